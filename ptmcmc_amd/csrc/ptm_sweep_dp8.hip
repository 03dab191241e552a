// fused sweep / evaluate / init kernels for state dimension padded to 8
#define PTM_DP 8
#include "ptm_sweep_inst.inc"
