// fused sweep / evaluate / init kernels for state dimension padded to 16
#define PTM_DP 16
#include "ptm_sweep_inst.inc"
