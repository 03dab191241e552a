// fused sweep / evaluate / init kernels for state dimension padded to 64
#define PTM_DP 64
#include "ptm_sweep_inst.inc"
