// fused sweep / evaluate / init kernels for state dimension padded to 32
#define PTM_DP 32
#include "ptm_sweep_inst.inc"
