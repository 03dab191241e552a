// fused sweep / evaluate / init kernels for state dimension padded to 4
#define PTM_DP 4
#include "ptm_sweep_inst.inc"
