// fused sweep / evaluate / init kernels for state dimension padded to 128 (65..128 dimensions: the lanes kernel, two dimensions per lane)
#define PTM_DP 128
#include "ptm_sweep_inst.inc"
