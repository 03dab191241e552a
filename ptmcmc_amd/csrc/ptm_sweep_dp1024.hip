// fused sweep / evaluate / init kernels for state dimension padded to 1024 (the lanes kernel, sixteen dimensions per lane; functional, not tuned)
#define PTM_DP 1024
#include "ptm_sweep_inst.inc"
