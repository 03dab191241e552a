// fused sweep / evaluate / init kernels for state dimension padded to 256 (the lanes kernel, four dimensions per lane; functional, not tuned)
#define PTM_DP 256
#include "ptm_sweep_inst.inc"
