// fused sweep / evaluate / init kernels for state dimension padded to 512 (the lanes kernel, eight dimensions per lane; functional, not tuned)
#define PTM_DP 512
#include "ptm_sweep_inst.inc"
