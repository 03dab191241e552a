// ptm_launch.hpp -- host-side launch entry points of the per-dimension translation units.
// The fused sweep kernel is instantiated for DP in {4,8,16,32,64,128,256,512,1024}; each DP lives in its own .hip file so the
// (large, fully unrolled) kernels compile in parallel.
#pragma once
#include <hip/hip_runtime.h>

#include "ptm_decide.hpp"
#include "ptm_kernels.hpp"
#include "ptm_ladder_args.hpp"

// the lanes kernel (ptm_lanes_kernel.hpp) takes launches of at most this many lanes (chains x padded dimension)
#define PTM_LANES_MAX (1ll << 20)

namespace ptm {
struct SweepSel {
  int kind;     // KIND_DENSE / KIND_DIAG / KIND_LOWER
  bool uni;     // W % 64 == 0: wave-uniform rung
  bool plain;   // open bounds, all-uniform prior, zero mean, no 1-D moves, no mixture, fixed ladder, device target
  bool simple;  // uni && plain
  bool lean_ev; // uni, and plain but for evolving ladders (per-chain temperatures): the lean MFMA build that reads them
  bool callback;  // host-callback likelihood (propose / accept passes): general VALU kernel only
  bool host_prop; // host-side proposals (ptm_set_proposal_callback): the lanes kernel's general build, whatever the population
  bool de;        // differential evolution drawn on the device (ptm_set_proposal_de): the general VALU kernel or the lanes kernel, not the MFMA kernels
};
#define PTM_DECL_DP(N)                                                                                              \
  hipError_t launch_sweep_##N(const Dev& p, SweepSel s, hipStream_t st);                                            \
  hipError_t launch_eval_##N(const Dev& p, int n, double* x, int* valid, double* lp, double* ll, int eval_like,     \
                             hipStream_t st);                                                                       \
  hipError_t launch_init_##N(const Dev& p, double* x, double* ll, double* lp, int* fail, long long cb_attempt,          \
                             unsigned char* pending, hipStream_t st);
// small ladders: nsteps whole PT steps per launch, one block per walker-ladder (ptm_fused_kernel.hpp; built for DP <= 16)
#define PTM_DECL_FUSED(N) \
  hipError_t launch_fused_##N(const Dev& p, const Decide& d, bool diag, int nsteps, int* swap_log_base, int log_head, size_t decide_lds, hipStream_t st);
PTM_DECL_FUSED(4)
PTM_DECL_FUSED(8)
PTM_DECL_FUSED(16)
#undef PTM_DECL_FUSED
// long ladders of few walkers: many PT steps per launch on a grid of resident workgroups (ptm_ladder_kernel.hpp; DP 16 and 32).
// ladder_blocks_N: how many of its workgroups the device holds at once (0: the kernel cannot run); launch_ladder_N: the launch
#define PTM_DECL_LADDER(N)                                  \
  size_t ladder_lds_##N(int Nt, int ms, bool ev);           \
  int ladder_blocks_##N(bool diag, int fl, size_t lds);     \
  hipError_t launch_ladder_##N(const Dev& p, const LadderArgs& a, bool diag, int fl, int grid, size_t lds, hipStream_t st);
PTM_DECL_LADDER(4)
PTM_DECL_LADDER(8)
PTM_DECL_LADDER(16)
PTM_DECL_LADDER(32)
#undef PTM_DECL_LADDER
PTM_DECL_DP(4)
PTM_DECL_DP(8)
PTM_DECL_DP(16)
PTM_DECL_DP(32)
PTM_DECL_DP(64)
PTM_DECL_DP(128)
PTM_DECL_DP(256)
PTM_DECL_DP(512)
PTM_DECL_DP(1024)
#undef PTM_DECL_DP
}  // namespace ptm
