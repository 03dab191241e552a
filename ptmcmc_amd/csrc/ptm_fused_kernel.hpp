// ptm_fused_kernel.hpp -- SMALL LADDERS: K whole parallel_tempering_chains::step calls per launch.
//
// A ladder whose rungs x padded dimensions fit 256 lanes of ONE workgroup (BASELINE configs[0] 8 rungs x 4; a sampler's
// 20-rung ladder of 6 parameters) is all latency: its step is an exchange block of ten
// dependent phases and a sweep of a few waves, two launches of 5-10 us each however little they compute.  Here one block per
// walker-ladder keeps the tables in LDS and loops over the steps: exchange phase (decide_body, the very code of
// decide_kernel) -> barrier -> one MH_chain::step per rung with a lane per dimension (lanes_body, the very code of
// sweep_lanes_kernel) -> barrier.  Walkers are independent ladders, so no barrier wider than the block is needed (SURVEY 7,
// step 5: "K sweeps per launch").  Chains are bit-identical to the two-launch path by construction: same functions, same
// random streams (keyed by walker, rung, step).
#pragma once
#include "ptm_decide.hpp"
#include "ptm_lanes_kernel.hpp"

namespace ptm {

// FUSED_THREADS: 64 or 256 -- the smaller that gives every (rung, dimension) of the ladder a lane (a barrier of one wave
// costs nothing).  (1024 threads -- BASELINE configs[1], 64 rungs x 16 -- were tried: 128 registers per lane do not hold the two
// bodies and what is hoisted out of the step loop, ~100 spill, and the step is no faster than two launches.)
template <int DP, int KIND, int FUSED_THREADS>
__global__ __launch_bounds__(FUSED_THREADS) void ladder_steps_kernel(const Dev p, const Decide d, int nsteps, int* swap_log_base, int log_head) {
  extern __shared__ __attribute__((aligned(16))) double lds_all[];
  const int w = blockIdx.x;                                   // this block's ladder
  unsigned char* dsm = reinterpret_cast<unsigned char*>(lds_all + ((lanes_lds_doubles<DP>(FUSED_THREADS / 64) + 1) & ~1));
  lanes_stage<DP>(p, lds_all);                                // Box-Muller tables and the precision matrix: once per launch
  const uint64_t step0 = p.step;
  const size_t logslot = (size_t)p.W * d.ms;
  for (int s = 0; s < nsteps; ++s) {
    // (the parameter blocks stay untouched -- scalar registers: written to, they would be copied into ~280 vector registers)
    const uint64_t step = step0 + (uint64_t)s;
    int* const swap_log = swap_log_base + (size_t)((log_head + s) % PTM_LOG_RING) * logslot;
    const int wv = w;
    if (p.Nt > 1) decide_body<FUSED_THREADS>(d, dsm, wv, step, swap_log);
    __syncthreads();                                          // rows, llikes and touch flags of the exchange phase: visible to the block
    lanes_body<DP, KIND, true>(p, lds_all, 0, wv, p.W, p.Nt, step);  // chain (rung k, walker w) = k * W + w
    __syncthreads();
  }
}

}  // namespace ptm
