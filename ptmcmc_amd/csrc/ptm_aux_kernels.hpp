// ptm_aux_kernels.hpp -- non-template kernels (exchange phase, verification hooks); included by ptm_engine.hip only.
#pragma once
#include "ptm_kernels.hpp"

namespace ptm {

// ------------------------------------------------------------------------------------------------
// exchange phase of parallel_tempering_chains::step (chain.cc:1410-1537), one wave per walker-ladder.
// Candidate draws are parallel over lanes; the in-order filter and trials (quirk Q6: later picks see the
// in-place updated view) run on lane 0 over LDS copies of the picked rungs' llikes.
// The kernel moves no state: it names, for every local row that took part, the slot it moves to (dst[]) and the
// number of add_state calls its rung received (touch[]); the sweep kernel does the move.
// ------------------------------------------------------------------------------------------------
struct Decide {
  int DP, Nt, r0, nloc, W, Nc, ms;
  uint64_t seed, step;
  double thresh;              // (Ntemps-1)*swap_rate/maxswapsperstep (chain.cc:1413)
  const double* beta;         // [Nt]
  const double* ll_below;     // [W]      llike of rung r0-1 (top rung of the shard below), null on the first shard
  const double* ll_above;     // [H][W]   llike of rungs r1 .. r1+H-1 (bottom rungs of the shard above), null on the last
  int H;                      // halo depth actually available above (0 on the last shard)
  const double* x_in;         // local state planes (for packing departures)
  const double* ll_in;
  const double* lp_in;
  int* dst;
  unsigned char* touch;
  int *arr_below, *arr_above;      // [W]
  long long *swap_try, *swap_acc;  // [W][Nt-1]
  int *last_pairs, *last_acc;      // [W][ms]
  double *send_up, *send_down;     // [(DP+2)][W] or null
  int* err;
};

// llike of global rung r for walker w, r inside the shard's window
__device__ __forceinline__ double win_llike(const Decide& p, int r, int w) {
  const int r1 = p.r0 + p.nloc;
  if (r < p.r0) return p.ll_below[w];
  if (r >= r1) return p.ll_above[(size_t)(r - r1) * p.W + w];
  return p.ll_in[(size_t)(r - p.r0) * p.W + w];
}

__global__ __launch_bounds__(64) void decide_kernel(const Decide p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int w = blockIdx.x;
  const int lane = threadIdx.x;
  const int Nt = p.Nt, ms = p.ms;
  // LDS carve (all offsets multiples of 8)
  double* llc = reinterpret_cast<double*>(smem);                              // [Nt]   llike view of the picked rungs
  double* lu = llc + Nt;                                                      // [ms]   log(u_accept) per candidate
  double* db = lu + ms;                                                       // [ms]   beta[i+1]-beta[i] per candidate
  int* cand = reinterpret_cast<int*>(db + ms);                                // [ms]
  int* accf = cand + ((ms + 1) & ~1);                                         // [ms]
  unsigned short* perm = reinterpret_cast<unsigned short*>(accf + ((ms + 1) & ~1));  // [Nt]
  unsigned char* tch = reinterpret_cast<unsigned char*>(perm + ((Nt + 3) & ~3));     // [Nt]
  unsigned char* mark = tch + ((Nt + 7) & ~7);                                       // [Nt+1]
  int* down_src_p = reinterpret_cast<int*>(mark + ((Nt + 1 + 7) & ~7));              // [1]
  // (no static __shared__: it would precede the dynamic region and break its 16-byte base alignment)

  // window of rungs whose llike this shard knows: its own, one below, H above
  const int wlo = p.r0 - (p.ll_below ? 1 : 0), whi = p.r0 + p.nloc - 1 + p.H;
  for (int i = lane; i < Nt + 1; i += 64) mark[i] = 0;
  if (lane == 0) *down_src_p = -1;
  // -- candidate draws (chain.cc:1410-1416): block k of the ladder stream gives {u_try, u_pick, u_accept}
  for (int k = lane; k < ms; k += 64) {
    const u32x4 o = draw_block(p.seed, TAG_PT, (uint32_t)w, p.step, (uint32_t)k);
    int n = -2;
    if (Nt > 1 && u01(o.v0) < p.thresh) n = (int)(u01(o.v1) * (Nt - 1));
    cand[k] = n;
    lu[k] = dlog_u01(o.v2);  // the accept uniform's slot is reserved whether or not it is needed (cf. Q5)
    accf[k] = 0;
  }
  __syncthreads();
  // -- drop a pick equal to, or one above, an earlier surviving pick (chain.cc:1417-1418)
  if (lane == 0) {
    for (int k = 0; k < ms; ++k) {
      const int n = cand[k];
      if (n < 0) continue;
      if (mark[n]) cand[k] = -2;
      else { mark[n] = 1; mark[n + 1] = 1; }
    }
  }
  __syncthreads();
  // -- working copy of the picked rungs (gather_llikes, chain.cc:1434)
  for (int k = lane; k < ms; k += 64) {
    const int n = cand[k];
    if (n < 0) continue;
    // only pairs inside the window [r0-1, r1+H) can concern this shard (exchanges propagate downwards only)
    if (n < wlo || n + 1 > whi) continue;
    llc[n] = win_llike(p, n, w);
    llc[n + 1] = win_llike(p, n + 1, w);
    db[k] = p.beta[n + 1] - p.beta[n];
    perm[n] = (unsigned short)n;
    perm[n + 1] = (unsigned short)(n + 1);
    tch[n] = 0;
    tch[n + 1] = 0;
  }
  __syncthreads();
  // -- trials in pick order (chain.cc:1436-1537)
  if (lane == 0) {
    // `taint`: lowest rung (>= r1) whose content is unknown because an exchange above the window may have changed it;
    // it moves down with every later pick right below it.  Reaching the shard boundary means the halo was too shallow.
    int taint = Nt + 1;
    const int r1s = p.r0 + p.nloc;
    for (int k = 0; k < ms; ++k) {
      const int i = cand[k];
      if (i < 0) continue;
      if (i + 1 > whi || i < wlo) {                       // pair outside the window
        if (i == whi && i + 1 < Nt) taint = i;            // ... but it may have replaced the window's top rung
        cand[k] = -3;                                     // (not logged as a local decision; -3 = "not ours")
        continue;
      }
      if (i + 1 >= taint) {                               // upper rung's content unknown
        if (i + 1 <= r1s) atomicOr(p.err, 2);             // would decide a local / straddling exchange blindly
        if (i < taint) taint = i;
        cand[k] = -3;
        continue;
      }
      double lla = llc[i];
      if (!(lla > -1e200)) lla = -1e200;
      double llb = llc[i + 1];
      if (!(llb > -1e200)) llb = -1e200;
      const double logH = -db[k] * (llb - lla);
      bool acc = true;
      if (logH < 0) acc = lu[k] < logH;
      if (acc) {
        if (i + 1 == p.r0) *down_src_p = perm[i + 1];  // the row that leaves this shard downwards
        const double t = llc[i]; llc[i] = llc[i + 1]; llc[i + 1] = t;
        const unsigned short s = perm[i]; perm[i] = perm[i + 1]; perm[i + 1] = s;
        accf[k] = 1;
      }
      tch[i] += 1;
      tch[i + 1] += 1;
    }
  }
  __syncthreads();
  // -- publish: hand-off arrays for local rungs, the step's log, departures
  const int r1 = p.r0 + p.nloc;
  const int DP = p.DP;
  for (int k = lane; k < ms; k += 64) {
    const int i = cand[k];
    p.last_pairs[(size_t)w * ms + k] = i;                 // -2: no candidate / dropped, -3: outside this shard's window
    p.last_acc[(size_t)w * ms + k] = accf[k];
    if (i < 0) continue;
    // swap_count / swap_accept_count (chain.cc:1498,1536): a pair is tried at most once per step, so no two lanes
    // of this wave (the only writer of walker w's counters) touch the same entry
    // (a pair is counted by the shard that owns its lower rung, so per-shard counters add up to the ladder's)
    if (i >= p.r0 && i < r1) {
      p.swap_try[(size_t)w * (Nt - 1) + i] += 1;
      if (accf[k]) p.swap_acc[(size_t)w * (Nt - 1) + i] += 1;
    }
    for (int r = i; r <= i + 1; ++r) {
      // rung r ends the phase holding the row that started the step at rung s = perm[r]: publish the move from the
      // row's point of view (dst of the source slot) -- or, for a row coming from another shard, its landing slot
      const int s = perm[r];
      const bool r_local = r >= p.r0 && r < r1, s_local = s >= p.r0 && s < r1;
      if (r_local) p.touch[(r - p.r0) * p.W + w] = tch[r];
      if (s_local) p.dst[(s - p.r0) * p.W + w] = r_local ? (r - p.r0) * p.W + w : DST_GONE;
      else if (r_local) (s >= r1 ? p.arr_above : p.arr_below)[w] = (r - p.r0) * p.W + w;
    }
    if (accf[k] && i + 1 == r1 && r1 < Nt && p.send_up) {
      // exchange across the upper shard boundary: our top rung's row (always its start-of-step content) goes up
      const int cs = (i - p.r0) * p.W + w;
      for (int d = 0; d < DP; ++d) p.send_up[(size_t)d * p.W + w] = p.x_in[(size_t)d * p.Nc + cs];
      p.send_up[(size_t)DP * p.W + w] = p.ll_in[cs];
      p.send_up[(size_t)(DP + 1) * p.W + w] = p.lp_in[cs];
    }
    if (accf[k] && i + 1 == p.r0 && p.send_down) {
      const int s = *down_src_p;
      if (s < p.r0 || s >= r1) {
        atomicOr(p.err, 1);  // the departing row is not ours: it crossed two boundaries in one step
      } else {
        const int cs = (s - p.r0) * p.W + w;
        for (int d = 0; d < DP; ++d) p.send_down[(size_t)d * p.W + w] = p.x_in[(size_t)d * p.Nc + cs];
        p.send_down[(size_t)DP * p.W + w] = p.ll_in[cs];
        p.send_down[(size_t)(DP + 1) * p.W + w] = p.lp_in[cs];
      }
    }
  }
}

// verification hooks
__global__ void debug_eval_kernel(int fn, const double* a, const double* b, double* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double r = 0;
  switch (fn) {
    case 0: r = dlog(a[i]); break;
    case 1: r = dexp(a[i]); break;
    case 2: r = dsin_0_pi(a[i]); break;
    case 3: r = dcos_hpi(a[i]); break;
    case 4: r = dsqrt(a[i]); break;
    case 5: r = a[i] / b[i]; break;
    case 6: r = __builtin_sqrt(a[i]); break;  // raw compiler expansion (to measure how often the fix-up fires)
  }
  out[i] = r;
}
__global__ void debug_philox_kernel(uint64_t seed, int tag, uint32_t stream, uint64_t step, uint32_t block, uint32_t* out) {
  const u32x4 o = draw_block(seed, tag, stream, step, block);
  out[0] = o.v0; out[1] = o.v1; out[2] = o.v2; out[3] = o.v3;
}
__global__ void debug_boxmuller_kernel(const uint32_t* k1, const uint32_t* k2, double* z0, double* z1, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  boxmuller(k1[i], k2[i], z0[i], z1[i]);
}
// exhaustive scan over all 2^32 first arguments of Box-Muller: counts the k for which the raw sqrt expansion
// differs from the corrected one (i.e. is not already correctly rounded) on r = sqrt(-2 log((k+.5)/2^32))
__global__ void debug_sqrt_scan_kernel(unsigned long long* mismatches) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t stride = gridDim.x * blockDim.x;
  unsigned long long bad = 0;
  for (uint64_t k = tid; k < (1ull << 32); k += stride) {
    const double a = -2.0 * dlog_u01((uint32_t)k);
    if (__builtin_sqrt(a) != dsqrt(a)) bad++;
  }
  if (bad) atomicAdd(mismatches, bad);
}

}  // namespace ptm
