// ptm_aux_kernels.hpp -- non-template kernels (exchange phase, verification hooks); included by ptm_engine.hip only.
#pragma once
#include "ptm_kernels.hpp"

namespace ptm {

// ------------------------------------------------------------------------------------------------
// exchange phase of parallel_tempering_chains::step (chain.cc:1410-1537), one wave per walker-ladder.
// Candidate draws are parallel over lanes; the in-order filter and trials (quirk Q6: later picks see the
// in-place updated view) run on lane 0 over LDS copies of the picked rungs' llikes.
// After the decisions the kernel exchanges the rows themselves, in place (whole contiguous rows), packs the rows that
// leave the shard and names the landing slot of arrivals; touch[] tells the sweep kernel which rungs skip their MH move.
// ------------------------------------------------------------------------------------------------
struct Decide {
  int DP, Nt, r0, nloc, W, Nc, ms;
  uint64_t seed, step;
  double thresh;              // (Ntemps-1)*swap_rate/maxswapsperstep (chain.cc:1413)
  const double* beta;         // [Nt]
  const double* ll_below;     // [W]      llike of rung r0-1 (top rung of the shard below), null on the first shard
  const double* ll_above;     // [H][W]   llike of rungs r1 .. r1+H-1 (bottom rungs of the shard above), null on the last
  int H;                      // halo depth actually available above (0 on the last shard)
  double* x;                  // [Nc][DP] rows, moved in place
  double* ll;
  double* lp;
  unsigned char* touch;
  int *arr_below, *arr_above;      // [W]
  long long *swap_try, *swap_acc;  // [W][Nt-1]
  int *last_pairs, *last_acc;      // [W][ms]
  double *send_up, *send_down;     // [W][DP+2] rows {x, llike, lprior} or null
  int* err;
};

// llike of global rung r for walker w, r inside the shard's window
__device__ __forceinline__ double win_llike(const Decide& p, int r, int w) {
  const int r1 = p.r0 + p.nloc;
  if (r < p.r0) return p.ll_below[w];
  if (r >= r1) return p.ll_above[(size_t)(r - r1) * p.W + w];
  return p.ll[(size_t)(r - p.r0) * p.W + w];
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
  unsigned short* owner = reinterpret_cast<unsigned short*>(mark + ((Nt + 1 + 7) & ~7));  // [Nt] pick that handles the rung
  unsigned short* inv = owner + ((Nt + 3) & ~3);                                         // [Nt] inverse of perm
  // (no static __shared__: it would precede the dynamic region and break its 16-byte base alignment)

  // window of rungs whose llike this shard knows: its own, one below, H above
  const int wlo = p.r0 - (p.ll_below ? 1 : 0), whi = p.r0 + p.nloc - 1 + p.H;
  for (int i = lane; i < Nt + 1; i += 64) mark[i] = 0;
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
    inv[n] = (unsigned short)n;
    inv[n + 1] = (unsigned short)(n + 1);
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
        // the row that leaves this shard downwards must be one of ours (else it crossed two boundaries in one step)
        if (i + 1 == p.r0 && (perm[i + 1] < p.r0 || perm[i + 1] >= p.r0 + p.nloc)) atomicOr(p.err, 1);
        const double t = llc[i]; llc[i] = llc[i + 1]; llc[i + 1] = t;
        const unsigned short s = perm[i]; perm[i] = perm[i + 1]; perm[i + 1] = s;
        accf[k] = 1;
      }
      tch[i] += 1;
      tch[i + 1] += 1;
    }
  }
  __syncthreads();
  // -- publish the step's log, counters and the touch counts of the local rungs; name one OWNER pick per touched rung
  const int r1 = p.r0 + p.nloc;
  const int DP = p.DP;
  for (int k = lane; k < ms; k += 64) {
    const int i = cand[k];
    p.last_pairs[(size_t)w * ms + k] = i;                 // -2: no candidate / dropped, -3: outside this shard's window
    p.last_acc[(size_t)w * ms + k] = accf[k];
    if (i < 0) continue;
    // swap_count / swap_accept_count (chain.cc:1498,1536): a pair is tried at most once per step, so no two lanes
    // of this wave (the only writer of walker w's counters) touch the same entry; a pair is counted by the shard that
    // owns its lower rung, so per-shard counters add up to the ladder's
    if (i >= p.r0 && i < r1) {
      p.swap_try[(size_t)w * (Nt - 1) + i] += 1;
      if (accf[k]) p.swap_acc[(size_t)w * (Nt - 1) + i] += 1;
    }
    for (int r = i; r <= i + 1; ++r) {
      if (r >= p.r0 && r < r1) p.touch[(r - p.r0) * p.W + w] = tch[r];
      owner[r] = (unsigned short)k;                       // any single winner will do
      const int s = perm[r];
      if (s != r) inv[s] = (unsigned short)r;             // the row that started at s ends at r
    }
  }
  __syncthreads();
  // -- move the rows IN PLACE.  The phase's net effect on the touched rungs is a permutation that decomposes into
  //    disjoint closed cycles inside the shard and at most two open paths through its boundaries (one row leaves, the
  //    others shift by one rung, an arrival fills the hole later).  One lane per cycle / path: nobody else touches
  //    those rows, so plain loads and stores in path order are safe.
  for (int k = lane; k < ms; k += 64) {
    const int i = cand[k];
    if (i < 0) continue;
    for (int r = i; r <= i + 1; ++r) {
      if (owner[r] != k || r < p.r0 || r >= r1 || perm[r] == r) continue;
      const int to = inv[r];                              // where rung r's old row goes
      const bool departs = to < p.r0 || to >= r1;
      bool head = departs;
      if (!departs) {                                     // closed cycle? then the lowest member leads
        head = true;
        int cc = perm[r], guard = 0;
        while (cc != r) {
          if (cc < p.r0 || cc >= r1 || cc < r || ++guard > Nt) { head = false; break; }  // open path or not the minimum
          cc = perm[cc];
        }
      }
      if (!head) continue;
      const size_t rowlen = DP;
      double* X = p.x;
      const int c0 = (r - p.r0) * p.W + w;
      double tmp[34];                                      // the head's old row {x[0..DP), llike, lprior}, DP <= 32
      for (int d = 0; d < DP; ++d) tmp[d] = X[(size_t)c0 * rowlen + d];
      tmp[DP] = p.ll[c0];
      tmp[DP + 1] = p.lp[c0];
      if (departs) {
        double* sb = (to >= r1) ? p.send_up : p.send_down;
        if (!sb) { atomicOr(p.err, 1); continue; }
        for (int d = 0; d < DP + 2; ++d) sb[(size_t)w * (DP + 2) + d] = tmp[d];
      }
      int cur = r;
      for (int guard = 0; guard <= Nt; ++guard) {
        const int src = perm[cur];
        const int cc = (cur - p.r0) * p.W + w;
        if (src == r) {                                    // closed the cycle: the saved row lands here
          for (int d = 0; d < DP; ++d) X[(size_t)cc * rowlen + d] = tmp[d];
          p.ll[cc] = tmp[DP];
          p.lp[cc] = tmp[DP + 1];
          break;
        }
        if (src < p.r0 || src >= r1) {                     // the hole: an arrival from the adjacent shard lands here
          (src >= r1 ? p.arr_above : p.arr_below)[w] = cc;
          break;
        }
        const int cs = (src - p.r0) * p.W + w;
        for (int d = 0; d < DP; ++d) X[(size_t)cc * rowlen + d] = X[(size_t)cs * rowlen + d];
        p.ll[cc] = p.ll[cs];
        p.lp[cc] = p.lp[cs];
        cur = src;
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
