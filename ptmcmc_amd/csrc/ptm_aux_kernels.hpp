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
  int *mv_src, *mv_dst, *mv_n;     // [W][MVCAP], [W][MVCAP], [W]: the ladder's row moves for move_kernel
  int* err;
};

// llike of global rung r for walker w, r inside the shard's window
__device__ __forceinline__ double win_llike(const Decide& p, int r, int w) {
  const int r1 = p.r0 + p.nloc;
  if (r < p.r0) return p.ll_below[w];
  if (r >= r1) return p.ll_above[(size_t)(r - r1) * p.W + w];
  return p.ll[(size_t)(r - p.r0) * p.W + w];
}

constexpr int MVCAP = 256;  // rows one ladder can move per step on the register path (move_kernel)

// The reference decides the candidates strictly in pick order (chain.cc:1410-1537).  Two facts make that order
// parallel over the ladder without changing any outcome:
//   (1) filter: a pick n is dropped iff an earlier SURVIVING pick is n or n-1 (chain.cc:1417-1418).  Only the first
//       pick of a rung value can survive, and alive[n] = !(alive[n-1] && first[n-1] < first[n]): a recurrence along
//       RUNS of consecutive picked rungs, independent between runs;
//   (2) trials: two surviving picks on adjacent rungs (n, n+1) exist only if n+1 was picked first, and only then does
//       pick n see an updated upper rung -- so each run of consecutive surviving rungs is decided top-down, and runs
//       are independent of each other.
// One lane walks each run; everything else (draws, logs, counters, move list) is parallel over picks.
__global__ __launch_bounds__(64) void decide_kernel(const Decide p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int w = blockIdx.x;
  const int lane = threadIdx.x;
  const int Nt = p.Nt, ms = p.ms;
  const int NONE = 0x7fffffff;
  // LDS carve (all offsets multiples of 8)
  double* llc = reinterpret_cast<double*>(smem);                              // [Nt]   llike view of the touched rungs
  double* lu = llc + Nt;                                                      // [ms]   log(u_accept) per candidate
  int* first = reinterpret_cast<int*>(lu + ms);                               // [Nt]   first pick of each rung value
  int* cand = first + ((Nt + 1) & ~1);                                        // [ms]   rung of the pick / -2 none or dropped
  int* mvsrc = cand + ((ms + 1) & ~1);                                        // [MVCAP]
  int* mvdst = mvsrc + MVCAP;                                                 // [MVCAP]
  int* mvcnt = mvdst + MVCAP;                                                 // [2]
  unsigned short* perm = reinterpret_cast<unsigned short*>(mvcnt + 2);        // [Nt]   source rung of the row now at a rung
  unsigned short* inv = perm + ((Nt + 3) & ~3);                               // [Nt]   inverse of perm
  unsigned char* alive = reinterpret_cast<unsigned char*>(inv + ((Nt + 3) & ~3));  // [ms] 0 dropped, 1 survives and is ours,
                                                                                   //      2 survives, not ours to decide
  unsigned char* accf = alive + ((ms + 7) & ~7);                              // [ms]

  const int r1 = p.r0 + p.nloc;
  // window of rungs whose llike this shard knows: its own, one below, H above
  const int wlo = p.r0 - (p.ll_below ? 1 : 0), whi = r1 - 1 + p.H;
  cdp beta = as_c(p.beta);

  for (int i = lane; i < Nt; i += 64) first[i] = NONE;
  if (lane == 0) *mvcnt = 0;
  __syncthreads();
  // -- candidate draws (chain.cc:1410-1416): block k of the ladder stream gives {u_try, u_pick, u_accept}
  for (int k = lane; k < ms; k += 64) {
    const u32x4 o = draw_block(p.seed, TAG_PT, (uint32_t)w, p.step, (uint32_t)k);
    int n = -2;
    if (Nt > 1 && u01(o.v0) < p.thresh) n = (int)(u01(o.v1) * (Nt - 1));
    cand[k] = n;
    lu[k] = dlog_u01(o.v2);  // the accept uniform's slot is reserved whether or not it is needed (cf. Q5)
    alive[k] = 0;
    accf[k] = 0;
    if (n >= 0) atomicMin(&first[n], k);
  }
  __syncthreads();
  // -- filter (1): run heads walk their run upwards
  for (int k = lane; k < ms; k += 64) {
    const int n = cand[k];
    if (n < 0 || first[n] != k) continue;                      // repeated rung value: dropped
    if (n > 0 && first[n - 1] != NONE) continue;               // not a run head
    bool a = true;
    for (int m = n;; ++m) {
      alive[first[m]] = a ? 1 : 0;
      if (m + 1 > Nt - 2 || first[m + 1] == NONE) break;
      a = !(a && first[m] < first[m + 1]);
    }
  }
  __syncthreads();
#define PTM_ALIVE_RUNG(r) ((r) >= 0 && (r) <= Nt - 2 && first[(r)] != NONE && alive[first[(r)]])
  // -- working copy of the touched rungs inside the window (gather_llikes, chain.cc:1434); each touched rung is set up
  //    by exactly one lane: the pick whose lower rung it is, or -- for the top of a run -- the pick just below it
  for (int k = lane; k < ms; k += 64) {
    const int n = cand[k];
    if (n < 0) continue;
    if (!alive[k]) { cand[k] = -2; continue; }
    if (n < wlo || n + 1 > whi) { alive[k] = 2; continue; }     // survives, but is not this shard's to decide
    llc[n] = win_llike(p, n, w);
    perm[n] = (unsigned short)n;
    inv[n] = (unsigned short)n;
    if (!PTM_ALIVE_RUNG(n + 1) || n + 2 > whi) {               // (an alive pick above that lies outside the window sets up nothing)
      llc[n + 1] = win_llike(p, n + 1, w);
      perm[n + 1] = (unsigned short)(n + 1);
      inv[n + 1] = (unsigned short)(n + 1);
    }
  }
  __syncthreads();
  // -- trials (2): the top pick of each run of surviving rungs walks the run downwards (chain.cc:1436-1537)
  for (int k = lane; k < ms; k += 64) {
    const int n = cand[k];
    if (n < 0 || PTM_ALIVE_RUNG(n + 1)) continue;               // not the top of a run
    bool taint = false;  // the upper rung's content is unknown: an exchange above the window may have replaced it
    for (int i = n; i >= 0; --i) {
      const int kk = first[i];
      if (i < wlo) break;                                        // below the window: nothing further down concerns us
      if (i + 1 > whi) {                                         // pair above the window: not decided here ...
        taint = (i == whi);                                      // ... but the one right above it may replace its top rung
      } else if (taint) {
        if (i + 1 <= r1) atomicOr(p.err, 2);                     // would decide a local / straddling exchange blindly
        alive[kk] = 2;
      } else {
        double lla = llc[i];
        if (!(lla > -1e200)) lla = -1e200;
        double llb = llc[i + 1];
        if (!(llb > -1e200)) llb = -1e200;
        const double logH = -(beta[i + 1] - beta[i]) * (llb - lla);
        bool acc = true;
        if (logH < 0) acc = lu[kk] < logH;
        if (acc) {
          // the row that leaves this shard downwards must be one of ours (else it crossed two boundaries in one step)
          if (i + 1 == p.r0 && (perm[i + 1] < p.r0 || perm[i + 1] >= r1)) atomicOr(p.err, 1);
          const double t = llc[i]; llc[i] = llc[i + 1]; llc[i + 1] = t;
          const unsigned short s = perm[i]; perm[i] = perm[i + 1]; perm[i + 1] = s;
          accf[kk] = 1;
        }
      }
      if (!PTM_ALIVE_RUNG(i - 1)) break;
    }
  }
  __syncthreads();
  // -- publish the step's log, counters, the touch counts of the local rungs and the inverse permutation
  const int DP = p.DP;
  for (int k = lane; k < ms; k += 64) {
    const int i = alive[k] == 1 ? cand[k] : (alive[k] == 2 ? -3 : -2);
    p.last_pairs[(size_t)w * ms + k] = i;                 // -2: no candidate / dropped, -3: not this shard's to decide
    p.last_acc[(size_t)w * ms + k] = accf[k];
    if (i < 0) continue;
    // swap_count / swap_accept_count (chain.cc:1498,1536); a pair is counted by the shard that owns its lower rung, so
    // per-shard counters add up to the ladder's
    if (i >= p.r0 && i < r1) {
      p.swap_try[(size_t)w * (Nt - 1) + i] += 1;
      if (accf[k]) p.swap_acc[(size_t)w * (Nt - 1) + i] += 1;
    }
    const int rtop = (PTM_ALIVE_RUNG(i + 1) && alive[first[i + 1]] == 1) ? i : i + 1;  // rung i+1 belongs to the pick above, if ours
    for (int r = i; r <= rtop; ++r) {
      if (r >= p.r0 && r < r1) {
        const int below_alive = (r > 0 && PTM_ALIVE_RUNG(r - 1) && alive[first[r - 1]] == 1) ? 1 : 0;
        const int self_alive = (r == i) ? 1 : 0;
        p.touch[(r - p.r0) * p.W + w] = (unsigned char)(below_alive + self_alive);
      }
      const int s = perm[r];
      if (s != r) inv[s] = (unsigned short)r;             // the row that started at s ends at r
    }
  }
  __syncthreads();
  // -- the row moves.  The phase's net effect on the touched rungs is a permutation of rows: new row[r] =
  //    old row[perm[r]].  List every move (source slot -> destination slot, or -> send buffer for a row that leaves
  //    the shard) for move_kernel; the hole an arrival will fill is named in arr_above / arr_below.
  for (int k = lane; k < ms; k += 64) {
    if (alive[k] != 1) continue;
    const int i = cand[k];
    const int rtop = (PTM_ALIVE_RUNG(i + 1) && alive[first[i + 1]] == 1) ? i : i + 1;
    for (int r = i; r <= rtop; ++r) {
      if (r < p.r0 || r >= r1 || perm[r] == r) continue;
      const int s = perm[r], to = inv[r];
      const int cr = (r - p.r0) * p.W + w;
      if (s >= p.r0 && s < r1) {                            // local -> local
        const int j = atomicAdd(mvcnt, 1);
        if (j < MVCAP) { mvsrc[j] = (s - p.r0) * p.W + w; mvdst[j] = cr; }
      } else {
        (s >= r1 ? p.arr_above : p.arr_below)[w] = cr;       // the hole: an arrival from the adjacent shard lands here
      }
      if (to < p.r0 || to >= r1) {                           // rung r's old row leaves the shard
        const int j = atomicAdd(mvcnt, 1);
        if (j < MVCAP) { mvsrc[j] = cr; mvdst[j] = (to >= r1) ? -1 : -2; }
        if (!((to >= r1) ? p.send_up : p.send_down)) atomicOr(p.err, 1);
      }
    }
  }
  __syncthreads();
  const int nmv = *mvcnt;
  if (nmv > MVCAP) {
    // rare overflow of the register path (more than MVCAP moved rows in one ladder and step): the permutation
    // decomposes into disjoint closed cycles inside the shard and at most two open paths through its boundaries;
    // one lane walks each in path order, so plain loads and stores are safe.  Slow, correct.
    for (int k = lane; k < ms; k += 64) {
      if (alive[k] != 1) continue;
      const int i = cand[k];
      const int rtop = (PTM_ALIVE_RUNG(i + 1) && alive[first[i + 1]] == 1) ? i : i + 1;
      for (int r = i; r <= rtop; ++r) {
        if (r < p.r0 || r >= r1 || perm[r] == r) continue;
        const int to = inv[r];
        const bool departs = to < p.r0 || to >= r1;
        bool head = departs;
        if (!departs) {                                     // closed cycle? then the lowest member leads
          head = true;
          int cc = perm[r], guard = 0;
          while (cc != r) {
            if (cc < p.r0 || cc >= r1 || cc < r || ++guard > Nt) { head = false; break; }
            cc = perm[cc];
          }
        }
        if (!head) continue;
        double* X = p.x;
        const int c0 = (r - p.r0) * p.W + w;
        double tmp[34];                                      // the head's old row {x[0..DP), llike, lprior}, DP <= 32
        for (int d = 0; d < DP; ++d) tmp[d] = X[(size_t)c0 * DP + d];
        tmp[DP] = p.ll[c0];
        tmp[DP + 1] = p.lp[c0];
        if (departs) {
          double* sb = (to >= r1) ? p.send_up : p.send_down;
          if (!sb) continue;
          for (int d = 0; d < DP + 2; ++d) sb[(size_t)w * (DP + 2) + d] = tmp[d];
        }
        int cur = r;
        for (int guard = 0; guard <= Nt; ++guard) {
          const int src = perm[cur];
          const int cc = (cur - p.r0) * p.W + w;
          if (src == r) {
            for (int d = 0; d < DP; ++d) X[(size_t)cc * DP + d] = tmp[d];
            p.ll[cc] = tmp[DP];
            p.lp[cc] = tmp[DP + 1];
            break;
          }
          if (src < p.r0 || src >= r1) break;                // the hole (named above)
          const int cs = (src - p.r0) * p.W + w;
          for (int d = 0; d < DP; ++d) X[(size_t)cc * DP + d] = X[(size_t)cs * DP + d];
          p.ll[cc] = p.ll[cs];
          p.lp[cc] = p.lp[cs];
          cur = src;
        }
      }
    }
    if (lane == 0) p.mv_n[w] = 0;
    return;
  }
  // hand the list to move_kernel (a register-heavy gather/scatter that would cost this kernel its occupancy)
  int* gs = p.mv_src + (size_t)w * MVCAP;
  int* gd = p.mv_dst + (size_t)w * MVCAP;
  for (int j = lane; j < nmv; j += 64) { gs[j] = mvsrc[j]; gd[j] = mvdst[j]; }
  if (lane == 0) p.mv_n[w] = nmv;
#undef PTM_ALIVE_RUNG
}

// ------------------------------------------------------------------------------------------------
// applies one ladder's row moves IN PLACE, one wave per ladder: GATHER every moved row into registers (16 lanes x 16 B =
// one 256-B row per quarter wave, four rows per load instruction, up to MVCAP rows), wait for all loads, then SCATTER.
// With every read finished before the first write no ordering between the moves is needed (the moves of one ladder
// form cycles over its own rows; other ladders' rows are never touched).
// ------------------------------------------------------------------------------------------------
struct Move {
  int DP, W;
  double *x, *ll, *lp;
  double *send_up, *send_down;
  const int *mv_src, *mv_dst;
  int* mv_n;
};
typedef double d2_t __attribute__((ext_vector_type(2)));  // (HIP's double2 struct does not stay in registers as an array)

__global__ __launch_bounds__(64, 1) void move_kernel(const Move p) {
  __shared__ int s_src[MVCAP], s_dst[MVCAP];
  const int w = blockIdx.x, lane = threadIdx.x;
  const int nmv = p.mv_n[w];
  if (nmv <= 0) return;
  const int DP = p.DP;
  // the list goes through LDS so that the row gathers below are not chained behind index loads from memory
  for (int j = lane; j < MVCAP; j += 64) {
    const bool in = j < nmv;
    s_src[j] = in ? p.mv_src[(size_t)w * MVCAP + j] : 0;
    s_dst[j] = in ? p.mv_dst[(size_t)w * MVCAP + j] : -3;
  }
  __syncthreads();
  const int g = lane >> 4, sub = lane & 15;
  const bool act = 2 * sub < DP;          // DP/2 lanes of 16 carry a row (16 B each)
  const int col = act ? 2 * sub : 0;      // idle lanes re-read column 0 (harmless) so that no load is predicated
  d2_t v[MVCAP / 4];
  double sl[MVCAP / 64], sp[MVCAP / 64];
#pragma unroll
  for (int q = 0; q < MVCAP / 4; ++q)     // unconditional loads: entries past the list read row 0 and are never stored
    v[q] = *reinterpret_cast<const d2_t*>(p.x + (size_t)s_src[4 * q + g] * DP + col);
#pragma unroll
  for (int q = 0; q < MVCAP / 64; ++q) {
    sl[q] = p.ll[s_src[64 * q + lane]];
    sp[q] = p.lp[s_src[64 * q + lane]];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every gather has landed before the first scatter
#pragma unroll
  for (int q = 0; q < MVCAP / 4; ++q) {
    const int d = s_dst[4 * q + g];
    if (d != -3 && act) {
      double* dstp = d >= 0 ? p.x + (size_t)d * DP : (d == -1 ? p.send_up : p.send_down) + (size_t)w * (DP + 2);
      *reinterpret_cast<d2_t*>(dstp + col) = v[q];
    }
  }
#pragma unroll
  for (int q = 0; q < MVCAP / 64; ++q) {
    const int d = s_dst[64 * q + lane];
    if (d >= 0) { p.ll[d] = sl[q]; p.lp[d] = sp[q]; }
    else if (d != -3) { double* sb = (d == -1 ? p.send_up : p.send_down) + (size_t)w * (DP + 2); sb[DP] = sl[q]; sb[DP + 1] = sp[q]; }
  }
  if (lane == 0) p.mv_n[w] = 0;
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
