// ptm_decide.hpp -- the exchange phase's decision kernel (template: included by the engine unit through ptm_aux_kernels.hpp and
// by the per-dimension units through ptm_fused_kernel.hpp).
#pragma once
#include "ptm_kernels.hpp"

namespace ptm {

// ------------------------------------------------------------------------------------------------
// exchange phase of parallel_tempering_chains::step (chain.cc:1410-1537), one block per walker-ladder.
// decide_kernel replays the step's candidate draws, decides every exchange that concerns the shard and lists the row
// moves; move_kernel applies them in place (whole contiguous rows) and packs the rows that leave the shard;
// install_kernel lands the rows that arrive from the adjacent shards.  touch[] tells the sweep kernel which rungs skip
// their MH move.
//
// Boundary message (one per direction and step): an int32 row count in the first 8 bytes (+8 bytes padding), then
// `row_cap` slots of RD = DP+4 doubles {x[DP], llike, lprior, walker, 0}.  Rows are appended in no particular order.
// ------------------------------------------------------------------------------------------------
constexpr int MSG_HDR = 2;    // doubles before the first row slot
constexpr int ROW_EXTRA = 4;  // {llike, lprior, walker, pad}: keeps row slots 32-byte multiples

// claims the next slot of a boundary message; null (and error bit 4) when the message is full
__device__ __forceinline__ double* claim_row(double* buf, int cap, int RD, int* err) {
  const int slot = atomicAdd(reinterpret_cast<int*>(buf), 1);
  if (slot >= cap) { atomicOr(err, 4); return nullptr; }
  return buf + MSG_HDR + (size_t)slot * RD;
}

struct Decide {
  int DP, Nt, r0, nloc, W, Nc, ms;
  int w_off;                  // global index of local walker 0: the ladder streams are keyed by the global walker
  uint64_t seed, step;
  double thresh;              // (Ntemps-1)*swap_rate/maxswapsperstep (chain.cc:1413)
  const double* beta;         // [Nt]
  const double* ll_below;     // [W]      llike of rung r0-1 (top rung of the shard below), null on the first shard
  const double* ll_above;     // [H][W]   llike of rungs r1 .. r1+H-1 (bottom rungs of the shard above), null on the last
  int H;                      // halo depth actually available above (0 on the last shard)
  // The whole ladder's llikes (and lpriors) [Nt][W], gathered from all shards by the caller -- what the reference's MPI ranks
  // have after gather_llikes / gather_lposts (chain.cc:1433-1435,1950-1972).  With them the window is the ladder: no halo, no
  // run can reach past it, and an EVOLVING ladder (whose every accepted exchange renormalises all gaps, so that every later trial
  // of the step depends on it) can be replayed exactly on every shard.  null: the halo form.
  const double* ll_all;
  const double* lp_all;       // needed by the posterior-ordering cut only
  // Recovery of a run of surviving picks longer than a halo (ptm_set_shard_map).  Whether ANY shard of the ladder is blind in a
  // step is a property of the candidate draws, which every shard replays: with the shards' boundaries known, every shard finds
  // the same ladders.  Such a ladder is left alone by every shard in the halo pass (flagged, counted) and decided afterwards
  // from the whole ladder's llikes (redo_only: the gathered form, for the flagged ladders alone).
  const int* shard_ends;      // [nshards] global rung where each shard ends (the last one = Nt), or null: no recovery
  int nshards, halo_nominal;  // the halo depth every shard asks for (a shard's own is clipped to the shard above it)
  int* redo_flag;             // [W]
  int* redo_count;            // [1]
  int redo_only;
  double* x;                  // [Nc][DP] rows (only the overflow path moves rows here)
  double* ll;
  double* lp;
  unsigned char* touch;
  int *arr_below, *arr_above;      // [W]  landing slot of the row arriving across the lower / upper boundary, or -1
  int* swap_log;                   // [W][ms]  this step's slot of the log ring; per candidate: -2 none/dropped, -3 not this
                                   //          shard's, else rung | accepted<<30
  double *send_up, *send_down;     // boundary messages or null
  int row_cap;
  int *mv_src, *mv_dst, *mv_n;     // [W][MVCAP], [W][MVCAP], [W]: the ladder's row moves for move_kernel
  int* err;
  // history (Hist, ptm_kernels.hpp): a rung touched twice in one step makes two add_state calls; the first one sees the
  // row the rung held BETWEEN its two exchanges.  If that call is one that saves, the row is copied by move_kernel.
  Hist hist;
  int add_every_n;
  const unsigned int* nhist;
  const int *naccept, *ntries, *last_type;
  MapT map;   // MAP tracking: the in-between row is a candidate too (its log-posterior at this rung's temperature)
  // evolving ladders (parallel_tempering_chains::evolve_temps, chain.hh:302-307): every accepted exchange pries its gap
  // apart (pry_temps, chain.cc:1501-1518,1809-1846), so each ladder owns its temperatures.  Whole-ladder shards only.
  double evolve_rate;   // 0: fixed ladder (beta[] rules)
  double evolve_cut;    // evolve_temp_lpost_cut (chain.hh:254,302-307; chain.cc:1819-1827): < 0 off, the default
  double* beta_w;       // [W][Nt] the ladders' inverse temperatures, rewritten after a step that pried
  int lp_is_const;      // every chain's lprior is lp_const (all-uniform prior, every state inside the box): the exchange moves no lprior
  double lp_const;
  double* betaC_direct; // [Nc] few ladders: the chain-indexed image is written here as well (no transposition launch); else null
  double* beta_add;     // [Nc] with history / MAP tracking: the temperature each touched rung had at its last add_state of
                        // the phase (the sweep kernel saves that row); null otherwise
};
constexpr int HIST_DST = -(1 << 30);   // move-list destination code: HIST_DST - c = "into chain c's history"
constexpr int MAP_DST = -(1 << 29);    //                             MAP_DST - c  = "chain c's new MAP" (c < 2^29)

// llike of global rung r for walker w, r inside the shard's window
__device__ __forceinline__ double win_llike(const Decide& p, int r, int w) {
  if (p.ll_all) return p.ll_all[(size_t)r * p.W + w];
  const int r1 = p.r0 + p.nloc;
  if (r < p.r0) return p.ll_below[w];
  if (r >= r1) return p.ll_above[(size_t)(r - r1) * p.W + w];
  return p.ll[(size_t)(r - p.r0) * p.W + w];
}

// bijection block -> walker that gives XCD k (blocks k, k+8, ...) the k-th contiguous eighth of the walkers
__device__ __forceinline__ int xcd_walker(int b, int W) {
  const int q = W >> 3, rem = W & 7, xcd = b & 7;
  return xcd * q + (xcd < rem ? xcd : rem) + (b >> 3);
}

constexpr int PTM_LOG_RING = 16;   // steps of candidate logs kept before they are folded into the swap counters
constexpr int MVCAP = 256;  // rows one ladder can move per step on the register path (move_kernel)
typedef double d2_t __attribute__((ext_vector_type(2)));  // (HIP's double2 struct does not stay in registers as an array)

// do perm / inv / the move list share the space of an evolving ladder's trial operands (decide_body's carve)?  Host and device agree.
__host__ __device__ inline bool decide_aliased(int ms, int WN, bool evolve) {
  return evolve && (size_t)2 * ((WN + 3) & ~3) * 2 + (size_t)2 * MVCAP * 4 <= (size_t)4 * ((ms + 3) & ~3) * 8;
}
// atomic minimum of a 16-bit LDS entry (ds_cmpst on the word that holds it; contention: two picks of one rung pair, rare)
__device__ __forceinline__ void lds_min_u16(unsigned short* a, int idx, unsigned int v) {
  unsigned int* wd = reinterpret_cast<unsigned int*>(a) + (idx >> 1);
  const int sh = (idx & 1) * 16;
  unsigned int old = *wd;
  while (((old >> sh) & 0xffffu) > v) {
    const unsigned int prev = atomicCAS(wd, old, (old & ~(0xffffu << sh)) | (v << sh));
    if (prev == old) break;
    old = prev;
  }
}

// exclusive prefix of nq chunk totals, left to right, by ONE lane: off[q] = tot[0] + ... + tot[q - 1] (in that order), returns the grand
// total.  Whole groups of eight without a test per element: a lone lane issues an instruction every ~10 cycles whatever it is
// (the persistent ladder kernel's scan of the same name gained 0.8 us per call at 32 totals from exactly this).
__device__ __forceinline__ double decide_totals_scan(const double* tot, double* off, int nq) {
  double run = 0.0;
  int q0 = 0;
#pragma unroll 1
  for (; q0 + 8 <= nq; q0 += 8) {
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = tot[q0 + j];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const double t = run; run = run + r[j]; r[j] = t; }
#pragma unroll
    for (int j = 0; j < 8; ++j) off[q0 + j] = r[j];
  }
  for (; q0 < nq; ++q0) { const double t = tot[q0]; off[q0] = run; run = run + t; }
  return run;
}

// The reference decides the candidates strictly in pick order (chain.cc:1410-1537).  Two facts make that order
// parallel over the ladder without changing any outcome:
//   (1) filter: a pick n is dropped iff an earlier SURVIVING pick is n or n-1 (chain.cc:1417-1418).  Only the first
//       pick of a rung value can survive, and alive[n] = !(alive[n-1] && first[n-1] < first[n]): a recurrence along
//       RUNS of consecutive picked rungs, independent between runs;
//   (2) trials: two surviving picks on adjacent rungs (n, n+1) exist only if n+1 was picked first, and only then does
//       pick n see an updated upper rung -- so each run of consecutive surviving rungs is decided top-down, and runs
//       are independent of each other.
// One lane walks each run; everything else (draws, counters, move list) is parallel over picks.  The draws and the
// filter cover the whole ladder (they are replicated on every shard); everything after them runs over the compacted
// list of the surviving picks inside the shard's window, on LDS arrays indexed by window rung.
// DECIDE_THREADS = 256 when the shard sees most of the ladder (~200 candidates and ~100 survivors in ONE pass per phase),
// 64 for a short shard of many ladders, where the per-block fixed costs are what counts.
// The body is a device function of the block's DECIDE_THREADS threads working for ladder (walker) w on `smem`: decide_kernel
// below is its plain launch; the fused small-ladder kernel (ptm_fused_kernel.hpp) calls it once per step from inside its loop.
template <int DECIDE_THREADS, bool CUT = false>   // CUT: the build that knows evolve_temps' posterior-ordering cut (the others carry none of it)
__device__ __forceinline__ void decide_body(const Decide& p, unsigned char* smem, const int w, const uint64_t step, int* const swap_log) {
  // (step and swap_log are parameters of their own: a caller that walks several steps -- the fused small-ladder kernel -- must
  //  not write into its copy of the parameter block, which would then live in vector registers)
  const int lane = threadIdx.x;
  const int Nt = p.Nt, ms = p.ms;
  const int NONE = 0xffff;
  const int r1 = p.r0 + p.nloc;
  // window of rungs whose llike this shard knows: its own, one below, H above
  const int wlo = p.ll_all ? 0 : p.r0 - (p.ll_below ? 1 : 0), whi = p.ll_all ? Nt - 1 : r1 - 1 + p.H;
  const int WN = whi - wlo + 1;
  // LDS carve (mirrored by decide_lds_bytes on the host; all offsets multiples of 8)
  double* llc_ = reinterpret_cast<double*>(smem);                             // [WN]  llike view of the touched rungs
  unsigned short* first = reinterpret_cast<unsigned short*>(llc_ + WN);       // [Nt]  first pick of each rung value (NONE: not picked)
  int* cand = reinterpret_cast<int*>(first + ((Nt + 3) & ~3));                // [ms]  rung of the pick / -2 none or dropped
  uint32_t* ua = reinterpret_cast<uint32_t*>(cand + ((ms + 1) & ~1));         // [ms]  accept uniform of the pick (raw)
  int* cnt = reinterpret_cast<int*>(ua + ((ms + 1) & ~1));                    // [4]   list length, move count, ambiguous trial seen, pries
  // perm [WN] (source rung of the row now at a rung), inv [WN] (its inverse) and the move list [2][MVCAP] are not needed before
  // the trials of an evolving ladder are over, whose operand arrays (tlu .. tllb below) are dead by then: they share that space
  // when it is big enough (decide_aliased) -- 6 KB less per block, a fifth block per CU for 1024-rung ladders
  const bool al = decide_aliased(ms, WN, p.evolve_rate > 0);
  const int WNp = (WN + 3) & ~3;
  unsigned short* perm_ = reinterpret_cast<unsigned short*>(cnt + 4);         // [WN]
  unsigned short* inv_ = perm_ + (al ? 0 : WNp);                              // [WN]
  unsigned short* list = inv_ + (al ? 0 : WNp);                               // [ms]  surviving picks that are ours
  unsigned short* midk = list + ((ms + 3) & ~3);                              // [ms]  (by pick) row the pick's lower rung held before a later pick
                                                                              //       on the pair below exchanged it again
  unsigned char* alive = reinterpret_cast<unsigned char*>(midk + ((ms + 3) & ~3));  // [ms] 0 dropped, 1 survives and is
                                                                                    //      ours, 2 survives, not ours
  unsigned char* accf = alive + ((ms + 7) & ~7);                              // [ms]
  // The block applies the ladder's row moves itself (below) from a list kept in LDS, up to FCAP rows (16 rounds of one
  // row per 16-lane group); a longer list (64-thread form only) goes to move_kernel through global memory.
  constexpr int FCAP = DECIDE_THREADS < MVCAP ? DECIDE_THREADS : MVCAP;   // (the list itself holds MVCAP moves)
  int* lmv = reinterpret_cast<int*>(accf + ((ms + 7) & ~7));                  // [2][MVCAP]
  // evolving ladders only: gaps (in the end their local prefix sums), chunk totals / offsets, {normaliser, pries}, and the
  // surviving picks in PICK ORDER (position t): what the chain of dependent trials needs, laid out for one lane to stream
  const int msp = (ms + 3) & ~3;
  double* gap = reinterpret_cast<double*>(lmv + (al ? 0 : 2 * MVCAP));        // [Nt]
  double* ct = gap + Nt;                                                      // [2][(Nt + 31) / 32]
  double* ev = ct + 2 * ((Nt + 31) / 32);                                     // [2]
  double* tlu = ev + 2;                                                       // [ms]  log of the pick's accept uniform
  double* tgap = tlu + msp;                                                   // [ms]  the pick's gap (a pair is tried once: never pried before)
  double* tlla = tgap + msp;                                                  // [ms]  llike of the lower rung (nothing earlier can change it)
  double* tllb = tlla + msp;                                                  // [ms]  llike of the upper rung (an earlier pick above may change it)
  unsigned short* olist = reinterpret_cast<unsigned short*>(tllb + msp);      // [ms]  position -> pick
  unsigned short* opos = olist + msp;                                         // [ms]  pick -> position
  unsigned short* ti = opos + msp;                                            // [ms]  the pick's lower rung
  short* tdep = reinterpret_cast<short*>(ti + msp);                           // [ms]  position of the (later) pick on the pair below, or -1
  unsigned char* tacc = reinterpret_cast<unsigned char*>(tdep + msp);         // [ms]  accepted
  // ... with history / MAP tracking on top (an add_state of the phase sees the temperature BETWEEN two pries of the step):
  double* p0 = reinterpret_cast<double*>(tacc + ((ms + 7) & ~7));             // [max(Nt, MVCAP)]  prefix sums of the step's first gaps
  double* tS = p0 + (Nt > MVCAP ? Nt : MVCAP);                                // [ms]  sum of the gaps the pick saw (0: nothing pried yet)
  double* tdl = tS + msp;                                                     // [ms]  what the pick added to its gap
  double* bklo = tdl + msp;                                                   // [ms]  (by pick) temperature of the pick's lower rung then
  double* tbl = bklo + msp;                                                   // [ms]  stored temperature of the pick's lower rung (asked for with the other operands:
  double* tbh = tbl + msp;                                                    // [ms]  ... and of its upper rung     a trip to memory off the phase's critical path)
  double* gb = p0;                                                            // [MVCAP] temperature for a HIST / MAP move (p0 is done by then)
  // ... with a posterior-ordering cut (evolve_cut >= 0; the carve above is then taken as with history): the current llike and
  // lprior of EVERY rung, exchanged as the picks are decided
  double* llv = tbh + msp;                                                    // [Nt]
  double* lpv = llv + Nt;                                                     // [Nt]
  if (al) {
    inv_ = reinterpret_cast<unsigned short*>(tlu);
    perm_ = inv_ + WNp;
    lmv = reinterpret_cast<int*>(perm_ + WNp);
  }
  double* llc = llc_ - wlo;                // indexed by global rung
  unsigned short* perm = perm_ - wlo;
  unsigned short* inv = inv_ - wlo;
  cdp beta = as_c(p.beta);

  if (p.redo_only && !p.redo_flag[w]) return;          // the second pass works for the ladders the halo pass left alone
  for (int i = lane; i < ((Nt + 3) & ~3) / 2; i += DECIDE_THREADS) reinterpret_cast<uint32_t*>(first)[i] = 0xffffffffu;
  if (lane == 0) { cnt[0] = 0; cnt[1] = 0; p.arr_below[w] = -1; p.arr_above[w] = -1; }
  __syncthreads();
  // -- candidate draws (chain.cc:1410-1416): block k of the ladder stream gives {u_try, u_pick, u_accept}
  const uint64_t try_below = u01_below_bound(p.thresh);
  for (int k = lane; k < ms; k += DECIDE_THREADS) {
    const u32x4 o = draw_block(p.seed, TAG_PT, (uint32_t)(w + p.w_off), step, (uint32_t)k);
    int n = -2;
    if (Nt > 1 && (uint64_t)o.v0 < try_below) n = u01_times(o.v1, Nt - 1);   // u01(v0) < thresh, (int)(u01(v1) (Nt - 1)): ptm_device_math.hpp
    cand[k] = n;
    ua[k] = o.v2;  // the accept uniform's slot is reserved whether or not it is needed (cf. Q5)
    alive[k] = 0;
    accf[k] = 0;
    if (n >= 0) lds_min_u16(first, n, (unsigned int)k);
  }
  const bool evolve = p.evolve_rate > 0 && Nt > 1;
  const bool evb = evolve && p.beta_add != nullptr;   // history / MAP tracking of evolving ladders
  const bool cutmode = CUT && evolve && p.evolve_cut >= 0;   // ... with a posterior-ordering cut
  double blast = 0.0;   // the ladder's last inverse temperature (chain.cc:1833 needs 1 - it): asked for here, used phases later
  if (evolve) {
    const double* bw = p.beta_w + (size_t)w * Nt;
    blast = bw[Nt - 1];
    for (int k = lane; k < Nt - 1; k += DECIDE_THREADS) gap[k] = bw[k] - bw[k + 1];   // chain.cc:1816
  }
  __syncthreads();
  // -- filter (1): run heads walk their run upwards
  for (int k = lane; k < ms; k += DECIDE_THREADS) {
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
  if (p.redo_only) {
    if (lane == 0) p.redo_flag[w] = 0;
  } else if (p.shard_ends && !p.ll_all) {
    // is some shard of this ladder blind this step?  Shard k (ending at rung B) sees the llikes of h = min(halo, size of shard
    // k + 1) rungs above it; it cannot decide its top pairs iff the picks B - 1 .. B - 1 + h all survive (see the trials below)
    if (lane == 0) cnt[2] = 0;
    __syncthreads();
    for (int k = lane; k + 1 < p.nshards; k += DECIDE_THREADS) {
      const int B = p.shard_ends[k];
      const int above = p.shard_ends[k + 1] - B;
      const int h = p.halo_nominal < above ? p.halo_nominal : above;
      bool all = true;
      for (int t = 0; t <= h && all; ++t) all = PTM_ALIVE_RUNG(B - 1 + t);
      if (all) cnt[2] = 1;
    }
    __syncthreads();
    if (cnt[2]) {                                      // every shard leaves this ladder to the gathered pass
      if (lane == 0) { p.redo_flag[w] = 1; atomicAdd(p.redo_count, 1); }
      for (int k = lane; k < ms; k += DECIDE_THREADS) swap_log[(size_t)w * ms + k] = -2;   // (nothing of this ladder is logged by this pass)
      return;
    }
  }
  // -- compaction: the surviving picks whose pair lies inside the window
  for (int k = lane; k < ms; k += DECIDE_THREADS) {
    const int n = cand[k];
    if (n < 0) continue;
    if (!alive[k]) { cand[k] = -2; continue; }
    if (n < wlo || n + 1 > whi) { alive[k] = 2; continue; }     // survives, but is not this shard's to decide
    list[atomicAdd(&cnt[0], 1)] = (unsigned short)k;
  }
  __syncthreads();
  const int nl = cnt[0];
  // -- working copy of the touched rungs (gather_llikes, chain.cc:1434); each touched rung is set up by exactly one
  //    lane: the pick whose lower rung it is, or -- for the top of a run -- the pick just below it
  for (int j = lane; j < nl; j += DECIDE_THREADS) {
    const int n = cand[list[j]];
    const bool top = !PTM_ALIVE_RUNG(n + 1) || n + 2 > whi;    // (an alive pick above that lies outside the window sets up nothing)
    const double a = win_llike(p, n, w);
    const double b = top ? win_llike(p, n + 1, w) : 0.0;
    llc[n] = a;
    if (!al) perm[n] = (unsigned short)n;                      // (sharing the trial operands' space: set up after the trials)
    if (top) {
      llc[n + 1] = b;
      if (!al) perm[n + 1] = (unsigned short)(n + 1);
    }
  }
  __syncthreads();
  // -- evolving ladder: every accepted exchange changes the normalisation of ALL the gaps (pry_temps renormalises the
  //    ladder, chain.cc:1829-1844), so the trials are one chain in pick order.  The gaps stay lazily normalised: a pry is
  //    gap[i] *= 1 + rate and S += the increase; the gap a later trial sees is gap[i] / (S / (1 - beta_last)) -- O(1) per
  //    exchange, and the very bits of the stored temperatures until the step's first accepted exchange.  The Metropolis
  //    test is taken with both sides multiplied by S, so the chain of dependent trials holds no division.
  if (evolve) {
    const int nch = (Nt - 1 + 31) / 32;
    // the surviving picks in pick order (one wave: ballot + prefix count)
    if (lane < 64) {
      int base = 0;
      for (int k0 = 0; k0 < ms; k0 += 64) {
        const int k = k0 + lane;
        const bool f = k < ms && alive[k] == 1;
        const unsigned long long m = __ballot(f);
        if (f) {
          const int t = base + __builtin_popcountll(m & ((1ull << lane) - 1ull));
          olist[t] = (unsigned short)k;
          opos[k] = (unsigned short)t;
        }
        base += __builtin_popcountll(m);
      }
    }
    for (int q = lane; q < nch; q += DECIDE_THREADS) {   // S: chunks of 32 left to right, then the chunk totals
      // (eight operands asked for before the chain of dependent additions: a lone lane pays ~100 cycles per dependent LDS read)
      double loc = 0.0;
#pragma unroll 1
      for (int j0 = 0; j0 < 32; j0 += 8) {
        double r8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int k = 32 * q + j0 + j; r8[j] = k < Nt - 1 ? gap[k] : 0.0; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = 32 * q + j0 + j;
          if (k < Nt - 1) { if (evb || cutmode) p0[k] = loc; loc = loc + r8[j]; }
        }
      }
      ct[q] = loc;
    }
    __syncthreads();
    for (int t = lane; t < nl; t += DECIDE_THREADS) {
      const int k = olist[t];
      const int i = cand[k];
      ti[t] = (unsigned short)i;
      tlu[t] = dlog_u01(ua[k]);
      tgap[t] = gap[i];
      tlla[t] = llc[i];
      tllb[t] = llc[i + 1];
      tdep[t] = PTM_ALIVE_RUNG(i - 1) ? (short)opos[first[i - 1]] : (short)-1;
      tacc[t] = 0;
      if (evb && !cutmode) { const double* bw = p.beta_w + (size_t)w * Nt; tbl[t] = bw[i]; tbh[t] = bw[i + 1]; }
    }
    __syncthreads();
    if (cutmode) {
      // evolve_temp_lpost_cut >= 0 (chain.cc:1819-1827): every pry also widens each gap whose two chains' log-posteriors are out
      // of order by more than cut * invtemp -- any gap of the ladder, so nothing stays lazy: the picks are decided one after the
      // other in pick order, and after every accepted exchange the block goes over all the gaps and rebuilds their prefix sums
      // (chunks of 32 left to right, then the chunk totals: ptmo_chunk_prefix).  Once something was pried a rung's temperature
      // is 1 - P0 / normaliser.  ev[0]: the gaps' running total, ev[1]: pries so far.
      const double* bw = p.beta_w + (size_t)w * Nt;
      {
        const double* lla = p.ll_all ? p.ll_all : p.ll;   // (a whole-ladder engine's own arrays ARE the whole ladder's)
        const double* lpa = p.ll_all ? p.lp_all : p.lp;
        for (int r = lane; r < Nt; r += DECIDE_THREADS) { llv[r] = lla[(size_t)r * p.W + w]; lpv[r] = lpa[(size_t)r * p.W + w]; }
      }
      if (lane == 0) {
        const double S = decide_totals_scan(ct, ct + nch, nch);
        ev[0] = S; ev[1] = 0.0;
      }
      __syncthreads();
      for (int k = lane; k < Nt - 1; k += DECIDE_THREADS) p0[k] = ct[nch + (k >> 5)] + p0[k];
      __syncthreads();
      const double c1 = 1 - blast;   // chain.cc:1833
      const double grow = 1.0 + p.evolve_rate;
      for (int t = 0; t < nl; ++t) {
        const int k = olist[t], i = ti[t];
        if (lane == 0) {
          const double S = ev[0];
          const bool pried = ev[1] > 0;
          const double nrm = S / c1;
          double lla = llv[i];
          if (!(lla > -1e200)) lla = -1e200;
          double llb = llv[i + 1];
          if (!(llb > -1e200)) llb = -1e200;
          const double g = gap[i];
          bool acc = true;
          if (pried) {
            const double tt = (g * c1) * (llb - lla);
            if (tt < 0) acc = tlu[t] * S < tt;
          } else {
            const double logH = g * (llb - lla);
            if (logH < 0) acc = tlu[t] < logH;
          }
          tacc[t] = acc ? 1 : 0;
          if (evb) {   // the two rungs' temperatures as this pick's add_state calls see them (before its own pry)
            double ba = bw[i], bb = bw[i + 1];
            if (pried) {
              if (i > 0) ba = 1 - p0[i] / nrm;
              if (i + 1 < Nt - 1) bb = 1 - p0[i + 1] / nrm;
            }
            bklo[k] = ba;
            // (chain-indexed: the shard's own rungs -- a rung shard replays the whole ladder's picks, ptm_exchange_decide_gathered)
            if (i + 1 >= p.r0 && i + 1 < r1) p.beta_add[(size_t)(i + 1 - p.r0) * p.W + w] = bb;
            if (!PTM_ALIVE_RUNG(i - 1) && i >= p.r0 && i < r1) p.beta_add[(size_t)(i - p.r0) * p.W + w] = ba;
          }
          if (acc) {
            const double a = llv[i]; llv[i] = llv[i + 1]; llv[i + 1] = a;
            const double b = lpv[i]; lpv[i] = lpv[i + 1]; lpv[i + 1] = b;
          }
        }
        __syncthreads();
        if (tacc[t]) {   // pry_temps({i}, rate, invtemps, gather_lposts()): chain.cc:1516-1517,1809-1846
          const double S = ev[0];
          const bool pried = ev[1] > 0;
          const double nrm = S / c1;
          for (int kk = lane; kk < Nt - 1; kk += DECIDE_THREADS) {
            const double b0 = (kk == 0 || !pried) ? bw[kk] : 1 - p0[kk] / nrm;
            const double b1 = (kk + 1 == Nt - 1 || !pried) ? bw[kk + 1] : 1 - p0[kk + 1] / nrm;
            const double t0 = b0 * llv[kk], t1 = b1 * llv[kk + 1];
            const double l0 = lpv[kk] + t0, l1 = lpv[kk + 1] + t1;   // the chains' current log-posteriors (MH_chain::resetTemp, chain.cc:1088-1091)
            double gk = gap[kk];
            if (l0 - l1 > p.evolve_cut * b0) gk = gk * grow;          // :1819-1827
            if (kk == i) gk = gk * grow;                              // :1829
            gap[kk] = gk;
          }
          __syncthreads();
          for (int q = lane; q < nch; q += DECIDE_THREADS) {
            double loc = 0.0;
            for (int kk = 32 * q; kk < Nt - 1 && kk < 32 * q + 32; ++kk) { p0[kk] = loc; loc = loc + gap[kk]; }
            ct[q] = loc;
          }
          __syncthreads();
          if (lane == 0) {
            const double S2 = decide_totals_scan(ct, ct + nch, nch);
            ev[0] = S2; ev[1] = ev[1] + 1.0;
          }
          __syncthreads();
          for (int kk = lane; kk < Nt - 1; kk += DECIDE_THREADS) p0[kk] = ct[nch + (kk >> 5)] + p0[kk];
          __syncthreads();
        }
      }
    }
    // The trials of an evolving ladder are ONE chain in pick order only through the normaliser S, and S moves little: by
    // rate x (the pried gaps) -- a relative 1e-3 at most.  So first every run of surviving picks is walked by its own lane
    // (as for a fixed ladder) with S known only to lie in [S0, S0 + all the increments any pick could add]: lu S is
    // monotone in S, so a trial that passes at S0 passes for every S of the interval and one that fails at the upper end fails
    // everywhere -- and as long as it also gives the same answer by the first-trial form (nothing pried yet: no S at all) the
    // decision is the sequential walk's, whatever came before it.  Only a ladder with a trial inside that window (about one
    // in fifty at 1024 rungs) takes the sequential walk below.  (History / MAP tracking want the normaliser every pick SAW: with the
    // decisions known it is a sum of the accepted picks' increases in pick order -- one lane's chain of additions, no trial on it.)
    bool walked = false;
    if (!cutmode) {
      const double c1 = 1 - blast;   // chain.cc:1833
      const double grow = 1.0 + p.evolve_rate;
      if (lane == 0) {
        const double S = decide_totals_scan(ct, ct + nch, nch);
        ev[0] = S; ev[1] = 0.0;
        cnt[2] = 0; cnt[3] = 0;
      }
      __syncthreads();
      {   // an upper bound of every increment together (any order: it is a bound, taken with a margin)
        double part = 0.0;
        for (int t = lane; t < nl; t += DECIDE_THREADS) part += tgap[t] * grow - tgap[t];
        if (part != 0.0) atomicAdd(&ev[1], part);
      }
      __syncthreads();
      const double S_lo = ev[0], S_hi = (ev[0] + ev[1]) * (1.0 + 1e-12);
      for (int j = lane; j < nl; j += DECIDE_THREADS) {
        const int n = cand[list[j]];
        if (PTM_ALIVE_RUNG(n + 1)) continue;                       // not the top of a run
        bool carried = false;
        double carry = 0.0;
        for (int i = n;; --i) {
          const int t = opos[first[i]];
          const double lu = tlu[t], g = tgap[t];
          const double llb_raw = carried ? carry : tllb[t];
          double lla = tlla[t];
          if (!(lla > -1e200)) lla = -1e200;
          double llb = llb_raw;
          if (!(llb > -1e200)) llb = -1e200;
          const double dl = llb - lla;
          const double logH = g * dl, tt = (g * c1) * dl;
          const bool accA = !(logH < 0) || lu < logH;               // nothing pried yet (chain.cc:1463-1467 on the stored temperatures)
          const bool acc_lo = !(tt < 0) || lu * S_lo < tt;          // the hardest S to pass
          const bool acc_hi = !(tt < 0) || lu * S_hi < tt;          // the easiest
          if (accA != acc_lo || accA != acc_hi) { cnt[2] = 1; break; }   // inside the window: this ladder walks in order
          tacc[t] = accA ? 1 : 0;
          carried = accA;
          carry = llb_raw;                                          // the row now on rung i came from rung i + 1
          if (!PTM_ALIVE_RUNG(i - 1)) break;
        }
      }
      __syncthreads();
      if (cnt[2] == 0) {
        for (int t = lane; t < nl; t += DECIDE_THREADS)
          if (tacc[t]) { gap[ti[t]] = tgap[t] * grow; atomicAdd(&cnt[3], 1); }   // chain.cc:1829
        __syncthreads();
        if (lane == 0) ev[1] = (double)cnt[3];
        if (evb && lane == 0) {
          // what the sequential walk would have noted per pick: the normaliser it saw (0: nothing pried yet), what it added
          double S = ev[0];
          int npry = 0;
          for (int t0 = 0; t0 < nl; t0 += 8) {   // (operands first: a lone lane pays ~100 cycles per dependent LDS read)
            double g8[8];
            bool a8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { g8[j] = t0 + j < nl ? tgap[t0 + j] : 0.0; a8[j] = t0 + j < nl && tacc[t0 + j] != 0; }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              if (t0 + j >= nl) break;
              tS[t0 + j] = npry ? S : 0.0;
              double inc = 0.0;
              if (a8[j]) { const double sn = g8[j] * grow; inc = sn - g8[j]; S = S + inc; ++npry; }   // chain.cc:1829
              tdl[t0 + j] = inc;
            }
          }
        }
        walked = true;
      } else {
        for (int t = lane; t < nl; t += DECIDE_THREADS) tacc[t] = 0;
      }
      __syncthreads();
    }
    if (!cutmode && !walked && lane == 0) {
      double S = decide_totals_scan(ct, ct + nch, nch);
      const double c1 = 1 - blast;   // chain.cc:1833
      const double grow = 1.0 + p.evolve_rate;
      int npry = 0;
      // one lane streams the picks; the next pick's operands are asked for before this one is decided
      double n_lu = 0, n_gap = 0, n_lla = 0, n_llb = 0, fwd = 0;
      int n_dep = -1, n_i = 0, fwd_to = -1;
      if (nl > 0) { n_lu = tlu[0]; n_gap = tgap[0]; n_lla = tlla[0]; n_llb = tllb[0]; n_dep = tdep[0]; n_i = ti[0]; }
      for (int t = 0; t < nl; ++t) {
        const double lu = n_lu, g = n_gap, lla_raw = n_lla, llb_raw = (fwd_to == t) ? fwd : n_llb;
        const int dep = n_dep, i = n_i;
        if (t + 1 < nl) { n_lu = tlu[t + 1]; n_gap = tgap[t + 1]; n_lla = tlla[t + 1]; n_llb = tllb[t + 1]; n_dep = tdep[t + 1]; n_i = ti[t + 1]; }
        double lla = lla_raw;
        if (!(lla > -1e200)) lla = -1e200;
        double llb = llb_raw;
        if (!(llb > -1e200)) llb = -1e200;
        // log u < logH with logH = (gap / (S / c1)) * (llb - lla), both sides times S > 0: no division on this chain of
        // dependent trials.  Until the step's first pry the gap is the stored temperatures' own difference.
        bool acc = true;
        if (npry) {
          const double tt = (g * c1) * (llb - lla);
          if (tt < 0) acc = lu * S < tt;
        } else {
          const double logH = g * (llb - lla);
          if (logH < 0) acc = lu < logH;
        }
        if (evb) { tS[t] = npry ? S : 0.0; tdl[t] = 0.0; }
        if (acc) {
          tacc[t] = 1;
          const double sn = g * grow;   // chain.cc:1829
          const double inc = sn - g;
          S = S + inc;
          gap[i] = sn;
          ++npry;
          if (evb) tdl[t] = inc;
          if (dep >= 0) { tllb[dep] = llb_raw; fwd = llb_raw; fwd_to = dep; }   // the row now on rung i came from rung i + 1
        }
      }
      ev[1] = (double)npry;
    }
    __syncthreads();
    if (al) {   // the trial operands are dead: perm moves in
      for (int j = lane; j < nl; j += DECIDE_THREADS) {
        const int n = cand[list[j]];
        perm[n] = (unsigned short)n;
        if (!PTM_ALIVE_RUNG(n + 1) || n + 2 > whi) perm[n + 1] = (unsigned short)(n + 1);
      }
      __syncthreads();
    }
  }
  // -- trials (2): the top pick of each run of surviving rungs walks the run downwards (chain.cc:1436-1537); with an
  //    evolving ladder the decisions are the ones just taken
  for (int j = lane; j < nl; j += DECIDE_THREADS) {
    const int n = cand[list[j]];
    const bool up = PTM_ALIVE_RUNG(n + 1);
    if (up && n + 1 < whi) continue;                             // not the top of a run (the pick above is in the list)
    if (up) {
      // the pick above decides the pair (whi, whi+1), outside the window, and may replace rung whi: the content of
      // every rung of this run is then unknown here.  Harmless as long as the run ends above the shard's own rungs.
      for (int i = n; i >= wlo; --i) {
        if (i + 1 <= r1) atomicOr(p.err, 2);                     // would decide a local / straddling exchange blindly
        alive[first[i]] = 2;
        if (!PTM_ALIVE_RUNG(i - 1)) break;
      }
      continue;
    }
    for (int i = n; i >= wlo; --i) {                             // below the window nothing concerns us
      const int kk = first[i];
      bool acc = true;
      if (evolve) {
        acc = tacc[opos[kk]] != 0;
      } else {
        double lla = llc[i];
        if (!(lla > -1e200)) lla = -1e200;
        double llb = llc[i + 1];
        if (!(llb > -1e200)) llb = -1e200;
        const double logH = -(beta[i + 1] - beta[i]) * (llb - lla);
        if (logH < 0) acc = dlog_u01(ua[kk]) < logH;
      }
      if (acc) {
        // the row that leaves this shard downwards must be one of ours (else it crossed two boundaries in one step)
        if (i + 1 == p.r0 && (perm[i + 1] < p.r0 || perm[i + 1] >= r1)) atomicOr(p.err, 1);
        const double t = llc[i]; llc[i] = llc[i + 1]; llc[i + 1] = t;
        const unsigned short s = perm[i]; perm[i] = perm[i + 1]; perm[i + 1] = s;
        accf[kk] = 1;
      }
      if (!PTM_ALIVE_RUNG(i - 1)) break;
      midk[kk] = perm[i];   // the pick below exchanges rung i again: this is what it held in between
    }
  }
  __syncthreads();
  if (evolve) {
    const int nch = (Nt - 1 + 31) / 32;
    if (evb && !cutmode) {
      // The temperature a rung had when a pick's add_state calls reached it (both rungs of the pair, before the pick's own
      // pry; chain.cc:1487-1490,1531-1534): 1 - (P0 + D) / normaliser, P0 = prefix of the step's first gaps, D = what
      // the earlier accepted picks added to the gaps below the rung, in pick order.  No gap between the pair's two rungs
      // has been pried yet (a pair is tried once), so D is the same for both.
      for (int k = lane; k < Nt - 1; k += DECIDE_THREADS) p0[k] = ct[nch + (k >> 5)] + p0[k];
      __syncthreads();
      for (int j = lane; j < nl; j += DECIDE_THREADS) {
        const int k = list[j];
        const int i = cand[k];
        const int t = opos[k];
        double D = 0.0;
        for (int t2 = 0; t2 < t; ++t2)
          if (tacc[t2] && ti[t2] < i) D = D + tdl[t2];
        const double Sk = tS[t];
        const double nk = Sk / (1 - blast);   // the normaliser then (chain.cc:1833)
        const double blo = (Sk == 0.0 || i == 0) ? tbl[t] : 1 - (p0[i] + D) / nk;
        const double bhi = (Sk == 0.0 || i + 1 == Nt - 1) ? tbh[t] : 1 - (p0[i + 1] + D) / nk;
        bklo[k] = blo;
        // last add of the phase: always for the upper rung (a pick on the pair above came earlier), for the lower rung
        // unless a later pick exchanges it again
        if (i + 1 >= p.r0 && i + 1 < r1) p.beta_add[(size_t)(i + 1 - p.r0) * p.W + w] = bhi;
        if (!PTM_ALIVE_RUNG(i - 1) && i >= p.r0 && i < r1) p.beta_add[(size_t)(i - p.r0) * p.W + w] = blo;
      }
      __syncthreads();
    }
    if (ev[1] > 0) {   // the new temperatures (chain.cc:1834-1844): beta_k = 1 - P_k / (total / (1 - beta_last))
      for (int q = lane; q < nch; q += DECIDE_THREADS) {
        double loc = 0.0;
#pragma unroll 1
        for (int j0 = 0; j0 < 32; j0 += 8) {
          double r8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) { const int k = 32 * q + j0 + j; r8[j] = k < Nt - 1 ? gap[k] : 0.0; }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int k = 32 * q + j0 + j;
            if (k < Nt - 1) { gap[k] = loc; loc = loc + r8[j]; }
          }
        }
        ct[q] = loc;
      }
      __syncthreads();
      if (lane == 0) {
        const double off = decide_totals_scan(ct, ct + nch, nch);
        ev[0] = off / (1 - blast);
      }
      __syncthreads();
      const double nn = ev[0];
      for (int k = 1 + lane; k < Nt - 1; k += DECIDE_THREADS) {
        const double bk = 1 - (ct[nch + (k >> 5)] + gap[k]) / nn;
        p.beta_w[(size_t)w * Nt + k] = bk;
        if (p.betaC_direct && k >= p.r0 && k < r1) p.betaC_direct[(size_t)(k - p.r0) * p.W + w] = bk;
      }
    }
  }
  // -- the step's log
  for (int k = lane; k < ms; k += DECIDE_THREADS)
    swap_log[(size_t)w * ms + k] = alive[k] == 1 ? (cand[k] | (accf[k] ? 0x40000000 : 0)) : (alive[k] == 2 ? -3 : -2);
  // -- counters, the touch counts of the local rungs and the inverse permutation
  for (int j = lane; j < nl; j += DECIDE_THREADS) {
    const int k = list[j];
    if (alive[k] != 1) continue;
    const int i = cand[k];
    // (swap_count / swap_accept_count, chain.cc:1498,1536, are not touched here: 185 scattered read-modify-writes per
    //  ladder and step.  fold_swap_log_kernel adds the logged steps to them every PTM_LOG_RING steps and on demand.)
    const int rtop = (PTM_ALIVE_RUNG(i + 1) && alive[first[i + 1]] == 1) ? i : i + 1;  // rung i+1 belongs to the pick above, if ours
    for (int r = i; r <= rtop; ++r) {
      if (r >= p.r0 && r < r1) {
        const int below_alive = (r > 0 && PTM_ALIVE_RUNG(r - 1) && alive[first[r - 1]] == 1) ? 1 : 0;
        const int self_alive = (r == i) ? 1 : 0;
        p.touch[(r - p.r0) * p.W + w] = (unsigned char)(below_alive + self_alive);
      }
      inv[perm[r]] = (unsigned short)r;                   // the row that started at perm[r] ends at r (perm permutes the touched rungs)
    }
  }
  __syncthreads();
  // -- the row moves.  The phase's net effect on the touched rungs is a permutation of rows: new row[r] =
  //    old row[perm[r]].  List every move (source slot -> destination slot, or -> boundary message for a row that
  //    leaves the shard) for move_kernel; the hole an arrival will fill is named in arr_above / arr_below.
  int* gs = lmv;
  int* gd = lmv + MVCAP;
  // source slot of the in-between row of rung r if its FIRST add_state of this step is one that saves, else -1
  auto hist_mid_src = [&](int r) -> int {
    if (r - p.r0 >= p.hist.rungs || r < p.r0 || r >= r1) return -1;
    if (!(r > 0 && PTM_ALIVE_RUNG(r - 1) && alive[first[r - 1]] == 1)) return -1;   // touched once only
    const int c = (r - p.r0) * p.W + w;
    if (p.nhist[c] % (unsigned int)p.add_every_n != 0u) return -1;
    const int s1 = midk[first[r]];
    if (s1 < p.r0 || s1 >= r1) { atomicOr(p.err, 16); return -1; }   // (the host keeps recorded rungs away from shard tops)
    return (s1 - p.r0) * p.W + w;
  };
  // the same for MAP tracking: source slot of rung r's in-between row if its log-posterior at rung r's temperature
  // beats the rung's MAP (the first of the two add_state calls sees it, chain.cc:931-934), else -1
  auto map_mid_src = [&](int r, double bmid) -> int {
    if (r - p.r0 >= p.map.rungs || r < p.r0 || r >= r1) return -1;
    if (!(r > 0 && PTM_ALIVE_RUNG(r - 1) && alive[first[r - 1]] == 1)) return -1;   // touched once only
    const int s1 = midk[first[r]];
    if (s1 < p.r0 || s1 >= r1) { atomicOr(p.err, 16); return -1; }
    const int cs = (s1 - p.r0) * p.W + w;
    const double t = bmid * p.ll[cs];
    return (p.lp[cs] + t > p.map.lpost[(r - p.r0) * p.W + w]) ? cs : -1;
  };
  for (int j = lane; j < nl; j += DECIDE_THREADS) {
    const int k = list[j];
    if (alive[k] != 1) continue;
    const int i = cand[k];
    const int rtop = (PTM_ALIVE_RUNG(i + 1) && alive[first[i + 1]] == 1) ? i : i + 1;
    const double bmid = evb ? bklo[k] : beta[i];   // rung i's temperature at this pick's add_state
    if (p.hist.rungs) {
      const int hs = hist_mid_src(i);
      if (hs >= 0) {
        const int m = atomicAdd(&cnt[1], 1);
        if (m < MVCAP) { gs[m] = hs; gd[m] = HIST_DST - ((i - p.r0) * p.W + w); if (evb) gb[m] = bmid; }
      }
    }
    if (p.map.rungs) {
      const int ms_ = map_mid_src(i, bmid);
      if (ms_ >= 0) {
        const int m = atomicAdd(&cnt[1], 1);
        if (m < MVCAP) { gs[m] = ms_; gd[m] = MAP_DST - ((i - p.r0) * p.W + w); if (evb) gb[m] = bmid; }
      }
    }
    for (int r = i; r <= rtop; ++r) {
      if (r < p.r0 || r >= r1 || perm[r] == r) continue;
      const int s = perm[r], to = inv[r];
      const int cr = (r - p.r0) * p.W + w;
      if (s >= p.r0 && s < r1) {                            // local -> local
        const int m = atomicAdd(&cnt[1], 1);
        if (m < MVCAP) { gs[m] = (s - p.r0) * p.W + w; gd[m] = cr; }
      } else {
        (s >= r1 ? p.arr_above : p.arr_below)[w] = cr;       // the hole: an arrival from the adjacent shard lands here
      }
      if (to < p.r0 || to >= r1) {                           // rung r's old row leaves the shard
        const int m = atomicAdd(&cnt[1], 1);
        if (m < MVCAP) { gs[m] = cr; gd[m] = (to >= r1) ? -1 : -2; }
        if (!((to >= r1) ? p.send_up : p.send_down)) atomicOr(p.err, 1);
      }
    }
  }
  __syncthreads();
  const int nmv = cnt[1];
  if (nmv > FCAP && nmv <= MVCAP) {   // too long for this block's registers: move_kernel takes it from here
    for (int j = lane; j < nmv; j += DECIDE_THREADS) {
      p.mv_src[(size_t)w * MVCAP + j] = gs[j];
      p.mv_dst[(size_t)w * MVCAP + j] = gd[j];
    }
    if (lane == 0) p.mv_n[w] = nmv;
    return;
  }
  if (nmv <= FCAP) {
    // ---- the moves, in place, by this block: GATHER every moved row into registers (16 lanes x 16 B = one 256-B row,
    //      one row per 16-lane group and round, up to 16 rounds), wait for all loads of all threads, then SCATTER.  With every read
    //      finished before the first write no ordering between the moves is needed (they form cycles over this ladder's
    //      own rows; other ladders' rows are never touched).
    const int DPm = p.DP, RD = DPm + ROW_EXTRA;
    for (int j = lane; j < nmv; j += DECIDE_THREADS) {
      const int dv = gd[j];
      if (dv == -1 || dv == -2) {   // a row that leaves the shard: claim its slot in the boundary message
        const int dir = dv == -1 ? 0 : 1;
        const int slot = atomicAdd(reinterpret_cast<int*>(dir ? p.send_down : p.send_up), 1);
        if (slot >= p.row_cap) { atomicOr(p.err, 4); gd[j] = -3; }
        else gd[j] = -4 - (2 * slot + dir);
      }
    }
    __syncthreads();
    const int g = lane >> 4, sub = lane & 15;
    const bool act = 2 * sub < DPm;         // DP/2 lanes of 16 carry a row (16 B each)
    const int col = act ? 2 * sub : 0;      // idle lanes re-read column 0 (harmless) so that no load is predicated
    constexpr int GR = DECIDE_THREADS / 16;   // rows per round
    constexpr int NQ = (FCAP + GR - 1) / GR;  // rounds that can hold a listed row (16 for 64 and 256 threads, 4 for the fused kernel's 1024)
    d2_t v[NQ];
    const int msrc = lane < nmv ? gs[lane] : 0;
    const int dme = lane < nmv ? gd[lane] : -3;
    // the row's scalars: a local destination's llike is already in LDS (the exchanged view of its rung), an all-uniform
    // prior's lprior is one constant -- most moves then touch no scalar line at all on the reading side
    const double sl = dme >= 0 ? llc[p.r0 + dme / p.W] : p.ll[msrc];
    const double sp = p.lp_is_const ? p.lp_const : p.lp[msrc];
    for (int hc = 0; hc < DPm; hc += 32) {   // (rows of 64 dimensions: their second 256 bytes the same way)
      const int colh = col + hc;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int j = GR * q + g;
        v[q] = *reinterpret_cast<const d2_t*>(p.x + (size_t)(j < nmv ? gs[j] : 0) * DPm + colh);   // past the list: row 0, never stored
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();   // every gather of every thread has landed before the first scatter
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int j = GR * q + g;
        const int d = j < nmv ? gd[j] : -3;
        if (d != -3 && act) {
          double* dstp;
          if (d >= 0) dstp = p.x + (size_t)d * DPm;
          else if (d <= HIST_DST) {
            const int c = HIST_DST - d;
            dstp = p.hist.x + hist_slot(p.hist, 1 + (long long)(p.nhist[c] / (unsigned int)p.add_every_n), c) * DPm;
          } else if (d <= MAP_DST) {
            dstp = p.map.x + (size_t)(MAP_DST - d) * DPm;
          } else { const int e = -d - 4; dstp = ((e & 1) ? p.send_down : p.send_up) + MSG_HDR + (size_t)(e >> 1) * RD; }
          *reinterpret_cast<d2_t*>(dstp + colh) = v[q];
        }
      }
    }
    if (lane < nmv) {
      const int d = gd[lane];
      if (d >= 0) { p.ll[d] = sl; if (!p.lp_is_const) p.lp[d] = sp; }
      else if (d <= HIST_DST) {
        const int c = HIST_DST - d;
        const long long hrow = 1 + (long long)(p.nhist[c] / (unsigned int)p.add_every_n);
        hist_scalars(p.hist, hist_slot(p.hist, hrow, c), hrow, sl, sp, p.naccept[c], p.ntries[c], p.last_type[c],
                     evb ? gb[lane] : beta[p.r0 + c / p.W]);
      } else if (d <= MAP_DST) {
        const int c = MAP_DST - d;
        const double t = (evb ? gb[lane] : beta[p.r0 + c / p.W]) * sl;
        p.map.lpost[c] = sp + t; p.map.ll[c] = sl; p.map.lp[c] = sp;
      } else if (d != -3) {
        const int e = -d - 4;
        double* row = ((e & 1) ? p.send_down : p.send_up) + MSG_HDR + (size_t)(e >> 1) * RD;
        row[DPm] = sl; row[DPm + 1] = sp; row[DPm + 2] = (double)w; row[DPm + 3] = 0.0;
      }
    }
    return;
  }
  // rare overflow of the register path (more than MVCAP moved rows in one ladder and step): the permutation
  // decomposes into disjoint closed cycles inside the shard and at most two open paths through its boundaries;
  // one lane walks each in path order, so plain loads and stores are safe.  Slow, correct.
  const int DP = p.DP;
  if (p.hist.rungs) {   // in-between rows first: they are read from rows nobody has moved yet
    for (int j = lane; j < nl; j += DECIDE_THREADS) {
      const int k = list[j];
      if (alive[k] != 1) continue;
      const int i = cand[k];
      const int hs = hist_mid_src(i);
      if (hs < 0) continue;
      const int c = (i - p.r0) * p.W + w;
      const long long hrow = 1 + (long long)(p.nhist[c] / (unsigned int)p.add_every_n);
      const size_t o = hist_slot(p.hist, hrow, c);
      for (int d = 0; d < DP; ++d) p.hist.x[o * DP + d] = p.x[(size_t)hs * DP + d];
      hist_scalars(p.hist, o, hrow, p.ll[hs], p.lp[hs], p.naccept[c], p.ntries[c], p.last_type[c], evb ? bklo[k] : beta[i]);
    }
    __syncthreads();
  }
  if (p.map.rungs) {
    for (int j = lane; j < nl; j += DECIDE_THREADS) {
      const int k = list[j];
      if (alive[k] != 1) continue;
      const int i = cand[k];
      const double bmid = evb ? bklo[k] : beta[i];
      const int cs = map_mid_src(i, bmid);
      if (cs < 0) continue;
      const int c = (i - p.r0) * p.W + w;
      const double t = bmid * p.ll[cs];
      p.map.lpost[c] = p.lp[cs] + t; p.map.ll[c] = p.ll[cs]; p.map.lp[c] = p.lp[cs];
      for (int d = 0; d < DP; ++d) p.map.x[(size_t)c * DP + d] = p.x[(size_t)cs * DP + d];
    }
    __syncthreads();
  }
  for (int j = lane; j < nl; j += DECIDE_THREADS) {
    const int k = list[j];
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
      if (departs) {
        double* sb = (to >= r1) ? p.send_up : p.send_down;
        if (!sb) continue;
        double* row = claim_row(sb, p.row_cap, DP + ROW_EXTRA, p.err);
        if (row) {
          for (int d = 0; d < DP; ++d) row[d] = X[(size_t)c0 * DP + d];
          row[DP] = p.ll[c0];
          row[DP + 1] = p.lp[c0];
          row[DP + 2] = (double)w;
          row[DP + 3] = 0.0;
        }
      }
      // one element of the row at a time (x[0..DP), llike, lprior), the head's old value carried in a register: no private
      // array -- a row-sized one would cost EVERY launch of this kernel a kilobyte of scratch per lane
      for (int el = 0; el < DP + 2; ++el) {
        auto at = [&](int c) -> double* { return el < DP ? X + (size_t)c * DP + el : (el == DP ? p.ll + c : p.lp + c); };
        const double carry = *at(c0);
        int cur = r;
        for (int guard = 0; guard <= Nt; ++guard) {
          const int src = perm[cur];
          const int cc = (cur - p.r0) * p.W + w;
          if (src == r) { *at(cc) = carry; break; }
          if (src < p.r0 || src >= r1) break;                // the hole (named above)
          const int cs = (src - p.r0) * p.W + w;
          *at(cc) = *at(cs);
          cur = src;
        }
      }
    }
  }
  if (lane == 0) p.mv_n[w] = 0;
#undef PTM_ALIVE_RUNG
}

template <int DECIDE_THREADS, bool CUT = false>
__global__ __launch_bounds__(DECIDE_THREADS) void decide_kernel(const Decide p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // walker of this block.  Consecutive workgroups go round the 8 XCDs, each with its own L2, while ll / lp / touch are
  // [rung][walker]: a 128-byte line holds 16 (128 for touch) neighbouring walkers.  Handing each XCD a contiguous range of
  // walkers keeps the ladders that share those lines on one L2.
  decide_body<DECIDE_THREADS, CUT>(p, smem, xcd_walker(blockIdx.x, p.W), p.step, p.swap_log);
}

// dynamic LDS of decide_body (host side)
inline size_t decide_lds_bytes(int Nt, int ms, int WN, bool evolve, bool evb, bool cut = false) {
  if (cut) return decide_lds_bytes(Nt, ms, WN, evolve, true) + (size_t)2 * Nt * 8;   // (the history carve + every rung's llike, lprior)
  // mirrors the carve at the top of decide_body
  const bool al = decide_aliased(ms, WN, evolve);
  const size_t msp = (size_t)((ms + 3) & ~3), WNp = (size_t)((WN + 3) & ~3);
  return (size_t)WN * 8 + (size_t)((Nt + 3) & ~3) * 2 + (size_t)((ms + 1) & ~1) * 4 * 2 + 16 + (al ? 0 : 2 * WNp * 2) + 2 * msp * 2 +
         (size_t)((ms + 7) & ~7) * 2 + (al ? 0 : (size_t)2 * MVCAP * 4) + 32 +
         (evolve ? ((size_t)Nt + 2 * ((Nt + 31) / 32) + 2 + 4 * msp) * 8 + 4 * msp * 2 + ((ms + 7) & ~7) : 0) +
         (evb ? ((size_t)(Nt > MVCAP ? Nt : MVCAP) + 5 * msp) * 8 : 0);
}


}  // namespace ptm
