// ptm_mfma_kernel.hpp -- the hot kernel of the BASELINE workload: one fused MH_chain::step (chain.cc:966-1022) for a
// Gaussian target of 17..32 dimensions with a dense or Cholesky proposal factor per rung.  The lean build assumes a
// uniform box prior and open bounds (the benchmark); the general build adds every boundary / prior / mean / 1-D move.
// One wave = 64 chains of ONE rung (W % 64 == 0).  Both matrix products run on the f64 matrix cores
// (v_mfma_f64_16x16x4_f64), the VALU is left with the random numbers:
//
//   offset (32 x 64) = T (32 x 32, the rung's factor)  x  Z (32 x 64 normals)          gaussian_prop::draw
//   S      (32 x 64) = P2 (32 x 32, lower triangle of the precision, off-diagonal doubled) x X' (32 x 64 proposals)
//   x'.S, reduced per chain                                                              the Gaussian log-likelihood
//
// Lane roles.  Lane l = 16q + j.  Per-chain scalars (llike, counters, the Metropolis test) belong to chain l of the
// wave.  Matrix data follow the MFMA operand maps (A[i][k]: lane 16k+i; B[k][n]: lane 16k+n; D[q+4r][n]: lane 16q+n,
// register r): for the four 16-chain groups g = 0..3 lane (q, j) works on chain 16g + j and on the eight dimensions
// d = q (mod 4) -- its normals are Philox blocks q and 4+q of that chain's stream (dimensions 4q..4q+3 and
// 16+4q..16+4q+3), and it holds x'[q + 4m], m = 0..7, which is why stored rows are laid out by (m/2, q, m%2)
// (row_pos).  An accumulator tile is the next product's B operand as it stands: step m of S = P2 X' needs dimension
// 4m + k on lane group k, and that is register m of the accumulators.  No lane ever moves data to another lane except
// the final 4-way reduction of x'.S (through 2 KB of LDS per wave) and two ballots.
//
// The accumulation order of every sum is the one the CPU checker states (oracle/ptm_oracle.c: ptmo_column_order,
// ptmo_llike): an f64 MFMA accumulates its four products as one fma chain in k order on top of C (measured:
// tools/probes/mfma_f64_probe.hip), so results are bit-identical to the VALU kernels and to the checker.
#pragma once
#include <type_traits>

#include "ptm_kernels.hpp"

namespace ptm {

typedef double mf_d4 __attribute__((ext_vector_type(4)));
typedef double mf_d2 __attribute__((ext_vector_type(2)));

// scheduling fence: nothing moves across it.  The kernel is written as stages ("issue loads" / "arithmetic that does
// not need them yet"); left alone, the machine scheduler sinks every load next to its first use to save registers and
// the wave then pays each HBM latency in full.
#define PTM_STAGE() __builtin_amdgcn_sched_barrier(0)

#ifndef PTM_MFMA_WAVES
#define PTM_MFMA_WAVES 3   // waves per SIMD the register budget is cut for (lean build)
#endif
#ifndef PTM_MFMA_GEN_WAVES
#define PTM_MFMA_GEN_WAVES 3   // ... of the general builds
#endif
#ifndef PTM_MFMA_GG
#define PTM_MFMA_GG 0          // 16-chain groups worked together in a pass (2: two passes of 32 chains; 1: four passes of 16 -- half the live
                               // set, four waves per SIMD); 0: by build -- 1 for the everything-general build (0.90 -> 0.70 ms), 2 for the others
                               // (the lean build loses 5 % with 1: twice the operand reads and pass overheads, and occupancy buys it nothing)
#endif
#ifndef PTM_MFMA_G1C_GG
#define PTM_MFMA_G1C_GG 2      // groups per pass of the compacted box-bounds build
#endif
#ifndef PTM_MFMA_G1C_WAVES
#define PTM_MFMA_G1C_WAVES 2   // (2: 193-198 registers, no spills -- at 3 the compacted build spilled 13-22 and its sweep-in-step of the default Gaussian recipe took 2.40 instead of 2.01 ms)
#endif
#ifndef PTM_MFMA_PRIO
#define PTM_MFMA_PRIO 2        // 1: a wave raises its issue priority over its matrix blocks (measured: nothing); 2: over its vector (draw)
                               // blocks instead (1.3-3 % on the benchmark sizes)
#endif
#ifndef PTM_MFMA_GEN_PERSIST
#define PTM_MFMA_GEN_PERSIST 0   // 1: the general builds walk several tiles per block too (measured: 15-25 % slower -- spills)
#endif
// HIST: the engine keeps a history.  GEN > 0: the general state space / prior / target -- boundaries of any kind
// (boundary::enforce, states.cc:11-58), mixed priors (probability_function.cc:281-304), a mean, one-dimensional moves
// (proposal_distribution.hh:196-206).  Both are compiled apart: the hot build (false, false) carries none of it.
// All of the general work is per dimension, so it runs in the accumulator layout as it stands: a lane enforces and
// prices its own eight dimensions of each chain, and the chain's four lanes meet in two more LDS reductions / ballots.
// EV: a GEN 1 build that reads a per-chain beta (evolving ladders; the GEN 2 build always can)
// CPT (lean and GEN 1 builds without history): the sweep visits the moving chains alone, through the per-rung lists of partition_kernel
// (ptm_kernels.hpp) -- tiles of 256 LISTED walkers of one rung, enumerated rung by rung; lane l of a wave works for the l-th
// listed walker of its group instead of walker w0 + l.
template <int KIND, bool HIST, int GEN, bool EV = false, bool CPT = false>   // GEN: 0 lean, 1 box boundaries + uniform prior (+ mean, 1-D moves, mixtures), 2 everything,
                                                                            //      3 box boundaries + uniform prior and nothing else (GEN 1 without what it only carries)
__global__ __launch_bounds__(256, (GEN == 0 ? PTM_MFMA_WAVES : ((PTM_MFMA_GG == 0 && GEN == 2 && KIND == KIND_LOWER) ? 4 : ((GEN == 1 && CPT) ? PTM_MFMA_G1C_WAVES : PTM_MFMA_GEN_WAVES)))) void sweep_mfma32_kernel(const Dev p) {
  constexpr bool PERSIST = GEN == 0 || PTM_MFMA_GEN_PERSIST != 0 || CPT;   // (a compacted sweep walks its tiles: an idle tile must cost nothing)
  static_assert(!CPT || ((GEN <= 1 || GEN == 3) && !HIST), "the compacted sweep exists for the lean and the box-bounds builds, without history");
  constexpr bool GENX = GEN == 1 || GEN == 2;   // the builds that carry a mean, one-dimensional moves and scale mixtures
  constexpr int DP = 32;
  constexpr int GG = PTM_MFMA_GG ? PTM_MFMA_GG : (GEN == 2 ? 1 : (((GEN == 1 || GEN == 3) && CPT) ? PTM_MFMA_G1C_GG : 2)), NP = 4 / GG, PL = 16 * GG;   // groups per pass, passes per tile, chains (= stage-5 lanes) per pass
  constexpr bool LOW = KIND == KIND_LOWER;
  static_assert(GEN != 2 || GG == 1, "the mixture scales share the GEN 2 build's red2 slots: its prior products must stay below index 64");
  // LDS: [2560] Box-Muller tables | [12][64] precision tiles | [64] prior box (all shared by the block's waves) |
  //      128 doubles per wave
  extern __shared__ __attribute__((aligned(16))) double lds_all[];
  double* ptile = lds_all + BM_TABLE_DOUBLES;   // tile (row tile 1, step m) at m*64, m = 0..7; (row tile 0, step m) at (8+m)*64, m = 0..3
  double* lbox = ptile + 12 * 64;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  double* red = lbox + 64 + wave * 128;
  // GEN: per-dimension tables of the state space and the prior, natural index (after the waves' reduction slots)
  double* gtab = lbox + 64 + 4 * 128;                   // bmin | bmax | plo | phi | pcoef | mean, 32 each
  int* gint = reinterpret_cast<int*>(gtab + 6 * 32);    // blo | bhi | ptype, 32 each
  double* red2 = reinterpret_cast<double*>(gint + 3 * 32) + wave * 128;   // the prior's partial products
  // the scales of the rung's mixture members (at most 64), this wave's copy: in the half of red2 the prior's products never reach (they are
  // GEN 2's, which works one group per pass: indices below 64), or all of it in the builds without them
  double* mixs = red2 + (GEN == 2 ? 64 : 0);
  // all boundaries open or `limit` (the usual case): enforcing is a box test -- lower | upper limits in row layout
  double* ebox = reinterpret_cast<double*>(gint + 3 * 32) + 4 * 128;
  const int q = l >> 4, j = l & 15;
  const double* pimg = ptile + l;
  const mf_d2* box = reinterpret_cast<const mf_d2*>(lbox) + q;   // lo piece t at 4t, hi piece t at 16 + 4t (row layout)
  const mf_d2* ebx = reinterpret_cast<const mf_d2*>(ebox) + q;

  // ---- the block's tables: staged ONCE.  The grid is persistent -- a block per resident slot of the chip, each walking
  //      the launch's 256-chain tiles with stride gridDim.x -- so the 26 KB of tables (Box-Muller, precision tiles, box) are
  //      read once per resident block instead of once per 256 chains (65536 times per sweep of the benchmark).
  //      The general builds keep one tile per block (PERSIST false): their live sets are larger, and there the tables' loads
  //      travel together with the tile's first rows / scalars and go to LDS only when those have been asked for.
  bm_d2 st_bm[BM_TABLE_DOUBLES / 512];
#pragma unroll
  for (int t = 0; t < BM_TABLE_DOUBLES / 512; ++t) st_bm[t] = reinterpret_cast<const bm_d2*>(BM_TABLE)[threadIdx.x + 256 * t];
  double st_p[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int e = threadIdx.x + 256 * t, tile = e >> 6;
    const int src = tile < 8 ? (tile * 2 + 1) : ((tile - 8) * 2 + 0);
    st_p[t] = p.P2_tiles[src * 64 + (e & 63)];
  }
  const double st_box = p.box_row[threadIdx.x & 63];
  double st_g[6] = {0, 0, 0, 0, 0, 0};
  int st_i[3] = {0, 0, 0};
  if (GEN && threadIdx.x < 32) {
    const int d = threadIdx.x;
    st_g[0] = p.bmin[d]; st_g[1] = p.bmax[d]; st_g[2] = p.plo[d]; st_g[3] = p.phi[d]; st_g[4] = p.pcoef[d];
    st_g[5] = p.has_mean ? p.mean[d] : 0.0;
    st_i[0] = p.blo[d]; st_i[1] = p.bhi[d]; st_i[2] = p.ptype[d];
  }
  auto stage_tables = [&]() {
#pragma unroll
    for (int t = 0; t < BM_TABLE_DOUBLES / 512; ++t) reinterpret_cast<bm_d2*>(lds_all)[threadIdx.x + 256 * t] = st_bm[t];
#pragma unroll
    for (int t = 0; t < 3; ++t) ptile[threadIdx.x + 256 * t] = st_p[t];
    if (threadIdx.x < 64) lbox[threadIdx.x] = st_box;
    if (GEN && threadIdx.x < 32) {
#pragma unroll
      for (int t = 0; t < 6; ++t) gtab[32 * t + threadIdx.x] = st_g[t];
#pragma unroll
      for (int t = 0; t < 3; ++t) gint[32 * t + threadIdx.x] = st_i[t];
      const int pos = row_pos<32>(threadIdx.x);
      ebox[pos] = st_i[0] == B_LIMIT ? st_g[0] : -__builtin_inf();
      ebox[32 + pos] = st_i[1] == B_LIMIT ? st_g[1] : __builtin_inf();
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  if (PERSIST) stage_tables();

  // CPT: the launch's rungs, their listed-walker counts as an inclusive prefix of 256-walker tiles (LDS, after everything else)
  const int rung0 = p.c_begin / p.W, nrung = (p.c_end - p.c_begin) / p.W;
  int* tpre = reinterpret_cast<int*>(GEN == 0 ? lbox + 64 + 4 * 128 : ebox + 64);   // [nrung] (CPT; the host sizes the LDS for it; the lean build has no general tables)
  int ntiles = (p.c_end - p.c_begin + 255) >> 8;   // 256-chain tiles of this launch (ranges are multiples of 64)
  if constexpr (CPT) {
    // inclusive scan of ceil(count / 256) over the rungs: each thread sums a contiguous chunk, one wave scans the chunk totals
    const int per = (nrung + 255) / 256;
    int loc = 0;
    for (int k = 0; k < per; ++k) {
      const int r = threadIdx.x * per + k;
      if (r < nrung) { loc += (p.ccnt[rung0 + r] + 255) >> 8; tpre[r] = loc; }
    }
    int* tsum = tpre + nrung;                       // [256] chunk totals -> exclusive offsets
    tsum[threadIdx.x] = loc;
    __syncthreads();
    if (threadIdx.x == 0) { int run = 0; for (int t = 0; t < 256; ++t) { const int v = tsum[t]; tsum[t] = run; run += v; } tsum[256] = run; }
    __syncthreads();
    const int off = tsum[threadIdx.x];
    for (int k = 0; k < per; ++k) {
      const int r = threadIdx.x * per + k;
      if (r < nrung) tpre[r] += off;
    }
    __syncthreads();
    ntiles = tsum[256];
  }
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  int c0, rl, w0, nact = 64, lbase = 0;
  bool live = true;
  if constexpr (CPT) {
    // the rung of compacted tile `tile`: first r with tpre[r] > tile (wave-uniform binary search in LDS)
    int lo = 0, hi = nrung - 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (tpre[mid] > tile) hi = mid; else lo = mid + 1; }
    const int r = __builtin_amdgcn_readfirstlane(lo);
    const int kb = tile - (r ? tpre[r - 1] : 0);
    rl = rung0 + r;
    const int start = (kb * 4 + wave) * 64, cnt = p.ccnt[rl];
    if (start >= cnt) continue;                        // (only wave-level barriers below)
    nact = cnt - start < 64 ? cnt - start : 64;
    lbase = rl * p.W + start;                           // this wave's slice of the rung's list
    c0 = rl * p.W; w0 = 0;
  } else {
    c0 = p.c_begin + (tile * 4 + wave) * 64;   // first chain of the wave
    if (c0 >= p.c_end) {
      if (PERSIST) continue;                               // (only wave-level barriers below)
      c0 = p.c_begin; live = false;                        // one tile per block: a wave past the end still helps to stage the tables
    }
    rl = __builtin_amdgcn_readfirstlane(c0 / p.W);
    w0 = c0 - rl * p.W;
  }
  const int c0s = c0;
  const int rg = p.r0 + rl;
  // walker of wave-chain index i (0..63): listed (CPT; lanes past the list shadow its last entry and write nothing) or in place
  auto walker_of = [&](int i) -> int {
    if constexpr (CPT) return p.cidx[lbase + (i < nact ? i : nact - 1)];
    else return w0 + i;
  };
  const int wl = walker_of(l);
  const bool dead = CPT && l >= nact;
  const int c = CPT ? rl * p.W + wl : c0s + l;   // "my" chain for the per-chain work
  int wq[4];   // the walkers this lane works for in the matrix products: chain 16 g + j of the wave, g = 0..3
#pragma unroll
  for (int g = 0; g < 4; ++g) wq[g] = walker_of(16 * g + j);
  const double* timg = p.prop_tiles + (size_t)rl * (16 * 64) + l;   // tile t = (half*4 + slot)*2 + row tile

  // A tile's 64 chains are worked in two passes of two 16-chain groups (g = 2 gp + gg): every live set is halved.
  mf_d2 rowv[GG][4];   // the pass's rows, asked for one pass ahead
  mf_d2* rowp[GG];
  auto ask_rows = [&](int gpp, mf_d2 (&rv)[GG][4], mf_d2* (&rp)[GG]) {
#pragma unroll
    for (int gg = 0; gg < GG; ++gg) {
      rp[gg] = reinterpret_cast<mf_d2*>(p.x + ((size_t)rl * p.W + wq[GG * gpp + gg]) * DP) + q;   // piece t at [4t]
#pragma unroll
      for (int t = 0; t < 4; ++t) rv[gg][t] = rp[gg][4 * t];
    }
  };
  ask_rows(0, rowv, rowp);
  // per-chain scalars: used at the very end
  const int tc = CPT ? 0 : p.touch[c];  // > 0: the rung took part in that many exchange attempts => no MH move this step
  const double ll = p.ll[c], lp = p.lp[c];
  const int ntries0 = p.ntries[c], naccept0 = p.naccept[c];
  const unsigned int nhist0 = CPT ? 0u : p.nhist[c];
  // (a per-chain beta in the plain GEN 1 build would cost its fixed-ladder users 4 %: evolving ladders have their own)
  constexpr bool PERCHAIN = GEN == 2 || EV;
  // (the lean build of evolving ladders takes its temperature from the ladder-major image the exchange kernel keeps: the chain-indexed
  //  one is then brought up to date only when somebody else asks for it)
  const double beta = (GEN == 0 && EV) ? p.beta_w[(size_t)wl * p.Nt + rg] : ((PERCHAIN && p.betaC) ? p.betaC[c] : as_c(p.beta)[rg]);
  // log of the chain's accept uniform (block 0 of its stream): drawn here once for all 64 chains -- the Metropolis test
  // itself runs per pass on half the lanes, and this is its expensive part.  (The reference draws the uniform only when
  // logH < 0, chain.cc:998; a counter-based stream makes the draw free of side effects, so drawing it always is the same.)
  const u32x4 o0 = draw_block(p.seed, TAG_MH, (uint32_t)(wl + p.w_off) * (uint32_t)p.Nt + (uint32_t)rg, p.step, 0);
  const double log_u = dlog_u01(o0.v0);
  // one-dimensional move of "my" chain (proposal_distribution.hh:196-206): its axis, or -1
  // ... and its member of a scale mixture (proposal_distribution_set::draw, proposal_distribution.cc:99-129), packed into ONE register:
  // (axis + 1) | member << 8.  The member's scale is looked up where it is used, in this wave's copy of the rung's scales in LDS
  // (as axis / member / scale of "my" chain and of the pass's chains these were ten registers of a kernel at its register cap)
  int my_meta = 0;
  if (GENX) {
    double f = as_c(p.onedfrac)[rl];
    int kmix = 0, axis1 = 0;
    if (p.mix_K > 0) {
      cdp mx = as_c(p.mix) + (size_t)rl * p.mix_K * 3;
      const double xs = p.mix_K > 1 ? u01(o0.v3) : 0.0;
      kmix = p.mix_K - 1;
      for (int k = p.mix_K - 2; k >= 0; --k)
        if (xs < mx[3 * k]) kmix = k;
      f = mx[3 * kmix + 2];
      if (l < p.mix_K) mixs[l] = mx[3 * l + 1];
    }
    if (p.any_oned && f > 0 && u01(o0.v1) < f) axis1 = 1 + (int)(p.D * u01(o0.v2));
    my_meta = axis1 | (kmix << 8);
  }

  if (!PERSIST) {   // the tables go to LDS behind the tile's first loads; the block meets once
    stage_tables();
    if (!live) break;
  }
  // (a generic lambda called with compile-time pass numbers: `#pragma unroll` gives up on a body of this size in the
  //  general build, and a pass number known only at run time costs dynamic register indexing)
  auto pass = [&](auto gpc) {
    constexpr int gp = decltype(gpc)::value;
    const int qd = opaque_copy(q);   // (the draws' counter word: not to be pre-multiplied outside the tile loop)
    if (PTM_MFMA_PRIO == 2) __builtin_amdgcn_s_setprio(2);
    // ---- stage 1: ask for the first half's factor tiles (L2-resident; behind them the first draw)
    double ta[4][2];
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      ta[sl][0] = timg[((0 * 4 + sl) * 2 + 0) * 64];
      ta[sl][1] = timg[((0 * 4 + sl) * 2 + 1) * 64];
    }
    PTM_STAGE();
    // ---- stage 2: T x Z, one 16-column half at a time (the half's normals: one Philox block per chain)
    mf_d4 acc[GG][2];
    double tb[4][2];
    int meta[GG];        // GEN: axis + 1 | member << 8 of chain (GG gp + gg, j), from its own lane
#pragma unroll
    for (int gg = 0; gg < GG; ++gg) {
      acc[gg][0] = mf_d4{0.0, 0.0, 0.0, 0.0};
      acc[gg][1] = mf_d4{0.0, 0.0, 0.0, 0.0};
      meta[gg] = 0;
    }
    if (GENX && (p.any_oned || p.mix_K > 0)) {
#pragma unroll
      for (int gg = 0; gg < GG; ++gg) meta[gg] = __builtin_amdgcn_ds_bpermute(4 * (PL * gp + 16 * gg + j), my_meta);
    }
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      double z[GG][4];
#pragma unroll
      for (int gg = 0; gg < GG; ++gg) {
        const uint32_t stream = (uint32_t)(wq[GG * gp + gg] + p.w_off) * (uint32_t)p.Nt + (uint32_t)rg;
        const u32x4 o = draw_block(p.seed, TAG_MH, stream, p.step, (uint32_t)(1 + 4 * hb + qd));
        boxmuller(o.v0, o.v1, (const double*)lds_all, z[gg][0], z[gg][1]);
        boxmuller(o.v2, o.v3, (const double*)lds_all, z[gg][2], z[gg][3]);
        if (GENX && (meta[gg] & 0xFF) != 0) {   // one-dimensional move: every other normal is dropped (proposal_distribution.hh:197-205)
#pragma unroll
          for (int sl = 0; sl < 4; ++sl)
            if (16 * hb + 4 * q + sl + 1 != (meta[gg] & 0xFF)) z[gg][sl] = 0.0;
        }
        PTM_STAGE();   // one chain's draw at a time: the temporaries of two interleaved draws cost 40 registers
      }
      if (PTM_MFMA_PRIO == 1) __builtin_amdgcn_s_setprio(2);
      if (PTM_MFMA_PRIO == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          if (LOW && hb == 1 && rt == 0) continue;   // columns >= 16 never reach rows < 16
          const double a = hb == 0 ? ta[sl][rt] : tb[sl][rt];
#pragma unroll
          for (int gg = 0; gg < GG; ++gg) {
            acc[gg][rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, z[gg][sl], acc[gg][rt], 0, 0, 0);
          }
        }
      }
      if (PTM_MFMA_PRIO == 1) __builtin_amdgcn_s_setprio(0);
      if (PTM_MFMA_PRIO == 2) __builtin_amdgcn_s_setprio(2);
      if (hb == 0) {   // ask for the second half's tiles while the second half's normals are drawn
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
          tb[sl][0] = LOW ? 0.0 : timg[((1 * 4 + sl) * 2 + 0) * 64];
          tb[sl][1] = timg[((1 * 4 + sl) * 2 + 1) * 64];
        }
      }
      PTM_STAGE();
    }
    // ---- stage 3: x' = x + offset (state::add, states.cc:205-214); boundaries; the prior
    double xp[GG][8];
    uint64_t inbox = 0;     // bit 16 gg + j: chain (GG gp + gg, j) is inside the box of an all-uniform prior
    uint64_t validb = ~0ull; // GEN: bit 16 gg + j: the chain's state is valid (stateSpace::enforce, states.cc:86-102)
    const bool boxed = GEN != 2 || p.all_uniform;
#pragma unroll
    for (int gg = 0; gg < GG; ++gg) {
      bool ok = true, vok = true;
      double pp = 1.0;      // GEN: this lane's partial product of the prior's factors (its dimensions, ascending)
      double msc = 1.0;     // GEN: the scale of the chain's mixture member
      if (GENX && p.mix_K > 0) msc = mixs[meta[gg] >> 8];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const mf_d2 v = rowv[gg][t], lo = box[4 * t], hi = box[16 + 4 * t];
        const int m = 2 * t;   // registers m, m+1 <-> dimensions q + 4m, q + 4m + 4
        if (GENX && p.mix_K > 0) {   // the member is scale_k times the rung's factor
          xp[gg][m] = v.x + msc * acc[gg][m >> 2][m & 3];
          xp[gg][m + 1] = v.y + msc * acc[gg][(m + 1) >> 2][(m + 1) & 3];
        } else {
          xp[gg][m] = v.x + acc[gg][m >> 2][m & 3];
          xp[gg][m + 1] = v.y + acc[gg][(m + 1) >> 2][(m + 1) & 3];
        }
        if (GEN && p.has_bounds && p.bounds_box) {   // boundary::enforce for open / limit sides (states.cc:53-55)
          const mf_d2 el = ebx[4 * t], eh = ebx[16 + 4 * t];
          vok = vok & !(xp[gg][m] < el.x) & !(xp[gg][m] > eh.x) & !(xp[gg][m + 1] < el.y) & !(xp[gg][m + 1] > eh.y);
        }
        if (GEN == 2) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int d = q + 4 * (m + u);
            if (p.has_bounds && !p.bounds_box)
              vok = vok & boundary_enforce(gint[d], gint[32 + d], gtab[d], gtab[32 + d], xp[gg][m + u]);
            if (!boxed) pp *= prior_pdf(gint[64 + d], gtab[64 + d], gtab[96 + d], gtab[128 + d], xp[gg][m + u]);
          }
        }
        ok = ok & !(xp[gg][m] < lo.x) & !(xp[gg][m] > hi.x) & !(xp[gg][m + 1] < lo.y) & !(xp[gg][m + 1] > hi.y);
      }
      uint64_t b = __builtin_amdgcn_ballot_w64(ok);
      b &= b >> 32;
      b &= b >> 16;                                  // bit jj: all four lanes (q, jj) of chain (g, jj) are inside
      inbox |= (b & 0xFFFFull) << (16 * gg);
      if (GEN) {
        uint64_t vb = __builtin_amdgcn_ballot_w64(vok);
        vb &= vb >> 32;
        vb &= vb >> 16;
        validb = (validb & ~(0xFFFFull << (16 * gg))) | ((vb & 0xFFFFull) << (16 * gg));
        if (GEN == 2) red2[(gg * 4 + q) * 16 + j] = pp;
      }
    }
    PTM_STAGE();
    // the second pass's rows are asked for here: the registers of the first pass's rows have just been freed
    mf_d2 rown[GG][4];
    mf_d2* rowpn[GG];
#pragma unroll
    for (int gg = 0; gg < GG; ++gg) rowpn[gg] = rowp[gg];
    if (gp + 1 < NP) ask_rows(gp + 1, rown, rowpn);
    PTM_STAGE();
    // ---- stage 4: S = P2 x Y, Y = X' (- mean), and the four partial dot products of each chain
    if (PTM_MFMA_PRIO == 1) __builtin_amdgcn_s_setprio(2);
    if (PTM_MFMA_PRIO == 2) __builtin_amdgcn_s_setprio(0);
    auto yv = [&](int gg, int m) -> double { return (GENX && p.has_mean) ? xp[gg][m] - gtab[160 + q + 4 * m] : xp[gg][m]; };
    mf_d4 sacc[GG][2];
#pragma unroll
    for (int gg = 0; gg < GG; ++gg) {
      sacc[gg][0] = mf_d4{0.0, 0.0, 0.0, 0.0};
      sacc[gg][1] = mf_d4{0.0, 0.0, 0.0, 0.0};
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        if (rt == 0 && m >= 4) continue;
        const double a = pimg[(rt ? m : 8 + m) * 64];
#pragma unroll
        for (int gg = 0; gg < GG; ++gg) {
          sacc[gg][rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, yv(gg, m), sacc[gg][rt], 0, 0, 0);
        }
      }
    }
#define PTM_SACC(gg, m) sacc[gg][(m) >> 2][(m) & 3]
    if (PTM_MFMA_PRIO == 1) __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int gg = 0; gg < GG; ++gg) {
      double pq = 0.0;
#pragma unroll
      for (int m = 0; m < 8; ++m) pq = __builtin_fma(yv(gg, m), PTM_SACC(gg, m), pq);
      // chain (g, j)'s four partial sums sit on lanes (0..3, j): hand them to lane 16 g + j through this wave's LDS
      red[(gg * 4 + q) * 16 + j] = pq;
    }
    __builtin_amdgcn_wave_barrier();
    // ---- stage 5, lanes PL gp .. PL gp + PL - 1 (chain = lane): Metropolis test and add_state counters
    //      (chain.cc:973-1019, 916-949)
    bool accept = false;
    int hrow = -1;   // >= 0: this add_state call saves a history row (chain.cc:935-946), the ring row index
    bool mapw = false;   // this add_state call sets a new MAP (chain.cc:931-934): the row is copied below
    const bool hist_on = HIST && rl < p.hist.rungs;
    const bool map_on = HIST && rl < p.map.rungs;
    if (l / PL == gp && !dead) {
      const double* mine = red + ((q % GG) * 4) * 16 + j;
      const double quad = ((mine[0] + mine[16]) + mine[32]) + mine[48];
      if (tc) {
        p.nhist[c] = nhist0 + (unsigned int)tc;
        p.touch[c] = 0;
        // history: the last of these adds saw the row as it is now (the exchange kernels save an earlier one of two)
        const unsigned int a = nhist0 + (unsigned int)tc - 1u;
        if (hist_on && a % (unsigned int)p.add_every_n == 0u) {
          hrow = 1 + (int)(a / (unsigned int)p.add_every_n);
          hist_scalars(p.hist, hist_slot(p.hist, hrow, c), hrow, ll, lp, naccept0, ntries0, p.last_type[c],
                       (PERCHAIN && p.beta_add) ? p.beta_add[c] : beta);
        }
        if (map_on) {   // at the temperature the rung had at that add (evolving ladders: between two pries of the step)
          const double tb = ((PERCHAIN && p.beta_add) ? p.beta_add[c] : beta) * ll;
          mapw = map_try(p.map, c, lp + tb, ll, lp);
        }
      } else {
        const double bl = beta * ll;
        const double cur_lpost = lp + bl;
        const double oldlprior = cur_lpost - bl;  // chain.cc:973
        const bool in = ((inbox >> (l & (PL - 1))) & 1ull) != 0;
        // Q9: state::add builds on an enforced zero state -- an origin outside a `limit` bound invalidates every proposal
        const bool valid = !GEN || (p.origin_valid != 0 && ((validb >> (l & (PL - 1))) & 1ull) != 0);
        double newlprior = in ? p.lprior_const : -__builtin_inf();
        if (GEN == 2 && !p.all_uniform) {   // log of the product of the factors, ((p0 p1) p2) p3 (probability_function.hh:59)
          const double* pm = red2 + ((q % GG) * 4) * 16 + j;
          newlprior = dlog(((pm[0] * pm[16]) * pm[32]) * pm[48]);
        }
        if (!valid) newlprior = -__builtin_inf();
        const bool want_like = valid && (newlprior > -1e200 || newlprior - oldlprior > p.min_prior);  // chain.cc:980 (Q1)
        double newlike = p.like0 - 0.5 * quad;
        double newlpost = newlike * beta + newlprior;
        if (!want_like) newlike = newlpost = -__builtin_inf();
        const double logH = newlpost - cur_lpost;
        accept = valid;
        if (accept && logH < 0) accept = log_u < logH;  // chain.cc:998-1001 (NaN stays accepted)
        int type = (GENX && (my_meta & 0xFF) != 0) ? 1 : 0;
        if (GENX && p.mix_K > 0) type = (my_meta >> 8) + 10 * type;   // proposal_distribution.cc:117
        p.ntries[c] = ntries0 + 1;
        if (!CPT) p.nhist[c] = nhist0 + 1u;   // (compacted: the engine counts the step for everybody, ptm_aux_kernels.hpp)
        if (hist_on && nhist0 % (unsigned int)p.add_every_n == 0u) {
          hrow = 1 + (int)(nhist0 / (unsigned int)p.add_every_n);
          const size_t o = hist_slot(p.hist, hrow, c);
          if (accept) hist_scalars(p.hist, o, hrow, newlike, newlprior, naccept0 + 1, ntries0 + 1, type, beta);
          else hist_scalars(p.hist, o, hrow, ll, lp, naccept0, ntries0 + 1, p.last_type[c], beta);
        }
        if (map_on && accept) mapw = map_try(p.map, c, newlpost, newlike, newlprior);
        // an evolving ladder: the state that stays is added at a NEW temperature and may beat the MAP with it
        else if (map_on && PERCHAIN && p.betaC) mapw = map_try(p.map, c, cur_lpost, ll, lp);
        if (accept) {
          p.naccept[c] = naccept0 + 1;
          p.last_type[c] = type;
          p.ll[c] = newlike;
          p.lp[c] = newlprior;
        }
      }
    }
    // ---- accepted proposals replace their rows; each of a chain's four lanes writes its 64 bytes
    const uint64_t acc_bits = __builtin_amdgcn_ballot_w64(accept) >> (PL * gp);
#pragma unroll
    for (int gg = 0; gg < GG; ++gg) {
      if ((acc_bits >> (16 * gg + j)) & 1ull) {
#pragma unroll
        for (int t = 0; t < 4; ++t) rowp[gg][4 * t] = mf_d2{xp[gg][2 * t], xp[gg][2 * t + 1]};
      }
    }
    // ---- history rows (rare: every add_every_N-th add of the recorded rungs): the chain's four lanes copy the state
    //      the add saw -- the proposal if it was accepted, else the row as it stands in memory
    if (hist_on) {
      const uint64_t rec_bits = __builtin_amdgcn_ballot_w64(hrow >= 0) >> (PL * gp);
      if (rec_bits & ((1ull << PL) - 1ull)) {
#pragma unroll
        for (int gg = 0; gg < GG; ++gg) {
          const int src_lane = PL * gp + 16 * gg + j;
          const int hr = __builtin_amdgcn_ds_bpermute(4 * src_lane, hrow);
          if ((rec_bits >> (16 * gg + j)) & 1ull) {
            const int cg = c0 + src_lane;
            mf_d2* dst = reinterpret_cast<mf_d2*>(p.hist.x + hist_slot(p.hist, hr, cg) * DP) + q;
            const bool took = (acc_bits >> (16 * gg + j)) & 1ull;
#pragma unroll
            for (int t = 0; t < 4; ++t) dst[4 * t] = took ? mf_d2{xp[gg][2 * t], xp[gg][2 * t + 1]} : rowp[gg][4 * t];
          }
        }
      }
    }
    if (map_on) {   // the new MAP's row: the proposal if it was accepted, else (an exchanged rung) the row in memory
      const uint64_t mb = __builtin_amdgcn_ballot_w64(mapw) >> (PL * gp);
      if (mb & ((1ull << PL) - 1ull)) {
#pragma unroll
        for (int gg = 0; gg < GG; ++gg) {
          if ((mb >> (16 * gg + j)) & 1ull) {
            const int cg = c0 + PL * gp + 16 * gg + j;
            mf_d2* dst = reinterpret_cast<mf_d2*>(p.map.x + (size_t)cg * DP) + q;
            const bool took = (acc_bits >> (16 * gg + j)) & 1ull;
#pragma unroll
            for (int t = 0; t < 4; ++t) dst[4 * t] = took ? mf_d2{xp[gg][2 * t], xp[gg][2 * t + 1]} : rowp[gg][4 * t];
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();   // the next pass reuses the LDS slots
    if (gp + 1 < NP) {
#pragma unroll
      for (int gg = 0; gg < GG; ++gg) {
        rowp[gg] = rowpn[gg];
#pragma unroll
        for (int t = 0; t < 4; ++t) rowv[gg][t] = rown[gg][t];
      }
    }
  };
  pass(std::integral_constant<int, 0>{});
  pass(std::integral_constant<int, 1>{});
  if constexpr (NP == 4) {
    pass(std::integral_constant<int, 2>{});
    pass(std::integral_constant<int, 3>{});
  }
  if (!PERSIST) break;   // (one tile per block: the launch gives every tile its own block)
  }   // tiles
}
#undef PTM_STAGE
#undef PTM_SACC

}  // namespace ptm
