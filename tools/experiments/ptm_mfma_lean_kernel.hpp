// ptm_mfma_lean_kernel.hpp -- EXPERIMENT, off by default (PTM_LEAN_PIPE=1 engages it): measured 8 % SLOWER than the lean build of
// ptm_mfma_kernel.hpp on the benchmark sizes (1.73 against 1.60 ms per sweep-in-step; round 2) -- kept
// because the result is the evidence: spreading a wave's vector work behind its own matrix instructions buys nothing on
// gfx950, the three waves of a SIMD already overlap the two pipes as far as the hardware lets them.
//
// The BASELINE workload's own build of the MFMA sweep (ptm_mfma_kernel.hpp states the method:
// lane roles, operand maps, accumulation order, row layout -- all the same here): a Gaussian target of 17..32 dimensions,
// dense or Cholesky factor per rung, uniform box prior, open bounds, no history / MAP.  What differs is the SCHEDULE.
//
// The f64 matrix pipe (64 cycles per v_mfma_f64_16x16x4_f64) and the vector ALU run side by side -- a wave can issue vector
// work behind its own matrix instruction, and another wave's vector work goes on at ~60 % of its rate under a saturated
// matrix pipe (tools/probes/mfma_f64_4x4_probe.hip, valu_cost_probe.hip) -- but a kernel written as "draw the normals, THEN
// multiply" leaves that to the chance that its three waves per SIMD are in different phases.  Here every matrix block of a
// wave carries the NEXT block's random numbers behind it:
//
//     pass g (16 chains; four per tile):   A  T x Z, columns 0..15  ||  Philox + Box-Muller of columns 16..31
//                                          B  T x Z, columns 16..31
//                                          C  x' = x + offset, box test
//                                          D  P2 x X'               ||  Philox + Box-Muller of pass g+1's columns 0..15
//                                          E  reduction, Metropolis test, counters, accepted rows
//
// The interleave is written out: a draw is cut into slices (Philox rounds, the radius and the angle half of each Box-Muller
// pair), one slice behind each matrix instruction, scheduling fences in between (the compiler's own group barriers are
// dropped when the mixed order costs registers).
// One 16-chain group per pass instead of two halves every live set again (145 VGPRs at three waves per SIMD; cut to 128 for
// four it spills and loses another 9 %).
#pragma once
#include <type_traits>
#include <utility>

#include "ptm_mfma_kernel.hpp"

namespace ptm {

#ifndef PTM_LEAN_WAVES
#define PTM_LEAN_WAVES 3      // waves per SIMD the register budget is cut for
#endif
template <int... I, class F>
__device__ __forceinline__ void lean_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void lean_for(F&& f) { lean_for_impl(std::make_integer_sequence<int, N>{}, f); }
struct lean_draw { philox_state ps; double r; };
// the value is needed HERE: scheduling fences stop the scheduler, not the passes that sink a computation towards its use
template <class T>
__device__ __forceinline__ void lean_pin(T& x) { asm volatile("" : "+v"(x)); }
#define PTM_STAGE() __builtin_amdgcn_sched_barrier(0)

template <int KIND, bool CPT>
__global__ __launch_bounds__(256, PTM_LEAN_WAVES) void sweep_mfma32_lean_kernel(const Dev p) {
  constexpr int DP = 32;
  constexpr bool LOW = KIND == KIND_LOWER;
  // LDS: [2560] Box-Muller tables | [12][64] precision tiles | [64] prior box | 64 doubles per wave | (CPT) tile prefix
  extern __shared__ __attribute__((aligned(16))) double lds_all[];
  double* ptile = lds_all + BM_TABLE_DOUBLES;   // tile (row tile 1, step m) at m*64, m = 0..7; (row tile 0, step m) at (8+m)*64, m = 0..3
  double* lbox = ptile + 12 * 64;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  double* red = lbox + 64 + wave * 64;
  const int q = l >> 4, j = l & 15;
  const double* pimg = ptile + l;
  const mf_d2* box = reinterpret_cast<const mf_d2*>(lbox) + q;   // lo piece t at 4t, hi piece t at 16 + 4t (row layout)

  // ---- the block's tables, staged once: the grid is persistent (a block per resident slot, tiles walked with stride gridDim.x)
  {
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
#pragma unroll
    for (int t = 0; t < BM_TABLE_DOUBLES / 512; ++t) reinterpret_cast<bm_d2*>(lds_all)[threadIdx.x + 256 * t] = st_bm[t];
#pragma unroll
    for (int t = 0; t < 3; ++t) ptile[threadIdx.x + 256 * t] = st_p[t];
    if (threadIdx.x < 64) lbox[threadIdx.x] = st_box;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }

  // CPT: the launch's rungs, their listed-walker counts as an inclusive prefix of 256-walker tiles
  const int rung0 = p.c_begin / p.W, nrung = (p.c_end - p.c_begin) / p.W;
  int* tpre = reinterpret_cast<int*>(lbox + 64 + 4 * 64);   // [nrung] + [257] (the host sizes the LDS for it)
  int ntiles = (p.c_end - p.c_begin + 255) >> 8;
  if constexpr (CPT) {
    const int per = (nrung + 255) / 256;
    int loc = 0;
    for (int k = 0; k < per; ++k) {
      const int r = threadIdx.x * per + k;
      if (r < nrung) { loc += (p.ccnt[rung0 + r] + 255) >> 8; tpre[r] = loc; }
    }
    int* tsum = tpre + nrung;
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
    if constexpr (CPT) {
      int lo = 0, hi = nrung - 1;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (tpre[mid] > tile) hi = mid; else lo = mid + 1; }
      const int r = __builtin_amdgcn_readfirstlane(lo);
      const int kb = tile - (r ? tpre[r - 1] : 0);
      rl = rung0 + r;
      const int start = (kb * 4 + wave) * 64, cnt = p.ccnt[rl];
      if (start >= cnt) continue;                        // (only wave-level barriers below)
      nact = cnt - start < 64 ? cnt - start : 64;
      lbase = rl * p.W + start;
      c0 = rl * p.W; w0 = 0;
    } else {
      c0 = p.c_begin + (tile * 4 + wave) * 64;
      if (c0 >= p.c_end) continue;
      rl = __builtin_amdgcn_readfirstlane(c0 / p.W);
      w0 = c0 - rl * p.W;
    }
    const int rg = p.r0 + rl;
    auto walker_of = [&](int i) -> int {
      if constexpr (CPT) return p.cidx[lbase + (i < nact ? i : nact - 1)];
      else return w0 + i;
    };
    const int wl = walker_of(l);
    const bool dead = CPT && l >= nact;
    const int c = CPT ? rl * p.W + wl : c0 + l;
    int wq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) wq[g] = walker_of(16 * g + j);
    const double* timg = p.prop_tiles + (size_t)rl * (16 * 64) + l;   // tile t = (half*4 + slot)*2 + row tile

    mf_d2 rowv[4];
    mf_d2* rowp;
    auto ask_rows = [&](int g, mf_d2 (&rv)[4], mf_d2*& rp) {
      rp = reinterpret_cast<mf_d2*>(p.x + ((size_t)rl * p.W + wq[g]) * DP) + q;   // piece t at [4t]
#pragma unroll
      for (int t = 0; t < 4; ++t) rv[t] = rp[4 * t];
    };
    ask_rows(0, rowv, rowp);
    double ta[4][2];   // the first half's factor tiles of the NEXT segment A (L2-resident; asked for one block ahead)
    auto ask_ta = [&]() {
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) {
        ta[sl][0] = timg[((0 * 4 + sl) * 2 + 0) * 64];
        ta[sl][1] = timg[((0 * 4 + sl) * 2 + 1) * 64];
      }
    };
    ask_ta();
    const int tc = CPT ? 0 : p.touch[c];
    const double ll = p.ll[c], lp = p.lp[c];
    const int ntries0 = p.ntries[c], naccept0 = p.naccept[c];
    const unsigned int nhist0 = CPT ? 0u : p.nhist[c];
    const double beta = as_c(p.beta)[rg];
    const u32x4 o0 = draw_block(p.seed, TAG_MH, (uint32_t)(wl + p.w_off) * (uint32_t)p.Nt + (uint32_t)rg, p.step, 0);
    const double log_u = dlog_u01(o0.v0);

    // the normals of chain (g, j), dimensions 16 hb + 4 q .. + 3: one Philox block, two Box-Muller pairs -- as NS slices
    // (NS = 7: rounds 3 3 3 1+radius | angle | radius | angle;  NS = 9: rounds 2 2 2 2 2 | radius | angle | radius | angle)
    const double* tab = (const double*)lds_all;
    auto draw_step = [&](auto nsc, auto ic, lean_draw& d, int g, int hb, double (&zz)[4]) {
      constexpr int NS = decltype(nsc)::value, I = decltype(ic)::value;
      constexpr int NR = NS == 7 ? (I < 3 ? 3 : (I == 3 ? 1 : 0)) : (I < 5 ? 2 : 0);
      if constexpr (I == 0) {
        const int qd = opaque_copy(q);
        d.ps = draw_block_begin(p.seed, TAG_MH, (uint32_t)(wq[g] + p.w_off) * (uint32_t)p.Nt + (uint32_t)rg, p.step, (uint32_t)(1 + 4 * hb + qd));
      }
#pragma unroll
      for (int r = 0; r < NR; ++r) philox_round(d.ps);
      if constexpr (NR > 0) { lean_pin(d.ps.c0); lean_pin(d.ps.c1); lean_pin(d.ps.c2); lean_pin(d.ps.c3); }
      if constexpr (I == NS - 4) { d.r = bm_sqrt(bm_neg2log(d.ps.c0, tab)); lean_pin(d.r); }
      if constexpr (I == NS - 3) { boxmuller_finish(d.r, d.ps.c1, tab, zz[0], zz[1]); lean_pin(zz[0]); lean_pin(zz[1]); }
      if constexpr (I == NS - 2) { d.r = bm_sqrt(bm_neg2log(d.ps.c2, tab)); lean_pin(d.r); }
      if constexpr (I == NS - 1) { boxmuller_finish(d.r, d.ps.c3, tab, zz[2], zz[3]); lean_pin(zz[2]); lean_pin(zz[3]); }
    };
    double z0[4];
    {
      lean_draw d;
      lean_for<7>([&](auto ic) { draw_step(std::integral_constant<int, 7>{}, ic, d, 0, 0, z0); });
    }
    PTM_STAGE();

    auto pass = [&](auto gpc) {
      constexpr int gp = decltype(gpc)::value;
      // ---- A: T x Z, first half of the columns  ||  the second half's normals
      double tb[4][2];
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) {
        tb[sl][0] = LOW ? 0.0 : timg[((1 * 4 + sl) * 2 + 0) * 64];
        tb[sl][1] = timg[((1 * 4 + sl) * 2 + 1) * 64];
      }
      mf_d4 acc[2] = {mf_d4{0.0, 0.0, 0.0, 0.0}, mf_d4{0.0, 0.0, 0.0, 0.0}};
      double z1[4];
      {
        lean_draw d;
        lean_for<8>([&](auto ic) {
          constexpr int i = decltype(ic)::value, sl = i >> 1, rt = i & 1;
          acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[sl][rt], z0[sl], acc[rt], 0, 0, 0);
          PTM_STAGE();
          if constexpr (i < 7) {
            draw_step(std::integral_constant<int, 7>{}, ic, d, gp, 1, z1);
            PTM_STAGE();
          }
        });
      }
      // ---- B: T x Z, second half
#pragma unroll
      for (int sl = 0; sl < 4; ++sl)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          if (LOW && rt == 0) continue;   // columns >= 16 never reach rows < 16
          acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(tb[sl][rt], z1[sl], acc[rt], 0, 0, 0);
        }
      PTM_STAGE();
      // ---- C: x' = x + offset (state::add, states.cc:205-214); the box of the all-uniform prior
      double xp[8];
      bool ok = true;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const mf_d2 v = rowv[t], lo = box[4 * t], hi = box[16 + 4 * t];
        const int m = 2 * t;   // registers m, m+1 <-> dimensions q + 4m, q + 4m + 4
        xp[m] = v.x + acc[m >> 2][m & 3];
        xp[m + 1] = v.y + acc[(m + 1) >> 2][(m + 1) & 3];
        ok = ok & !(xp[m] < lo.x) & !(xp[m] > hi.x) & !(xp[m + 1] < lo.y) & !(xp[m + 1] > hi.y);
      }
      uint64_t inbox = __builtin_amdgcn_ballot_w64(ok);
      inbox &= inbox >> 32;
      inbox &= inbox >> 16;                              // bit jj: all four lanes (q, jj) of chain (gp, jj) are inside
      PTM_STAGE();
      // ---- D: S = P2 x X' and the chain's four partial dot products  ||  the next pass's first normals
      mf_d2 rown[4];
      mf_d2* rowpn = rowp;
      double z0n[4] = {0.0, 0.0, 0.0, 0.0};
      if (gp < 3) {
        ask_rows(gp + 1, rown, rowpn);
        ask_ta();
      }
      PTM_STAGE();
      mf_d4 sacc[2] = {mf_d4{0.0, 0.0, 0.0, 0.0}, mf_d4{0.0, 0.0, 0.0, 0.0}};
      {
        lean_draw d;
        lean_for<12>([&](auto ic) {
          // instruction i: column block m of both row tiles while m < 4, then of the lower row tile alone
          constexpr int i = decltype(ic)::value, m = i < 8 ? i >> 1 : i - 4, rt = i < 8 ? (i & 1) : 1;
          sacc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(pimg[(rt ? m : 8 + m) * 64], xp[m], sacc[rt], 0, 0, 0);
          PTM_STAGE();
          if constexpr (gp < 3 && i < 9) {
            draw_step(std::integral_constant<int, 9>{}, ic, d, gp + 1, 0, z0n);
            PTM_STAGE();
          }
        });
      }
      double pq = 0.0;
#pragma unroll
      for (int m = 0; m < 8; ++m) pq = __builtin_fma(xp[m], sacc[m >> 2][m & 3], pq);
      red[q * 16 + j] = pq;   // chain (gp, j)'s four partial sums sit on lanes (0..3, j): hand them to lane 16 gp + j
      __builtin_amdgcn_wave_barrier();
      // ---- E, lanes 16 gp .. 16 gp + 15 (chain = lane): Metropolis test and add_state counters (chain.cc:973-1019, 916-949)
      bool accept = false;
      if (q == gp && !dead) {
        const double* mine = red + j;
        const double quad = ((mine[0] + mine[16]) + mine[32]) + mine[48];
        if (tc) {
          p.nhist[c] = nhist0 + (unsigned int)tc;
          p.touch[c] = 0;
        } else {
          const double bl = beta * ll;
          const double cur_lpost = lp + bl;
          const double oldlprior = cur_lpost - bl;  // chain.cc:973
          const bool in = ((inbox >> j) & 1ull) != 0;
          const double newlprior = in ? p.lprior_const : -__builtin_inf();
          const bool want_like = newlprior > -1e200 || newlprior - oldlprior > p.min_prior;  // chain.cc:980 (Q1)
          double newlike = p.like0 - 0.5 * quad;
          double newlpost = newlike * beta + newlprior;
          if (!want_like) newlike = newlpost = -__builtin_inf();
          const double logH = newlpost - cur_lpost;
          accept = true;
          if (logH < 0) accept = log_u < logH;  // chain.cc:998-1001 (NaN stays accepted)
          p.ntries[c] = ntries0 + 1;
          if (!CPT) p.nhist[c] = nhist0 + 1u;   // (compacted: the engine counts the step for everybody, ptm_aux_kernels.hpp)
          if (accept) {
            p.naccept[c] = naccept0 + 1;
            p.last_type[c] = 0;
            p.ll[c] = newlike;
            p.lp[c] = newlprior;
          }
        }
      }
      // accepted proposals replace their rows; each of a chain's four lanes writes its 64 bytes
      const uint64_t acc_bits = __builtin_amdgcn_ballot_w64(accept) >> (16 * gp);
      if ((acc_bits >> j) & 1ull) {
#pragma unroll
        for (int t = 0; t < 4; ++t) rowp[4 * t] = mf_d2{xp[2 * t], xp[2 * t + 1]};
      }
      __builtin_amdgcn_wave_barrier();   // the next pass reuses the LDS slots
      if (gp < 3) {
        rowp = rowpn;
#pragma unroll
        for (int t = 0; t < 4; ++t) rowv[t] = rown[t];
#pragma unroll
        for (int t = 0; t < 4; ++t) z0[t] = z0n[t];
      }
      PTM_STAGE();
    };
    pass(std::integral_constant<int, 0>{});
    pass(std::integral_constant<int, 1>{});
    pass(std::integral_constant<int, 2>{});
    pass(std::integral_constant<int, 3>{});
  }   // tiles
}
#undef PTM_STAGE

}  // namespace ptm
