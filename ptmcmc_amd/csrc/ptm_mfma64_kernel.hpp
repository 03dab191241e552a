// ptm_mfma64_kernel.hpp -- the fused MH_chain::step (chain.cc:966-1022) for Gaussian targets of 33..64 dimensions on the f64
// matrix cores: the 32-dimensional kernel's construction (ptm_mfma_kernel.hpp) with 4 x 4 tiles of 16.
//
// The reference's gaussian_prop has no dimension limit (proposal_distribution.hh:207-215: a dense D x D transform per draw), and at
// 64 dimensions both products of a step -- offset = T z (64 x 64) and s = P2 x' (the lower triangle of the precision) -- are
// the genuine dense contractions north_star reserves the matrix cores for: 8192 + 4160 multiply-adds per chain and step, which
// the lanes kernel (a lane per dimension, ptm_lanes_kernel.hpp) walks as 64-long dependent chains with every operand read from
// memory per chain.  Here, as at 32 dimensions, one wave = 64 chains of ONE rung (W % 64 == 0):
//
//   offset (64 x 64 chains) = T (64 x 64)  x  Z (64 x 64 normals)     16 k-steps x 4 row tiles of v_mfma_f64_16x16x4_f64 per group
//   S      (64 x 64 chains) = P2 (lower)   x  X'                      40 tiles on and under the diagonal
//   x'.S per chain                                                     16 fma per lane, 4 lanes per chain through LDS
//
// Lane l = 16 q + j works, for each of the wave's four 16-chain groups g, on chain 16 g + j and on the sixteen dimensions
// d = q + 4 m, m = 0..15: its normals are Philox blocks 1 + 4 hb + q (hb = 0..3) of that chain's stream, the accumulator
// registers of T z ARE the B operands of P2 x' (register m = dimension 4 m + k on lane group k), and stored rows keep
// dimension q + 4 m at 16-byte slot 4 (m / 2) + q (row_pos<64>), so that a chain's four lanes read 64 contiguous bytes per
// load.  The rung's factor -- 64 operand tiles, 32 KB -- is read per group from the L2, one 16-column half ahead of its use.
// Sums accumulate in the order every other path and the CPU checker share (ptmo_column_order: halves of 16 columns, inside a
// half s + 4 k with s outer, k inner = the MFMA's own k order; the precision rows j ascending), structural zeros of a Cholesky
// factor / above the precision's diagonal are skipped or stored as zeros (fma(0, z, acc) == acc): chains are bit-identical to the
// lanes kernel's and the checker's.
// Builds: uniform box prior, zero mean, no one-dimensional moves, no mixture, no history; open bounds or (BND) open / `limit`
// bounds; a fixed ladder or (EV) per-chain temperatures of evolving ladders
// (everything else at 33..64 dimensions keeps the lanes kernel).
#pragma once
#include <type_traits>

#include "ptm_kernels.hpp"

namespace ptm {

typedef double m64_d4 __attribute__((ext_vector_type(4)));
typedef double m64_d2 __attribute__((ext_vector_type(2)));

constexpr int M64_P2_TILES = 40;   // (row tile rt, step m) with m <= 4 rt + 3: at m64_p2_base(rt) + m
__host__ __device__ constexpr int m64_p2_base(int rt) { return rt == 0 ? 0 : (rt == 1 ? 4 : (rt == 2 ? 12 : 24)); }
// LDS in doubles: Box-Muller tables | precision tiles | prior box lo | hi (row layout) | 64 reduction slots per wave
constexpr int m64_lds_doubles() { return BM_TABLE_DOUBLES + M64_P2_TILES * 64 + 128 + 4 * 64 + 128; }   // (... | boundary box lo | hi)

#define PTM_M64_STAGE() __builtin_amdgcn_sched_barrier(0)

template <int KIND, bool BND = false, bool EV = false>   // BND: open / `limit` boundaries on top of the prior's box; EV: per-chain temperatures (evolving ladders)
__global__ __launch_bounds__(256, 2) void sweep_mfma64_kernel(const Dev p) {
  constexpr int DP = 64;
  constexpr bool LOW = KIND == KIND_LOWER;
  extern __shared__ __attribute__((aligned(16))) double lds_all[];
  double* ptile = lds_all + BM_TABLE_DOUBLES;
  double* lbox = ptile + M64_P2_TILES * 64;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  double* red = lbox + 128 + wave * 64;
  double* lebox = lbox + 128 + 4 * 64;   // [2][64] boundary::enforce for open / limit sides (states.cc:53-55) as a box, row layout
  const int q = l >> 4, j = l & 15;
  const double* pimg = ptile + l;
  const m64_d2* box = reinterpret_cast<const m64_d2*>(lbox) + q;   // lo piece t at [4t], hi piece t at [32 + 4t]
  const m64_d2* ebx = reinterpret_cast<const m64_d2*>(lebox) + q;

  // the block's tables, once: the grid is persistent (a block per resident slot walks the launch's 256-chain tiles)
  for (int i = threadIdx.x; i < BM_TABLE_DOUBLES / 2; i += 256) reinterpret_cast<bm_d2*>(lds_all)[i] = reinterpret_cast<const bm_d2*>(BM_TABLE)[i];
  for (int i = threadIdx.x; i < M64_P2_TILES * 64; i += 256) ptile[i] = p.P2_tiles[i];
  if (threadIdx.x < 128) lbox[threadIdx.x] = p.box_row[threadIdx.x];
  if (BND && threadIdx.x < 64) {
    const int d = threadIdx.x, pos = row_pos<64>(d);
    lebox[pos] = p.blo[d] == B_LIMIT ? p.bmin[d] : -__builtin_inf();
    lebox[64 + pos] = p.bhi[d] == B_LIMIT ? p.bmax[d] : __builtin_inf();
  }
  __syncthreads();

  const int ntiles = (p.c_end - p.c_begin + 255) >> 8;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int c0 = p.c_begin + (tile * 4 + wave) * 64;   // first chain of the wave (ranges are multiples of 64)
    if (c0 >= p.c_end) continue;                          // (only wave-level barriers below)
    const int rl = __builtin_amdgcn_readfirstlane(c0 / p.W);
    const int w0 = c0 - rl * p.W;
    const int rg = p.r0 + rl;
    const int c = c0 + l;                                 // "my" chain for the per-chain work
    // tile (hb * 4 + sl) * 4 + rt of the rung's factor: a wave-uniform base (scalar registers) + the lane's 8 bytes -- a tile load is
    // `global_load v, v_lane, s[base] offset:..`, the 32 KB of tile offsets fold into scalar adds instead of per-lane 64-bit pointers
    const char* const tbase = reinterpret_cast<const char*>(p.prop_tiles + (size_t)rl * (64 * 64));
    const unsigned int lb8 = 8u * (unsigned int)l;

    // per-chain scalars: used at the very end
    const int tc = p.touch[c];  // > 0: the rung took part in that many exchange attempts => no MH move this step
    const double ll = p.ll[c], lp = p.lp[c];
    const int ntries0 = p.ntries[c], naccept0 = p.naccept[c];
    const unsigned int nhist0 = p.nhist[c];
    const double beta = EV ? p.betaC[c] : as_c(p.beta)[rg];
    const u32x4 o0 = draw_block(p.seed, TAG_MH, (uint32_t)(w0 + l + p.w_off) * (uint32_t)p.Nt + (uint32_t)rg, p.step, 0);
    const double log_u = dlog_u01(o0.v0);

    auto pass = [&](auto gc) {
      constexpr int g = decltype(gc)::value;
      const int qd = opaque_copy(q);
      const unsigned int lane8 = (unsigned int)opaque_copy((int)lb8);   // (a value of this pass: the passes must not share tile addresses)
      const uint32_t stream = (uint32_t)(w0 + 16 * g + j + p.w_off) * (uint32_t)p.Nt + (uint32_t)rg;
      m64_d2* const rowp = reinterpret_cast<m64_d2*>(p.x + (size_t)(c0 + 16 * g + j) * DP) + q;   // the group's rows: piece t at [4t]
      m64_d2 rowv[8];
      // ---- T x Z, one 16-column half at a time.  The operand tiles travel one half ahead in a ring of four k-step slots: when
      //      k-step (hb, sl) has issued, its slot asks for (hb + 1, sl) -- in flight over the rest of the half and the next draw.
      m64_d4 acc[4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc[rt] = m64_d4{0.0, 0.0, 0.0, 0.0};
      double ta[4][4];   // [sl][rt]
      auto ask_tiles = [&](int hb, int sl) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) ta[sl][rt] = (LOW && hb > rt) ? 0.0 : *reinterpret_cast<const double*>(tbase + ((hb * 4 + sl) * 4 + rt) * 512 + lane8);
      };
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) ask_tiles(0, sl);
      PTM_M64_STAGE();
#pragma unroll
      for (int hb = 0; hb < 4; ++hb) {
        double z[4];
        {
          const u32x4 o = draw_block(p.seed, TAG_MH, stream, p.step, (uint32_t)(1 + 4 * hb + qd));
          boxmuller(o.v0, o.v1, (const double*)lds_all, z[0], z[1]);
          boxmuller(o.v2, o.v3, (const double*)lds_all, z[2], z[3]);
        }
        PTM_M64_STAGE();
        if (hb == 2) {   // the rows: needed after the last half (two halves of arithmetic away)
#pragma unroll
          for (int t = 0; t < 8; ++t) rowv[t] = rowp[4 * t];
        }
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) {
            if (LOW && hb > rt) continue;   // columns >= 16 hb never reach rows < 16 hb of a Cholesky factor
            acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[sl][rt], z[sl], acc[rt], 0, 0, 0);
          }
          PTM_M64_STAGE();
          if (hb + 1 < 4) ask_tiles(hb + 1, sl);
          PTM_M64_STAGE();
        }
      }
      // ---- x' = x + offset (state::add, states.cc:205-214) and the box of the uniform prior
      double xp[16];
      bool ok = true, vok = true;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const m64_d2 v = rowv[t], lo = box[4 * t], hi = box[32 + 4 * t];
        const int m = 2 * t;   // registers m, m + 1 <-> dimensions q + 4 m, q + 4 m + 4
        xp[m] = v.x + acc[m >> 2][m & 3];
        xp[m + 1] = v.y + acc[(m + 1) >> 2][(m + 1) & 3];
        ok = ok & !(xp[m] < lo.x) & !(xp[m] > hi.x) & !(xp[m + 1] < lo.y) & !(xp[m + 1] > hi.y);
        if (BND) {
          const m64_d2 el = ebx[4 * t], eh = ebx[32 + 4 * t];
          vok = vok & !(xp[m] < el.x) & !(xp[m] > eh.x) & !(xp[m + 1] < el.y) & !(xp[m + 1] > eh.y);
        }
      }
      uint64_t inb = __builtin_amdgcn_ballot_w64(ok);
      inb &= inb >> 32;
      inb &= inb >> 16;                                  // bit jj: all four lanes (q, jj) of chain (g, jj) are inside
      uint64_t vb = ~0ull;                                // BND: bit jj: the chain's state is valid (stateSpace::enforce, states.cc:86-102)
      if (BND) {
        vb = __builtin_amdgcn_ballot_w64(vok);
        vb &= vb >> 32;
        vb &= vb >> 16;
      }
      PTM_M64_STAGE();
      // ---- S = P2 x X' (the accumulator layout of x' is the B operand's) and the lane's part of x'.S
      m64_d4 sacc[4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) sacc[rt] = m64_d4{0.0, 0.0, 0.0, 0.0};
      // (operand tiles from LDS, read two k-steps ahead of their use; the fences keep the scheduler from reading all forty at once)
      double pa[3][4];
      auto read_p2 = [&](int m) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
          if (m <= 4 * rt + 3) pa[m % 3][rt] = pimg[(m64_p2_base(rt) + m) * 64];
      };
      read_p2(0);
      read_p2(1);
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        if (m + 2 < 16) read_p2(m + 2);
        PTM_M64_STAGE();
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          if (m > 4 * rt + 3) continue;                  // above the diagonal
          sacc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[m % 3][rt], xp[m], sacc[rt], 0, 0, 0);
        }
        PTM_M64_STAGE();
      }
      double pq = 0.0;
#pragma unroll
      for (int m = 0; m < 16; ++m) pq = __builtin_fma(xp[m], sacc[m >> 2][m & 3], pq);
      red[q * 16 + j] = pq;   // chain (g, j)'s four partial sums sit on lanes (0..3, j): to lane 16 g + j through this wave's LDS
      __builtin_amdgcn_wave_barrier();
      // ---- lanes 16 g .. 16 g + 15 (chain = lane): Metropolis test and add_state counters (chain.cc:973-1019, 916-949)
      bool accept = false;
      if ((l >> 4) == g) {
        const double* mine = red + j;
        const double quad = ((mine[0] + mine[16]) + mine[32]) + mine[48];
        if (tc) {
          p.nhist[c] = nhist0 + (unsigned int)tc;
          p.touch[c] = 0;
        } else {
          const double bl = beta * ll;
          const double cur_lpost = lp + bl;
          const double oldlprior = cur_lpost - bl;  // chain.cc:973
          const bool in = ((inb >> j) & 1ull) != 0;
          // Q9: state::add builds on an enforced zero state -- an origin outside a `limit` bound invalidates every proposal
          const bool valid = !BND || (p.origin_valid != 0 && ((vb >> j) & 1ull) != 0);
          double newlprior = in ? p.lprior_const : -__builtin_inf();
          if (!valid) newlprior = -__builtin_inf();
          const bool want_like = valid && (newlprior > -1e200 || newlprior - oldlprior > p.min_prior);  // chain.cc:980 (Q1)
          double newlike = p.like0 - 0.5 * quad;
          double newlpost = newlike * beta + newlprior;
          if (!want_like) newlike = newlpost = -__builtin_inf();
          const double logH = newlpost - cur_lpost;
          accept = valid;
          if (accept && logH < 0) accept = log_u < logH;  // chain.cc:998-1001 (NaN stays accepted)
          p.ntries[c] = ntries0 + 1;
          p.nhist[c] = nhist0 + 1u;
          if (accept) {
            p.naccept[c] = naccept0 + 1;
            p.last_type[c] = 0;
            p.ll[c] = newlike;
            p.lp[c] = newlprior;
          }
        }
      }
      // ---- accepted proposals replace their rows; each of a chain's four lanes writes its 128 bytes
      const uint64_t acc_bits = __builtin_amdgcn_ballot_w64(accept) >> (16 * g);
      if ((acc_bits >> j) & 1ull) {
#pragma unroll
        for (int t = 0; t < 8; ++t) rowp[4 * t] = m64_d2{xp[2 * t], xp[2 * t + 1]};
      }
      __builtin_amdgcn_wave_barrier();   // the next pass reuses the LDS slots
    };
    pass(std::integral_constant<int, 0>{});
    pass(std::integral_constant<int, 1>{});
    pass(std::integral_constant<int, 2>{});
    pass(std::integral_constant<int, 3>{});
  }
}
#undef PTM_M64_STAGE

}  // namespace ptm
