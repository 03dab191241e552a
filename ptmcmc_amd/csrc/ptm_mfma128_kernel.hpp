// ptm_mfma128_kernel.hpp -- the fused MH_chain::step (chain.cc:966-1022) for Gaussian targets of 65..128 dimensions on the f64
// matrix cores: the construction of ptm_mfma_kernel.hpp (32) and ptm_mfma64_kernel.hpp (64) with 8 x 8 tiles of 16.
//
// gaussian_prop::draw is a dense D x D transform per draw at any dimension (proposal_distribution.hh:207-215); at 128 both
// products of a step -- offset = T z (128 x 128) and s = P2 x' (the lower triangle of the precision) -- are 16384 + 8256
// multiply-adds per chain and step, which the lanes kernel (two dimensions per lane, ptm_lanes_kernel.hpp) walks as 128-long
// dependent chains with every operand read from memory per chain.  Here one wave = 64 chains of ONE rung (W % 64 == 0), worked
// as four groups of 16 chains, one group per pass:
//
//   offset (128 x 16 chains) = T (128 x 128)  x  Z (128 x 16 normals)   32 k-steps x 8 row tiles of v_mfma_f64_16x16x4_f64
//   S      (128 x 16 chains) = P2 (lower)     x  X'                     144 tiles on and under the diagonal
//   x'.S per chain                                                       32 fma per lane, 4 lanes per chain through LDS
//
// Lane l = 16 q + j works for chain 16 g + j of its wave on the thirty-two dimensions d = q + 4 m, m = 0..31: its normals are
// Philox blocks 1 + 4 hb + q (hb = 0..7: the halves of 16 columns) of that chain's stream, the accumulator registers of T z ARE
// the B operands of P2 x' (register m = dimension 4 m + k on lane group k), and stored rows keep dimension q + 4 m at 16-byte
// slot 4 (m / 2) + q (row_pos<128>), so that a chain's four lanes read 64 contiguous bytes per load.  The rung's factor -- 256
// operand tiles, 128 KB -- is read per group from the L2 through a ring of four k-step slots; the precision's 144 tiles (72 KB)
// are staged in LDS once per block (persistent grid, one block per CU).
// Sums accumulate in the order every other path and the CPU checker share (ptmo_column_order: halves of 16 columns, inside a
// half s + 4 k with s outer, k inner = the MFMA's own k order; the precision rows j ascending, ptmo_llike), structural zeros of
// a Cholesky factor / above the precision's diagonal are skipped or stored as zeros (fma(0, z, acc) == acc): chains are
// bit-identical to the lanes kernel's and the checker's.
// Builds: uniform box prior, zero mean, no one-dimensional moves, no mixture, no history; open bounds or (BND) open / `limit`
// bounds; a fixed ladder or (EV) per-chain temperatures of evolving ladders
// (everything else at 65..128 dimensions keeps the lanes kernel).
#pragma once
#include <type_traits>

#include "ptm_kernels.hpp"

namespace ptm {

typedef double m128_d4 __attribute__((ext_vector_type(4)));
typedef double m128_d2 __attribute__((ext_vector_type(2)));

constexpr int M128_NT = 8;           // row tiles of 16
constexpr int M128_KS = 32;          // k-steps of 4 columns
constexpr int M128_P2_TILES = 144;   // (row tile rt, step m) with m <= 4 rt + 3: at m128_p2_base(rt) + m
__host__ __device__ constexpr int m128_p2_base(int rt) { return 2 * rt * (rt + 1); }
// LDS in doubles: Box-Muller tables | precision tiles | prior box lo | hi (row layout) | 64 reduction slots per wave
constexpr int M128_THREADS = 512;     // eight waves share the block's tables: two per SIMD at 256 registers each
constexpr int m128_lds_doubles() { return BM_TABLE_DOUBLES + M128_P2_TILES * 64 + 256 + (M128_THREADS / 64) * 64 + 256; }   // (... | boundary box lo | hi)

#define PTM_M128_STAGE() __builtin_amdgcn_sched_barrier(0)

template <int KIND, bool BND = false, bool EV = false>   // BND: open / `limit` boundaries on top of the prior's box; EV: per-chain temperatures (evolving ladders)
__global__ __launch_bounds__(M128_THREADS, 1) void sweep_mfma128_kernel(const Dev p) {
  constexpr int DP = 128;
  constexpr int NT = M128_NT;
  constexpr bool LOW = KIND == KIND_LOWER;
  constexpr int RING = 3;   // k-step slots of factor tiles in flight
  extern __shared__ __attribute__((aligned(16))) double lds_all[];
  double* ptile = lds_all + BM_TABLE_DOUBLES;
  double* lbox = ptile + M128_P2_TILES * 64;
  const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
  double* red = lbox + 256 + wave * 64;
  double* lebox = lbox + 256 + (M128_THREADS / 64) * 64;   // [2][128] boundary::enforce for open / limit sides (states.cc:53-55) as a box, row layout
  const int q = l >> 4, j = l & 15;
  const double* pimg = ptile + l;
  const m128_d2* box = reinterpret_cast<const m128_d2*>(lbox) + q;   // lo piece t at [4t], hi piece t at [64 + 4t]
  const m128_d2* ebx = reinterpret_cast<const m128_d2*>(lebox) + q;

  // the block's tables, once: the grid is persistent (a block per CU walks the launch's 512-chain tiles)
  for (int i = threadIdx.x; i < BM_TABLE_DOUBLES / 2; i += M128_THREADS) reinterpret_cast<bm_d2*>(lds_all)[i] = reinterpret_cast<const bm_d2*>(BM_TABLE)[i];
  for (int i = threadIdx.x; i < M128_P2_TILES * 64; i += M128_THREADS) ptile[i] = p.P2_tiles[i];
  if (threadIdx.x < 256) lbox[threadIdx.x] = p.box_row[threadIdx.x];
  if (BND && threadIdx.x < 128) {
    const int d = threadIdx.x, pos = row_pos<128>(d);
    lebox[pos] = p.blo[d] == B_LIMIT ? p.bmin[d] : -__builtin_inf();
    lebox[128 + pos] = p.bhi[d] == B_LIMIT ? p.bmax[d] : __builtin_inf();
  }
  __syncthreads();

  const int ntiles = (p.c_end - p.c_begin + M128_THREADS - 1) / M128_THREADS;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int c0 = p.c_begin + (tile * (M128_THREADS / 64) + wave) * 64;   // first chain of the wave (ranges are multiples of 64)
    if (c0 >= p.c_end) continue;                          // (only wave-level barriers below)
    const int rl = __builtin_amdgcn_readfirstlane(c0 / p.W);
    const int w0 = c0 - rl * p.W;
    const int rg = p.r0 + rl;
    const int c = c0 + l;                                 // "my" chain for the per-chain work
    // tile (hb * 4 + sl) * NT + rt of the rung's factor: a wave-uniform base (scalar registers) + the lane's 8 bytes, so that a
    // tile load is `global_load v, v_lane, s[base] offset:..` with the 128 KB of tile offsets folded into scalar adds -- as a
    // per-lane 64-bit pointer every 4 KB window of them cost two vector registers, hoisted to the top of the pass
    const char* const tbase = reinterpret_cast<const char*>(p.prop_tiles + (size_t)rl * (NT * M128_KS * 64));
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
      const unsigned int lane8 = (unsigned int)opaque_copy((int)lb8);   // (a value of this pass: the passes must not share 144 tile addresses)
      const uint32_t stream = (uint32_t)(w0 + 16 * g + j + p.w_off) * (uint32_t)p.Nt + (uint32_t)rg;
      m128_d2* const rowp = reinterpret_cast<m128_d2*>(p.x + (size_t)(c0 + 16 * g + j) * DP) + q;   // the group's rows: piece t at [4t]
      m128_d2 rowv[16];
      // ---- T x Z, one 16-column half at a time.  The operand tiles of a k-step (up to eight, one per row tile) travel RING
      //      k-steps ahead of their use: when k-step ks has issued, its slot asks for k-step ks + RING.
      m128_d4 acc[NT];
#pragma unroll
      for (int rt = 0; rt < NT; ++rt) acc[rt] = m128_d4{0.0, 0.0, 0.0, 0.0};
      double ta[RING][NT];
      auto ask_tiles = [&](int ks) {
        const int hb = ks >> 2;
#pragma unroll
        for (int rt = 0; rt < NT; ++rt) ta[ks % RING][rt] = (LOW && hb > rt) ? 0.0 : *reinterpret_cast<const double*>(tbase + (ks * NT + rt) * 512 + lane8);
      };
#pragma unroll
      for (int ks = 0; ks < RING; ++ks) ask_tiles(ks);
      PTM_M128_STAGE();
#pragma unroll
      for (int hb = 0; hb < 8; ++hb) {
        double z[4];
        {
          const u32x4 o = draw_block(p.seed, TAG_MH, stream, p.step, (uint32_t)(1 + 4 * hb + qd));
          boxmuller(o.v0, o.v1, (const double*)lds_all, z[0], z[1]);
          boxmuller(o.v2, o.v3, (const double*)lds_all, z[2], z[3]);
        }
        PTM_M128_STAGE();
        if (hb == 6) {   // the rows: needed after the last half (two halves of arithmetic away)
#pragma unroll
          for (int t = 0; t < 16; ++t) rowv[t] = rowp[4 * t];
        }
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
          const int ks = hb * 4 + sl;
#pragma unroll
          for (int rt = 0; rt < NT; ++rt) {
            if (LOW && hb > rt) continue;   // columns >= 16 hb never reach rows < 16 hb of a Cholesky factor
            acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[ks % RING][rt], z[sl], acc[rt], 0, 0, 0);
          }
          PTM_M128_STAGE();
          if (ks + RING < M128_KS) ask_tiles(ks + RING);
          PTM_M128_STAGE();
        }
      }
      // ---- x' = x + offset (state::add, states.cc:205-214) and the box of the uniform prior
      double xp[32];
      bool ok = true, vok = true;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const m128_d2 v = rowv[t], lo = box[4 * t], hi = box[64 + 4 * t];
        const int m = 2 * t;   // registers m, m + 1 <-> dimensions q + 4 m, q + 4 m + 4
        xp[m] = v.x + acc[m >> 2][m & 3];
        xp[m + 1] = v.y + acc[(m + 1) >> 2][(m + 1) & 3];
        ok = ok & !(xp[m] < lo.x) & !(xp[m] > hi.x) & !(xp[m + 1] < lo.y) & !(xp[m + 1] > hi.y);
        if (BND) {
          const m128_d2 el = ebx[4 * t], eh = ebx[64 + 4 * t];
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
      PTM_M128_STAGE();
      // ---- S = P2 x X' (the accumulator layout of x' is the B operand's) and the lane's part of x'.S
      m128_d4 sacc[NT];
#pragma unroll
      for (int rt = 0; rt < NT; ++rt) sacc[rt] = m128_d4{0.0, 0.0, 0.0, 0.0};
      // (operand tiles from LDS, read two k-steps ahead of their use; the fences keep the scheduler from reading all at once)
      double pa[3][NT];
      auto read_p2 = [&](int m) {
#pragma unroll
        for (int rt = 0; rt < NT; ++rt)
          if (m <= 4 * rt + 3) pa[m % 3][rt] = pimg[(m128_p2_base(rt) + m) * 64];
      };
      read_p2(0);
      read_p2(1);
#pragma unroll
      for (int m = 0; m < 32; ++m) {
        if (m + 2 < 32) read_p2(m + 2);
        PTM_M128_STAGE();
#pragma unroll
        for (int rt = 0; rt < NT; ++rt) {
          if (m > 4 * rt + 3) continue;                  // above the diagonal
          sacc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[m % 3][rt], xp[m], sacc[rt], 0, 0, 0);
        }
        PTM_M128_STAGE();
      }
      double pq = 0.0;
#pragma unroll
      for (int m = 0; m < 32; ++m) pq = __builtin_fma(xp[m], sacc[m >> 2][m & 3], pq);
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
      // ---- accepted proposals replace their rows; each of a chain's four lanes writes its 256 bytes
      const uint64_t acc_bits = __builtin_amdgcn_ballot_w64(accept) >> (16 * g);
      if ((acc_bits >> j) & 1ull) {
#pragma unroll
        for (int t = 0; t < 16; ++t) rowp[4 * t] = m128_d2{xp[2 * t], xp[2 * t + 1]};
      }
      __builtin_amdgcn_wave_barrier();   // the next pass reuses the LDS slots
    };
    pass(std::integral_constant<int, 0>{});
    pass(std::integral_constant<int, 1>{});
    pass(std::integral_constant<int, 2>{});
    pass(std::integral_constant<int, 3>{});
  }
}
#undef PTM_M128_STAGE

}  // namespace ptm
