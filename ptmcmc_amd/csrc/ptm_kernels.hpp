// ptm_kernels.hpp -- gfx950 kernels of the parallel-tempering step engine: the data layout, the shared device
// functions (state space, prior, Gaussian target, proposal product) and the GENERAL fused sweep kernel (one lane = one
// chain).  The 32-dimensional workload has its own kernel on the f64 matrix cores (ptm_mfma_kernel.hpp); the exchange
// phase is in ptm_aux_kernels.hpp.
//
// Data layout in HBM (per engine = per GPU shard), Nc = rung_count * W chains, chain c = rung_local * W + walker:
//   x      [Nc][DP]  double   one contiguous ROW per chain (DP = dimension padded to 4/8/16/32; pad entries stay 0,
//                             pad factor rows / precision rows are 0), updated IN PLACE; position of dimension d
//                             inside the row: row_pos (identity except for DP = 32, 64 and 128)
//   llike, lprior [Nc] double (lpost is always fl(lprior + fl(beta*llike)), chain.cc:928)
//   ntries, naccept, last_type [Nc] int32; nhist [Nc] uint32          (MH_chain counters, chain.hh:150-170;
//                             Nsize is a function of Nhist: 1 + ceil(nhist / add_every_N), chain.cc:935-947)
//   touch [Nc] uint8          exchange phase -> sweep hand-off: add_state calls the rung received (=> no MH move);
//   arr_above / arr_below [W] slot where a row arriving from the adjacent shard lands.
// Why rows: a Metropolis step rejects ~3/4 of its proposals and the exchange phase moves ~1/5 of the rows.  With rows
// contiguous (256 B at D=32 = 4 full cache lines) a lane reads its row with 16-byte loads (lanes 256 B apart: measured
// 5.1 TB/s vs 5.8 TB/s for the plane-wise layout), writes it back ONLY when the move is accepted, and the exchange
// kernel swaps whole rows in place -- no copy-through of unchanged rows, no second buffer.  The plane-wise (SoA)
// ping-pong layout of the first version moved 3.7 GB per sweep against 2.3 GB algorithmic (profiles/r01_*).
// Walker is the fastest index, so the 64 lanes of a wave hold 64 walkers of ONE rung whenever W % 64 == 0: beta, the
// proposal factor and the precision matrix are then wave-uniform (LDS-staged / scalar-cache operands).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ptm_device_math.hpp"

// minimum waves per SIMD the general sweep kernel is compiled for (register budget 512 / waves)
#ifndef PTM_SWEEP_WAVES
#define PTM_SWEEP_WAVES 3
#endif

namespace ptm {

enum { KIND_DENSE = 0, KIND_DIAG = 1, KIND_LOWER = 2 };
enum { B_OPEN = 0, B_LIMIT = 1, B_REFLECT = 2, B_WRAP = 3 };
enum { P_FLAT = 0, P_UNIFORM = 1, P_GAUSSIAN = 2, P_POLAR = 3, P_COPOLAR = 4, P_LOG = 5 };

// read-only tables written by the host before any launch: reading them through the constant address space lets
// the backend use scalar loads whenever the address is wave-uniform, with no alias analysis in the way.
typedef const double __attribute__((address_space(4))) * cdp;
typedef const int __attribute__((address_space(4))) * cip;
__device__ __forceinline__ cdp as_c(const double* p) { return (cdp)(uintptr_t)p; }
__device__ __forceinline__ cip as_c(const int* p) { return (cip)(uintptr_t)p; }

// problem description + state pointers; passed BY VALUE to every kernel (kernarg segment => scalar loads)
// Optional history of the first `rungs` local rungs -- what MH_chain::add_state pushes every add_every_N-th call
// (chain.cc:935-946: state, llike, lpost, acceptance ratio, type).  A ring of `cap` rows per chain: saved row s of chain
// c sits in slot s % cap -- x at [slot][c][DP] (row layout), llike / lprior at [slot][c], meta = {Naccept, Ntries,
// last_type, s} at [slot][c].  Row 0 is the initial state; the add number a (a % add_every_N == 0) saves row 1 + a/N.
struct Hist {
  int rungs, cap, HC;   // rungs == 0: off; HC = rungs * W chains
  double *x, *ll, *lp;
  int4* meta;
  double* beta;         // [slot][c] the chain's inverse temperature when the row was saved (MH_chain::invtemps,
                        // chain.cc:943) -- kept only once the ladders evolve (else null: the ladder says it)
};
__device__ __forceinline__ size_t hist_slot(const Hist& h, long long row, int c) { return (size_t)(row % h.cap) * h.HC + c; }
__device__ __forceinline__ void hist_scalars(const Hist& h, size_t o, long long row, double ll, double lp, int nacc, int ntry, int type,
                                             double beta) {
  h.ll[o] = ll; h.lp[o] = lp;
  h.meta[o] = make_int4(nacc, ntry, type, (int)row);
  if (h.beta) h.beta[o] = beta;
}

// Optional MAP tracking of the first `rungs` local rungs: MH_chain::add_state keeps the state of the largest log-posterior
// it was ever called with (chain.cc:931-934; MAPlpost starts at -1e200, chain.hh:69; the ladder's MAP is the cold
// rung's, chain.cc:1570-1571).  lpost [MC], llike / lprior [MC], x [MC][DP] (row layout), MC = rungs * W.
struct MapT {
  int rungs, MC;
  double *lpost, *ll, *lp, *x;
};
// records the candidate's scalars if it beats the stored MAP; the caller then copies the row
__device__ __forceinline__ bool map_try(const MapT& m, int c, double lpost, double ll, double lp) {
  if (!(lpost > m.lpost[c])) return false;
  m.lpost[c] = lpost; m.ll[c] = ll; m.lp[c] = lp;
  return true;
}

struct Dev {
  int D, DP, Nt, r0, nloc, W, Nc;
  int w_off;            // global index of local walker 0 (ptm_config.walker_begin): random streams are keyed by the global walker
  int c_begin, c_end;   // chains [c_begin, c_end) are swept by this launch (whole rungs; the full range is [0, Nc))
  Hist hist;
  MapT map;
  uint64_t seed, step;
  int add_every_n;
  double min_prior;
  // state space (states.hh:29-48)
  int has_bounds, origin_valid;
  int bounds_box;   // every boundary side is open or `limit`: enforcing is a box test
  const int *blo, *bhi;
  const double *bmin, *bmax;
  // prior (probability_function.cc:219-262)
  int all_uniform;
  double lprior_const;
  const int* ptype;
  const double *plo, *phi, *pcoef;
  // Gaussian target: packed rows over DP dims, row i = {2P_i0 .. 2P_i,i-1, P_ii}
  const double* P2;
  const double* mean;
  int has_mean;
  double like0;
  // ladder + proposals
  const double* beta;      // [Nt] global
  const double* betaC;     // [Nc] per-chain inverse temperatures once the ladders evolve (evolve_temps), else null; the
                           // SIMPLE / GEN 0 / GEN 1 builds never see it
  const double* beta_w;    // [W][Nt] the same, ladder-major: what the exchange kernel keeps.  The lean MFMA build for evolving
                           // ladders reads it directly (strided: free in an issue-bound kernel) and spares the step the transposition
  const double* beta_add;  // [Nc] evolving ladders with history / MAP tracking: the inverse temperature a rung touched by the
                           // exchange phase had at its LAST add_state of that phase (between two pries of the step)
  const double* prop;      // [nloc][prop_stride]  factor, dense column-major [col][row] (DP*DP), or sigmas (DP)
  // operand images of the MFMA kernel (DP == 32 only, ptm_mfma_kernel.hpp): 64-lane A tiles, and the box in row layout
  const double* prop_tiles;  // [nloc][16][64]  tile (half*4 + slot)*2 + rowtile, lane 16k+i: T[16 rowtile + i][16 half + 4k + slot]
                             // (DP == 64 / 128: [nloc][NT*NT*4][64], NT = DP / 16, tile (half*4 + slot)*NT + rowtile, ptm_mfma64 / 128_kernel.hpp)
  const double* P2_tiles;    // [16][64]        tile step*2 + rowtile, lane 16k+i: P2[16 rowtile + i][4 step + k] (lower, doubled); then [36][16] 4x4 blocks (R,C), [k][i]
  const double* box_row;     // [2][32]         prior box lo | hi at row_pos  (DP == 64 / 128: [2][DP])
  const double* onedfrac;  // [nloc]
  // optional scale mixture (a proposal_distribution_set of Gaussian members that are scalar multiples of the rung's
  // factor, proposal_distribution.cc:99-129, the sampler's default Gaussian recipe ptmcmc.cc:117-139):
  // mix [nloc][mix_K][3] = {cumulative share, scale, oneDfrac}; mix_K == 0: none
  int mix_K;
  const double* mix;
  int prop_stride, any_oned;
  // differential evolution (ptm_set_proposal_de) as the member of NEGATIVE scale of the mixture: a move drawn from the chain's own
  // saved history (proposal_distribution.cc:476-592), rows 0 .. de_init_extra - 1 from de_init (what MH_chain::initialize(n) saved
  // in front of the start state), the others from the history ring
  int de_on, de_init_extra;
  const double* de_init;      // [de_init_extra][Nc][DP] rows (row layout)
  double de_snooker, de_gamma_one, de_gamma_std, de_gamma_div, de_ignore;   // gamma_std = 1.68 / sqrt(D) / reduce_gamma, made by the host
  double* de_hast;            // [Nc] host-callback likelihood: the propose pass hands its log-Hastings ratio and type to the accept pass
  int* de_type;               // [Nc]
  // state (in place)
  double* x;                              // [Nc][DP] rows
  double *ll, *lp;                        // [Nc]
  int *ntries, *naccept, *last_type;
  unsigned int* nhist;
  unsigned char* touch;
  int* err;
  // host-callback likelihood (bayes_likelihood::register_evaluate_log surface): the sweep is split around the host
  //   mode 0: fused (device target)   mode 1: propose only -> xprop/lprior_new/gate   mode 2: accept with llike_new
  int mode;
  double* xprop;            // [Nc][DP] proposed (enforced) states
  double* lprior_new;       // [Nc]
  unsigned char* gate;      // [Nc] bit0: state valid, bit1: likelihood wanted (chain.cc:980)
  const double* llike_new;  // [Nc] filled by the host for gated chains
  // host-side proposals (ptm_set_proposal_callback; lanes kernel, general build): the proposed states arrive in xprop
  // (whole states, row layout) with their log-Hastings ratio, type code and validity; acc_out gets the outcome
  // compacted sweep (lean MFMA build, ptm_mfma_kernel.hpp): per local rung the walkers that make a Metropolis move this
  // step, packed at cidx[rl * W ..) by partition_kernel; ccnt[rl] of them.  null: every chain is visited in place.
  const int* cidx;
  const int* ccnt;
  uint64_t init_base;            // init_prior_kernel: first attempt number of this initial draw (ptm_init_from_prior_k)
  int host_prop;
  const double* hastings;        // [Nc]
  const int* htype;              // [Nc]
  const unsigned char* hvalid;   // [Nc]
  unsigned char* acc_out;        // [Nc] 1 accepted, 0 rejected, 2 no move (exchanged rung)
};

// ------------------------------------------------------------------------------------------------
// boundary::enforce (states.cc:11-58)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool boundary_enforce(int lo, int hi, double xmin, double xmax, double& x) {
  if ((lo == B_WRAP) != (hi == B_WRAP)) return false;
  if (lo == B_WRAP) {
    const double width = xmax - xmin;
    if (width <= 0) return false;
    double xt = fmod_det(x - xmin, width);
    if (xt < 0) xt += width;
    x = xmin + xt;
    return true;
  }
  if (lo == B_REFLECT && hi == B_REFLECT) {
    const double halfwidth = xmax - xmin;
    if (halfwidth <= 0) return false;
    const double width = 2 * halfwidth;
    double xt = fmod_det(x - xmin, width);
    if (xt < 0) xt += width;
    if (xt >= halfwidth) xt = halfwidth - xt;  // as the reference folds it (states.cc:41)
    x = xmin + xt;
    return true;
  }
  if (lo == B_REFLECT && x < xmin) x = xmin + (xmin - x);
  else if (hi == B_REFLECT && x > xmax) x = xmax - (x - xmax);
  if (lo == B_LIMIT && x < xmin) return false;
  if (hi == B_LIMIT && x > xmax) return false;
  return true;
}

// one factor of mixed_dist_product::evaluate (probability_function.cc:281-304; pdfs ProbabilityDist.h:88-247)
__device__ __forceinline__ double prior_pdf(int type, double lo, double hi, double coef, double x) {
  switch (type) {
    case P_FLAT: return 1.0;
    case P_UNIFORM:
      if (x < lo) return 0.0;
      if (x > hi) return 0.0;
      return coef;
    case P_GAUSSIAN: {
      const double xn = (x - lo) / hi;
      return dexp(-xn * xn / 2) / 2.5066282746310002 / hi;
    }
    case P_POLAR:
      if (x < lo) return 0.0;
      if (x > hi) return 0.0;
      return dsin_0_pi(x) / coef;
    case P_COPOLAR:
      if (x < lo) return 0.0;
      if (x > hi) return 0.0;
      return dcos_hpi(x) / coef;
    case P_LOG:
      if (x < lo) return 0.0;
      if (x > hi) return 0.0;
      return 1 / coef / x;
  }
  return __builtin_nan("");
}

// stateSpace::enforce (states.cc:86-102) + sampleable_probability_function::evaluate_log (probability_function.hh:59).
// Runs over the true dimensions only (pad dimensions are open/flat by construction).  Beyond 64 padded dimensions the
// per-lane loops of this file stay loops (`#pragma unroll DP_UNROLL`): only the set-up kernels (evaluate / init) get
// there, and their unrolled forms were hundreds of kilobytes of code and minutes of compile time.
#ifndef PTM_ENFORCE_INLINE_MAX
#define PTM_ENFORCE_INLINE_MAX 32   // padded dimensions up to which enforce_and_lprior is inlined: a call keeps the state array and the parameter block in scratch -- 0.6-1.1 KB per lane that every launch of a general kernel paid for, and twice the registers (DP = 8: 130 -> 76 VGPRs)
#endif
template <int DP>
__device__ __attribute__((noinline)) double enforce_and_lprior_call(const Dev& p, double (&x)[DP], bool& valid);
template <int DP>
__device__ __forceinline__ double enforce_and_lprior_body(const Dev& p, double (&x)[DP], bool& valid) {
  constexpr int DP_UNROLL = DP > 64 ? 1 : DP;   // (beyond 64 padded dimensions the per-dimension loops stay loops)
  cip blo = as_c(p.blo), bhi = as_c(p.bhi), pt = as_c(p.ptype);
  cdp bmin = as_c(p.bmin), bmax = as_c(p.bmax), plo = as_c(p.plo), phi = as_c(p.phi), pco = as_c(p.pcoef);
  if (valid && p.has_bounds) {
#pragma unroll DP_UNROLL
    for (int d = 0; d < DP; ++d)
      if (d < p.D && valid) valid = boundary_enforce(blo[d], bhi[d], bmin[d], bmax[d], x[d]);
  }
  if (!valid) return -__builtin_inf();
  if (p.all_uniform) {
    bool in = true;
#pragma unroll DP_UNROLL
    for (int d = 0; d < DP; ++d)
      if (d < p.D) in = in && !(x[d] < plo[d]) && !(x[d] > phi[d]);
    return in ? p.lprior_const : -__builtin_inf();
  }
  // four interleaved partial products, combined as ((p0 p1) p2) p3: the order every path and the CPU checker share
  double pq[4] = {1.0, 1.0, 1.0, 1.0};
#pragma unroll DP_UNROLL
  for (int d = 0; d < DP; ++d)
    if (d < p.D) pq[d & 3] *= prior_pdf(pt[d], plo[d], phi[d], pco[d], x[d]);
  const double result = ((pq[0] * pq[1]) * pq[2]) * pq[3];
  return dlog(result);
}
template <int DP>
__device__ __attribute__((noinline)) double enforce_and_lprior_call(const Dev& p, double (&x)[DP], bool& valid) { return enforce_and_lprior_body<DP>(p, x, valid); }
template <int DP>
__device__ __forceinline__ double enforce_and_lprior(const Dev& p, double (&x)[DP], bool& valid) {
  if constexpr (DP <= PTM_ENFORCE_INLINE_MAX) return enforce_and_lprior_body<DP>(p, x, valid);
  else return enforce_and_lprior_call<DP>(p, x, valid);
}

// like0 - 1/2 y^T P y in the order shared by every path (and the CPU checker): s_i = P_ii y_i + sum_{j<i} 2P_ij y_j as
// one fma chain per row (j ascending: exactly what a column of f64 MFMA tiles accumulates), then the dot product y.s
// in four interleaved partial sums p_q = sum_{i = q mod 4} y_i s_i combined as ((p0 + p1) + p2) + p3.
template <int DP, bool MEAN, class XV>
__device__ __forceinline__ double gauss_llike(const Dev& p, const XV& x) {
  constexpr int DP_UNROLL = DP > 64 ? 1 : DP;   // (beyond 64 padded dimensions the per-dimension loops stay loops)
  double pq[4] = {0.0, 0.0, 0.0, 0.0};
  cdp row = as_c(p.P2);
  cdp mean = as_c(p.mean);
#pragma unroll DP_UNROLL
  for (int i = 0; i < DP; ++i) {
    double s = 0;
#pragma unroll DP_UNROLL
    for (int j = 0; j < i; ++j) {
      const double yj = MEAN ? x[j] - mean[j] : x[j];
      s = __builtin_fma(row[j], yj, s);
    }
    const double yi = MEAN ? x[i] - mean[i] : x[i];
    s = __builtin_fma(row[i], yi, s);
    pq[i & 3] = __builtin_fma(yi, s, pq[i & 3]);
    row += i + 1;
  }
  const double q = ((pq[0] + pq[1]) + pq[2]) + pq[3];
  return p.like0 - 0.5 * q;
}

// ------------------------------------------------------------------------------------------------
// gaussian_prop::draw (proposal_distribution.hh:194-218): offset = factor * z, z ~ N(0,1)^D drawn four at a time
// (one Philox block = two Box-Muller pairs).  Written as REAL loops over Philox blocks and factor columns -- only
// the row index is unrolled, so the accumulators stay in registers, the code stays small, and the staged-table
// reads of one column sit next to their FMAs (a fully unrolled version lets the scheduler hoist every table read
// to the top and spill).  Lower-triangular factors are stored in column panels of 8 (4 for DP=4): panel P holds
// columns [PC*P, PC*P+PC) x rows [PC*P, DP); the (few) structural zeros inside a panel contribute fma(0,z,acc)=acc.
// ------------------------------------------------------------------------------------------------
struct DrawCtx {
  uint64_t seed, step;
  uint32_t stream;
  int axis;  // >= 0: one-dimensional move along that axis (proposal_distribution.hh:197-205), -1: full move
  const double* bmtab;  // Box-Muller radius table (the block's LDS copy)
};
__device__ __forceinline__ void draw4(const DrawCtx& dc, int b, double (&z)[4]) {
  const u32x4 o = draw_block(dc.seed, TAG_MH, dc.stream, dc.step, (uint32_t)(b + 1));
  boxmuller(o.v0, o.v1, dc.bmtab, z[0], z[1]);
  boxmuller(o.v2, o.v3, dc.bmtab, z[2], z[3]);
  if (dc.axis >= 0) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (4 * b + t != dc.axis) z[t] = 0.0;
  }
}

// Position of dimension d inside a stored row.  For DP == 32, 64 and 128 rows are kept in the accumulator layout of the f64 MFMA
// kernel: lane group q of a wave owns the eight dimensions d = q + 4m of a chain and moves them as four 16-byte
// pieces {m = 2t, 2t+1}; piece t of lane group q sits at 16-byte slot 4t + q, so that ONE load instruction of the
// chain's four lanes covers 64 contiguous bytes (a quarter of the row) instead of four scattered 16-byte pieces.
// Every [n][DP] row image (states, proposals handed to a host callback, boundary messages) uses this layout; the
// host converts on set / get.
template <int DP>
__device__ __forceinline__ constexpr int row_pos(int d) {
  return (DP == 32 || DP == 64 || DP == 128) ? 8 * ((d >> 2) >> 1) + 2 * (d & 3) + ((d >> 2) & 1) : d;
}

// Column order of the product (shared with the CPU checker and the MFMA kernel, whose tile steps fix it): natural for
// DP <= 8; for DP >= 16 in halves of 16 columns, inside a half s + 4k with s = 0..3 outer and k = 0..3 inner -- i.e.
// slot s of the half's four Philox blocks k = 0..3.  Factors are stored dense column-major [col][row] (DP*DP per rung,
// zeros above the diagonal of a Cholesky factor); LOWER skips the leading rows of a column block, whose entries are
// structural zeros (fma(0, z, acc) == acc: same bits either way).
template <int DP, int KIND, class TP>
__device__ __forceinline__ void factor_product(const DrawCtx& dc, TP tab, double (&acc)[DP]) {
  if constexpr (DP <= 8) {
#pragma unroll 1
    for (int bb = 0; bb < DP / 4; ++bb) {
      double z[4];
      draw4(dc, bb, z);
#pragma unroll 1
      for (int t = 0; t < 4; ++t) {
        const double zj = z[0];
        z[0] = z[1]; z[1] = z[2]; z[2] = z[3]; z[3] = zj;  // rotate: keeps the register index static
        TP col = tab + (bb * 4 + t) * DP;
#pragma unroll
        for (int i = 0; i < DP; ++i) acc[i] = __builtin_fma(col[i], zj, acc[i]);
      }
    }
  } else {
#pragma unroll
    for (int h = 0; h < DP / 16; ++h) {
      double zz[4][4];
#pragma unroll
      for (int k = 0; k < 4; ++k) draw4(dc, 4 * h + k, zz[k]);
#pragma unroll 1
      for (int sl = 0; sl < 4; ++sl) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const double zj = zz[k][0];
          zz[k][0] = zz[k][1]; zz[k][1] = zz[k][2]; zz[k][2] = zz[k][3]; zz[k][3] = zj;
          TP col = tab + (16 * h + 4 * k + sl) * DP;
          constexpr int dummy = 0; (void)dummy;
          const int R0 = (KIND == KIND_LOWER) ? 16 * h + 4 * k : 0;   // compile-time after unrolling h and k
#pragma unroll
          for (int i = 0; i < DP; ++i)
            if (i >= R0) acc[i] = __builtin_fma(col[i], zj, acc[i]);
        }
      }
    }
  }
}

// ---- wave-uniform rung: broadcast-by-DPP factor product ------------------------------------------------------
// The rung's factor (dense column-major in LDS, zeros above the diagonal for a Cholesky factor) is the same for all
// 64 lanes.  Feeding it to v_fma_f64 as a broadcast LDS read costs a full 512-B register write-back per operand
// (4 LDS clocks each: a quarter of the FMA rate), and as SGPR operands it misses the scalar cache (one factor per
// rung).  Instead each column is read ONCE per 16 rows with every lane fetching "its" row element
// (lane l <- T[l & 15 (+16)][j], 512 B per instruction, conflict-free), and the FMA takes its table operand from
// lane i of each 16-lane row through the DPP row_newbcast control -- one VALU instruction per multiply-add, no
// operand traffic.  All 64 lanes must be active (DPP reads lanes, not memory): callers run this with full EXEC.
#define PTM_DPP_FMA(I) "v_fmac_f64_dpp %" #I ", %16, %17 row_newbcast:" #I " row_mask:0xf bank_mask:0xf\n\t"
template <int NROW16>  // how many of the 16 rows of this half are live (16, or 8 for the half-height panels)
__device__ __forceinline__ void dpp_fma16(double* a, double col, double z) {
  // rows [16 - NROW16, 16) of the half: the leading rows of a lower-triangular panel are structural zeros
  if constexpr (NROW16 == 16) {
    asm("s_nop 1\n\t" PTM_DPP_FMA(0) PTM_DPP_FMA(1) PTM_DPP_FMA(2) PTM_DPP_FMA(3) PTM_DPP_FMA(4) PTM_DPP_FMA(5) PTM_DPP_FMA(6)
        PTM_DPP_FMA(7) PTM_DPP_FMA(8) PTM_DPP_FMA(9) PTM_DPP_FMA(10) PTM_DPP_FMA(11) PTM_DPP_FMA(12) PTM_DPP_FMA(13)
        PTM_DPP_FMA(14) PTM_DPP_FMA(15)
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
          "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
        : "v"(col), "v"(z));
  } else {
    asm("s_nop 1\n\t" PTM_DPP_FMA(8) PTM_DPP_FMA(9) PTM_DPP_FMA(10) PTM_DPP_FMA(11) PTM_DPP_FMA(12) PTM_DPP_FMA(13)
        PTM_DPP_FMA(14) PTM_DPP_FMA(15)
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
          "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
        : "v"(col), "v"(z));
  }
}

// half h (columns [16h, 16h+16)) of a 32-dimensional factor staged dense column-major at `tab` (LDS), in the shared
// column order; LOWER skips the row blocks that are structurally zero
template <int KIND, int H>
__device__ __forceinline__ void dpp_half32(const DrawCtx& dc, const double* tab, int lane16, double (&acc)[32]) {
  double zz[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) draw4(dc, 4 * H + k, zz[k]);
#pragma unroll 1
  for (int sl = 0; sl < 4; ++sl) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double zj = zz[k][0];
      zz[k][0] = zz[k][1]; zz[k][1] = zz[k][2]; zz[k][2] = zz[k][3]; zz[k][3] = zj;
      const double* col = tab + (16 * H + 4 * k + sl) * 32 + lane16;
      if (KIND == KIND_DENSE || H == 0) {          // rows 0..15: only columns < 16 reach them in a Cholesky factor
        const double ca = col[0];
        if (KIND == KIND_DENSE || k < 2) dpp_fma16<16>(&acc[0], ca, zj); else dpp_fma16<8>(&acc[0], ca, zj);
      }
      const double cb = col[16];                   // rows 16..31
      if (KIND == KIND_DENSE || H == 0 || k < 2) dpp_fma16<16>(&acc[16], cb, zj); else dpp_fma16<8>(&acc[16], cb, zj);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// differential_evolution::draw (proposal_distribution.cc:476-592; draw_i_from_chain :745-801 with unlikely_alpha = 0, no temperature
// mixing: the reference sampler's defaults, ptmcmc.cc:81-91) for ONE chain on one lane, from the chain's own saved history on the
// device: rows 0 .. de_init_extra - 1 of de_init, then the history ring.  The oracle states the same operations (ptmo_de_draw):
// state::scalar_mult then state::add -- every product rounded before its sum --, innerprod in index order.  History rows are
// read per dimension as they are used (no register image of a row).  The draw's uniforms: Philox block 0x0DE00000 of the chain's
// MH stream = {snooker test, gamma, pick of z1, pick of z2}; blocks 0x0DE00001 + t: the t-th attempt at the snooker move's z.
// Returns the move's type (0 parallel, 1 snooker), xn = the proposed state (natural order, pad dimensions 0).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool de_ready(const Dev& p, unsigned int nh0) {   // differential_evolution::is_ready(): 10 D rows
  const long long rows = p.de_init_extra + 1 + (long long)((nh0 + (unsigned int)p.add_every_n - 1u) / (unsigned int)p.add_every_n);
  return rows >= 10ll * p.D;
}
template <int DP>
__device__ __forceinline__ int de_draw(const Dev& p, int c, uint32_t stream, uint64_t step, unsigned int nh0, const double* __restrict__ row, double (&xn)[DP],
                                       double& log_hastings) {
  // (every loop runs over the padded dimensions with a guard: xn stays in registers -- a run-time trip count would index it dynamically)
  constexpr int DU = DP > 64 ? 1 : DP;
  const int D = p.D;
  const long long saved = 1 + (long long)((nh0 + (unsigned int)p.add_every_n - 1u) / (unsigned int)p.add_every_n);   // rows of the ring so far
  const long long rows = p.de_init_extra + saved;
  auto pick = [&](double u) -> const double* {
    const long long spare = rows - 100ll * D;
    const long long first = (spare * (1 - p.de_ignore) > 10ll * D) ? (long long)(spare * p.de_ignore) : 0;
    const long long r = (long long)(first + (rows - first) * u);
    if (r < p.de_init_extra) return p.de_init + ((size_t)r * p.Nc + c) * DP;
    const long long hr = r - p.de_init_extra;
    if (saved - hr > p.hist.cap) atomicOr(p.err, 64);   // the ring has lost that row: it must hold the whole run (ptm_set_proposal_de)
    return p.hist.x + hist_slot(p.hist, hr, c) * DP;
  };
  const u32x4 b0 = draw_block(p.seed, TAG_MH, stream, step, 0x0DE00000u);
  if (!(p.de_snooker > u01(b0.v0))) {                  // draw_standard
    const double gamma = u01(b0.v1) < p.de_gamma_one ? 1.0 : p.de_gamma_std;
    const double* z1 = pick(u01(b0.v2));
    const double* z2 = pick(u01(b0.v3));
#pragma unroll DU
    for (int d = 0; d < DP; ++d) {
      xn[d] = 0.0;
      if (d < D) {
        const int q = row_pos<DP>(d);
        const double t1 = z1[q] * gamma;
        const double a = row[q] + t1;
        const double t2 = z2[q] * (-gamma);
        xn[d] = a + t2;
      }
    }
    log_hastings = 0.0;
    return 0;
  }
  const double gamma = (1.2 + u01(b0.v1)) / p.de_gamma_div;   // draw_snooker
  const double* z = row;
  double axis2 = 0.0;
  for (int tries = 0; axis2 == 0.0; ++tries) {         // the history repeats states: z must differ from the current state
    if (tries > 1000) {                                // (the reference exits here; the engine raises an error bit and rejects the move)
      atomicOr(p.err, 128);
      log_hastings = __builtin_nan("");
#pragma unroll DU
      for (int d = 0; d < DP; ++d) xn[d] = d < D ? row[row_pos<DP>(d)] : 0.0;
      return 1;
    }
    const u32x4 bt = draw_block(p.seed, TAG_MH, stream, step, 0x0DE00001u + (uint32_t)tries);
    z = pick(u01(bt.v0));
    axis2 = 0.0;
#pragma unroll 1
    for (int d = 0; d < D; ++d) { const int q = row_pos<DP>(d); const double a = row[q] + z[q] * (-1.0); axis2 = axis2 + a * a; }
  }
  const double* z1 = pick(u01(b0.v2));
  const double* z2 = pick(u01(b0.v3));
  double proj = 0.0;
#pragma unroll 1
  for (int d = 0; d < D; ++d) {
    const int q = row_pos<DP>(d);
    const double a = z1[q] * gamma, b = z2[q] * (-gamma);
    const double diff = a + b;
    const double ax = row[q] + z[q] * (-1.0);
    proj = proj + diff * ax;
  }
  proj = proj / axis2;
  double fz2 = 0.0;
#pragma unroll DU
  for (int d = 0; d < DP; ++d) {
    xn[d] = 0.0;
    if (d < D) {
      const int q = row_pos<DP>(d);
      const double ax = row[q] + z[q] * (-1.0);
      const double t = ax * proj;
      const double y = row[q] + t;
      xn[d] = y;
      const double f = y + z[q] * (-1.0);
      fz2 = fz2 + f * f;
    }
  }
  log_hastings = (dlog(fz2) - dlog(axis2)) * (double)(D - 1) / 2.0;
  return 1;
}

// ------------------------------------------------------------------------------------------------
// The general fused sweep kernel: one MH_chain::step (chain.cc:966-1022) per lane, all rungs x walkers per launch:
//   gaussian_prop::draw (proposal_distribution.hh:194-218) -> state::add / enforce (states.cc:205-214,161-166)
//   -> prior -> Gaussian likelihood -> Metropolis test -> add_state counters and history (chain.cc:916-949).
// Rungs that took part in an exchange attempt this step (touch[] > 0) make no move (chain.cc:1553-1557).
//   UNI    the wave's 64 chains share one rung (W % 64 == 0): factor / beta addresses are wave-uniform (SGPR operands)
//   SIMPLE open boundaries, all-uniform prior, zero mean, no one-dimensional moves (the BASELINE workload):
//          the general state-space / prior code is not even compiled in.
// ------------------------------------------------------------------------------------------------
template <int DP, int KIND, bool UNI, bool SIMPLE>
__global__ __launch_bounds__(256, PTM_SWEEP_WAVES) void sweep_kernel(const Dev p) {
  // Per-wave LDS staging of the rung's proposal factor (UNI only): every rung has its own D x D factor, so unlike
  // the shared precision matrix it misses the scalar cache; one coalesced 512-B-per-instruction copy into LDS per
  // wave, then wave-uniform (broadcast) LDS reads feed the mat-vec.  Same-wave LDS traffic only: no barrier.
  // The first 20 KB of the block's LDS hold the Box-Muller tables (per-lane gathers, one 16-byte entry per draw each).
  extern __shared__ __attribute__((aligned(16))) double lds_all[];
  double* lds_fac = lds_all + BM_TABLE_DOUBLES;
#pragma unroll
  for (int t = 0; t < BM_TABLE_DOUBLES / 512; ++t)
    reinterpret_cast<bm_d2*>(lds_all)[threadIdx.x + 256 * t] = reinterpret_cast<const bm_d2*>(BM_TABLE)[threadIdx.x + 256 * t];
  const int c = p.c_begin + blockIdx.x * 256 + threadIdx.x;
  int rl = (c < p.c_end ? c : p.c_end - 1) / p.W;
  if (UNI) rl = __builtin_amdgcn_readfirstlane(rl);
  // (DP == 32: the DPP product wants the dense column-major image, which the host keeps beside the packed one)
  const int fstride = p.prop_stride;
  double* myfac = lds_fac + (threadIdx.x >> 6) * fstride;
  if (UNI && KIND != KIND_DIAG) {
    const double* g = p.prop + (size_t)rl * fstride;
    for (int k = threadIdx.x & 63; k < fstride; k += 64) myfac[k] = g[k];
  }
  // the table is shared by the block's four waves: wait for this wave's LDS writes only (a full __syncthreads would
  // also drain the factor loads just issued) and meet the others
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (c >= p.c_end) return;
  const int w = c - rl * p.W;
  const int rg = p.r0 + rl;

  const int tc = p.touch[c];  // > 0: the rung took part in that many exchange attempts => no MH move this step
  const bool propose_only = !SIMPLE && p.mode == 1;
  if (tc && !propose_only) {
    const unsigned int nh0 = p.nhist[c];
    p.nhist[c] = nh0 + (unsigned int)tc;  // one add_state per attempt (chain.cc:1487-1490,1531-1534,1554-1557)
    p.touch[c] = 0;                       // (the exchange kernel already moved the rows)
    // history: the LAST of these adds saw the row as it is now (an earlier one of two saw the intermediate row: the
    // exchange kernels save that one, ptm_aux_kernels.hpp)
    const unsigned int a = nh0 + (unsigned int)tc - 1u;
    if (rl < p.hist.rungs && a % (unsigned int)p.add_every_n == 0u) {
      const long long row = 1 + (long long)(a / (unsigned int)p.add_every_n);
      const size_t o = hist_slot(p.hist, row, c);
      for (int d = 0; d < DP; ++d) p.hist.x[o * DP + d] = p.x[(size_t)c * DP + d];
      hist_scalars(p.hist, o, row, p.ll[c], p.lp[c], p.naccept[c], p.ntries[c], p.last_type[c],
                   p.beta_add ? p.beta_add[c] : as_c(p.beta)[rg]);
    }
    if (rl < p.map.rungs) {   // MAP: the row the last add saw, at the temperature the rung had then
      const double tl = p.ll[c], tp = p.lp[c];
      const double tb = (p.beta_add ? p.beta_add[c] : as_c(p.beta)[rg]) * tl;
      if (map_try(p.map, c, tp + tb, tl, tp))
        for (int d = 0; d < DP; ++d) p.map.x[(size_t)c * DP + d] = p.x[(size_t)c * DP + d];
    }
  }
  // ---- MH_chain::step for the untouched rungs; touched lanes idle through the draw loops
  const uint32_t stream = (uint32_t)(w + p.w_off) * (uint32_t)p.Nt + (uint32_t)rg;
  const u32x4 o0 = draw_block(p.seed, TAG_MH, stream, p.step, 0);

  // -- gaussian_prop::draw: D normals, optional one-dimensional move, offset = factor * z
  int type = 0, axis = -1, kmix = 0;
  double mix_scale = 1.0;
  if (!SIMPLE) {
    double f = as_c(p.onedfrac)[rl];
    if (p.mix_K > 0) {   // proposal_distribution_set::draw: one uniform picks the member (a set of one draws nothing)
      cdp mx = as_c(p.mix) + (size_t)rl * p.mix_K * 3;
      const double xs = p.mix_K > 1 ? u01(o0.v3) : 0.0;
      kmix = p.mix_K - 1;
      for (int k = p.mix_K - 2; k >= 0; --k)
        if (xs < mx[3 * k]) kmix = k;
      // differential evolution: a member that is not ready yet (fewer than 10 D saved rows) is passed over, as
      // proposal_distribution_set::draw passes over it (proposal_distribution.cc:111): the next member's bin is met
      if (p.de_on && mx[3 * kmix + 1] < 0 && kmix + 1 < p.mix_K && !de_ready(p, p.nhist[c])) kmix += 1;
      mix_scale = mx[3 * kmix + 1];
      f = mx[3 * kmix + 2];
    }
    if (p.any_oned && !tc && f > 0 && u01(o0.v1) < f) { axis = (int)(p.D * u01(o0.v2)); type = 1; }
  }
  const bool de_move = !SIMPLE && p.de_on && mix_scale < 0;
  double de_hast = 0.0;
  double xn[DP];  // accumulates the offset, then becomes the proposed state
#pragma unroll
  for (int i = 0; i < DP; ++i) xn[i] = 0.0;
  const DrawCtx dc{p.seed, p.step, stream, (!SIMPLE) ? axis : -1, lds_all};
  const int mode = SIMPLE ? 0 : p.mode;

  // (touched lanes run the draw too: the DPP product needs every lane of the wave active, and they would idle anyway)
  if (mode == 2) {
    // accept pass of the host-callback path: the proposal was drawn and stored by the propose pass
  } else if (KIND == KIND_DIAG) {
    cdp fac = as_c(p.prop) + (size_t)rl * p.prop_stride;
#pragma unroll
    for (int b = 0; b < DP / 4; ++b) {
      double z[4];
      draw4(dc, b, z);
#pragma unroll
      for (int t = 0; t < 4; ++t) xn[4 * b + t] = fac[4 * b + t] * z[t];
    }
  } else if (UNI && DP == 32) {
    if constexpr (DP == 32) {
      const int lane16 = threadIdx.x & 15;
      dpp_half32<KIND, 0>(dc, myfac, lane16, xn);
      dpp_half32<KIND, 1>(dc, myfac, lane16, xn);
    }
  } else if (UNI) {
    factor_product<DP, KIND>(dc, (const double*)myfac, xn);
  } else {
    factor_product<DP, KIND>(dc, as_c(p.prop) + (size_t)rl * p.prop_stride, xn);
  }

  if (tc) {        // touched rung: no MH move (its add_state calls were counted above)
    if (!SIMPLE && mode == 1) p.gate[c] = 0;   // propose pass of the host-callback path: nothing to evaluate
    return;
  }

  // -- current state: read only now (its registers are not live across the draw loops) and folded straight into the
  //    proposal.  The row is written back only if the move is accepted.
  const double ll = p.ll[c], lp = p.lp[c];
  double* __restrict__ row = p.x + (size_t)c * DP;
  if (de_move) {
    // the member is differential evolution: the proposed state and its log-Hastings ratio from the chain's saved history (the accept
    // pass of a host-callback likelihood takes ratio and type from the propose pass)
    int dt;
    if (mode == 2) { de_hast = p.de_hast[c]; dt = p.de_type[c]; }
    else {
      dt = de_draw<DP>(p, c, stream, p.step, p.nhist[c], row, xn, de_hast);
      if (mode == 1) { p.de_hast[c] = de_hast; p.de_type[c] = dt; }
    }
    type = kmix + 10 * dt;     // proposal_distribution.cc:117
  } else if (!SIMPLE && p.mix_K > 0) {
    type = kmix + 10 * type;   // proposal_distribution.cc:117
    if (mode != 2) {
#pragma unroll
      for (int d = 0; d < DP; ++d) xn[d] = mix_scale * xn[d];   // the member is scale_k times the rung's factor
    }
  }
  if (mode != 2 && !de_move) {
#pragma unroll
    for (int d = 0; d < DP; ++d) xn[d] = row[row_pos<DP>(d)] + xn[d];  // state::add (states.cc:205-214)
  }
  const double beta = (!SIMPLE && p.betaC) ? p.betaC[c] : as_c(p.beta)[rg];
  const double bl = beta * ll;
  const double cur_lpost = lp + bl;
  const double oldlprior = cur_lpost - bl;  // chain.cc:973

  bool valid;
  double newlprior;
  if (SIMPLE) {
    valid = true;
    // the box of the all-uniform prior, eight dimensions at a time: one live lane mask, not 2*DP compare results
    uint64_t box_mask = ~0ull;
    cdp plo = as_c(p.plo), phi = as_c(p.phi);
#pragma unroll
    for (int d0 = 0; d0 < DP; d0 += (DP < 8 ? DP : 8)) {
      bool ok8 = true;
#pragma unroll
      for (int d = d0; d < d0 + (DP < 8 ? DP : 8); ++d) ok8 = ok8 & !(xn[d] < plo[d]) & !(xn[d] > phi[d]);
      box_mask &= __builtin_amdgcn_ballot_w64(ok8);
      asm volatile("" : "+s"(box_mask));
    }
    const bool in = ((box_mask >> (threadIdx.x & 63)) & 1ull) != 0;
    newlprior = in ? p.lprior_const : -__builtin_inf();
  } else if (mode == 2) {
    valid = (p.gate[c] & 1) != 0;
    newlprior = p.lprior_new[c];
#pragma unroll
    for (int d = 0; d < DP; ++d) xn[d] = p.xprop[(size_t)c * DP + row_pos<DP>(d)];
  } else {
    valid = p.origin_valid != 0;  // Q9: the sum is built on an enforced zero state
    newlprior = enforce_and_lprior<DP>(p, xn, valid);
  }
  const bool want_like = valid && (newlprior > -1e200 || newlprior - oldlprior > p.min_prior);  // chain.cc:980 (Q1)
  if (mode == 1) {
    // propose pass: hand the proposal to the host, change nothing else
#pragma unroll
    for (int d = 0; d < DP; ++d) p.xprop[(size_t)c * DP + row_pos<DP>(d)] = xn[d];
    p.lprior_new[c] = newlprior;
    p.gate[c] = (unsigned char)((valid ? 1 : 0) | (want_like ? 2 : 0));
    return;
  }
  double newlike, newlpost;
  if (mode == 2) {
    newlike = want_like ? p.llike_new[c] : -__builtin_inf();
    newlpost = want_like ? newlike * beta + newlprior : -__builtin_inf();
  } else if (want_like) {
    newlike = (!SIMPLE && p.has_mean) ? gauss_llike<DP, true>(p, xn) : gauss_llike<DP, false>(p, xn);
    newlpost = newlike * beta + newlprior;
  } else {
    newlike = newlpost = -__builtin_inf();
  }
  double logH = newlpost - cur_lpost;  // gaussian_prop: log_hastings_ratio() == 0
  bool accept = valid;
  if (de_move) {                       // chain.cc:989-994: prop.log_hastings_ratio(), NaN => reject
    if (de_hast != de_hast) accept = false;
    logH = de_hast + logH;
  }
  if (accept && logH < 0) accept = dlog_u01(o0.v0) < logH;  // chain.cc:998-1001 (NaN stays accepted)

  const int ntries1 = p.ntries[c] + 1;
  p.ntries[c] = ntries1;
  const unsigned int nh0 = p.nhist[c];
  p.nhist[c] = nh0 + 1u;
  if (rl < p.hist.rungs && nh0 % (unsigned int)p.add_every_n == 0u) {   // add_state saves this one (chain.cc:935-946)
    const long long hrow = 1 + (long long)(nh0 / (unsigned int)p.add_every_n);
    const size_t o = hist_slot(p.hist, hrow, c);
    if (accept) {
#pragma unroll
      for (int d = 0; d < DP; ++d) p.hist.x[o * DP + row_pos<DP>(d)] = xn[d];
      hist_scalars(p.hist, o, hrow, newlike, newlprior, p.naccept[c] + 1, ntries1, type, beta);
    } else {
      for (int d = 0; d < DP; ++d) p.hist.x[o * DP + d] = row[d];
      hist_scalars(p.hist, o, hrow, ll, lp, p.naccept[c], ntries1, p.last_type[c], beta);
    }
  }
  if (accept && rl < p.map.rungs && map_try(p.map, c, newlpost, newlike, newlprior)) {   // MAP (chain.cc:931-934)
#pragma unroll
    for (int d = 0; d < DP; ++d) p.map.x[(size_t)c * DP + row_pos<DP>(d)] = xn[d];
  } else if (!SIMPLE && !accept && p.betaC && rl < p.map.rungs && map_try(p.map, c, cur_lpost, ll, lp)) {
    // an evolving ladder: the state that stays is added at a NEW temperature and may beat the MAP with it
    for (int d = 0; d < DP; ++d) p.map.x[(size_t)c * DP + d] = row[d];
  }
  if (accept) {
    p.naccept[c] += 1;
    p.last_type[c] = type;
#pragma unroll
    for (int d = 0; d < DP; ++d) row[row_pos<DP>(d)] = xn[d];
    p.ll[c] = newlike;
    p.lp[c] = newlprior;
  }
}

// ------------------------------------------------------------------------------------------------
// evaluation of given states (set_states / debug_evaluate): enforce, lprior, llike
// ------------------------------------------------------------------------------------------------
template <int DP>
__global__ __launch_bounds__(256) void evaluate_kernel(const Dev p, int n, double* x_io /*[n][DP]*/, int* valid_out,
                                                        double* lprior_out, double* llike_out, int eval_like) {
  constexpr int DP_UNROLL = DP > 64 ? 1 : DP;   // (beyond 64 padded dimensions the per-dimension loops stay loops)
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= n) return;
  double x[DP];
#pragma unroll DP_UNROLL
  for (int d = 0; d < DP; ++d) x[d] = x_io[(size_t)c * DP + row_pos<DP>(d)];
  bool valid = true;  // state(space, values) constructor: valid unless enforce fails (states.cc:194-199)
  const double lp = enforce_and_lprior<DP>(p, x, valid);
#pragma unroll DP_UNROLL
  for (int d = 0; d < DP; ++d) x_io[(size_t)c * DP + row_pos<DP>(d)] = x[d];
  if (valid_out) valid_out[c] = valid ? 1 : 0;
  lprior_out[c] = lp;
  if (eval_like) llike_out[c] = p.has_mean ? gauss_llike<DP, true>(p, x) : gauss_llike<DP, false>(p, x);
}

// MH_chain::initialize(1) (chain.cc:846-876): redraw from the prior until valid and llike >= -1e100.
// Dimension d of attempt a uses block d of the chain's INIT stream with step = a.
// With a host-callback likelihood (cb_attempt >= 0) one launch makes ONE attempt for the chains not yet done and
// reports validity in `pending` (1 = drawn a valid state that awaits its likelihood, 2 = done earlier); the host
// evaluates the plug-in and relaunches for the rest.
template <int DP>
__global__ __launch_bounds__(256) void init_prior_kernel(const Dev p, double* x_out, double* ll_out, double* lp_out,
                                                          int* fail, long long cb_attempt, unsigned char* pending) {
  constexpr int DP_UNROLL = DP > 64 ? 1 : DP;   // (beyond 64 padded dimensions the per-dimension loops stay loops)
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= p.Nc) return;
  if (cb_attempt >= 0 && pending[c] == 2) return;
  const int rl = c / p.W, w = c - rl * p.W;
  const uint32_t stream = (uint32_t)(w + p.w_off) * (uint32_t)p.Nt + (uint32_t)(p.r0 + rl);
  cip pt = as_c(p.ptype);
  cdp plo = as_c(p.plo), phi = as_c(p.phi);
  double x[DP];
  double ll = 0, lp = 0;
  bool done = false;
  const uint64_t a_begin = p.init_base + (cb_attempt >= 0 ? (uint64_t)cb_attempt : 0), a_end = cb_attempt >= 0 ? a_begin + 1 : p.init_base + 100000;
  for (uint64_t a = a_begin; a < a_end && !done; ++a) {
#pragma unroll DP_UNROLL
    for (int d = 0; d < DP; ++d) {
      x[d] = 0.0;
      if (d < p.D) {
        const u32x4 o = draw_block(p.seed, TAG_INIT, stream, a, (uint32_t)d);
        if (pt[d] == P_UNIFORM) x[d] = u01(o.v0) * (phi[d] - plo[d]) + plo[d];
        else if (pt[d] == P_GAUSSIAN) { double z0, z1; boxmuller(o.v0, o.v1, (const double*)BM_TABLE, z0, z1); x[d] = z0 * phi[d] + plo[d]; }
        else if (pt[d] == P_POLAR) x[d] = draw_polar(u01(o.v0), plo[d], phi[d]);
        else if (pt[d] == P_COPOLAR) x[d] = draw_copolar(u01(o.v0), plo[d], phi[d]);
        else if (pt[d] == P_LOG) x[d] = draw_log(u01(o.v0), plo[d], phi[d]);
        else x[d] = __builtin_nan("");
      }
    }
    bool valid = true;
    lp = enforce_and_lprior<DP>(p, x, valid);
    if (!valid) continue;
    if (cb_attempt >= 0) { done = true; break; }   // the host evaluates the likelihood of this draw
    ll = p.has_mean ? gauss_llike<DP, true>(p, x) : gauss_llike<DP, false>(p, x);
    if (ll < -1e100) continue;
    done = true;
  }
  if (cb_attempt >= 0) {
    pending[c] = done ? 1 : 0;
    if (!done) return;
  } else if (!done) atomicOr(fail, 1);
#pragma unroll DP_UNROLL
  for (int d = 0; d < DP; ++d) x_out[(size_t)c * DP + row_pos<DP>(d)] = x[d];
  ll_out[c] = ll;
  lp_out[c] = lp;
}

}  // namespace ptm
