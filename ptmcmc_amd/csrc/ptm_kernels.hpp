// ptm_kernels.hpp -- gfx950 kernels of the parallel-tempering step engine.
//
// Data layout in HBM (per engine = per GPU shard), Nc = rung_count * W chains, chain c = rung_local * W + walker:
//   x      [2][DP][Nc] double  SoA state planes, ping-pong (a step reads buffer `cur`, writes the other);
//                              DP = dimension padded to 4/8/16/32 (pad planes stay 0, pad factor rows are 0)
//   llike  [2][Nc], lprior [2][Nc]  double         (lpost is always fl(lprior + fl(beta*llike)), chain.cc:928)
//   ntries, naccept, last_type [Nc] int32; nhist [Nc] uint32          (MH_chain counters, chain.hh:150-170;
//                              Nsize is a function of Nhist: 1 + ceil(nhist / add_every_N), chain.cc:935-947)
//   src [Nc] int32, touch [Nc] uint8   exchange phase -> sweep hand-off: where the chain's state comes from
//                                      this step and how many add_state calls it already received
// Walker is the fastest index, so the 64 lanes of a wave hold 64 walkers of ONE rung whenever W % 64 == 0:
// beta, the proposal factor and the precision matrix are then wave-uniform and are fetched through the scalar
// cache straight into SGPR operands of v_fma_f64 (constant-address-space loads), while every state plane is read
// and written with fully coalesced 512-B wave accesses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ptm_device_math.hpp"

// keeps the scheduler from hoisting every scalar table load of the unrolled mat-vecs to the top of the kernel
// (which spills SGPRs by the thousand); one fence per factor column / precision row
#ifndef PTM_SCHED_FENCE
#define PTM_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

namespace ptm {

enum { KIND_DENSE = 0, KIND_DIAG = 1, KIND_LOWER = 2 };
enum { SRC_ABOVE = -2, SRC_BELOW = -3 };
enum { B_OPEN = 0, B_LIMIT = 1, B_REFLECT = 2, B_WRAP = 3 };
enum { P_FLAT = 0, P_UNIFORM = 1, P_GAUSSIAN = 2, P_POLAR = 3, P_COPOLAR = 4, P_LOG = 5 };

// read-only tables written by the host before any launch: reading them through the constant address space lets
// the backend use scalar loads whenever the address is wave-uniform, with no alias analysis in the way.
typedef const double __attribute__((address_space(4))) * cdp;
typedef const int __attribute__((address_space(4))) * cip;
__device__ __forceinline__ cdp as_c(const double* p) { return (cdp)(uintptr_t)p; }
__device__ __forceinline__ cip as_c(const int* p) { return (cip)(uintptr_t)p; }

// problem description + state pointers; passed BY VALUE to every kernel (kernarg segment => scalar loads)
struct Dev {
  int D, DP, Nt, r0, nloc, W, Nc;
  uint64_t seed, step;
  int add_every_n;
  double min_prior;
  // state space (states.hh:29-48)
  int has_bounds, origin_valid;
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
  const double* prop;      // [nloc][prop_stride]  column-packed factor or sigmas
  const double* onedfrac;  // [nloc]
  int prop_stride, any_oned;
  // state
  const double* x_in;  double* x_out;
  const double* ll_in; double* ll_out;
  const double* lp_in; double* lp_out;
  int *ntries, *naccept, *last_type;
  unsigned int* nhist;
  int* src;
  unsigned char* touch;
  const double *recv_below, *recv_above;  // [(DP+2)][W] rows that crossed the shard boundary this step
  int* err;
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
// Runs over the true dimensions only (pad dimensions are open/flat by construction); not unrolled: this is the
// general path, the lean kernels (SIMPLE) never reach it.
template <int DP>
__device__ __noinline__ double enforce_and_lprior(const Dev& p, double (&x)[DP], bool& valid) {
  cip blo = as_c(p.blo), bhi = as_c(p.bhi), pt = as_c(p.ptype);
  cdp bmin = as_c(p.bmin), bmax = as_c(p.bmax), plo = as_c(p.plo), phi = as_c(p.phi), pco = as_c(p.pcoef);
  if (valid && p.has_bounds) {
#pragma unroll
    for (int d = 0; d < DP; ++d)
      if (d < p.D && valid) valid = boundary_enforce(blo[d], bhi[d], bmin[d], bmax[d], x[d]);
  }
  if (!valid) return -__builtin_inf();
  if (p.all_uniform) {
    bool in = true;
#pragma unroll
    for (int d = 0; d < DP; ++d)
      if (d < p.D) in = in && !(x[d] < plo[d]) && !(x[d] > phi[d]);
    return in ? p.lprior_const : -__builtin_inf();
  }
  double result = 1;
#pragma unroll
  for (int d = 0; d < DP; ++d)
    if (d < p.D) result *= prior_pdf(pt[d], plo[d], phi[d], pco[d], x[d]);
  return dlog(result);
}

// like0 - 1/2 y^T P y in the symmetric-packed order q = sum_i y_i (P_ii y_i + sum_{j<i} 2P_ij y_j)
template <int DP, bool MEAN>
__device__ __forceinline__ double gauss_llike(const Dev& p, const double (&x)[DP]) {
  double q = 0;
  cdp row = as_c(p.P2);
  cdp mean = as_c(p.mean);
#pragma unroll
  for (int i = 0; i < DP; ++i) {
    double s = 0;
#pragma unroll
    for (int j = 0; j < i; ++j) {
      const double yj = MEAN ? x[j] - mean[j] : x[j];
      s = __builtin_fma(row[j], yj, s);
    }
    const double yi = MEAN ? x[i] - mean[i] : x[i];
    s = __builtin_fma(row[i], yi, s);
    q = __builtin_fma(yi, s, q);
    row += i + 1;
    PTM_SCHED_FENCE();
  }
  return p.like0 - 0.5 * q;
}

// ------------------------------------------------------------------------------------------------
// THE hot kernel: one fused MH_chain::step (chain.cc:966-1022) per chain, all rungs x walkers per launch:
//   gaussian_prop::draw (proposal_distribution.hh:194-218) -> state::add / enforce (states.cc:205-214,161-166)
//   -> prior -> Gaussian likelihood -> Metropolis test -> add_state counters (chain.cc:916-949),
// fused with the state hand-off of the exchange phase (rows named by src[]).
//   UNI    the wave's 64 chains share one rung (W % 64 == 0): factor / beta addresses are wave-uniform (SGPR operands)
//   SIMPLE open boundaries, all-uniform prior, zero mean, no one-dimensional moves (the BASELINE workload):
//          the general state-space / prior code is not even compiled in.
// ------------------------------------------------------------------------------------------------
template <int DP, int KIND, bool UNI, bool SIMPLE>
__global__ __launch_bounds__(256) void sweep_kernel(const Dev p) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= p.Nc) return;
  int rl = c / p.W;
  const int w = c - rl * p.W;
  if (UNI) rl = __builtin_amdgcn_readfirstlane(rl);
  const int rg = p.r0 + rl;
  const int Nc = p.Nc;

  const int sc = p.src[c];
  const int tc = p.touch[c];

  // ---- gather the chain's current state (its own row, a neighbour rung's row after an exchange, or an arrival)
  double x[DP];
  double ll, lp;
  if (sc >= 0) {
#pragma unroll
    for (int d = 0; d < DP; ++d) x[d] = p.x_in[(size_t)d * Nc + sc];
    ll = p.ll_in[sc];
    lp = p.lp_in[sc];
  } else {
    const double* rb = (sc == SRC_ABOVE) ? p.recv_above : p.recv_below;
#pragma unroll
    for (int d = 0; d < DP; ++d) x[d] = rb[(size_t)d * p.W + w];
    ll = rb[(size_t)DP * p.W + w];
    lp = rb[(size_t)(DP + 1) * p.W + w];
  }
  unsigned int nhist = p.nhist[c];  // add_state calls since initialisation (chain.cc:947)

  if (tc) {
    // rung took part in 1 or 2 exchange attempts: no MH move this step, one add_state per attempt
    // (chain.cc:1487-1490,1531-1534,1554-1557)
    nhist += (unsigned int)tc;
    p.touch[c] = 0;
    p.src[c] = c;
  } else {
    const double beta = as_c(p.beta)[rg];
    const double bl = beta * ll;
    const double cur_lpost = lp + bl;
    const double oldlprior = cur_lpost - bl;  // chain.cc:973
    const uint32_t stream = (uint32_t)w * (uint32_t)p.Nt + (uint32_t)rg;
    const u32x4 o0 = draw_block(p.seed, TAG_MH, stream, p.step, 0);

    // -- gaussian_prop::draw: D normals, optional one-dimensional move, offset = factor * z
    int type = 0, axis = -1;
    if (!SIMPLE && p.any_oned) {
      const double f = as_c(p.onedfrac)[rl];
      if (f > 0 && u01(o0.v1) < f) { axis = (int)(p.D * u01(o0.v2)); type = 1; }
    }
    cdp fac = as_c(p.prop) + (size_t)rl * p.prop_stride;
    double acc[DP];
#pragma unroll
    for (int i = 0; i < DP; ++i) acc[i] = 0.0;
#pragma unroll
    for (int b = 0; b < DP / 4; ++b) {
      const u32x4 o = draw_block(p.seed, TAG_MH, stream, p.step, (uint32_t)(b + 1));
      double z[4];
      boxmuller(o.v0, o.v1, z[0], z[1]);
      boxmuller(o.v2, o.v3, z[2], z[3]);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int j = 4 * b + t;
        double zj = z[t];
        if (!SIMPLE && axis >= 0 && j != axis) zj = 0.0;
        if (KIND == KIND_DIAG) {
          acc[j] = fac[j] * zj;
        } else if (KIND == KIND_LOWER) {
          // column j of the packed lower factor: rows j..DP-1 at offset j*DP - j(j-1)/2
          cdp col = fac + (j * DP - (j * (j - 1)) / 2);
#pragma unroll
          for (int i = j; i < DP; ++i) acc[i] = __builtin_fma(col[i - j], zj, acc[i]);
        } else {
          cdp col = fac + j * DP;
#pragma unroll
          for (int i = 0; i < DP; ++i) acc[i] = __builtin_fma(col[i], zj, acc[i]);
        }
        PTM_SCHED_FENCE();
      }
    }
    double xn[DP];
#pragma unroll
    for (int d = 0; d < DP; ++d) xn[d] = x[d] + acc[d];  // state::add (states.cc:205-214)

    bool valid;
    double newlprior;
    if (SIMPLE) {
      valid = true;
      bool in = true;
      cdp plo = as_c(p.plo), phi = as_c(p.phi);
#pragma unroll
      for (int d = 0; d < DP; ++d) in = in && !(xn[d] < plo[d]) && !(xn[d] > phi[d]);
      newlprior = in ? p.lprior_const : -__builtin_inf();
    } else {
      valid = p.origin_valid != 0;  // Q9: the sum is built on an enforced zero state
      newlprior = enforce_and_lprior<DP>(p, xn, valid);
    }
    double newlike, newlpost;
    if (valid && (newlprior > -1e200 || newlprior - oldlprior > p.min_prior)) {  // chain.cc:980 (Q1)
      newlike = (!SIMPLE && p.has_mean) ? gauss_llike<DP, true>(p, xn) : gauss_llike<DP, false>(p, xn);
      newlpost = newlike * beta + newlprior;
    } else {
      newlike = newlpost = -__builtin_inf();
    }
    const double logH = newlpost - cur_lpost;  // gaussian_prop: log_hastings_ratio() == 0
    bool accept = valid;
    if (accept && logH < 0) accept = dlog_u01(o0.v0) < logH;  // chain.cc:998-1001 (NaN stays accepted)

    p.ntries[c] += 1;
    if (accept) {
      p.naccept[c] += 1;
      p.last_type[c] = type;
#pragma unroll
      for (int d = 0; d < DP; ++d) x[d] = xn[d];
      ll = newlike;
      lp = newlprior;
    }
    nhist += 1u;
  }
#pragma unroll
  for (int d = 0; d < DP; ++d) p.x_out[(size_t)d * Nc + c] = x[d];
  p.ll_out[c] = ll;
  p.lp_out[c] = lp;
  p.nhist[c] = nhist;
}

// ------------------------------------------------------------------------------------------------
// evaluation of given states (set_states / debug_evaluate): enforce, lprior, llike
// ------------------------------------------------------------------------------------------------
template <int DP>
__global__ __launch_bounds__(256) void evaluate_kernel(const Dev p, int n, double* x_io /*[DP][n]*/, int* valid_out,
                                                        double* lprior_out, double* llike_out, int eval_like) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= n) return;
  double x[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) x[d] = x_io[(size_t)d * n + c];
  bool valid = true;  // state(space, values) constructor: valid unless enforce fails (states.cc:194-199)
  const double lp = enforce_and_lprior<DP>(p, x, valid);
#pragma unroll
  for (int d = 0; d < DP; ++d) x_io[(size_t)d * n + c] = x[d];
  if (valid_out) valid_out[c] = valid ? 1 : 0;
  lprior_out[c] = lp;
  if (eval_like) llike_out[c] = p.has_mean ? gauss_llike<DP, true>(p, x) : gauss_llike<DP, false>(p, x);
}

// MH_chain::initialize(1) (chain.cc:846-876): redraw from the prior until valid and llike >= -1e100.
// Dimension d of attempt a uses block d of the chain's INIT stream with step = a.
template <int DP>
__global__ __launch_bounds__(256) void init_prior_kernel(const Dev p, double* x_out, double* ll_out, double* lp_out,
                                                          int* fail) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= p.Nc) return;
  const int rl = c / p.W, w = c - rl * p.W;
  const uint32_t stream = (uint32_t)w * (uint32_t)p.Nt + (uint32_t)(p.r0 + rl);
  cip pt = as_c(p.ptype);
  cdp plo = as_c(p.plo), phi = as_c(p.phi);
  double x[DP];
  double ll = 0, lp = 0;
  bool done = false;
  for (uint64_t a = 0; a < 100000 && !done; ++a) {
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      x[d] = 0.0;
      if (d < p.D) {
        const u32x4 o = draw_block(p.seed, TAG_INIT, stream, a, (uint32_t)d);
        if (pt[d] == P_UNIFORM) x[d] = u01(o.v0) * (phi[d] - plo[d]) + plo[d];
        else if (pt[d] == P_GAUSSIAN) { double z0, z1; boxmuller(o.v0, o.v1, z0, z1); x[d] = z0 * phi[d] + plo[d]; }
        else x[d] = __builtin_nan("");
      }
    }
    bool valid = true;
    lp = enforce_and_lprior<DP>(p, x, valid);
    if (!valid) continue;
    ll = p.has_mean ? gauss_llike<DP, true>(p, x) : gauss_llike<DP, false>(p, x);
    if (ll < -1e100) continue;
    done = true;
  }
  if (!done) atomicOr(fail, 1);
#pragma unroll
  for (int d = 0; d < DP; ++d) x_out[(size_t)d * p.Nc + c] = x[d];
  ll_out[c] = ll;
  lp_out[c] = lp;
}

}  // namespace ptm
