// ptm_kernels.hpp -- gfx950 kernels of the parallel-tempering step engine.
//
// Data layout in HBM (per engine = per GPU shard), Nc = rung_count * W chains, chain c = rung_local * W + walker:
//   x      [2][D][Nc]  double  SoA state planes, ping-pong (a step reads buffer `cur`, writes the other)
//   llike  [2][Nc], lprior [2][Nc]  double         (lpost is always fl(lprior + fl(beta*llike)), chain.cc:928)
//   ntries, naccept, last_type [Nc] int32; nhist, nsize [Nc] int64   (MH_chain counters, chain.hh:150-170)
//   src [Nc] int32, touch [Nc] uint8   exchange phase -> sweep hand-off: where the chain's state comes from
//                                      this step and how many add_state calls it already received
// Walker is the fastest index, so the 64 lanes of a wave hold 64 walkers of ONE rung whenever W % 64 == 0:
// beta, the proposal factor and the Philox key material are then wave-uniform and travel through the scalar
// unit / scalar cache, while every state plane is read and written with fully coalesced 512-B wave accesses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ptm_device_math.hpp"

namespace ptm {

enum { KIND_DENSE = 0, KIND_DIAG = 1, KIND_LOWER = 2 };
enum { SRC_ABOVE = -2, SRC_BELOW = -3 };
enum { B_OPEN = 0, B_LIMIT = 1, B_REFLECT = 2, B_WRAP = 3 };
enum { P_FLAT = 0, P_UNIFORM = 1, P_GAUSSIAN = 2, P_POLAR = 3, P_COPOLAR = 4, P_LOG = 5 };

// problem description + state pointers; passed BY VALUE to every kernel (kernarg segment => scalar loads)
struct Dev {
  int D, Nt, r0, nloc, W, Nc;
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
  // Gaussian target: packed rows, row i = {2P_i0 .. 2P_i,i-1, P_ii}
  const double* P2;
  const double* mean;
  int has_mean;
  double like0;
  // ladder + proposals
  const double* beta;      // [Nt] global
  const double* prop;      // [nloc][prop_stride]  column-packed factor (see pack_factor) or sigmas
  const double* onedfrac;  // [nloc]
  int prop_stride, any_oned;
  // state
  const double* x_in;  double* x_out;
  const double* ll_in; double* ll_out;
  const double* lp_in; double* lp_out;
  int *ntries, *naccept, *last_type;
  long long *nhist, *nsize;
  int* src;
  unsigned char* touch;
  const double *recv_below, *recv_above;  // [(D+2)][W] rows that crossed the shard boundary this step
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

// stateSpace::enforce (states.cc:86-102) + sampleable_probability_function::evaluate_log (probability_function.hh:59)
template <int DP>
__device__ __forceinline__ double enforce_and_lprior(const Dev& p, double (&x)[DP], bool& valid) {
  if (valid && p.has_bounds) {
#pragma unroll
    for (int d = 0; d < DP; ++d)
      if (d < p.D && valid) valid = boundary_enforce(p.blo[d], p.bhi[d], p.bmin[d], p.bmax[d], x[d]);
  }
  if (!valid) return -__builtin_inf();
  if (p.all_uniform) {
    bool in = true;
#pragma unroll
    for (int d = 0; d < DP; ++d)
      if (d < p.D) in = in && !(x[d] < p.plo[d]) && !(x[d] > p.phi[d]);
    return in ? p.lprior_const : -__builtin_inf();
  }
  double result = 1;
#pragma unroll
  for (int d = 0; d < DP; ++d)
    if (d < p.D) result *= prior_pdf(p.ptype[d], p.plo[d], p.phi[d], p.pcoef[d], x[d]);
  return dlog(result);
}

// like0 - 1/2 y^T P y in the symmetric-packed order q = sum_i y_i (P_ii y_i + sum_{j<i} 2P_ij y_j)
template <int DP>
__device__ __forceinline__ double gauss_llike(const Dev& p, const double (&x)[DP]) {
  double q = 0;
  const double* __restrict__ row = p.P2;
#pragma unroll
  for (int i = 0; i < DP; ++i) {
    if (i < p.D) {
      double s = 0;
#pragma unroll
      for (int j = 0; j < i; ++j) {
        const double yj = p.has_mean ? x[j] - p.mean[j] : x[j];
        s = __builtin_fma(row[j], yj, s);
      }
      const double yi = p.has_mean ? x[i] - p.mean[i] : x[i];
      s = __builtin_fma(row[i], yi, s);
      q = __builtin_fma(yi, s, q);
    }
    row += i + 1;
  }
  return p.like0 - 0.5 * q;
}

// MH_chain::add_state bookkeeping (chain.cc:935-947)
__device__ __forceinline__ void add_state_count(long long& nhist, long long& nsize, int every) {
  if (nhist % every == 0) nsize++;
  nhist++;
}

// ------------------------------------------------------------------------------------------------
// THE hot kernel: one fused MH_chain::step (chain.cc:966-1022) per chain, all rungs x walkers per launch:
//   gaussian_prop::draw (proposal_distribution.hh:194-218) -> state::add / enforce (states.cc:205-214,161-166)
//   -> prior -> Gaussian likelihood -> Metropolis test -> add_state counters (chain.cc:916-949),
// fused with the state hand-off of the exchange phase (rows named by src[]).
// UNI: the wave's 64 chains share one rung (W % 64 == 0) => factor/beta addresses are wave-uniform.
// ------------------------------------------------------------------------------------------------
template <int DP, int KIND, bool UNI>
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
    for (int d = 0; d < DP; ++d) x[d] = (d < p.D) ? p.x_in[(size_t)d * Nc + sc] : 0.0;
    ll = p.ll_in[sc];
    lp = p.lp_in[sc];
  } else {
    const double* rb = (sc == SRC_ABOVE) ? p.recv_above : p.recv_below;
#pragma unroll
    for (int d = 0; d < DP; ++d) x[d] = (d < p.D) ? rb[(size_t)d * p.W + w] : 0.0;
    ll = rb[(size_t)p.D * p.W + w];
    lp = rb[(size_t)(p.D + 1) * p.W + w];
  }
  long long nhist = p.nhist[c], nsize = p.nsize[c];

  if (tc) {
    // rung took part in 1 or 2 exchange attempts: no MH move this step, one add_state per attempt
    // (chain.cc:1487-1490,1531-1534,1554-1557)
    for (int k = 0; k < tc; ++k) add_state_count(nhist, nsize, p.add_every_n);
    p.touch[c] = 0;
    p.src[c] = c;
  } else {
    const double beta = p.beta[rg];
    const double bl = beta * ll;
    const double cur_lpost = lp + bl;
    const double oldlprior = cur_lpost - bl;  // chain.cc:973
    const uint32_t stream = (uint32_t)w * (uint32_t)p.Nt + (uint32_t)rg;
    const u32x4 o0 = draw_block(p.seed, TAG_MH, stream, p.step, 0);

    // -- gaussian_prop::draw: D normals, optional one-dimensional move, offset = factor * z
    int type = 0, axis = -1;
    if (p.any_oned) {
      const double f = p.onedfrac[rl];
      if (f > 0 && u01(o0.v1) < f) { axis = (int)(p.D * u01(o0.v2)); type = 1; }
    }
    const double* __restrict__ fac = p.prop + (size_t)rl * p.prop_stride;
    double acc[DP];
#pragma unroll
    for (int i = 0; i < DP; ++i) acc[i] = 0.0;
#pragma unroll
    for (int b = 0; b < DP / 4; ++b) {
      if (4 * b < p.D) {
        const u32x4 o = draw_block(p.seed, TAG_MH, stream, p.step, (uint32_t)(b + 1));
        double z[4];
        boxmuller(o.v0, o.v1, z[0], z[1]);
        boxmuller(o.v2, o.v3, z[2], z[3]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int j = 4 * b + t;
          double zj = z[t];
          if (axis >= 0 && j != axis) zj = 0.0;
          if (KIND == KIND_DIAG) {
            acc[j] = (j < p.D) ? fac[j] * zj : 0.0;
          } else if (KIND == KIND_LOWER) {
            // column j of the packed lower factor: rows j..DP-1 at offset j*DP - j(j-1)/2
            const double* __restrict__ col = fac + (j * DP - (j * (j - 1)) / 2);
#pragma unroll
            for (int i = j; i < DP; ++i) acc[i] = __builtin_fma(col[i - j], zj, acc[i]);
          } else {
            const double* __restrict__ col = fac + j * DP;
#pragma unroll
            for (int i = 0; i < DP; ++i) acc[i] = __builtin_fma(col[i], zj, acc[i]);
          }
        }
      }
    }
    double xn[DP];
#pragma unroll
    for (int d = 0; d < DP; ++d) xn[d] = x[d] + acc[d];  // state::add (states.cc:205-214)

    bool valid = p.origin_valid != 0;  // Q9: the sum is built on an enforced zero state
    const double newlprior = enforce_and_lprior<DP>(p, xn, valid);
    double newlike, newlpost;
    if (valid && (newlprior > -1e200 || newlprior - oldlprior > p.min_prior)) {  // chain.cc:980 (Q1)
      newlike = gauss_llike<DP>(p, xn);
      newlpost = newlike * beta + newlprior;
    } else {
      newlike = newlpost = -__builtin_inf();
    }
    const double logH = newlpost - cur_lpost;  // gaussian_prop: log_hastings_ratio() == 0
    bool accept = valid;
    if (accept && logH < 0) accept = dlog(u01(o0.v0)) < logH;  // chain.cc:998-1001 (NaN stays accepted)

    int ntries = p.ntries[c] + 1;
    p.ntries[c] = ntries;
    if (accept) {
      p.naccept[c] += 1;
      p.last_type[c] = type;
#pragma unroll
      for (int d = 0; d < DP; ++d) x[d] = xn[d];
      ll = newlike;
      lp = newlprior;
    }
    add_state_count(nhist, nsize, p.add_every_n);
  }
#pragma unroll
  for (int d = 0; d < DP; ++d)
    if (d < p.D) p.x_out[(size_t)d * Nc + c] = x[d];
  p.ll_out[c] = ll;
  p.lp_out[c] = lp;
  p.nhist[c] = nhist;
  p.nsize[c] = nsize;
}

// ------------------------------------------------------------------------------------------------
// exchange phase of parallel_tempering_chains::step (chain.cc:1410-1537), one wave per walker-ladder.
// Candidate draws are parallel over lanes; the in-order filter and trials (quirk Q6: later picks see the
// in-place updated view) run on lane 0 over LDS copies of the picked rungs' llikes.
// The kernel moves no state: it names, for every local rung that took part, the row its state comes from
// (src[]) and the number of add_state calls it received (touch[]); the sweep kernel does the move.
// ------------------------------------------------------------------------------------------------
struct Decide {
  int D, Nt, r0, nloc, W, Nc, ms;
  uint64_t seed, step;
  double thresh;              // (Ntemps-1)*swap_rate/maxswapsperstep (chain.cc:1413)
  const double* beta;         // [Nt]
  const double* llg;          // GLOBAL llike [Nt][W]
  const double* x_in;         // local state planes (for packing departures)
  const double* ll_in;
  const double* lp_in;
  int* src;
  unsigned char* touch;
  long long *swap_try, *swap_acc;  // [W][Nt-1]
  int *last_pairs, *last_acc;      // [W][ms]
  double *send_up, *send_down;     // [(D+2)][W] or null
  int* err;
};

__global__ __launch_bounds__(64) void decide_kernel(const Decide p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int w = blockIdx.x;
  const int lane = threadIdx.x;
  const int Nt = p.Nt, ms = p.ms;
  // LDS carve (all offsets multiples of 8)
  double* llc = reinterpret_cast<double*>(smem);                              // [Nt]
  unsigned int* ukey = reinterpret_cast<unsigned int*>(llc + Nt);             // [ms]
  int* cand = reinterpret_cast<int*>(ukey + ((ms + 1) & ~1));                 // [ms]
  int* accf = cand + ((ms + 1) & ~1);                                         // [ms]
  unsigned short* perm = reinterpret_cast<unsigned short*>(accf + ((ms + 1) & ~1));  // [Nt]
  unsigned char* tch = reinterpret_cast<unsigned char*>(perm + ((Nt + 3) & ~3));     // [Nt]
  unsigned char* mark = tch + ((Nt + 7) & ~7);                                       // [Nt+1]
  int* down_src_p = reinterpret_cast<int*>(mark + ((Nt + 1 + 7) & ~7));              // [1]
  // (no static __shared__: it would precede the dynamic region and break its 16-byte base alignment)

  for (int i = lane; i < Nt + 1; i += 64) mark[i] = 0;
  if (lane == 0) *down_src_p = -1;
  // -- candidate draws (chain.cc:1410-1416): block k of the ladder stream gives {u_try, u_pick, u_accept}
  for (int k = lane; k < ms; k += 64) {
    const u32x4 o = draw_block(p.seed, TAG_PT, (uint32_t)w, p.step, (uint32_t)k);
    int n = -2;
    if (Nt > 1 && u01(o.v0) < p.thresh) n = (int)(u01(o.v1) * (Nt - 1));
    cand[k] = n;
    ukey[k] = o.v2;
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
    llc[n] = p.llg[(size_t)n * p.W + w];
    llc[n + 1] = p.llg[(size_t)(n + 1) * p.W + w];
    perm[n] = (unsigned short)n;
    perm[n + 1] = (unsigned short)(n + 1);
    tch[n] = 0;
    tch[n + 1] = 0;
  }
  __syncthreads();
  // -- trials in pick order (chain.cc:1436-1537)
  if (lane == 0) {
    long long* st = p.swap_try + (size_t)w * (Nt - 1);
    long long* sa = p.swap_acc + (size_t)w * (Nt - 1);
    for (int k = 0; k < ms; ++k) {
      const int i = cand[k];
      if (i < 0) continue;
      double lla = llc[i];
      if (!(lla > -1e200)) lla = -1e200;
      double llb = llc[i + 1];
      if (!(llb > -1e200)) llb = -1e200;
      const double logH = -(p.beta[i + 1] - p.beta[i]) * (llb - lla);
      bool acc = true;
      if (logH < 0) acc = dlog(u01(ukey[k])) < logH;
      if (acc) {
        if (i + 1 == p.r0) *down_src_p = perm[i + 1];  // the row that leaves this shard downwards
        const double t = llc[i]; llc[i] = llc[i + 1]; llc[i + 1] = t;
        const unsigned short s = perm[i]; perm[i] = perm[i + 1]; perm[i + 1] = s;
        sa[i] += 1;
        accf[k] = 1;
      }
      tch[i] += 1;
      tch[i + 1] += 1;
      st[i] += 1;
    }
  }
  __syncthreads();
  // -- publish: hand-off arrays for local rungs, the step's log, departures
  const int r1 = p.r0 + p.nloc;
  for (int k = lane; k < ms; k += 64) {
    const int i = cand[k];
    p.last_pairs[(size_t)w * ms + k] = i;
    p.last_acc[(size_t)w * ms + k] = accf[k];
    if (i < 0) continue;
    for (int r = i; r <= i + 1; ++r) {
      if (r < p.r0 || r >= r1) continue;
      const int c = (r - p.r0) * p.W + w;
      const int s = perm[r];
      p.touch[c] = tch[r];
      p.src[c] = (s >= p.r0 && s < r1) ? (s - p.r0) * p.W + w : (s >= r1 ? SRC_ABOVE : SRC_BELOW);
    }
    if (accf[k] && i + 1 == r1 && r1 < Nt && p.send_up) {
      // exchange across the upper shard boundary: our top rung's row (always its start-of-step content) goes up
      const int cs = (i - p.r0) * p.W + w;
      for (int d = 0; d < p.D; ++d) p.send_up[(size_t)d * p.W + w] = p.x_in[(size_t)d * p.Nc + cs];
      p.send_up[(size_t)p.D * p.W + w] = p.ll_in[cs];
      p.send_up[(size_t)(p.D + 1) * p.W + w] = p.lp_in[cs];
    }
    if (accf[k] && i + 1 == p.r0 && p.send_down) {
      const int s = *down_src_p;
      if (s < p.r0 || s >= r1) {
        atomicOr(p.err, 1);  // the departing row is not ours: it crossed two boundaries in one step
      } else {
        const int cs = (s - p.r0) * p.W + w;
        for (int d = 0; d < p.D; ++d) p.send_down[(size_t)d * p.W + w] = p.x_in[(size_t)d * p.Nc + cs];
        p.send_down[(size_t)p.D * p.W + w] = p.ll_in[cs];
        p.send_down[(size_t)(p.D + 1) * p.W + w] = p.lp_in[cs];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// evaluation of given states (set_states / debug_evaluate): enforce, lprior, llike
// ------------------------------------------------------------------------------------------------
template <int DP>
__global__ __launch_bounds__(256) void evaluate_kernel(const Dev p, int n, double* x_io /*[D][n]*/, int* valid_out,
                                                        double* lprior_out, double* llike_out, int eval_like) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= n) return;
  double x[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) x[d] = (d < p.D) ? x_io[(size_t)d * n + c] : 0.0;
  bool valid = true;  // state(space, values) constructor: valid unless enforce fails (states.cc:194-199)
  const double lp = enforce_and_lprior<DP>(p, x, valid);
#pragma unroll
  for (int d = 0; d < DP; ++d)
    if (d < p.D) x_io[(size_t)d * n + c] = x[d];
  if (valid_out) valid_out[c] = valid ? 1 : 0;
  lprior_out[c] = lp;
  if (eval_like) llike_out[c] = gauss_llike<DP>(p, x);
}

// MH_chain::initialize(1) (chain.cc:846-876): redraw from the prior until valid and llike >= -1e100.
// Dimension d of attempt a uses block d of the chain's INIT stream with step = a.
template <int DP>
__global__ __launch_bounds__(256) void init_prior_kernel(const Dev p, double* x_out, double* ll_out, double* lp_out,
                                                          int* fail) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= p.Nc) return;
  const int rl = c / p.W, w = c - rl * p.W;
  // INIT stream id = w*Nt + global rung, as for the MH stream
  const uint32_t stream = (uint32_t)w * (uint32_t)p.Nt + (uint32_t)(p.r0 + rl);
  double x[DP];
  double ll = 0, lp = 0;
  bool done = false;
  for (uint64_t a = 0; a < 100000 && !done; ++a) {
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      x[d] = 0.0;
      if (d < p.D) {
        const u32x4 o = draw_block(p.seed, TAG_INIT, stream, a, (uint32_t)d);
        if (p.ptype[d] == P_UNIFORM) x[d] = u01(o.v0) * (p.phi[d] - p.plo[d]) + p.plo[d];
        else if (p.ptype[d] == P_GAUSSIAN) { double z0, z1; boxmuller(o.v0, o.v1, z0, z1); x[d] = z0 * p.phi[d] + p.plo[d]; }
        else x[d] = __builtin_nan("");
      }
    }
    bool valid = true;
    lp = enforce_and_lprior<DP>(p, x, valid);
    if (!valid) continue;
    ll = gauss_llike<DP>(p, x);
    if (ll < -1e100) continue;
    done = true;
  }
  if (!done) atomicOr(fail, 1);
#pragma unroll
  for (int d = 0; d < DP; ++d)
    if (d < p.D) x_out[(size_t)d * p.Nc + c] = x[d];
  ll_out[c] = ll;
  lp_out[c] = lp;
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

}  // namespace ptm
