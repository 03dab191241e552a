/* oracle/ptm_oracle.c -- TEST INFRASTRUCTURE ONLY (see ptm_oracle.h for the contract).
 *
 * Plain-C restatement of ptmcmc's chain::step() hot path.  Scalar, one chain at a time,
 * written for legibility; the only concession to speed is an optional OpenMP loop over
 * chains in the MH sweep (used by bench.py's cpu_baseline leg, kind "port").
 *
 * Arithmetic contract shared with the HIP kernels (DESIGN.md "Numerics"): every floating
 * point result below is produced by IEEE-754 binary64 +, -, *, /, sqrt and explicitly
 * written fma() in the order written here; the file is compiled with -ffp-contract=off.
 * log / exp / sin / cos are the deterministic implementations in this file (classic
 * fdlibm-style argument reduction + polynomial, see each function), never libm, except
 * where a value is a per-problem CONSTANT that the reference also computes with libm on
 * the host (prior normalisations, the ladder).
 */
#define _GNU_SOURCE
#include "ptm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ============================================================================================
 * Philox4x32-10  (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3")
 * ============================================================================================ */
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

void ptmo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
    uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += PHILOX_W0; k1 += PHILOX_W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* newran1.cxx:432  MotherOfAll::Next(): ((double)seed + 0.5) / 4294967296.0 */
double ptmo_u01(uint32_t k) { return ((double)k + 0.5) * (1.0 / 4294967296.0); }

void ptmo_draw_block(uint64_t seed, int tag, uint32_t stream, uint64_t step, uint32_t block, uint32_t out[4]) {
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t ctr[4] = {block, stream, (uint32_t)step, ((uint32_t)(step >> 32) & 0x00FFFFFFu) | ((uint32_t)tag << 24)};
  ptmo_philox4x32_10(ctr, key, out);
}

/* ============================================================================================
 * deterministic elementary functions
 * ============================================================================================ */
static inline uint64_t d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

/* log(x): x = 2^e * m, m in (sqrt(1/2), sqrt(2)];  f = m-1, s = f/(2+f);
 * log(1+f) = 2s + (2/3)s^3 + ... via the degree-14 even polynomial of the classic
 * FreeBSD/fdlibm e_log.c algorithm (coefficients Lg1..Lg7 published there).  < 1 ulp. */
double ptmo_log(double x) {
  static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  static const double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01, L3 = 2.857142874366239149e-01,
                      L4 = 2.222219843214978396e-01, L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
                      L7 = 1.479819860511658591e-01;
  int e = 0;
  if (x != x) return x;
  if (x < 0.0) return NAN;
  if (x == 0.0) return -INFINITY;
  if (x == INFINITY) return x;
  if (x < 2.2250738585072014e-308) { x *= 18014398509481984.0; e = -54; } /* subnormal: scale by 2^54 */
  uint64_t b = d2u(x);
  e += (int)(b >> 52) - 1023;
  double m = u2d((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull); /* [1,2) */
  if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
  double f = m - 1.0;
  double s = f / (2.0 + f);
  double z = s * s;
  double r = L7;
  r = fma(r, z, L6); r = fma(r, z, L5); r = fma(r, z, L4); r = fma(r, z, L3); r = fma(r, z, L2); r = fma(r, z, L1);
  double R = r * z;
  double hfsq = 0.5 * f * f;
  double dk = (double)e;
  return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

/* exp(x): k = round(x/ln2), r = x - k ln2 (hi/lo split), exp(r) = 1 + 2r/(2-c)... with the degree-10
 * even polynomial of the classic fdlibm e_exp.c algorithm (coefficients P1..P5 published there);
 * result scaled by 2^k with gradual underflow.  < 1 ulp in the normal range. */
double ptmo_exp(double x) {
  static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                      invln2 = 1.44269504088896338700e+00;
  static const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                      P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  if (x != x) return x;
  if (x > 709.782712893384) return INFINITY;
  if (x < -745.1332191019412) return 0.0;
  int k = (int)(x * invln2 + (x < 0.0 ? -0.5 : 0.5));
  double dk = (double)k;
  double hi = x - dk * ln2_hi;
  double lo = dk * ln2_lo;
  double r = hi - lo;
  double t = r * r;
  double p = P5;
  p = fma(p, t, P4); p = fma(p, t, P3); p = fma(p, t, P2); p = fma(p, t, P1);
  double c = r - t * p;
  double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
  if (k >= -1021 && k <= 1023) return y * u2d((uint64_t)(k + 1023) << 52);
  if (k > 1023) return y * 2.0 * u2d((uint64_t)(k - 1 + 1023) << 52);
  return (y * u2d((uint64_t)(k + 1000 + 1023) << 52)) * u2d((uint64_t)(-1000 + 1023) << 52);
}

/* sin / cos on [0, pi/4]: Taylor series in Horner/fma form (truncation < 1e-17) */
static inline double sin_k(double p) {
  double z = p * p;
  double r = -1.0 / 1307674368000.0;
  r = fma(r, z, 1.0 / 6227020800.0);
  r = fma(r, z, -1.0 / 39916800.0);
  r = fma(r, z, 1.0 / 362880.0);
  r = fma(r, z, -1.0 / 5040.0);
  r = fma(r, z, 1.0 / 120.0);
  r = fma(r, z, -1.0 / 6.0);
  return fma(p * z, r, p);
}
static inline double cos_k(double p) {
  double z = p * p;
  double r = 1.0 / 20922789888000.0;
  r = fma(r, z, -1.0 / 87178291200.0);
  r = fma(r, z, 1.0 / 479001600.0);
  r = fma(r, z, -1.0 / 3628800.0);
  r = fma(r, z, 1.0 / 40320.0);
  r = fma(r, z, -1.0 / 720.0);
  r = fma(r, z, 1.0 / 24.0);
  return fma(z * z, r, fma(-0.5, z, 1.0));
}
#define PI_HI 3.141592653589793116e+00 /* 0x1.921fb54442d18p+1 */
#define PI_LO 1.224646799147353207e-16 /* pi - PI_HI */
#define HPI_HI 1.570796326794896558e+00
#define HPI_LO 6.123233995736766036e-17
#define QPI 7.853981633974482790e-01

/* sin on [0,pi] (UniformPolarDist::pdf argument range, ProbabilityDist.h:197-201) */
double ptmo_sin_0_pi(double x) {
  if (x > HPI_HI) x = (PI_HI - x) + PI_LO;   /* sin(pi - x) */
  if (x <= QPI) return sin_k(x);
  return cos_k((HPI_HI - x) + HPI_LO);
}
/* cos on [-pi/2,pi/2] (UniformCoPolarDist::pdf, ProbabilityDist.h:243-247) */
double ptmo_cos_hpi(double x) {
  x = fabs(x);
  if (x <= QPI) return cos_k(x);
  return sin_k((HPI_HI - x) + HPI_LO);
}

/* -2 ln(u) for u = (k+0.5)/2^32, the Box-Muller radius argument: table of {fl(1/c_i), A_i} over the top 8 mantissa
 * bits (oracle/ptm_tables.inc, generated DATA, re-derived by tests/test_tables.py) + degree-6 log1p polynomial:
 *   x = k+0.5 = 2^E m;  t = fma(m, 1/c_i, -1);  q = 1 - (2/3)t + (1/2)t^2 - (2/5)t^3 + (1/3)t^4;
 *   -2 ln u = [(E-32+adj)(-2 ln2_hi) + A_i] + [(E-32+adj)(-2 ln2_lo) + (t^2 q - 2t)],  adj = [i >= SPLIT]. */
#include "ptm_tables.inc"
static const double bm_table[512] = {PTM_BMTAB_VALUES};
double ptmo_bm_neg2log(uint32_t k) {
  const double M2LN2_HI = -2.0 * 6.93147180369123816490e-01, M2LN2_LO = -2.0 * 1.90821492927058770002e-10;
  double x = (double)k + 0.5;
  uint64_t b;
  memcpy(&b, &x, 8);
  uint32_t hi = (uint32_t)(b >> 32);
  uint32_t idx = (hi >> 12) & 255u;
  int e = (int)(hi >> 20) - (1023 + 32) + (idx >= PTM_BMTAB_SPLIT ? 1 : 0);
  uint64_t mb = (b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
  double m;
  memcpy(&m, &mb, 8);
  double rc = bm_table[2 * idx], A = bm_table[2 * idx + 1];
  double t = fma(m, rc, -1.0);
  double q = 1.0 / 3.0;
  q = fma(q, t, -0.4);
  q = fma(q, t, 0.5);
  q = fma(q, t, -2.0 / 3.0);
  q = fma(q, t, 1.0);
  double l = fma(t * t, q, -2.0 * t);
  double dk = (double)e;
  return fma(dk, M2LN2_HI, A) + fma(dk, M2LN2_LO, l);
}

/* Box-Muller on two 32-bit draws.  r = sqrt(-2 ln u1) (IEEE sqrt).  theta = 2 pi (k2+.5)/2^32: bit 31 of k2 is the
 * half turn (sign of r), bits 30..21 select {sin a_i, cos a_i}, a_i = (i+.5) pi/1024, from the generated table, the low
 * 21 bits give delta = theta - a_i (|delta| <= pi/2048), added with
 *   sin(a+d) = S + (S (cos d - 1) + C sin d),  cos(a+d) = C + (C (cos d - 1) - S sin d),
 *   sin d = d + d^3 (-1/6 + d^2/120),  cos d - 1 = d^2 (-1/2 + d^2/24). */
static const double trig_table[2048] = {PTM_TRIGTAB_VALUES};
void ptmo_boxmuller(uint32_t k1, uint32_t k2, double* z0, double* z1) {
  double r = sqrt(ptmo_bm_neg2log(k1));
  uint32_t idx = (k2 >> 21) & 1023u;
  double S = trig_table[2 * idx], C = trig_table[2 * idx + 1];
  double d = ((double)((int)(k2 & 0x1FFFFFu) - (1 << 20)) + 0.5) * 1.4629180792671596e-09; /* 2 pi / 2^32 = 0x1.921fb54442d18p-30 */
  double d2 = d * d;
  double sd = fma(d * d2, fma(d2, 1.0 / 120.0, -1.0 / 6.0), d);
  double cm1 = d2 * fma(d2, 1.0 / 24.0, -0.5);
  double sn = fma(C, sd, S * cm1) + S;
  double cs = fma(-S, sd, C * cm1) + C;
  if (k2 >> 31) r = -r;
  *z0 = r * cs;
  *z1 = r * sn;
}

/* ============================================================================================
 * state algebra: boundary::enforce (states.cc:11-58), stateSpace::enforce (states.cc:86-102)
 * ============================================================================================ */
/* fmod restated with exact semantics for the magnitudes that occur (|q| < 2^52):
 * r = x - w*trunc(x/w) evaluated with one fma, then corrected into (-w, w) with x's sign. */
static double fmod_det(double x, double w) {
  double q = trunc(x / w);
  double r = fma(-q, w, x);
  if (x >= 0.0) { if (r < 0.0) r += w; else if (r >= w) r -= w; }
  else          { if (r > 0.0) r -= w; else if (r <= -w) r += w; }
  return r;
}

int ptmo_boundary_enforce(int lo, int hi, double xmin, double xmax, double* px) {
  double x = *px;
  if ((lo == PTMO_WRAP) != (hi == PTMO_WRAP)) return 0;         /* states.cc:14-16 inconsistent wrap */
  if (lo == PTMO_WRAP) {                                         /* states.cc:17-29 */
    double width = xmax - xmin;
    if (width <= 0) return 0;
    double xt = fmod_det(x - xmin, width);
    if (xt < 0) xt += width;
    *px = xmin + xt;
    return 1;
  }
  if (lo == PTMO_REFLECT && hi == PTMO_REFLECT) {                /* states.cc:31-45 */
    double halfwidth = xmax - xmin;
    if (halfwidth <= 0) return 0;
    double width = 2 * halfwidth;
    double xt = fmod_det(x - xmin, width);
    if (xt < 0) xt += width;
    if (xt >= halfwidth) xt = halfwidth - xt;                    /* sic: the reference folds to a NEGATIVE offset */
    *px = xmin + xt;
    return 1;
  }
  if (lo == PTMO_REFLECT && x < xmin) x = xmin + (xmin - x);     /* states.cc:46-47 */
  else if (hi == PTMO_REFLECT && x > xmax) x = xmax - (x - xmax);
  *px = x;
  if (lo == PTMO_LIMIT && x < xmin) return 0;                    /* states.cc:48-55 */
  if (hi == PTMO_LIMIT && x > xmax) return 0;
  return 1;
}

int ptmo_enforce(const ptmo_problem* pb, double* x) {
  for (int i = 0; i < pb->D; i++)
    if (!ptmo_boundary_enforce(pb->blo[i], pb->bhi[i], pb->bmin[i], pb->bmax[i], &x[i])) return 0; /* stops at first failure */
  return 1;
}

/* ============================================================================================
 * prior: sampleable_probability_function::evaluate_log = log(evaluate) (probability_function.hh:59)
 * with mixed_dist_product::evaluate = prod_i pdf_i(x_i)  (probability_function.cc:281-304) and the
 * pdfs of ProbabilityDist.h:88-93 (uniform), :126-130 (log), :153-156 (gaussian), :197-201 (polar),
 * :243-247 (copolar).  Invalid state => 0 => log(0) = -inf.
 * ============================================================================================ */
static double pdf1(const ptmo_problem* pb, int i, double x) {
  double lo = pb->plo[i], hi = pb->phi[i];
  switch (pb->ptype[i]) {
    case PTMO_FLAT: return 1;
    case PTMO_UNIFORM:
      if (x < lo) return 0;
      if (x > hi) return 0;
      return pb->pcoef[i];                               /* 1/(xmax-xmin) */
    case PTMO_GAUSSIAN: {
      double xn = (x - lo) / hi;                         /* lo = x0, hi = sigma */
      return ptmo_exp(-xn * xn / 2) / 2.5066282746310002 / hi;  /* exp(-xnorm*xnorm/2)/sqrt(2*M_PI)/sigma */
    }
    case PTMO_POLAR:
      if (x < lo) return 0;
      if (x > hi) return 0;
      return ptmo_sin_0_pi(x) / pb->pcoef[i];
    case PTMO_COPOLAR:
      if (x < lo) return 0;
      if (x > hi) return 0;
      return ptmo_cos_hpi(x) / pb->pcoef[i];
    case PTMO_LOG:
      if (x < lo) return 0;
      if (x > hi) return 0;
      return 1 / pb->pcoef[i] / x;                       /* 1/(log_xmax-log_xmin)/x */
  }
  return NAN;
}

double ptmo_lprior(const ptmo_problem* pb, const double* x, int valid) {
  if (!valid) return -INFINITY;                          /* evaluate() returns 0 for an invalid state */
  if (pb->prior_fn) return pb->prior_fn(pb->prior_user, x, pb->D);
  if (pb->all_uniform) {
    for (int i = 0; i < pb->D; i++) {
      if (x[i] < pb->plo[i]) return -INFINITY;
      if (x[i] > pb->phi[i]) return -INFINITY;
    }
    return pb->lprior_const;
  }
  /* mixed_dist_product::evaluate (probability_function.cc:281-304) multiplies the factors in dimension order; here (and
   * in the kernels, whose MFMA layout spreads a chain's dimensions over four lanes) they are multiplied in four
   * interleaved partial products p_q = prod_{i = q mod 4} pdf_i (i ascending), combined as ((p0 p1) p2) p3 */
  double pq[4] = {1, 1, 1, 1};
  for (int i = 0; i < pb->D; i++) pq[i & 3] *= pdf1(pb, i, x[i]);
  double result = ((pq[0] * pq[1]) * pq[2]) * pq[3];
  return ptmo_log(result);
}

/* Gaussian target of cython/exampleGaussian.py:103-109: like0 - 0.5 x^T invcov x, evaluated in the order shared with
 * the kernels:  s_i = P_ii y_i + sum_{j<i} 2 P_ij y_j  (one fma chain per row, j ascending -- on the GPU a column of
 * 16x16x4 f64 MFMA tiles, whose accumulation is exactly this chain), then the dot product y.s in four interleaved
 * partial sums p_q = sum_{i = q mod 4} y_i s_i (i ascending) combined as ((p0 + p1) + p2) + p3. */
double ptmo_llike(const ptmo_problem* pb, const double* x) {
  if (pb->user_fn) return pb->user_fn(pb->user, x, pb->D);
  int D = pb->D;
  double pq[4] = {0.0, 0.0, 0.0, 0.0};
  for (int i = 0; i < D; i++) {
    double s = 0;
    for (int j = 0; j < i; j++) {
      double yj = pb->mean ? x[j] - pb->mean[j] : x[j];
      s = fma(pb->P2[i * D + j], yj, s);
    }
    double yi = pb->mean ? x[i] - pb->mean[i] : x[i];
    s = fma(pb->P2[i * D + i], yi, s);
    pq[i & 3] = fma(yi, s, pq[i & 3]);
  }
  double q = ((pq[0] + pq[1]) + pq[2]) + pq[3];
  return pb->like0 - 0.5 * q;
}

/* order in which the columns of a proposal factor are accumulated (the kernels' tile order): natural for padded
 * dimension <= 8; otherwise in halves of 16 columns, inside a half s + 4k with s = 0..3 outer, k = 0..3 inner.
 * Padded dimension = D rounded up to 4, 8, 16, ..., 1024. */
int ptmo_column_order(int D, int* ord) {
  int DP = 4;
  while (DP < D) DP *= 2;
  int n = 0;
  if (DP <= 8) {
    for (int j = 0; j < D; j++) ord[n++] = j;
    return n;
  }
  for (int h = 0; 16 * h < DP; h++)
    for (int sct = 0; sct < 4; sct++)
      for (int k = 0; k < 4; k++) {
        int col = 16 * h + 4 * k + sct;
        if (col < D) ord[n++] = col;
      }
  return n;
}

/* chain.cc:928 / :982 / :1090: lprior + invtemp*llike, product rounded before the sum */
double ptmo_lpost(double lprior, double beta, double llike) {
  double t = beta * llike;
  return lprior + t;
}

/* chain.cc:1181-1183 (tratio=exp(log(Tmax)/(Ntemps-1)); temps[i]=temps[i-1]*tratio) and :1340 (invtemp=1/temps[i]).
 * Per-problem constants: libm here exactly as in the reference. */
void ptmo_ladder(int Nt, double Tmax, double* beta) {
  double tratio = exp(log(Tmax) / (Nt - 1));
  double t = 1;
  beta[0] = 1 / t;
  for (int i = 1; i < Nt; i++) { t = t * tratio; beta[i] = 1 / t; }
}

/* ============================================================================================
 * problem set-up
 * ============================================================================================ */
ptmo_problem* ptmo_problem_create(int D) {
  ptmo_problem* p = (ptmo_problem*)calloc(1, sizeof *p);
  p->D = D;
  p->blo = (int*)calloc(D, sizeof(int)); p->bhi = (int*)calloc(D, sizeof(int));
  p->bmin = (double*)calloc(D, sizeof(double)); p->bmax = (double*)calloc(D, sizeof(double));
  p->ptype = (int*)calloc(D, sizeof(int));
  p->plo = (double*)calloc(D, sizeof(double)); p->phi = (double*)calloc(D, sizeof(double));
  p->pcoef = (double*)calloc(D, sizeof(double));
  for (int i = 0; i < D; i++) { p->bmin[i] = -INFINITY; p->bmax[i] = INFINITY; p->plo[i] = -INFINITY; p->phi[i] = INFINITY; p->pcoef[i] = 1; }
  p->origin_valid = 1; p->all_uniform = 1; p->lprior_const = 0; p->minPrior = -30;
  p->P2 = (double*)calloc((size_t)D * D, sizeof(double));
  return p;
}
void ptmo_problem_free(ptmo_problem* p) {
  if (!p) return;
  free(p->blo); free(p->bhi); free(p->bmin); free(p->bmax); free(p->ptype); free(p->plo); free(p->phi); free(p->pcoef);
  free(p->mean); free(p->P2); free(p);
}
void ptmo_problem_set_bounds(ptmo_problem* p, const int* lo, const int* hi, const double* xmin, const double* xmax) {
  for (int i = 0; i < p->D; i++) { p->blo[i] = lo[i]; p->bhi[i] = hi[i]; p->bmin[i] = xmin[i]; p->bmax[i] = xmax[i]; }
  /* Q9: state(space,n) is a zero vector passed through enforce() (states.cc:183-192) */
  double* z = (double*)calloc(p->D, sizeof(double));
  p->origin_valid = ptmo_enforce(p, z);
  free(z);
}
void ptmo_problem_set_prior(ptmo_problem* p, const int* types, const double* c, const double* h) {
  /* mixed_dist_product ctor, probability_function.cc:232-254; Uniform{Polar,CoPolar}Dist ctors clamp the
   * PARAMETER copies of xmin/xmax for the normalisation but keep the unclamped members for the support
   * test (ProbabilityDist.h:181-186, 227-232: the ctor arguments shadow the members). */
  p->all_uniform = 1;
  double prod = 1;
  for (int i = 0; i < p->D; i++) {
    p->ptype[i] = types[i];
    switch (types[i]) {
      case PTMO_UNIFORM: p->plo[i] = c[i] - h[i]; p->phi[i] = c[i] + h[i]; p->pcoef[i] = 1 / (p->phi[i] - p->plo[i]); break;
      case PTMO_GAUSSIAN: p->plo[i] = c[i]; p->phi[i] = h[i]; p->pcoef[i] = 0; p->all_uniform = 0; break;
      case PTMO_POLAR: {
        double a = c[i] - h[i], b = c[i] + h[i];
        p->plo[i] = a; p->phi[i] = b;
        if (a < 0) a = 0;
        if (b > M_PI) b = M_PI;
        p->pcoef[i] = -cos(b) + cos(a); p->all_uniform = 0; break;
      }
      case PTMO_COPOLAR: {
        double a = c[i] - h[i], b = c[i] + h[i];
        p->plo[i] = a; p->phi[i] = b;
        if (a < -M_PI / 2) a = -M_PI / 2;
        if (b > M_PI / 2) b = M_PI / 2;
        p->pcoef[i] = sin(b) - sin(a); p->all_uniform = 0; break;
      }
      case PTMO_FLAT: p->pcoef[i] = 1; break;
      case PTMO_LOG: p->plo[i] = c[i] / h[i]; p->phi[i] = c[i] * h[i]; p->pcoef[i] = log(p->phi[i]) - log(p->plo[i]); p->all_uniform = 0; break;
    }
    if (types[i] == PTMO_UNIFORM || types[i] == PTMO_FLAT) prod *= p->pcoef[i];
  }
  p->lprior_const = log(prod);   /* the reference takes libm log of the running product every call */
}
void ptmo_problem_set_gauss(ptmo_problem* p, const double* mean, const double* P, double like0) {
  int D = p->D;
  p->have_gauss = 1; p->like0 = like0; p->user_fn = 0;
  free(p->mean); p->mean = 0;
  if (mean) { p->mean = (double*)malloc(D * sizeof(double)); memcpy(p->mean, mean, D * sizeof(double)); }
  for (int i = 0; i < D; i++)
    for (int j = 0; j < D; j++)
      p->P2[i * D + j] = (j < i) ? (P[i * D + j] + P[j * D + i]) : (j == i ? P[i * D + i] : 0.0);
}
void ptmo_problem_set_user(ptmo_problem* p, ptmo_loglike_fn fn, void* user) { p->user_fn = fn; p->user = user; }
void ptmo_problem_set_user_prior(ptmo_problem* p, ptmo_loglike_fn fn, void* user) { p->prior_fn = fn; p->prior_user = user; }

/* ============================================================================================
 * ladder state
 * ============================================================================================ */
ptmo_pt* ptmo_pt_create(int D, int Nt, int W, const double* beta, double swap_rate, int add_every_N) {
  ptmo_pt* s = (ptmo_pt*)calloc(1, sizeof *s);
  size_t N = (size_t)Nt * W;
  s->D = D; s->Nt = Nt; s->W = W; s->swap_rate = swap_rate; s->add_every_N = add_every_N;
  s->evolve_cut = -1;
  s->maxswaps = (int)(1 + 2 * swap_rate * Nt);                  /* chain.cc:1192 (int = double truncation) */
  s->beta = (double*)malloc(Nt * sizeof(double)); memcpy(s->beta, beta, Nt * sizeof(double));
  s->x = (double*)calloc(N * D, sizeof(double));
  s->llike = (double*)calloc(N, sizeof(double)); s->lprior = (double*)calloc(N, sizeof(double));
  s->ntries = (int32_t*)malloc(N * 4); s->naccept = (int32_t*)malloc(N * 4); s->last_type = (int32_t*)malloc(N * 4);
  s->nhist = (int64_t*)calloc(N, 8); s->nsize = (int64_t*)calloc(N, 8);
  for (size_t c = 0; c < N; c++) { s->ntries[c] = 1; s->naccept[c] = 1; s->last_type[c] = -1; } /* chain.cc:649 */
  size_t np = (size_t)W * (Nt > 1 ? Nt - 1 : 1);
  s->swap_count = (int64_t*)calloc(np, 8); s->swap_accept_count = (int64_t*)calloc(np, 8);
  s->last_pairs = (int*)calloc((size_t)W * s->maxswaps, sizeof(int));
  s->last_accept = (int*)calloc((size_t)W * s->maxswaps, sizeof(int));
  s->touched = (uint8_t*)calloc(N, 1);
  s->last_accept_mh = (uint8_t*)calloc(N, 1);
  s->map_lpost = (double*)malloc(N * sizeof(double));
  s->map_x = (double*)calloc(N * D, sizeof(double));
  for (size_t c = 0; c < N; c++) s->map_lpost[c] = -1e200;        /* chain.hh:69 */
  return s;
}
void ptmo_pt_free(ptmo_pt* s) {
  if (!s) return;
  free(s->beta); free(s->x); free(s->llike); free(s->lprior); free(s->ntries); free(s->naccept); free(s->last_type);
  free(s->nhist); free(s->nsize); free(s->swap_count); free(s->swap_accept_count); free(s->last_pairs); free(s->last_accept);
  free(s->touched); free(s->last_accept_mh); free(s->map_lpost); free(s->map_x); free(s->betaw);
  free(s->de_init);
  free(s->hist_x); free(s->hist_ll); free(s->hist_lp); free(s->hist_beta); free(s->hist_nacc); free(s->hist_ntry); free(s->hist_type);
  free(s);
}
void ptmo_pt_evolve_lpost_cut(ptmo_pt* s, double cut) { s->evolve_cut = cut; }
void ptmo_pt_evolve_temps(ptmo_pt* s, double rate) {
  s->evolve_rate = rate;
  free(s->betaw);
  s->betaw = NULL;
  if (!(rate > 0)) return;
  s->betaw = (double*)malloc((size_t)s->W * s->Nt * sizeof(double));
  for (int w = 0; w < s->W; w++) memcpy(s->betaw + (size_t)w * s->Nt, s->beta, s->Nt * sizeof(double));
}
/* inverse temperature of chain (w, r) */
static inline double chain_beta(const ptmo_pt* s, int w, int r) { return s->betaw ? s->betaw[(size_t)w * s->Nt + r] : s->beta[r]; }

double ptmo_chunk_prefix(const double* v, int n, double* P) {
  double off = 0.0;
  for (int q = 0; 32 * q < n; q++) {
    double loc = 0.0;
    for (int k = 32 * q; k < n && k < 32 * q + 32; k++) {
      if (P) P[k] = loc;            /* local part first ... */
      loc = loc + v[k];
    }
    if (P)
      for (int k = 32 * q; k < n && k < 32 * q + 32; k++) P[k] = off + P[k];   /* ... then the chunk's offset */
    off = off + loc;
  }
  return off;
}
void ptmo_pt_enable_history(ptmo_pt* s, int cap) {
  size_t N = (size_t)s->Nt * s->W;
  s->hist_cap = cap;
  s->hist_x = (double*)calloc(N * cap * s->D, sizeof(double));
  s->hist_ll = (double*)calloc(N * cap, sizeof(double)); s->hist_lp = (double*)calloc(N * cap, sizeof(double));
  s->hist_beta = (double*)calloc(N * cap, sizeof(double));
  s->hist_nacc = (int32_t*)calloc(N * cap, 4); s->hist_ntry = (int32_t*)calloc(N * cap, 4); s->hist_type = (int32_t*)calloc(N * cap, 4);
}
/* the push_back block of MH_chain::add_state (chain.cc:935-946); row index = Nsize before the push */
static void hist_push(ptmo_pt* s, size_t c, int64_t row, double beta) {
  if (!s->hist_cap || row >= s->hist_cap) return;
  size_t o = c * s->hist_cap + (size_t)row;
  s->hist_beta[o] = beta;
  memcpy(s->hist_x + o * s->D, s->x + c * s->D, s->D * sizeof(double));
  s->hist_ll[o] = s->llike[c]; s->hist_lp[o] = s->lprior[c];
  s->hist_nacc[o] = s->naccept[c]; s->hist_ntry[o] = s->ntries[c]; s->hist_type[o] = s->last_type[c];
}

/* MH_chain::add_state bookkeeping (chain.cc:935-947) */
/* the MAP update of add_state (chain.cc:931-934) for the state chain c holds now, at its own temperature */
static void map_update(ptmo_pt* s, size_t c, double beta) {
  double lpost = ptmo_lpost(s->lprior[c], beta, s->llike[c]);
  if (lpost > s->map_lpost[c]) {
    s->map_lpost[c] = lpost;
    memcpy(s->map_x + c * s->D, s->x + c * s->D, s->D * sizeof(double));
  }
}
/* beta: the chain's inverse temperature at this call */
static inline void add_state_count(ptmo_pt* s, size_t c, double beta) {
  map_update(s, c, beta);
  if (s->nhist[c] % s->add_every_N == 0) { hist_push(s, c, s->nsize[c], beta); s->nsize[c]++; }
  s->nhist[c]++;
}

void ptmo_pt_set_states(ptmo_pt* s, const ptmo_problem* pb, const double* x, const double* llike) {
  size_t N = (size_t)s->Nt * s->W;
  memcpy(s->x, x, N * s->D * sizeof(double));
  for (size_t c = 0; c < N; c++) {
    double* xc = s->x + c * s->D;
    int valid = ptmo_enforce(pb, xc);                     /* state ctor enforces (states.cc:194-199) */
    s->lprior[c] = ptmo_lprior(pb, xc, valid);
    s->llike[c] = llike ? llike[c] : ptmo_llike(pb, xc);
    s->nhist[c] = 0; s->nsize[c] = 1;                     /* MH_chain::initialize(1): one row, Nhist reset (chain.cc:871-875) */
    s->map_lpost[c] = -1e200;
    map_update(s, c, chain_beta(s, (int)(c / s->Nt), (int)(c % s->Nt)));
    hist_push(s, c, 0, chain_beta(s, (int)(c / s->Nt), (int)(c % s->Nt)));
  }
}

/* ============================================================================================
 * differential_evolution::draw (proposal_distribution.cc:476-592; draw_i_from_chain :745-801 with unlikely_alpha = 0;
 * ter Braak & Vrugt, Stat Comput 18 (2008) 435: eqs. 2-4).  state::scalar_mult followed by state::add: every product is
 * rounded before its sum; innerprod sums in index order.
 * ============================================================================================ */
int ptmo_de_ready(int D, long rows) { return rows >= 10L * D; }
static long de_pick_row(int D, long rows, double ignore_frac, double u) {
  long spare = rows - 100L * D;                       /* size - get_min_cut_size() */
  long first = (spare * (1 - ignore_frac) > 10L * D) ? (long)(spare * ignore_frac) : 0;
  return (long)(first + (rows - first) * u);
}
int ptmo_de_draw(int D, const double* x, long rows, const ptmo_de_params* q, ptmo_de_uniform_fn u, void* uctx, ptmo_de_row_fn rowf,
                 void* rctx, double* xn, double* log_hastings) {
  const double u_snooker = u(uctx, 0);
  if (!(q->snooker > u_snooker)) {                    /* draw_standard (:487-536) */
    const double ug = u(uctx, 1);
    double gamma = 1.68 / sqrt((double)D) / q->reduce_gamma;
    if (ug < q->gamma_one_frac) gamma = 1;
    const double* z1 = rowf(rctx, de_pick_row(D, rows, q->ignore_frac, u(uctx, 2)));
    const double* z2 = rowf(rctx, de_pick_row(D, rows, q->ignore_frac, u(uctx, 3)));
    for (int i = 0; i < D; i++) {
      const double t1 = z1[i] * gamma;
      const double a = x[i] + t1;
      const double t2 = z2[i] * (-gamma);
      xn[i] = a + t2;
    }
    *log_hastings = 0;
    return 0;
  }
  /* draw_snooker (:539-592) */
  const double gamma = (1.2 + u(uctx, 1)) / q->reduce_gamma;
  const double* z = NULL;
  double axis2 = 0;
  for (int tries = 0; axis2 == 0; tries++) {          /* the history repeats states: z must differ from x */
    if (tries > 1000) return -1;
    z = rowf(rctx, de_pick_row(D, rows, q->ignore_frac, u(uctx, 4 + tries)));
    axis2 = 0;
    for (int i = 0; i < D; i++) { const double a = x[i] + z[i] * (-1.0); axis2 = axis2 + a * a; }
  }
  const double* z1 = rowf(rctx, de_pick_row(D, rows, q->ignore_frac, u(uctx, 2)));
  const double* z2 = rowf(rctx, de_pick_row(D, rows, q->ignore_frac, u(uctx, 3)));
  double proj = 0;
  for (int i = 0; i < D; i++) {
    const double a = z1[i] * gamma, b = z2[i] * (-gamma);
    const double diff = a + b;
    const double ax = x[i] + z[i] * (-1.0);
    proj = proj + diff * ax;
  }
  proj = proj / axis2;
  double fz2 = 0;
  for (int i = 0; i < D; i++) {
    const double ax = x[i] + z[i] * (-1.0);
    const double t = ax * proj;
    xn[i] = x[i] + t;
    const double f = xn[i] + z[i] * (-1.0);
    fz2 = fz2 + f * f;
  }
  *log_hastings = (ptmo_log(fz2) - ptmo_log(axis2)) * (D - 1) / 2.0;
  return 1;
}
void ptmo_pt_set_de(ptmo_pt* s, const ptmo_de_params* q, int n_init_extra, const double* init_rows) {
  s->de_on = q != NULL;
  free(s->de_init);
  s->de_init = NULL; s->de_init_extra = 0;
  if (!q) return;
  s->de = *q;
  if (n_init_extra > 0 && init_rows) {
    size_t n = (size_t)n_init_extra * s->Nt * s->W * s->D;
    s->de_init = (double*)malloc(n * sizeof(double));
    memcpy(s->de_init, init_rows, n * sizeof(double));
    s->de_init_extra = n_init_extra;
  }
}
typedef struct { const ptmo_pt* s; size_t c; } de_rows_ctx;
static const double* de_row_of_chain(void* v, long row) {
  de_rows_ctx* q = (de_rows_ctx*)v;
  const ptmo_pt* s = q->s;
  if (row < s->de_init_extra) return s->de_init + ((size_t)row * s->Nt * s->W + q->c) * s->D;
  return s->hist_x + (q->c * s->hist_cap + (size_t)(row - s->de_init_extra)) * s->D;
}
typedef struct { const ptmo_rng* rng; int w, r; uint64_t step; } de_uni_ctx;
static double de_uniform_of_chain(void* v, int slot) {
  de_uni_ctx* q = (de_uni_ctx*)v;
  return q->rng->de_uniform(q->rng->ctx, q->w, q->r, q->step, slot);
}

/* ============================================================================================
 * MH_chain::step  (chain.cc:966-1022) with gaussian_prop::draw (proposal_distribution.hh:194-218)
 * ============================================================================================ */
int ptmo_mh_step(ptmo_pt* s, const ptmo_problem* pb, const ptmo_proposal* prop, const ptmo_rng* rng, int w, int r) {
  int D = s->D;
  size_t c = (size_t)w * s->Nt + r;
  double beta = chain_beta(s, w, r);
  double* x = s->x + c * D;
  double cur_llike = s->llike[c], cur_lprior = s->lprior[c];
  double cur_lpost = ptmo_lpost(cur_lprior, beta, cur_llike);
  double oldlprior = cur_lpost - beta * cur_llike;                       /* :973 */
  double xn[PTMO_MAX_DIM], off[PTMO_MAX_DIM];
  double hast = 0.0;
  int type, valid;
  if (s->host_prop) {                                                    /* :975 prop.draw -- any proposal, evaluated by the caller */
    int32_t rr = r, ww = w, ty = 0, va = 1;
    s->host_prop(s->host_prop_user, 1, D, x, &rr, &ww, s->step, xn, &hast, &ty, &va);
    type = ty;
    valid = va != 0;                                                     /* the returned state's own validity (state::invalid()) */
  } else {
    /* a proposal set with differential evolution as a member (the member of negative scale): proposal_distribution_set::draw
     * (proposal_distribution.cc:99-129) picks the first READY member whose bin the uniform falls below */
    ptmo_proposal local = *prop;
    double lmix[3 * 64];
    int de_member = -1;
    if (s->de_on && prop->K > 0 && prop->K <= 64 && rng->de_uniform) {
      const double xs = prop->K > 1 ? rng->chain_uniform(rng->ctx, w, r, s->step, 3) : 0.0;
      int kmix = prop->K - 1;
      for (int k = 0; k < prop->K; k++)
        if (xs < prop->mix[3 * k]) { kmix = k; break; }
      if (prop->mix[3 * kmix + 1] < 0) {
        const long rows = s->de_init_extra + s->nsize[c];
        if (ptmo_de_ready(D, rows)) de_member = kmix;
        else {                                                           /* not ready: passed over -- its bin closes, the next member's is met */
          memcpy(lmix, prop->mix, (size_t)prop->K * 3 * sizeof(double));
          lmix[3 * kmix] = kmix > 0 ? lmix[3 * (kmix - 1)] : -1.0;
          local.mix = lmix;
        }
      }
    }
    if (de_member >= 0) {
      de_rows_ctx rc = {s, c};
      de_uni_ctx uc = {rng, w, r, s->step};
      const long rows = s->de_init_extra + s->nsize[c];
      const int t = ptmo_de_draw(D, x, rows, &s->de, de_uniform_of_chain, &uc, de_row_of_chain, &rc, xn, &hast);
      type = de_member + 10 * (t < 0 ? 0 : t);                           /* proposal_distribution.cc:117 */
      valid = t < 0 ? 0 : pb->origin_valid;                              /* state::add on an enforced origin: Q9 */
    } else {
      type = rng->draw_offset(rng->ctx, w, r, s->step, &local, D, off);   /* :975 prop.draw (gaussian_prop / scripted offsets) */
      if (rng->log_hastings) hast = rng->log_hastings(rng->ctx, w, r, s->step);
      for (int i = 0; i < D; i++) xn[i] = x[i] + off[i];                 /* state::add, states.cc:205-214 */
      valid = pb->origin_valid;                                          /* Q9 */
    }
  }
  if (valid) valid = ptmo_enforce(pb, xn);                               /* :976 newstate.enforce() */
  double newlprior = ptmo_lprior(pb, xn, valid);                         /* :977 */
  double newlike, newlpost;
  /* :980  (!current_lpost>-1e200 parses as (!current_lpost)>-1e200 == true: SURVEY Q1) */
  if (valid && (newlprior > -1e200 || newlprior - oldlprior > pb->minPrior)) {
    newlike = ptmo_llike(pb, xn);                                        /* :981 */
    newlpost = newlike * beta + newlprior;                               /* :982 */
  } else {
    newlike = newlpost = -INFINITY;                                      /* :986 */
  }
  double logH = hast;                                                    /* :989 prop.log_hastings_ratio() (gaussian_prop: 0) */
  int accept = 1;
  if (isnan(logH)) accept = 0;                                           /* :990-993 */
  logH += newlpost - cur_lpost;                                          /* :994 */
  if (!valid) accept = 0;                                                /* :996 */
  if (accept && logH < 0) {                                              /* :998 (NaN => stays accepted) */
    double u = rng->chain_uniform(rng->ctx, w, r, s->step, 0);
    accept = (ptmo_log(u) < logH);
  }
  s->ntries[c]++;                                                        /* :1005 */
  if (accept) {
    s->naccept[c]++;
    s->last_type[c] = type;
    memcpy(x, xn, D * sizeof(double));
    s->llike[c] = newlike; s->lprior[c] = newlprior;
  }
  add_state_count(s, c, beta);
  s->last_accept_mh[c] = (uint8_t)accept;
  return accept;
}

void ptmo_pt_set_host_proposal(ptmo_pt* s, ptmo_propose_fn fn, void* user) { s->host_prop = fn; s->host_prop_user = user; }

void ptmo_sweep(ptmo_pt* s, const ptmo_problem* pb, const ptmo_proposal* props, const ptmo_rng* rng, int nthreads) {
  long N = (long)s->Nt * s->W;
  (void)nthreads;
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
  for (long c = 0; c < N; c++) {
    int w = (int)(c / s->Nt), r = (int)(c % s->Nt);
    ptmo_mh_step(s, pb, &props[r], rng, w, r);
  }
  s->step++;
}

/* ============================================================================================
 * parallel_tempering_chains::step  (chain.cc:1393-1571)
 *
 * Evolving ladder (do_evolve_temps, chain.cc:1501-1518): after EVERY accepted exchange the reference widens that gap,
 * renormalises all the gaps so that they add up to 1 - beta_last again and resets every temperature (pry_temps,
 * chain.cc:1809-1846, lpost_cut < 0).  The same ladder is kept here in lazily normalised form: gaps sp[] and their sum S;
 * a pry is sp[i] *= 1 + rate, S += the increase; the gap a later trial of the step sees is sp[i] / (S / (1 - beta_last)) --
 * (its Metropolis test is taken with both sides multiplied by S) -- the reference's value up to rounding (O(1) per exchange instead of O(Nt); exact same bits until the step's first
 * accepted exchange).  At the end of the step the temperatures are rebuilt from the gaps' prefix sums
 * (ptmo_chunk_prefix), beta_k = 1 - P_k / (total / (1 - beta_last)) as the reference does (invtemp starts at 1).
 * ============================================================================================ */
static void swap_phase(ptmo_pt* s, const ptmo_rng* rng, int w) {
  int Nt = s->Nt, D = s->D, ms = s->maxswaps;
  int* iswaps = s->last_pairs + (size_t)w * ms;
  int* acc = s->last_accept + (size_t)w * ms;
  size_t base = (size_t)w * Nt;
  /* :1410-1420 candidate selection */
  for (int i = 0; i < ms; i++) {
    iswaps[i] = -2; acc[i] = 0;
    double x = rng->pt_uniform(rng->ctx, w, s->step, i, 0);
    if (Nt > 1 && x < (Nt - 1) * s->swap_rate / ms) {
      x = rng->pt_uniform(rng->ctx, w, s->step, i, 1);
      iswaps[i] = (int)(x * (Nt - 1));
      for (int j = 0; j < i; j++)
        if (iswaps[j] == iswaps[i] || iswaps[j] + 1 == iswaps[i]) iswaps[i] = -2;
    }
  }
  /* :1436-1537 trials, in pick order, on the in-place updated view */
  double tmp[PTMO_MAX_DIM];
  double *bw = NULL, *sp = NULL, *P0 = NULL, *inc = NULL, S = 0.0, c1 = 0.0, nrm = 1.0;
  int* ipry = NULL;
  const double grow = 1.0 + s->evolve_rate;
  int npry = 0;
  /* evolve_temp_lpost_cut >= 0 (chain.cc:1819-1827): every pry ALSO widens each gap whose two chains' log-posteriors are out
   * of order by more than cut * invtemp -- any gap of the ladder, so nothing stays lazy: after every accepted exchange the
   * gaps' prefix sums P0 and their total are rebuilt (ptmo_chunk_prefix), and a rung's temperature is 1 - P0 / normaliser */
  const int cutmode = s->evolve_cut >= 0;
  if (s->betaw && Nt > 1) {
    bw = s->betaw + base;
    sp = (double*)malloc((size_t)Nt * sizeof(double));
    P0 = (double*)malloc((size_t)Nt * sizeof(double));
    inc = (double*)malloc((size_t)ms * sizeof(double));
    ipry = (int*)malloc((size_t)ms * sizeof(int));
    for (int k = 0; k < Nt - 1; k++) sp[k] = bw[k] - bw[k + 1];        /* :1816 splits */
    S = ptmo_chunk_prefix(sp, Nt - 1, P0);
    c1 = 1 - bw[Nt - 1];                                                /* :1833 */
  }
  for (int j = 0; j < ms; j++) {
    int i = iswaps[j];
    if (i < 0) continue;
    size_t a = base + i, b = base + i + 1;
    double lla = s->llike[a]; if (!(lla > -1e200)) lla = -1e200;        /* :1459 */
    double llb = s->llike[b]; if (!(llb > -1e200)) llb = -1e200;        /* :1461 */
    /* the two rungs' temperatures as this pick's add_state calls see them (before the pick's own pry): the stored ones
     * until something was pried this step, then 1 - (P0 + D) / normaliser with P0 the prefix sum of the step's first gaps
     * and D what the earlier pries added to the gaps below the rung, in pick order; the ladder's ends never move */
    double ba = s->beta[i], bb = s->beta[i + 1];
    if (sp) {
      ba = bw[i]; bb = bw[i + 1];
      if (npry && cutmode) {
        if (i > 0) ba = 1 - P0[i] / nrm;
        if (i + 1 < Nt - 1) bb = 1 - P0[i + 1] / nrm;
      } else if (npry) {
        double Da = 0.0, Db = 0.0;
        for (int q = 0; q < npry; q++) {
          if (ipry[q] < i) Da = Da + inc[q];
          if (ipry[q] < i + 1) Db = Db + inc[q];
        }
        if (i > 0) ba = 1 - (P0[i] + Da) / nrm;
        if (i + 1 < Nt - 1) bb = 1 - (P0[i + 1] + Db) / nrm;
      }
    }
    int accept = 1;
    if (sp && npry) {
      /* log u < logH with logH = (sp[i] / (S / c1)) * (llb - lla), both sides multiplied by S > 0 (no division in the chain
       * of dependent trials) */
      double t = (sp[i] * c1) * (llb - lla);
      if (t < 0) {
        double u = rng->pt_uniform(rng->ctx, w, s->step, j, 2);
        accept = (ptmo_log(u) * S < t);
      }
    } else {
      double db = sp ? -sp[i] : s->beta[i + 1] - s->beta[i];            /* (nothing pried yet: the stored temperatures' difference) */
      double logH = -db * (llb - lla);                                  /* :1463 */
      if (logH < 0) {
        double u = rng->pt_uniform(rng->ctx, w, s->step, j, 2);
        accept = (ptmo_log(u) < logH);                                   /* :1464-1467 */
      }
    }
    if (accept) {                                                        /* :1487-1492 exchange states, temps stay */
      memcpy(tmp, s->x + a * D, D * sizeof(double));
      memcpy(s->x + a * D, s->x + b * D, D * sizeof(double));
      memcpy(s->x + b * D, tmp, D * sizeof(double));
      double t = s->llike[a]; s->llike[a] = s->llike[b]; s->llike[b] = t;
      t = s->lprior[a]; s->lprior[a] = s->lprior[b]; s->lprior[b] = t;   /* lprior is a pure function of the state */
      s->swap_accept_count[(size_t)w * (Nt - 1) + i]++;                   /* :1498 */
      if (sp && cutmode) {                                               /* :1516-1517 pry_temps({i}, rate, invtemps, gather_lposts()) */
        /* the chains' current log-posteriors: lprior + invtemp * llike at the temperatures of this moment (every pry resets
         * them, MH_chain::resetTemp chain.cc:1088-1091), the exchanged pair already in place */
        double lpk = 0.0, bk = 0.0;
        for (int k = 0; k < Nt; k++) {
          double bnext = (k == 0 || k == Nt - 1 || !npry) ? bw[k] : 1 - P0[k] / nrm;
          double lnext = ptmo_lpost(s->lprior[base + k], bnext, s->llike[base + k]);
          if (k > 0 && lpk - lnext > s->evolve_cut * bk) sp[k - 1] = sp[k - 1] * grow;   /* :1819-1827, pow(1+rate, 1) */
          lpk = lnext; bk = bnext;
        }
        sp[i] = sp[i] * grow;                                            /* :1829 */
        S = ptmo_chunk_prefix(sp, Nt - 1, P0);
        nrm = S / c1;                                                    /* :1833 */
        npry++;
      } else if (sp) {                                                   /* :1501-1518 pry_temps({i}) */
        double sn = sp[i] * grow;                                        /* :1829 */
        ipry[npry] = i; inc[npry] = sn - sp[i];
        S = S + inc[npry];
        sp[i] = sn;
        nrm = S / c1;                                                    /* :1833 */
        npry++;
      }
    }
    acc[j] = accept;
    add_state_count(s, a, ba); add_state_count(s, b, bb);                /* add_state on both rungs either way (:1487-1490,:1531-1534) */
    s->touched[a]++; s->touched[b]++;
    s->swap_count[(size_t)w * (Nt - 1) + i]++;                            /* :1536 */
  }
  if (npry && cutmode) {
    for (int k = 1; k < Nt - 1; k++) bw[k] = 1 - P0[k] / nrm;            /* :1834-1844 */
  } else if (npry) {                                                     /* :1834-1844 the new temperatures */
    double* P = (double*)malloc((size_t)Nt * sizeof(double));
    double total = ptmo_chunk_prefix(sp, Nt - 1, P);
    double nn = total / c1;
    for (int k = 1; k < Nt - 1; k++) bw[k] = 1 - P[k] / nn;
    free(P);
  }
  free(sp); free(P0); free(inc); free(ipry);
}

/* the exchange phase of a step alone (chain.cc:1410-1537) -- for the CPU stand-in of a rung shard (tests/oracle_shard.py), which
 * runs it on a replica of the whole ladder and makes the Metropolis moves of its own rungs itself */
void ptmo_exchange_phase(ptmo_pt* s, const ptmo_rng* rng) {
  long N = (long)s->Nt * s->W;
  memset(s->touched, 0, (size_t)N);
  for (int w = 0; w < s->W; w++) swap_phase(s, rng, w);
}

void ptmo_pt_step(ptmo_pt* s, const ptmo_problem* pb, const ptmo_proposal* props, const ptmo_rng* rng, int nthreads) {
  long N = (long)s->Nt * s->W;
  memset(s->touched, 0, (size_t)N);
  (void)nthreads;
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
  for (int w = 0; w < s->W; w++) swap_phase(s, rng, w);
  /* :1544-1559 MH move for every rung not involved in a swap attempt this step */
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
  for (long c = 0; c < N; c++) {
    if (s->touched[c]) { s->last_accept_mh[c] = 2; continue; }
    int w = (int)(c / s->Nt), r = (int)(c % s->Nt);
    ptmo_mh_step(s, pb, &props[r], rng, w, r);
  }
  s->step++;
}

/* ============================================================================================
 * Census of the candidate selection of parallel_tempering_chains::step (chain.cc:1410-1420), for the multi-GPU halo.
 *
 * A shard that holds rungs [.., b) replays the exchanges that reach it from above out of the llikes of the `H` rungs
 * above b (ptm_exchange_decide).  That fails -- loudly -- exactly when the SURVIVING picks of a step cover every pair
 * (b-1, b), (b, b+1), ..., (b-1+H, b+H): a run of H+1 consecutive surviving picks starting at rung b-1, accepted or not.
 * This function replays the step's candidate draws of `W` ladders over `nsteps` steps with the engine's own ladder
 * streams and counts, for every boundary b in bounds[nb], how often the run of consecutive surviving picks that starts at
 * rung b-1 has length L: hist[i*(Lmax+1) + min(L, Lmax)].  (A pick n is dropped iff an earlier surviving pick is n or n-1,
 * chain.cc:1417-1418: kept here as a per-rung flag instead of the reference's loop over the earlier picks.)
 * ============================================================================================ */
void ptmo_selection_run_census(uint64_t seed, int Nt, double swap_rate, int W, uint64_t step0, int nsteps, const int* bounds,
                               int nb, int Lmax, int64_t* hist, int nthreads) {
  const int ms = (int)(1 + 2 * swap_rate * Nt);                                      /* chain.cc:1192 */
  const double thresh = (Nt - 1) * swap_rate / ms;                                   /* chain.cc:1413 */
  (void)nthreads;
#pragma omp parallel num_threads(nthreads) if (nthreads > 1)
  {
    unsigned char* alive = (unsigned char*)calloc((size_t)Nt + 1, 1);
    int* picked = (int*)malloc((size_t)ms * sizeof(int));
    int64_t* h = (int64_t*)calloc((size_t)nb * (Lmax + 1), sizeof(int64_t));
#pragma omp for schedule(static)
    for (int w = 0; w < W; w++)
      for (int t = 0; t < nsteps; t++) {
        int np = 0;
        for (int k = 0; k < ms; k++) {
          uint32_t o[4];
          ptmo_draw_block(seed, PTMO_TAG_PT, (uint32_t)w, step0 + (uint64_t)t, (uint32_t)k, o);
          if (!(Nt > 1 && ptmo_u01(o[0]) < thresh)) continue;                         /* :1413 */
          int n = (int)(ptmo_u01(o[1]) * (Nt - 1));                                   /* :1415 */
          if (alive[n] || (n > 0 && alive[n - 1])) continue;                          /* :1417-1418 */
          alive[n] = 1;
          picked[np++] = n;
        }
        for (int i = 0; i < nb; i++) {
          int L = 0;
          for (int r = bounds[i] - 1; r >= 0 && r < Nt - 1 && alive[r]; r++) L++;
          h[(size_t)i * (Lmax + 1) + (L < Lmax ? L : Lmax)]++;
        }
        for (int k = 0; k < np; k++) alive[picked[k]] = 0;
      }
#pragma omp critical
    for (int i = 0; i < nb * (Lmax + 1); i++) hist[i] += h[i];
    free(alive); free(picked); free(h);
  }
}

/* ============================================================================================
 * RNG providers
 * ============================================================================================ */
typedef struct { uint64_t seed; int Nt; } philox_ctx;

static double ph_chain_uniform(void* vctx, int w, int r, uint64_t step, int slot) {
  philox_ctx* c = (philox_ctx*)vctx;
  uint32_t o[4];
  ptmo_draw_block(c->seed, PTMO_TAG_MH, (uint32_t)((uint64_t)w * c->Nt + r), step, 0, o);
  return ptmo_u01(o[slot]);
}
static int ph_draw_offset(void* vctx, int w, int r, uint64_t step, const ptmo_proposal* p, int D, double* off) {
  philox_ctx* c = (philox_ctx*)vctx;
  uint32_t stream = (uint32_t)((uint64_t)w * c->Nt + r);
  double z[PTMO_MAX_DIM + 4];
  for (int b = 0; 4 * b < D; b++) {                       /* gaussian_dist_product::drawSample: D normals (probability_function.cc:37-47) */
    uint32_t o[4];
    ptmo_draw_block(c->seed, PTMO_TAG_MH, stream, step, (uint32_t)(b + 1), o);
    ptmo_boxmuller(o[0], o[1], &z[4 * b], &z[4 * b + 1]);
    ptmo_boxmuller(o[2], o[3], &z[4 * b + 2], &z[4 * b + 3]);
  }
  int type = 0, kmix = 0;
  double odf = p->oneDfrac, scale = 1.0;
  if (p->K > 0) {                                         /* proposal_distribution_set::draw (proposal_distribution.cc:99-129) */
    double xs = 0.0;
    if (p->K > 1) {                                       /* a set of one draws nothing (:105-110) */
      uint32_t o[4];
      ptmo_draw_block(c->seed, PTMO_TAG_MH, stream, step, 0, o);
      xs = ptmo_u01(o[3]);
    }
    kmix = p->K - 1;
    for (int k = 0; k < p->K; k++)
      if (xs < p->mix[3 * k]) { kmix = k; break; }
    scale = p->mix[3 * kmix + 1];
    odf = p->mix[3 * kmix + 2];
  }
  if (odf > 0) {                                          /* proposal_distribution.hh:196-206 */
    uint32_t o[4];
    ptmo_draw_block(c->seed, PTMO_TAG_MH, stream, step, 0, o);
    double x = ptmo_u01(o[1]);
    if (x < odf) {
      int ax = (int)(D * ptmo_u01(o[2]));
      for (int j = 0; j < D; j++) if (j != ax) z[j] = 0.0;
      type = 1;
    }
  }
  if (p->kind == PTMO_PROP_DIAG) {
    for (int i = 0; i < D; i++) off[i] = p->M[i] * z[i];
  } else {
    int ord[PTMO_MAX_DIM];
    int n = ptmo_column_order(D, ord);
    for (int i = 0; i < D; i++) {
      double a = 0.0;
      for (int t = 0; t < n; t++) a = fma(p->M[i * D + ord[t]], z[ord[t]], a);
      off[i] = a;
    }
  }
  if (p->K > 0) {
    for (int i = 0; i < D; i++) off[i] = scale * off[i];  /* the member is scale_k times the base factor */
    type = kmix + 10 * type;                              /* proposal_distribution.cc:117 */
  }
  return type;
}
static double ph_pt_uniform(void* vctx, int w, uint64_t step, int k, int slot) {
  philox_ctx* c = (philox_ctx*)vctx;
  uint32_t o[4];
  ptmo_draw_block(c->seed, PTMO_TAG_PT, (uint32_t)w, step, (uint32_t)k, o);
  return ptmo_u01(o[slot]);
}
/* uniform `slot` of the chain's differential-evolution draw of this step (ptm_oracle.h: ptmo_de_uniform_fn) */
static double ph_de_uniform(void* vctx, int w, int r, uint64_t step, int slot) {
  philox_ctx* c = (philox_ctx*)vctx;
  uint32_t o[4];
  ptmo_draw_block(c->seed, PTMO_TAG_MH, (uint32_t)((uint64_t)w * c->Nt + r), step, slot < 4 ? 0x0DE00000u : 0x0DE00001u + (uint32_t)(slot - 4), o);
  return ptmo_u01(o[slot < 4 ? slot : 0]);
}
ptmo_rng* ptmo_rng_philox(uint64_t seed, int Nt) {
  ptmo_rng* r = (ptmo_rng*)calloc(1, sizeof *r);
  philox_ctx* c = (philox_ctx*)calloc(1, sizeof *c);
  c->seed = seed; c->Nt = Nt;
  r->ctx = c; r->chain_uniform = ph_chain_uniform; r->draw_offset = ph_draw_offset; r->pt_uniform = ph_pt_uniform;
  r->de_uniform = ph_de_uniform;
  return r;
}

typedef struct {
  int W, Nt, D, len_c, len_p, nsteps;
  const double *chain_tapes, *pt_tapes, *deltas;
  int *cpos, *ppos, *dpos;
  const double* hastings;   /* [W*Nt][nsteps] scripted log-Hastings ratio of offset k of the chain, or NULL */
  const int32_t* types;     /* [W*Nt][nsteps] its type code, or NULL (0) */
} tape_ctx;
static double tp_chain_uniform(void* v, int w, int r, uint64_t step, int slot) {
  tape_ctx* t = (tape_ctx*)v; (void)step; (void)slot;
  size_t c = (size_t)w * t->Nt + r;
  return t->chain_tapes[c * t->len_c + t->cpos[c]++];
}
static int tp_draw_offset(void* v, int w, int r, uint64_t step, const ptmo_proposal* p, int D, double* off) {
  tape_ctx* t = (tape_ctx*)v; (void)step; (void)p;
  size_t c = (size_t)w * t->Nt + r;
  const int k = t->dpos[c]++;
  const double* d = t->deltas + (c * t->nsteps + k) * D;
  for (int i = 0; i < D; i++) off[i] = d[i];
  return t->types ? t->types[c * t->nsteps + k] : 0;
}
static double tp_log_hastings(void* v, int w, int r, uint64_t step) {   /* of the offset drawn last */
  tape_ctx* t = (tape_ctx*)v; (void)step;
  size_t c = (size_t)w * t->Nt + r;
  return t->hastings[c * t->nsteps + t->dpos[c] - 1];
}
void ptmo_rng_tape_hastings(ptmo_rng* r, const double* log_hastings, const int32_t* types) {
  tape_ctx* t = (tape_ctx*)r->ctx;
  t->hastings = log_hastings; t->types = types;
  r->log_hastings = log_hastings ? tp_log_hastings : NULL;
}
static double tp_pt_uniform(void* v, int w, uint64_t step, int k, int slot) {
  tape_ctx* t = (tape_ctx*)v; (void)step; (void)k; (void)slot;
  return t->pt_tapes[(size_t)w * t->len_p + t->ppos[w]++];
}
ptmo_rng* ptmo_rng_tape(int W, int Nt, int D, const double* chain_tapes, int len_c, const double* pt_tapes, int len_p,
                        const double* deltas, int nsteps) {
  ptmo_rng* r = (ptmo_rng*)calloc(1, sizeof *r);
  tape_ctx* t = (tape_ctx*)calloc(1, sizeof *t);
  t->W = W; t->Nt = Nt; t->D = D; t->len_c = len_c; t->len_p = len_p; t->nsteps = nsteps;
  t->chain_tapes = chain_tapes; t->pt_tapes = pt_tapes; t->deltas = deltas;
  t->cpos = (int*)calloc((size_t)W * Nt, sizeof(int)); t->ppos = (int*)calloc(W, sizeof(int)); t->dpos = (int*)calloc((size_t)W * Nt, sizeof(int));
  r->ctx = t; r->chain_uniform = tp_chain_uniform; r->draw_offset = tp_draw_offset; r->pt_uniform = tp_pt_uniform;
  return r;
}
void ptmo_rng_free(ptmo_rng* r) {
  if (!r) return;
  if (r->chain_uniform == tp_chain_uniform) { tape_ctx* t = (tape_ctx*)r->ctx; free(t->cpos); free(t->ppos); free(t->dpos); }
  free(r->ctx); free(r);
}

/* inverse cdfs of UniformPolarDist / UniformCoPolarDist (ProbabilityDist.h:108-110,149-151) by 64 bisections on the
 * deterministic cos / sin above, and of UniformLogDist (:32-34): the engine's own draw procedure, restated */
static double cos_0_pi(double x) { return x <= HPI_HI ? ptmo_cos_hpi(x) : -ptmo_cos_hpi((PI_HI - x) + PI_LO); }
static double sin_hpi(double x) { return x >= 0 ? ptmo_sin_0_pi(x) : -ptmo_sin_0_pi(-x); }
static double draw_polar(double u, double lo, double hi) {
  double cl = cos_0_pi(lo), ch = cos_0_pi(hi);
  double y = cl - u * (cl - ch);
  double a = lo, b = hi;
  for (int k = 0; k < 64; k++) {
    double m = 0.5 * (a + b);
    if (cos_0_pi(m) > y) a = m; else b = m;
  }
  return 0.5 * (a + b);
}
static double draw_copolar(double u, double lo, double hi) {
  double sl = sin_hpi(lo), sh = sin_hpi(hi);
  double y = sl + u * (sh - sl);
  double a = lo, b = hi;
  for (int k = 0; k < 64; k++) {
    double m = 0.5 * (a + b);
    if (sin_hpi(m) < y) a = m; else b = m;
  }
  return 0.5 * (a + b);
}
static double draw_log(double u, double lo, double hi) {
  double l0 = ptmo_log(lo);
  return ptmo_exp(u * (ptmo_log(hi) - l0) + l0);
}

/* MH_chain::initialize(1) (chain.cc:846-876): draw from the prior until valid with llike >= -1e100.
 * Dimension d of attempt a uses block d of stream (chain), step = a, tag INIT:
 * uniform dims x = u*(hi-lo)+lo (ProbabilityDist.h:94-97), gaussian dims x = z*sigma+x0 (ProbabilityDist.cxx:79). */
void ptmo_init_from_prior(ptmo_pt* s, const ptmo_problem* pb, uint64_t seed) {
  int D = s->D;
  size_t N = (size_t)s->Nt * s->W;
  for (size_t c = 0; c < N; c++) {
    double* x = s->x + c * D;
    for (uint64_t a = 0;; a++) {
      for (int d = 0; d < D; d++) {
        uint32_t o[4];
        ptmo_draw_block(seed, PTMO_TAG_INIT, (uint32_t)c, a, (uint32_t)d, o);
        if (pb->ptype[d] == PTMO_UNIFORM) x[d] = ptmo_u01(o[0]) * (pb->phi[d] - pb->plo[d]) + pb->plo[d];
        else if (pb->ptype[d] == PTMO_GAUSSIAN) { double z0, z1; ptmo_boxmuller(o[0], o[1], &z0, &z1); x[d] = z0 * pb->phi[d] + pb->plo[d]; }
        else if (pb->ptype[d] == PTMO_POLAR) x[d] = draw_polar(ptmo_u01(o[0]), pb->plo[d], pb->phi[d]);
        else if (pb->ptype[d] == PTMO_COPOLAR) x[d] = draw_copolar(ptmo_u01(o[0]), pb->plo[d], pb->phi[d]);
        else if (pb->ptype[d] == PTMO_LOG) x[d] = draw_log(ptmo_u01(o[0]), pb->plo[d], pb->phi[d]);
        else x[d] = NAN;
      }
      int valid = ptmo_enforce(pb, x);
      if (!valid) continue;
      double ll = ptmo_llike(pb, x);
      if (ll < -1e100) continue;
      s->llike[c] = ll;
      s->lprior[c] = ptmo_lprior(pb, x, 1);
      break;
    }
    s->nhist[c] = 0; s->nsize[c] = 1;
    s->map_lpost[c] = -1e200;
    map_update(s, c, chain_beta(s, (int)(c / s->Nt), (int)(c % s->Nt)));
    hist_push(s, c, 0, chain_beta(s, (int)(c / s->Nt), (int)(c % s->Nt)));
  }
}
