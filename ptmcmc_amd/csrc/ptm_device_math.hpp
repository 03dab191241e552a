// ptm_device_math.hpp -- gfx950 device functions: Philox4x32-10, the open-interval uniform map,
// Box-Muller, and the deterministic log / exp / sin / cos used on the hot path.
//
// Numerics contract (DESIGN.md "Numerics"): every result is produced by IEEE-754 binary64
// +, -, *, /, sqrt and explicitly written fma in the order written here; the translation unit is
// compiled with -ffp-contract=off so hipcc adds no contractions of its own.  The CPU checker
// (oracle/ptm_oracle.c) states the same sequences independently; tests compare bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptm {

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11).  Replaces newran's MotherOfAll
// (ProbabilityDist/newran1.cxx:383-432) -- counter based, so a chain's stream depends only on
// (seed, chain identity, step), never on launch geometry.
// ------------------------------------------------------------------------------------------------
struct u32x4 { uint32_t v0, v1, v2, v3; };
// rounds of the generator: ONE constant shared with the checker (oracle/ptm_oracle.c: PTM_PHILOX_ROUNDS); Random123 publishes
// known answers for 7 and for 10 rounds (tests/test_oracle_golden.py holds both sets)
#ifndef PTM_PHILOX_ROUNDS
#define PTM_PHILOX_ROUNDS 10
#endif

// one round and its key bump (a kernel that spreads a block's ten rounds over its schedule calls this itself)
struct philox_state { uint32_t c0, c1, c2, c3, k0, k1; };
__device__ __forceinline__ void philox_round(philox_state& s) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * s.c0;
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * s.c2;
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ s.c1 ^ s.k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ s.c3 ^ s.k1;
  const uint32_t n3 = (uint32_t)p0;
  s.c0 = n0; s.c1 = n1; s.c2 = n2; s.c3 = n3;
  s.k0 += 0x9E3779B9u; s.k1 += 0xBB67AE85u;
}
__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                uint32_t k1) {
  philox_state s = {c0, c1, c2, c3, k0, k1};
#pragma unroll
  for (int r = 0; r < PTM_PHILOX_ROUNDS; ++r) philox_round(s);
  return {s.c0, s.c1, s.c2, s.c3};
}

// counter layout of the engine: c0 = block, c1 = stream, c2 = step[31:0], c3 = step[55:32] | tag << 24
enum { TAG_MH = 0, TAG_PT = 1, TAG_INIT = 2 };
__device__ __forceinline__ philox_state draw_block_begin(uint64_t seed, int tag, uint32_t stream, uint64_t step, uint32_t block) {
  return {block, stream, (uint32_t)step, ((uint32_t)(step >> 32) & 0x00FFFFFFu) | ((uint32_t)tag << 24), (uint32_t)seed, (uint32_t)(seed >> 32)};
}
__device__ __forceinline__ u32x4 draw_block(uint64_t seed, int tag, uint32_t stream, uint64_t step, uint32_t block) {
  return philox4x32_10(block, stream, (uint32_t)step, ((uint32_t)(step >> 32) & 0x00FFFFFFu) | ((uint32_t)tag << 24),
                       (uint32_t)seed, (uint32_t)(seed >> 32));
}

// (k + 0.5) / 2^32: the open-interval map of MotherOfAll::Next (newran1.cxx:432), so log(u) is finite
// (one fma: k + 0.5 is exact and the scale a power of two, so fma(k, 2^-32, 2^-33) rounds the same real number -- the same bits as
//  the checker's add-then-multiply, one instruction less)
__device__ __forceinline__ double u01(uint32_t k) { return __builtin_fma((double)k, 1.0 / 4294967296.0, 0.5 / 4294967296.0); }
// The two tests the exchange phase makes on such uniforms, in integers -- both exact, seven f64-rate instructions less per candidate:
//   u01(k) < t          <=>  k + 1/2 < t 2^32  <=>  k < ceil(t 2^32 - 1/2)   (t 2^32 and the difference are exact: t <= 1 has an ulp below 2^-20 there)
//   (int)(u01(k) * m)    =   floor((2k + 1) m / 2^33)                         (the product is exact in f64 for m < 2^20: the same floor)
__device__ __forceinline__ uint64_t u01_below_bound(double t) {
  const double T = t * 4294967296.0 - 0.5;
  if (!(T > 0.0)) return 0ull;
  if (T >= 4294967296.0) return 4294967296ull;
  return (uint64_t)__builtin_ceil(T);
}
__device__ __forceinline__ int u01_times(uint32_t k, int m) { return (int)(((2ull * k + 1ull) * (uint64_t)m) >> 33); }

// ------------------------------------------------------------------------------------------------
// log / exp: classic argument reduction + polynomial (coefficients of the FreeBSD/fdlibm e_log.c /
// e_exp.c algorithms), written with explicit fma Horner steps.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double dlog(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01, L3 = 2.857142874366239149e-01,
               L4 = 2.222219843214978396e-01, L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
               L7 = 1.479819860511658591e-01;
  int e = 0;
  if (x != x) return x;
  if (x < 0.0) return __builtin_nan("");
  if (x == 0.0) return -__builtin_inf();
  if (x == __builtin_inf()) return x;
  if (x < 2.2250738585072014e-308) { x *= 18014398509481984.0; e = -54; }
  const uint64_t b = (uint64_t)__double_as_longlong(x);
  e += (int)(b >> 52) - 1023;
  double m = __longlong_as_double((long long)((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull));
  if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
  const double f = m - 1.0;
  const double s = f / (2.0 + f);
  const double z = s * s;
  double r = L7;
  r = __builtin_fma(r, z, L6); r = __builtin_fma(r, z, L5); r = __builtin_fma(r, z, L4);
  r = __builtin_fma(r, z, L3); r = __builtin_fma(r, z, L2); r = __builtin_fma(r, z, L1);
  const double R = r * z;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)e;
  return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

// A 64-bit constant that lives in a scalar register pair made where it is used.  A v_fma_f64 cannot carry a 64-bit literal,
// so the compiler hoists such constants out of the caller's main loop into VECTOR registers and keeps (or spills) them there
// for the whole kernel -- fourteen VGPRs for the log polynomial of ONE accept uniform per tile in the hot kernel, which is
// at its register cap.  The volatile statement cannot be hoisted; two s_mov per constant per use cost nothing.
template <uint32_t HI, uint32_t LO>
__device__ __forceinline__ double sconst() {
  uint32_t a, b;
  asm volatile("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3" : "=s"(a), "=s"(b) : "n"(LO), "n"(HI));
  return __longlong_as_double((long long)(((uint64_t)b << 32) | a));
}

// A copy of x the optimiser knows nothing about: what is computed from it stays where it is written instead of being hoisted
// out of the enclosing loop into registers that live for the whole kernel (e.g. the first Philox round of a lane-constant
// counter word: ten VGPRs in the hot kernel).
__device__ __forceinline__ int opaque_copy(int x) {
  asm volatile("" : "+v"(x));
  return x;
}

// log of an open-interval uniform u = (k+0.5)/2^32: u is a positive normal number in [2^-33, 1), so none of
// dlog()'s special cases can fire; this is the same operation sequence without them (bit-identical results).
__device__ __forceinline__ double dlog_u01(uint32_t k) {
  // (the values of dlog(): ln2_hi 6.93147180369123816490e-01, ln2_lo 1.90821492927058770002e-10, L1..L7 of e_log.c)
  const double ln2_hi = sconst<0x3fe62e42u, 0xfee00000u>(), ln2_lo = sconst<0x3dea39efu, 0x35793c76u>();
  const double L1 = sconst<0x3fe55555u, 0x55555593u>(), L2 = sconst<0x3fd99999u, 0x9997fa04u>(), L3 = sconst<0x3fd24924u, 0x94229359u>(),
               L4 = sconst<0x3fcc71c5u, 0x1d8e78afu>(), L5 = sconst<0x3fc74664u, 0x96cb03deu>(), L6 = sconst<0x3fc39a09u, 0xd078c69fu>(),
               L7 = sconst<0x3fc2f112u, 0xdf3e5244u>();
  const double x = u01(k);
  const uint64_t b = (uint64_t)__double_as_longlong(x);
  int e = (int)(b >> 52) - 1023;
  double m = __longlong_as_double((long long)((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull));
  const bool big = m > 1.4142135623730951;
  m = big ? m * 0.5 : m;
  e += big ? 1 : 0;
  const double f = m - 1.0;
  const double s = f / (2.0 + f);
  const double z = s * s;
  double r = L7;
  r = __builtin_fma(r, z, L6); r = __builtin_fma(r, z, L5); r = __builtin_fma(r, z, L4);
  r = __builtin_fma(r, z, L3); r = __builtin_fma(r, z, L2); r = __builtin_fma(r, z, L1);
  const double R = r * z;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)e;
  return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

__device__ __forceinline__ double dexp(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
               invln2 = 1.44269504088896338700e+00;
  const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
               P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  if (x != x) return x;
  if (x > 709.782712893384) return __builtin_inf();
  if (x < -745.1332191019412) return 0.0;
  const int k = (int)(x * invln2 + (x < 0.0 ? -0.5 : 0.5));
  const double dk = (double)k;
  const double hi = x - dk * ln2_hi;
  const double lo = dk * ln2_lo;
  const double r = hi - lo;
  const double t = r * r;
  double p = P5;
  p = __builtin_fma(p, t, P4); p = __builtin_fma(p, t, P3); p = __builtin_fma(p, t, P2); p = __builtin_fma(p, t, P1);
  const double c = r - t * p;
  const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
  if (k >= -1021 && k <= 1023) return y * __longlong_as_double((long long)((uint64_t)(k + 1023) << 52));
  if (k > 1023) return y * 2.0 * __longlong_as_double((long long)((uint64_t)(k - 1 + 1023) << 52));
  return (y * __longlong_as_double((long long)((uint64_t)(k + 1000 + 1023) << 52))) *
         __longlong_as_double((long long)((uint64_t)(-1000 + 1023) << 52));
}

// sin / cos on [0, pi/4]: Taylor polynomials, Horner with fma
__device__ __forceinline__ double sin_k(double p) {
  const double z = p * p;
  double r = -1.0 / 1307674368000.0;
  r = __builtin_fma(r, z, 1.0 / 6227020800.0);
  r = __builtin_fma(r, z, -1.0 / 39916800.0);
  r = __builtin_fma(r, z, 1.0 / 362880.0);
  r = __builtin_fma(r, z, -1.0 / 5040.0);
  r = __builtin_fma(r, z, 1.0 / 120.0);
  r = __builtin_fma(r, z, -1.0 / 6.0);
  return __builtin_fma(p * z, r, p);
}
__device__ __forceinline__ double cos_k(double p) {
  const double z = p * p;
  double r = 1.0 / 20922789888000.0;
  r = __builtin_fma(r, z, -1.0 / 87178291200.0);
  r = __builtin_fma(r, z, 1.0 / 479001600.0);
  r = __builtin_fma(r, z, -1.0 / 3628800.0);
  r = __builtin_fma(r, z, 1.0 / 40320.0);
  r = __builtin_fma(r, z, -1.0 / 720.0);
  r = __builtin_fma(r, z, 1.0 / 24.0);
  return __builtin_fma(z * z, r, __builtin_fma(-0.5, z, 1.0));
}

constexpr double PI_HI = 3.141592653589793116e+00, PI_LO = 1.224646799147353207e-16;
constexpr double HPI_HI = 1.570796326794896558e+00, HPI_LO = 6.123233995736766036e-17;
constexpr double QPI = 7.853981633974482790e-01;

// sin on [0, pi]: UniformPolarDist::pdf (ProbabilityDist.h:197-201)
__device__ __forceinline__ double dsin_0_pi(double x) {
  if (x > HPI_HI) x = (PI_HI - x) + PI_LO;
  if (x <= QPI) return sin_k(x);
  return cos_k((HPI_HI - x) + HPI_LO);
}
// cos on [-pi/2, pi/2]: UniformCoPolarDist::pdf (ProbabilityDist.h:243-247)
__device__ __forceinline__ double dcos_hpi(double x) {
  x = __builtin_fabs(x);
  if (x <= QPI) return cos_k(x);
  return sin_k((HPI_HI - x) + HPI_LO);
}

// Prior draws of MH_chain::initialize (chain.cc:846-876) for the support types without a closed form in the functions
// above: the inverse cdfs of UniformPolarDist / UniformCoPolarDist (ProbabilityDist.h:108-110,149-151: acos / asin of a
// uniform in the cosines / sines of the limits) by 64 bisections on the monotone dcos / dsin -- set-up code, exactness of
// the operation sequence (shared with the CPU checker) matters, speed does not; UniformLogDist::invcdf (:32-34) directly.
__device__ __forceinline__ double dcos_0_pi(double x) { return x <= HPI_HI ? dcos_hpi(x) : -dcos_hpi((PI_HI - x) + PI_LO); }
__device__ __forceinline__ double dsin_hpi(double x) { return x >= 0 ? dsin_0_pi(x) : -dsin_0_pi(-x); }
__device__ __forceinline__ double draw_polar(double u, double lo, double hi) {
  const double cl = dcos_0_pi(lo), ch = dcos_0_pi(hi);
  const double y = cl - u * (cl - ch);
  double a = lo, b = hi;
#pragma unroll 1
  for (int k = 0; k < 64; ++k) {
    const double m = 0.5 * (a + b);
    if (dcos_0_pi(m) > y) a = m; else b = m;
  }
  return 0.5 * (a + b);
}
__device__ __forceinline__ double draw_copolar(double u, double lo, double hi) {
  const double sl = dsin_hpi(lo), sh = dsin_hpi(hi);
  const double y = sl + u * (sh - sl);
  double a = lo, b = hi;
#pragma unroll 1
  for (int k = 0; k < 64; ++k) {
    const double m = 0.5 * (a + b);
    if (dsin_hpi(m) < y) a = m; else b = m;
  }
  return 0.5 * (a + b);
}
__device__ __forceinline__ double draw_log(double u, double lo, double hi) {
  const double l0 = dlog(lo);
  return dexp(u * (dlog(hi) - l0) + l0);
}

// correctly rounded square root: the compiler's f64 sqrt expansion (faithful, <= 1 ulp) followed by one
// exact-residual decision, so that the result is the IEEE value the CPU checker's sqrt() returns.
//   r = a - g*g (sign exact via fma).  If r != 0 the root lies between g and its neighbour gn on that side; g is
//   correctly rounded iff the root is on g's side of the midpoint, i.e. a vs ((g+gn)/2)^2 = g*gn + d^2/4.  In units
//   of ulp^2 the quantity a - g*gn is an integer N and d^2/4 < 1, so the test is N > 0 (r > 0) / N <= 0 (r < 0).
// Branch-free; valid for finite a > 0 (the only arguments the hot path produces); other inputs pass through.
__device__ __forceinline__ double dsqrt(double a) {
  const double g = __builtin_sqrt(a);
  const double r = __builtin_fma(-g, g, a);
  const double gn = __longlong_as_double(__double_as_longlong(g) + (r > 0.0 ? 1 : -1));
  const double t = __builtin_fma(-g, gn, a);
  const bool move = (r > 0.0) ? (t > 0.0) : ((r < 0.0) ? (t <= 0.0) : false);
  const bool ok = (a > 0.0) && (a < __builtin_inf());
  return (ok && move) ? gn : g;
}

// ------------------------------------------------------------------------------------------------
// Box-Muller on two 32-bit draws; replaces newran's table-rejection Normal (newran2.cxx:164-217)
// behind gaussian_dist_product::drawSample (probability_function.cc:37-47).
//
// The radius is the hot half: 32 normals per chain and step.  -2 ln(u) for u = (k+0.5)/2^32 comes from a 256-entry
// table of {fl(1/c_i), A_i} (tools/gen_tables.py) and a degree-6 log1p polynomial -- no division:
//     x = k + 0.5 = 2^E m,  i = top 8 mantissa bits,  t = fma(m, 1/c_i, -1)  (|t| <= 2^-9, exact for i = 255),
//     -2 ln(u) = [(E-32+adj_i)(-2 ln2_hi) + A_i] + [(E-32+adj_i)(-2 ln2_lo) + (t^2 q(t) - 2t)],
//     q(t) = 1 - (2/3)t + (1/2)t^2 - (2/5)t^3 + (1/3)t^4         (truncation < 2^-56 relative).
// Its square root needs none of the generic sqrt's range scaling (the argument lies in [2^-32, 46]) and is the
// v_rsq_f64 seed + one coupled Newton step + one residual correction; tests/test_gpu_parity.py scans all 2^32
// arguments on the GPU to prove it equal to the correctly rounded dsqrt() -- the CPU checker calls sqrt().
// ------------------------------------------------------------------------------------------------
#include "ptm_tables.inc"
// One table image, staged whole into LDS by the sweep kernels: [0, 512) the radius table, [512, 2560) the angle table
// {sin a_i, cos a_i}, a_i = (i + 0.5) pi / 1024.
constexpr int BM_TABLE_DOUBLES = 512 + 2048;
__device__ __attribute__((aligned(16))) const double BM_TABLE[BM_TABLE_DOUBLES] = {PTM_BMTAB_VALUES, PTM_TRIGTAB_VALUES};

typedef double bm_d2 __attribute__((ext_vector_type(2)));

template <class Tab>   // Tab: pointer to 256 {rc, A} pairs, 16-byte aligned (LDS or global)
__device__ __forceinline__ double bm_neg2log(uint32_t k, Tab tab) {
  const double M2LN2_HI = -2.0 * 6.93147180369123816490e-01, M2LN2_LO = -2.0 * 1.90821492927058770002e-10;
  const double x = (double)k + 0.5;
  const uint64_t b = (uint64_t)__double_as_longlong(x);
  const uint32_t hi = (uint32_t)(b >> 32);
  const uint32_t idx = (hi >> 12) & 255u;
  const int e = (int)(hi >> 20) - (1023 + 32) + (idx >= (uint32_t)PTM_BMTAB_SPLIT ? 1 : 0);
  const double m = __longlong_as_double((long long)((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull));
  const bm_d2 ra = *reinterpret_cast<const bm_d2*>(tab + 2 * idx);
  const double t = __builtin_fma(m, ra.x, -1.0);
  double q = 1.0 / 3.0;
  q = __builtin_fma(q, t, -0.4);
  q = __builtin_fma(q, t, 0.5);
  q = __builtin_fma(q, t, -2.0 / 3.0);
  q = __builtin_fma(q, t, 1.0);
  const double l = __builtin_fma(t * t, q, -2.0 * t);
  const double dk = (double)e;
  return __builtin_fma(dk, M2LN2_HI, ra.y) + __builtin_fma(dk, M2LN2_LO, l);
}

// sqrt for a in [2^-32, 64): no scaling, no special cases
__device__ __forceinline__ double bm_sqrt(double a) {
  const double y = __builtin_amdgcn_rsq(a);
  double g = a * y;
  double h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  const double d = __builtin_fma(-g, g, a);
  return __builtin_fma(d, h, g);   // (a second residual step changes none of the 2^32 results: scanned)
}

// The angle theta = 2 pi (k2 + 0.5) / 2^32: bit 31 of k2 is the half turn (a sign flip of the radius), bits 30..21
// pick the table interval, and the low 21 bits give delta = theta - a_i, |delta| <= pi/2048, added with
//   sin(a+d) = S + (S (cos d - 1) + C sin d),   cos(a+d) = C + (C (cos d - 1) - S sin d),
//   sin d = d + d^3 (-1/6 + d^2/120),  cos d - 1 = d^2 (-1/2 + d^2/24)        (truncation < 2^-65).
template <class Tab>   // Tab: pointer to the BM_TABLE image (LDS or global)
__device__ __forceinline__ void boxmuller_finish(double r, uint32_t k2, Tab tab, double& z0, double& z1) {
  const uint32_t idx = (k2 >> 21) & 1023u;
  const bm_d2 sc = *reinterpret_cast<const bm_d2*>(tab + 512 + 2 * idx);
  // 2 pi / 2^32 x (i + 0.5): i + 0.5 is exact, so fma(i, c, c / 2) rounds the same real number as the checker's (i + 0.5) * c
  const double d = __builtin_fma((double)((int)(k2 & 0x1FFFFFu) - (1 << 20)), 1.4629180792671596e-09, 0.5 * 1.4629180792671596e-09);
  const double d2 = d * d;
  const double sd = __builtin_fma(d * d2, __builtin_fma(d2, 1.0 / 120.0, -1.0 / 6.0), d);
  const double cm1 = d2 * __builtin_fma(d2, 1.0 / 24.0, -0.5);
  const double sn = __builtin_fma(sc.y, sd, sc.x * cm1) + sc.x;
  const double cs = __builtin_fma(-sc.x, sd, sc.y * cm1) + sc.y;
  r = __longlong_as_double(__double_as_longlong(r) ^ ((long long)(k2 >> 31) << 63));
  z0 = r * cs;
  z1 = r * sn;
}
// (the two halves apart, for a kernel that places them itself: radius = bm_sqrt(bm_neg2log(k1)), then boxmuller_finish)
template <typename Tab>
__device__ __forceinline__ void boxmuller(uint32_t k1, uint32_t k2, Tab tab, double& z0, double& z1) {
  boxmuller_finish(bm_sqrt(bm_neg2log(k1, tab)), k2, tab, z0, z1);
}

// fmod restated for the boundary wrap (states.cc:24,39): exact for |x/w| < 2^52
__device__ __forceinline__ double fmod_det(double x, double w) {
  const double q = __builtin_trunc(x / w);
  double r = __builtin_fma(-q, w, x);
  if (x >= 0.0) { if (r < 0.0) r += w; else if (r >= w) r -= w; }
  else          { if (r > 0.0) r -= w; else if (r <= -w) r += w; }
  return r;
}

}  // namespace ptm
