// probe: operand layout and rounding behaviour of v_mfma_f64_16x16x4_f64 on gfx950.
//  (1) layout: integer-valued A, B -> exact D, compared with the documented maps
//  (2) rounding: wide-dynamic-range data; which CPU model reproduces D bit for bit?
//      m0: c -> fma(a0,b0,.) -> fma(a1,b1,.) -> fma(a2,b2,.) -> fma(a3,b3,.)     (k ascending, chained on C)
//      m1: same, k descending            m2: ((a0b0+a1b1)+(a2b2+a3b3)) + c with fma pairs       m3: products rounded, then added k ascending
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, const double* C, double* D) {
  // A[16][4], B[4][16], C[16][16] row-major
  const int l = threadIdx.x;
  const double a = A[(l & 15) * 4 + (l >> 4)];
  const double b = B[(l >> 4) * 16 + (l & 15)];
  d4 c;
  for (int r = 0; r < 4; ++r) c[r] = C[((l >> 4) + 4 * r) * 16 + (l & 15)];
  d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = d[r];
}
// two chained instructions: does the second see exactly the rounded D of the first? (it must: D is a register)
int main() {
  double *dA, *dB, *dC, *dD;
  CHK(hipMalloc(&dA, 64 * 8)); CHK(hipMalloc(&dB, 64 * 8)); CHK(hipMalloc(&dC, 256 * 8)); CHK(hipMalloc(&dD, 256 * 8));
  std::vector<double> A(64), B(64), C(256), D(256);
  // (1) layout
  for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 4; ++kk) A[i * 4 + kk] = 1 + i + 100 * kk;
  for (int kk = 0; kk < 4; ++kk) for (int j = 0; j < 16; ++j) B[kk * 16 + j] = 3 + 7 * j + 1000 * kk;
  for (int i = 0; i < 256; ++i) C[i] = i;
  CHK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CHK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
  CHK(hipMemcpy(dC, C.data(), 2048, hipMemcpyHostToDevice));
  k<<<1, 64>>>(dA, dB, dC, dD);
  CHK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    double w = C[i * 16 + j];
    for (int kk = 0; kk < 4; ++kk) w += A[i * 4 + kk] * B[kk * 16 + j];
    if (w != D[i * 16 + j]) bad++;
  }
  printf("layout: %d wrong of 256\n", bad);
  // (2) rounding
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> U(-1, 1);
  long miss[5] = {0, 0, 0, 0, 0}, total = 0;
  for (int trial = 0; trial < 400; ++trial) {
    const int spread = trial % 4 == 0 ? 0 : (trial % 4 == 1 ? 20 : (trial % 4 == 2 ? 60 : 300));
    auto val = [&]() { return U(rng) * std::ldexp(1.0, spread ? (int)(rng() % (2 * spread)) - spread : 0); };
    for (auto& v : A) v = val();
    for (auto& v : B) v = val();
    for (auto& v : C) v = (trial % 8 < 4) ? val() : 0.0;
    if (trial % 16 == 15) { for (int i = 0; i < 64; i += 3) A[i] = -A[(i + 1) % 64]; }   // cancellations
    CHK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CHK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dC, C.data(), 2048, hipMemcpyHostToDevice));
    k<<<1, 64>>>(dA, dB, dC, dD);
    CHK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost));
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      const double c = C[i * 16 + j];
      double a[4], b[4];
      for (int kk = 0; kk < 4; ++kk) { a[kk] = A[i * 4 + kk]; b[kk] = B[kk * 16 + j]; }
      double m0 = c; for (int kk = 0; kk < 4; ++kk) m0 = std::fma(a[kk], b[kk], m0);
      double m1 = c; for (int kk = 3; kk >= 0; --kk) m1 = std::fma(a[kk], b[kk], m1);
      double m2 = std::fma(a[0], b[0], a[1] * b[1]) + std::fma(a[2], b[2], a[3] * b[3]) + c;
      double m3 = c; for (int kk = 0; kk < 4; ++kk) m3 += a[kk] * b[kk];
      // m4: exact sum rounded once (long double has 64 bits: not exact in general, indicative only)
      long double e = c; for (int kk = 0; kk < 4; ++kk) e += (long double)a[kk] * b[kk];
      const double m4 = (double)e;
      const double d = D[i * 16 + j];
      const double m[5] = {m0, m1, m2, m3, m4};
      for (int q = 0; q < 5; ++q) if (std::memcmp(&m[q], &d, 8)) miss[q]++;
      total++;
    }
  }
  printf("rounding: of %ld results, mismatches  m0(k asc fma chain)=%ld  m1(k desc)=%ld  m2(pairwise)=%ld  m3(rounded products)=%ld  m4(long double)=%ld\n",
         total, miss[0], miss[1], miss[2], miss[3], miss[4]);
  return 0;
}
