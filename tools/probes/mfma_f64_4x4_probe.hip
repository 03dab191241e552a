// probe: v_mfma_f64_4x4x4_4b_f64 on gfx950 -- operand layout, rounding model and issue cost beside v_mfma_f64_16x16x4_f64.
//  (1) layout: one-hot A lane x one-hot B lane -> which D lanes light up; fitted to lane = 16*block + 4*x + y forms
//  (2) rounding: k-ascending fma chain on C (the model the 16x16x4 shape follows)?
//  (3) cost: cycles per instruction, independent accumulators and one dependent chain, one wave and four waves per SIMD
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void onehot(unsigned long long* out) {
  const int l = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = l == la ? 1.0 : 0.0, b = l == lb ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      const unsigned long long m = __ballot(d != 0.0);
      if (l == 0) out[la * 64 + lb] = m;
    }
}
__global__ void one(const double* A, const double* B, const double* C, double* D) {
  const int l = threadIdx.x;
  D[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], C[l], 0, 0, 0);
}
template <int SHAPE, int DEP>
__global__ void cost(double* out, long long* cyc, int iters) {
  const int l = threadIdx.x & 63;
  const double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
  double s4[8];
  d4 s16[8];
  for (int i = 0; i < 8; ++i) { s4[i] = i; s16[i] = d4{(double)i, 0, 0, 0}; }
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = DEP ? 0 : i;
      if (SHAPE == 4) s4[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s4[j], 0, 0, 0);
      else s16[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, s16[j], 0, 0, 0);
    }
  }
  const long long t1 = clock64();
  double r = 0;
  for (int i = 0; i < 8; ++i) r += SHAPE == 4 ? s4[i] : s16[i][0] + s16[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// mixed roles on one SIMD: waves with (wave_id / 4) % 2 == 0 issue MFMAs, the others a chain-free stream of v_fma_f64 (role mask)
template <int SHAPE>
__global__ void mixed(double* out, long long* cyc, int iters, int valu_mask) {
  const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;      // waves 0-3 land on SIMDs 0-3, waves 4-7 again on 0-3 ...
  const bool valu = (valu_mask >> (wv >> 2)) & 1;
  const double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
  double s4[8];
  d4 s16[8];
  for (int i = 0; i < 8; ++i) { s4[i] = i; s16[i] = d4{(double)i, 0, 0, 0}; }
  const long long t0 = clock64();
  if (valu) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) s4[i] = __builtin_fma(s4[i], a, b);
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (SHAPE == 4) s4[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s4[i], 0, 0, 0);
        else s16[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, s16[i], 0, 0, 0);
      }
    }
  }
  const long long t1 = clock64();
  double r = 0;
  for (int i = 0; i < 8; ++i) r += s4[i] + s16[i][0] + s16[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (l == 0 && blockIdx.x == 0) cyc[wv] = t1 - t0;
}

int main() {
  // (1) layout
  unsigned long long* dm;
  CHK(hipMalloc(&dm, 4096 * 8));
  onehot<<<1, 64>>>(dm);
  std::vector<unsigned long long> M(4096);
  CHK(hipMemcpy(M.data(), dm, 4096 * 8, hipMemcpyDeviceToHost));
  // candidate maps: A lane (b,i,k), B lane (b,k,n), D lane (b,i,n), each "16b+4x+y" with (x,y) in either order
  for (int fa = 0; fa < 2; ++fa) for (int fb = 0; fb < 2; ++fb) for (int fd = 0; fd < 2; ++fd) {
    int bad = 0;
    for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) {
      const int ba = la >> 4, bb = lb >> 4;
      const int ia = fa ? (la >> 2) & 3 : la & 3, ka = fa ? la & 3 : (la >> 2) & 3;
      const int kb = fb ? (lb >> 2) & 3 : lb & 3, nb = fb ? lb & 3 : (lb >> 2) & 3;
      unsigned long long want = 0;
      if (ba == bb && ka == kb) want = 1ull << (16 * ba + (fd ? 4 * ia + nb : 4 * nb + ia));
      if (want != M[la * 64 + lb]) bad++;
    }
    printf("layout A:%s B:%s D:%s -> %d wrong of 4096\n", fa ? "16b+4i+k" : "16b+4k+i", fb ? "16b+4k+n" : "16b+4n+k",
           fd ? "16b+4i+n" : "16b+4n+i", bad);
  }
  printf("sample: la=1 lb=1 -> %016llx   la=4 lb=4 -> %016llx  la=1 lb=4 -> %016llx  la=4 lb=1 -> %016llx  la=17 lb=17 -> %016llx\n",
         M[1 * 64 + 1], M[4 * 64 + 4], M[1 * 64 + 4], M[4 * 64 + 1], M[17 * 64 + 17]);
  if (getenv("PROBE_DUMP"))
    for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) if (M[la * 64 + lb]) {
      printf("pair %d %d ->", la, lb);
      for (int ld = 0; ld < 64; ++ld) if (M[la * 64 + lb] >> ld & 1) printf(" %d", ld);
      printf("\n");
    }
  if (getenv("PROBE_LAYOUT_ONLY")) return 0;
  // (2) rounding, with the layout found generically from the one-hot table: for D lane ld, its contributing (la, lb) pairs
  double *dA, *dB, *dC, *dD;
  CHK(hipMalloc(&dA, 512)); CHK(hipMalloc(&dB, 512)); CHK(hipMalloc(&dC, 512)); CHK(hipMalloc(&dD, 512));
  std::vector<std::vector<std::pair<int, int>>> src(64);
  for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb)
    for (int ld = 0; ld < 64; ++ld) if (M[la * 64 + lb] >> ld & 1) src[ld].push_back({la, lb});
  int four = 0;
  for (int ld = 0; ld < 64; ++ld) four += src[ld].size() == 4;
  printf("D lanes with exactly four contributing pairs: %d of 64\n", four);
  std::mt19937_64 rng(11);
  std::uniform_real_distribution<double> U(-1, 1);
  long miss[3] = {0, 0, 0}, total = 0;
  std::vector<double> A(64), B(64), C(64), D(64);
  for (int trial = 0; trial < 1600; ++trial) {
    const int spread = trial % 4 == 0 ? 0 : (trial % 4 == 1 ? 20 : (trial % 4 == 2 ? 60 : 300));
    auto val = [&]() { return U(rng) * std::ldexp(1.0, spread ? (int)(rng() % (2 * spread)) - spread : 0); };
    for (auto& v : A) v = val();
    for (auto& v : B) v = val();
    for (auto& v : C) v = (trial % 8 < 4) ? val() : 0.0;
    CHK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CHK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dC, C.data(), 512, hipMemcpyHostToDevice));
    one<<<1, 64>>>(dA, dB, dC, dD);
    CHK(hipMemcpy(D.data(), dD, 512, hipMemcpyDeviceToHost));
    for (int ld = 0; ld < 64; ++ld) {
      if (src[ld].size() != 4) continue;
      // pairs come out ordered by la then lb; order them by the A lane's k under either A map: try ascending / descending pair order
      double m0 = C[ld], m1 = C[ld], m3 = C[ld];
      for (int q = 0; q < 4; ++q) m0 = std::fma(A[src[ld][q].first], B[src[ld][q].second], m0);
      for (int q = 3; q >= 0; --q) m1 = std::fma(A[src[ld][q].first], B[src[ld][q].second], m1);
      for (int q = 0; q < 4; ++q) m3 += A[src[ld][q].first] * B[src[ld][q].second];
      const double m[3] = {m0, m1, m3};
      for (int q = 0; q < 3; ++q) if (std::memcmp(&m[q], &D[ld], 8)) miss[q]++;
      total++;
    }
  }
  printf("rounding: of %ld results, mismatches  m0(fma chain, pairs ascending)=%ld  m1(descending)=%ld  m3(rounded products)=%ld\n", total,
         miss[0], miss[1], miss[2]);
  // (3) cost
  double* dout;
  long long* dc;
  CHK(hipMalloc(&dout, 8 << 20)); CHK(hipMalloc(&dc, 8));   // 1 M doubles: the largest launch below writes 512 x 1024
  const int iters = 2000;
  long long c = 0;
#define RUN(S, DEPC, BLK, THR, label)                                                         \
  cost<S, DEPC><<<BLK, THR>>>(dout, dc, iters); CHK(hipDeviceSynchronize());                      \
  cost<S, DEPC><<<BLK, THR>>>(dout, dc, iters); CHK(hipDeviceSynchronize());                      \
  CHK(hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost));                                               \
  printf("%-44s %8.2f clock64 ticks per instruction (wave 0)\n", label, (double)c / (iters * 8.0));
  RUN(16, 0, 1, 64, "16x16x4 independent, 1 wave");
  RUN(16, 1, 1, 64, "16x16x4 dependent chain, 1 wave");
  RUN(4, 0, 1, 64, "4x4x4_4b independent, 1 wave");
  RUN(4, 1, 1, 64, "4x4x4_4b dependent chain, 1 wave");
  RUN(16, 0, 1, 256, "16x16x4 independent, 1 wave/SIMD (4 waves)");
  RUN(4, 0, 1, 256, "4x4x4_4b independent, 1 wave/SIMD (4 waves)");
  RUN(16, 0, 1, 512, "16x16x4 independent, 2 waves/SIMD");
  RUN(4, 0, 1, 512, "4x4x4_4b independent, 2 waves/SIMD");
  RUN(4, 1, 1, 512, "4x4x4_4b dependent, 2 waves/SIMD");
  RUN(4, 1, 1, 1024, "4x4x4_4b dependent, 4 waves/SIMD");
  // whole-chip throughput by the host's clock
  {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int its = 20000;
#define CHIP(S, THR, BLKS, label)                                                                          \
    cost<S, 0><<<BLKS, THR>>>(dout, dc, 100); CHK(hipDeviceSynchronize());                                     \
    CHK(hipEventRecord(e0)); cost<S, 0><<<BLKS, THR>>>(dout, dc, its); CHK(hipEventRecord(e1));                \
    CHK(hipEventSynchronize(e1));                                                                              \
    { float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); CHK(hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost));        \
      const double fl = 2.0 * (S == 4 ? 256.0 : 1024.0) * 8 * its * (THR / 64) * (double)BLKS;                \
      printf("%-40s %7.3f ms  %7.2f TFLOP/s   wave0: %lld ticks = %.2f per instr, %.1f ticks/us\n", label, ms, fl / ms * 1e-9, c, \
             (double)c / (its * 8.0), c / (ms * 1e3)); }
    CHIP(16, 256, 256, "16x16x4 256 blocks x 4 waves");
    CHIP(16, 512, 256, "16x16x4 256 blocks x 8 waves");
    CHIP(16, 1024, 256, "16x16x4 256 blocks x 16 waves");
    CHIP(16, 1024, 512, "16x16x4 512 blocks x 16 waves");
    CHIP(4, 256, 256, "4x4x4 256 blocks x 4 waves");
    CHIP(4, 512, 256, "4x4x4 256 blocks x 8 waves");
    CHIP(4, 1024, 256, "4x4x4 256 blocks x 16 waves");
    CHIP(4, 1024, 512, "4x4x4 512 blocks x 16 waves");
  }
  // mixed roles
  {
    long long* dcy;
    CHK(hipMalloc(&dcy, 16 * 8));
    long long cy[16];
#define MIX(S, THR, MASK, label)                                                                   \
    mixed<S><<<1, THR>>>(dout, dcy, 4000, MASK); CHK(hipDeviceSynchronize());                          \
    mixed<S><<<1, THR>>>(dout, dcy, 4000, MASK); CHK(hipDeviceSynchronize());                          \
    CHK(hipMemcpy(cy, dcy, 16 * 8, hipMemcpyDeviceToHost));                                             \
    printf("%-52s", label);                                                                             \
    for (int w = 0; w < THR / 64; w += 4) printf("  wave %d (%s): %.2f/instr", w, (MASK >> (w >> 2)) & 1 ? "fma" : "mfma", \
      cy[w] / (4000.0 * (((MASK >> (w >> 2)) & 1) ? 32 : 8)));                                              \
    printf("\n");
    MIX(16, 256, 1, "v_fma_f64 alone, 1 wave/SIMD");
    MIX(16, 512, 3, "v_fma_f64 x2 waves/SIMD");
    MIX(16, 512, 2, "16x16x4 wave + fma wave per SIMD");
    MIX(4, 512, 2, "4x4x4 wave + fma wave per SIMD");
    MIX(16, 1024, 12, "2 x 16x16x4 waves + 2 fma waves per SIMD");
    MIX(16, 1024, 14, "1 x 16x16x4 wave + 3 fma waves per SIMD");
    MIX(16, 1024, 0, "4 x 16x16x4 waves per SIMD");
    MIX(4, 1024, 0, "4 x 4x4x4 waves per SIMD");
  }
  return 0;
}
