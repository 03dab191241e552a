// probe: what a per-step JOIN of two streams costs on gfx950 -- per step, stream A and stream B each run one short kernel
// (~5 us of dependent work) and then wait for the other's kernel through events (hipEventRecord / hipStreamWaitEvent), against
// the same two kernels back to back on ONE stream.  Tells whether a latency-bound step can be split over two streams.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void spin(double* p, int iters) {   // a chain of dependent fmas: iters * ~8 cycles on one wave
  double a = p[threadIdx.x];
  for (int i = 0; i < iters; ++i) a = __builtin_fma(a, 1.0000001, 1e-9);
  p[threadIdx.x] = a;
}

int main() {
  double *a, *b;
  CHK(hipMalloc(&a, 64 * 8)); CHK(hipMalloc(&b, 64 * 8));
  CHK(hipMemset(a, 0, 64 * 8)); CHK(hipMemset(b, 0, 64 * 8));
  hipStream_t sa, sb;
  CHK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  const int K = 2000, iters = 1500;   // ~5 us per kernel
  hipEvent_t ea[2], eb[2];
  for (int i = 0; i < 2; ++i) { CHK(hipEventCreateWithFlags(&ea[i], hipEventDisableTiming)); CHK(hipEventCreateWithFlags(&eb[i], hipEventDisableTiming)); }
  auto now = [] { return std::chrono::steady_clock::now(); };
  for (int rep = 0; rep < 2; ++rep) {
    // one stream, two kernels per step
    CHK(hipDeviceSynchronize());
    auto t0 = now();
    for (int k = 0; k < K; ++k) { spin<<<1, 64, 0, sa>>>(a, iters); spin<<<1, 64, 0, sa>>>(b, iters); }
    CHK(hipDeviceSynchronize());
    const double one = std::chrono::duration<double, std::micro>(now() - t0).count() / K;
    // one stream, one kernel per step (what a perfect overlap would give)
    t0 = now();
    for (int k = 0; k < K; ++k) spin<<<1, 64, 0, sa>>>(a, iters);
    CHK(hipDeviceSynchronize());
    const double half = std::chrono::duration<double, std::micro>(now() - t0).count() / K;
    // two streams, joined after every step
    t0 = now();
    for (int k = 0; k < K; ++k) {
      spin<<<1, 64, 0, sa>>>(a, iters);
      spin<<<1, 64, 0, sb>>>(b, iters);
      CHK(hipEventRecord(ea[k & 1], sa)); CHK(hipEventRecord(eb[k & 1], sb));
      CHK(hipStreamWaitEvent(sa, eb[k & 1], 0)); CHK(hipStreamWaitEvent(sb, ea[k & 1], 0));
    }
    CHK(hipDeviceSynchronize());
    const double two = std::chrono::duration<double, std::micro>(now() - t0).count() / K;
    printf("per step: two kernels on one stream %.2f us; one kernel alone %.2f us; one kernel on each of two streams + join %.2f us\n", one, half, two);
  }
  return 0;
}
