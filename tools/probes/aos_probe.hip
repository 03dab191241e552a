// probe: is a per-lane 256-B row (AoS, 16 x 16-B loads per lane, lanes 256 B apart) read at HBM speed on MI355X?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int DP = 32;
__global__ __launch_bounds__(256) void soa_read(const double* __restrict__ x, double* __restrict__ out, int n) {
  int c = blockIdx.x * 256 + threadIdx.x; if (c >= n) return;
  double s = 0;
#pragma unroll
  for (int d = 0; d < DP; ++d) s += x[(size_t)d * n + c];
  out[c] = s;
}
__global__ __launch_bounds__(256) void aos_read(const double2* __restrict__ x, double* __restrict__ out, int n) {
  int c = blockIdx.x * 256 + threadIdx.x; if (c >= n) return;
  const double2* r = x + (size_t)c * (DP / 2);
  double s = 0;
#pragma unroll
  for (int k = 0; k < DP / 2; ++k) { double2 v = r[k]; s += v.x + v.y; }
  out[c] = s;
}
// read every row, write back 1 row in 5 (the accepted ones), in place
__global__ __launch_bounds__(256) void aos_rw(double2* __restrict__ x, double* __restrict__ out, int n) {
  int c = blockIdx.x * 256 + threadIdx.x; if (c >= n) return;
  double2* r = x + (size_t)c * (DP / 2);
  double2 v[DP / 2]; double s = 0;
#pragma unroll
  for (int k = 0; k < DP / 2; ++k) { v[k] = r[k]; s += v[k].x + v[k].y; }
  if ((c * 2654435761u >> 16) % 5 == 0) {
#pragma unroll
    for (int k = 0; k < DP / 2; ++k) { v[k].x += 1.0; r[k] = v[k]; }
  }
  out[c] = s;
}
__global__ __launch_bounds__(256) void soa_rw(const double* __restrict__ x, double* __restrict__ y, double* __restrict__ out, int n) {
  int c = blockIdx.x * 256 + threadIdx.x; if (c >= n) return;
  double s = 0;
#pragma unroll
  for (int d = 0; d < DP; ++d) { double v = x[(size_t)d * n + c]; s += v; y[(size_t)d * n + c] = v + 1.0; }
  out[c] = s;
}
int main() {
  const int n = 1024 * 4096;
  double *x, *y, *out;
  CHK(hipMalloc(&x, (size_t)n * DP * 8)); CHK(hipMalloc(&y, (size_t)n * DP * 8)); CHK(hipMalloc(&out, (size_t)n * 8));
  CHK(hipMemset(x, 0, (size_t)n * DP * 8)); CHK(hipMemset(y, 0, (size_t)n * DP * 8));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto time = [&](const char* name, auto launch, double bytes) {
    for (int i = 0; i < 3; i++) launch();
    hipEventRecord(a);
    for (int i = 0; i < 20; i++) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 20;
    printf("%-10s %.4f ms  %.0f GB/s\n", name, ms, bytes / (ms * 1e-3) / 1e9);
  };
  dim3 g((n + 255) / 256), bl(256);
  time("soa_read", [&] { hipLaunchKernelGGL(soa_read, g, bl, 0, 0, x, out, n); }, (double)n * (DP * 8 + 8));
  time("aos_read", [&] { hipLaunchKernelGGL(aos_read, g, bl, 0, 0, (const double2*)x, out, n); }, (double)n * (DP * 8 + 8));
  time("soa_rw", [&] { hipLaunchKernelGGL(soa_rw, g, bl, 0, 0, x, y, out, n); }, (double)n * (2 * DP * 8 + 8));
  time("aos_rw20%", [&] { hipLaunchKernelGGL(aos_rw, g, bl, 0, 0, (double2*)x, out, n); }, (double)n * (1.2 * DP * 8 + 8));
  return 0;
}
