// probe: what a plain device-to-device copy kernel reaches on this MI355X (GB/s of bytes read + written), by launch shape --
// the figure ptm_calibrate should report as the box's copy ceiling (MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define uint4 u4
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int U, bool NT>
__global__ __launch_bounds__(256) void copyk(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n16; i += U * stride) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride); else dst[i + u * stride] = v[u]; }
  }
  for (; i < n16; i += stride) dst[i] = src[i];
}
// each block owns a contiguous chunk (the rows of a streaming kernel), 4 loads in flight per lane
__global__ __launch_bounds__(256) void copy_chunk(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
  const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n16 ? b0 + per : n16;
  for (size_t i = b0 + threadIdx.x; i < b1; i += 1024) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i + 256 * u < b1) v[u] = src[i + 256 * u];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i + 256 * u < b1) dst[i + 256 * u] = v[u];
  }
}

int main() {
  const size_t half = (size_t)2400 << 20, n16 = half / 16;
  void *s, *d;
  CHK(hipMalloc(&s, half)); CHK(hipMalloc(&d, half));
  CHK(hipMemset(s, 0x3c, half)); CHK(hipMemset(d, 0, half));
  hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  auto run = [&](const char* name, int blocks, auto launch) {
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
      hipEventRecord(a); launch(blocks); hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (rep && ms < best) best = ms;
    }
    printf("%-28s blocks %6d  %.3f ms  %.0f GB/s\n", name, blocks, best, 2.0 * half / (best * 1e-3) / 1e9);
    return 0;
  };
  for (int bpc : {4, 8, 16, 32, 64}) {
    run("grid-stride U=1", 256 * bpc, [&](int g) { hipLaunchKernelGGL((copyk<1, false>), dim3(g), dim3(256), 0, 0, (const uint4*)s, (uint4*)d, n16); });
    run("grid-stride U=4", 256 * bpc, [&](int g) { hipLaunchKernelGGL((copyk<4, false>), dim3(g), dim3(256), 0, 0, (const uint4*)s, (uint4*)d, n16); });
    run("grid-stride U=4 nontemporal", 256 * bpc, [&](int g) { hipLaunchKernelGGL((copyk<4, true>), dim3(g), dim3(256), 0, 0, (const uint4*)s, (uint4*)d, n16); });
    run("chunk per block", 256 * bpc, [&](int g) { hipLaunchKernelGGL(copy_chunk, dim3(g), dim3(256), 0, 0, (const uint4*)s, (uint4*)d, n16); });
  }
  run("one block per 4 KB", (int)(n16 / 256), [&](int g) { hipLaunchKernelGGL((copyk<1, false>), dim3(g), dim3(256), 0, 0, (const uint4*)s, (uint4*)d, n16); });
  { float best = 1e30f; for (int rep = 0; rep < 4; ++rep) { hipEventRecord(a); hipMemcpyAsync(d, s, half, hipMemcpyDeviceToDevice, 0); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (rep && ms < best) best = ms; }
    printf("%-28s               %.3f ms  %.0f GB/s\n", "hipMemcpyAsync D2D", best, 2.0 * half / (best * 1e-3) / 1e9); }
  return 0;
}
