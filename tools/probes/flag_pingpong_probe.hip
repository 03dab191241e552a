// flag_pingpong_probe.hip -- what does it cost two RESIDENT workgroups to hand each other a row through memory?
// (the persistent ladder kernel's neighbour hand-over, ptm_ladder_kernel.hpp).  Workgroup 0 and workgroup `peer` of one launch
// play ping-pong: write a 256-byte payload, raise a flag, wait for the other's flag, read its payload.  Measured per round trip
// (two hand-overs), for: peers on the same XCD (workgroup ids equal mod 8) or on different XCDs; ordinary device memory with
// release / acquire atomics at agent scope (the C++ memory model's way: L2 write-back + invalidate around every hand-over),
// or relaxed atomics with sc1 accesses on ordinary / fine-grained / uncached memory and a bare s_waitcnt before the flag.
// Also: the price of one __threadfence().
//   hipcc --offload-arch=gfx950 -O2 -o flag_pingpong_probe flag_pingpong_probe.hip && ./flag_pingpong_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>   // 0: release / acquire + plain payload; 1: relaxed agent-scope atomics for flag and payload, s_waitcnt before the flag
__global__ __launch_bounds__(256) void pingpong(double* pay, int* flags, int peer, int rounds, long long* ticks, int* bad) {
  const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == peer ? 1 : -1);
  if (me < 0) return;
  const int tid = threadIdx.x;
  double* mine = pay + me * 64;
  double* theirs = pay + (1 - me) * 64;
  long long t0 = 0;
  int wrong = 0;
  for (int i = 1; i <= rounds; ++i) {
    if (i == 11 && tid == 0) t0 = wall_clock64();
    if (me == 1) {   // wait for the ping first
      if (tid == 0) {
        if (MODE == 0) while (__hip_atomic_load(&flags[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < i) {}
        else while (__hip_atomic_load(&flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < i) {}
      }
      __syncthreads();
      if (tid < 32) {
        const double v = MODE == 0 ? theirs[tid] : __hip_atomic_load(&theirs[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v != (double)(i * 1000 + tid)) wrong++;
      }
    }
    if (tid < 32) {
      const double v = (double)(i * 1000 + tid + (me ? 500 : 0));
      if (MODE == 0) mine[tid] = v; else __hip_atomic_store(&mine[tid], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (MODE == 0) __syncthreads();
    else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
    if (tid == 0) {
      if (MODE == 0) __hip_atomic_store(&flags[me], i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_store(&flags[me], i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (me == 0) {   // wait for the pong
      if (tid == 0) {
        if (MODE == 0) while (__hip_atomic_load(&flags[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < i) {}
        else while (__hip_atomic_load(&flags[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < i) {}
      }
      __syncthreads();
      if (tid < 32) {
        const double v = MODE == 0 ? theirs[tid] : __hip_atomic_load(&theirs[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v != (double)(i * 1000 + tid + 500)) wrong++;
      }
    }
  }
  if (me == 0 && tid == 0) ticks[0] = wall_clock64() - t0;
  if (wrong) atomicAdd(bad, wrong);
}

__global__ void fence_cost(int n, long long* ticks, double* sink) {
  const long long t0 = wall_clock64();
  for (int i = 0; i < n; ++i) { sink[threadIdx.x] = (double)i; __threadfence(); }
  if (threadIdx.x == 0) ticks[0] = wall_clock64() - t0;
}

int main() {
  const int rounds = 2010;
  long long* ticks; int* bad;
  CHK(hipHostMalloc((void**)&ticks, 64, 0));
  CHK(hipHostMalloc((void**)&bad, 64, 0));
  struct memkind { const char* name; int kind; };
  const memkind kinds[] = {{"ordinary", 0}, {"fine-grained", 1}, {"uncached", 2}};
  for (const memkind& mk : kinds) {
    void* buf = nullptr;
    hipError_t e = mk.kind == 0 ? hipMalloc(&buf, 4096) : hipExtMallocWithFlags(&buf, 4096, mk.kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
    if (e != hipSuccess) { printf("%-13s allocation failed: %s\n", mk.name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
    double* pay = (double*)buf;
    int* flags = (int*)((char*)buf + 2048);
    for (int peer : {8, 1}) {
      for (int mode = 0; mode < 2; ++mode) {
        CHK(hipMemset(buf, 0, 4096));
        *bad = 0; *ticks = 0;
        if (mode == 0) hipLaunchKernelGGL(pingpong<0>, dim3(16), dim3(256), 0, 0, pay, flags, peer, rounds, ticks, bad);
        else hipLaunchKernelGGL(pingpong<1>, dim3(16), dim3(256), 0, 0, pay, flags, peer, rounds, ticks, bad);
        CHK(hipDeviceSynchronize());
        printf("%-13s %-22s %-28s %7.3f us per round trip   payload errors %d\n", mk.name, peer == 8 ? "same XCD (wg 0, 8)" : "two XCDs (wg 0, 1)",
               mode == 0 ? "release/acquire + plain data" : "relaxed sc1 + s_waitcnt", *ticks * 0.01 / (rounds - 10), *bad);
      }
    }
    CHK(hipFree(buf));
  }
  double* sink;
  CHK(hipMalloc((void**)&sink, 4096));
  hipLaunchKernelGGL(fence_cost, dim3(1), dim3(64), 0, 0, 1000, ticks, sink);
  CHK(hipDeviceSynchronize());
  printf("__threadfence() after a store: %.3f us each (one wave)\n", *ticks * 0.01 / 1000);
  return 0;
}
