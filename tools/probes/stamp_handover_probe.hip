// stamp_handover_probe.hip -- can a hand-over between two resident workgroups do without the flag?
// The persistent ladder kernel (ptm_ladder_kernel.hpp) hands rungs over as: payload stores -> s_waitcnt -> barrier -> flag store;
// the receiver looks at the flag (one round trip through the L2), then fetches the payload (another).  If every payload word carried
// its own step stamp IN THE SAME 16-BYTE STORE, the receiver could fetch speculatively and validate what it got: one round trip.
// That needs 16-byte accesses that are never torn.  This probe measures both and hammers the assumption:
//   (A) flag + payload, relaxed agent-scope (sc1) accesses: the kernel's present scheme;
//   (B) 32 lanes each store {value, stamp} with ONE global_store_dwordx4 sc1; the receiver's 32 lanes poll their own 16 bytes;
//   (C) torn-read hammer: a writer keeps storing {k, k} pairs, a reader keeps loading 16 bytes: halves that differ are counted.
//   hipcc --offload-arch=gfx950 -O2 -o stamp_handover_probe stamp_handover_probe.hip && ./stamp_handover_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double d2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ d2_t load16(const d2_t* p) {
  d2_t v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void store16(d2_t* p, d2_t v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory"); }

template <int MODE>   // 0: flag + payload (two round trips at the receiver), 1: stamped 16-byte payload words (one)
__global__ __launch_bounds__(256) void pingpong(d2_t* pay, int* flags, int peer, int rounds, long long* ticks, int* bad) {
  const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == peer ? 1 : -1);
  if (me < 0) return;
  const int tid = threadIdx.x;
  d2_t* mine = pay + me * 64;
  d2_t* theirs = pay + (1 - me) * 64;
  long long t0 = 0;
  int wrong = 0;
  auto receive = [&](int i, double expect_off) {
    if (MODE == 0) {
      if (tid == 0) while (__hip_atomic_load(&flags[1 - me], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < i) {}
      __syncthreads();
      if (tid < 32) { const d2_t v = load16(&theirs[tid]); if (v.x != (double)(i * 1000 + tid) + expect_off) wrong++; }
    } else {
      if (tid < 32) {
        d2_t v;
        do { v = load16(&theirs[tid]); } while (v.y != (double)i);   // my own word validates itself
        if (v.x != (double)(i * 1000 + tid) + expect_off) wrong++;
      }
      __syncthreads();
    }
  };
  auto send = [&](int i, double off) {
    if (tid < 32) store16(&mine[tid], d2_t{(double)(i * 1000 + tid) + off, (double)i});
    if (MODE == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(&flags[me], i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  for (int i = 1; i <= rounds; ++i) {
    if (i == 11 && tid == 0) t0 = wall_clock64();
    if (me == 1) receive(i, 0.0);
    send(i, me ? 500.0 : 0.0);
    if (me == 0) receive(i, 500.0);
  }
  if (me == 0 && tid == 0) ticks[0] = wall_clock64() - t0;
  if (wrong) atomicAdd(bad, wrong);
}

// torn-read hammer: block 0 writes {k, k}, block `peer` reads; a pair whose halves differ was torn
__global__ __launch_bounds__(64) void hammer(d2_t* word, int peer, int n, int* torn, int* seen) {
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) for (int k = 1; k <= n; ++k) store16(word, d2_t{(double)k, (double)k});
  } else if (blockIdx.x == peer) {
    if (threadIdx.x == 0) {
      int t = 0, distinct = 0;
      double last = 0;
      for (int k = 0; k < n; ++k) {
        const d2_t v = load16(word);
        if (v.x != v.y) t++;
        if (v.x != last) { distinct++; last = v.x; }
      }
      *torn = t; *seen = distinct;
    }
  }
}

int main() {
  const int rounds = 4010;
  long long* ticks; int* bad; int* torn; int* seen;
  CHK(hipHostMalloc((void**)&ticks, 64, 0)); CHK(hipHostMalloc((void**)&bad, 64, 0)); CHK(hipHostMalloc((void**)&torn, 64, 0)); CHK(hipHostMalloc((void**)&seen, 64, 0));
  void* buf = nullptr;
  if (hipExtMallocWithFlags(&buf, 8192, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); CHK(hipMalloc(&buf, 8192)); printf("(ordinary device memory)\n"); }
  d2_t* pay = (d2_t*)buf;
  int* flags = (int*)((char*)buf + 4096);
  for (int peer : {8, 1}) {   // same XCD (ids equal mod 8) / neighbouring XCDs
    for (int mode = 0; mode < 2; ++mode) {
      CHK(hipMemset(buf, 0, 8192)); *bad = 0;
      if (mode == 0) hipLaunchKernelGGL(pingpong<0>, dim3(16), dim3(256), 0, 0, pay, flags, peer, rounds, ticks, bad);
      else hipLaunchKernelGGL(pingpong<1>, dim3(16), dim3(256), 0, 0, pay, flags, peer, rounds, ticks, bad);
      CHK(hipDeviceSynchronize());
      printf("%-26s %-32s %.2f us per hand-over   (%d wrong values)\n", peer == 8 ? "peers on one XCD" : "peers on neighbouring XCDs",
             mode == 0 ? "flag, then payload" : "stamped 16-byte payload words", ticks[0] * 0.01 / (2.0 * (rounds - 10)), *bad);
    }
    CHK(hipMemset(buf, 0, 8192)); *torn = -1; *seen = 0;
    hipLaunchKernelGGL(hammer, dim3(16), dim3(64), 0, 0, pay, peer, 2000000, torn, seen);
    CHK(hipDeviceSynchronize());
    printf("%-26s torn-read hammer: %d torn of 2000000 loads (%d distinct values seen)\n", peer == 8 ? "peers on one XCD" : "peers on neighbouring XCDs", *torn, *seen);
  }
  return 0;
}
