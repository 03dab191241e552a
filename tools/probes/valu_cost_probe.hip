// probe: issue cost (cycles per wave-instruction, per SIMD) of the VALU instruction kinds the sweep kernel is made of, gfx950.
// Each kernel runs 8 independent chains of one instruction; 1 wave per SIMD gives the single-wave issue interval, 4 waves per
// SIMD the pipe's throughput (per-SIMD cycles per instruction = wave 0's cycles / instructions / waves, measured on the whole
// block's span).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define BODY8(stmt) stmt(0) stmt(1) stmt(2) stmt(3) stmt(4) stmt(5) stmt(6) stmt(7)

template <int K>
__global__ void cost(unsigned long long* out, long long* cyc, int iters) {
  const int l = threadIdx.x;
  unsigned int u[8], w[8];
  double d[8], e[8];
  unsigned long long q[8];
  for (int i = 0; i < 8; ++i) { u[i] = l * 2654435761u + i; w[i] = l + 77 * i + 1; d[i] = 1.0 + 1e-3 * (l + i); e[i] = 0.5 + 1e-4 * i; q[i] = l + i; }
  const unsigned int M = 0xD2511F53u;
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (K == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(q[i]) : "v"(u[i]), "s"(M) : "vcc");
      if (K == 1) asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(w[i]) : "v"(u[i]), "s"(M));
      if (K == 2) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(w[i]) : "v"(u[i]), "s"(M));
      if (K == 3) asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(w[i]) : "v"(u[i]), "v"(w[i]));
      if (K == 4) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(w[i]) : "v"(u[i]), "v"(w[i]));
      if (K == 5) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(e[i]), "v"(e[i]));
      if (K == 6) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e[i]));
      if (K == 7) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e[i]));
      if (K == 8) asm volatile("v_rsq_f64 %0, %1" : "=v"(d[i]) : "v"(e[i]));
      if (K == 9) asm volatile("v_rcp_f64 %0, %1" : "=v"(d[i]) : "v"(e[i]));
      if (K == 10) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(u[i]));
      if (K == 11) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(w[i]) : "v"(d[i]));
      if (K == 12) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(w[i]) : "v"(u[i]), "v"(w[i]) : "vcc");
      if (K == 13) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[i]), "v"(e[i]) : "vcc");
      if (K == 14) asm volatile("v_lshlrev_b64 %0, 3, %1" : "=v"(q[i]) : "v"(q[i]));
      if (K == 15) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(w[i]) : "v"(u[i]), "v"(u[i]));
      if (K == 16) asm volatile("v_ldexp_f64 %0, %1, %2" : "=v"(d[i]) : "v"(e[i]), "v"(u[i]));
      if (K == 17) asm volatile("v_add_co_u32 %0, vcc, %1, %2" : "=v"(w[i]) : "v"(u[i]), "v"(w[i]) : "vcc");
      if (K == 18) asm volatile("v_mov_b32 %0, %1" : "=v"(w[i]) : "v"(u[i]));
      if (K == 19) asm volatile("v_div_scale_f64 %0, vcc, %1, %1, %2" : "=v"(d[i]) : "v"(e[i]), "v"(e[i]) : "vcc");
      if (K == 20) asm volatile("v_div_fmas_f64 %0, %1, %2, %3" : "=v"(d[i]) : "v"(e[i]), "v"(e[i]), "v"(e[i]) : "vcc");
      if (K == 21) asm volatile("v_div_fixup_f64 %0, %1, %2, %3" : "=v"(d[i]) : "v"(e[i]), "v"(e[i]), "v"(e[i]));
      if (K == 22) asm volatile("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(w[i]) : "v"(u[i]), "v"(w[i]));
      if (K == 23) asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(w[i]) : "v"(u[i]), "v"(w[i]), "v"(u[i]));
    }
  }
  const long long t1 = clock64();
  unsigned long long r = 0;
  for (int i = 0; i < 8; ++i) r += u[i] + w[i] + q[i] + (unsigned long long)d[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int K>
int run(const char* name, unsigned long long* dout, long long* dc) {
  const int iters = 2000;
  long long c[16];
  printf("%-22s", name);
  for (int waves = 4; waves <= 16; waves *= 2) {   // 1, 2, 4 waves per SIMD
    cost<K><<<1, 64 * waves>>>(dout, dc, iters);
    CHK(hipDeviceSynchronize());
    cost<K><<<1, 64 * waves>>>(dout, dc, iters);
    CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(c, dc, sizeof(long long) * waves, hipMemcpyDeviceToHost));
    long long mx = 0;
    for (int w = 0; w < waves; ++w) mx = c[w] > mx ? c[w] : mx;
    printf("  %d/SIMD: wave0 %6.2f  pipe %6.2f", waves / 4, (double)c[0] / (iters * 8.0), (double)mx / (iters * 8.0 * (waves / 4)));
  }
  printf("\n");
  return 0;
}

int main() {
  unsigned long long* dout;
  long long* dc;
  CHK(hipMalloc(&dout, 1024 * 8));   // one block of at most 1024 threads
  CHK(hipMalloc(&dc, 16 * 8));
  printf("cycles per wave-instruction: one wave's issue interval; `pipe` = slowest wave's span / instructions / waves per SIMD\n");
  run<0>("v_mad_u64_u32", dout, dc);
  run<1>("v_mul_hi_u32", dout, dc);
  run<2>("v_mul_lo_u32", dout, dc);
  run<3>("v_mul_u32_u24", dout, dc);
  run<22>("v_mul_hi_u32_u24", dout, dc);
  run<23>("v_mad_u32_u24", dout, dc);
  run<4>("v_xor_b32", dout, dc);
  run<18>("v_mov_b32", dout, dc);
  run<17>("v_add_co_u32", dout, dc);
  run<12>("v_cndmask_b32", dout, dc);
  run<14>("v_lshlrev_b64", dout, dc);
  run<15>("v_fma_f32", dout, dc);
  run<5>("v_fma_f64", dout, dc);
  run<6>("v_mul_f64", dout, dc);
  run<7>("v_add_f64", dout, dc);
  run<13>("v_cmp_lt_f64", dout, dc);
  run<16>("v_ldexp_f64", dout, dc);
  run<8>("v_rsq_f64", dout, dc);
  run<9>("v_rcp_f64", dout, dc);
  run<10>("v_cvt_f64_u32", dout, dc);
  run<11>("v_cvt_u32_f64", dout, dc);
  run<19>("v_div_scale_f64", dout, dc);
  run<20>("v_div_fmas_f64", dout, dc);
  run<21>("v_div_fixup_f64", dout, dc);
  return 0;
}
