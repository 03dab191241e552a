// ptm_aux_kernels.hpp -- non-template kernels (exchange phase, verification hooks); included by ptm_engine.hip only.
#pragma once
#include "ptm_kernels.hpp"
#include "ptm_decide.hpp"

namespace ptm {

// ------------------------------------------------------------------------------------------------
// Partition of a sweep's chains (lean MFMA build on big populations): ~1/6 of a long ladder's chains took part in an exchange
// attempt this step and make no Metropolis move (chain.cc:1553-1557).  Riding along they cost a full chain's arithmetic (the
// kernel is bound by instruction issue, not by memory).  This pass packs, for every local rung, the walkers that DO move into
// cidx[rl * W ..) (any order: a chain's random stream is keyed by its identity, not by its place) and settles the others --
// one add_state per attempt, nothing to record in this build -- so the sweep visits moving chains only.
// One block per rung (and 16384 walkers): 16 touch bytes per thread in one load, a block-wide prefix count, one atomic per block.
// ------------------------------------------------------------------------------------------------
constexpr int PART_CHUNK = 16384;   // walkers per block: 1024 threads x 16 touch bytes (one 16-byte load each)
// nhist is NOT advanced here or by the compacted sweep for the chains that make exactly one add_state call this step (every
// moving chain, every rung exchanged once): the engine counts those steps and adds them to all of nhist in one pass when
// somebody needs the array (nhist_flush_kernel).  Only a rung exchanged twice in the step gets its extra add at once.
__global__ __launch_bounds__(1024) void partition_kernel(int W, int rung0, int nchunk, unsigned char* __restrict__ touch,
                                                         unsigned int* __restrict__ nhist, int* __restrict__ cidx, int* __restrict__ ccnt) {
  extern __shared__ int plist[];   // [PART_CHUNK] the block's listed walkers, then written out in one coalesced sweep
  __shared__ int wsum[16];
  __shared__ int sbase, stotal;
  const int rl = rung0 + blockIdx.x / nchunk, ch = blockIdx.x % nchunk;
  const int w0 = ch * PART_CHUNK + threadIdx.x * 16;      // this thread's 16 walkers (W is a multiple of 64: all or none)
  const size_t c0 = (size_t)rl * W + w0;
  uint4 v = make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u);   // past the end: "touched", nothing to list
  const bool in = w0 < W;
  if (in) v = *reinterpret_cast<const uint4*>(touch + c0);
  const unsigned int word[4] = {v.x, v.y, v.z, v.w};
  int mine = 0;
  unsigned int zmask = 0, twice = 0;   // bit k: walker w0 + k moves / was exchanged twice
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const unsigned int t = (word[k >> 2] >> (8 * (k & 3))) & 0xFFu;
    zmask |= (t == 0u ? 1u : 0u) << k;
    twice |= (t > 1u ? 1u : 0u) << k;
    mine += t == 0u ? 1 : 0;
  }
  // block-wide exclusive prefix of `mine`: inclusive scan inside the wave (shuffles), then the 16 wave totals
  int incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d, 64);
    if ((int)(threadIdx.x & 63) >= d) incl += up;
  }
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int k = 0; k < 16; ++k) { const int t = wsum[k]; wsum[k] = run; run += t; }
    stotal = run;
    sbase = run ? atomicAdd(&ccnt[rl], run) : 0;    // (several chunks of a long rung share the rung's list)
  }
  __syncthreads();
  if (in) {
    int pos = wsum[threadIdx.x >> 6] + incl - mine;
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if ((zmask >> k) & 1u) plist[pos++] = w0 + k;
    if (zmask != 0xFFFFu) *reinterpret_cast<uint4*>(touch + c0) = make_uint4(0, 0, 0, 0);
    if (twice) {
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if ((twice >> k) & 1u) nhist[c0 + k] += ((word[k >> 2] >> (8 * (k & 3))) & 0xFFu) - 1u;
    }
  }
  __syncthreads();
  int* out = cidx + (size_t)rl * W + sbase;
  for (int i = threadIdx.x; i < stotal; i += 1024) out[i] = plist[i];
}
// adds `n` to every chain's add_state counter (the steps the compacted sweep did not count one by one)
__global__ __launch_bounds__(256) void nhist_flush_kernel(unsigned int* __restrict__ nhist, size_t Nc, unsigned int n) {
  const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (c < Nc) nhist[c] += n;
}


// swap_count / swap_accept_count (chain.cc:1498,1536; chain.hh:244-245) from the candidate logs of the last `nslots`
// steps (a ring of PTM_LOG_RING slots): one block per ladder counts in LDS and adds to the ladder's counters, {tries,
// accepts} side by side, with coalesced read-modify-writes.  A pair is counted by the shard that owns its lower rung, so
// per-shard counters add up to the ladder's.
__global__ __launch_bounds__(256) void fold_swap_log_kernel(const int* __restrict__ ring, long long* __restrict__ cnt, int W, int ms, int Nt,
                                                           int r0, int r1, int first_slot, int nslots) {
  extern __shared__ int fold_sc[];   // [2][Nt - 1]
  const int w = blockIdx.x, np = Nt - 1;
  for (int i = threadIdx.x; i < 2 * np; i += 256) fold_sc[i] = 0;
  __syncthreads();
  for (int sidx = 0; sidx < nslots; ++sidx) {
    const int slot = (first_slot + sidx) % PTM_LOG_RING;
    const int* L = ring + ((size_t)slot * W + w) * ms;
    for (int k = threadIdx.x; k < ms; k += 256) {
      const int v = L[k];
      if (v < 0) continue;
      const int i = v & 0x3fffffff;
      if (i < r0 || i >= r1) continue;
      atomicAdd(&fold_sc[i], 1);
      if (v & 0x40000000) atomicAdd(&fold_sc[np + i], 1);
    }
  }
  __syncthreads();
  long long* c = cnt + 2 * (size_t)w * np;
  for (int i = threadIdx.x; i < np; i += 256) {
    const int t = fold_sc[i], a = fold_sc[np + i];
    if (t) c[2 * i] += t;
    if (a) c[2 * i + 1] += a;
  }
}

// [W][Nt] -> [Nt][W]: the exchange kernel keeps each evolving ladder's temperatures together, the sweep kernels read a
// chain's inverse temperature at its chain index (rung-major).  32 x 32 tiles through LDS, 256 threads.
__global__ __launch_bounds__(256) void beta_transpose_kernel(const double* __restrict__ in, double* __restrict__ out, int W, int Nt) {
  __shared__ double t[32][33];
  const int r_0 = blockIdx.x * 32, w_0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int w = w_0 + ty + 8 * j, r = r_0 + tx;
    if (w < W && r < Nt) t[ty + 8 * j][tx] = in[(size_t)w * Nt + r];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = r_0 + ty + 8 * j, w = w_0 + tx;
    if (w < W && r < Nt) out[(size_t)r * W + w] = t[tx][ty + 8 * j];
  }
}

// the same for a rung shard of evolving ladders: the chain-indexed temperatures of its own rungs [r0, r0 + nloc)
__global__ __launch_bounds__(256) void beta_local_kernel(const double* __restrict__ in, double* __restrict__ out, int W, int Nt, int r0, int nloc) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)nloc * W) return;
  const int rl = (int)(i / W), w = (int)(i - (size_t)rl * W);
  out[i] = in[(size_t)w * Nt + r0 + rl];
}

// ------------------------------------------------------------------------------------------------
// applies one ladder's row moves IN PLACE, one wave per ladder: GATHER every moved row into registers (16 lanes x 16 B =
// one 256-B row per quarter wave, four rows per load instruction, up to MV rows), wait for all loads, then SCATTER.
// With every read finished before the first write no ordering between the moves is needed (the moves of one ladder
// form cycles over its own rows; other ladders' rows are never touched).
// Two instances: MV = 64 (four ladders per 256-thread block, few registers: short lists, i.e. shards of a few hundred
// rungs) and MV = MVCAP (one ladder per block).  Each handles the ladders whose list fits it and clears mv_n.
// ------------------------------------------------------------------------------------------------
struct Move {
  int DP, W, row_cap;
  double *x, *ll, *lp;
  double *send_up, *send_down;
  const int *mv_src, *mv_dst;
  int* mv_n;
  int* err;
  Hist hist;                        // destination codes <= HIST_DST: the row goes to that chain's history
  int add_every_n;
  const unsigned int* nhist;
  const int *naccept, *ntries, *last_type;
  MapT map;                         // destination codes in (HIST_DST, MAP_DST]: the row is that chain's new MAP
  const double* beta;
  int r0;
};
// One ladder's listed moves, by one wave (s_src / s_dst: the wave's [MV] ints of LDS each).
template <int MV, bool HIST>   // HIST: history destinations exist (compiled apart: the hot build carries none of it)
__device__ __forceinline__ void move_ladder(const Move& p, const int w, const int nmv, int* s_src, int* s_dst, const int lane) {
  // s_dst: >= 0 row slot, -3 nothing, <= -4: slot (-d-4)>>1 of the up (bit 0 clear) / down message
  const int DP = p.DP, RD = DP + ROW_EXTRA;
  // the list goes through LDS (this wave's private part: no block barrier) so that the row gathers below are not
  // chained behind index loads from memory
  for (int j = lane; j < MV; j += 64) {
    const bool in = j < nmv;
    const int sv = in ? p.mv_src[(size_t)w * MVCAP + j] : 0;
    int dv = in ? p.mv_dst[(size_t)w * MVCAP + j] : -3;
    if (dv == -1 || dv == -2) {   // a row that leaves the shard: claim its slot in the boundary message
      const int dir = dv == -1 ? 0 : 1;
      const int slot = atomicAdd(reinterpret_cast<int*>(dir ? p.send_down : p.send_up), 1);
      if (slot >= p.row_cap) { atomicOr(p.err, 4); dv = -3; }
      else dv = -4 - (2 * slot + dir);
    }
    s_src[j] = sv;
    s_dst[j] = dv;
  }
  __builtin_amdgcn_wave_barrier();
  const int g = lane >> 4, sub = lane & 15;
  const bool act = 2 * sub < DP;          // DP/2 lanes of 16 carry a row (16 B each)
  const int col = act ? 2 * sub : 0;      // idle lanes re-read column 0 (harmless) so that no load is predicated
  d2_t v[MV / 4];
  double sl[MV / 64], sp[MV / 64];
#pragma unroll
  for (int q = 0; q < MV / 64; ++q) {
    sl[q] = p.ll[s_src[64 * q + lane]];
    sp[q] = p.lp[s_src[64 * q + lane]];
  }
  for (int hc = 0; hc < DP; hc += 32) {   // (rows of 64 dimensions: their second 256 bytes the same way)
    const int colh = col + hc;
#pragma unroll
    for (int q = 0; q < MV / 4; ++q)        // unconditional loads: entries past the list read row 0 and are never stored
      v[q] = *reinterpret_cast<const d2_t*>(p.x + (size_t)s_src[4 * q + g] * DP + colh);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every gather has landed before the first scatter
#pragma unroll
    for (int q = 0; q < MV / 4; ++q) {
      const int d = s_dst[4 * q + g];
      if (d != -3 && act) {
        double* dstp;
        if (d >= 0) dstp = p.x + (size_t)d * DP;
        else if (HIST && d <= MAP_DST && d > HIST_DST) dstp = p.map.x + (size_t)(MAP_DST - d) * DP;
        else if (HIST && d <= HIST_DST) {
          const int c = HIST_DST - d;
          dstp = p.hist.x + hist_slot(p.hist, 1 + (long long)(p.nhist[c] / (unsigned int)p.add_every_n), c) * DP;
        } else { const int e = -d - 4; dstp = ((e & 1) ? p.send_down : p.send_up) + MSG_HDR + (size_t)(e >> 1) * RD; }
        *reinterpret_cast<d2_t*>(dstp + colh) = v[q];
      }
    }
  }
#pragma unroll
  for (int q = 0; q < MV / 64; ++q) {
    const int d = s_dst[64 * q + lane];
    if (d >= 0) { p.ll[d] = sl[q]; p.lp[d] = sp[q]; }
    else if (HIST && d <= MAP_DST && d > HIST_DST) {
      const int c = MAP_DST - d;
      const double t = p.beta[p.r0 + c / p.W] * sl[q];
      p.map.lpost[c] = sp[q] + t; p.map.ll[c] = sl[q]; p.map.lp[c] = sp[q];
    } else if (HIST && d <= HIST_DST) {
      const int c = HIST_DST - d;
      const long long hrow = 1 + (long long)(p.nhist[c] / (unsigned int)p.add_every_n);
      hist_scalars(p.hist, hist_slot(p.hist, hrow, c), hrow, sl[q], sp[q], p.naccept[c], p.ntries[c], p.last_type[c],
                   p.beta[p.r0 + c / p.W]);   // (evolving ladders never come here: their exchange kernel moves the rows itself)
    } else if (d != -3) {
      const int e = -d - 4;
      double* row = ((e & 1) ? p.send_down : p.send_up) + MSG_HDR + (size_t)(e >> 1) * RD;
      row[DP] = sl[q]; row[DP + 1] = sp[q]; row[DP + 2] = (double)w; row[DP + 3] = 0.0;
    }
  }
  if (lane == 0) p.mv_n[w] = 0;
  __builtin_amdgcn_wave_barrier();   // (the wave's LDS lists are reused for its next ladder)
}
// A wave looks at the list lengths of 64 consecutive ladders in ONE load and works through the few that have a list: the ladders
// whose moves fitted their exchange block's registers (nearly all) cost nothing here -- a block per ladder spent 63 us at the
// 8-GPU size's 131072 ladders finding that out.
template <int MV, int WPB, bool HIST>
__global__ __launch_bounds__(64 * WPB, MV > 64 ? 1 : 4) void move_kernel(const Move p) {
  __shared__ int s_all[WPB][2][MV];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int base = (blockIdx.x * WPB + wv) * 64;
  if (base >= p.W) return;
  const int n = base + lane < p.W ? p.mv_n[base + lane] : 0;
  unsigned long long todo = __builtin_amdgcn_ballot_w64(n > 0 && n <= MV);
  while (todo) {
    const int k = __builtin_ctzll(todo);
    todo &= todo - 1;
    const int nmv = __builtin_amdgcn_readlane(n, k);
    move_ladder<MV, HIST>(p, base + k, nmv, s_all[wv][0], s_all[wv][1], lane);
  }
}

// ------------------------------------------------------------------------------------------------
// lands the rows of the two boundary messages in the holes decide_kernel named (arr_below / arr_above); 16 lanes per
// row.  blockIdx.y = 0: message from below, 1: from above.
// ------------------------------------------------------------------------------------------------
struct Install {
  int DP, W, row_cap;
  double *x, *ll, *lp;
  const double *recv_below, *recv_above;
  int *arr_below, *arr_above;
  int* err;
};
__global__ __launch_bounds__(256) void install_kernel(const Install p) {
  const double* msg = blockIdx.y ? p.recv_above : p.recv_below;
  if (!msg) return;
  int* arr = blockIdx.y ? p.arr_above : p.arr_below;
  int n = *reinterpret_cast<const int*>(msg);
  if (n > p.row_cap) n = p.row_cap;         // (the sender has flagged the overflow)
  const int j = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
  if (j >= n) return;
  const int DP = p.DP, RD = DP + ROW_EXTRA;
  const double* row = msg + MSG_HDR + (size_t)j * RD;
  const int w = (int)row[DP + 2];
  const int a = (w >= 0 && w < p.W) ? arr[w] : -1;
  if (a < 0) { atomicOr(p.err, 8); return; }  // a row nobody expects: the two shards disagree about the step
  for (int col = 2 * sub; col < DP; col += 32)
    *reinterpret_cast<d2_t*>(p.x + (size_t)a * DP + col) = *reinterpret_cast<const d2_t*>(row + col);
  if (sub == 0) { p.ll[a] = row[DP]; p.lp[a] = row[DP + 1]; }
}

// history row 0: the initial state (MH_chain::initialize -> add_state, chain.cc:871-875)
__global__ void hist_init_kernel(Hist h, int DP, const double* x, const double* ll, const double* lp, const int* naccept,
                                 const int* ntries, const int* last_type, const double* beta, const double* betaC, int W, int r0) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= h.HC) return;
  const size_t o = hist_slot(h, 0, c);
  for (int d = 0; d < DP; ++d) h.x[o * DP + d] = x[(size_t)c * DP + d];
  hist_scalars(h, o, 0, ll[c], lp[c], naccept[c], ntries[c], last_type[c], betaC ? betaC[c] : beta[r0 + c / W]);
}
// the ladders start to evolve with rows already saved: those rows were saved at the common ladder's temperatures
__global__ void hist_beta_fill_kernel(Hist h, const double* beta, int W, int r0) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)h.cap * h.HC) return;
  h.beta[i] = beta[r0 + (int)(i % h.HC) / W];
}

// MAP after initialize: the initial state, if its log-posterior beats -1e200 (chain.hh:69, chain.cc:931-934)
__global__ void map_init_kernel(MapT m, int DP, int W, int r0, const double* beta, const double* x, const double* ll, const double* lp) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= m.MC) return;
  m.lpost[c] = -1e200;
  const double t = beta[r0 + c / W] * ll[c];
  if (map_try(m, c, lp[c] + t, ll[c], lp[c]))
    for (int d = 0; d < DP; ++d) m.x[(size_t)c * DP + d] = x[(size_t)c * DP + d];
}

// verification hooks
__global__ void debug_eval_kernel(int fn, const double* a, const double* b, double* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double r = 0;
  switch (fn) {
    case 0: r = dlog(a[i]); break;
    case 1: r = dexp(a[i]); break;
    case 2: r = dsin_0_pi(a[i]); break;
    case 3: r = dcos_hpi(a[i]); break;
    case 4: r = dsqrt(a[i]); break;
    case 5: r = a[i] / b[i]; break;
    case 6: r = __builtin_sqrt(a[i]); break;  // raw compiler expansion (to measure how often the fix-up fires)
  }
  out[i] = r;
}
__global__ void debug_philox_kernel(uint64_t seed, int tag, uint32_t stream, uint64_t step, uint32_t block, uint32_t* out) {
  const u32x4 o = draw_block(seed, tag, stream, step, block);
  out[0] = o.v0; out[1] = o.v1; out[2] = o.v2; out[3] = o.v3;
}
__global__ void debug_boxmuller_kernel(const uint32_t* k1, const uint32_t* k2, double* z0, double* z1, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  boxmuller(k1[i], k2[i], (const double*)BM_TABLE, z0[i], z1[i]);
}
// exhaustive scan over all 2^32 first arguments of Box-Muller: counts the k for which the unscaled Newton sqrt of
// the hot path (bm_sqrt) differs from the correctly rounded dsqrt() on a = -2 ln((k+.5)/2^32), and the k with a
// outside (0, 64) (bm_sqrt's domain)
__global__ void debug_sqrt_scan_kernel(unsigned long long* mismatches) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t stride = gridDim.x * blockDim.x;
  unsigned long long bad = 0, out = 0;
  for (uint64_t k = tid; k < (1ull << 32); k += stride) {
    const double a = bm_neg2log((uint32_t)k, (const double*)BM_TABLE);
    if (bm_sqrt(a) != dsqrt(a)) bad++;
    if (!(a > 0.0 && a < 64.0)) out++;
  }
  if (bad) atomicAdd(mismatches, bad);
  if (out) atomicAdd(mismatches + 1, out);
}

// ------------------------------------------------------------------------------------------------
// Calibration of the box a measurement runs on (ptm_calibrate; bench.py puts the figures beside its roofline): MI355X devices of
// one pool differ by several per cent in the clock they hold under an f64 load, and a bench line cannot otherwise tell a slow
// box from a slow kernel.  (1) a plain streaming copy, 16 bytes per lane, grid-stride: what HBM gives this device;
// (2) an f64 fma issue loop, four waves per SIMD on every CU, eight independent chains per lane: what the f64 vector pipe
// gives it, with the shader clock it held meanwhile (s_memtime ticks per 100 MHz s_memrealtime tick).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void calib_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}
// sums of the MH_chain counters over the engine's chains (ptm_get_counter_sums): a measurement reads two numbers instead of two arrays
__global__ __launch_bounds__(256) void counter_sums_kernel(const int* __restrict__ ntries, const int* __restrict__ naccept, size_t n, unsigned long long* __restrict__ out) {
  __shared__ unsigned long long sh[2][4];
  unsigned long long t = 0, a = 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { t += (unsigned long long)(long long)ntries[i]; a += (unsigned long long)(long long)naccept[i]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { t += __shfl_down(t, o); a += __shfl_down(a, o); }
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = t; sh[1][threadIdx.x >> 6] = a; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(out, sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]);
    atomicAdd(out + 1, sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]);
  }
}
__global__ __launch_bounds__(256) void calib_fma_kernel(double* __restrict__ out, long long* __restrict__ clk, int iters) {
  double d[8];
  const double e0 = 1.0 - 1e-9 * (threadIdx.x & 7), e1 = 1e-12 * (1 + (threadIdx.x & 3));
#pragma unroll
  for (int i = 0; i < 8; ++i) d[i] = 1.0 + 1e-3 * (threadIdx.x + i);
  __syncthreads();
  const long long c0 = (long long)__builtin_amdgcn_s_memtime(), r0 = (long long)__builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(e0), "v"(e1));
  }
  const long long c1 = (long long)__builtin_amdgcn_s_memtime(), r1 = (long long)__builtin_amdgcn_s_memrealtime();
  double r = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += d[i];
  if (r == 12345.678) out[0] = r;   // (keeps the chains alive; never true)
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

}  // namespace ptm
