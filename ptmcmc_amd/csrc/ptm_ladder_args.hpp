// ptm_ladder_args.hpp -- argument block and LDS sizes of the persistent ladder kernel (ptm_ladder_kernel.hpp), shared with the host
#pragma once
#include <stddef.h>

namespace ptm {

constexpr int LADDER_H = 8;   // halo depth in rungs (parallel.DEFAULT_HALO has the run-length statistics)
constexpr int LADDER_ABORT_BIT = 1 << 30;   // in the arrivals word ctl[3]: some workgroup of the launch has given up
constexpr int LADDER_THREADS = 512;   // four waves of chains (256 lanes: a lane per dimension) + four bookkeeper waves

struct LadderArgs {
  int nsteps;           // steps asked for
  int NB;               // workgroups per walker-ladder
  int ms;               // maxswapsperstep (chain.cc:1192)
  double thresh;        // (Ntemps-1)*swap_rate/maxswapsperstep (chain.cc:1413)
  double* pub_x;        // [2][Nc][DP]  published rows (dimension d at d), by step parity
  double* pub_ll;       // [2][Nc]
  double* pub_lp;       // [2][Nc]
  double* pub_s;        // [2][Nc][2]   evolving ladders: {llike, stamp} in one 16-byte word (stamp = launch number x 2^24 + step + 1)
  int* flags;           // [W * NB]  steps published by each workgroup since the launch began
  int* ctl;             // [0] abort (a neighbour never showed up), [1] steps done (set by workgroup 0), [2] whole-ladder steps taken,
                        // [3] workgroups that have finished all their steps | LADDER_ABORT_BIT (the all-or-nothing word)
  int* done_seq;        // sequence number of the engine's last launch of this kernel that was committed (persistent; not cleared per launch)
  int seq;              // this launch's number: it runs only if *done_seq == seq - 1
  int* slow_done;       // [W] workgroups that finished a whole-ladder exchange (a barrier of the ladder's workgroups)
  long long* swap_cnt;  // [W][Nt-1][2] {tries, accepts} (chain.hh:244-245)
  int* swap_log;        // [W][ms] candidate log of the LAST step asked for (ptm_get_last_swaps)
  long long spin_limit; // wall-clock ticks a workgroup waits for a neighbour before it gives up
  long long* prof;      // null, or [W * NB][8] phase clocks (diagnostics)
  int prof_tid;         // ... of this thread: 0 a chains' wave, 256 a replay wave, 384 a window wave (PTM_LADDER_PROF=1 / 2 / 3)
  double evolve_rate;   // evolving ladders (FL bit 2): the rate of pry_temps (chain.cc:1829)
  int max_run;          // longest run of surviving picks on consecutive rungs a step may hold (<= LADDER_H; tests lower it)
};

// LDS of the decide replay (bytes): first[Nt] | cand[ms] | ua[ms] | alive[ms] (+pad) | flag words
inline size_t ladder_decide_lds_bytes(int Nt, int ms) { return (size_t)((Nt + 1) & ~1) * 4 + (size_t)((ms + 1) & ~1) * 8 + (size_t)((ms + 7) & ~7) + 32 + (size_t)Nt * 8 + (size_t)((Nt + 1) & ~1) * 8 + 16; }
// ... and of the window: llike (working + original), lprior, rows, perm | tries / accepts of the own pairs
inline size_t ladder_window_lds_bytes(int DP) {
  const int R = 256 / DP, WN = 1 + R + LADDER_H;
  return (size_t)WN * 8 * 5 + (size_t)WN * DP * 8 + (size_t)((WN + 3) & ~3) * 8 + (size_t)R * 8 + 64;
}

// ... and, at 32 dimensions with a full factor, the precision matrix as padded rows (stride DP + 1: conflict-free for a lane per row):
// with the bookkeeper wave two waves share a SIMD and a lane has 256 registers, not 512 -- its row of the matrix no longer fits beside
// its row of the factor
inline size_t ladder_psq_lds_bytes(int DP) { return DP == 32 ? (size_t)DP * (DP + 1) * 8 : 0; }

// ... and of an evolving ladder: temperatures, gaps, prefix sums [Nt] each | chunk totals and offsets | per candidate: log-uniform,
// normaliser, increase (doubles), pries before, pried pair, pick list (ints) | two counters
inline size_t ladder_ev_lds_bytes(int Nt, int ms) { return (size_t)Nt * 24 + (size_t)(2 * ((Nt + 31) / 32) + 4) * 8 + (size_t)ms * 24 + (size_t)ms * 12 + 64 + (size_t)((ms + 7) & ~7); }

}  // namespace ptm
