// ptm_ladder_kernel.hpp -- LONG LADDERS OF FEW WALKERS: many parallel_tempering_chains::step calls in ONE launch, the chains'
// state resident in registers.
//
// The reference's own shape -- ONE ladder of 1024 rungs (BASELINE: D = 32, 1024 temperatures) -- is 512 waves of work per
// step and all latency: as two launches per step (exchange kernel, lanes kernel) it costs ~19 us, most of it launch
// boundaries and the ~10 dependent trips to memory between them.  Here a grid of resident workgroups walks the steps of a
// ptm_step(n) call without returning to the host (the caller's loop this replaces: ptmcmc.cc:563-599):
//
//   * a workgroup owns R = 256 / DP consecutive rungs of one walker's ladder (8 at DP = 32).  Its waves 0..3 (256 lanes) hold the
//     chains, a lane per dimension as in the lanes kernel (ptm_lanes_kernel.hpp): the state row sits in ONE register per lane,
//     llike / lprior / the MH_chain counters in registers too, the rung's row of the proposal factor (it never changes) in 32
//     registers, tables in LDS -- loaded once per launch.  Waves 4..7 are BOOKKEEPERS: they take the exchange phase's replay and
//     the hand-over off the chains' waves (a lone wave uses less than half of its SIMD's issue slots; two share each SIMD here);
//   * every workgroup replays the step's candidate draws and the survivor filter of the WHOLE ladder (chain.cc:1410-1418: they
//     depend on the ladder's random stream only), exactly as the shards of a multi-GPU run do (ptm_decide.hpp), and decides the
//     exchanges that can change its own rungs from its own llikes, the top rung of the workgroup below and the bottom H = 8
//     rungs of the workgroup above;
//   * after its Metropolis moves a workgroup PUBLISHES its rows, llikes and lpriors (a double buffer by step parity) and raises
//     its flag; before the trials of the next step it waits for the flags of its two NEIGHBOURS only -- no grid-wide barrier --
//     and reads their published rungs (the halo llikes and the rows that may move in: exchanges propagate downwards only,
//     chain.cc:1417-1418, so a row comes from at most H rungs above or one below).  The candidate draws, the filter, the
//     step's normals and the product factor . z do not depend on anybody's state: they are computed while the flags travel;
//   * a run of H + 1 surviving picks on consecutive rungs would reach past a halo (probability ~ swap_rate^9 / 9! per rung and
//     step).  It is a property of the draws alone and every workgroup of the ladder sees it: that step's exchange phase is then
//     taken from the WHOLE ladder's publications -- all the ladder's workgroups wait for each other, replay every trial, and
//     nobody publishes again before everybody has read.  Nothing is ever decided blindly, and no step leaves the kernel.
//   * the Metropolis moves of a step are made BEFORE its exchange phase is looked at: which rungs an exchange attempt touches
//     (they make no move) is known from the draws, the moves of the others need nothing of the neighbours -- the flags travel
//     meanwhile.
//
// Arithmetic: the operation sequences of lanes_body / decide_body, number for number -- the chains are bit-identical to the
// two-launch path and to the CPU checker (the parity suite runs through this kernel wherever it applies).
// Scope: device target (a mean included), DP = 4 .. 32, open or `limit` bounds and an all-uniform prior -- or, bit 4 of FL, any boundary and
// any per-dimension prior; template flags FL: bit 0 =
// one-dimensional moves and scale mixtures (the reference sampler's default Gaussian recipe, ptmcmc.cc:117-139), bit 1 = the history
// ring and MAP tracking of MH_chain::add_state (chain.cc:931-946), rows of a rung exchanged twice in a step included (quirk Q6); bit 2 =
// EVOLVING LADDERS (parallel_tempering_chains::evolve_temps: pry_temps after every accepted exchange, chain.cc:1501-1518,1809-1846,
// without the posterior-ordering cut).  A pry renormalises ALL gaps and every later trial of the step sees it, so every workgroup
// replays every trial of the ladder in pick order -- every step takes the whole ladder's published llikes (the form a run of picks
// longer than the halo takes on a fixed ladder), keeps the ladder's temperatures in LDS, and the Metropolis tests wait for the
// step's new temperatures.  With every step in that form the double buffer of the publications needs no second barrier: nobody
// can publish step s + 2 before everybody has published step s + 1, i.e. has read step s.
//
// ALL OR NOTHING.  A workgroup that waits in vain for a neighbour (a grid that is not resident: a shared device, a second engine's
// kernel in the way) gives up, and then nothing of the launch may stay: the chains live in registers until every workgroup of the grid
// has finished every step -- each adds itself to ONE counter word and waits for it to reach the grid size; giving up sets a bit in the
// same word, so no workgroup can ever read "all there" once anybody has given up, and a workgroup that gives up waiting for the
// counter itself learns from the value its atomic-or returns whether everybody had arrived after all.  Only then are rows, scalars,
// counters and swap counters written back and the launch's sequence number stored (LadderArgs::done_seq): the host finds a launch
// that gave up at its next look (ladder_settle, ptm_engine.hip), the engine's arrays untouched, and repeats its steps on the
// two-launch path.  A launch that finds its predecessor's number missing does nothing.  History rows and MAP entries are written as
// the steps produce them: the repeated steps produce the same rows into the same slots and the same maxima.
//
// MEMORY ORDER of the hand-over (the form MI355X_MICROARCH.md lists under "Hand-offs measured with sc1 loads in place of the
// acquire", first row): every published word is stored with an agent-scope relaxed atomic (an sc1 store: written through to memory,
// dropped from the XCD's L2); every storing wave waits for its stores (s_waitcnt vmcnt(0)) BEFORE the workgroup's barrier; ONE lane
// raises the workgroup's flag (an sc1 store) AFTER that barrier; the consumer polls the flag with sc1 loads (they bypass its L1, no
// stale line can answer), and the wave that polled issues the loads of the published words -- sc1 loads to registers, every one of
// them -- only after its poll has matched: vector-memory loads of a wave return in issue order, and a compiler barrier after every
// look at a flag keeps the compiler from hoisting a payload load above it; the other waves read what that wave put into LDS behind
// a workgroup barrier.  This is measured hardware behaviour of gfx950 / ROCm 7.2, not a guarantee of the HIP memory model (which
// would ask for a release / acquire pair: +0.4 us per hand-over, tools/probes/flag_pingpong_probe.hip); the parity suite, the soak
// runs and the forced-give-up test run through it.
#pragma once
#include "ptm_decide.hpp"
#include "ptm_ladder_args.hpp"
#include "ptm_lanes_kernel.hpp"

namespace ptm {

// a published word that validates itself: {value, stamp} in ONE 16-byte sc1 store, read by one 16-byte sc1 load (no torn pair seen in
// 2 x 10^6 racing loads and stores, tools/probes/stamp_handover_probe.hip; MI355X_MICROARCH.md: "R2's granule needs no ordering at all")
typedef double lad_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ lad_d2 lad_load16(const lad_d2* q) {
  lad_d2 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(q) : "memory");
  return v;
}
__device__ __forceinline__ void lad_store16(lad_d2* q, lad_d2 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(q), "v"(v) : "memory"); }

template <int DP, int KIND, int FL>
__global__ __launch_bounds__(LADDER_THREADS) void ladder_persistent_kernel(const Dev p, const LadderArgs a) {
  static_assert(DP == 4 || DP == 8 || DP == 16 || DP == 32, "persistent ladder kernel: DP 4 .. 32");
  constexpr bool GENX = (FL & 1) != 0;   // one-dimensional moves, scale mixtures
  constexpr bool HIST = (FL & 2) != 0;   // history ring, MAP tracking
  constexpr bool EV = (FL & 4) != 0;     // evolving ladders
  constexpr bool DEB = (FL & 8) != 0;    // differential evolution from the chain's saved history (with GENX and HIST: FL = 11, 15)
  constexpr bool GENS = (FL & 16) != 0;  // general state space: any boundary (wrap, reflect) and any per-dimension prior (built with GENX and HIST: 19, 23, 27, 31)
  static_assert(!DEB || (GENX && HIST), "differential evolution is a member of a proposal set and draws from the history ring");
  // a launch whose predecessor gave up does nothing: the host repeats that launch's steps, and this one's, on the two-launch path
  if (__hip_atomic_load(a.done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.seq - 1) return;
  constexpr int R = 256 / DP;            // rungs per workgroup
  constexpr int CPW = 64 / DP;           // chains per wave
  constexpr int H = LADDER_H;
  constexpr int WNMAX = 1 + R + H;
  static_assert(R >= H, "a halo must fit the neighbouring workgroup");
  extern __shared__ __attribute__((aligned(16))) double lds_all[];
  const int tid = threadIdx.x;
  // Waves 0..3 hold the workgroup's chains (a lane per dimension); waves 4..7 are the BOOKKEEPERS of the exchange phase (step 2 below).
  const bool helper = tid >= 256;
  const int ht = tid - 256;   // a bookkeeper's thread number: 0..127 replay the draws (waves 4, 5), 128..255 fetch the window (waves 6, 7)
  const bool drole = helper && ht < 128, wrole = helper && ht >= 128;
  const int wt = ht - 128;
  const int Nt = p.Nt, ms = a.ms, NB = a.NB;
  const uint64_t try_below = u01_below_bound(a.thresh);
  const int NONE = 0x7fffffff;

  // -- which ladder, which rungs.  Consecutive workgroup ids go round the 8 XCDs; neighbours in the ladder should share an L2.
  const int G = gridDim.x;
  int L = blockIdx.x;
  if ((G & 7) == 0) L = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int w = L / NB, b = L - w * NB;
  const int r0 = b * R;
  const int r1 = r0 + R < Nt ? r0 + R : Nt;
  const int wlo = r0 > 0 ? r0 - 1 : 0;
  const int whi = (r1 - 1 + H < Nt - 1) ? r1 - 1 + H : Nt - 1;
  const int WN = whi - wlo + 1;

  // -- LDS carve
  double* p2s = lds_all + BM_TABLE_DOUBLES;
  double* wsc = p2s + lanes_p2_doubles<DP>() + ((tid >> 6) & 3) * (3 * 64 + 4 * CPW);   // (the bookkeeper never touches it)
  double* vbuf = wsc;
  double* sbuf = wsc + 64;
  double* pbuf = wsc + 3 * 64;
  unsigned char* dsm = reinterpret_cast<unsigned char*>(lds_all + ((lanes_lds_doubles<DP>(4) + 1) & ~1));
  int* first = reinterpret_cast<int*>(dsm);                               // [Nt]
  int* cand = first + ((Nt + 1) & ~1);                                    // [ms]
  uint32_t* ua = reinterpret_cast<uint32_t*>(cand + ((ms + 1) & ~1));     // [ms]
  unsigned char* alive = reinterpret_cast<unsigned char*>(ua + ((ms + 1) & ~1));   // [ms]
  int* sflag = reinterpret_cast<int*>(alive + ((ms + 7) & ~7));           // [0] run longer than the halo, [1], [2] gave up waiting
  double* wll = reinterpret_cast<double*>(sflag + 8);                     // [WN] llike view, exchanged as the picks are decided
  double* wll0 = wll + WNMAX;                                             // [WN] llikes as published
  double* wlp0 = wll0 + WNMAX;                                            // [WN] lpriors as published
  double* wdb = wlp0 + WNMAX;                                             // [WN] -(beta[n+1] - beta[n]) of the window's pairs
  double* wlu = wdb + WNMAX;                                              // [WN] log of the accept uniform of the pair's surviving pick
  double* wx = wlu + WNMAX;                                               // [WN][DP] rows as published
  int* wperm = reinterpret_cast<int*>(wx + WNMAX * DP);                   // [WN] source rung of the row now at a rung
  int* wmid = wperm + ((WNMAX + 3) & ~3);                                 // [WN] source rung of the row a rung held BETWEEN its two exchanges of this step
  int* ptry = wmid + ((WNMAX + 3) & ~3);                                  // [R] exchange attempts of the own pairs (lower rung here)
  int* pacc = ptry + R;                                                   // [R] ... accepted
  double* llall = reinterpret_cast<double*>(pacc + R + ((2 * R) & 1));    // [Nt] whole-ladder llike view (steps with a run longer than the halo)
  int* permall = reinterpret_cast<int*>(llall + Nt);                      // [Nt] ... and row map
  int* midall = permall + ((Nt + 1) & ~1);                                // [Nt] ... and in-between rows
  constexpr bool PROW_LDS = DP == 32 && KIND != KIND_DIAG;                // the precision matrix's rows from LDS (ptm_ladder_args.hpp)
  double* psq = reinterpret_cast<double*>(midall + ((Nt + 1) & ~1));      // [DP][DP + 1], zeros above the diagonal
  // evolving ladders: the ladder's inverse temperatures (kept over the steps), the step's gaps and their prefix sums, chunk totals,
  // per candidate: the log of its accept uniform, the normaliser S and the number of pries BEFORE its trial; per pry: pair and increase
  double* bwl = psq + (DP == 32 ? DP * (DP + 1) : 0);                     // [Nt]
  double* spl = bwl + Nt;                                                 // [Nt]
  double* P0l = spl + Nt;                                                 // [Nt] prefix sums of the step's first gaps (chunk order)
  const int nchunk = (Nt + 31) / 32;                                      // chunks of 32 gaps (ptmo_chunk_prefix)
  double* cts = P0l + Nt;                                                 // [nchunk] chunk totals, then the new prefix's offsets | [nchunk] the new normaliser |
                                                                          // [nchunk + 1] S at the step's start | [nchunk + 2 ..) the first prefix's offsets
  double* lul = cts + 2 * nchunk + 4;                                     // [ms]
  double* kSl = lul + ms;                                                 // [ms]
  double* incl = kSl + ms;                                                // [ms]
  int* knp = reinterpret_cast<int*>(incl + ms);                           // [ms]
  int* ipry = knp + ms;                                                   // [ms]
  int* plist = ipry + ms;                                                 // [ms] surviving picks in pick order
  int* evi = plist + ms;                                                  // [0] their number, [1] pries of this step

  lanes_stage<DP>(p, lds_all);
  for (int i = tid; i < Nt; i += LADDER_THREADS) first[i] = NONE;
  if (tid < R) { ptry[tid] = 0; pacc[tid] = 0; }
  if (tid < 8) sflag[tid] = 0;   // ([5], [6]: arrivals of the bookkeeper waves at their own barriers, counted over the steps)
  if (tid < WN - 1) wdb[tid] = -(p.beta[wlo + tid + 1] - p.beta[wlo + tid]);   // chain.cc:1463

  // -- this lane's chain
  const int lane = tid & 63;
  const int d = lane % DP, g = lane / DP;
  const int slot = (tid >> 6) * CPW + g;        // rung of the workgroup
  const bool live = !helper && r0 + slot < r1;
  const int rg = live ? r0 + slot : r1 - 1;     // dead lanes shadow the last rung and write nothing
  const int c = rg * p.W + w;
  const bool lead = d == 0;
  const int pos = row_pos<DP>(d);
  const bool hist_on = HIST && rg < p.hist.rungs, map_on = HIST && rg < p.map.rungs;
  const unsigned int every = (unsigned int)p.add_every_n;
  auto sync_wave = [] { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); };
  constexpr unsigned long long GM = DP == 64 ? ~0ull : ((1ull << (DP & 63)) - 1ull);
  auto all_of_chain = [&](bool v) { return ((__builtin_amdgcn_ballot_w64(v) >> (g * DP)) & GM) == GM; };

  double xd = p.x[(size_t)c * DP + pos];
  double ll = p.ll[c], lp = p.lp[c];
  int ntries = p.ntries[c], naccept = p.naccept[c], last_type = p.last_type[c];
  unsigned int nhist = p.nhist[c];
  // the rung's MAP log-posterior in a register of every lane of the chain (the rung's MAP is this workgroup's alone for the launch;
  // a look at memory per add_state would be a round trip on the chains' critical path, and every lane of a chain holds the same
  // candidate scalars: all of them take the same decision, no lane exchange)
  double map_lp = map_on ? p.map.lpost[c] : 0.0;
  auto map_take = [&](double lpost, double l_, double p_) -> bool {   // chain.cc:931-934
    if (!(lpost > map_lp)) return false;
    map_lp = lpost;
    if (lead && live) { p.map.lpost[c] = lpost; p.map.ll[c] = l_; p.map.lp[c] = p_; }
    return true;
  };
  double beta = p.beta[rg];   // (evolving ladders: the ladder-major image, refreshed after every step's exchange phase)
  if (EV) {
    for (int k = tid; k < Nt; k += LADDER_THREADS) bwl[k] = p.beta_w[(size_t)w * Nt + k];
    beta = p.beta_w[(size_t)w * Nt + rg];
  }
  const double grow = 1.0 + a.evolve_rate;
  const double plo = p.plo[d], phi = p.phi[d];
  const double mean_d = p.has_mean ? p.mean[d] : 0.0;   // the target's mean (x - 0.0 is x, bit for bit: no second code path)
  // open / `limit` boundaries (boundary::enforce, states.cc:53-55) as a second box: without bounds it holds everything
  const double elo = (p.has_bounds && p.blo[d] == B_LIMIT) ? p.bmin[d] : -__builtin_inf();
  const double ehi = (p.has_bounds && p.bhi[d] == B_LIMIT) ? p.bmax[d] : __builtin_inf();
  const uint32_t stream = (uint32_t)(w + p.w_off) * (uint32_t)Nt + (uint32_t)rg;
  // row d of the rung's factor (column-major [col][row]); the sigma of a diagonal proposal
  double tcol[KIND == KIND_DIAG ? 1 : DP];
  if (KIND == KIND_DIAG) tcol[0] = p.prop[(size_t)rg * p.prop_stride + d];
  else {
#pragma unroll
    for (int j = 0; j < DP; ++j) tcol[KIND == KIND_DIAG ? 0 : j] = p.prop[(size_t)rg * p.prop_stride + (size_t)j * DP + d];
  }
  // row d of the packed precision matrix {2P_d0 .. 2P_d,d-1, P_dd}, zeros behind it (fma(0, y, s) == s: the chain below runs over all DP)
  double prow[PROW_LDS ? 1 : DP];
  if (PROW_LDS) {
    for (int i = tid; i < DP * DP; i += LADDER_THREADS) {
      const int r = i / DP, j = i - r * DP;
      psq[r * (DP + 1) + j] = j <= r ? p2s[(size_t)r * (r + 1) / 2 + j] : 0.0;
    }
  } else {
#pragma unroll
    for (int j = 0; j < DP; ++j) prow[PROW_LDS ? 0 : j] = j <= d ? p2s[(size_t)d * (d + 1) / 2 + j] : 0.0;
  }
  const double* const prl = psq + d * (DP + 1);
  const size_t NcDP = (size_t)p.Nc * DP;
  const int blk = w * NB + b;
  __syncthreads();

  // optional phase clock (a.prof != null: PTM_LADDER_PROF=1, tools/w1_probe.py): 100 MHz ticks per phase, summed over the steps
  long long tick_sum[7] = {0, 0, 0, 0, 0, 0, 0}, tick_last = 0;
#define PTM_LADDER_TICK(k) do { if (a.prof && tid == a.prof_tid) { const long long t_ = wall_clock64(); if ((k) > 0) tick_sum[(k)] += t_ - tick_last; tick_last = t_; } } while (0)
#define PTM_LADDER_ALIVE(r) ((r) >= 0 && (r) <= Nt - 2 && first[(r)] != NONE && alive[first[(r)]])
  // one trial (chain.cc:1459-1467) on a llike view `lv` / row map `pm` indexed from rung `base`; own pairs are counted and logged
  auto trial = [&](double* lv, int* pm, int* md, int base, int i, bool last_step, double dbeta, double lu) {
    const int kk = first[i];
    double lla = lv[i - base];
    if (!(lla > -1e200)) lla = -1e200;
    double llb = lv[i + 1 - base];
    if (!(llb > -1e200)) llb = -1e200;
    const double logH = dbeta * (llb - lla);
    bool acc = true;
    if (logH < 0) acc = lu < logH;
    if (acc) {
      const double t = lv[i - base]; lv[i - base] = lv[i + 1 - base]; lv[i + 1 - base] = t;
      const int q = pm[i - base]; pm[i - base] = pm[i + 1 - base]; pm[i + 1 - base] = q;
    }
    // rung i is exchanged once more if the pair below survived too: what it holds now is the row its FIRST add_state of the
    // step sees (chain.cc:1487-1490; quirk Q6)
    if (HIST && PTM_LADDER_ALIVE(i - 1)) md[i - base] = pm[i - base];
    if (i >= r0 && i < r1) {                              // an own pair: its counters, its line of the log
      ptry[i - r0] += 1;
      if (acc) pacc[i - r0] += 1;
      if (last_step) a.swap_log[(size_t)w * ms + kk] = i | (acc ? 0x40000000 : 0);
    }
  };
  // thread 0 waits until workgroup `nb` of this ladder has published step s (false: gave up)
  // (a look at a flag is a round trip through the L2, ~0.8 us: the abort word travels beside it, not behind it)
  auto wait_for = [&](int nb, int s, long long t0) {
    int* f = &a.flags[w * NB + nb];
    for (;;) {
      const int fv = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int ab = __hip_atomic_load(&a.ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (fv >= s + 1) { asm volatile("" ::: "memory"); return true; }   // (no load of the published words may be hoisted above this look)
      if (ab != 0 || wall_clock64() - t0 > a.spin_limit) return false;
      __builtin_amdgcn_s_sleep(1);
    }
  };
  // ... and both neighbours' flags in one round trip
  auto wait_for_neighbours = [&](int s, long long t0) {
    int* flo = &a.flags[blk - (b > 0 ? 1 : 0)];
    int* fhi = &a.flags[blk + (b + 1 < NB ? 1 : 0)];
    for (;;) {
      const int vlo = __hip_atomic_load(flo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int vhi = __hip_atomic_load(fhi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int ab = __hip_atomic_load(&a.ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((b == 0 || vlo >= s + 1) && (b + 1 >= NB || vhi >= s + 1)) { asm volatile("" ::: "memory"); return true; }
      if (ab != 0 || wall_clock64() - t0 > a.spin_limit) return false;
      __builtin_amdgcn_s_sleep(1);
    }
  };
  // The bookkeeper waves meet between the phases of the replay on counters in LDS (the workgroup's barrier would stop the chains'
  // waves, which need nothing of the replay before the moves are committed): the waves of `arrives` count themselves in sflag[slot],
  // those of `waits` go on when the count -- kept over the steps -- has reached `target`.
  auto replay_sync = [&](int slot, int target, bool arrives, bool waits) {
    if (!arrives) return;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      __hip_atomic_fetch_add(&sflag[slot], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (waits)
        while (__hip_atomic_load(&sflag[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(0);
    }
    __builtin_amdgcn_wave_barrier();
  };
  // ptmo_chunk_prefix's two levels, with the operands fetched from LDS BEFORE the chain of dependent adds (a lone lane pays ~100
  // cycles per dependent LDS read): chunk q of v[0 .. n): P[k] = the sum of the chunk's entries before k, returns the chunk's total ...
  auto chunk_scan = [&](const double* v, double* P, int q, int n) -> double {
    double loc = 0.0;
#pragma unroll 1
    for (int j0 = 0; j0 < 32; j0 += 8) {   // (eight operands in flight: the kernel is short of registers, not of LDS bandwidth)
      double r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int k = 32 * q + j0 + j; r[j] = k < n ? v[k] : 0.0; }
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int k = 32 * q + j0 + j; if (k < n) P[k] = loc; loc = loc + r[j]; }   // (+ 0.0 behind the end changes nothing)
    }
    return loc;
  };
  // ... and the chunk totals tot[0 .. nq) left to right: off[q] = the sum of the totals before q (may overwrite tot), returns the grand total
  auto totals_scan = [&](const double* tot, double* off, int nq) -> double {
    double run = 0.0;
    int q0 = 0;
#pragma unroll 1
    for (; q0 + 8 <= nq; q0 += 8) {   // (whole groups of eight without a test per element; the last few one by one)
      double r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = tot[q0 + j];
#pragma unroll
      for (int j = 0; j < 8; ++j) { const double t = run; run = run + r[j]; r[j] = t; }
#pragma unroll
      for (int j = 0; j < 8; ++j) off[q0 + j] = r[j];
    }
    for (; q0 < nq; ++q0) { const double t = tot[q0]; off[q0] = run; run = run + t; }
    return run;
  };
  int done = 0, nslow = 0;
  bool aborted = false;
  for (int s = 0; s < a.nsteps; ++s) {
    const uint64_t step = p.step + (uint64_t)s;
    const int par = s & 1;
    const bool last_step = s == a.nsteps - 1;
    // ---- 1. publish the state this step starts from, raise the flag.  Every published word and every flag is written and read
    //      with agent-scope atomics (sc1 accesses: coherent across the XCDs' L2s by themselves, no cache-wide write-back or
    //      invalidate); a wave's stores have completed before it reaches the barrier, the flag goes out after the barrier
    //      (tools/probes/flag_pingpong_probe.hip: 0.8-1.2 us per hand-over, against 0.4 us for every __threadfence on top)
    PTM_LADDER_TICK(0);
    if (live) {
      __hip_atomic_store(&a.pub_x[par * NcDP + (size_t)c * DP + d], xd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (lead) {
        __hip_atomic_store(&a.pub_ll[(size_t)par * p.Nc + c], ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&a.pub_lp[(size_t)par * p.Nc + c], lp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&a.flags[blk], s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // evolving ladders: every workgroup reads every rung's llike every step -- the llike travels WITH its stamp (this launch, this step),
    // stored after the barrier like the flag: a reader that finds the stamp knows the rung's row is published too
    const double stamp = (double)a.seq * 16777216.0 + (double)(s + 1);
    if (EV && live && lead) lad_store16(reinterpret_cast<lad_d2*>(a.pub_s) + (size_t)par * p.Nc + c, lad_d2{ll, stamp});
    PTM_LADDER_TICK(1);

    // ---- 2. the step in four segments, each ended by a workgroup barrier.  The bookkeepers (waves 4..7) replay the candidate draws
    //      (chain.cc:1410-1416) and the survivor filter (:1417-1418) of the whole ladder -- decide_body's, verbatim --, wait for the
    //      neighbours' flags, fetch the window and take the trials; the chains' waves (0..3) meanwhile do MH_chain::step
    //      (chain.cc:966-1022) for every rung -- which needs nothing of the neighbours, and of the draws only WHICH rungs an exchange
    //      touches (they make no Metropolis move: their lanes run along and the result is dropped when the moves are committed).
    //      Two waves share each SIMD; a lone wave leaves more than half of a SIMD's issue slots idle (tools/probes/valu_cost_probe.hip).
    int fl_lo = s + 1, fl_hi = s + 1;
    auto ask_flags = [&] {
      if (lane == 0) {
        if (b > 0) fl_lo = __hip_atomic_load(&a.flags[blk - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b + 1 < NB) fl_hi = __hip_atomic_load(&a.flags[blk + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    };
    auto flags_up = [&] {
      const bool up = __builtin_amdgcn_readfirstlane(fl_lo) >= s + 1 && __builtin_amdgcn_readfirstlane(fl_hi) >= s + 1;
      asm volatile("" ::: "memory");   // (the window's loads stay behind this look at the flags)
      return up;
    };
    // the window: rows, llikes and lpriors of rungs wlo .. whi as published for this step, NWR words per bookkeeper thread
    constexpr int NWR = (WNMAX * DP + 2 * WNMAX + 127) / 128;
    double wr[NWR];
    auto ask_window = [&] {
      const double* px = a.pub_x + par * NcDP;
      const double* pl = a.pub_ll + (size_t)par * p.Nc;
      const double* pp = a.pub_lp + (size_t)par * p.Nc;
#pragma unroll
      for (int q = 0; q < NWR; ++q) {
        const int i = wt + 128 * q;
        const double* src = nullptr;
        if (i < WN * DP) src = px + (size_t)((wlo + i / DP) * p.W + w) * DP + i % DP;
        else if (i < WN * DP + WN) src = pl + (size_t)(wlo + i - WN * DP) * p.W + w;
        else if (i < WN * DP + 2 * WN) src = pp + (size_t)(wlo + i - WN * DP - WN) * p.W + w;
        wr[q] = src ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
      }
    };
    double xn = 0.0, newlike = 0.0, newlprior = 0.0, newlpost = 0.0, off = 0.0;
    bool accept = false;
    int type = 0, axis = -1, kmix = 0;   // GENX: the step's proposal type code, its one-dimensional move's axis, its mixture member
    double mix_scale = 1.0;
    u32x4 o0 = u32x4{0u, 0u, 0u, 0u}, o = u32x4{0u, 0u, 0u, 0u};
    bool valid = false;
    double cur_lpost = 0.0;
    bool de_move = false;        // DEB: the step's member is differential evolution; its log-Hastings ratio, the error bits of its draw
    double de_hast = 0.0;
    int de_err = 0;
    // the part of MH_chain::step that depends on the rung's temperature (chain.cc:973,980-1001)
    auto metropolis = [&] {
      const double bl = beta * ll;
      cur_lpost = lp + bl;
      const double oldlprior = cur_lpost - bl;                    // chain.cc:973
      const bool want_like = valid && (newlprior > -1e200 || newlprior - oldlprior > p.min_prior);   // chain.cc:980 (Q1)
      newlpost = newlike * beta + newlprior;
      if (!want_like) newlike = newlpost = -__builtin_inf();
      double logH = newlpost - cur_lpost;
      accept = valid;
      if (DEB && de_move) {                                       // chain.cc:989-994: prop.log_hastings_ratio(), NaN => reject
        if (de_hast != de_hast) accept = false;
        logH = de_hast + logH;
      }
      if (accept && logH < 0) accept = dlog_u01(o0.v0) < logH;    // chain.cc:998-1001 (NaN stays accepted)
    };

    // -- segment A: candidate draws | the chains' random blocks
    if (helper) {   // (all four bookkeeper waves draw; the window's two have asked for the neighbours' flags and look at them afterwards)
      if (wrole && !EV) ask_flags();
      for (int k = ht; k < ms; k += 256) {
        const u32x4 oc = draw_block(p.seed, TAG_PT, (uint32_t)(w + p.w_off), step, (uint32_t)k);
        int n = -2;
        if (Nt > 1 && (uint64_t)oc.v0 < try_below) n = u01_times(oc.v1, Nt - 1);   // u01(v0) < thresh, (int)(u01(v1) (Nt - 1)): ptm_device_math.hpp
        cand[k] = n;
        ua[k] = oc.v2;
        alive[k] = 0;
        if (n >= 0) atomicMin(&first[n], k);
      }
    } else {
      o0 = draw_block(p.seed, TAG_MH, stream, step, 0);
      o = draw_block(p.seed, TAG_MH, stream, step, (uint32_t)((d >> 2) + 1));
      if (GENX) {   // lanes_body's, verbatim (proposal_distribution_set::draw, proposal_distribution.cc:99-129; gaussian_prop::draw, .hh:196-206)
        double f = p.onedfrac[rg];
        if (p.mix_K > 0) {
          const double* mx = p.mix + (size_t)rg * p.mix_K * 3;
          const double xs = p.mix_K > 1 ? u01(o0.v3) : 0.0;
          kmix = p.mix_K - 1;
          for (int k = p.mix_K - 2; k >= 0; --k)
            if (xs < mx[3 * k]) kmix = k;
          // differential evolution that is not ready yet is passed over (proposal_distribution.cc:111)
          if (DEB && mx[3 * kmix + 1] < 0 && kmix + 1 < p.mix_K && !de_ready(p, nhist)) kmix += 1;
          mix_scale = mx[3 * kmix + 1];
          f = mx[3 * kmix + 2];
        }
        if (p.any_oned && f > 0 && u01(o0.v1) < f) { axis = (int)(p.D * u01(o0.v2)); type = 1; }
        de_move = DEB && mix_scale < 0;
      }
    }
    replay_sync(5, 4 * (s + 1), helper, drole);   // every bookkeeper wave has drawn: the two replay waves go on when all four have
    PTM_LADDER_TICK(2);

    // -- segment B: survivor filter | the proposal's offset = factor . z of this lane's rung (gaussian_prop::draw, proposal_distribution.hh:194-218)
    if (EV && wrole) {
      // evolving ladders: every step takes the WHOLE ladder's llikes: the two window waves fetch the stamped words of all rungs and
      // look again at those that were not there yet (one round trip where a flag and then the data would be two)
      {
        // (stamp first, value second: a wave's loads return in issue order and the word was stored whole, so a value read behind its
        //  matching stamp is that step's; eight rungs per lane in flight at once)
        const double* ps = a.pub_s + 2 * (size_t)par * p.Nc;
        const long long t0 = wall_clock64();
        bool ok = true;
        for (int rb = 0; rb < Nt && ok; rb += 128 * 8) {
          unsigned int pending = 0;
#pragma unroll
          for (int q = 0; q < 8; ++q) if (rb + wt + 128 * q < Nt) pending |= 1u << q;
          while (pending && ok) {
            double st[8], vx[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) if ((pending >> q) & 1u) st[q] = __hip_atomic_load(ps + 2 * ((size_t)(rb + wt + 128 * q) * p.W + w) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int q = 0; q < 8; ++q) if ((pending >> q) & 1u) vx[q] = __hip_atomic_load(ps + 2 * ((size_t)(rb + wt + 128 * q) * p.W + w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int q = 0; q < 8; ++q)
              if (((pending >> q) & 1u) && st[q] == stamp) { const int r = rb + wt + 128 * q; llall[r] = vx[q]; permall[r] = r; pending &= ~(1u << q); }
            if (pending) {
              if (__hip_atomic_load(&a.ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || wall_clock64() - t0 > a.spin_limit) ok = false;
              __builtin_amdgcn_s_sleep(1);
            }
          }
        }
        if (!ok) { __hip_atomic_store(&a.ctl[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); sflag[1] = 1; }
        asm volatile("" ::: "memory");
      }
    } else if (wrole) {
      // the window, as soon as both neighbours have published (their flags were asked for a segment ago; a wave that finds them
      // down waits for them here)
      if (!flags_up()) {
        if (lane == 0) {
          const long long t0 = wall_clock64();
          const bool ok = wait_for_neighbours(s, t0);
          if (!ok) { __hip_atomic_store(&a.ctl[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); sflag[1] = 1; }
        }
        __builtin_amdgcn_wave_barrier();
      }
      ask_window();
    }
    if (drole) {
      for (int k = ht; k < ms; k += 128) {
        const int n = cand[k];
        if (n < 0 || first[n] != k) continue;                      // repeated rung value: dropped
        if (n > 0 && first[n - 1] != NONE) continue;               // not a run head
        bool al = true;
        for (int m = n;; ++m) {
          alive[first[m]] = al ? 1 : 0;
          if (m + 1 > Nt - 2 || first[m + 1] == NONE) break;
          al = !(al && first[m] < first[m + 1]);
        }
      }
    } else if (!helper) {
      const bool hi = (d & 2) != 0;
      double z0, z1;
      boxmuller(hi ? o.v2 : o.v0, hi ? o.v3 : o.v1, lds_all, z0, z1);
      double zd = (d & 1) ? z1 : z0;
      if (GENX && axis >= 0 && d != axis) zd = 0.0;   // one-dimensional move (proposal_distribution.hh:197-205)
      if (KIND == KIND_DIAG) off = tcol[0] * zd;
      else {
        vbuf[g * DP + d] = zd;
        sync_wave();
        double acc = 0.0;
        if constexpr (DP <= 8) {   // the shared column order (ptmo_column_order): natural up to 8 dimensions ...
#pragma unroll
          for (int j = 0; j < DP; ++j) acc = __builtin_fma(tcol[KIND == KIND_DIAG ? 0 : j], vbuf[g * DP + j], acc);
        } else {                   // ... else halves of 16 columns, inside a half s + 4k with s outer, k inner
#pragma unroll
          for (int h = 0; h < DP / 16; ++h)
#pragma unroll
            for (int sl = 0; sl < 4; ++sl)
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const int j = 16 * h + 4 * k + sl;
                acc = __builtin_fma(tcol[KIND == KIND_DIAG ? 0 : j], vbuf[g * DP + j], acc);
              }
        }
        off = acc;
        sync_wave();
      }
      if (GENX && p.mix_K > 0) {
        type = kmix + 10 * type;        // proposal_distribution.cc:117
        off = mix_scale * off;          // the member is scale_k times the rung's factor
      }
    }
    replay_sync(6, 2 * (s + 1), drole, drole);    // the filter is done
    PTM_LADDER_TICK(3);

    // -- segment C: long runs, the picks' log-uniforms, flags, the window | prior box, likelihood, the Metropolis test
    if (EV && drole) {
      // evolving ladders: the log of every surviving pick's accept uniform, the step's gaps (chain.cc:1816) with their prefix sums
      // in the checker's order (ptmo_chunk_prefix: chunks of 32 left to right, then the chunk totals), the surviving picks in pick order
      const int nch = (Nt - 1 + 31) / 32;
      for (int k = ht; k < ms; k += 128)
        if (alive[k]) lul[k] = dlog_u01(ua[k]);
      for (int k = ht; k < Nt - 1; k += 128) spl[k] = bwl[k] - bwl[k + 1];
      replay_sync(7, 2 * (s + 1), true, true);
      for (int q = ht; q < nch; q += 128) cts[q] = chunk_scan(spl, P0l, q, Nt - 1);
      replay_sync(4, 2 * (s + 1), true, true);
      if (ht == 0) cts[nchunk + 1] = totals_scan(cts, cts + nchunk + 2, nch);   // S at the start of the step
      if (ht < 64) {
        int cnt = 0;
        double v = 0.0;   // the increases of ALL surviving picks: how far the normaliser can move this step
        for (int base = 0; base < ms; base += 64) {
          const int k = base + ht;
          const bool al = k < ms && alive[k];
          const unsigned long long m = __builtin_amdgcn_ballot_w64(al);
          if (al) { plist[cnt + __builtin_popcountll(m & ((1ull << ht) - 1ull))] = k; const double gq = spl[cand[k]]; v += gq * grow - gq; }
          cnt += __builtin_popcountll(m);
        }
#pragma unroll
        for (int o_ = 32; o_ > 0; o_ >>= 1) v += __shfl_down(v, o_);
        if (ht == 0) { evi[0] = cnt; evi[2] = 0; cts[2 * nchunk + 2] = v; }
      }
    } else if (drole) {
      // a run of more than H surviving picks on consecutive rungs anywhere in the ladder: the halos do not cover this step (every
      // workgroup of the ladder sees the same draws): it takes the whole-ladder form below
      for (int k = ht; k < ms; k += 128) {
        const int n = cand[k];
        if (n < 0 || !alive[k] || PTM_LADDER_ALIVE(n - 1)) continue;   // bottoms of runs of surviving picks
        int len = 1;
        while (PTM_LADDER_ALIVE(n + len)) ++len;
        if (len > a.max_run) sflag[0] = 1;
      }
      // log of the accept uniform of every surviving pick in the window (the uniform's slot is the pick's whether needed or not: Q5),
      // taken off the chain of dependent trials
      if (ht < WN - 1) {
        const int n = wlo + ht;
        if (PTM_LADDER_ALIVE(n)) wlu[n - wlo] = dlog_u01(ua[first[n]]);
      }
    }
    if (wrole && !EV) {   // the window into LDS
#pragma unroll
      for (int q = 0; q < NWR; ++q) {
        const int i = wt + 128 * q;
        if (i < WN * DP) wx[i] = wr[q];
        else if (i < WN * DP + WN) { wll[i - WN * DP] = wr[q]; wll0[i - WN * DP] = wr[q]; wperm[i - WN * DP] = wlo + i - WN * DP; }
        else if (i < WN * DP + 2 * WN) wlp0[i - WN * DP - WN] = wr[q];
      }
    }
    if (!helper) {
      xn = xd + off;                                              // state::add (states.cc:205-214)
      if (DEB && __builtin_amdgcn_ballot_w64(de_move) != 0ull) {
        // differential_evolution::draw (proposal_distribution.cc:476-592) -- lanes_body's, with the state in its register and the
        // history rows read past the L1 (this workgroup wrote the newest of them itself, some steps ago in this launch).  A chain that
        // turns out to be touched by the exchange phase drops the result like any other move's, and its error bits with it.
        auto hload = [](const double* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        const int D = p.D;
        const long long saved = 1 + (long long)((nhist + every - 1u) / every);   // rows of the ring so far
        const long long rows = p.de_init_extra + saved;
        auto pick = [&](double u) -> const double* {
          const long long spare = rows - 100ll * D;
          const long long first_ = (spare * (1 - p.de_ignore) > 10ll * D) ? (long long)(spare * p.de_ignore) : 0;
          const long long r = (long long)(first_ + (rows - first_) * u);
          if (r < p.de_init_extra) return p.de_init + ((size_t)r * p.Nc + c) * DP;
          const long long hr = r - p.de_init_extra;
          if (saved - hr > p.hist.cap) de_err |= 64;         // the ring has lost that row
          return p.hist.x + hist_slot(p.hist, hr, c) * DP;
        };
        const u32x4 b0 = draw_block(p.seed, TAG_MH, stream, step, 0x0DE00000u);
        const bool snk = de_move && p.de_snooker > u01(b0.v0);
        double z1d = 0.0, z2d = 0.0, xde = 0.0;
        if (de_move) { z1d = hload(pick(u01(b0.v2)) + pos); z2d = hload(pick(u01(b0.v3)) + pos); }
        if (de_move && !snk) {                               // draw_standard
          const double gamma = u01(b0.v1) < p.de_gamma_one ? 1.0 : p.de_gamma_std;
          const double t1 = z1d * gamma;
          const double a_ = xd + t1;
          const double t2 = z2d * (-gamma);
          xde = d < D ? a_ + t2 : 0.0;
        }
        int dt = 0;
        if (__builtin_amdgcn_ballot_w64(snk) != 0ull) {      // draw_snooker (the whole wave walks along: the LDS hand-overs are the wave's)
          const double gamma = (1.2 + u01(b0.v1)) / p.de_gamma_div;
          auto chain_sum = [&](double term, bool mine, double& out) {   // the chain's sum of its lanes' terms, in index order
            sbuf[g * DP + d] = term;
            sync_wave();
            if (mine) {
              double sm = 0.0;
              for (int j = 0; j < D; ++j) sm = sm + sbuf[g * DP + j];
              out = sm;
            }
            sync_wave();
          };
          double zz = 0.0, ax = 0.0, axis2 = 0.0;
          bool failed = false;
          for (int tries = 0;; ++tries) {                    // the history repeats states: z must differ from the current state
            bool need = snk && !failed && axis2 == 0.0;
            if (need && tries > 1000) { failed = true; need = false; }
            if (__builtin_amdgcn_ballot_w64(need) == 0ull) break;
            if (need) {
              const u32x4 bt = draw_block(p.seed, TAG_MH, stream, step, 0x0DE00001u + (uint32_t)tries);
              zz = hload(pick(u01(bt.v0)) + pos);
              ax = xd + zz * (-1.0);
            }
            chain_sum(ax * ax, need, axis2);
          }
          const bool go = snk && !failed;
          double proj = 0.0, fz2 = 0.0;
          {
            const double a_ = z1d * gamma, b_ = z2d * (-gamma);
            const double diff = a_ + b_;
            chain_sum(diff * ax, go, proj);
          }
          if (go) proj = proj / axis2;
          const double t = ax * proj;
          const double y = xd + t;
          if (go) xde = d < D ? y : 0.0;
          const double f_ = y + zz * (-1.0);
          chain_sum(f_ * f_, go, fz2);
          if (go) de_hast = (dlog(fz2) - dlog(axis2)) * (double)(D - 1) / 2.0;
          if (snk && failed) {                               // (the reference exits here; the engine raises an error bit and rejects the move)
            de_err |= 128;
            de_hast = __builtin_nan("");
            xde = d < D ? xd : 0.0;
          }
          if (snk) dt = 1;
        }
        if (de_move) { xn = xde; type = kmix + 10 * dt; }    // proposal_distribution.cc:117
      }
      if constexpr (GENS) {
        // lanes_body's general form, each dimension on its lane: stateSpace::enforce (states.cc:86-102; Q9: the sum is built on an enforced
        // zero state), then the box of an all-uniform prior or mixed_dist_product::evaluate -- the factors in four interleaved partial
        // products, combined ((p0 p1) p2) p3
        bool vd = true;
        if (p.has_bounds && d < p.D) vd = boundary_enforce(p.blo[d], p.bhi[d], p.bmin[d], p.bmax[d], xn);
        valid = p.origin_valid != 0 && all_of_chain(vd);
        if (p.all_uniform) {
          const bool in = all_of_chain(!(xn < plo) && !(xn > phi));
          newlprior = (valid && in) ? p.lprior_const : -__builtin_inf();
        } else {
          sbuf[g * DP + d] = d < p.D ? prior_pdf(p.ptype[d], plo, phi, p.pcoef[d], xn) : 1.0;
          sync_wave();
          if (d < 4) {
            double pq = 1.0;
#pragma unroll
            for (int t = 0; t < DP / 4; ++t) pq *= sbuf[g * DP + d + 4 * t];
            pbuf[g * 4 + d] = pq;
          }
          sync_wave();
          const double result = ((pbuf[g * 4 + 0] * pbuf[g * 4 + 1]) * pbuf[g * 4 + 2]) * pbuf[g * 4 + 3];
          sync_wave();   // (sbuf / pbuf are reused by the likelihood)
          newlprior = valid ? dlog(result) : -__builtin_inf();
        }
      } else {
        const bool ind = !(xn < plo) && !(xn > phi);
        const bool in = all_of_chain(ind);
        // Q9: state::add builds on an enforced zero state -- an origin outside a `limit` bound invalidates every proposal
        valid = p.origin_valid != 0 && all_of_chain(!(xn < elo) && !(xn > ehi));   // stateSpace::enforce, states.cc:86-102
        newlprior = in ? p.lprior_const : -__builtin_inf();
        if (!valid) newlprior = -__builtin_inf();
      }
      vbuf[g * DP + d] = xn - mean_d;
      sync_wave();
      {
        // s_d = sum_{j<=d} P2_dj y_j, one fma chain with j ascending (gauss_llike's order); the terms behind the diagonal add
        // fma(0, y_j, s) = s exactly for finite y_j, and change nothing -- they keep the loop free of a lane-dependent trip count,
        // so that the operand reads run ahead of the chain.  A non-finite proposal never gets here with a wanted likelihood:
        // the prior box has rejected it (want_like false => the likelihood is dropped below).
        const double* y = vbuf + g * DP;
        double sacc = 0.0;
#pragma unroll
        for (int j = 0; j < DP; ++j) sacc = __builtin_fma(PROW_LDS ? prl[j] : prow[PROW_LDS ? 0 : j], y[j], sacc);
        sbuf[g * DP + d] = sacc;
      }
      sync_wave();
      if (d < 4) {
        double pq = 0.0;
#pragma unroll
        for (int t = 0; t < DP / 4; ++t) pq = __builtin_fma(vbuf[g * DP + d + 4 * t], sbuf[g * DP + d + 4 * t], pq);
        pbuf[g * 4 + d] = pq;
      }
      sync_wave();
      const double quad = ((pbuf[g * 4 + 0] + pbuf[g * 4 + 1]) + pbuf[g * 4 + 2]) + pbuf[g * 4 + 3];
      sync_wave();   // (vbuf / pbuf are rewritten by the next step's draw)
      newlike = p.like0 - 0.5 * quad;
      if (!EV) metropolis();   // (an evolving ladder's test waits for the temperature this step's exchange phase leaves the rung with)
    }
    __syncthreads();
    if (sflag[1]) { aborted = true; break; }
    PTM_LADDER_TICK(4);
    // rungs an exchange attempt touches make no Metropolis move this step, one add_state per attempt (chain.cc:1487-1490,
    // 1531-1534,1553-1557): known from the draws alone
    const int tc = helper ? 0 : (PTM_LADDER_ALIVE(rg) ? 1 : 0) + (PTM_LADDER_ALIVE(rg - 1) ? 1 : 0);
    const unsigned int nh0 = nhist;   // add_state calls before this step's
    if (DEB && !helper && !tc && live && de_err) atomicOr(p.err, de_err);
    const double beta_old = beta;     // (evolving ladders: the rung's temperature before this step's pries)
    // evolving ladders: the row an exchanged rung receives (and the one it held in between) are asked for as soon as the row map is
    // known, and taken when the temperatures and the Metropolis tests are done
    double ex_x = 0.0, ex_ll = 0.0, ex_lp = 0.0, exm_x = 0.0, exm_ll = 0.0, exm_lp = 0.0;
    int ex_src = rg;
    if (EV) {
      // ---- the exchange phase of an evolving ladder.  The trials are one chain only through the normaliser S, and S moves by
      //      rate x (the pried gaps) -- a relative 1e-3 over a step.  lu * S is monotone in S: a trial that gives the same answer by
      //      the first-trial form (nothing pried yet: no S) and by the pried form at BOTH ends of [S0, S0 + every increase a surviving
      //      pick could add] gives that answer whatever came before it.  So every run of surviving picks is walked by its own lane,
      //      top-down as on a fixed ladder; a step with a trial INSIDE that window (one in fifty at 1024 rungs) is walked in pick
      //      order by one lane instead -- the checker's swap_phase operation for operation.  Then the pries, in pick order.
      const double c1 = 1 - bwl[Nt - 1];                           // chain.cc:1833
      const double S0 = cts[nchunk + 1];
      const int np = evi[0];
      // the largest S any order of acceptances could reach: S0 + the increases of ALL surviving picks (+ a margin for the roundings)
      const double Shi = (S0 + cts[2 * nchunk + 2]) * (1.0 + 1e-9);
      // decisions, a lane per run (no side effects yet): accb[k]
      unsigned char* accb = reinterpret_cast<unsigned char*>(evi + 8);   // [ms]
      auto decide3 = [&](int k, double lla, double llb, bool& acc) -> bool {   // false: the answer depends on what was pried before
        const int i = cand[k];
        const double gq = spl[i], dl = llb - lla, lu = lul[k];
        const double logH = gq * dl;                                // chain.cc:1463 with the stored temperatures' difference
        const bool a0 = logH < 0 ? lu < logH : true;
        const double tt = (gq * c1) * dl;
        const bool a1 = tt < 0 ? lu * S0 < tt : true;
        const bool a2 = tt < 0 ? lu * Shi < tt : true;
        acc = a0;
        return a0 == a1 && a1 == a2;
      };
      for (int k = tid; k < ms; k += LADDER_THREADS) {
        const int n = cand[k];
        if (n < 0 || !alive[k] || PTM_LADDER_ALIVE(n + 1)) continue;   // tops of runs of surviving picks
        double up = llall[n + 1];                                   // the upper rung's llike as the pick sees it
        for (int i = n; i >= 0; --i) {
          const int kk = first[i];
          double lla = llall[i];
          if (!(lla > -1e200)) lla = -1e200;
          double llb = up;
          if (!(llb > -1e200)) llb = -1e200;
          bool acc;
          if (!decide3(kk, lla, llb, acc)) evi[2] = 1;
          accb[kk] = acc ? 1 : 0;
          up = acc ? up : llall[i];                                 // what rung i holds after the trial: the next pick's upper rung
          if (!PTM_LADDER_ALIVE(i - 1)) break;
        }
      }
      __syncthreads();
      if (evi[2]) {
        // the rare step: in pick order, one lane (llall is exchanged as the picks are decided)
        if (tid == 256) {
          double S = S0;
          int npry = 0;
          for (int t = 0; t < np; ++t) {
            const int k = plist[t], i = cand[k];
            double lla = llall[i];
            if (!(lla > -1e200)) lla = -1e200;
            double llb = llall[i + 1];
            if (!(llb > -1e200)) llb = -1e200;
            bool acc = true;
            if (npry) {
              const double tt = (spl[i] * c1) * (llb - lla);
              if (tt < 0) acc = lul[k] * S < tt;
            } else {
              const double logH = spl[i] * (llb - lla);
              if (logH < 0) acc = lul[k] < logH;
            }
            accb[k] = acc ? 1 : 0;
            if (acc) {
              const double tl = llall[i]; llall[i] = llall[i + 1]; llall[i + 1] = tl;
              S = S + (spl[i] * grow - spl[i]);                     // (spl itself is pried below, with everybody's)
              npry++;
            }
          }
        }
        __syncthreads();
      }
      // the exchanges applied, a lane per run: the row map, the in-between rows, the own pairs' counters and log lines
      for (int k = tid; k < ms; k += LADDER_THREADS) {
        const int n = cand[k];
        if (n < 0 || !alive[k] || PTM_LADDER_ALIVE(n + 1)) continue;
        for (int i = n; i >= 0; --i) {
          const int kk = first[i];
          const bool acc = accb[kk] != 0;
          if (acc) { const int q = permall[i]; permall[i] = permall[i + 1]; permall[i + 1] = q; }
          if (HIST && PTM_LADDER_ALIVE(i - 1)) midall[i] = permall[i];
          if (i >= r0 && i < r1) {
            ptry[i - r0] += 1;
            if (acc) pacc[i - r0] += 1;
            if (last_step) a.swap_log[(size_t)w * ms + kk] = i | (acc ? 0x40000000 : 0);
          }
          if (!PTM_LADDER_ALIVE(i - 1)) break;
        }
      }
      // the pries in pick order (chain.cc:1829): pair and increase of the q-th accepted pick; per pick the number of pries before it
      if (tid < 64) {
        int cnt = 0;
        for (int base = 0; base < np; base += 64) {
          const int t = base + tid;
          const int k = t < np ? plist[t] : 0;
          const bool ac = t < np && accb[k] != 0;
          const unsigned long long m = __builtin_amdgcn_ballot_w64(ac);
          const int q = cnt + __builtin_popcountll(m & ((1ull << tid) - 1ull));
          if (t < np) knp[k] = q;
          if (ac) {
            const int i = cand[k];
            const double gq = spl[i], sn = gq * grow;
            ipry[q] = i; incl[q] = sn - gq; spl[i] = sn;
          }
          cnt += __builtin_popcountll(m);
        }
        if (tid == 0) evi[1] = cnt;
      }
      __syncthreads();
      if (tc) {
        const double* pl = a.pub_ll + (size_t)par * p.Nc;
        ex_src = permall[rg];
        if (ex_src != rg) {
          const size_t cs = (size_t)ex_src * p.W + w;
          ex_x = __hip_atomic_load(a.pub_x + par * NcDP + cs * DP + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ex_ll = __hip_atomic_load(pl + cs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ex_lp = __hip_atomic_load(a.pub_lp + (size_t)par * p.Nc + cs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (HIST && tc == 2) {
          const size_t cm = (size_t)midall[rg] * p.W + w;
          exm_x = __hip_atomic_load(a.pub_x + par * NcDP + cm * DP + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          exm_ll = __hip_atomic_load(pl + cm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          exm_lp = __hip_atomic_load(a.pub_lp + (size_t)par * p.Nc + cm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (HIST && tid == 256) {   // the normaliser after each pry, summed in pick order (kSl[q]: after pry q): what the in-phase temperatures need
        double S = S0;
        const int nq = evi[1];
        // (a lone lane issues an instruction every ~10 cycles whatever it is: whole groups of eight without a test per element -- paired
        //  LDS reads, eight additions, paired writes --, the last few one by one)
        int q0 = 0;
#pragma unroll 1
        for (; q0 + 8 <= nq; q0 += 8) {
          double ic[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) ic[j] = incl[q0 + j];
#pragma unroll
          for (int j = 0; j < 8; ++j) { S = S + ic[j]; ic[j] = S; }
#pragma unroll
          for (int j = 0; j < 8; ++j) kSl[q0 + j] = ic[j];
        }
        for (; q0 < nq; ++q0) { S = S + incl[q0]; kSl[q0] = S; }
      }
      PTM_LADDER_TICK(5);
      if (evi[1] > 0) {   // the new temperatures (chain.cc:1834-1844): beta_k = 1 - P_k / (total / (1 - beta_last)), P in the checker's order
        const int nch = (Nt - 1 + 31) / 32;
        double* Pn = llall;   // (the llike view is done with)
        for (int q = tid; q < nch; q += LADDER_THREADS) cts[q] = chunk_scan(spl, Pn, q, Nt - 1);
        __syncthreads();
        if (tid == 0) cts[nchunk] = totals_scan(cts, cts, nch) / (1 - bwl[Nt - 1]);
        __syncthreads();
        const double nn = cts[nchunk];
        for (int k = 1 + tid; k < Nt - 1; k += LADDER_THREADS) bwl[k] = 1 - (cts[k >> 5] + Pn[k]) / nn;
        __syncthreads();
        beta = bwl[rg];
      }
      if (!helper) metropolis();
    }
    if (!helper) {
      if (!tc) {
        ntries += 1;
        nhist += 1u;
        if (HIST) {
          // this add_state saves a row (chain.cc:935-946): the proposal if it was accepted, else the state as it stands
          if (hist_on && nh0 % every == 0u) {
            const long long hrow = 1 + (long long)(nh0 / every);
            const size_t o = hist_slot(p.hist, hrow, c);
            if (live) p.hist.x[o * DP + pos] = accept ? xn : xd;
            if (live && lead) {
              if (accept) hist_scalars(p.hist, o, hrow, newlike, newlprior, naccept + 1, ntries, GENX ? type : 0, beta);
              else hist_scalars(p.hist, o, hrow, ll, lp, naccept, ntries, last_type, beta);
            }
          }
          if (map_on) {
            int mapw = 0;
            if (accept) mapw = map_take(newlpost, newlike, newlprior) ? 1 : 0;
            // an evolving ladder: the state that stays is added at a NEW temperature and may beat the MAP with it
            else if (EV) mapw = map_take(cur_lpost, ll, lp) ? 2 : 0;
            if (mapw && live) p.map.x[(size_t)c * DP + pos] = mapw == 1 ? xn : xd;
          }
        }
        if (accept) { xd = xn; ll = newlike; lp = newlprior; naccept += 1; last_type = GENX ? type : 0; }
      } else nhist += (unsigned int)tc;
    }
    // the add_state calls of a rung the exchange phase touched (one per attempt; the rung makes no Metropolis move): history and MAP
    // see the row the rung holds at each call -- the in-between row `mid` at the first of two (its scalars, the rung's own counters)
    // (evolving ladders: each add sees the temperature its rung had THEN -- before the pick's own pry, after the earlier ones:
    //  1 - (P0 + D) / normaliser with P0 the prefix sum of the step's first gaps, D what the earlier pries added to the gaps below the
    //  rung, in pick order; the ladder's ends never move.  The checker's swap_phase, chain.cc:1487-1490,1531-1534)
    auto beta_at = [&](int k) -> double {
      if (!EV) return beta;
      const int np_ = knp[k];
      if (np_ == 0 || rg == 0 || rg == Nt - 1) return beta_old;
      // (the operands eight pries ahead of the chain of additions: a lone dependent LDS read costs ~100 cycles, and a 1024-rung ladder
      //  pries ~80 times a step; a pry at or above the rung adds +0.0, which changes nothing.  One walk for both temperatures of a rung
      //  exchanged twice -- the earlier sum is a prefix of the later -- was tried: its two snapshot tests per pry cost more than the
      //  second walk, 20.1 -> 21.4 us per step)
      double Dr = 0.0;
      int q0 = 0;
#pragma unroll 1
      for (; q0 + 8 <= np_; q0 += 8) {   // (whole groups of eight without a test for the end, the last few one by one)
        int ip[8];
        double ic[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { ip[j] = ipry[q0 + j]; ic[j] = incl[q0 + j]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) Dr = Dr + (ip[j] < rg ? ic[j] : 0.0);
      }
      for (; q0 < np_; ++q0) Dr = Dr + (ipry[q0] < rg ? incl[q0] : 0.0);
      const double nrm = kSl[np_ - 1] / (1 - bwl[Nt - 1]);   // S after the pries before this pick
      return 1 - ((cts[nchunk + 2 + (rg >> 5)] + P0l[rg]) + Dr) / nrm;
    };
    auto exchanged_adds = [&](double xmid, double llmid, double lpmid) {
      if (!HIST || !tc) return;
      // the pick on the pair above comes first in pick order whenever both survive (chain.cc:1417-1418)
      const double bmid = tc == 2 ? beta_at(first[rg]) : 0.0;
      const double beta = beta_at(PTM_LADDER_ALIVE(rg - 1) ? first[rg - 1] : first[rg]);   // (shadows the chain's: the LAST add's temperature)
      if (tc == 2) {
        if (hist_on && nh0 % every == 0u) {
          const long long hrow = 1 + (long long)(nh0 / every);
          const size_t o = hist_slot(p.hist, hrow, c);
          if (live) p.hist.x[o * DP + pos] = xmid;
          if (live && lead) hist_scalars(p.hist, o, hrow, llmid, lpmid, naccept, ntries, last_type, EV ? bmid : beta);
        }
        if (map_on) {
          const double tb = (EV ? bmid : beta) * llmid;
          if (map_take(lpmid + tb, llmid, lpmid) && live) p.map.x[(size_t)c * DP + pos] = xmid;
        }
      }
      const unsigned int al = nh0 + (unsigned int)tc - 1u;   // the LAST of the adds saw the row as it is now
      if (hist_on && al % every == 0u) {
        const long long hrow = 1 + (long long)(al / every);
        const size_t o = hist_slot(p.hist, hrow, c);
        if (live) p.hist.x[o * DP + pos] = xd;
        if (live && lead) hist_scalars(p.hist, o, hrow, ll, lp, naccept, ntries, last_type, beta);
      }
      if (map_on) {
        const double tb = beta * ll;
        if (map_take(lp + tb, ll, lp) && live) p.map.x[(size_t)c * DP + pos] = xd;
      }
    };

    if (EV) {
      // ---- 4''. evolving ladders: the walk above has decided every pick of the ladder; the rows of the own rungs come from wherever
      //      the exchanges took them (the whole ladder's publications of this step; nobody can overwrite them before this workgroup
      //      has published its next step)
      if (tc) {
        if (ex_src != rg) { xd = ex_x; ll = ex_ll; lp = ex_lp; }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (!sflag[0]) {
      // ---- 4. the exchange phase from the neighbours' publications
      // trials (chain.cc:1436-1537): the top pick of each run of surviving picks inside the window walks it downwards
      if (helper && ht < WN - 1) {
        const int n = wlo + ht;                                    // pair (n, n + 1), both inside the window
        // A pick above that is inside the window walks this pick too.  One that lies outside (n + 1 == whi) may replace rung
        // whi, so the run below it is unknown here -- and it has at most H picks (checked above): it ends above this
        // workgroup's rungs.  Somebody else's.
        if (PTM_LADDER_ALIVE(n) && !PTM_LADDER_ALIVE(n + 1))
          for (int i = n; i >= wlo; --i) {
            trial(wll, wperm, wmid, wlo, i, last_step, wdb[i - wlo], wlu[i - wlo]);
            if (!PTM_LADDER_ALIVE(i - 1)) break;
          }
      }
      __syncthreads();
      PTM_LADDER_TICK(5);
      // the rows the exchanges brought
      if (tc) {
        const int src = wperm[rg - wlo];
        if (src != rg) { xd = wx[(src - wlo) * DP + d]; ll = wll0[src - wlo]; lp = wlp0[src - wlo]; }
        if (HIST && tc == 2) { const int sm = wmid[rg - wlo]; exm_x = wx[(sm - wlo) * DP + d]; exm_ll = wll0[sm - wlo]; exm_lp = wlp0[sm - wlo]; }
      }
    } else {
      // ---- 4'. the same from the WHOLE ladder's publications (a run of surviving picks longer than the halo: rare).  Every
      //      workgroup of the ladder waits for all of them, replays every trial of the ladder on a full llike view, takes the rows
      //      of its own rungs from wherever they come, and nobody goes on before everybody has read.
      nslow += 1;
      if (tid == 0) sflag[2] = 0;
      __syncthreads();
      {
        const long long t0 = wall_clock64();
        bool ok = true;
        for (int nb = tid; nb < NB && ok; nb += LADDER_THREADS)
          if (nb != b) ok = wait_for(nb, s, t0);
        if (!ok) sflag[2] = 1;
      }
      __syncthreads();
      if (sflag[2]) { if (tid == 0) __hip_atomic_store(&a.ctl[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); aborted = true; break; }
      const double* pl = a.pub_ll + (size_t)par * p.Nc;
      for (int r = tid; r < Nt; r += LADDER_THREADS) {
        llall[r] = __hip_atomic_load(pl + (size_t)r * p.W + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        permall[r] = r;
      }
      __syncthreads();
      for (int k = tid; k < ms; k += LADDER_THREADS) {
        const int n = cand[k];
        if (n < 0 || !alive[k] || PTM_LADDER_ALIVE(n + 1)) continue;   // tops of runs of surviving picks
        for (int i = n; i >= 0; --i) {
          trial(llall, permall, midall, 0, i, last_step, -(p.beta[i + 1] - p.beta[i]), dlog_u01(ua[first[i]]));
          if (!PTM_LADDER_ALIVE(i - 1)) break;
        }
      }
      __syncthreads();
      if (tc) {
        const int src = permall[rg];
        if (src != rg) {
          const size_t cs = (size_t)src * p.W + w;
          xd = __hip_atomic_load(a.pub_x + par * NcDP + cs * DP + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ll = __hip_atomic_load(pl + cs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          lp = __hip_atomic_load(a.pub_lp + (size_t)par * p.Nc + cs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (HIST && tc == 2) {
          const size_t cm = (size_t)midall[rg] * p.W + w;
          exm_x = __hip_atomic_load(a.pub_x + par * NcDP + cm * DP + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          exm_ll = __hip_atomic_load(pl + cm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          exm_lp = __hip_atomic_load(a.pub_lp + (size_t)par * p.Nc + cm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {   // everybody has read: only then may anybody publish again
        sflag[0] = 0;
        __hip_atomic_fetch_add(&a.slow_done[w], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(&a.slow_done[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < NB * nslow) {
          if (__hip_atomic_load(&a.ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || wall_clock64() - t0 > a.spin_limit) { sflag[2] = 1; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      __syncthreads();
      if (sflag[2]) { if (tid == 0) __hip_atomic_store(&a.ctl[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); aborted = true; break; }
    }
    // the add_state calls of the exchanged rungs (history rows, MAP): one place for the three forms of the exchange phase above
    if (HIST && tc) exchanged_adds(exm_x, exm_ll, exm_lp);
    if (last_step && b == 0)                                      // the log lines of the picks that did not survive
      for (int k = tid; k < ms; k += LADDER_THREADS)
        if (!alive[k]) a.swap_log[(size_t)w * ms + k] = -2;
    __syncthreads();   // (first / alive / window are rewritten by the next step)
    for (int k = tid; k < ms; k += LADDER_THREADS) { const int n = cand[k]; if (n >= 0) first[n] = NONE; }
    PTM_LADDER_TICK(6);
    done = s + 1;
  }
#undef PTM_LADDER_ALIVE

  // -- all or nothing (header): every workgroup has finished every step, or nothing of this launch stays
  if (tid == 0) {
    int ok = 0;
    if (aborted) __hip_atomic_fetch_or(&a.ctl[3], LADDER_ABORT_BIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else {
      const int old = __hip_atomic_fetch_add(&a.ctl[3], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!(old & LADDER_ABORT_BIT)) {
        const long long t0 = wall_clock64();
        for (;;) {
          const int v = __hip_atomic_load(&a.ctl[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (v == G) { ok = 1; break; }
          if (v & LADDER_ABORT_BIT) break;
          if (wall_clock64() - t0 > a.spin_limit) {
            // (whoever sets the bit learns from the value it replaces whether everybody had arrived after all)
            ok = __hip_atomic_fetch_or(&a.ctl[3], LADDER_ABORT_BIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == G ? 1 : 0;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
    }
    sflag[3] = ok;
  }
  __syncthreads();
  const bool commit = sflag[3] != 0;
  if (commit && live) {
    p.x[(size_t)c * DP + pos] = xd;
    if (lead) {
      p.ll[c] = ll; p.lp[c] = lp;
      p.ntries[c] = ntries; p.naccept[c] = naccept; p.last_type[c] = last_type; p.nhist[c] = nhist;
    }
  }
  if (EV && commit && b == 0)   // the ladder's temperatures (the chain-indexed image is brought up to date by the host when somebody asks for it)
    for (int k = tid; k < Nt; k += LADDER_THREADS) const_cast<double*>(p.beta_w)[(size_t)w * Nt + k] = bwl[k];
  if (commit && tid < R && r0 + tid < Nt - 1 && r0 + tid < r1) {
    long long* sc = a.swap_cnt + ((size_t)w * (Nt - 1) + (r0 + tid)) * 2;
    sc[0] += ptry[tid];
    sc[1] += pacc[tid];
  }
  if (a.prof && tid == a.prof_tid)
    for (int k = 0; k < 7; ++k) a.prof[(size_t)blk * 8 + k] = tick_sum[k];
#undef PTM_LADDER_TICK
  if (L == 0 && tid == 0) {
    a.ctl[1] = commit ? done : -1; a.ctl[2] = nslow;
    if (commit) *a.done_seq = a.seq;   // (visible to the next launch and to the host at the kernel's end)
  }
  // (a launch is asynchronous: a workgroup that gave up says so in the engine's deferred error word too; the host looks at done_seq)
  if (!commit && tid == 0) atomicOr(p.err, 32);
}

}  // namespace ptm
