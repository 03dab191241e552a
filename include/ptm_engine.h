/* include/ptm_engine.h -- C ABI of the MI355X parallel-tempering step engine.
 *
 * This is the drop-in boundary for ONE hot path of JohnGBaker/ptmcmc: chain::step()
 *   MH_chain::step                       chain.cc:966-1022
 *   MH_chain::add_state                  chain.cc:916-949
 *   parallel_tempering_chains::step      chain.cc:1393-1571
 * behind the reference's plug-in surface (bayes_likelihood / probability_function /
 * proposal_distribution, ptmcmc_sampler).  Plain C: pointers and sizes only, no C++ or
 * torch types.  All HOST pointers unless a parameter says "device".
 *
 * Conventions
 *   - every function returns PTM_OK (0) or a negative ptm_status; ptm_last_error() gives text.
 *   - a "chain" is one (rung, walker) pair.  Walkers are W independent ladders batched together
 *     (the reference's nearest notion is Nchain independent repeats, testMH.cpp:17,206).
 *   - an engine holds the contiguous rung block [rung_begin, rung_begin+rung_count) of a global ladder
 *     of n_rungs rungs, for all W walkers.  Local chain index  c = (rung - rung_begin) * W + walker.
 *     Host-visible arrays are chain-major in that order; states are [chain][dim] row-major.
 *   - random streams are keyed by GLOBAL identities (seed, global rung, walker, step), so results do
 *     not depend on how the ladder is sharded or how kernels are launched.
 */
#ifndef PTM_ENGINE_H
#define PTM_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTM_ABI_VERSION 3   /* 2: ptm_config.walker_begin (a v1 caller's shorter struct is still accepted: walker_begin = 0); 3: ptm_calibrate, ptm_get_counter_sums, ptm_get_ladder_stats (additions only) */

typedef struct ptm_engine ptm_engine;

typedef enum {
  PTM_OK = 0,
  PTM_ERR_INVALID = -1,      /* bad argument / call order */
  PTM_ERR_UNSUPPORTED = -2,  /* feature not available on the device path (caller must use the host path) */
  PTM_ERR_HIP = -3,          /* HIP runtime error (text in ptm_last_error) */
  PTM_ERR_NO_DEVICE = -4,    /* no gfx950 device / code object cannot load */
  PTM_ERR_FAR_MOVE = -5,     /* neighbour-exchange mode saw a state cross more than one shard boundary in one step */
  PTM_ERR_CALLBACK = -6
} ptm_status;

/* boundary types -- values of boundary::{open,limit,reflect,wrap}, states.hh:34-37 */
enum { PTM_BOUND_OPEN = 0, PTM_BOUND_LIMIT = 1, PTM_BOUND_REFLECT = 2, PTM_BOUND_WRAP = 3 };
/* per-dimension prior types -- values of mixed_dist_product::{uniform,gaussian,polar,copolar,log},
 * probability_function.hh:151-155; 0 = flat (base probability_function, probability_function.hh:40) */
enum { PTM_PRIOR_FLAT = 0, PTM_PRIOR_UNIFORM = 1, PTM_PRIOR_GAUSSIAN = 2, PTM_PRIOR_POLAR = 3, PTM_PRIOR_COPOLAR = 4,
       PTM_PRIOR_LOG = 5 };
/* Gaussian step proposal storage (gaussian_prop, proposal_distribution.hh:145-227) */
enum {
  PTM_PROP_DENSE = 0, /* offset = M z, M dense D x D (e.g. eigenvectors * sqrt(eigenvalues), hh:173-176,207-213) */
  PTM_PROP_DIAG = 1,  /* offset_i = sigma_i z_i  (identity_trans, hh:155-163) */
  PTM_PROP_LOWER = 2  /* offset = L z, L lower triangular (Cholesky factor); same numbers as DENSE with zeros */
};

typedef struct ptm_config {
  uint32_t struct_size;   /* sizeof(ptm_config), for ABI evolution */
  int32_t dim;            /* D, 1..1024 (PTM_ERR_UNSUPPORTED above; 33..1024: functional, not tuned) */
  int32_t n_rungs;        /* global ladder length (Ntemps) */
  int32_t rung_begin;     /* first global rung held by this engine */
  int32_t rung_count;     /* rungs held by this engine (== n_rungs on one GPU) */
  int32_t n_walkers;      /* W independent ladders */
  uint64_t seed;          /* Philox key */
  double swap_rate;       /* parallel_tempering_chains ctor arg (chain.cc:1163); maxswapsperstep = 1+2*swap_rate*n_rungs */
  int32_t add_every_n;    /* history stride (save_every), >= 1 */
  double min_prior;       /* dpriormin / minPrior, MH_chain ctor (chain.cc:647); sampler default -30 */
  int32_t device;         /* HIP device ordinal, -1 = current device */
  void* stream;           /* hipStream_t to launch on; NULL = the engine creates its own */
  int32_t time_kernels;   /* !=0: bracket every sweep-kernel launch with HIP events (ptm_get_kernel_times) */
  int32_t swap_log_steps; /* reserved, must be 0 (the per-candidate swap log of the LAST step is always kept: ptm_get_last_swaps) */
  int32_t exchange_row_capacity; /* row slots per boundary message (multi-GPU); 0 = automatic:
                                  * min(n_walkers, n_walkers*swap_rate + 8 sigma + 64).  More rows crossing one boundary in
                                  * one step than this is reported as PTM_ERR_FAR_MOVE by ptm_sync, never silently dropped */
  int32_t history_rungs;    /* >0: keep the history MH_chain::add_state pushes (chain.cc:935-946) for the first
                             * history_rungs rungs held by this engine; 0 = off */
  int32_t history_capacity; /* rows per chain kept on the device (a ring: saved row s sits in slot s % capacity) */
  int32_t map_rungs;        /* >0: track the MAP state MH_chain::add_state keeps (chain.cc:931-934) for the first map_rungs
                             * rungs held by this engine (the ladder's MAP is rung 0's, chain.cc:1570-1571); 0 = off */
  int32_t walker_begin;     /* (ABI 2) GLOBAL index of this engine's first walker: the engine holds walkers [walker_begin,
                             * walker_begin + n_walkers) of a larger population of independent ladders, and every random stream
                             * is keyed by the global walker -- so a population split by WALKERS over several engines / GPUs
                             * (whole ladders each: no exchange between engines at all, evolving ladders included) gives the
                             * very chains of one engine holding them all.  0 for a single engine. */
} ptm_config;

/* user plug-in likelihood, batched: the C shape of bayes_likelihood::register_evaluate_log
 * (bayesian.hh:536-552: double(*)(void* object, const state& s)).  X is [n][dim] row-major. */
typedef void (*ptm_loglike_batch_fn)(void* user, const double* X, int n, int dim, double* out_llike);

/* A proposal evaluated on the HOST: the C shape of proposal_distribution::draw(state&, chain*) + log_hastings_ratio() +
 * type() (proposal_distribution.hh:65-87), batched over the chains that make a Metropolis move this step (rungs touched by
 * an exchange attempt make none, chain.cc:1553-1557).  In: current states X_cur[n][dim] row-major, each chain's GLOBAL rung
 * and walker, the PT step number (ptm_step_count).  Out: proposed states X_prop[n][dim] -- whole states, not offsets; the
 * engine enforces the boundaries, prices prior and likelihood and takes the Metropolis test on the device with
 * logH = log_hastings + newlpost - current_lpost (chain.cc:976-1002; a NaN log_hastings rejects, :990-993) --,
 * log_hastings[n], type[n] (what MH_chain::last_type becomes on acceptance) and valid[n] (preset to 1; 0 = the proposed
 * state is invalid as state::invalid() says, e.g. state::add on a space whose origin violates a `limit` bound). */
typedef void (*ptm_propose_batch_fn)(void* user, int n, int dim, const double* X_cur, const int32_t* rung, const int32_t* walker,
                                     uint64_t step, double* X_prop, double* log_hastings, int32_t* type, int32_t* valid);
/* proposal_distribution::accept() / reject() (proposal_distribution.hh:70-71; MH_chain::step, chain.cc:1009,1015): the
 * outcome of the moves proposed by the last ptm_propose_batch_fn call, same n / order; accepted[i] in {0, 1} */
typedef void (*ptm_proposal_result_fn)(void* user, int n, const int32_t* rung, const int32_t* walker, const int32_t* accepted);

const char* ptm_last_error(void);
int ptm_abi_version(void);
/* number of usable gfx950 devices (0 if none); never fails */
int ptm_device_count(void);

int ptm_engine_create(const ptm_config* cfg, ptm_engine** out);
int ptm_engine_destroy(ptm_engine* e);

/* ---- problem description (what bayes_likelihood::basic_setup + stateSpace describe) ---------------- */
/* stateSpace::set_bound per dimension (states.hh:70-76) */
int ptm_set_bounds(ptm_engine* e, const int32_t* lower_type, const int32_t* upper_type, const double* xmin,
                   const double* xmax);
/* mixed_dist_product(space, types, centers, halfwidths) (probability_function.cc:219-262) */
int ptm_set_prior(ptm_engine* e, const int32_t* types, const double* centers, const double* halfwidths);
/* llike(x) = like0 - 1/2 (x-mean)^T P (x-mean);  P row-major D x D symmetric; mean may be NULL
 * (cython/exampleGaussian.py:53-54,61,103-109) */
int ptm_set_target_gaussian(ptm_engine* e, const double* mean, const double* precision, double like0);
/* user likelihood evaluated on the host between a propose and an accept kernel */
int ptm_set_target_callback(ptm_engine* e, ptm_loglike_batch_fn fn, void* user);
/* A PRIOR evaluated on the host: the C shape of probability_function::evaluate_log(state&) (probability_function.hh:31-44,59)
 * for priors ptm_set_prior cannot describe -- independent_dist_product, transformed_dist, chain_distribution
 * (probability_function.hh:184-246) or a user's own subclass.  Batched like the likelihood (X [n][dim] row-major, one log-prior
 * each; -inf = outside the support).  The engine calls it for the VALID proposals of a step (stateSpace::enforce stays on the
 * device, and an invalid state has probability 0 without asking: probability_function.cc:283) before the likelihood, applies the
 * reference's prior gate (chain.cc:980) to the answer, and for every state it is handed by ptm_set_states / ptm_restore.  Needs
 * the host-callback likelihood (ptm_set_target_callback): the step is then propose kernel -> prior -> likelihood -> accept kernel.
 * The device-side prior (ptm_set_prior) is ignored while a prior callback is set, and ptm_init_from_prior refuses: draw the start
 * states with the prior's own drawSample and hand them over with ptm_set_states.  fn = NULL removes it. */
typedef void (*ptm_logprior_batch_fn)(void* user, const double* X, int n, int dim, double* out_lprior);
int ptm_set_prior_callback(ptm_engine* e, ptm_logprior_batch_fn fn, void* user);
/* inverse temperatures of the GLOBAL ladder, beta[n_rungs] (chain.cc:1181-1183,1340) */
int ptm_set_ladder(ptm_engine* e, const double* beta);
/* parallel_tempering_chains::evolve_temps(rate, lpost_cut) (chain.hh:302-307; default-on in the sampler with rate 0.01,
 * lpost_cut -1: ptmcmc.cc:389-390,512).  After every accepted exchange of rungs (i, i+1) the reference widens that gap of
 * inverse temperatures by (1 + rate), renormalises all gaps and resets every rung's temperature (pry_temps,
 * chain.cc:1501-1518,1809-1846).  Each of the n_walkers ladders then owns its temperatures.  The engine keeps a ladder's
 * gaps lazily normalised inside a step (same values up to rounding, pinned against the reference by tests/golden traces
 * 5 and 6) and rebuilds the temperatures once per step.  Call after ptm_set_ladder.  History rows and MAP values taken
 * during an exchange phase carry the temperature their rung had at that add_state, between two pries of the step, as in
 * the reference (chain.cc:1487-1490,1531-1534).  lpost_cut >= 0 (chain.cc:1819-1827): every pry also widens each gap whose two
 * chains' current log-posteriors are out of order by more than lpost_cut * invtemp; the exchange kernel then decides a
 * ladder's picks one after the other and goes over all its gaps after every accepted exchange (pinned against the reference by
 * traces 11 and 12).  On a rung shard (no history / MAP there): the exchange phase must then be fed the whole ladder's llikes,
 * ptm_exchange_decide_gathered. */
int ptm_set_evolve_temps(ptm_engine* e, double rate, double lpost_cut);
/* every ladder's inverse temperatures, beta[n_walkers][n_rungs] (the common ladder repeated while nothing evolves);
 * ptm_set_invtemps puts them back (checkpoint / resume of an evolving run; needs ptm_set_evolve_temps first) */
int ptm_get_invtemps(ptm_engine* e, double* beta);
int ptm_set_invtemps(ptm_engine* e, const double* beta);
/* proposals of the LOCAL rungs: factors[rung_count][D*D] row-major (DENSE/LOWER) or [rung_count][D] (DIAG);
 * one_d_frac[rung_count] (gaussian_prop oneDfrac) may be NULL (= 0).  One clone per rung as
 * parallel_tempering_chains::set_proposal does (chain.cc:1367-1386). */
int ptm_set_proposals(ptm_engine* e, int kind, const double* factors, const double* one_d_frac);
/* Replace ONE rung's factor between steps (same kind and layout as given to ptm_set_proposals; one_d_frac < 0 keeps the
 * rung's value): the engine side of user_gaussian_prop::check_update / reset_dist (proposal_distribution.cc:406-441,
 * 340-403), whose user callback hands a chain a new covariance during the run. */
int ptm_set_proposal_rung(ptm_engine* e, int local_rung, const double* factor, double one_d_frac);
/* Scale mixture on top of the rungs' factors: a proposal_distribution_set (proposal_distribution.cc:99-129) of K Gaussian
 * members that are scalar multiples of the rung's factor -- the sampler's default Gaussian recipe (ptmcmc.cc:117-139).
 * Arrays [rung_count][K]: cumulative shares (the set's bin_max: one uniform x picks the first member with x < share),
 * scales, oneDfracs.  offset = scale_k * (factor z); last_type = k + 10 * (1 if the move was one-dimensional).
 * K = 0 removes the mixture. */
int ptm_set_proposal_mixture(ptm_engine* e, int K, const double* cum_shares, const double* scales, const double* one_d_fracs);

/* Host-side proposals -- the "host fallback step" for everything that is not a Gaussian the device can draw itself
 * (differential evolution from the chain history, involutions, adaptive sets, user proposals with callbacks ...):
 * every sweep fetches the current states, calls `propose` once for all moving chains, and runs the rest of
 * MH_chain::step on the device.  `result` (may be NULL) is told the outcomes after the accept kernel.  Replaces the
 * proposals of ptm_set_proposals; propose == NULL goes back to them.  Whole-shard sweeps only (ptm_sweep, ptm_step,
 * ptm_exchange_finish_and_sweep; not ptm_sweep_rungs). */
int ptm_set_proposal_callback(ptm_engine* e, ptm_propose_batch_fn propose, ptm_proposal_result_fn result, void* user);
/* Differential evolution (ter Braak & Vrugt 2008) drawn ON THE DEVICE from each chain's own saved history -- the reference's
 * differential_evolution (proposal_distribution.cc:476-801) as the reference sampler configures it by default (ptmcmc.cc:81-91: no
 * temperature mixing, unlikely_alpha = 0).  It becomes the member of the rungs' proposal sets (ptm_set_proposal_mixture) whose SCALE
 * IS NEGATIVE: the set's uniform picks the first ready member whose cumulative share it falls below
 * (proposal_distribution_set::draw, proposal_distribution.cc:99-129; differential evolution is ready once the chain has saved
 * 10 * dim rows, and is passed over until then -- give it a member behind it); type code = member + 10 * (0 parallel move, 1 snooker).
 * The engine must have been created with history_rungs = rung_count and a history_capacity that holds EVERY row of the run (a row the
 * ring has lost is reported by ptm_sync).  init_rows [n_init_extra][n_local_chains][dim], chain (local rung r, walker w) at
 * r * n_walkers + w: the states MH_chain::initialize(n) saved in FRONT of the start state (chain.cc:846-876: n_init_extra = n - 1,
 * oldest first), or NULL with n_init_extra = 0.  Needs dim <= 128 and the whole ladder on this engine (rung_count == n_rungs; populations
 * split by walkers are fine: PTM_ERR_UNSUPPORTED on a rung shard, whose top rung's history cannot be complete).  q == NULL switches it off.  Host-side draws with temperature
 * mixing or unlikely_alpha stay possible through ptm_set_proposal_callback. */
typedef struct ptm_de_params {
  double snooker;         /* probability of a snooker move (differential_evolution's first constructor argument; sampler: 0.1) */
  double gamma_one_frac;  /* probability of gamma = 1 in a parallel move (de_g1_frac: 0.3) */
  double reduce_gamma;    /* differential_evolution::reduce_gamma (de_reduce_gamma: 4) */
  double ignore_frac;     /* early fraction of a long history that is not drawn from (sampler: 0) */
} ptm_de_params;
int ptm_set_proposal_de(ptm_engine* e, const ptm_de_params* q, int n_init_extra, const double* init_rows);

/* ---- state ------------------------------------------------------------------------------------------ */
/* X[n_local_chains][D]; llike may be NULL (the device target evaluates it).  Resets counters the way
 * MH_chain::initialize(1) leaves them (chain.cc:649,846-876). */
int ptm_set_states(ptm_engine* e, const double* X, const double* llike);
/* MH_chain::initialize(1): draw each chain's start from the prior until valid and of finite likelihood (chain.cc:846-876);
 * every mixed_dist_product type but the flat one can be drawn (PTM_ERR_UNSUPPORTED for a flat dimension) */
int ptm_init_from_prior(ptm_engine* e);
/* MH_chain::initialize(n) draws n states per chain (chain.cc:846-876); each of them is add_state'd, the last is where the chain
 * starts.  ptm_init_from_prior_k draws the k-th of them (k = 0: what ptm_init_from_prior draws) from its own slice of the
 * chain's initialisation stream, so a caller that wants n initial samples (the history seed of differential evolution,
 * ptmcmc.cc:86) calls it for k = n-1 .. 0, reading the states back in between.  Resets the counters like ptm_set_states. */
int ptm_init_from_prior_k(ptm_engine* e, int k);
/* Draws k_begin .. k_begin + n - 1 of every chain into host arrays -- x_out [n][n_local_chains][dim], ll_out / lp_out
 * [n][n_local_chains], chain (local rung r, walker w) at r * n_walkers + w: exactly the states, log-likelihoods and log-priors that
 * ptm_init_from_prior_k(e, k) would leave in the engine, whose own state stays untouched.  MH_chain::initialize(n)'s n - 1 rows in
 * front of the start state (chain.cc:846-876) in one call (then ptm_init_from_prior_k(e, 0) for the start state itself). */
int ptm_draw_prior_rows(ptm_engine* e, int k_begin, int n, double* x_out, double* ll_out, double* lp_out);

/* ---- the hot path ------------------------------------------------------------------------------------- */
/* n x { MH_chain::step for every chain } -- no exchange phase */
int ptm_sweep(ptm_engine* e, int n);
/* n x parallel_tempering_chains::step (chain.cc:1393-1571; single-shard engines: rung_count == n_rungs).  ASYNCHRONOUS: the call
 * queues work on the engine's stream and returns; results are there when a getter, a setter or ptm_sync looks.  On the persistent
 * ladder kernel's path (populations of up to 256 workgroups, ptm_step_kernel_name) calls of fewer than 64 steps are only COUNTED and launched
 * together at that next look (or when 1024 have gathered) -- nothing can observe the difference, and a loop of ptm_step(1) calls, the
 * reference sampler's own (ptmcmc.cc:563-599), then costs what ptm_step(n) costs per step.  PTM_LADDER_DEFER=0 launches every call. */
int ptm_step(ptm_engine* e, int n);
/* wait for all queued work */
int ptm_sync(ptm_engine* e);

/* ---- multi-GPU building blocks (one engine per GPU; the caller moves bytes with RCCL) ---------------- */
/* Exchanges only ever propagate DOWN the ladder within one step (a candidate one above an earlier pick is dropped,
 * chain.cc:1417-1418), so a shard can replay every decision that concerns its rungs from: its own llikes, the llike
 * of the top rung of the shard below (W doubles) and the llikes of the bottom `halo_rungs` rungs of the shard above
 * ([halo_rungs][W]).  A chain of accepted exchanges longer than the halo sets PTM_ERR_FAR_MOVE (reported by ptm_sync). */
/* copy n_rungs rungs of the local llike array (current step), starting at local rung first_local_rung, to a device
 * buffer ([n_rungs][W] doubles) -- the halo a neighbour needs; asynchronous on the engine's stream */
int ptm_copy_llike(ptm_engine* e, int first_local_rung, int n_rungs, void* dst_dev);
/* device pointer to the local llike array of the CURRENT step, [rung_count*W] doubles, chain-major */
int ptm_llike_device_ptr(ptm_engine* e, void** dev_ptr);
/* exchange phase, part 1: decide all exchanges that concern this shard, move the rows that stay inside it and pack the
 * rows that leave it into the boundary messages send_up / send_down (device buffers of ptm_exchange_buffer_doubles()
 * doubles each; ignored at the ladder's ends).  A message is opaque to the caller: it is delivered whole to the
 * neighbour's ptm_exchange_finish_and_sweep.  (Layout: int32 row count, then ptm_exchange_row_capacity() slots of
 * {state[padded dim], llike, lprior, walker, 0}.)
 * ll_below_dev: [W] (ignored on the first shard); ll_above_dev: [halo_rungs][W] (ignored on the last shard). */
int ptm_exchange_decide(ptm_engine* e, const void* ll_below_dev, const void* ll_above_dev, int halo_rungs, void* send_up_dev,
                        void* send_down_dev);
/* The exchange phase of a shard from the WHOLE ladder's llikes -- what the reference's MPI ranks work from (gather_llikes /
 * gather_lposts: an MPI_Allgather per step, chain.cc:1433-1435,1905-1972).  ll_all_dev: [n_rungs][W] doubles, every shard's llike
 * array in rung order (the caller gathers them: ptm_copy_llike of each shard, an all-gather); lp_all_dev: the lpriors the same way
 * (ptm_copy_lprior), needed only with a posterior-ordering cut (else NULL).  No halo, no run of picks can reach past the view --
 * and it is the form EVOLVING ladders need on rung shards (ptm_set_evolve_temps): every accepted exchange renormalises all gaps
 * and every later trial of the step sees it, so every shard replays the whole ladder's trials and keeps the whole ladder's
 * temperatures.  Costs an all-gather of n_rungs x W doubles per step instead of two neighbour messages: for populations, prefer
 * splitting by walkers (ptm_config.walker_begin).  Boundary rows travel as with ptm_exchange_decide. */
int ptm_exchange_decide_gathered(ptm_engine* e, const void* ll_all_dev, const void* lp_all_dev, void* send_up_dev, void* send_down_dev);
/* RECOVERY of a run of surviving picks longer than a halo (instead of PTM_ERR_FAR_MOVE).  Whether some shard of a ladder is blind
 * in a step is a property of the step's candidate draws, which every shard replays: told the whole shard map (every shard's rung
 * count, in order, and the halo depth all of them ask for), ptm_exchange_decide leaves such a ladder alone ON EVERY SHARD -- the same
 * ladders everywhere, no message needed to agree -- and counts it.  The caller asks for the count after the decide pass
 * (ptm_exchange_redo_count: one wait on the device); if it is not zero -- on every rank alike -- it gathers the whole ladder's llikes
 * as for ptm_exchange_decide_gathered (the arrays of the ladders left alone are untouched) and ptm_exchange_redo decides those ladders
 * from the full view, appending their boundary rows to the same messages.  Then the step goes on as ever.  The reference never
 * fails here: its ranks gather everything every step (chain.cc:1905-1967).  ptmcmc_amd.parallel.ShardedLadder(recover=True) is the
 * driver; without a shard map a blind shard still says so loudly (PTM_ERR_FAR_MOVE). */
int ptm_set_shard_map(ptm_engine* e, int n_shards, const int32_t* rung_counts, int halo_rungs);
int ptm_exchange_redo_count(ptm_engine* e, int* n_ladders);
int ptm_exchange_redo(ptm_engine* e, const void* ll_all_dev, const void* lp_all_dev, void* send_up_dev, void* send_down_dev);
/* ptm_copy_llike's twin for the lpriors */
int ptm_copy_lprior(ptm_engine* e, int first_local_rung, int n_rungs, void* dst_dev);
/* exchange phase, part 2 + MH sweep: lands the rows of the neighbours' messages (device buffers of the same size; the
 * message from below is required unless this is the first shard, the one from above unless it is the last) */
int ptm_exchange_finish_and_sweep(ptm_engine* e, const void* recv_from_below_dev, const void* recv_from_above_dev);
/* The same in pieces, for callers that overlap the messages with arithmetic: the Metropolis moves of a step do not
 * depend on the rows in flight (their landing slots are exchanged rungs, which make no move), only the llike halo of
 * the NEXT step does.  ptm_sweep_rungs sweeps local rungs [first_local_rung, first_local_rung + n_rungs); every local
 * rung must be swept exactly once per step and the last call passes closes_step != 0. */
int ptm_exchange_install(ptm_engine* e, const void* recv_from_below_dev, const void* recv_from_above_dev);
int ptm_sweep_rungs(ptm_engine* e, int first_local_rung, int n_rungs, int closes_step);
/* size of one boundary message in doubles: 2 + row capacity * (padded dim + 4) */
int ptm_exchange_buffer_doubles(ptm_engine* e);
int ptm_exchange_row_capacity(ptm_engine* e);

/* ---- the same, driven natively over RCCL (for hosts that stay C/C++; ptmcmc_amd/parallel.py is the torch.distributed twin) --
 * One engine per process and GPU holds a contiguous rung block (rung_begin / rung_count); ptm_shard_step runs the whole sharded
 * PT step: llike halos and boundary rows travel as ncclSend / ncclRecv pairs between NEIGHBOUR ranks on a side stream, hidden
 * behind the sweep of the interior rungs (halos of the next step behind the second half).  RCCL is loaded at run time
 * (dlopen "librccl.so.1") by the first of these calls; nothing else in the library needs it.
 *   rank 0:   ptm_shard_unique_id(id)  -> hand the 128 bytes to every rank (file, socket, MPI, a launcher's environment ...)
 *   all:      ptm_shard_init(e, id, rank, world, rung_counts, halo)   rung_counts[world]: every rank's block length
 *             (rank r holds the block after rank r-1's; must agree with this engine's rung_begin / rung_count); halo <= 0: 12
 *             ptm_shard_step(e, n) ... ptm_sync(e) ...   ptm_shard_finalize(e) before ptm_engine_destroy
 * Replaces the reference's MPI layer for this path: rank/size `chain.cc:1199-1209`, the cyclic rung map `:1298-1309` and the
 * three MPI_Allgathers of every step `:1433-1435,1879-1972`. */
#define PTM_SHARD_ID_BYTES 128
int ptm_shard_unique_id(void* id_out);
int ptm_shard_init(ptm_engine* e, const void* id, int rank, int world, const int32_t* rung_counts, int halo_rungs);
int ptm_shard_step(ptm_engine* e, int n);
int ptm_shard_finalize(ptm_engine* e);

/* ---- read-back ------------------------------------------------------------------------------------------ */
enum {
  PTM_ARR_LLIKE = 0,  /* double */
  PTM_ARR_LPRIOR = 1, /* double */
  PTM_ARR_LPOST = 2,  /* double: lprior + beta*llike (chain.cc:928) */
  PTM_ARR_NTRIES = 3, /* int32: MH_chain::Ntries (starts at 1, chain.cc:649) */
  PTM_ARR_NACCEPT = 4,/* int32 */
  PTM_ARR_LAST_TYPE = 5, /* int32 */
  PTM_ARR_NHIST = 6,  /* int64: add_state calls */
  PTM_ARR_NSIZE = 7   /* int64: history rows (chain.cc:935-946) */
};
int ptm_get_states(ptm_engine* e, double* X);
int ptm_get_array(ptm_engine* e, int which, void* out);
/* per walker, per adjacent pair: attempts and accepts (swap_count / swap_accept_count, chain.hh:244-245);
 * each [W][n_rungs-1] int64 */
int ptm_get_swap_counts(ptm_engine* e, int64_t* tries, int64_t* accepts);
/* candidates of the most recent step: pairs[W][maxswaps] (lower rung or -2), accepted[W][maxswaps] */
int ptm_get_last_swaps(ptm_engine* e, int32_t* pairs, int32_t* accepted);
/* Several reads with one wait.  Every ptm_get_* call is an asynchronous copy on the engine's stream plus a wait; a host that
 * reads several arrays after every step (the facade's history mirror for host-side proposals: chain.cc:935-946 rows, the
 * swap log, the temperatures) brackets them: between ptm_batch_begin and ptm_batch_end the ptm_get_* calls only queue
 * their copies and return at once; the output buffers are filled when ptm_batch_end returns (possibly earlier).  Only
 * reads go between the two (ptm_step there is refused).  Brackets nest.  No counterpart in the reference (its arrays are host
 * memory); it exists because a wait on the device costs what ~10 small copies do. */
int ptm_batch_begin(ptm_engine* e);
int ptm_batch_end(ptm_engine* e);
/* Checkpoint / resume.  Everything a run's future depends on is: the states and their llikes (ptm_get_states,
 * PTM_ARR_LLIKE), the MH_chain counters (PTM_ARR_NTRIES / NACCEPT / LAST_TYPE / NHIST), the step count (the random
 * streams are counters of it) and, for the bookkeeping, the swap counters.  ptm_restore puts them back into an engine
 * configured like the one they came from (same seed, ladder, proposals, target); the run then continues bit for bit.
 * swap_tries / swap_accepts may be NULL (counters restart at 0).  An engine that keeps a history ring or a MAP starts
 * them afresh from the restored state; ptm_set_history / ptm_set_map (below) put saved ones back, and for an evolving
 * run ptm_set_evolve_temps + ptm_set_invtemps the ladders. */
int ptm_restore(ptm_engine* e, const double* X, const double* llike, const int32_t* ntries, const int32_t* naccept,
                const int32_t* last_type, const int64_t* nhist, uint64_t step_count, const int64_t* swap_tries,
                const int64_t* swap_accepts);
int ptm_max_swaps_per_step(ptm_engine* e);
/* History of the recorded rungs (ptm_config.history_rungs): for chain (local rung r, walker w) at index r*W + w and ring
 * slot k: X[(k*HC + index)*dim ..], llike / lprior [k*HC + index], meta [4*(k*HC + index)] = {Naccept, Ntries,
 * last_type, saved row number} as MH_chain::add_state saw them (acceptance_ratio = Naccept/Ntries, lpost = lprior +
 * beta*llike); HC = history_rungs * n_walkers.  Saved row s is in slot s % history_capacity; row 0 is the initial state
 * and a chain has saved Nsize rows (PTM_ARR_NSIZE).  Any output pointer may be NULL. */
int ptm_get_history(ptm_engine* e, double* X, double* llike, double* lprior, int32_t* meta);
/* the inverse temperature each saved row was saved at (MH_chain::invtemps, chain.cc:943), beta[k*HC + index]: the
 * ladder's value while the ladder is fixed, the chain's own once the ladders evolve */
int ptm_get_history_invtemps(ptm_engine* e, double* beta);
/* The ring entries of history chains [chain_begin, chain_begin + chain_count) only (chain = local rung * n_walkers + walker), written into
 * host arrays of ptm_get_history's FULL layout -- [capacity][history chains][..] -- whose other entries stay untouched; any of the five
 * pointers may be NULL.  What a chain-file writer needs (chain::dumpChain, chain.cc:1110-1140, for the rungs it dumps) without reading
 * every rung's ring: with differential evolution on the device the ring holds all rungs for the whole run. */
int ptm_get_history_chains(ptm_engine* e, int chain_begin, int chain_count, double* X, double* llike, double* lprior, int32_t* meta, double* invtemps);
/* put a saved ring / MAP back (after ptm_restore): the arrays exactly as ptm_get_history (+ ptm_get_history_invtemps;
 * invtemps may be NULL while the ladder is fixed) and ptm_get_map returned them */
int ptm_set_history(ptm_engine* e, const double* X, const double* llike, const double* lprior, const int32_t* meta,
                    const double* invtemps);
int ptm_set_map(ptm_engine* e, const double* X, const double* lpost, const double* llike, const double* lprior);
/* MAP of the tracked rungs (ptm_config.map_rungs): chain (local rung r, walker w) at index r*W + w: X[index*dim ..], its
 * log-posterior at the rung's temperature (MH_chain::getMAPlpost / getMAPstate, chain.hh:116-117), llike, lprior.
 * lpost is -1e200 while no valid state was seen.  Any output pointer may be NULL. */
int ptm_get_map(ptm_engine* e, double* X, double* lpost, double* llike, double* lprior);
uint64_t ptm_step_count(ptm_engine* e);

/* ---- measurement ---------------------------------------------------------------------------------------- */
/* HIP events on the engine's stream */
int ptm_timer_start(ptm_engine* e);
int ptm_timer_stop(ptm_engine* e, float* elapsed_ms); /* synchronises on the stop event */
/* per-launch durations of the fused sweep kernel recorded since the last call (cfg.time_kernels); ms == NULL: the records are
 * dropped unread (no event query: a measurement that discards its warm-up records must not leave the device idle meanwhile) */
int ptm_get_kernel_times(ptm_engine* e, float* ms, int capacity, int* count);
/* name of the sweep kernel variant in use (for matching rocprofv3 traces) */
const char* ptm_sweep_kernel_name(ptm_engine* e);
/* what a ptm_step call of this engine launches: one kernel for many steps -- "ladder_steps_kernel<..>" (small ladders, a block per
 * ladder) or "ladder_persistent_kernel<..>" (populations whose ladders fit 256 resident workgroups, chains in registers) -- or
 * "decide_kernel + <sweep kernel>" per step */
const char* ptm_step_kernel_name(ptm_engine* e);

/* Long ladders of few walkers step through ONE persistent kernel per ptm_step(n) call ("ladder_persistent_kernel<..>" in
 * ptm_step_kernel_name).  A launch of it commits all of its steps or none: if its workgroups cannot all be resident (a shared device)
 * one of them gives up waiting, nothing of the launch is kept, and the engine repeats the steps on the two-launch path at its
 * next look at the device (any getter, setter, ptm_sync) and keeps that path from then on.  out[0] launches of that kernel,
 * out[1] launches that gave up and were repeated, out[2] steps that took its whole-ladder exchange form (counted in diagnostics
 * runs only), out[3] 1 if the kernel is switched off for this engine.  No reference counterpart. */
int ptm_get_ladder_stats(ptm_engine* e, int64_t out[4]);
/* Sums of MH_chain::Ntries and ::Naccept (chain.hh:151-152) over this engine's chains, reduced on the device: a measurement
 * that wants "Metropolis moves made so far" reads 16 bytes instead of two arrays (and leaves the GPU no idle gap to drop
 * its clocks in).  Waits for the engine's stream. */
int ptm_get_counter_sums(ptm_engine* e, int64_t* ntries_sum, int64_t* naccept_sum);
/* What this device gives a plain streaming copy and an f64 fma issue loop right now, on the engine's stream (bench.py prints it
 * beside its roofline: devices of one pool differ by several per cent in the clock they hold under load, and a line without it
 * cannot tell a slow box from a slow kernel).  copy_GBs: bytes read + written by a 16-byte-per-lane copy of copy_bytes / 2 bytes,
 * best of 5; f64_fma_TFs: 2 flops x v_fma_f64 issued by four waves per SIMD on every CU; sclk_MHz: the shader clock held during
 * that loop (s_memtime ticks per s_memrealtime tick x 100 MHz, median over workgroups).  A measurement aid like ptm_timer_*: it
 * reads and writes scratch buffers of its own and touches nothing of the engine's state.  No reference counterpart. */
typedef struct ptm_calibration {
  double copy_GBs, copy_bytes, copy_ms;
  double f64_fma_TFs, fma_ms, sclk_MHz;
  int32_t compute_units, reserved;
} ptm_calibration;
int ptm_calibrate(ptm_engine* e, ptm_calibration* out);

/* ---- verification hooks (used by tests/ only; evaluate device functions on arrays) ---------------------- */
enum { PTM_FN_LOG = 0, PTM_FN_EXP = 1, PTM_FN_SIN_0_PI = 2, PTM_FN_COS_HPI = 3, PTM_FN_SQRT = 4, PTM_FN_DIV = 5,
       PTM_FN_SQRT_RAW = 6 };
int ptm_debug_eval(int device, int fn, const double* a, const double* b, double* out, int n);
int ptm_debug_philox(int device, uint64_t seed, int tag, uint32_t stream, uint64_t step, uint32_t block, uint32_t out[4]);
int ptm_debug_boxmuller(int device, const uint32_t* k1, const uint32_t* k2, double* z0, double* z1, int n);
/* exhaustive scan of all 2^32 Box-Muller radius arguments a = -2 ln((k+.5)/2^32): for how many is the hot path's
 * unscaled Newton square root NOT the correctly rounded one (or a outside its domain).  Must be 0. */
int ptm_debug_sqrt_scan(int device, uint64_t* mismatches);
/* evaluate lprior / llike of arbitrary states with the engine's problem description: X[n][D] */
/* device-memory helpers for callers without a GPU array library (tests, tools) */
int ptm_dev_alloc(size_t bytes, void** out);
int ptm_dev_free(void* p);
int ptm_dev_copy(void* dst, const void* src, size_t bytes); /* any direction, synchronous */
int ptm_debug_evaluate(ptm_engine* e, const double* X, int n, int32_t* valid, double* X_enforced, double* lprior,
                       double* llike);

#ifdef __cplusplus
}
#endif
#endif /* PTM_ENGINE_H */
