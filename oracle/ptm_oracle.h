/* oracle/ptm_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the one hot path this repository accelerates:
 * ptmcmc's chain::step()  (MH_chain::step + parallel_tempering_chains::step).
 * It exists to CHECK the HIP engine; nothing in the product (ptmcmc_amd/, include/,
 * bench.py's GPU leg) may link, import or call it.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_*.py) against
 *   - golden vectors produced by the real reference compiled from /root/reference
 *     (oracle/Makefile target `ref`, generator tests/golden/make_golden.py):
 *     boundary::enforce tables, mixed_dist_product::evaluate_log tables, the ladder,
 *     and six full parallel_tempering_chains traces replayed from recorded RNG tapes (two of them with
 *     evolve_temps on), every history row's log-posterior included;
 *   - the reference's own golden file test/exampleLISA/exampleLISA_test_0_t0.dat
 *     (31 prior-draw rows: lpost, llike, parameters);
 *   - the Random123 known-answer vectors for Philox4x32-10.
 *
 * Every function cites the reference file:line (relative to /root/reference) it follows.
 */
#ifndef PTM_ORACLE_H
#define PTM_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* boundary types: states.hh:34-37 */
enum { PTMO_OPEN = 0, PTMO_LIMIT = 1, PTMO_REFLECT = 2, PTMO_WRAP = 3 };
/* prior types: the values of mixed_dist_product::{uniform,gaussian,polar,copolar,log} (probability_function.hh:151-155);
   0 = flat (the base probability_function: evaluate_log == 0, probability_function.hh:40) */
enum { PTMO_FLAT = 0, PTMO_UNIFORM = 1, PTMO_GAUSSIAN = 2, PTMO_POLAR = 3, PTMO_COPOLAR = 4, PTMO_LOG = 5 };
/* proposal kinds */
enum { PTMO_PROP_DENSE = 0, PTMO_PROP_DIAG = 1 };
/* RNG domains (counter word 3, top byte) */
enum { PTMO_TAG_MH = 0, PTMO_TAG_PT = 1, PTMO_TAG_INIT = 2 };

typedef double (*ptmo_loglike_fn)(void* user, const double* x, int dim);

#define PTMO_MAX_DIM 1024   /* the engine pads a state to 4 .. 1024 dimensions */

typedef struct {
  int D;
  /* per-dimension boundary descriptors (states.hh:29-48) */
  int *blo, *bhi;
  double *bmin, *bmax;
  int origin_valid;   /* quirk Q9: state::add() builds its result from an enforced zero vector
                         (states.cc:183-192,205-214); if 0 is outside a `limit` bound every
                         proposed state is invalid */
  /* per-dimension prior descriptors (probability_function.cc:219-262) */
  int *ptype;
  double *plo, *phi;  /* support [lo,hi] (uniform/polar/copolar/log) or {x0,sigma} (gaussian) */
  double *pcoef;      /* uniform: 1/(hi-lo); gaussian: unused; polar/copolar: norm; log: log(hi)-log(lo) */
  int all_uniform;
  double lprior_const; /* log(prod 1/(hi-lo)) when all_uniform */
  /* target */
  int have_gauss;
  double *mean;       /* may be NULL */
  double *P2;         /* D*D row-major: strictly-lower part holds 2*P_ij, diagonal holds P_ii */
  double like0;
  ptmo_loglike_fn user_fn;
  void* user;
  double minPrior;    /* MH_chain ctor arg (chain.cc:647), sampler default -30 */
  ptmo_loglike_fn prior_fn;   /* a prior the checker is handed as a function (probability_function::evaluate_log of any subclass, */
  void* prior_user;           /* probability_function.hh:31-44): replaces the per-dimension description above for VALID states     */
} ptmo_problem;

typedef struct {
  int kind;        /* PTMO_PROP_DENSE / PTMO_PROP_DIAG */
  double* M;       /* DENSE: D*D row-major factor (offset = M z);  DIAG: D sigmas */
  double oneDfrac; /* gaussian_prop oneDfrac (proposal_distribution.hh:194-206) */
  /* optional scale mixture -- a proposal_distribution_set (proposal_distribution.cc:99-129) of K Gaussian members that
   * are scalar multiples of M (the sampler's default Gaussian recipe, ptmcmc.cc:117-139): mix[k] = {cumulative share,
   * scale, oneDfrac}; one uniform picks the first k with x < cumulative share; offset = scale_k * (M z);
   * type = k + 10 * (member's type).  K = 0: no mixture (M and oneDfrac alone). */
  int K;
  const double* mix;
} ptmo_proposal;

/* RNG provider: Philox in production parity tests, tapes for the reference-trace fixtures */
typedef struct ptmo_rng {
  void* ctx;
  /* uniform for chain (walker w, global rung r) at PT step `step`; slot 0 = MH accept */
  double (*chain_uniform)(void* ctx, int w, int r, uint64_t step, int slot);
  /* proposal offset for the chain; returns proposal type (0 full, 1 one-dimensional) */
  int (*draw_offset)(void* ctx, int w, int r, uint64_t step, const ptmo_proposal* p, int D, double* offset);
  /* uniform of the ladder-level generator of walker w; k = candidate slot, slot 0 try,1 pick,2 accept */
  double (*pt_uniform)(void* ctx, int w, uint64_t step, int k, int slot);
  /* proposal_distribution::log_hastings_ratio() of the offset just drawn (proposal_distribution.hh:68; MH_chain::step,
   * chain.cc:989-994); NULL = 0 (every gaussian_prop) */
  double (*log_hastings)(void* ctx, int w, int r, uint64_t step);
  /* uniform `slot` of the chain's differential-evolution draw of this step (ptmo_de_uniform_fn); NULL: the provider has none */
  double (*de_uniform)(void* ctx, int w, int r, uint64_t step, int slot);
} ptmo_rng;

/* differential_evolution (proposal_distribution.cc:476-801; ter Braak & Vrugt 2008): the parameters the draw depends on without
 * temperature mixing and with unlikely_alpha = 0 (the sampler's defaults, ptmcmc.cc:81-91) */
typedef struct {
  double snooker;         /* probability of a snooker move (differential_evolution ctor arg 1) */
  double gamma_one_frac;  /* probability of gamma = 1 in a parallel move (arg 2) */
  double reduce_gamma;    /* reduce_gamma(factor): both gammas are divided by it */
  double ignore_frac;     /* the early fraction of a long history that is not drawn from (arg 4) */
} ptmo_de_params;
/* uniform number `slot` of ONE draw: 0 the snooker test, 1 gamma, 2 the pick of z1, 3 the pick of z2, 4 + t the t-th attempt at the
 * snooker move's z.  The function asks for them in the order the reference's generator delivers them (0, 1, [4, 5, ..], 2, 3), so
 * a tape of the reference's uniforms can answer in sequence. */
typedef double (*ptmo_de_uniform_fn)(void* ctx, int slot);
typedef const double* (*ptmo_de_row_fn)(void* ctx, long row);   /* row `row` of the caller's saved history (raw indexing, chain.hh:155) */
/* one differential_evolution::draw for the state x[D] of a chain whose history holds `rows` rows: the proposed state, its
 * log-Hastings ratio; returns the move's type (0 parallel, 1 snooker; proposal_distribution::type()), -1 if a thousand history
 * states in a row equalled x (the reference exits there) */
int ptmo_de_draw(int D, const double* x, long rows, const ptmo_de_params* q, ptmo_de_uniform_fn u, void* uctx, ptmo_de_row_fn rowf,
                 void* rctx, double* xn, double* log_hastings);
int ptmo_de_ready(int D, long rows);   /* differential_evolution::is_ready(): the history holds 10 D rows */

/* A proposal evaluated by the CALLER (the engine's ptm_propose_batch_fn, include/ptm_engine.h): any
 * proposal_distribution::draw(state&, chain*) (proposal_distribution.hh:65-87).  For n chains: current states X_cur[n][dim],
 * their global rung and walker, the PT step -> proposed states X_prop[n][dim] (whole states, not offsets), the proposals'
 * log-Hastings ratios, type codes (proposal_distribution::type()) and validity (state::invalid(); preset to 1). */
typedef void (*ptmo_propose_fn)(void* user, int n, int dim, const double* X_cur, const int32_t* rung, const int32_t* walker, uint64_t step,
                                double* X_prop, double* log_hastings, int32_t* type, int32_t* valid);

typedef struct {
  int D, Nt, W;        /* Nt = GLOBAL number of rungs; chain (w,r) lives at index w*Nt + r */
  double* beta;        /* [Nt] */
  double swap_rate;
  int maxswaps;        /* 1 + 2*swap_rate*Nt (chain.cc:1192) */
  int add_every_N;
  uint64_t step;       /* PT steps taken */
  double* x;           /* [W*Nt][D] */
  double *llike, *lprior; /* lpost is always fl(lprior + fl(beta*llike)) */
  int32_t *ntries, *naccept, *last_type; /* init 1,1,-1 (chain.cc:649) */
  int64_t *nhist, *nsize;               /* add_state counters (chain.cc:916-949) */
  /* swap diagnostics per walker: [W][Nt-1] */
  int64_t *swap_count, *swap_accept_count;
  /* per-step log of the last step's swap attempts (walker-major): pair index or -2, and accept flag */
  int* last_pairs;     /* [W][maxswaps] */
  int* last_accept;    /* [W][maxswaps] */
  uint8_t* touched;    /* [W*Nt] scratch: add_state calls received in the swap phase of the last step */
  /* optional history (MH_chain::states / llikes / acceptance_ratio / types, chain.cc:935-946): row s of chain c is
   * what add_state pushed when Nsize was s; row 0 is the initial state.  hist_cap rows per chain (no ring here). */
  int hist_cap;
  double* hist_x;      /* [W*Nt][hist_cap][D] */
  double *hist_ll, *hist_lp;             /* [W*Nt][hist_cap] */
  int32_t *hist_nacc, *hist_ntry, *hist_type; /* [W*Nt][hist_cap] */
  /* MAP tracking of MH_chain::add_state (chain.cc:931-934; MAPlpost starts at -1e200, chain.hh:69) */
  double* map_lpost;   /* [W*Nt] */
  double* map_x;       /* [W*Nt][D] */
  /* temperature evolution (parallel_tempering_chains::evolve_temps, chain.hh:302-307; pry_temps chain.cc:1809-1846):
   * every ladder re-tunes its own temperatures, so the inverse temperatures become per walker.  evolve_rate == 0: off
   * (betaw NULL, beta[] rules). */
  double evolve_rate;
  double* betaw;       /* [W][Nt] */
  double* hist_beta;   /* [W*Nt][hist_cap] the chain's inverse temperature when the row was pushed (MH_chain::invtemps, chain.cc:943) */
  /* optional caller-evaluated proposal (replaces the rng's draw_offset / props[] in ptmo_mh_step) */
  ptmo_propose_fn host_prop;
  void* host_prop_user;
  uint8_t* last_accept_mh; /* [W*Nt] outcome of the last step's MH move: 1 accepted, 0 rejected, 2 no move (exchanged rung) */
  double evolve_cut;       /* evolve_temp_lpost_cut (chain.hh:254,302-307): < 0 off (the default) */
  /* differential evolution as a member of the proposal set (ptmo_pt_set_de): the mixture member whose scale is negative */
  int de_on;
  ptmo_de_params de;
  int de_init_extra;       /* rows of MH_chain::initialize(n) in front of the start state: n - 1 (chain.cc:846-876) */
  double* de_init;         /* [de_init_extra][W*Nt][D] */
} ptmo_pt;

/* ---- RNG: Philox4x32-10 (Salmon et al., SC'11; Random123) -------------------------------- */
void ptmo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* open-interval map (k+0.5)/2^32 of newran1.cxx:432 */
double ptmo_u01(uint32_t k);
int ptmo_column_order(int D, int* ord); /* accumulation order of a proposal factor's columns; returns D */
double ptmo_bm_neg2log(uint32_t k); /* -2 ln((k+.5)/2^32), the Box-Muller radius argument */
/* engine counter layout: c0 = block, c1 = stream, c2 = step low, c3 = step high(24) | tag<<24 */
void ptmo_draw_block(uint64_t seed, int tag, uint32_t stream, uint64_t step, uint32_t block, uint32_t out[4]);
void ptmo_boxmuller(uint32_t k1, uint32_t k2, double* z0, double* z1);

/* ---- deterministic elementary functions (same IEEE op sequence as the HIP kernels) ------- */
double ptmo_log(double x);
double ptmo_exp(double x);
double ptmo_sin_0_pi(double x);      /* x in [0, pi]        */
double ptmo_cos_hpi(double x);       /* x in [-pi/2, pi/2]  */

/* ---- reference restatements -------------------------------------------------------------- */
int ptmo_boundary_enforce(int lo, int hi, double xmin, double xmax, double* x); /* states.cc:11-58 */
int ptmo_enforce(const ptmo_problem* pb, double* x);                              /* states.cc:86-102 */
void ptmo_problem_set_user_prior(ptmo_problem* p, ptmo_loglike_fn fn, void* user);
double ptmo_lprior(const ptmo_problem* pb, const double* x, int valid);          /* probability_function.hh:59, .cc:281-304 */
double ptmo_llike(const ptmo_problem* pb, const double* x);
double ptmo_lpost(double lprior, double beta, double llike);
void ptmo_ladder(int Nt, double Tmax, double* beta);                              /* chain.cc:1181-1183,1340 */

ptmo_problem* ptmo_problem_create(int D);
void ptmo_problem_free(ptmo_problem*);
void ptmo_problem_set_bounds(ptmo_problem*, const int* lo, const int* hi, const double* xmin, const double* xmax);
/* (types, centers, halfwidths) exactly as mixed_dist_product's constructor takes them */
void ptmo_problem_set_prior(ptmo_problem*, const int* types, const double* centers, const double* halfwidths);
void ptmo_problem_set_gauss(ptmo_problem*, const double* mean, const double* P, double like0);
void ptmo_problem_set_user(ptmo_problem*, ptmo_loglike_fn fn, void* user);

ptmo_pt* ptmo_pt_create(int D, int Nt, int W, const double* beta, double swap_rate, int add_every_N);
void ptmo_pt_free(ptmo_pt*);
/* evolve_temps(rate) with lpost_cut < 0 (the sampler's defaults, ptmcmc.cc:389-390,512).  History rows and MAP values taken
 * DURING a swap phase see the rung's temperature between two pries of the step (swap_phase in ptm_oracle.c). */
void ptmo_pt_evolve_temps(ptmo_pt*, double rate);
/* evolve_temps(rate, lpost_cut) with lpost_cut >= 0 (chain.cc:1819-1827): call before or after ptmo_pt_evolve_temps */
void ptmo_pt_evolve_lpost_cut(ptmo_pt*, double cut);
/* exclusive prefix sums of v[0..n) in the order the engine uses: chunks of 32 summed left to right from 0, then the chunk
 * totals summed left to right; P (may be NULL) gets P[k] = offset[k/32] + local sum before k.  Returns the total. */
double ptmo_chunk_prefix(const double* v, int n, double* P);
void ptmo_pt_enable_history(ptmo_pt*, int rows_per_chain); /* call before ptmo_pt_set_states / ptmo_init_from_prior */
/* set states and (re)evaluate lprior/llike; llike may be NULL => evaluate the target */
void ptmo_pt_set_states(ptmo_pt*, const ptmo_problem*, const double* x, const double* llike);

/* one MH_chain::step for chain index c (chain.cc:966-1022); returns 1 if accepted */
int ptmo_mh_step(ptmo_pt*, const ptmo_problem*, const ptmo_proposal* prop, const ptmo_rng*, int w, int r);
/* one parallel_tempering_chains::step for every walker (chain.cc:1393-1571); props[Nt] per rung */
void ptmo_pt_step(ptmo_pt*, const ptmo_problem*, const ptmo_proposal* props, const ptmo_rng*, int nthreads);
void ptmo_exchange_phase(ptmo_pt*, const ptmo_rng*);   /* the exchange phase of ptmo_pt_step alone (the step count stays) */
/* MH sweep only (no swap phase) */
void ptmo_sweep(ptmo_pt*, const ptmo_problem*, const ptmo_proposal* props, const ptmo_rng*, int nthreads);

/* run-length census of the surviving exchange picks above shard boundaries (sizes the multi-GPU llike halo; see the .c) */
void ptmo_selection_run_census(uint64_t seed, int Nt, double swap_rate, int W, uint64_t step0, int nsteps, const int* bounds,
                               int nb, int Lmax, int64_t* hist, int nthreads);

/* Philox provider */
ptmo_rng* ptmo_rng_philox(uint64_t seed, int Nt);
/* tape provider: chain tapes [W*Nt][len_c], pt tapes [W][len_p], deltas [W*Nt][nsteps][D] */
ptmo_rng* ptmo_rng_tape(int W, int Nt, int D, const double* chain_tapes, int len_c, const double* pt_tapes, int len_p,
                        const double* deltas, int nsteps);
/* scripted log-Hastings ratios and type codes for the tape provider's offsets: [W*Nt][nsteps] each (types may be NULL) */
void ptmo_rng_tape_hastings(ptmo_rng*, const double* log_hastings, const int32_t* types);
void ptmo_pt_set_host_proposal(ptmo_pt*, ptmo_propose_fn fn, void* user);
/* Differential evolution from each chain's OWN saved history as the member of the rungs' proposal sets whose scale is negative
 * (ptmo_proposal::mix): needs the history (ptmo_pt_enable_history) with room for every row of the run.  init_rows: the
 * n_init_extra states MH_chain::initialize(n) saved in front of the start state, [n_init_extra][W*Nt][D], or NULL.  A member that
 * is not ready (fewer than 10 D rows) is passed over as proposal_distribution_set::draw passes over it (proposal_distribution.cc:111).
 * The uniforms of a draw: Philox block 0x0DE00000 of the chain's MH stream (slots 0..3), blocks 0x0DE00001 + t (slot 4 + t). */
void ptmo_pt_set_de(ptmo_pt*, const ptmo_de_params* q, int n_init_extra, const double* init_rows);
void ptmo_rng_free(ptmo_rng*);

/* prior draw used by init (uniform / gaussian dims only; others return NaN) */
void ptmo_init_from_prior(ptmo_pt*, const ptmo_problem*, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
