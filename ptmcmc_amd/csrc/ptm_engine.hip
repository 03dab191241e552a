// ptm_engine.hip -- host side of the C ABI declared in include/ptm_engine.h.
//
// Thin by design: it owns device memory, packs the problem description the way the kernels want it,
// and enqueues kernels on one HIP stream.  No algorithmic work happens on the host; there is no CPU
// fallback -- without a gfx950 device every entry point that computes returns PTM_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "../../include/ptm_engine.h"
#include "ptm_aux_kernels.hpp"
#include "ptm_launch.hpp"
#include "ptm_shard_rccl.hpp"

using namespace ptm;

static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIPCHK(x)                                                                                         \
  do {                                                                                                    \
    hipError_t _e = (x);                                                                                  \
    if (_e != hipSuccess) return fail(PTM_ERR_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

// host staging buffers of the callback path live in pinned memory: a hipMemcpyAsync from / to pageable memory is staged
// by the runtime (22 us per copy measured against ~5)
template <class T>
struct PinnedAlloc {
  typedef T value_type;
  PinnedAlloc() {}
  template <class U> PinnedAlloc(const PinnedAlloc<U>&) {}
  T* allocate(size_t n) {
    void* p = nullptr;
    if (hipHostMalloc(&p, n * sizeof(T), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) throw std::bad_alloc();
    return (T*)p;
  }
  void deallocate(T* p, size_t) { (void)hipHostFree(p); }
  template <class U> bool operator==(const PinnedAlloc<U>&) const { return true; }
  template <class U> bool operator!=(const PinnedAlloc<U>&) const { return false; }
};
template <class T> using pinned_vector = std::vector<T, PinnedAlloc<T>>;

struct ptm_engine {
  ptm_config cfg;
  int D = 0, DP = 0, Nt = 0, r0 = 0, nloc = 0, W = 0, Nc = 0, ms = 0;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  uint64_t step = 0;
  // device state
  double *x = nullptr, *ll = nullptr, *lp = nullptr;   // rows [Nc][DP] and per-chain scalars, updated in place
  int *ntries = nullptr, *naccept = nullptr, *last_type = nullptr, *err = nullptr;
  int *arr_below = nullptr, *arr_above = nullptr;
  int *mv_src = nullptr, *mv_dst = nullptr, *mv_n = nullptr;   // per-ladder move lists (exchange kernel -> move kernel)
  unsigned int* nhist = nullptr;
  long long* swap_cnt = nullptr;   // [W][Nt-1][2] {tries, accepts}
  unsigned char* touch = nullptr;
  Hist hist = {0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr};   // optional history ring (ptm_config.history_rungs)
  MapT map = {0, 0, nullptr, nullptr, nullptr, nullptr};       // optional MAP tracking (ptm_config.map_rungs)
  int* swap_log = nullptr;   // [PTM_LOG_RING][W][ms] candidate logs of the last steps; the swap counters lag behind by
  int log_head = 0, log_pending = 0;   //  the `log_pending` newest of them (fold_swap_log)
  int row_cap = 0;           // row slots per boundary message
  // device problem description
  int *blo = nullptr, *bhi = nullptr, *ptype = nullptr;
  double *bmin = nullptr, *bmax = nullptr, *plo = nullptr, *phi = nullptr, *pcoef = nullptr;
  double *P2 = nullptr, *mean = nullptr, *beta = nullptr, *prop = nullptr, *prop_tiles = nullptr, *P2_tiles = nullptr, *box_row = nullptr, *onedfrac = nullptr, *mix = nullptr;
  int mix_K = 0;
  // differential evolution on the device (ptm_set_proposal_de)
  bool de_on = false;
  ptm_de_params de = {0, 0, 0, 0};
  int de_init_extra = 0;
  double *de_init = nullptr, *de_hast = nullptr;
  int* de_type = nullptr;
  // evolving ladders (ptm_set_evolve_temps): per-ladder inverse temperatures [W][Nt] and their chain-indexed image [Nc]
  double evolve_rate = 0, evolve_cut = -1;
  double *beta_w = nullptr, *betaC = nullptr, *beta_add = nullptr;
  bool betaC_stale = false;   // the chain-indexed image of evolving temperatures lags the ladder-major one (ensure_betaC)
  // host copies / flags
  int has_bounds = 0, origin_valid = 1, all_uniform = 1, has_mean = 0, have_target = 0, have_ladder = 0,
      have_prop = 0, have_state = 0, prop_kind = KIND_DIAG, prop_stride = 0, any_oned = 0, bounds_box = 1;
  double lprior_const = 0, like0 = 0, thresh = 0;
  bool lp_is_const = false;   // every chain's lprior equals lprior_const (checked when states are set; sweeps keep it so)
  std::vector<double> h_beta;
  std::vector<int> h_ptype;
  std::vector<double> h_plo, h_phi;
  ptm_loglike_batch_fn cb = nullptr;
  void* cb_user = nullptr;
  ptm_logprior_batch_fn prior_cb = nullptr;   // host-evaluated prior (ptm_set_prior_callback)
  void* prior_user = nullptr;
  // host-side proposals (ptm_set_proposal_callback)
  ptm_propose_batch_fn pcb = nullptr;
  ptm_proposal_result_fn pres = nullptr;
  void* pcb_user = nullptr;
  // compacted sweep (partition_kernel): per-rung lists of the walkers that move, their counts; touched = an exchange phase
  // ran since the last sweep
  int *cidx = nullptr, *ccnt = nullptr;
  bool touched = false;
  bool compact_step = false;   // this step's (partial) sweeps are compacted
  ShardComm* shard = nullptr;   // native RCCL sharding (ptm_shard_*)
  // recovery of runs longer than a halo (ptm_set_shard_map): the shards' ends, the nominal halo, per-ladder flags | count
  int* shard_ends = nullptr;
  int nshards = 0, halo_nominal = 0;
  int* redo_flag = nullptr;     // [W] flags, then [1] the count
  long long redo_total = 0;     // ladders decided by the gathered second pass so far
  // persistent ladder kernel (ptm_ladder_kernel.hpp): published rows / llikes / lpriors by step parity, the workgroups' flags
  double *pub_x = nullptr, *pub_ll = nullptr, *pub_lp = nullptr;   // (one allocation: pub_x)
  int *lad_flags = nullptr, *lad_ctl = nullptr;                     // (one allocation: lad_flags)
  long long* lad_prof = nullptr;
  int lad_capacity[2][32] = {{-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1},
                             {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1}};   // workgroups of each build of that kernel [diagonal][FL] the device holds at once (-1: not asked yet)
  long long ladder_launches = 0, ladder_whole_steps = 0;   // launches of that kernel; steps (of walker 0) whose exchange phase needed the whole ladder
  // A launch of that kernel commits all of its steps or none (ptm_ladder_kernel.hpp) and is ASYNCHRONOUS: the host learns at its
  // next look (ladder_settle) whether the launches since the last look were committed -- the device keeps the number of the last one
  // that was (err[2]) -- and, if one gave up, repeats its steps and its successors' on the two-launch path.
  struct LadLaunch { int seq; uint64_t step_before; int nsteps; int log_head_before; };
  std::vector<LadLaunch> lad_log;   // launches not yet looked at
  // Steps asked for in small portions (a host loop that calls ptm_step(1): the reference sampler's, ptmcmc.cc:563-599) are only COUNTED
  // here and launched together at the engine's next look at the device -- any getter, setter or ptm_sync -- or when enough have
  // gathered: nothing can observe the difference, and a launch of the persistent kernel costs ~15 us on top of its 5 us steps.
  int lad_deferred = 0;
  int lad_seq = 0;                  // number of the last launch issued
  bool lad_disabled = false;        // a launch gave up once: this engine keeps the two-launch path from then on
  long long ladder_fallbacks = 0;   // launches that gave up (their steps were repeated on the two-launch path)
  hipEvent_t lad_event = nullptr;   // end of this engine's last launch of that kernel (two engines' grids must not share the device)
  unsigned int nhist_pending = 0;   // steps whose one-add-per-chain the compacted sweep left uncounted (flush_nhist)
  double* hastings = nullptr;
  int* htype = nullptr;
  unsigned char *hvalid = nullptr, *acc_out = nullptr;
  pinned_vector<double> h_rows, h_hast;
  pinned_vector<int> h_type;
  pinned_vector<unsigned char> h_valid, h_touch, h_acc;
  std::vector<double> p_xcur, p_xprop, p_hast;
  std::vector<int32_t> p_rung, p_walker, p_type, p_valid, p_acc;
  std::vector<size_t> p_pick;
  double *xprop = nullptr, *lprior_new = nullptr, *llike_new = nullptr;  // device buffers of the callback path
  // reads of device arrays by the ptm_get_* calls: asynchronous copies into one pinned arena, finished by ONE wait (a
  // synchronous hipMemcpy into pageable memory costs ~20 us whatever its size); between ptm_batch_begin and ptm_batch_end the
  // calls only queue, and their output buffers are filled by ptm_batch_end
  pinned_vector<unsigned char> fetch_arena;
  size_t fetch_used = 0;
  int fetch_depth = 0;
  std::vector<std::function<void()>> fetch_after;
  std::vector<std::shared_ptr<std::vector<unsigned char>>> fetch_big;
  std::vector<std::pair<unsigned char*, size_t>> host_blocks;   // arrays kept in mapped host memory (a small history ring): read in place
  bool shared_handover = false, hist_on_host = false;   // ... or, for a small population, the pinned host images themselves (mapped into the device)
  unsigned char* gate = nullptr;
  pinned_vector<double> h_xprop, h_llnew;
  std::vector<double> h_batch, h_llbatch;
  unsigned char* h_gate = nullptr;   // view into h_xprop's tail
  unsigned long long *sums = nullptr, *h_sums = nullptr;   // ptm_get_counter_sums
  // timing
  hipEvent_t t0 = nullptr, t1 = nullptr;
  std::vector<hipEvent_t> kev;  // pairs
  size_t kev_used = 0;
  std::string kname;
};

// Persistent ladder kernels of different engines (streams) must not run at the same time: each sizes its grid for a device of its
// own, two at once may not be resident together and would wait for each other's workgroups until they give up.  A launch
// therefore waits for the event of the last such launch of any OTHER engine of the process.
#include <mutex>
static std::mutex g_lad_mutex;
static ptm_engine* g_lad_last = nullptr;

static int round_dp(int D) {
  if (D <= 4) return 4;
  if (D <= 8) return 8;
  if (D <= 16) return 16;
  if (D <= 32) return 32;
  if (D <= 64) return 64;
  if (D <= 128) return 128;
  if (D <= 256) return 256;
  if (D <= 512) return 512;
  return 1024;
}

template <class T>
static int dalloc(T** p, size_t n) {
  HIPCHK(hipMalloc((void**)p, (n ? n : 1) * sizeof(T)));
  return PTM_OK;
}
// a copy between a device buffer and its host image; nothing to do when the two are one (small populations keep the
// hand-over buffers of the host paths in mapped host memory, alloc_proposal_buffers)
static hipError_t copy_unless_shared(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s) {
  if (dst == src) return hipSuccess;
  return hipMemcpyAsync(dst, src, bytes, kind, s);
}
struct ptm_engine;
static int fetch_flush(ptm_engine* e);
static int fetch(ptm_engine* e, const void* dev, size_t bytes, const unsigned char** staged);
static int fetch_done(ptm_engine* e);
// small write-only outputs of the kernels that the host reads after every step (the history ring of a small population):
// mapped, coherent host memory -- the kernels' writes go over the bus as they happen, the host reads them with no copy
static bool small_outputs_on_host(size_t bytes) {
  const char* z = getenv("PTM_SHARED_HANDOVER");
  return bytes <= ((size_t)1 << 20) && !(z && *z == '0');
}
template <class T>
static int halloc(ptm_engine* e, T** p, size_t n);
template <class T>
static int upload(T* d, const T* h, size_t n, hipStream_t s) {
  HIPCHK(hipMemcpyAsync(d, h, n * sizeof(T), hipMemcpyDefault, s));
  HIPCHK(hipStreamSynchronize(s));
  return PTM_OK;
}

extern "C" const char* ptm_last_error(void) { return g_err.c_str(); }
extern "C" int ptm_abi_version(void) { return PTM_ABI_VERSION; }

extern "C" int ptm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  int ok = 0;
  for (int i = 0; i < n; ++i) {
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, i) == hipSuccess && strncmp(pr.gcnArchName, "gfx950", 6) == 0) ok++;
  }
  return ok;
}

static int need_device() {
  if (ptm_device_count() <= 0)
    return fail(PTM_ERR_NO_DEVICE, "no gfx950 (MI355X) device visible: the engine has no CPU fallback");
  return PTM_OK;
}

extern "C" int ptm_engine_destroy(ptm_engine* e);
extern "C" int ptm_shard_finalize(ptm_engine* e);
static int build_engine(ptm_engine* e, const ptm_config* cfg);
static int launch_beta_transpose(ptm_engine* e);
static int ensure_betaC(ptm_engine* e);
static bool lean_ev_sweeps(const ptm_engine* e);
static int ladder_settle(ptm_engine* e);
static int fill_evolving_ladders(ptm_engine* e);

extern "C" int ptm_engine_create(const ptm_config* cfg, ptm_engine** out) {
  if (!cfg || !out) return fail(PTM_ERR_INVALID, "null argument");
  // ABI evolution: a caller built against an older header hands a shorter struct; the fields it does not know are zero
  ptm_config cfg_full;
  memset(&cfg_full, 0, sizeof cfg_full);
  if (cfg->struct_size > sizeof(ptm_config) || cfg->struct_size < offsetof(ptm_config, walker_begin)) return fail(PTM_ERR_INVALID, "ptm_config size mismatch (ABI)");
  memcpy(&cfg_full, cfg, cfg->struct_size);
  cfg_full.struct_size = sizeof(ptm_config);
  cfg = &cfg_full;
  if (cfg->walker_begin < 0) return fail(PTM_ERR_INVALID, "walker_begin must be >= 0");
  if (cfg->dim < 1) return fail(PTM_ERR_INVALID, "dim must be >= 1");
  if (cfg->dim > 1024) return fail(PTM_ERR_UNSUPPORTED, "dim > 1024 is not built (kernels exist for padded dimensions 4, 8, 16, ..., 1024)");
  if (cfg->n_rungs < 1 || cfg->n_rungs > 65535) return fail(PTM_ERR_INVALID, "n_rungs must be in 1..65535");
  if (cfg->rung_begin < 0 || cfg->rung_count < 1 || cfg->rung_begin + cfg->rung_count > cfg->n_rungs)
    return fail(PTM_ERR_INVALID, "rung block out of range");
  if (cfg->n_walkers < 1) return fail(PTM_ERR_INVALID, "n_walkers must be >= 1");
  if (cfg->add_every_n < 1) return fail(PTM_ERR_INVALID, "add_every_n must be >= 1");
  if ((double)cfg->rung_count * cfg->n_walkers > 2.0e9) return fail(PTM_ERR_INVALID, "too many chains for one engine");
  // a chain's random stream id is (global walker) * n_rungs + (global rung), 32 bits
  if (((double)cfg->walker_begin + cfg->n_walkers) * cfg->n_rungs > 4294967296.0) return fail(PTM_ERR_INVALID, "too many chains for 32-bit stream ids: (walker_begin + n_walkers) * n_rungs must stay below 2^32");
  int rc = need_device();
  if (rc) return rc;
  ptm_engine* e = new ptm_engine();
  rc = build_engine(e, cfg);
  if (rc) { (void)ptm_engine_destroy(e); return rc; }   // nothing of a half-built engine survives
  *out = e;
  return PTM_OK;
}

static int build_engine(ptm_engine* e, const ptm_config* cfg) {
  int rc;
  e->cfg = *cfg;
  e->D = cfg->dim; e->DP = round_dp(cfg->dim); e->Nt = cfg->n_rungs; e->r0 = cfg->rung_begin; e->nloc = cfg->rung_count;
  e->W = cfg->n_walkers; e->Nc = e->nloc * e->W;
  e->ms = (int)(1 + 2 * cfg->swap_rate * cfg->n_rungs);                      // chain.cc:1192
  {
    // Rows that cross one shard boundary in one direction per step: at most one per walker, and a walker's boundary pair
    // is a candidate with probability <= swap_rate (maxswapsperstep * thresh / (Ntemps-1), chain.cc:1413-1416).
    // Default capacity: that binomial's mean + 8 sigma + 64, or every walker if that is less.
    const double pr = cfg->swap_rate < 1 ? (cfg->swap_rate > 0 ? cfg->swap_rate : 0) : 1;
    const double want = e->W * pr + 8 * std::sqrt(e->W * pr) + 64;
    e->row_cap = cfg->exchange_row_capacity > 0 ? cfg->exchange_row_capacity : (want < e->W ? (int)want : e->W);
    if (e->row_cap > e->W) e->row_cap = e->W;
  }
  e->thresh = (e->Nt - 1) * cfg->swap_rate / e->ms;                          // chain.cc:1413
  if (cfg->history_rungs < 0 || cfg->history_rungs > cfg->rung_count) { return fail(PTM_ERR_INVALID, "history_rungs out of range"); }
  if (cfg->history_rungs > 0) {
    if (cfg->history_capacity < 2) { return fail(PTM_ERR_INVALID, "history_capacity must be >= 2"); }
    // a rung touched twice in a step saves the row it held in between, which may be the row of the rung above: on a
    // shard that is not the ladder's last, the shard's top rung therefore cannot be recorded
    if (cfg->history_rungs == cfg->rung_count && cfg->rung_begin + cfg->rung_count < cfg->n_rungs) {
      return fail(PTM_ERR_UNSUPPORTED, "the top rung of a shard below the ladder's top cannot be recorded: history_rungs < rung_count");
    }
    if ((double)cfg->history_rungs * cfg->n_walkers >= (double)(1 << 30)) { return fail(PTM_ERR_INVALID, "too many recorded chains"); }
    e->hist.rungs = cfg->history_rungs; e->hist.cap = cfg->history_capacity; e->hist.HC = cfg->history_rungs * cfg->n_walkers;
  }
  if (cfg->device >= 0) { HIPCHK(hipSetDevice(cfg->device)); e->device = cfg->device; } else HIPCHK(hipGetDevice(&e->device));
  if (cfg->stream) e->stream = (hipStream_t)cfg->stream;
  else { HIPCHK(hipStreamCreate(&e->stream)); e->own_stream = true; }
  const size_t Nc = e->Nc, D = e->DP;  // every per-dimension table is padded to DP
  if ((rc = dalloc(&e->x, Nc * D)) || (rc = dalloc(&e->ll, Nc)) || (rc = dalloc(&e->lp, Nc))) return rc;
  HIPCHK(hipMemsetAsync(e->x, 0, Nc * D * 8, e->stream));
  if ((rc = dalloc(&e->ntries, Nc)) || (rc = dalloc(&e->naccept, Nc)) || (rc = dalloc(&e->last_type, Nc)) ||
      (rc = dalloc(&e->arr_below, (size_t)cfg->n_walkers)) || (rc = dalloc(&e->mv_src, (size_t)cfg->n_walkers * MVCAP)) ||
      (rc = dalloc(&e->mv_dst, (size_t)cfg->n_walkers * MVCAP)) || (rc = dalloc(&e->mv_n, (size_t)cfg->n_walkers)) ||
      (rc = dalloc(&e->arr_above, (size_t)cfg->n_walkers)) || (rc = dalloc(&e->touch, Nc)) || (rc = dalloc(&e->nhist, Nc)) ||
      (rc = dalloc(&e->err, 4)))
    return rc;
  if (cfg->map_rungs < 0 || cfg->map_rungs > cfg->rung_count) return fail(PTM_ERR_INVALID, "map_rungs out of range");
  if (cfg->map_rungs > 0) {
    if (cfg->map_rungs == cfg->rung_count && cfg->rung_begin + cfg->rung_count < cfg->n_rungs)
      return fail(PTM_ERR_UNSUPPORTED, "the top rung of a shard below the ladder's top cannot be tracked: map_rungs < rung_count");
    if ((double)cfg->map_rungs * cfg->n_walkers >= (double)(1 << 29)) return fail(PTM_ERR_INVALID, "too many MAP-tracked chains");
    e->map.rungs = cfg->map_rungs; e->map.MC = cfg->map_rungs * cfg->n_walkers;
    const size_t n = (size_t)e->map.MC;
    if ((rc = dalloc(&e->map.lpost, n)) || (rc = dalloc(&e->map.ll, n)) || (rc = dalloc(&e->map.lp, n)) || (rc = dalloc(&e->map.x, n * D)))
      return rc;
  }
  if (e->hist.rungs) {
    const size_t n = (size_t)e->hist.cap * e->hist.HC;
    e->hist_on_host = small_outputs_on_host(n * D * 8);
    if (e->hist_on_host) {
      if ((rc = halloc(e, &e->hist.x, n * D)) || (rc = halloc(e, &e->hist.ll, n)) || (rc = halloc(e, &e->hist.lp, n)) || (rc = halloc(e, &e->hist.meta, n)))
        return rc;
    } else if ((rc = dalloc(&e->hist.x, n * D)) || (rc = dalloc(&e->hist.ll, n)) || (rc = dalloc(&e->hist.lp, n)) || (rc = dalloc(&e->hist.meta, n)))
      return rc;
    HIPCHK(hipMemsetAsync(e->hist.meta, 0xFF, n * sizeof(int4), e->stream));   // saved row number -1: empty slot
  }
  const size_t np = (size_t)e->W * (e->Nt > 1 ? e->Nt - 1 : 1);
  if ((rc = dalloc(&e->swap_cnt, 2 * np)) || (rc = dalloc(&e->swap_log, (size_t)PTM_LOG_RING * e->W * e->ms)))
    return rc;
  HIPCHK(hipMemsetAsync(e->swap_cnt, 0, 2 * np * 8, e->stream));
  HIPCHK(hipMemsetAsync(e->swap_log, 0xFE, (size_t)PTM_LOG_RING * e->W * e->ms * 4, e->stream));  // 0xFEFEFEFE < 0: "none"
  HIPCHK(hipMemsetAsync(e->err, 0, 16, e->stream));
  HIPCHK(hipMemsetAsync(e->mv_n, 0, (size_t)e->W * 4, e->stream));
  if ((rc = dalloc(&e->blo, D)) || (rc = dalloc(&e->bhi, D)) || (rc = dalloc(&e->ptype, D)) || (rc = dalloc(&e->bmin, D)) ||
      (rc = dalloc(&e->bmax, D)) || (rc = dalloc(&e->plo, D)) || (rc = dalloc(&e->phi, D)) || (rc = dalloc(&e->pcoef, D)) ||
      (rc = dalloc(&e->P2, D * (D + 1) / 2)) || (rc = dalloc(&e->mean, D)) || (rc = dalloc(&e->beta, (size_t)e->Nt)) ||
      (rc = dalloc(&e->onedfrac, (size_t)e->nloc)) || (rc = dalloc(&e->P2_tiles, 144 * 64)) || (rc = dalloc(&e->box_row, 256)))
    return rc;
  {
    const int bd = e->DP == 128 ? 128 : (e->DP == 64 ? 64 : 32);
    std::vector<double> box(256);
    for (int d = 0; d < bd; ++d) { box[d] = -INFINITY; box[bd + d] = INFINITY; }
    if ((rc = upload(e->box_row, box.data(), 256, e->stream))) return rc;
    HIPCHK(hipMemsetAsync(e->P2_tiles, 0, (144 * 64) * 8, e->stream));
  }
  // defaults (and the permanent content of the pad dimensions): open bounds, flat prior with unbounded support
  std::vector<int> zi(D, 0);
  std::vector<double> zd(D, 0.0), one(D, 1.0), ninf(D, -INFINITY), pinf(D, INFINITY);
  if ((rc = upload(e->blo, zi.data(), D, e->stream)) || (rc = upload(e->bhi, zi.data(), D, e->stream)) ||
      (rc = upload(e->ptype, zi.data(), D, e->stream)) || (rc = upload(e->bmin, zd.data(), D, e->stream)) ||
      (rc = upload(e->bmax, zd.data(), D, e->stream)) || (rc = upload(e->plo, ninf.data(), D, e->stream)) ||
      (rc = upload(e->phi, pinf.data(), D, e->stream)) || (rc = upload(e->pcoef, one.data(), D, e->stream)) ||
      (rc = upload(e->mean, zd.data(), D, e->stream)))
    return rc;
  e->h_ptype.assign(D, 0); e->h_plo.assign(D, -INFINITY); e->h_phi.assign(D, INFINITY);
  e->all_uniform = 0;  // flat prior goes through the general product (pdf == 1)
  HIPCHK(hipEventCreate(&e->t0));
  HIPCHK(hipEventCreate(&e->t1));
  HIPCHK(hipStreamSynchronize(e->stream));
  return PTM_OK;
}

template <class T>
static int halloc(ptm_engine* e, T** p, size_t n) {
  void* h = nullptr;
  const size_t bytes = (n ? n : 1) * sizeof(T);
  HIPCHK(hipHostMalloc(&h, bytes, hipHostMallocMapped | hipHostMallocCoherent));
  e->host_blocks.push_back(std::make_pair((unsigned char*)h, bytes));
  *p = (T*)h;
  return PTM_OK;
}

// ---- batched reads (see ptm_engine::fetch_arena) ------------------------------------------------------------------------
static const size_t FETCH_ARENA = (size_t)4 << 20;
static int fetch_flush(ptm_engine* e) {
  HIPCHK(hipStreamSynchronize(e->stream));
  for (auto& f : e->fetch_after) f();
  e->fetch_after.clear(); e->fetch_big.clear(); e->fetch_used = 0;
  return PTM_OK;
}
// queues the copy of `bytes` at device address `dev` (ordered on the engine's stream, so it sees the state at the time of the
// call); *staged is where the bytes will be once fetch_flush has waited -- valid until that flush returns
static int fetch_raw(ptm_engine* e, const void* dev, size_t bytes, const unsigned char** staged) {
  for (const auto& b : e->host_blocks)
    if ((const unsigned char*)dev >= b.first && (const unsigned char*)dev < b.first + b.second) {   // already on the host: the flush's wait is all it needs
      *staged = (const unsigned char*)dev;
      return PTM_OK;
    }
  if (e->fetch_arena.empty()) e->fetch_arena.resize(FETCH_ARENA);
  const size_t need = (bytes + 63) & ~(size_t)63;
  // A big array, or one the arena has no room left for: its own buffer, copied at once (for a big array bandwidth, not call
  // latency, is the cost).  The arena is never recycled before the flush: pointers staged earlier in the same call or the same
  // ptm_batch_begin / ptm_batch_end bracket stay valid until their lambdas have run.
  if (need > FETCH_ARENA / 2 || e->fetch_used + need > FETCH_ARENA) {
    auto buf = std::make_shared<std::vector<unsigned char>>(bytes);
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(buf->data(), dev, bytes, hipMemcpyDeviceToHost));
    e->fetch_big.push_back(buf);
    *staged = buf->data();
    return PTM_OK;
  }
  unsigned char* at = e->fetch_arena.data() + e->fetch_used;
  HIPCHK(hipMemcpyAsync(at, dev, bytes, hipMemcpyDeviceToHost, e->stream));
  e->fetch_used += need;
  *staged = at;
  return PTM_OK;
}
// a read that fails cancels everything queued (the outputs of this call and of an open bracket are then never written: no
// lambda is left pointing at a caller's buffer)
static int fetch(ptm_engine* e, const void* dev, size_t bytes, const unsigned char** staged) {
  const int rc = fetch_raw(e, dev, bytes, staged);
  if (rc) { e->fetch_after.clear(); e->fetch_big.clear(); e->fetch_used = 0; }
  return rc;
}
static int fetch_done(ptm_engine* e) { return e->fetch_depth > 0 ? PTM_OK : fetch_flush(e); }
// Between ptm_batch_begin and ptm_batch_end only reads are allowed: a queued read of mapped host memory (a small population's
// history ring) is served in place at the flush, so nothing may change the engine's arrays in between.
// the launches of the persistent ladder kernel since the last look: committed, or repeated on the two-launch path (ladder_settle)
#define SETTLE(e) do { if (e) { const int _rc = ladder_settle(e); if (_rc) return _rc; } } while (0)
#define NO_BATCH(e, name) do { if ((e) && (e)->fetch_depth > 0) return fail(PTM_ERR_INVALID, name " between ptm_batch_begin and ptm_batch_end (only reads go there)"); } while (0)
#define FETCH(ptr, dev, bytes) do { int _rc = fetch(e, (dev), (bytes), &(ptr)); if (_rc) return _rc; } while (0)

extern "C" int ptm_batch_begin(ptm_engine* e) {
  SETTLE(e);
  if (!e) return fail(PTM_ERR_INVALID, "null engine");
  e->fetch_depth++;
  return PTM_OK;
}
extern "C" int ptm_batch_end(ptm_engine* e) {
  if (!e) return fail(PTM_ERR_INVALID, "null engine");
  if (e->fetch_depth <= 0) return fail(PTM_ERR_INVALID, "ptm_batch_end without ptm_batch_begin");
  if (--e->fetch_depth > 0) return PTM_OK;
  return fetch_flush(e);
}

extern "C" int ptm_engine_destroy(ptm_engine* e) {
  if (!e) return PTM_OK;
  (void)ptm_shard_finalize(e);
  (void)hipStreamSynchronize(e->stream);
  {
    std::lock_guard<std::mutex> lock(g_lad_mutex);
    if (g_lad_last == e) g_lad_last = nullptr;
  }
  if (e->lad_event) (void)hipEventDestroy(e->lad_event);
  if (e->shared_handover) e->xprop = e->llike_new = nullptr, e->hastings = nullptr, e->htype = nullptr, e->hvalid = e->acc_out = nullptr;   // (these are the pinned vectors)
  if (e->hist_on_host) e->hist.x = e->hist.ll = e->hist.lp = e->hist.beta = nullptr, e->hist.meta = nullptr;
  for (auto& b : e->host_blocks) (void)hipHostFree(b.first);
  void* ptrs[] = {e->x, e->ll, e->lp, e->ntries, e->naccept, e->last_type, e->arr_below, e->arr_above, e->mv_src, e->mv_dst, e->mv_n,
                  e->err, e->nhist, e->swap_cnt, e->touch, e->swap_log, e->hist.x, e->hist.ll, e->hist.lp, e->hist.meta, e->map.lpost, e->map.ll, e->map.lp, e->map.x, e->blo,
                  e->bhi, e->ptype, e->bmin, e->bmax, e->plo, e->phi, e->pcoef, e->P2, e->mean, e->beta, e->prop, e->prop_tiles, e->P2_tiles, e->box_row, e->onedfrac, e->mix, e->beta_w, e->betaC, e->beta_add, e->hist.beta, e->xprop, e->lprior_new, e->llike_new, e->hastings, e->htype, e->hvalid, e->acc_out, e->cidx, e->ccnt,
                  e->pub_x, e->lad_flags, e->lad_prof, e->shard_ends, e->redo_flag, e->sums, e->de_init, e->de_hast, e->de_type};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (e->h_sums) (void)hipHostFree(e->h_sums);
  for (hipEvent_t ev : e->kev) (void)hipEventDestroy(ev);
  if (e->t0) (void)hipEventDestroy(e->t0);
  if (e->t1) (void)hipEventDestroy(e->t1);
  if (e->own_stream) (void)hipStreamDestroy(e->stream);
  delete e;
  return PTM_OK;
}

// ---- problem description ----------------------------------------------------------------------------------
extern "C" int ptm_set_bounds(ptm_engine* e, const int32_t* lo, const int32_t* hi, const double* xmin, const double* xmax) {
  SETTLE(e);
  if (!e || !lo || !hi || !xmin || !xmax) return fail(PTM_ERR_INVALID, "null argument");
  int rc;
  const int D = e->D;
  e->has_bounds = 0;
  e->bounds_box = 1;
  for (int d = 0; d < D; ++d) {
    if (lo[d] < 0 || lo[d] > 3 || hi[d] < 0 || hi[d] > 3) return fail(PTM_ERR_INVALID, "bad boundary type in dimension %d", d);
    if (lo[d] != PTM_BOUND_OPEN || hi[d] != PTM_BOUND_OPEN) e->has_bounds = 1;
    if ((lo[d] != PTM_BOUND_OPEN && lo[d] != PTM_BOUND_LIMIT) || (hi[d] != PTM_BOUND_OPEN && hi[d] != PTM_BOUND_LIMIT)) e->bounds_box = 0;
  }
  // Q9: state::add() builds on state(space,n) = enforced zero vector (states.cc:183-192,205-214).  Plain host
  // arithmetic on D constants; zero is only ever *rejected* by a `limit` bound, any other bound type maps it.
  e->origin_valid = 1;
  for (int d = 0; d < D; ++d) {
    const bool lw = lo[d] == PTM_BOUND_WRAP, hw = hi[d] == PTM_BOUND_WRAP;
    if (lw != hw) { e->origin_valid = 0; break; }
    if (lw) { if (xmax[d] - xmin[d] <= 0) { e->origin_valid = 0; break; } continue; }
    if (lo[d] == PTM_BOUND_REFLECT && hi[d] == PTM_BOUND_REFLECT) { if (xmax[d] - xmin[d] <= 0) { e->origin_valid = 0; break; } continue; }
    double z = 0.0;
    if (lo[d] == PTM_BOUND_REFLECT && z < xmin[d]) z = xmin[d] + (xmin[d] - z);
    else if (hi[d] == PTM_BOUND_REFLECT && z > xmax[d]) z = xmax[d] - (z - xmax[d]);
    if (lo[d] == PTM_BOUND_LIMIT && z < xmin[d]) { e->origin_valid = 0; break; }
    if (hi[d] == PTM_BOUND_LIMIT && z > xmax[d]) { e->origin_valid = 0; break; }
  }
  if ((rc = upload(e->blo, lo, D, e->stream)) || (rc = upload(e->bhi, hi, D, e->stream)) ||      // pad dimensions stay open
      (rc = upload(e->bmin, xmin, D, e->stream)) || (rc = upload(e->bmax, xmax, D, e->stream)))
    return rc;
  return PTM_OK;
}

static inline size_t host_row_pos(size_t DP, size_t d);
extern "C" int ptm_set_prior(ptm_engine* e, const int32_t* types, const double* c, const double* h) {
  SETTLE(e);
  if (!e || !types || !c || !h) return fail(PTM_ERR_INVALID, "null argument");
  const int D = e->D;
  std::vector<double> lo(D), hi(D), coef(D);
  std::vector<int> ty(D);
  e->all_uniform = 1;
  double prod = 1;
  for (int d = 0; d < D; ++d) {
    ty[d] = types[d];
    switch (types[d]) {  // mixed_dist_product ctor (probability_function.cc:232-254) and the 1-D ctors it calls
      case PTM_PRIOR_FLAT: lo[d] = -INFINITY; hi[d] = INFINITY; coef[d] = 1; e->all_uniform = 0; break;
      case PTM_PRIOR_UNIFORM: lo[d] = c[d] - h[d]; hi[d] = c[d] + h[d]; coef[d] = 1 / (hi[d] - lo[d]); prod *= coef[d]; break;
      case PTM_PRIOR_GAUSSIAN: lo[d] = c[d]; hi[d] = h[d]; coef[d] = 0; e->all_uniform = 0; break;
      case PTM_PRIOR_POLAR: {  // UniformPolarDist ctor: clamps only the normalisation (ProbabilityDist.h:181-186)
        double a = c[d] - h[d], b = c[d] + h[d];
        lo[d] = a; hi[d] = b;
        if (a < 0) a = 0;
        if (b > M_PI) b = M_PI;
        coef[d] = -std::cos(b) + std::cos(a);
        e->all_uniform = 0;
        break;
      }
      case PTM_PRIOR_COPOLAR: {  // ProbabilityDist.h:227-232
        double a = c[d] - h[d], b = c[d] + h[d];
        lo[d] = a; hi[d] = b;
        if (a < -M_PI / 2) a = -M_PI / 2;
        if (b > M_PI / 2) b = M_PI / 2;
        coef[d] = std::sin(b) - std::sin(a);
        e->all_uniform = 0;
        break;
      }
      case PTM_PRIOR_LOG:  // probability_function.cc:245-250, ProbabilityDist.h:109-117
        if (c[d] <= 0 || h[d] <= 1) return fail(PTM_ERR_INVALID, "log prior needs center>0 and halfwidth>1 (dimension %d)", d);
        lo[d] = c[d] / h[d]; hi[d] = c[d] * h[d]; coef[d] = std::log(hi[d]) - std::log(lo[d]);
        e->all_uniform = 0;
        break;
      default: return fail(PTM_ERR_INVALID, "unknown prior type %d in dimension %d", types[d], d);
    }
  }
  e->lprior_const = std::log(prod);  // the reference takes libm log of the product of these constants on every call
  e->lp_is_const = false;            // (re-established when states are set)
  for (int d = 0; d < D; ++d) { e->h_ptype[d] = ty[d]; e->h_plo[d] = lo[d]; e->h_phi[d] = hi[d]; }
  int rc;
  if ((rc = upload(e->ptype, ty.data(), D, e->stream)) || (rc = upload(e->plo, lo.data(), D, e->stream)) ||
      (rc = upload(e->phi, hi.data(), D, e->stream)) || (rc = upload(e->pcoef, coef.data(), D, e->stream)))
    return rc;
  if (e->DP == 32 || e->DP == 64 || e->DP == 128) {   // the box in row layout, for the MFMA kernels (pad dimensions stay unbounded)
    const int bd = e->DP;
    std::vector<double> box(2 * bd);
    for (int d = 0; d < bd; ++d) {
      box[host_row_pos(bd, d)] = d < D ? lo[d] : -INFINITY;
      box[bd + host_row_pos(bd, d)] = d < D ? hi[d] : INFINITY;
    }
    if ((rc = upload(e->box_row, box.data(), 2 * bd, e->stream))) return rc;
  }
  return PTM_OK;
}

extern "C" int ptm_set_target_gaussian(ptm_engine* e, const double* mean, const double* P, double like0) {
  SETTLE(e);
  if (!e || !P) return fail(PTM_ERR_INVALID, "null argument");
  const int D = e->D, DP = e->DP;
  std::vector<double> packed((size_t)DP * (DP + 1) / 2, 0.0);
  for (int i = 0; i < D; ++i) {
    const size_t o = (size_t)i * (i + 1) / 2;
    for (int j = 0; j < i; ++j) packed[o + j] = P[i * D + j] + P[j * D + i];
    packed[o + i] = P[i * D + i];
  }
  int rc;
  if ((rc = upload(e->P2, packed.data(), packed.size(), e->stream))) return rc;
  if (DP == 32) {
    // A-operand tiles of the MFMA kernel: tile t = step*2 + rowtile, lane 16k + i holds P2[16 rowtile + i][4 step + k]
    std::vector<double> tiles(16 * 64, 0.0);
    for (int t = 0; t < 16; ++t) {
      const int rt = t & 1, m = t >> 1;
      for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 16; ++i) {
          const int row = 16 * rt + i, col = 4 * m + k;
          if (row < D && col <= row) tiles[(size_t)t * 64 + 16 * k + i] = packed[(size_t)row * (row + 1) / 2 + col];
        }
    }
    if ((rc = upload(e->P2_tiles, tiles.data(), tiles.size(), e->stream))) return rc;
  }
  if (DP == 64 || DP == 128) {
    // A-operand tiles of ptm_mfma64_kernel.hpp / ptm_mfma128_kernel.hpp: (row tile rt, step m) with m <= 4 rt + 3 at base(rt) + m,
    // base(rt) = 2 rt (rt + 1) = 0, 4, 12, 24, ..; lane 16k + i holds P2[16 rt + i][4 m + k] (lower triangle, off-diagonals doubled;
    // zeros above the diagonal)
    const int NT = DP / 16;
    std::vector<double> tiles((size_t)2 * NT * (NT + 1) * 64, 0.0);
    for (int rt = 0; rt < NT; ++rt)
      for (int m = 0; m <= 4 * rt + 3; ++m)
        for (int k = 0; k < 4; ++k)
          for (int i = 0; i < 16; ++i) {
            const int row = 16 * rt + i, col = 4 * m + k;
            if (row < D && col <= row) tiles[(size_t)(2 * rt * (rt + 1) + m) * 64 + 16 * k + i] = packed[(size_t)row * (row + 1) / 2 + col];
          }
    if ((rc = upload(e->P2_tiles, tiles.data(), tiles.size(), e->stream))) return rc;
  }
  e->has_mean = mean ? 1 : 0;
  if (mean && (rc = upload(e->mean, mean, D, e->stream))) return rc;
  e->like0 = like0;
  e->have_target = 1;
  e->cb = nullptr;  // a device target replaces a host callback
  return PTM_OK;
}

// the proposal hand-over buffers both host paths use (callback likelihood: propose pass -> host; host-side proposals: host ->
// kernel): proposed states in row layout, the gate bytes right behind them (one device-to-host copy fetches both)
static int alloc_proposal_buffers(ptm_engine* e) {
  const size_t Nc = e->Nc, DP = e->DP;
  int rc;
  // A small population's step on the host paths is a chain of copies and waits, each ~5-10 us of runtime calls for a few
  // kilobytes: there the kernels read and write the pinned host images directly (coherent mapped memory; a kernel's writes are
  // on the host when its stream has drained, the host's are there for the next launch) and the copies fall away.
  const char* zc = getenv("PTM_SHARED_HANDOVER");
  if (!e->xprop) e->shared_handover = Nc * DP * 8 <= ((size_t)1 << 20) && !(zc && *zc == '0');
  e->h_xprop.resize(Nc * DP + (Nc + 7) / 8); e->h_llnew.assign(Nc, 0.0);
  e->h_gate = reinterpret_cast<unsigned char*>(e->h_xprop.data() + Nc * DP);
  if (e->shared_handover) {
    HIPCHK(hipHostGetDevicePointer((void**)&e->xprop, e->h_xprop.data(), 0));
    HIPCHK(hipHostGetDevicePointer((void**)&e->llike_new, e->h_llnew.data(), 0));
    if (!e->lprior_new && (rc = dalloc(&e->lprior_new, Nc))) return rc;
  } else if (!e->xprop && ((rc = dalloc(&e->xprop, Nc * DP + (Nc + 7) / 8)) || (rc = dalloc(&e->lprior_new, Nc)) || (rc = dalloc(&e->llike_new, Nc))))
    return rc;
  e->gate = reinterpret_cast<unsigned char*>(e->xprop + Nc * DP);
  return PTM_OK;
}

extern "C" int ptm_set_target_callback(ptm_engine* e, ptm_loglike_batch_fn fn, void* user) {
  SETTLE(e);
  if (!e || !fn) return fail(PTM_ERR_INVALID, "null argument");
  int rc = alloc_proposal_buffers(e);
  if (rc) return rc;
  e->cb = fn; e->cb_user = user;
  e->have_target = 1;
  return PTM_OK;
}

extern "C" int ptm_set_prior_callback(ptm_engine* e, ptm_logprior_batch_fn fn, void* user) {
  SETTLE(e);
  if (!e) return fail(PTM_ERR_INVALID, "null engine");
  e->prior_cb = fn; e->prior_user = fn ? user : nullptr;
  e->lp_is_const = false;
  return PTM_OK;
}

extern "C" int ptm_set_proposal_callback(ptm_engine* e, ptm_propose_batch_fn propose, ptm_proposal_result_fn result, void* user) {
  SETTLE(e);
  if (!e) return fail(PTM_ERR_INVALID, "null engine");
  if (!propose) {   // back to the device proposals, if any were set
    e->pcb = nullptr; e->pres = nullptr; e->pcb_user = nullptr;
    e->have_prop = e->prop ? 1 : 0;
    return PTM_OK;
  }
  const size_t Nc = e->Nc, DP = e->DP;
  int rc = alloc_proposal_buffers(e);
  if (rc) return rc;
  e->h_rows.resize(Nc * DP); e->h_hast.assign(Nc, 0.0); e->h_type.assign(Nc, 0); e->h_valid.assign(Nc, 0); e->h_touch.assign(Nc, 0); e->h_acc.assign(Nc, 0);
  if (e->shared_handover) {
    HIPCHK(hipHostGetDevicePointer((void**)&e->hastings, e->h_hast.data(), 0));
    HIPCHK(hipHostGetDevicePointer((void**)&e->htype, e->h_type.data(), 0));
    HIPCHK(hipHostGetDevicePointer((void**)&e->hvalid, e->h_valid.data(), 0));
    HIPCHK(hipHostGetDevicePointer((void**)&e->acc_out, e->h_acc.data(), 0));
  } else if (!e->hastings && ((rc = dalloc(&e->hastings, Nc)) || (rc = dalloc(&e->htype, Nc)) || (rc = dalloc(&e->hvalid, Nc)) || (rc = dalloc(&e->acc_out, Nc))))
    return rc;
  if (!e->onedfrac) return fail(PTM_ERR_INVALID, "engine not built");
  e->pcb = propose; e->pres = result; e->pcb_user = user;
  e->have_prop = 1;
  return PTM_OK;
}

// the user's batched log-likelihood on the chains picked by `pick` (row indices into the padded row image `rows`)
static int call_user(ptm_engine* e, const pinned_vector<double>& rows, const std::vector<size_t>& pick, std::vector<double>& out) {
  const size_t D = e->D, DP = e->DP, n = pick.size();
  out.assign(n, 0.0);
  if (!n) return PTM_OK;
  e->h_batch.resize(n * D);
  for (size_t k = 0; k < n; ++k)
    for (size_t d = 0; d < D; ++d) e->h_batch[k * D + d] = rows[pick[k] * DP + host_row_pos(DP, d)];
  e->cb(e->cb_user, e->h_batch.data(), (int)n, (int)D, out.data());
  return PTM_OK;
}

// the user's batched log-prior on the chains picked by `pick`
static int call_user_prior(ptm_engine* e, const pinned_vector<double>& rows, const std::vector<size_t>& pick, std::vector<double>& out) {
  const size_t D = e->D, DP = e->DP, n = pick.size();
  out.assign(n, 0.0);
  if (!n) return PTM_OK;
  e->h_batch.resize(n * D);
  for (size_t k = 0; k < n; ++k)
    for (size_t d = 0; d < D; ++d) e->h_batch[k * D + d] = rows[pick[k] * DP + host_row_pos(DP, d)];
  e->prior_cb(e->prior_user, e->h_batch.data(), (int)n, (int)D, out.data());
  return PTM_OK;
}
// host prior: the log-priors of the states now in e->x (their rows in e->h_xprop) replace what the device prior gave -- except
// where that is -inf by the state's invalidity (the device's flat stand-in prior has no other way to say -inf)
static int host_prior_of_states(ptm_engine* e) {
  const size_t Nc = e->Nc;
  std::vector<double> lp(Nc), out;
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpy(lp.data(), e->lp, Nc * 8, hipMemcpyDeviceToHost));
  std::vector<size_t> pick;
  for (size_t c = 0; c < Nc; ++c)
    if (lp[c] > -__builtin_inf()) pick.push_back(c);
  int rc = call_user_prior(e, e->h_xprop, pick, out);
  if (rc) return rc;
  for (size_t k = 0; k < pick.size(); ++k) lp[pick[k]] = out[k];
  return upload(e->lp, lp.data(), Nc, e->stream);
}

extern "C" int ptm_set_ladder(ptm_engine* e, const double* beta) {
  SETTLE(e);
  NO_BATCH(e, "ptm_set_ladder");
  if (!e || !beta) return fail(PTM_ERR_INVALID, "null argument");
  e->h_beta.assign(beta, beta + e->Nt);
  int rc = upload(e->beta, beta, (size_t)e->Nt, e->stream);
  if (rc) return rc;
  e->have_ladder = 1;
  if (e->beta_w) return fill_evolving_ladders(e);   // evolving ladders restart from the new common ladder
  return PTM_OK;
}

// every ladder starts from the common ladder
static int fill_evolving_ladders(ptm_engine* e) {
  std::vector<double> bw((size_t)e->W * e->Nt);
  for (int w = 0; w < e->W; ++w) std::copy(e->h_beta.begin(), e->h_beta.end(), bw.begin() + (size_t)w * e->Nt);
  int rc = upload(e->beta_w, bw.data(), bw.size(), e->stream);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(e->stream));   // bw leaves scope
  return launch_beta_transpose(e);
}

extern "C" int ptm_set_evolve_temps(ptm_engine* e, double rate, double lpost_cut) {
  SETTLE(e);
  if (!e) return fail(PTM_ERR_INVALID, "null engine");
  if (!(rate > 0)) {
    if (e->evolve_rate > 0) return fail(PTM_ERR_UNSUPPORTED, "an evolving ladder cannot be frozen again (the reference has no such call either)");
    return PTM_OK;   // evolve_temps is never called with rate <= 0 (ptmcmc.cc:512)
  }
  if (e->nloc != e->Nt) {
    // A rung shard: every accepted exchange renormalises ALL gaps and every later trial of the step sees it, so each shard replays
    // the whole ladder's trials -- from the whole ladder's llikes, which the caller gathers each step (ptm_exchange_decide_gathered;
    // the reference's MPI ranks do exactly that: gather_llikes, chain.cc:1433-1435,1950-1972).  Every shard keeps every ladder's
    // temperatures [W][Nt]; the gathered llikes (+ lpriors with a posterior-ordering cut) are [Nt][W] doubles per step.
    // (history / MAP tracking of the shard's own rungs: every shard replays every pick and so knows the temperature each of its
    //  rungs had at each add_state of the exchange phase -- beta_add, chain-indexed; the shard's top rung cannot be recorded unless
    //  it is the ladder's, as on a fixed ladder: ptm_engine_create)
    if ((double)e->W * e->Nt * 16.0 > 512.0 * 1024 * 1024)
      return fail(PTM_ERR_UNSUPPORTED, "evolving ladders on a rung shard gather %d x %d llikes (and lpriors) per step: more than 512 MB -- split such a population by walkers "
                                       "(ptm_config.walker_begin)", e->Nt, e->W);
  }
  if (!e->have_ladder) return fail(PTM_ERR_INVALID, "set the ladder first (ptm_set_ladder)");
  int rc;
  if (!e->beta_w) {
    if ((rc = dalloc(&e->beta_w, (size_t)e->W * e->Nt)) || (rc = dalloc(&e->betaC, (size_t)e->Nc))) return rc;
    if ((rc = fill_evolving_ladders(e))) return rc;
    // history / MAP tracking: an add_state of the exchange phase sees its rung's temperature between two pries of the step
    if ((e->hist.rungs || e->map.rungs) && (rc = dalloc(&e->beta_add, (size_t)e->Nc))) return rc;
    if (e->hist.rungs) {   // every saved row keeps the temperature it was saved at; the rows so far: the common ladder's
      const size_t n = (size_t)e->hist.cap * e->hist.HC;
      if ((rc = e->hist_on_host ? halloc(e, &e->hist.beta, n) : dalloc(&e->hist.beta, n))) return rc;
      hipLaunchKernelGGL(hist_beta_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, e->hist, e->beta, e->W, e->r0);
      HIPCHK(hipGetLastError());
    }
  }
  e->evolve_rate = rate;
  e->evolve_cut = lpost_cut >= 0 ? lpost_cut : -1;
  return PTM_OK;
}

extern "C" int ptm_get_invtemps(ptm_engine* e, double* beta) {
  SETTLE(e);
  if (!e || !beta) return fail(PTM_ERR_INVALID, "null argument");
  if (!e->have_ladder) return fail(PTM_ERR_INVALID, "no ladder set");
  if (e->beta_w) {
    const size_t n = (size_t)e->W * e->Nt * 8;
    const unsigned char* s;
    FETCH(s, e->beta_w, n);
    e->fetch_after.push_back([=] { memcpy(beta, s, n); });
    return fetch_done(e);
  }
  for (int w = 0; w < e->W; ++w) std::copy(e->h_beta.begin(), e->h_beta.end(), beta + (size_t)w * e->Nt);
  return PTM_OK;
}

extern "C" int ptm_get_history_invtemps(ptm_engine* e, double* beta) {
  SETTLE(e);
  if (!e || !beta) return fail(PTM_ERR_INVALID, "null argument");
  if (!e->hist.rungs) return fail(PTM_ERR_INVALID, "this engine keeps no history (ptm_config.history_rungs)");
  if (!e->have_ladder) return fail(PTM_ERR_INVALID, "no ladder set");
  const size_t n = (size_t)e->hist.cap * e->hist.HC;
  if (e->hist.beta) {
    const unsigned char* s;
    FETCH(s, e->hist.beta, n * 8);
    e->fetch_after.push_back([=] { memcpy(beta, s, n * 8); });
    return fetch_done(e);
  }
  for (size_t i = 0; i < n; ++i) beta[i] = e->h_beta[e->r0 + (int)(i % e->hist.HC) / e->W];
  return PTM_OK;
}

extern "C" int ptm_set_invtemps(ptm_engine* e, const double* beta) {
  SETTLE(e);
  NO_BATCH(e, "ptm_set_invtemps");
  if (!e || !beta) return fail(PTM_ERR_INVALID, "null argument");
  if (!e->beta_w) return fail(PTM_ERR_INVALID, "per-ladder temperatures exist only once the ladders evolve (ptm_set_evolve_temps)");
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpy(e->beta_w, beta, (size_t)e->W * e->Nt * 8, hipMemcpyHostToDevice));
  return launch_beta_transpose(e);
}

extern "C" int ptm_set_proposals(ptm_engine* e, int kind, const double* factors, const double* one_d_frac) {
  SETTLE(e);
  if (!e || !factors) return fail(PTM_ERR_INVALID, "null argument");
  const int D = e->D, DP = e->DP, nloc = e->nloc;
  int stride;
  std::vector<double> packed;
  if (kind == PTM_PROP_DIAG) {
    stride = DP;
    packed.assign((size_t)nloc * stride, 0.0);
    for (int r = 0; r < nloc; ++r)
      for (int d = 0; d < D; ++d) packed[(size_t)r * stride + d] = factors[(size_t)r * D + d];
  } else if (kind == PTM_PROP_DENSE || kind == PTM_PROP_LOWER) {
    stride = DP * DP;  // column-major, padded: element (i,j) at j*DP + i; a Cholesky factor keeps its zeros
    packed.assign((size_t)nloc * stride, 0.0);
    for (int r = 0; r < nloc; ++r)
      for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) packed[(size_t)r * stride + (size_t)j * DP + i] = factors[(size_t)r * D * D + (size_t)i * D + j];
    if (kind == PTM_PROP_LOWER)
      for (int r = 0; r < nloc; ++r)
        for (int i = 0; i < D; ++i)
          for (int j = i + 1; j < D; ++j)
            if (factors[(size_t)r * D * D + (size_t)i * D + j] != 0.0)
              return fail(PTM_ERR_INVALID, "PTM_PROP_LOWER factor of local rung %d has a non-zero above the diagonal", r);
  } else {
    return fail(PTM_ERR_INVALID, "unknown proposal kind %d", kind);
  }
  if (e->prop) { HIPCHK(hipStreamSynchronize(e->stream)); HIPCHK(hipFree(e->prop)); e->prop = nullptr; }
  if (e->prop_tiles) { HIPCHK(hipFree(e->prop_tiles)); e->prop_tiles = nullptr; }
  int rc;
  if ((rc = dalloc(&e->prop, packed.size())) || (rc = upload(e->prop, packed.data(), packed.size(), e->stream))) return rc;
  if (DP == 32) {
    // A-operand tiles of the MFMA kernel: tile t = (half*4 + slot)*2 + rowtile, lane 16k + i holds
    // T[16 rowtile + i][16 half + 4k + slot].  A diagonal proposal (sigmas) goes through the same kernel as the diagonal
    // matrix it is: fma(sigma_i, z_i, +0) is the product sigma_i z_i, and the zero terms around it change nothing.
    std::vector<double> tiles((size_t)nloc * 16 * 64, 0.0);
    for (int r = 0; r < nloc; ++r)
      for (int t = 0; t < 16; ++t) {
        const int rt = t & 1, sl = (t >> 1) & 3, hb = t >> 3;
        for (int k = 0; k < 4; ++k)
          for (int i = 0; i < 16; ++i) {
            const int row = 16 * rt + i, col = 16 * hb + 4 * k + sl;
            if (row < D && col < D)
              tiles[((size_t)r * 16 + t) * 64 + 16 * k + i] =
                  kind == PTM_PROP_DIAG ? (row == col ? factors[(size_t)r * D + row] : 0.0) : factors[(size_t)r * D * D + (size_t)row * D + col];
          }
      }
    if ((rc = dalloc(&e->prop_tiles, tiles.size())) || (rc = upload(e->prop_tiles, tiles.data(), tiles.size(), e->stream))) return rc;
  }
  if (DP == 64 || DP == 128) {
    // A-operand tiles of ptm_mfma64_kernel.hpp / ptm_mfma128_kernel.hpp: tile t = (half*4 + slot)*NT + rowtile (NT = DP / 16 row
    // tiles), lane 16k + i holds T[16 rowtile + i][16 half + 4k + slot]
    const int NT = DP / 16, ntile = NT * NT * 4;
    std::vector<double> tiles((size_t)nloc * ntile * 64, 0.0);
    for (int r = 0; r < nloc; ++r)
      for (int t = 0; t < ntile; ++t) {
        const int rt = t % NT, sl = (t / NT) & 3, hb = t / (4 * NT);
        for (int k = 0; k < 4; ++k)
          for (int i = 0; i < 16; ++i) {
            const int row = 16 * rt + i, col = 16 * hb + 4 * k + sl;
            if (row < D && col < D)
              tiles[((size_t)r * ntile + t) * 64 + 16 * k + i] =
                  kind == PTM_PROP_DIAG ? (row == col ? factors[(size_t)r * D + row] : 0.0) : factors[(size_t)r * D * D + (size_t)row * D + col];
          }
      }
    if ((rc = dalloc(&e->prop_tiles, tiles.size())) || (rc = upload(e->prop_tiles, tiles.data(), tiles.size(), e->stream))) return rc;
  }
  std::vector<double> f(nloc, 0.0);
  e->any_oned = 0;
  if (one_d_frac)
    for (int r = 0; r < nloc; ++r) {
      if (one_d_frac[r] < 0 || one_d_frac[r] > 1) return fail(PTM_ERR_INVALID, "oneDfrac must be in [0,1]");  // hh:159-162
      f[r] = one_d_frac[r];
      if (f[r] > 0) e->any_oned = 1;
    }
  if ((rc = upload(e->onedfrac, f.data(), (size_t)nloc, e->stream))) return rc;
  e->prop_kind = kind; e->prop_stride = stride; e->have_prop = 1;
  return PTM_OK;
}

// A proposal_distribution_set of Gaussian members (proposal_distribution.cc:99-129) whose members are scalar multiples of
// the rung's factor -- the sampler's default Gaussian recipe (six diagonal Gaussians a factor gauss_step_fac apart with
// doubling shares, ptmcmc.cc:117-139).  Per local rung K members: cumulative shares (proposal_distribution_set's
// bin_max), scales, oneDfracs.  K = 0 removes the mixture.  last_type becomes member + 10 * (member's type).
extern "C" int ptm_set_proposal_mixture(ptm_engine* e, int K, const double* cum_shares, const double* scales, const double* one_d_fracs) {
  SETTLE(e);
  if (!e) return fail(PTM_ERR_INVALID, "null engine");
  if (!e->have_prop) return fail(PTM_ERR_INVALID, "set the base proposals first (ptm_set_proposals)");
  if (K < 0 || K > 64) return fail(PTM_ERR_INVALID, "mixture size must be in 0..64");
  if (e->mix) { HIPCHK(hipStreamSynchronize(e->stream)); HIPCHK(hipFree(e->mix)); e->mix = nullptr; }
  e->mix_K = 0;
  if (K == 0) return PTM_OK;
  if (!cum_shares || !scales || !one_d_fracs) return fail(PTM_ERR_INVALID, "null argument");
  const int nloc = e->nloc;
  std::vector<double> t((size_t)nloc * K * 3);
  int oned = 0;
  for (int r = 0; r < nloc; ++r)
    for (int k = 0; k < K; ++k) {
      const size_t i = (size_t)r * K + k;
      if (k > 0 && cum_shares[i] < cum_shares[i - 1]) return fail(PTM_ERR_INVALID, "cumulative shares must not decrease (rung %d)", r);
      if (one_d_fracs[i] < 0 || one_d_fracs[i] > 1) return fail(PTM_ERR_INVALID, "oneDfrac must be in [0,1]");
      t[3 * i] = cum_shares[i]; t[3 * i + 1] = scales[i]; t[3 * i + 2] = one_d_fracs[i];
      if (one_d_fracs[i] > 0) oned = 1;
    }
  int rc;
  if ((rc = dalloc(&e->mix, t.size())) || (rc = upload(e->mix, t.data(), t.size(), e->stream))) return rc;
  e->mix_K = K;
  if (oned) e->any_oned = 1;
  return PTM_OK;
}

// Differential evolution drawn on the device (include/ptm_engine.h)
extern "C" int ptm_set_proposal_de(ptm_engine* e, const ptm_de_params* q, int n_init_extra, const double* init_rows) {
  SETTLE(e);
  if (!e) return fail(PTM_ERR_INVALID, "null engine");
  NO_BATCH(e, "ptm_set_proposal_de");
  HIPCHK(hipStreamSynchronize(e->stream));
  if (e->de_init) { HIPCHK(hipFree(e->de_init)); e->de_init = nullptr; }
  e->de_on = false; e->de_init_extra = 0;
  if (!q) return PTM_OK;
  if (e->DP > 128) return fail(PTM_ERR_UNSUPPORTED, "differential evolution on the device is built for up to 128 dimensions; above, draw it on the host (ptm_set_proposal_callback)");
  if (e->nloc != e->Nt)
    return fail(PTM_ERR_UNSUPPORTED, "differential evolution draws from EVERY rung's saved history, and a rung shard cannot record its top rung's: the row that "
                                     "rung holds between two exchanges of one step sits in the shard above (error bit 16).  Split the population by walkers "
                                     "(whole ladders per GPU: walker_begin) or draw on the host (ptm_set_proposal_callback)");
  if (e->hist.rungs != e->nloc || e->hist.cap < 2)
    return fail(PTM_ERR_INVALID, "differential evolution draws from every rung's saved history: create the engine with history_rungs = rung_count and a "
                                 "history_capacity that holds every row of the run");
  if (e->pcb) return fail(PTM_ERR_INVALID, "host-side proposals are set (ptm_set_proposal_callback): they replace every device proposal");
  if (!(q->snooker >= 0 && q->snooker <= 1) || !(q->gamma_one_frac >= 0 && q->gamma_one_frac <= 1) || !(q->reduce_gamma > 0) ||
      !(q->ignore_frac >= 0 && q->ignore_frac < 1) || n_init_extra < 0 || (n_init_extra > 0 && !init_rows))
    return fail(PTM_ERR_INVALID, "differential evolution: snooker and gamma_one_frac in [0, 1], reduce_gamma > 0, ignore_frac in [0, 1), init rows given");
  int rc;
  if (n_init_extra > 0) {
    const size_t Nc = e->Nc, D = e->D, DP = e->DP;
    std::vector<double> img((size_t)n_init_extra * Nc * DP, 0.0);
    for (size_t k = 0; k < (size_t)n_init_extra; ++k)
      for (size_t c = 0; c < Nc; ++c)
        for (size_t d = 0; d < D; ++d) img[(k * Nc + c) * DP + host_row_pos(DP, d)] = init_rows[(k * Nc + c) * D + d];
    if ((rc = dalloc(&e->de_init, img.size()))) return rc;
    HIPCHK(hipMemcpy(e->de_init, img.data(), img.size() * 8, hipMemcpyHostToDevice));
  }
  if (!e->de_hast && ((rc = dalloc(&e->de_hast, (size_t)e->Nc)) || (rc = dalloc(&e->de_type, (size_t)e->Nc)))) return rc;
  e->de = *q;
  e->de_init_extra = n_init_extra;
  e->de_on = true;
  return PTM_OK;
}

// One rung's factor replaced between steps -- what user_gaussian_prop::check_update achieves in the reference when its
// callback returns a new covariance for a chain (proposal_distribution.cc:406-441, reset_dist :340-403).  Same kind and
// storage as the factors set by ptm_set_proposals; one_d_frac < 0 keeps the rung's current value.
extern "C" int ptm_set_proposal_rung(ptm_engine* e, int local_rung, const double* factor, double one_d_frac) {
  SETTLE(e);
  if (!e || !factor) return fail(PTM_ERR_INVALID, "null argument");
  if (!e->have_prop) return fail(PTM_ERR_INVALID, "set all proposals first (ptm_set_proposals)");
  if (local_rung < 0 || local_rung >= e->nloc) return fail(PTM_ERR_INVALID, "rung out of the shard");
  const int D = e->D, DP = e->DP, kind = e->prop_kind == KIND_DIAG ? PTM_PROP_DIAG : (e->prop_kind == KIND_LOWER ? PTM_PROP_LOWER : PTM_PROP_DENSE);
  std::vector<double> packed((size_t)e->prop_stride, 0.0);
  if (kind == PTM_PROP_DIAG) {
    for (int d = 0; d < D; ++d) packed[d] = factor[d];
  } else {
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < D; ++j) {
        if (kind == PTM_PROP_LOWER && j > i && factor[(size_t)i * D + j] != 0.0)
          return fail(PTM_ERR_INVALID, "PTM_PROP_LOWER factor has a non-zero above the diagonal");
        packed[(size_t)j * DP + i] = factor[(size_t)i * D + j];
      }
  }
  int rc;
  if ((rc = upload(e->prop + (size_t)local_rung * e->prop_stride, packed.data(), packed.size(), e->stream))) return rc;
  if (DP == 32) {
    std::vector<double> tiles(16 * 64, 0.0);
    for (int t = 0; t < 16; ++t) {
      const int rt = t & 1, sl = (t >> 1) & 3, hb = t >> 3;
      for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 16; ++i) {
          const int row = 16 * rt + i, col = 16 * hb + 4 * k + sl;
          if (row < D && col < D)
            tiles[(size_t)t * 64 + 16 * k + i] = kind == PTM_PROP_DIAG ? (row == col ? factor[row] : 0.0) : factor[(size_t)row * D + col];
        }
    }
    if ((rc = upload(e->prop_tiles + (size_t)local_rung * 16 * 64, tiles.data(), tiles.size(), e->stream))) return rc;
  }
  if (DP == 64 || DP == 128) {
    const int NT = DP / 16, ntile = NT * NT * 4;
    std::vector<double> tiles((size_t)ntile * 64, 0.0);
    for (int t = 0; t < ntile; ++t) {
      const int rt = t % NT, sl = (t / NT) & 3, hb = t / (4 * NT);
      for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 16; ++i) {
          const int row = 16 * rt + i, col = 16 * hb + 4 * k + sl;
          if (row < D && col < D)
            tiles[(size_t)t * 64 + 16 * k + i] = kind == PTM_PROP_DIAG ? (row == col ? factor[row] : 0.0) : factor[(size_t)row * D + col];
        }
    }
    if ((rc = upload(e->prop_tiles + (size_t)local_rung * ntile * 64, tiles.data(), tiles.size(), e->stream))) return rc;
  }
  if (one_d_frac >= 0) {
    if (one_d_frac > 1) return fail(PTM_ERR_INVALID, "oneDfrac must be in [0,1]");
    if ((rc = upload(e->onedfrac + local_rung, &one_d_frac, 1, e->stream))) return rc;
    if (one_d_frac > 0) e->any_oned = 1;
  }
  return PTM_OK;
}

// ---- kernel argument block ----------------------------------------------------------------------------------
static Dev make_dev(ptm_engine* e) {
  Dev p;
  memset(&p, 0, sizeof p);
  p.D = e->D; p.DP = e->DP; p.Nt = e->Nt; p.r0 = e->r0; p.nloc = e->nloc; p.W = e->W; p.Nc = e->Nc; p.w_off = e->cfg.walker_begin;
  p.seed = e->cfg.seed; p.step = e->step; p.add_every_n = e->cfg.add_every_n; p.min_prior = e->cfg.min_prior;
  p.has_bounds = e->has_bounds; p.origin_valid = e->origin_valid; p.bounds_box = e->bounds_box;
  p.blo = e->blo; p.bhi = e->bhi; p.bmin = e->bmin; p.bmax = e->bmax;
  p.all_uniform = e->all_uniform; p.lprior_const = e->lprior_const;
  p.ptype = e->ptype; p.plo = e->plo; p.phi = e->phi; p.pcoef = e->pcoef;
  p.P2 = e->P2; p.mean = e->mean; p.has_mean = e->has_mean; p.like0 = e->like0;
  p.beta = e->beta; p.prop = e->prop; p.prop_tiles = e->prop_tiles; p.P2_tiles = e->P2_tiles; p.box_row = e->box_row; p.onedfrac = e->onedfrac; p.prop_stride = e->prop_stride; p.any_oned = e->any_oned;
  p.mix_K = e->mix_K; p.mix = e->mix;
  p.de_on = e->de_on ? 1 : 0; p.de_init_extra = e->de_init_extra; p.de_init = e->de_init; p.de_hast = e->de_hast; p.de_type = e->de_type;
  p.de_snooker = e->de.snooker; p.de_gamma_one = e->de.gamma_one_frac; p.de_gamma_div = e->de.reduce_gamma; p.de_ignore = e->de.ignore_frac;
  p.de_gamma_std = e->de_on ? 1.68 / std::sqrt((double)e->D) / e->de.reduce_gamma : 0.0;   // (the reference's own expression, proposal_distribution.cc:492)
  p.betaC = e->betaC; p.beta_add = e->beta_add; p.beta_w = e->beta_w;
  p.x = e->x; p.ll = e->ll; p.lp = e->lp;
  p.ntries = e->ntries; p.naccept = e->naccept; p.last_type = e->last_type; p.nhist = e->nhist;
  p.touch = e->touch; p.err = e->err;
  p.hist = e->hist;
  p.map = e->map;
  p.c_begin = 0; p.c_end = e->Nc;
  p.host_prop = e->pcb ? 1 : 0; p.hastings = e->hastings; p.htype = e->htype; p.hvalid = e->hvalid; p.acc_out = e->acc_out;
  return p;
}

static SweepSel sweep_sel(const ptm_engine* e) {
  SweepSel s;
  s.kind = e->prop_kind == PTM_PROP_DIAG ? KIND_DIAG : (e->prop_kind == PTM_PROP_LOWER ? KIND_LOWER : KIND_DENSE);
  s.uni = (e->W % 64) == 0;
  s.host_prop = e->pcb != nullptr;
  if (s.host_prop) s.kind = KIND_DIAG;   // (no factor is read: any instantiation serves)
  s.plain = !e->has_bounds && e->all_uniform && !e->has_mean && !e->any_oned && !e->cb && e->mix_K == 0 && !e->betaC && !s.host_prop;
  s.simple = s.uni && s.plain;
  s.lean_ev = s.uni && e->betaC && !e->has_bounds && e->all_uniform && !e->has_mean && !e->any_oned && !e->cb && e->mix_K == 0 && !s.host_prop &&
              !e->hist.rungs && !e->map.rungs;
  s.callback = e->cb != nullptr;
  s.de = e->de_on;
  return s;
}

static bool lean_ev_sweeps(const ptm_engine* e) {
  static const bool forced = [] { const char* v = getenv("PTM_FORCE_VALU"); return v && *v && *v != '0'; }();
  return e->DP == 32 && !forced && sweep_sel(e).lean_ev;
}

// the compacted sweep counts a step's add_state calls for all chains at once: bring nhist up to date before anything reads it
static int flush_nhist(ptm_engine* e) {
  if (!e->nhist_pending) return PTM_OK;
  hipLaunchKernelGGL(nhist_flush_kernel, dim3((unsigned)(((size_t)e->Nc + 255) / 256)), dim3(256), 0, e->stream, e->nhist, (size_t)e->Nc, e->nhist_pending);
  HIPCHK(hipGetLastError());
  e->nhist_pending = 0;
  return PTM_OK;
}

// one fused MH sweep over local rungs [rung0, rung0 + nr); `last` closes the step (the step count is the RNG position)
// host-callback likelihood, between the propose pass and the accept pass: the gated proposals to the host, the user's function on
// them (the prior's first if it is a callback too), the new llikes back
static int ensure_betaC(ptm_engine* e);
static int callback_host_part(ptm_engine* e) {
  const size_t Nc = e->Nc, DP = e->DP;
  HIPCHK(copy_unless_shared(e->h_xprop.data(), e->xprop, Nc * DP * 8 + Nc, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  if (e->prior_cb) {
    // host-evaluated prior: the valid proposals' log-priors, then the reference's prior gate on them -- want_like (chain.cc:980)
    // with oldlprior = current_lpost - invtemp * current_llike (:973), rounded as the kernels round it
    std::vector<double> hll(Nc), hlp(Nc), hbeta(Nc), out;
    HIPCHK(hipMemcpy(hll.data(), e->ll, Nc * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hlp.data(), e->lp, Nc * 8, hipMemcpyDeviceToHost));
    { int rc2 = ensure_betaC(e); if (rc2) return rc2; }
    if (e->betaC) HIPCHK(hipMemcpy(hbeta.data(), e->betaC, Nc * 8, hipMemcpyDeviceToHost));
    std::vector<size_t> vp;
    for (size_t c = 0; c < Nc; ++c)
      if (e->h_gate[c] & 1) vp.push_back(c);
    int rcp = call_user_prior(e, e->h_xprop, vp, out);
    if (rcp) return rcp;
    std::vector<double> lpn(Nc, -__builtin_inf());
    for (size_t k = 0; k < vp.size(); ++k) {
      const size_t c = vp[k];
      const double beta = e->betaC ? hbeta[c] : e->h_beta[e->r0 + c / e->W];
      const double bl = beta * hll[c];
      const double cur_lpost = hlp[c] + bl;
      const double oldlprior = cur_lpost - bl;
      const double nl = out[k];
      lpn[c] = nl;
      const bool want = nl > -1e200 || nl - oldlprior > e->cfg.min_prior;
      e->h_gate[c] = (unsigned char)(1 | (want ? 2 : 0));
    }
    HIPCHK(hipMemcpyAsync(e->lprior_new, lpn.data(), Nc * 8, hipMemcpyHostToDevice, e->stream));
    HIPCHK(copy_unless_shared(e->gate, e->h_gate, Nc, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));   // (lpn is a local)
  }
  std::vector<size_t> pick;
  for (size_t c = 0; c < Nc; ++c)
    if (e->h_gate[c] & 2) pick.push_back(c);
  int rc = call_user(e, e->h_xprop, pick, e->h_llbatch);
  if (rc) return rc;
  for (size_t k = 0; k < pick.size(); ++k) e->h_llnew[pick[k]] = e->h_llbatch[k];
  HIPCHK(copy_unless_shared(e->llike_new, e->h_llnew.data(), Nc * 8, hipMemcpyHostToDevice, e->stream));
  return PTM_OK;
}

static int launch_sweep(ptm_engine* e, int rung0 = 0, int nr = -1, bool last = true) {
  Dev p = make_dev(e);
  if (nr < 0) nr = e->nloc - rung0;
  if (rung0 < 0 || nr < 0 || rung0 + nr > e->nloc) return fail(PTM_ERR_INVALID, "rung range out of the shard");
  p.c_begin = rung0 * e->W; p.c_end = (rung0 + nr) * e->W;
  if ((e->cb || e->pcb) && (rung0 != 0 || nr != e->nloc)) return fail(PTM_ERR_UNSUPPORTED, "partial sweeps with a host-callback likelihood or host-side proposals are not built");
  auto close_step = [&]() {
    e->step += 1; e->touched = false;
    if (e->compact_step) { e->nhist_pending += 1; e->compact_step = false; }
  };
  if (nr == 0) { if (last) close_step(); return PTM_OK; }
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (e->cfg.time_kernels) {
    if (e->kev_used + 2 > e->kev.size()) {
      for (int k = 0; k < 2; ++k) { hipEvent_t ev; HIPCHK(hipEventCreate(&ev)); e->kev.push_back(ev); }
    }
    ev0 = e->kev[e->kev_used]; ev1 = e->kev[e->kev_used + 1];
    e->kev_used += 2;
  }
  const SweepSel sel = sweep_sel(e);
  if (e->betaC_stale && !lean_ev_sweeps(e)) { int rc = ensure_betaC(e); if (rc) return rc; }   // (a build that reads the chain-indexed temperatures)
  // Compacted sweep: after an exchange phase ~1/6 of a long ladder's chains make no move; the lean MFMA build on a big
  // population then visits the moving chains only (partition_kernel packs them per rung).  PTM_COMPACT=0 switches it off.
  static const bool compact_ok = [] { const char* v = getenv("PTM_COMPACT"); return !(v && *v == '0'); }();
  // ... and the box-bounds build (uniform priors, open / limit bounds, a mean, one-dimensional moves, scale mixtures, evolving ladders)
  const bool gen1 = sel.uni && !sel.callback && !sel.host_prop && !sel.de && e->all_uniform && (!e->has_bounds || e->bounds_box);
  const bool compact = compact_ok && e->touched && e->DP == 32 && (sel.simple || gen1) && !e->hist.rungs && !e->map.rungs && !getenv("PTM_FORCE_VALU") &&
                       e->W >= 1024 && e->nloc <= 4096;   // (the same answer for every partial sweep of a step)
  if (!compact) { int rc = flush_nhist(e); if (rc) return rc; }
  if (compact) {
    if (!e->cidx) { int rc; if ((rc = dalloc(&e->cidx, (size_t)e->Nc)) || (rc = dalloc(&e->ccnt, (size_t)e->nloc))) return rc; }
    HIPCHK(hipMemsetAsync(e->ccnt + rung0, 0, (size_t)nr * sizeof(int), e->stream));
    const int nchunk = (e->W + PART_CHUNK - 1) / PART_CHUNK;
    hipLaunchKernelGGL(partition_kernel, dim3((unsigned)((size_t)nr * nchunk)), dim3(1024), (size_t)PART_CHUNK * sizeof(int), e->stream, e->W, rung0, nchunk, e->touch, e->nhist,
                       e->cidx, e->ccnt);
    e->compact_step = true;   // (every partial sweep of this step goes the same way: the step is counted once, at its end)
    HIPCHK(hipGetLastError());
    p.cidx = e->cidx; p.ccnt = e->ccnt;
  }
  // the timed bracket holds the sweep kernel alone (its name: ptm_sweep_kernel_name): the list fill and partition_kernel of a
  // compacted sweep stay outside, so that the events' mean is what rocprofv3 reports for that kernel
  if (ev0) HIPCHK(hipEventRecord(ev0, e->stream));
  auto launch = [&](const Dev& q) -> hipError_t {
    switch (e->DP) {
      case 4: return launch_sweep_4(q, sel, e->stream);
      case 8: return launch_sweep_8(q, sel, e->stream);
      case 16: return launch_sweep_16(q, sel, e->stream);
      case 32: return launch_sweep_32(q, sel, e->stream);
      case 64: return launch_sweep_64(q, sel, e->stream);
      case 128: return launch_sweep_128(q, sel, e->stream);
      case 256: return launch_sweep_256(q, sel, e->stream);
      case 512: return launch_sweep_512(q, sel, e->stream);
      case 1024: return launch_sweep_1024(q, sel, e->stream);
    }
    return hipErrorInvalidValue;
  };

  size_t npick = 0;
  if (e->pcb) {
    // host-side proposals: fetch the rows and the exchange phase's touch flags, let the host propose for every chain that
    // moves this step, hand the proposals (whole states), their log-Hastings ratios, types and validity to the kernel
    const size_t Nc = e->Nc, D = e->D, DP = e->DP;
    HIPCHK(hipMemcpyAsync(e->h_rows.data(), e->x, Nc * DP * 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(e->h_touch.data(), e->touch, Nc, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->p_pick.clear();
    for (size_t c = 0; c < Nc; ++c)
      if (!e->h_touch[c]) e->p_pick.push_back(c);
    npick = e->p_pick.size();
    e->p_xcur.resize(npick * D); e->p_xprop.assign(npick * D, 0.0); e->p_hast.assign(npick, 0.0);
    e->p_rung.resize(npick); e->p_walker.resize(npick); e->p_type.assign(npick, 0); e->p_valid.assign(npick, 1);
    for (size_t k = 0; k < npick; ++k) {
      const size_t c = e->p_pick[k];
      for (size_t d = 0; d < D; ++d) e->p_xcur[k * D + d] = e->h_rows[c * DP + host_row_pos(DP, d)];
      e->p_rung[k] = e->r0 + (int)(c / e->W); e->p_walker[k] = e->cfg.walker_begin + (int)(c % e->W);   // GLOBAL rung and walker (ptm_engine.h)
    }
    if (npick)
      e->pcb(e->pcb_user, (int)npick, (int)D, e->p_xcur.data(), e->p_rung.data(), e->p_walker.data(), e->step, e->p_xprop.data(),
             e->p_hast.data(), e->p_type.data(), e->p_valid.data());
    // (rows of chains that make no move keep whatever the buffer held: the kernel never uses them)
    for (size_t k = 0; k < npick; ++k) {
      const size_t c = e->p_pick[k];
      for (size_t d = 0; d < DP; ++d) e->h_xprop[c * DP + d] = 0.0;
      for (size_t d = 0; d < D; ++d) e->h_xprop[c * DP + host_row_pos(DP, d)] = e->p_xprop[k * D + d];
      e->h_hast[c] = e->p_hast[k]; e->h_type[c] = e->p_type[k]; e->h_valid[c] = e->p_valid[k] ? 1 : 0;
    }
    HIPCHK(copy_unless_shared(e->xprop, e->h_xprop.data(), Nc * DP * 8, hipMemcpyHostToDevice, e->stream));
    HIPCHK(copy_unless_shared(e->hastings, e->h_hast.data(), Nc * 8, hipMemcpyHostToDevice, e->stream));
    HIPCHK(copy_unless_shared(e->htype, e->h_type.data(), Nc * 4, hipMemcpyHostToDevice, e->stream));
    HIPCHK(copy_unless_shared(e->hvalid, e->h_valid.data(), Nc, hipMemcpyHostToDevice, e->stream));
    p.xprop = e->xprop; p.lprior_new = e->lprior_new; p.gate = e->gate; p.llike_new = e->llike_new;
  }
  if (!e->cb) {
    HIPCHK(launch(p));
  } else {
    // host-callback likelihood: propose kernel -> user function on the gated proposals -> accept kernel
    const size_t Nc = e->Nc, DP = e->DP;
    p.xprop = e->xprop; p.lprior_new = e->lprior_new; p.gate = e->gate; p.llike_new = e->llike_new;
    p.mode = 1;
    HIPCHK(launch(p));   // (the propose pass writes every chain's gate byte: 0 for the rungs that make no move)
    { const int rc = callback_host_part(e); if (rc) return rc; }
    p.mode = 2;
    HIPCHK(launch(p));
  }
  if (e->pcb && e->pres && npick) {   // proposal_distribution::accept() / reject() (chain.cc:1009,1015)
    HIPCHK(copy_unless_shared(e->h_acc.data(), e->acc_out, (size_t)e->Nc, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->p_acc.resize(npick);
    for (size_t k = 0; k < npick; ++k) e->p_acc[k] = e->h_acc[e->p_pick[k]] == 1 ? 1 : 0;
    e->pres(e->pcb_user, (int)npick, e->p_rung.data(), e->p_walker.data(), e->p_acc.data());
  }
  if (ev1) HIPCHK(hipEventRecord(ev1, e->stream));
  if (last) close_step();
  return PTM_OK;
}

// adds the logged, not yet counted steps to the swap counters
static int fold_swap_log(ptm_engine* e) {
  if (!e->log_pending) return PTM_OK;
  const int first = ((e->log_head - e->log_pending) % PTM_LOG_RING + PTM_LOG_RING) % PTM_LOG_RING;
  if (e->Nt > 1) {
    hipLaunchKernelGGL(fold_swap_log_kernel, dim3(e->W), dim3(256), (size_t)2 * (e->Nt - 1) * sizeof(int), e->stream, e->swap_log, e->swap_cnt,
                       e->W, e->ms, e->Nt, e->r0, e->r0 + e->nloc, first, e->log_pending);
    HIPCHK(hipGetLastError());
  }
  e->log_pending = 0;
  return PTM_OK;
}

// chain-indexed image of the evolving ladders' temperatures, for the sweep kernels
static int launch_beta_transpose(ptm_engine* e) {
  e->betaC_stale = false;
  if (e->nloc != e->Nt) {   // a rung shard: its own rungs' temperatures out of the whole ladders'
    hipLaunchKernelGGL(beta_local_kernel, dim3((unsigned)(((size_t)e->Nc + 255) / 256)), dim3(256), 0, e->stream, e->beta_w, e->betaC, e->W, e->Nt, e->r0, e->nloc);
    HIPCHK(hipGetLastError());
    return PTM_OK;
  }
  hipLaunchKernelGGL(beta_transpose_kernel, dim3((e->Nt + 31) / 32, (e->W + 31) / 32), dim3(256), 0, e->stream, e->beta_w, e->betaC, e->W, e->Nt);
  HIPCHK(hipGetLastError());
  return PTM_OK;
}
// the chain-indexed image, for everybody but the lean MFMA build of evolving ladders (which reads the ladder-major one)
static int ensure_betaC(ptm_engine* e) {
  if (!e->betaC_stale) return PTM_OK;
  e->betaC_stale = false;
  return launch_beta_transpose(e);
}

static Decide make_decide(ptm_engine* e, const double* ll_below, const double* ll_above, int H, double* send_up, double* send_down,
                          const double* ll_all = nullptr, const double* lp_all = nullptr) {
  Decide p;
  memset(&p, 0, sizeof p);
  p.ll_all = ll_all; p.lp_all = lp_all;
  p.shard_ends = e->shard_ends; p.nshards = e->nshards; p.halo_nominal = e->halo_nominal;
  p.redo_flag = e->redo_flag; p.redo_count = e->redo_flag ? e->redo_flag + e->W : nullptr;
  p.DP = e->DP; p.Nt = e->Nt; p.r0 = e->r0; p.nloc = e->nloc; p.W = e->W; p.Nc = e->Nc; p.ms = e->ms; p.w_off = e->cfg.walker_begin;
  p.seed = e->cfg.seed; p.step = e->step; p.thresh = e->thresh;
  p.beta = e->beta; p.ll_below = ll_below; p.ll_above = ll_above; p.H = ll_above ? H : 0; p.x = e->x; p.ll = e->ll; p.lp = e->lp;
  p.touch = e->touch; p.arr_below = e->arr_below; p.arr_above = e->arr_above;
  p.swap_log = e->swap_log + (size_t)e->log_head * e->W * e->ms; p.send_up = send_up; p.send_down = send_down; p.row_cap = e->row_cap; p.err = e->err;
  p.mv_src = e->mv_src; p.mv_dst = e->mv_dst; p.mv_n = e->mv_n;
  p.hist = e->hist; p.add_every_n = e->cfg.add_every_n; p.nhist = e->nhist;
  p.naccept = e->naccept; p.ntries = e->ntries; p.last_type = e->last_type;
  p.map = e->map;
  p.evolve_rate = e->evolve_rate; p.evolve_cut = e->evolve_cut; p.beta_w = e->beta_w; p.beta_add = e->beta_add;
  p.lp_is_const = e->lp_is_const ? 1 : 0; p.lp_const = e->lprior_const;
  // few ladders: the exchange kernel scatters the new temperatures into their chain-indexed image itself.  For populations the
  // scattered 8-byte stores cost the kernel what the transposition launch costs (measured at 1024 rungs x 16384 ladders: 0.621 ms
  // against 0.564 + 0.049; PTM_BETA_DIRECT=1 selects it anyway)
  static const bool direct_all = [] { const char* v = getenv("PTM_BETA_DIRECT"); return v && *v == '1'; }();
  const bool beta_direct = e->evolve_rate > 0 && (e->W <= 64 || direct_all);
  p.betaC_direct = beta_direct ? e->betaC : nullptr;
  return p;
}

static int launch_decide(ptm_engine* e, const double* ll_below, const double* ll_above, int H, double* send_up, double* send_down,
                         const double* ll_all = nullptr, const double* lp_all = nullptr, bool redo = false) {
  Decide p = make_decide(e, ll_below, ll_above, H, send_up, send_down, ll_all, lp_all);
  p.redo_only = redo ? 1 : 0;
  if (!redo && e->log_pending >= PTM_LOG_RING) { int rc = fold_swap_log(e); if (rc) return rc; }
  if (redo) {   // the second pass of the SAME step: the log slot, the messages and the move lists are the first pass's
    const int slot = (e->log_head + PTM_LOG_RING - 1) % PTM_LOG_RING;
    p.swap_log = e->swap_log + (size_t)slot * e->W * e->ms;
  }
  const bool beta_direct = p.betaC_direct != nullptr;
  e->touched = true;
  const int WN = ll_all ? e->Nt : e->nloc + (ll_below ? 1 : 0) + p.H;
  const bool evb = e->evolve_rate > 0 && e->beta_add;
  const bool cut = e->evolve_rate > 0 && e->evolve_cut >= 0;
  const size_t lds = decide_lds_bytes(e->Nt, e->ms, WN, e->evolve_rate > 0, evb, cut);
  if (lds > 160 * 1024) return fail(PTM_ERR_UNSUPPORTED, "ladder too long for the LDS-resident exchange kernel (%zu B)", lds);
  if (e->ms >= 65535) return fail(PTM_ERR_UNSUPPORTED, "more than 65534 exchange candidates per step (the exchange kernel numbers them in 16 bits)");
  // expected candidates inside the window; evolving ladders with history / MAP tracking: the wide form alone carries the
  // saved rows' temperatures through its own row moves
  // ... and a few ladders only (latency regime) whose moves may overflow the 64-thread block: one launch instead of two
  const int per_pick = (e->nloc == e->Nt ? 2 : 4) + (e->hist.rungs ? 1 : 0) + (e->map.rungs ? 1 : 0);
  const bool wide = (double)e->ms * WN / e->Nt > 96.0 || evb || cut || (e->W <= 256 && per_pick * e->ms > 64);
  if (lds > 64 * 1024) {
    HIPCHK(hipFuncSetAttribute((const void*)decide_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPCHK(hipFuncSetAttribute((const void*)decide_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  // the boundary messages start empty (the row count lives in their first word)
  if (send_up && !redo) HIPCHK(hipMemsetAsync(send_up, 0, 16, e->stream));
  if (send_down && !redo) HIPCHK(hipMemsetAsync(send_down, 0, 16, e->stream));
  if (cut) {   // (always the wide form)
    if (lds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void*)decide_kernel<256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((decide_kernel<256, true>), dim3(e->W), dim3(256), lds, e->stream, p);
  } else if (wide) hipLaunchKernelGGL(decide_kernel<256>, dim3(e->W), dim3(256), lds, e->stream, p);
  else hipLaunchKernelGGL(decide_kernel<64>, dim3(e->W), dim3(64), lds, e->stream, p);
  HIPCHK(hipGetLastError());
  if (!redo) {   // (the ring is folded when the NEXT step finds it full: a second pass of this step still finds its slot pending)
    e->log_head = (e->log_head + 1) % PTM_LOG_RING;
    ++e->log_pending;
  }
  if (e->evolve_rate > 0 && !beta_direct) {
    if (lean_ev_sweeps(e)) e->betaC_stale = true;   // (nobody reads the chain-indexed image in such a step)
    else { e->betaC_stale = false; int rc = launch_beta_transpose(e); if (rc) return rc; }
  }
  if (wide) return PTM_OK;   // the 256-thread decide kernel has applied the moves itself
  // a pick lists at most four row moves (two rungs, each a local move and / or a departure) and one in-between row each for
  // the history and the MAP: short ladders
  // can never overflow the 64-thread block's own moves
  if (per_pick * e->ms <= 64) return PTM_OK;
  Move m;
  m.DP = e->DP; m.W = e->W; m.row_cap = e->row_cap; m.x = e->x; m.ll = e->ll; m.lp = e->lp; m.send_up = send_up; m.send_down = send_down;
  m.mv_src = e->mv_src; m.mv_dst = e->mv_dst; m.mv_n = e->mv_n; m.err = e->err;
  m.hist = e->hist; m.add_every_n = e->cfg.add_every_n; m.nhist = e->nhist;
  m.naccept = e->naccept; m.ntries = e->ntries; m.last_type = e->last_type;
  m.map = e->map; m.beta = e->beta; m.r0 = e->r0;
  // (64-thread form only) ladders whose list did not fit the decide block's registers: rare, the kernel exits at once
  // for everybody else
  const dim3 mgrid((unsigned)((e->W + 63) / 64));   // (a wave scans the list lengths of 64 ladders)
  if (e->hist.rungs || e->map.rungs) hipLaunchKernelGGL((move_kernel<MVCAP, 1, true>), mgrid, dim3(64), 0, e->stream, m);
  else hipLaunchKernelGGL((move_kernel<MVCAP, 1, false>), mgrid, dim3(64), 0, e->stream, m);
  HIPCHK(hipGetLastError());
  return PTM_OK;
}

static int launch_install(ptm_engine* e, const double* recv_below, const double* recv_above) {
  if (!recv_below && !recv_above) return PTM_OK;
  Install q;
  q.DP = e->DP; q.W = e->W; q.row_cap = e->row_cap; q.x = e->x; q.ll = e->ll; q.lp = e->lp;
  q.recv_below = recv_below; q.recv_above = recv_above; q.arr_below = e->arr_below; q.arr_above = e->arr_above; q.err = e->err;
  e->touched = true;
  hipLaunchKernelGGL(install_kernel, dim3((e->row_cap + 15) / 16, 2), dim3(256), 0, e->stream, q);
  HIPCHK(hipGetLastError());
  return PTM_OK;
}

static int ready(ptm_engine* e) {
  if (!e) return fail(PTM_ERR_INVALID, "null engine");
  if (!e->have_target) return fail(PTM_ERR_INVALID, "no target set (ptm_set_target_gaussian)");
  if (!e->have_ladder) return fail(PTM_ERR_INVALID, "no ladder set (ptm_set_ladder)");
  if (!e->have_prop) return fail(PTM_ERR_INVALID, "no proposals set (ptm_set_proposals)");
  if (!e->have_state) return fail(PTM_ERR_INVALID, "no states set (ptm_set_states / ptm_init_from_prior)");
  return PTM_OK;
}

// ---- state ------------------------------------------------------------------------------------------------------
// all-uniform prior with every state inside the box: one lprior for all chains, and the sweeps keep it so (a move out of
// the box is never accepted) -- the exchange kernel then moves no lprior
static int check_lp_const(ptm_engine* e) {
  e->lp_is_const = false;
  if (!e->all_uniform || e->prior_cb) return PTM_OK;
  std::vector<double> lp((size_t)e->Nc);
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpy(lp.data(), e->lp, lp.size() * 8, hipMemcpyDeviceToHost));
  for (double v : lp)
    if (!(v == e->lprior_const)) return PTM_OK;
  e->lp_is_const = true;
  return PTM_OK;
}
static int reset_counters(ptm_engine* e) {
  const size_t Nc = e->Nc;
  std::vector<int> one(Nc, 1), m1(Nc, -1);
  std::vector<unsigned int> z(Nc, 0u);
  int rc;
  if ((rc = upload(e->ntries, one.data(), Nc, e->stream)) || (rc = upload(e->naccept, one.data(), Nc, e->stream)) ||  // chain.cc:649
      (rc = upload(e->last_type, m1.data(), Nc, e->stream)) ||
      (rc = upload(e->nhist, z.data(), Nc, e->stream)))                                                        // chain.cc:871-875
    return rc;
  e->nhist_pending = 0; e->compact_step = false; e->touched = false;
  HIPCHK(hipMemsetAsync(e->touch, 0, Nc, e->stream));
  HIPCHK(hipMemsetAsync(e->arr_below, 0xFF, (size_t)e->W * 4, e->stream));
  HIPCHK(hipMemsetAsync(e->arr_above, 0xFF, (size_t)e->W * 4, e->stream));
  e->step = 0;
  if (e->map.rungs) {    // MAP = the initial state (MH_chain::initialize -> add_state)
    if (!e->have_ladder) return fail(PTM_ERR_INVALID, "set the ladder before the states when tracking the MAP (ptm_set_ladder)");
    hipLaunchKernelGGL(map_init_kernel, dim3((e->map.MC + 255) / 256), dim3(256), 0, e->stream, e->map, e->DP, e->W, e->r0, e->beta, e->x, e->ll,
                       e->lp);
    HIPCHK(hipGetLastError());
  }
  if (e->hist.rungs) {   // history row 0 = the initial state (chain.cc:871-875)
    HIPCHK(hipMemsetAsync(e->hist.meta, 0xFF, (size_t)e->hist.cap * e->hist.HC * sizeof(int4), e->stream));
    if (e->hist.beta && !e->have_ladder) return fail(PTM_ERR_INVALID, "set the ladder before the states (ptm_set_ladder)");
    hipLaunchKernelGGL(hist_init_kernel, dim3((e->hist.HC + 255) / 256), dim3(256), 0, e->stream, e->hist, e->DP, e->x, e->ll, e->lp,
                       e->naccept, e->ntries, e->last_type, e->beta, e->betaC, e->W, e->r0);
    HIPCHK(hipGetLastError());
  }
  return PTM_OK;
}

static int run_eval(ptm_engine* e, int n, double* x, int* valid, double* lp, double* ll, int eval_like) {
  Dev p = make_dev(e);
  switch (e->DP) {
    case 4: HIPCHK(launch_eval_4(p, n, x, valid, lp, ll, eval_like, e->stream)); break;
    case 8: HIPCHK(launch_eval_8(p, n, x, valid, lp, ll, eval_like, e->stream)); break;
    case 16: HIPCHK(launch_eval_16(p, n, x, valid, lp, ll, eval_like, e->stream)); break;
    case 32: HIPCHK(launch_eval_32(p, n, x, valid, lp, ll, eval_like, e->stream)); break;
    case 64: HIPCHK(launch_eval_64(p, n, x, valid, lp, ll, eval_like, e->stream)); break;
    case 128: HIPCHK(launch_eval_128(p, n, x, valid, lp, ll, eval_like, e->stream)); break;
    case 256: HIPCHK(launch_eval_256(p, n, x, valid, lp, ll, eval_like, e->stream)); break;
    case 512: HIPCHK(launch_eval_512(p, n, x, valid, lp, ll, eval_like, e->stream)); break;
    case 1024: HIPCHK(launch_eval_1024(p, n, x, valid, lp, ll, eval_like, e->stream)); break;
    default: return fail(PTM_ERR_UNSUPPORTED, "dim > 1024 is not built");
  }
  return PTM_OK;
}

// host rows [n][D] -> device row image [n][DP]: padded, and for DP == 32 / 64 / 128 with dimension d at position row_pos(d)
// (ptm_kernels.hpp: the MFMA accumulator layout)
static inline size_t host_row_pos(size_t DP, size_t d) { return (DP == 32 || DP == 64 || DP == 128) ? 8 * ((d >> 2) >> 1) + 2 * (d & 3) + ((d >> 2) & 1) : d; }
static std::vector<double> pad_rows(const double* X, size_t n, size_t D, size_t DP) {
  std::vector<double> r(n * DP, 0.0);
  for (size_t c = 0; c < n; ++c)
    for (size_t d = 0; d < D; ++d) r[c * DP + host_row_pos(DP, d)] = X[c * D + d];
  return r;
}
static void unpad_rows(const double* r, size_t n, size_t D, size_t DP, double* X) {
  for (size_t c = 0; c < n; ++c)
    for (size_t d = 0; d < D; ++d) X[c * D + d] = r[c * DP + host_row_pos(DP, d)];
}
static void unpad_rows(const std::vector<double>& r, size_t n, size_t D, size_t DP, double* X) { unpad_rows(r.data(), n, D, DP, X); }

extern "C" int ptm_set_states(ptm_engine* e, const double* X, const double* llike) {
  SETTLE(e);
  NO_BATCH(e, "ptm_set_states");
  if (!e || !X) return fail(PTM_ERR_INVALID, "null argument");
  if (!e->have_target && !llike) return fail(PTM_ERR_INVALID, "set the target first (or pass llike)");
  const size_t Nc = e->Nc, D = e->D, DP = e->DP;
  const std::vector<double> rows = pad_rows(X, Nc, D, DP);
  int rc;
  if ((rc = upload(e->x, rows.data(), Nc * DP, e->stream))) return rc;
  if (llike && (rc = upload(e->ll, llike, Nc, e->stream))) return rc;
  if ((rc = run_eval(e, (int)Nc, e->x, nullptr, e->lp, e->ll, (llike || e->cb) ? 0 : 1))) return rc;
  if (e->cb && !llike) {
    // MH_chain::add_state(s) with the 999 sentinel: the likelihood plug-in evaluates the (enforced) start states (chain.cc:925)
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(e->h_xprop.data(), e->x, Nc * DP * 8, hipMemcpyDeviceToHost));
    std::vector<size_t> all(Nc);
    for (size_t c = 0; c < Nc; ++c) all[c] = c;
    if ((rc = call_user(e, e->h_xprop, all, e->h_llbatch))) return rc;
    if ((rc = upload(e->ll, e->h_llbatch.data(), Nc, e->stream))) return rc;
  }
  if (e->prior_cb) {
    if (!e->cb) return fail(PTM_ERR_UNSUPPORTED, "a host-evaluated prior needs the host-callback likelihood (ptm_set_target_callback)");
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(e->h_xprop.data(), e->x, Nc * DP * 8, hipMemcpyDeviceToHost));
    if ((rc = host_prior_of_states(e))) return rc;
  }
  if ((rc = reset_counters(e))) return rc;
  HIPCHK(hipStreamSynchronize(e->stream));
  if ((rc = check_lp_const(e))) return rc;
  e->have_state = 1;
  return PTM_OK;
}

static int launch_init(ptm_engine* e, const Dev& p, long long attempt, unsigned char* pending, double* x = nullptr, double* ll = nullptr, double* lp = nullptr) {
  if (x) {   // (into a caller's staging arrays: ptm_draw_prior_rows)
    switch (e->DP) {
      case 4: HIPCHK(launch_init_4(p, x, ll, lp, e->err + 1, attempt, pending, e->stream)); break;
      case 8: HIPCHK(launch_init_8(p, x, ll, lp, e->err + 1, attempt, pending, e->stream)); break;
      case 16: HIPCHK(launch_init_16(p, x, ll, lp, e->err + 1, attempt, pending, e->stream)); break;
      case 32: HIPCHK(launch_init_32(p, x, ll, lp, e->err + 1, attempt, pending, e->stream)); break;
      case 64: HIPCHK(launch_init_64(p, x, ll, lp, e->err + 1, attempt, pending, e->stream)); break;
      case 128: HIPCHK(launch_init_128(p, x, ll, lp, e->err + 1, attempt, pending, e->stream)); break;
      case 256: HIPCHK(launch_init_256(p, x, ll, lp, e->err + 1, attempt, pending, e->stream)); break;
      case 512: HIPCHK(launch_init_512(p, x, ll, lp, e->err + 1, attempt, pending, e->stream)); break;
      case 1024: HIPCHK(launch_init_1024(p, x, ll, lp, e->err + 1, attempt, pending, e->stream)); break;
      default: return fail(PTM_ERR_UNSUPPORTED, "dim > 1024 is not built");
    }
    return PTM_OK;
  }
  switch (e->DP) {
    case 4: HIPCHK(launch_init_4(p, e->x, e->ll, e->lp, e->err + 1, attempt, pending, e->stream)); break;
    case 8: HIPCHK(launch_init_8(p, e->x, e->ll, e->lp, e->err + 1, attempt, pending, e->stream)); break;
    case 16: HIPCHK(launch_init_16(p, e->x, e->ll, e->lp, e->err + 1, attempt, pending, e->stream)); break;
    case 32: HIPCHK(launch_init_32(p, e->x, e->ll, e->lp, e->err + 1, attempt, pending, e->stream)); break;
    case 64: HIPCHK(launch_init_64(p, e->x, e->ll, e->lp, e->err + 1, attempt, pending, e->stream)); break;
    case 128: HIPCHK(launch_init_128(p, e->x, e->ll, e->lp, e->err + 1, attempt, pending, e->stream)); break;
    case 256: HIPCHK(launch_init_256(p, e->x, e->ll, e->lp, e->err + 1, attempt, pending, e->stream)); break;
    case 512: HIPCHK(launch_init_512(p, e->x, e->ll, e->lp, e->err + 1, attempt, pending, e->stream)); break;
    case 1024: HIPCHK(launch_init_1024(p, e->x, e->ll, e->lp, e->err + 1, attempt, pending, e->stream)); break;
    default: return fail(PTM_ERR_UNSUPPORTED, "dim > 1024 is not built");
  }
  return PTM_OK;
}

extern "C" int ptm_init_from_prior(ptm_engine* e) { return ptm_init_from_prior_k(e, 0); }

extern "C" int ptm_init_from_prior_k(ptm_engine* e, int kdraw) {
  SETTLE(e);
  NO_BATCH(e, "ptm_init_from_prior");
  if (!e) return fail(PTM_ERR_INVALID, "null engine");
  if (kdraw < 0 || kdraw > 8191) return fail(PTM_ERR_INVALID, "initial draw index out of range (0..8191)");
  if (!e->have_target) return fail(PTM_ERR_INVALID, "set the target first");
  if (e->prior_cb) return fail(PTM_ERR_UNSUPPORTED, "a host-evaluated prior cannot be drawn from here: draw the start states with its drawSample and pass them to ptm_set_states");
  for (int d = 0; d < e->D; ++d)
    if (e->h_ptype[d] == PTM_PRIOR_FLAT)
      return fail(PTM_ERR_UNSUPPORTED, "a flat (improper) prior cannot be drawn from (dimension %d): pass start states", d);
  Dev p = make_dev(e);
  p.init_base = (uint64_t)kdraw << 17;   // attempts of draw k count from k * 2^17 (a draw gives up after 100000 < 2^17 attempts)
  HIPCHK(hipMemsetAsync(e->err + 1, 0, 4, e->stream));
  int rc;
  if (!e->cb) {
    if ((rc = launch_init(e, p, -1, nullptr))) return rc;
  } else {
    // MH_chain::initialize's redraw loop (chain.cc:856-869) with the plug-in likelihood on the host: one attempt per
    // launch for the chains still pending; `gate` doubles as the pending flag array
    const size_t Nc = e->Nc, DP = e->DP;
    HIPCHK(hipMemsetAsync(e->gate, 0, Nc, e->stream));
    std::vector<double> llh(Nc, 0.0);
    size_t left = Nc;
    for (long long a = 0; left && a < 100000; ++a) {
      if ((rc = launch_init(e, p, a, e->gate))) return rc;
      HIPCHK(copy_unless_shared(e->h_gate, e->gate, Nc, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipMemcpyAsync(e->h_xprop.data(), e->x, Nc * DP * 8, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
      std::vector<size_t> pick;
      for (size_t c = 0; c < Nc; ++c)
        if (e->h_gate[c] == 1) pick.push_back(c);
      if ((rc = call_user(e, e->h_xprop, pick, e->h_llbatch))) return rc;
      for (size_t k = 0; k < pick.size(); ++k) {
        const double v = e->h_llbatch[k];
        if (v < -1e100) { e->h_gate[pick[k]] = 0; continue; }   // chain.cc:858: (slike=evaluate_log(s)) < -1e100 => redraw
        llh[pick[k]] = v;
        e->h_gate[pick[k]] = 2;
        left--;
      }
      HIPCHK(copy_unless_shared(e->gate, e->h_gate, Nc, hipMemcpyHostToDevice, e->stream));
    }
    if (left) return fail(PTM_ERR_INVALID, "could not draw a valid start state from the prior for some chain");
    if ((rc = upload(e->ll, llh.data(), Nc, e->stream))) return rc;
  }
  rc = reset_counters(e);
  if (rc) return rc;
  int flag = 0;
  HIPCHK(hipMemcpy(&flag, e->err + 1, 4, hipMemcpyDeviceToHost));
  if (flag) return fail(PTM_ERR_INVALID, "could not draw a valid start state from the prior for some chain");
  if ((rc = check_lp_const(e))) return rc;
  e->have_state = 1;
  return PTM_OK;
}

// Draws k_begin .. k_begin + n - 1 of every chain's initial draws (the rows ptm_init_from_prior_k(e, k) would leave in the engine) into
// host arrays, the engine's own state untouched: what MH_chain::initialize(n) saves in FRONT of the start state (chain.cc:846-876) in
// one call -- the sampler's default asks for 50 x dim of them, and one engine initialisation per row was most of a short run's time.
extern "C" int ptm_draw_prior_rows(ptm_engine* e, int k_begin, int n, double* x_out, double* ll_out, double* lp_out) {
  SETTLE(e);
  NO_BATCH(e, "ptm_draw_prior_rows");
  if (!e || (n > 0 && (!x_out || !ll_out || !lp_out))) return fail(PTM_ERR_INVALID, "null argument");
  if (n <= 0) return PTM_OK;
  if (k_begin < 0 || k_begin + n - 1 > 8191) return fail(PTM_ERR_INVALID, "initial draw index out of range (0..8191)");
  if (!e->have_target) return fail(PTM_ERR_INVALID, "set the target first");
  if (e->prior_cb) return fail(PTM_ERR_UNSUPPORTED, "a host-evaluated prior cannot be drawn from here: draw with its drawSample");
  for (int d = 0; d < e->D; ++d)
    if (e->h_ptype[d] == PTM_PRIOR_FLAT)
      return fail(PTM_ERR_UNSUPPORTED, "a flat (improper) prior cannot be drawn from (dimension %d)", d);
  const size_t Nc = e->Nc, D = e->D, DP = e->DP;
  // staging on the device: G draws at a time (at most ~256 MB of rows)
  const size_t per = Nc * DP * 8;
  size_t G = per ? (size_t)(256u << 20) / per : 1;
  if (G < 1) G = 1;
  if (G > (size_t)n) G = (size_t)n;
  if (e->cb) G = 1;
  double *xs = nullptr, *ls = nullptr, *ps = nullptr;
  int rc = PTM_OK;
  if ((rc = dalloc(&xs, G * Nc * DP)) || (rc = dalloc(&ls, G * Nc)) || (rc = dalloc(&ps, G * Nc))) { hipFree(xs); hipFree(ls); hipFree(ps); return rc; }
  std::vector<double> hx(G * Nc * DP), hl(G * Nc), hp(G * Nc);
  auto done = [&](int r) { hipFree(xs); hipFree(ls); hipFree(ps); return r; };
#define PTM_DRAW_CHK(call) do { const hipError_t e_ = (call); if (e_ != hipSuccess) return done(fail(PTM_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_))); } while (0)
  PTM_DRAW_CHK(hipMemsetAsync(e->err + 1, 0, 4, e->stream));
  for (int k0 = 0; k0 < n; k0 += (int)G) {
    const int g = std::min((int)G, n - k0);
    for (int j = 0; j < g; ++j) {
      Dev p = make_dev(e);
      p.init_base = (uint64_t)(k_begin + k0 + j) << 17;
      double *xj = xs + (size_t)j * Nc * DP, *lj = ls + (size_t)j * Nc, *pj = ps + (size_t)j * Nc;
      if (!e->cb) {
        if ((rc = launch_init(e, p, -1, nullptr, xj, lj, pj))) return done(rc);
      } else {
        // the redraw loop of ptm_init_from_prior_k (chain.cc:856-869) on the staging row
        PTM_DRAW_CHK(hipMemsetAsync(e->gate, 0, Nc, e->stream));
        std::vector<double> llh(Nc, 0.0);
        size_t left = Nc;
        for (long long a = 0; left && a < 100000; ++a) {
          if ((rc = launch_init(e, p, a, e->gate, xj, lj, pj))) return done(rc);
          PTM_DRAW_CHK(copy_unless_shared(e->h_gate, e->gate, Nc, hipMemcpyDeviceToHost, e->stream));
          PTM_DRAW_CHK(hipMemcpyAsync(e->h_xprop.data(), xj, Nc * DP * 8, hipMemcpyDeviceToHost, e->stream));
          PTM_DRAW_CHK(hipStreamSynchronize(e->stream));
          std::vector<size_t> pick;
          for (size_t c = 0; c < Nc; ++c)
            if (e->h_gate[c] == 1) pick.push_back(c);
          if ((rc = call_user(e, e->h_xprop, pick, e->h_llbatch))) return done(rc);
          for (size_t q = 0; q < pick.size(); ++q) {
            const double v = e->h_llbatch[q];
            if (v < -1e100) { e->h_gate[pick[q]] = 0; continue; }
            llh[pick[q]] = v;
            e->h_gate[pick[q]] = 2;
            left--;
          }
          PTM_DRAW_CHK(copy_unless_shared(e->gate, e->h_gate, Nc, hipMemcpyHostToDevice, e->stream));
        }
        if (left) return done(fail(PTM_ERR_INVALID, "could not draw a valid state from the prior for some chain"));
        PTM_DRAW_CHK(hipMemcpyAsync(lj, llh.data(), Nc * 8, hipMemcpyHostToDevice, e->stream));
        PTM_DRAW_CHK(hipStreamSynchronize(e->stream));   // (llh is a local)
      }
    }
    PTM_DRAW_CHK(hipMemcpyAsync(hx.data(), xs, (size_t)g * Nc * DP * 8, hipMemcpyDeviceToHost, e->stream));
    PTM_DRAW_CHK(hipMemcpyAsync(hl.data(), ls, (size_t)g * Nc * 8, hipMemcpyDeviceToHost, e->stream));
    PTM_DRAW_CHK(hipMemcpyAsync(hp.data(), ps, (size_t)g * Nc * 8, hipMemcpyDeviceToHost, e->stream));
    PTM_DRAW_CHK(hipStreamSynchronize(e->stream));
    for (int j = 0; j < g; ++j) {
      const size_t o = (size_t)(k0 + j);
      for (size_t c = 0; c < Nc; ++c) {
        for (size_t d = 0; d < D; ++d) x_out[(o * Nc + c) * D + d] = hx[((size_t)j * Nc + c) * DP + host_row_pos(DP, d)];
        ll_out[o * Nc + c] = hl[(size_t)j * Nc + c];
        lp_out[o * Nc + c] = hp[(size_t)j * Nc + c];
      }
    }
  }
  int flag = 0;
  PTM_DRAW_CHK(hipMemcpy(&flag, e->err + 1, 4, hipMemcpyDeviceToHost));
#undef PTM_DRAW_CHK
  if (flag) return done(fail(PTM_ERR_INVALID, "could not draw a valid state from the prior for some chain"));
  return done(PTM_OK);
}

// ---- hot path ----------------------------------------------------------------------------------------------------
extern "C" int ptm_sweep(ptm_engine* e, int n) {
  SETTLE(e);
  NO_BATCH(e, "ptm_sweep");
  int rc = ready(e);
  if (rc) return rc;
  for (int k = 0; k < n; ++k)
    if ((rc = launch_sweep(e))) return rc;
  return PTM_OK;
}

// Small ladders (rungs x padded dimensions <= the 1024 lanes of one workgroup, device target and proposals): whole PT steps in
// ONE launch per batch, a block per walker-ladder looping over the steps (ptm_fused_kernel.hpp).  PTM_FUSED=0 keeps the
// two-launch path.  Returns the steps taken (0: not this engine's case), or a negative status.
static int fused_steps(ptm_engine* e, int n) {
  static const bool fused_ok = [] { const char* v = getenv("PTM_FUSED"); return !(v && *v == '0'); }();
  if (!fused_ok || e->DP > 16 || (long long)e->Nt * e->DP > 256 || e->cb || e->pcb || e->cfg.time_kernels) return 0;
  if (e->evolve_rate > 0 && (e->W > 64 || e->evolve_cut >= 0)) return 0;   // (the new temperatures' chain-indexed image is then a separate launch)
  const bool evb = e->evolve_rate > 0 && e->beta_add;
  const size_t dlds = decide_lds_bytes(e->Nt, e->ms, e->Nt, e->evolve_rate > 0, evb);
  if (dlds > 96 * 1024) return 0;
  int rc = flush_nhist(e);
  if (rc) return rc;
  if (e->log_pending >= PTM_LOG_RING && (rc = fold_swap_log(e))) return rc;
  int done = 0;
  while (done < n) {
    const int k = std::min(n - done, PTM_LOG_RING - e->log_pending);   // the candidate logs of at most a ring's worth of steps
    Dev p = make_dev(e);
    const Decide d = make_decide(e, nullptr, nullptr, 0, nullptr, nullptr);
    const bool diag = e->prop_kind == PTM_PROP_DIAG;
    hipError_t he;
    switch (e->DP) {
      case 4: he = launch_fused_4(p, d, diag, k, e->swap_log, e->log_head, dlds, e->stream); break;
      case 8: he = launch_fused_8(p, d, diag, k, e->swap_log, e->log_head, dlds, e->stream); break;
      default: he = launch_fused_16(p, d, diag, k, e->swap_log, e->log_head, dlds, e->stream); break;
    }
    HIPCHK(he);
    e->step += (uint64_t)k;
    e->log_head = (e->log_head + k) % PTM_LOG_RING;
    e->log_pending += k;
    if (e->log_pending >= PTM_LOG_RING && (rc = fold_swap_log(e))) return rc;
    done += k;
  }
  e->touched = false;
  return done;
}

// Long ladders of few walkers (the reference's own shape: one ladder of 1024 rungs): the steps of a ptm_step(n) call in ONE launch
// of resident workgroups that keep the chains in registers and talk to their neighbours through flags (ptm_ladder_kernel.hpp).
// Plain workload only; PTM_LADDER=0 keeps the two-launch path.  Returns the steps taken (0: not this engine's case), or a negative
// status.
// does a ptm_step call of this engine go through the persistent ladder kernel?  (grid: its workgroups; lds: their LDS)
#ifndef PTM_LADDER_MIN_STEPS
#define PTM_LADDER_MIN_STEPS 1
#endif
// the build of the persistent ladder kernel an engine takes: bit 0 one-dimensional moves / scale mixtures, bit 1 history / MAP tracking,
// bit 2 evolving ladders (built plain, 4, and with everything, 7)
static int ladder_flavour(const ptm_engine* e) {
  const int fl = ((e->any_oned || e->mix_K > 0) ? 1 : 0) | ((e->hist.rungs || e->map.rungs) ? 2 : 0);
  // any boundary (wrap, reflect) or a prior that is not all uniform: the builds with the general state space, which carry the recipe's and
  // the history's code whether or not this engine uses them (19, 23, 27, 31)
  if ((e->has_bounds && !e->bounds_box) || !e->all_uniform) return 16 | 3 | (e->evolve_rate > 0 ? 4 : 0) | (e->de_on ? 8 : 0);
  if (e->de_on) return e->evolve_rate > 0 ? 15 : 11;   // differential evolution: a member of a set, drawn from the history ring
  return e->evolve_rate > 0 ? (fl ? 7 : 4) : fl;
}
static bool ladder_applies(ptm_engine* e, long long* grid_out = nullptr, size_t* lds_out = nullptr) {
  static const bool ladder_ok = [] { const char* v = getenv("PTM_LADDER"); return !(v && *v == '0'); }();
  if (!ladder_ok || e->lad_disabled || e->DP > 32 || e->Nt < 2 || e->nloc != e->Nt || e->cfg.time_kernels || (e->evolve_rate > 0 && e->evolve_cut >= 0) || e->shard)
    return false;
  // open / `limit` boundaries, all-uniform prior, zero mean, fixed ladder, device target and proposals (one-dimensional moves, scale
  // mixtures, history and MAP tracking have their builds: ladder_flavour); ANY population whose grid is resident at once (below): where
  // it fits, a step costs this kernel its ~6 us of latency whatever the walkers' number (64 walkers x 64 rungs of 12 dimensions with the
  // sampler's defaults: 14 us against 40 on two launches)
  if (e->cb || e->prior_cb || e->pcb) return false;
  const int R = 256 / e->DP, NB = (e->Nt + R - 1) / R;
  const long long grid = (long long)e->W * NB;
  const bool diag = e->prop_kind == PTM_PROP_DIAG;
  const int fl = ladder_flavour(e);
  const bool ev_ = e->evolve_rate > 0;
  const size_t lds = e->DP == 4 ? ladder_lds_4(e->Nt, e->ms, ev_) : e->DP == 8 ? ladder_lds_8(e->Nt, e->ms, ev_) : e->DP == 16 ? ladder_lds_16(e->Nt, e->ms, ev_) : ladder_lds_32(e->Nt, e->ms, ev_);
  if (lds > 160 * 1024) return false;
  int& cap = e->lad_capacity[diag ? 1 : 0][fl];   // (asked per build: the builds differ in registers, and the LDS attribute is per kernel)
  if (cap < 0) cap = e->DP == 4 ? ladder_blocks_4(diag, fl, lds) : e->DP == 8 ? ladder_blocks_8(diag, fl, lds) : e->DP == 16 ? ladder_blocks_16(diag, fl, lds) : ladder_blocks_32(diag, fl, lds);
  // every workgroup must be resident at once (they wait for each other)
  if (grid > cap || grid > 1024) return false;
  if (grid_out) *grid_out = grid;
  if (lds_out) *lds_out = lds;
  return true;
}
static int two_launch_steps(ptm_engine* e, int n);
// Have the launches of the persistent ladder kernel since the last look been committed?  If one gave up, the engine's arrays are as
// they were before it (the kernel commits all of a launch or nothing, and the launches behind it found its number missing and did
// nothing): its steps and theirs are repeated on the two-launch path, and this engine keeps that path from now on.
static int ladder_steps(ptm_engine* e, int n);
static int ladder_flush(ptm_engine* e) {
  if (!e->lad_deferred) return PTM_OK;
  int n = e->lad_deferred;
  e->lad_deferred = 0;
  const int f = ladder_steps(e, n);
  if (f < 0) return f;
  n -= f;
  if (n > 0) {   // (the kernel was switched off meanwhile: a launch gave up)
    HIPCHK(hipStreamSynchronize(e->stream));
    return two_launch_steps(e, n);
  }
  return PTM_OK;
}
static int ladder_settle(ptm_engine* e) {
  if (e->lad_deferred) { const int rc = ladder_flush(e); if (rc) return rc; }
  if (e->lad_log.empty()) return PTM_OK;
  HIPCHK(hipStreamSynchronize(e->stream));
  int w[3] = {0, 0, 0};
  HIPCHK(hipMemcpy(w, e->err, sizeof w, hipMemcpyDeviceToHost));
  if (w[2] == e->lad_seq) { e->lad_log.clear(); return PTM_OK; }
  size_t k = 0;
  while (k < e->lad_log.size() && e->lad_log[k].seq != w[2] + 1) ++k;
  if (k == e->lad_log.size()) return fail(PTM_ERR_HIP, "persistent ladder kernel: the device reports launch %d as the last one committed, the host has no record of its successor", w[2]);
  long long todo = 0;
  for (size_t j = k; j < e->lad_log.size(); ++j) todo += e->lad_log[j].nsteps;
  e->step = e->lad_log[k].step_before;
  e->log_head = e->lad_log[k].log_head_before;
  e->lad_log.clear();
  e->lad_disabled = true;
  e->ladder_fallbacks += 1;
  const int clr = w[0] & ~32;
  HIPCHK(hipMemcpy(e->err, &clr, 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->err + 2, &e->lad_seq, 4, hipMemcpyHostToDevice));
  static const bool quiet = [] { const char* v = getenv("PTM_QUIET"); return v && *v && *v != '0'; }();
  if (!quiet)
    fprintf(stderr, "[ptm] the persistent ladder kernel gave up waiting for a neighbouring workgroup (is the device shared?): nothing of that launch was "
                    "kept, its %lld steps are repeated on the two-launch path, which this engine keeps from now on\n", todo);
  while (todo > 0) {
    const int n = todo > (1 << 30) ? (1 << 30) : (int)todo;
    const int rc = two_launch_steps(e, n);
    if (rc) return rc;
    todo -= n;
  }
  return PTM_OK;
}
static int ladder_steps(ptm_engine* e, int n) {
  long long grid = 0;
  size_t lds = 0;
  // A launch of the persistent kernel stages its tables and loads its chains: a few microseconds on top of its steps
  // (tools/step1_probe.py).  A host loop that steps one at a time keeps the two-launch path.
  if (n < PTM_LADDER_MIN_STEPS) return 0;
  if (!ladder_applies(e, &grid, &lds)) return 0;
  const int R = 256 / e->DP, NB = (e->Nt + R - 1) / R;
  const bool diag = e->prop_kind == PTM_PROP_DIAG;
  const int fl = ladder_flavour(e);
  static const int max_run = [] { const char* v = getenv("PTM_LADDER_MAXRUN"); const int m = v && *v ? atoi(v) : LADDER_H; return m < 1 ? 1 : (m > LADDER_H ? LADDER_H : m); }();
  // how long a workgroup waits for a neighbour before it gives up: 3 s of the 100 MHz wall clock (a neighbour that is not there by
  // then never will be); PTM_LADDER_SPIN_US shortens it (tests: 0 makes every flag that is not up at the first look a reason to give up)
  static const long long spin_limit = [] { const char* v = getenv("PTM_LADDER_SPIN_US"); return v && *v ? atoll(v) * 100ll : 300000000ll; }();
  int rc = flush_nhist(e);
  if (rc) return rc;
  if (e->lad_log.size() >= 256 && (rc = ladder_settle(e))) return rc;   // (a host that never looks: look for it now and then)
  if (e->lad_disabled) return 0;
  const size_t Nc = e->Nc;
  if (!e->pub_x) {
    // what the workgroups publish for each other: fine-grained device memory where the runtime has it (a little faster across
    // XCDs: tools/probes/flag_pingpong_probe.hip), else ordinary -- the accesses are agent-scope atomics either way
    const size_t doubles = 2 * Nc * e->DP + 4 * Nc + 4 * Nc + 2;   // rows | llikes | lpriors | stamped llikes (16-byte aligned)
    void* buf = nullptr;
    if (hipExtMallocWithFlags(&buf, doubles * sizeof(double), hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
      HIPCHK(hipMalloc(&buf, doubles * sizeof(double)));
    }
    e->pub_x = (double*)buf; e->pub_ll = e->pub_x + 2 * Nc * e->DP; e->pub_lp = e->pub_ll + 2 * Nc;
    HIPCHK(hipMemsetAsync(e->pub_lp + 2 * Nc, 0, (4 * Nc + 2) * sizeof(double), e->stream));   // (no stale word may look like a stamp)
    void* fl_ = nullptr;
    const size_t flbytes = ((size_t)grid + 16 + (size_t)e->W) * sizeof(int);
    if (hipExtMallocWithFlags(&fl_, flbytes, hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
      HIPCHK(hipMalloc(&fl_, flbytes));
    }
    e->lad_flags = (int*)fl_; e->lad_ctl = e->lad_flags + grid;   // [grid] flags | [16] control words | [W] whole-ladder barrier counters
    HIPCHK(hipEventCreateWithFlags(&e->lad_event, hipEventDisableTiming));
  }
  static const bool prof_on = [] { const char* v = getenv("PTM_LADDER_PROF"); return v && *v && *v != '0'; }();
  if (prof_on && !e->lad_prof && (rc = dalloc(&e->lad_prof, (size_t)grid * 8))) return rc;
  int done = 0;
  while (done < n) {
    // (one launch walks at most 2^24 steps: its step-counted flags and LDS counters are ints, and a launch should end some day)
    const int k = n - done < (1 << 24) ? n - done : (1 << 24);
    Dev p = make_dev(e);
    LadderArgs a;
    a.nsteps = k; a.NB = NB; a.ms = e->ms; a.thresh = e->thresh;
    a.pub_x = e->pub_x; a.pub_ll = e->pub_ll; a.pub_lp = e->pub_lp; a.pub_s = e->pub_lp + 2 * (size_t)e->Nc + ((2 * (size_t)e->Nc * e->DP) & 1); a.flags = e->lad_flags; a.ctl = e->lad_ctl; a.slow_done = e->lad_ctl + 16;
    a.swap_cnt = e->swap_cnt; a.swap_log = e->swap_log + (size_t)e->log_head * e->W * e->ms;
    a.max_run = max_run;
    a.prof = e->lad_prof;
    { static const int pt = [] { const char* v = getenv("PTM_LADDER_PROF"); return (v && *v == '2') ? 256 : ((v && *v == '3') ? 384 : 0); }(); a.prof_tid = pt; }
    if ((rc = fold_swap_log(e))) return rc;   // (the kernel adds to the swap counters itself: nothing logged may be pending behind it)
    a.spin_limit = spin_limit;
    a.evolve_rate = e->evolve_rate;
    a.done_seq = e->err + 2;
    a.seq = e->lad_seq + 1;
    HIPCHK(hipMemsetAsync(e->lad_flags, 0, ((size_t)grid + 16 + (size_t)e->W) * sizeof(int), e->stream));
    {
      std::lock_guard<std::mutex> lock(g_lad_mutex);
      if (g_lad_last && g_lad_last != e && g_lad_last->lad_event) HIPCHK(hipStreamWaitEvent(e->stream, g_lad_last->lad_event, 0));
      HIPCHK(e->DP == 4 ? launch_ladder_4(p, a, diag, fl, (int)grid, lds, e->stream) : e->DP == 8 ? launch_ladder_8(p, a, diag, fl, (int)grid, lds, e->stream) :
             e->DP == 16 ? launch_ladder_16(p, a, diag, fl, (int)grid, lds, e->stream) : launch_ladder_32(p, a, diag, fl, (int)grid, lds, e->stream));
      // (the event costs a host call per launch: only a process with several engines on this path records it)
      static int engines_seen = 0;
      static ptm_engine* first_seen = nullptr;
      if (!first_seen) first_seen = e;
      if (first_seen != e) engines_seen = 2;
      if (engines_seen >= 2 || (g_lad_last && g_lad_last != e)) HIPCHK(hipEventRecord(e->lad_event, e->stream));
      g_lad_last = e;
    }
    e->ladder_launches++;
    e->lad_seq += 1;
    e->lad_log.push_back({e->lad_seq, e->step, k, e->log_head});
    if (e->lad_prof) {
      // diagnostics (PTM_LADDER_PROF): wait for the launch, read its outcome and its phase clocks -- mean microseconds per step and
      // phase over the workgroups, and the slowest workgroup's
      int ctl[4] = {0, 0, 0, 0};
      HIPCHK(hipMemcpyAsync(ctl, e->lad_ctl, sizeof ctl, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
      if (ctl[1] > 0) {
        std::vector<long long> pr((size_t)grid * 8);
        HIPCHK(hipMemcpy(pr.data(), e->lad_prof, pr.size() * 8, hipMemcpyDeviceToHost));
        static const char* const phase[7] = {"", "publish", "draws | random blocks", "filter | offsets", "window | Metropolis", "commit + trials (evolving: decisions + pries)", "rows (evolving: temperatures, Metropolis, rows)"};
        fprintf(stderr, "[ladder kernel] %d steps, %lld workgroups; us per step (mean / max over workgroups):", ctl[1], grid);
        for (int q = 1; q < 7; ++q) {
          double sum = 0, mx = 0;
          for (long long g2 = 0; g2 < grid; ++g2) { const double v = pr[(size_t)g2 * 8 + q] * 0.01 / ctl[1]; sum += v; if (v > mx) mx = v; }
          fprintf(stderr, " %s %.2f/%.2f", phase[q], sum / grid, mx);
        }
        fprintf(stderr, "\n");
        e->ladder_whole_steps += ctl[2];
      }
    }
    e->step += (uint64_t)k;
    done += k;
    if (e->evolve_rate > 0) e->betaC_stale = true;    // (the kernel keeps the ladder-major temperatures; the chain-indexed image on demand: ensure_betaC)
    e->log_head = (e->log_head + 1) % PTM_LOG_RING;   // (the last step's candidate log sits in the slot handed over)
  }
  e->touched = false;
  return done;
}

static int two_launch_steps(ptm_engine* e, int n) {
  int rc;
  for (int k = 0; k < n; ++k) {
    if (e->Nt > 1 && (rc = launch_decide(e, nullptr, nullptr, 0, nullptr, nullptr))) return rc;
    if ((rc = launch_sweep(e))) return rc;
  }
  return PTM_OK;
}

extern "C" int ptm_step(ptm_engine* e, int n) {
  int rc = ready(e);
  if (rc) return rc;
  if (e->nloc != e->Nt) return fail(PTM_ERR_INVALID, "ptm_step needs the whole ladder on this engine; sharded engines use ptm_exchange_*");
  NO_BATCH(e, "ptm_step");
  if (n > 0) {
    int f = 0;
    // (the persistent ladder kernel first: where both apply it steps a small ladder in half the fused kernel's time -- 20 rungs of 6
    //  dimensions with the sampler's defaults: 11 against 22 us)
    if (ladder_applies(e)) {
      // small portions are counted and launched together later (lad_deferred); PTM_LADDER_DEFER=0: every call launches
      static const bool defer_ok = [] { const char* v = getenv("PTM_LADDER_DEFER"); return !(v && *v == '0'); }();
      if (defer_ok && n < 64) {
        e->lad_deferred += n;
        return e->lad_deferred >= 1024 ? ladder_flush(e) : PTM_OK;
      }
      if ((rc = ladder_flush(e))) return rc;
      f = ladder_steps(e, n);   // (launches of that kernel follow each other without a look at the outcome of the one before: ladder_settle)
      if (f < 0) return f;
      n -= f;
    }
    if (n > 0) {
      if ((rc = ladder_settle(e))) return rc;
      f = fused_steps(e, n);
      if (f < 0) return f;
      n -= f;
    }
  }
  if (n > 0 && (rc = ladder_settle(e))) return rc;
  return two_launch_steps(e, n);
}

extern "C" int ptm_sync(ptm_engine* e) {
  if (!e) return fail(PTM_ERR_INVALID, "null engine");
  { const int rc = ladder_settle(e); if (rc) return rc; }
  HIPCHK(hipStreamSynchronize(e->stream));
  if (e->shard && e->shard->cstream) HIPCHK(hipStreamSynchronize(e->shard->cstream));   // (halos left in flight for the next step)
  int flag = 0;
  HIPCHK(hipMemcpy(&flag, e->err, 4, hipMemcpyDeviceToHost));
  if (flag & 1) return fail(PTM_ERR_FAR_MOVE, "a state crossed more than one shard boundary in one step (neighbour exchange mode)");
  if (flag & 2) return fail(PTM_ERR_FAR_MOVE, "an exchange chain reached past the llike halo: rerun with a deeper halo");
  if (flag & 4) return fail(PTM_ERR_FAR_MOVE, "a boundary message overflowed: more rows crossed a shard boundary in one step than "
                            "ptm_config.exchange_row_capacity slots (%d)", e->row_cap);
  if (flag & 8) return fail(PTM_ERR_FAR_MOVE, "a boundary message carried a row this shard did not expect (neighbour shards out of step?)");
  if (flag & 16) return fail(PTM_ERR_UNSUPPORTED, "a recorded rung's in-between history row belongs to the neighbour shard");
  if (flag & 64) return fail(PTM_ERR_INVALID, "differential evolution asked for a saved row the history ring no longer holds: history_capacity (%d rows) must hold the whole run", e->hist.cap);
  if (flag & 128) return fail(PTM_ERR_INVALID, "differential evolution: a thousand history states in a row equalled the current state (the reference exits here)");
  // (bit 32 -- a launch of the persistent ladder kernel gave up -- is not an error any more: ladder_settle has repeated its steps)
  return PTM_OK;
}

extern "C" int ptm_llike_device_ptr(ptm_engine* e, void** p) {
  SETTLE(e);
  if (!e || !p) return fail(PTM_ERR_INVALID, "null argument");
  *p = e->ll;
  return PTM_OK;
}

extern "C" int ptm_exchange_decide(ptm_engine* e, const void* ll_below, const void* ll_above, int halo_rungs, void* send_up,
                                   void* send_down) {
  SETTLE(e);
  NO_BATCH(e, "ptm_exchange_decide");
  int rc = ready(e);
  if (rc) return rc;
  if (e->evolve_rate > 0 && e->nloc != e->Nt)
    return fail(PTM_ERR_INVALID, "an evolving ladder's shard decides from the whole ladder's llikes: ptm_exchange_decide_gathered");
  const bool first = e->r0 == 0, last = e->r0 + e->nloc == e->Nt;
  if ((!first && !ll_below) || (!last && (!ll_above || halo_rungs < 1)))
    return fail(PTM_ERR_INVALID, "missing llike halo: a shard needs the top rung below it and >= 1 rung above it");
  if (!last && halo_rungs > e->Nt - (e->r0 + e->nloc)) return fail(PTM_ERR_INVALID, "halo deeper than the ladder above this shard");
  if (e->Nt > 1)
    return launch_decide(e, first ? nullptr : (const double*)ll_below, last ? nullptr : (const double*)ll_above, halo_rungs,
                         last ? nullptr : (double*)send_up, first ? nullptr : (double*)send_down);
  return PTM_OK;
}

extern "C" int ptm_exchange_decide_gathered(ptm_engine* e, const void* ll_all, const void* lp_all, void* send_up, void* send_down) {
  SETTLE(e);
  NO_BATCH(e, "ptm_exchange_decide_gathered");
  int rc = ready(e);
  if (rc) return rc;
  if (!ll_all) return fail(PTM_ERR_INVALID, "null argument");
  if (e->evolve_rate > 0 && e->evolve_cut >= 0 && !lp_all) return fail(PTM_ERR_INVALID, "a posterior-ordering cut needs the whole ladder's lpriors too");
  const bool first = e->r0 == 0, last = e->r0 + e->nloc == e->Nt;
  if ((!last && !send_up) || (!first && !send_down)) return fail(PTM_ERR_INVALID, "missing boundary message buffer");
  if (e->Nt > 1)
    return launch_decide(e, nullptr, nullptr, 0, last ? nullptr : (double*)send_up, first ? nullptr : (double*)send_down, (const double*)ll_all,
                         (const double*)lp_all);
  return PTM_OK;
}

extern "C" int ptm_set_shard_map(ptm_engine* e, int n_shards, const int32_t* rung_counts, int halo_rungs) {
  SETTLE(e);
  NO_BATCH(e, "ptm_set_shard_map");
  if (!e || !rung_counts || n_shards < 1 || halo_rungs < 1) return fail(PTM_ERR_INVALID, "bad argument");
  std::vector<int> ends(n_shards);
  int at = 0, mine = -1;
  for (int k = 0; k < n_shards; ++k) {
    if (rung_counts[k] < 1) return fail(PTM_ERR_INVALID, "every shard needs at least one rung");
    if (at == e->r0 && rung_counts[k] == e->nloc) mine = k;
    at += rung_counts[k];
    ends[k] = at;
  }
  if (at != e->Nt || mine < 0) return fail(PTM_ERR_INVALID, "the shard map does not describe this engine's block of the ladder");
  int rc;
  if (e->shard_ends) { HIPCHK(hipStreamSynchronize(e->stream)); HIPCHK(hipFree(e->shard_ends)); e->shard_ends = nullptr; }
  if ((rc = dalloc(&e->shard_ends, (size_t)n_shards)) || (rc = upload(e->shard_ends, ends.data(), (size_t)n_shards, e->stream))) return rc;
  if (!e->redo_flag) {
    if ((rc = dalloc(&e->redo_flag, (size_t)e->W + 4))) return rc;
    HIPCHK(hipMemsetAsync(e->redo_flag, 0, ((size_t)e->W + 4) * sizeof(int), e->stream));
  }
  e->nshards = n_shards; e->halo_nominal = halo_rungs;
  return PTM_OK;
}

extern "C" int ptm_exchange_redo_count(ptm_engine* e, int* n) {
  SETTLE(e);
  if (!e || !n) return fail(PTM_ERR_INVALID, "null argument");
  *n = 0;
  if (!e->redo_flag) return PTM_OK;
  HIPCHK(hipMemcpyAsync(n, e->redo_flag + e->W, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return PTM_OK;
}

extern "C" int ptm_exchange_redo(ptm_engine* e, const void* ll_all, const void* lp_all, void* send_up, void* send_down) {
  SETTLE(e);
  NO_BATCH(e, "ptm_exchange_redo");
  int rc = ready(e);
  if (rc) return rc;
  if (!e->redo_flag) return fail(PTM_ERR_INVALID, "ptm_set_shard_map first");
  if (!ll_all) return fail(PTM_ERR_INVALID, "null argument");
  int n = 0;
  if ((rc = ptm_exchange_redo_count(e, &n))) return rc;
  if (n == 0) return PTM_OK;
  e->redo_total += n;
  const bool first = e->r0 == 0, last = e->r0 + e->nloc == e->Nt;
  if ((!last && !send_up) || (!first && !send_down)) return fail(PTM_ERR_INVALID, "missing boundary message buffer");
  rc = launch_decide(e, nullptr, nullptr, 0, last ? nullptr : (double*)send_up, first ? nullptr : (double*)send_down, (const double*)ll_all, (const double*)lp_all, true);
  if (rc) return rc;
  HIPCHK(hipMemsetAsync(e->redo_flag + e->W, 0, sizeof(int), e->stream));
  return PTM_OK;
}

extern "C" int ptm_copy_lprior(ptm_engine* e, int first_local_rung, int n_rungs, void* dst_dev) {
  SETTLE(e);
  if (!e || !dst_dev) return fail(PTM_ERR_INVALID, "null argument");
  if (first_local_rung < 0 || n_rungs < 1 || first_local_rung + n_rungs > e->nloc) return fail(PTM_ERR_INVALID, "rung range out of the shard");
  HIPCHK(hipMemcpyAsync(dst_dev, e->lp + (size_t)first_local_rung * e->W, (size_t)n_rungs * e->W * 8, hipMemcpyDeviceToDevice, e->stream));
  return PTM_OK;
}

extern "C" int ptm_copy_llike(ptm_engine* e, int first_local_rung, int n_rungs, void* dst_dev) {
  SETTLE(e);
  if (!e || !dst_dev) return fail(PTM_ERR_INVALID, "null argument");
  if (first_local_rung < 0 || n_rungs < 1 || first_local_rung + n_rungs > e->nloc) return fail(PTM_ERR_INVALID, "rung range out of the shard");
  HIPCHK(hipMemcpyAsync(dst_dev, e->ll + (size_t)first_local_rung * e->W, (size_t)n_rungs * e->W * 8, hipMemcpyDeviceToDevice, e->stream));
  return PTM_OK;
}

// small device-memory helpers for callers without a GPU array library (tests, tools)
extern "C" int ptm_dev_alloc(size_t bytes, void** out) {
  int rc = need_device();
  if (rc) return rc;
  HIPCHK(hipMalloc(out, bytes ? bytes : 1));
  return PTM_OK;
}
extern "C" int ptm_dev_free(void* p) {
  if (p) HIPCHK(hipFree(p));
  return PTM_OK;
}
extern "C" int ptm_dev_copy(void* dst, const void* src, size_t bytes) {
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDefault));
  return PTM_OK;
}

extern "C" int ptm_exchange_install(ptm_engine* e, const void* recv_below, const void* recv_above) {
  SETTLE(e);
  NO_BATCH(e, "ptm_exchange_install");
  int rc = ready(e);
  if (rc) return rc;
  const bool first = e->r0 == 0, last = e->r0 + e->nloc == e->Nt;
  if ((!first && !recv_below) || (!last && !recv_above)) return fail(PTM_ERR_INVALID, "missing boundary message from a neighbour shard");
  return launch_install(e, first ? nullptr : (const double*)recv_below, last ? nullptr : (const double*)recv_above);
}
extern "C" int ptm_sweep_rungs(ptm_engine* e, int first_local_rung, int n_rungs, int closes_step) {
  SETTLE(e);
  NO_BATCH(e, "ptm_sweep_rungs");
  int rc = ready(e);
  if (rc) return rc;
  return launch_sweep(e, first_local_rung, n_rungs, closes_step != 0);
}

extern "C" int ptm_exchange_buffer_doubles(ptm_engine* e) { return e ? MSG_HDR + e->row_cap * (e->DP + ROW_EXTRA) : 0; }
extern "C" int ptm_exchange_row_capacity(ptm_engine* e) { return e ? e->row_cap : 0; }

extern "C" int ptm_exchange_finish_and_sweep(ptm_engine* e, const void* recv_below, const void* recv_above) {
  SETTLE(e);
  NO_BATCH(e, "ptm_exchange_finish_and_sweep");
  int rc = ready(e);
  if (rc) return rc;
  const bool first = e->r0 == 0, last = e->r0 + e->nloc == e->Nt;
  if ((!first && !recv_below) || (!last && !recv_above)) return fail(PTM_ERR_INVALID, "missing boundary message from a neighbour shard");
  if ((rc = launch_install(e, first ? nullptr : (const double*)recv_below, last ? nullptr : (const double*)recv_above))) return rc;
  return launch_sweep(e);
}


// ---- native RCCL sharding -----------------------------------------------------------------------------------------------------
#define NCCLCHK(x)                                                                                                     \
  do {                                                                                                                 \
    ncclResult_t _r = (x);                                                                                             \
    if (_r != ncclSuccess) return fail(PTM_ERR_HIP, "%s failed: %s (%s:%d)", #x, rccl().GetErrorString(_r), __FILE__, __LINE__); \
  } while (0)

extern "C" int ptm_shard_unique_id(void* id_out) {
  if (!id_out) return fail(PTM_ERR_INVALID, "null argument");
  const char* why = rccl().load();
  if (why) return fail(PTM_ERR_UNSUPPORTED, "RCCL is not available: %s", why);
  static_assert(sizeof(ncclUniqueId) == PTM_SHARD_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  NCCLCHK(rccl().GetUniqueId(&id));
  memcpy(id_out, &id, sizeof id);
  return PTM_OK;
}

extern "C" int ptm_shard_finalize(ptm_engine* e) {
  if (!e || !e->shard) return PTM_OK;
  ShardComm* s = e->shard;
  (void)hipStreamSynchronize(e->stream);
  if (s->cstream) (void)hipStreamSynchronize(s->cstream);
  if (s->comm) (void)rccl().CommDestroy(s->comm);
  double* bufs[] = {s->ll_top, s->ll_bottom, s->ll_below, s->ll_above, s->send_up, s->recv_above, s->send_down, s->recv_below, s->gsend, s->grecv, s->ll_all, s->lp_all};
  if (s->ev_gather) (void)hipEventDestroy(s->ev_gather);
  for (double* b : bufs) if (b) (void)hipFree(b);
  if (s->ev_ready) (void)hipEventDestroy(s->ev_ready);
  if (s->ev_rows) (void)hipEventDestroy(s->ev_rows);
  if (s->ev_halo) (void)hipEventDestroy(s->ev_halo);
  if (s->cstream) (void)hipStreamDestroy(s->cstream);
  delete s;
  e->shard = nullptr;
  return PTM_OK;
}

extern "C" int ptm_shard_init(ptm_engine* e, const void* id, int rank, int world, const int32_t* rung_counts, int halo_rungs) {
  SETTLE(e);
  if (!e || !id || !rung_counts) return fail(PTM_ERR_INVALID, "null argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(PTM_ERR_INVALID, "bad rank / world size");
  if (e->shard) return fail(PTM_ERR_INVALID, "this engine is sharded already (ptm_shard_finalize first)");
  int begin = 0, total = 0;
  for (int r = 0; r < world; ++r) { if (rung_counts[r] < 1) return fail(PTM_ERR_INVALID, "every rank needs at least one rung"); if (r < rank) begin += rung_counts[r]; total += rung_counts[r]; }
  if (total != e->Nt || begin != e->r0 || rung_counts[rank] != e->nloc)
    return fail(PTM_ERR_INVALID, "rung_counts do not describe this engine's block (rank %d: expected rungs %d..%d of %d, the engine holds %d..%d of %d)", rank, begin,
                begin + rung_counts[rank], total, e->r0, e->r0 + e->nloc, e->Nt);
  const char* why = rccl().load();
  if (why) return fail(PTM_ERR_UNSUPPORTED, "RCCL is not available: %s", why);
  ShardComm* s = new ShardComm();
  e->shard = s;
  s->rank = rank; s->world = world; s->halo = halo_rungs > 0 ? halo_rungs : 12;   // (ptmcmc_amd/parallel.py DEFAULT_HALO has the arithmetic)
  s->up = rank + 1 < world ? rank + 1 : -1;
  s->down = rank > 0 ? rank - 1 : -1;
  s->h_recv = s->up >= 0 ? (s->halo < rung_counts[rank + 1] ? s->halo : rung_counts[rank + 1]) : 0;   // what I receive from above is limited by the
  s->h_send = s->down >= 0 ? (s->halo < rung_counts[rank] ? s->halo : rung_counts[rank]) : 0;         // neighbour's size, what I send down by mine
  s->row_doubles = (size_t)ptm_exchange_buffer_doubles(e);
  HIPCHK(hipSetDevice(e->device));
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  NCCLCHK(rccl().CommInitRank(&s->comm, world, uid, rank));
  HIPCHK(hipStreamCreateWithFlags(&s->cstream, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&s->ev_ready, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&s->ev_rows, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&s->ev_halo, hipEventDisableTiming));
  const size_t W = (size_t)e->W;
  int rc;
  if (s->up >= 0 && ((rc = dalloc(&s->ll_top, W)) || (rc = dalloc(&s->ll_above, (size_t)s->h_recv * W)) || (rc = dalloc(&s->send_up, s->row_doubles)) ||
                     (rc = dalloc(&s->recv_above, s->row_doubles))))
    return rc;
  if (s->down >= 0 && ((rc = dalloc(&s->ll_bottom, (size_t)s->h_send * W)) || (rc = dalloc(&s->ll_below, W)) || (rc = dalloc(&s->send_down, s->row_doubles)) ||
                       (rc = dalloc(&s->recv_below, s->row_doubles))))
    return rc;
  s->counts.assign(rung_counts, rung_counts + world);
  for (int r = 0; r < world; ++r) s->maxn = rung_counts[r] > s->maxn ? rung_counts[r] : s->maxn;
  HIPCHK(hipEventCreateWithFlags(&s->ev_gather, hipEventDisableTiming));
  // a halo shallower than the default depth turns the recovery of longer runs on (PTM_SHARD_RECOVER=0/1 overrides; evolving
  // ladders gather every step anyway)
  const char* rv = getenv("PTM_SHARD_RECOVER");
  s->recover = world > 1 && e->evolve_rate <= 0 && (rv ? atoi(rv) != 0 : s->halo < 12);
  if (s->recover && (rc = ptm_set_shard_map(e, world, rung_counts, s->halo))) return rc;
  double* zero[] = {s->send_up, s->recv_above, s->send_down, s->recv_below};
  for (double* b : zero) if (b) HIPCHK(hipMemsetAsync(b, 0, s->row_doubles * 8, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return PTM_OK;
}

// One message round on the side stream: it starts when the engine's stream has reached `after`, and `done` marks its end.
// up_* / down_*: what goes to / comes from the upper / lower neighbour (counts in doubles).
static int shard_exchange(ptm_engine* e, const double* up_send, size_t up_ns, double* up_recv, size_t up_nr, const double* down_send, size_t down_ns,
                          double* down_recv, size_t down_nr, hipEvent_t done) {
  ShardComm* s = e->shard;
  HIPCHK(hipEventRecord(s->ev_ready, e->stream));
  HIPCHK(hipStreamWaitEvent(s->cstream, s->ev_ready, 0));
  NCCLCHK(rccl().GroupStart());
  if (s->up >= 0) {
    NCCLCHK(rccl().Send(up_send, up_ns, ncclDouble, s->up, s->comm, s->cstream));
    NCCLCHK(rccl().Recv(up_recv, up_nr, ncclDouble, s->up, s->comm, s->cstream));
  }
  if (s->down >= 0) {
    NCCLCHK(rccl().Send(down_send, down_ns, ncclDouble, s->down, s->comm, s->cstream));
    NCCLCHK(rccl().Recv(down_recv, down_nr, ncclDouble, s->down, s->comm, s->cstream));
  }
  NCCLCHK(rccl().GroupEnd());
  HIPCHK(hipEventRecord(done, s->cstream));
  return PTM_OK;
}
static int shard_stage_and_start_halos(ptm_engine* e) {
  ShardComm* s = e->shard;
  int rc;
  if (s->up >= 0 && (rc = ptm_copy_llike(e, e->nloc - 1, 1, s->ll_top))) return rc;
  if (s->down >= 0 && (rc = ptm_copy_llike(e, 0, s->h_send, s->ll_bottom))) return rc;
  const size_t W = (size_t)e->W;
  if ((rc = shard_exchange(e, s->ll_top, W, s->ll_above, (size_t)s->h_recv * W, s->ll_bottom, (size_t)s->h_send * W, s->ll_below, W, s->ev_halo))) return rc;
  s->halos_in_flight = true;
  return PTM_OK;
}

// the whole ladder's llikes and lpriors on every shard: one ncclAllGather of the shards' (padded) slabs, unpacked in rung order
// into ll_all / lp_all [Nt][W] on the engine's stream
static int shard_gather_ladder(ptm_engine* e) {
  ShardComm* s = e->shard;
  int rc;
  const size_t W = (size_t)e->W, slab = (size_t)s->maxn * W;
  if (!s->gsend && ((rc = dalloc(&s->gsend, 2 * slab)) || (rc = dalloc(&s->grecv, (size_t)s->world * 2 * slab)) || (rc = dalloc(&s->ll_all, (size_t)e->Nt * W)) ||
                    (rc = dalloc(&s->lp_all, (size_t)e->Nt * W))))
    return rc;
  if ((rc = ptm_copy_llike(e, 0, e->nloc, s->gsend)) || (rc = ptm_copy_lprior(e, 0, e->nloc, s->gsend + slab))) return rc;
  HIPCHK(hipEventRecord(s->ev_ready, e->stream));
  HIPCHK(hipStreamWaitEvent(s->cstream, s->ev_ready, 0));
  NCCLCHK(rccl().AllGather(s->gsend, s->grecv, 2 * slab, ncclDouble, s->comm, s->cstream));
  HIPCHK(hipEventRecord(s->ev_gather, s->cstream));
  HIPCHK(hipStreamWaitEvent(e->stream, s->ev_gather, 0));
  size_t at = 0;
  for (int r = 0; r < s->world; ++r) {   // the shards' slabs, unpadded, in rung order
    const size_t cnt = (size_t)s->counts[r] * W;
    HIPCHK(hipMemcpyAsync(s->ll_all + at, s->grecv + (size_t)r * 2 * slab, cnt * 8, hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(s->lp_all + at, s->grecv + (size_t)r * 2 * slab + slab, cnt * 8, hipMemcpyDeviceToDevice, e->stream));
    at += cnt;
  }
  return PTM_OK;
}

// Runs of surviving picks longer than the halo (ptm_set_shard_map): every shard left the same ladders alone; if there are any
// (one wait on the stream per step to know), the ladder is gathered and they are decided from the full view.
static int shard_recover(ptm_engine* e) {
  ShardComm* s = e->shard;
  if (!s->recover) return PTM_OK;
  int rc, pending = 0;
  if ((rc = ptm_exchange_redo_count(e, &pending)) || !pending) return rc;
  if ((rc = shard_gather_ladder(e))) return rc;
  return ptm_exchange_redo(e, s->ll_all, s->lp_all, s->send_up, s->send_down);
}

extern "C" int ptm_shard_step(ptm_engine* e, int n) {
  SETTLE(e);
  NO_BATCH(e, "ptm_shard_step");
  int rc = ready(e);
  if (rc) return rc;
  if (!e->shard) return fail(PTM_ERR_INVALID, "ptm_shard_init first");
  if (e->cb || e->pcb) return fail(PTM_ERR_UNSUPPORTED, "sharded steps with a host-callback likelihood or host-side proposals are not built");
  ShardComm* s = e->shard;
  if (e->evolve_rate > 0) {
    // Evolving ladders: every shard replays the whole ladder's trials from the whole ladder's llikes (and lpriors, with a
    // posterior-ordering cut) -- one ncclAllGather per step, the reference's gather_llikes / gather_lposts (chain.cc:1433-1435,
    // 1950-1972) -- then the boundary rows travel between neighbours as ever.  No overlap: the trials need the gathered view.
    for (int k = 0; k < n; ++k) {
      if ((rc = shard_gather_ladder(e))) return rc;
      if ((rc = ptm_exchange_decide_gathered(e, s->ll_all, s->lp_all, s->send_up, s->send_down))) return rc;
      if ((rc = shard_exchange(e, s->send_up, s->row_doubles, s->recv_above, s->row_doubles, s->send_down, s->row_doubles, s->recv_below, s->row_doubles, s->ev_rows))) return rc;
      HIPCHK(hipStreamWaitEvent(e->stream, s->ev_rows, 0));
      if ((rc = ptm_exchange_finish_and_sweep(e, s->recv_below, s->recv_above))) return rc;
    }
    return PTM_OK;
  }
  const bool overlap = !e->hist.rungs && !e->map.rungs;   // (a recorded exchanged rung reads its final row: needs the arrivals first)
  // sweep plan: the boundary rungs are the ones whose llikes the neighbours need as halos (bottom h_send rungs, the top rung)
  const int nl = e->nloc, nb = s->h_send < nl ? s->h_send : nl, nt = (s->up >= 0 && nl > nb) ? 1 : 0;
  const int lo = nb, hi = nl - nt, mid = lo + (hi - lo) / 2;
  for (int k = 0; k < n; ++k) {
    if (!s->halos_in_flight && (rc = shard_stage_and_start_halos(e))) return rc;
    HIPCHK(hipStreamWaitEvent(e->stream, s->ev_halo, 0));
    s->halos_in_flight = false;
    if ((rc = ptm_exchange_decide(e, s->ll_below, s->ll_above, s->h_recv, s->send_up, s->send_down))) return rc;
    if ((rc = shard_recover(e))) return rc;
    if ((rc = shard_exchange(e, s->send_up, s->row_doubles, s->recv_above, s->row_doubles, s->send_down, s->row_doubles, s->recv_below, s->row_doubles, s->ev_rows))) return rc;
    if (!overlap) {
      HIPCHK(hipStreamWaitEvent(e->stream, s->ev_rows, 0));
      if ((rc = ptm_exchange_finish_and_sweep(e, s->recv_below, s->recv_above))) return rc;
      continue;
    }
    if ((rc = ptm_sweep_rungs(e, lo, mid - lo, 0))) return rc;            // ... while the boundary rows travel
    HIPCHK(hipStreamWaitEvent(e->stream, s->ev_rows, 0));
    if ((rc = ptm_exchange_install(e, s->recv_below, s->recv_above))) return rc;
    if ((rc = ptm_sweep_rungs(e, 0, nb, 0))) return rc;
    if ((rc = ptm_sweep_rungs(e, nl - nt, nt, 0))) return rc;
    if ((rc = shard_stage_and_start_halos(e))) return rc;                 // the NEXT step's halos ...
    if ((rc = ptm_sweep_rungs(e, mid, hi - mid, 1))) return rc;           // ... travel behind the second half of the interior
  }
  return PTM_OK;
}

// ---- read-back ------------------------------------------------------------------------------------------------------
extern "C" int ptm_get_states(ptm_engine* e, double* X) {
  SETTLE(e);
  if (!e || !X) return fail(PTM_ERR_INVALID, "null argument");
  const size_t Nc = e->Nc, D = e->D, DP = e->DP;
  const unsigned char* s;
  FETCH(s, e->x, Nc * DP * 8);
  e->fetch_after.push_back([=] { unpad_rows((const double*)s, Nc, D, DP, X); });
  return fetch_done(e);
}

extern "C" int ptm_get_array(ptm_engine* e, int which, void* out) {
  SETTLE(e);
  if (!e || !out) return fail(PTM_ERR_INVALID, "null argument");
  const size_t Nc = e->Nc;
  const unsigned char* s;
  switch (which) {
    case PTM_ARR_LLIKE: FETCH(s, e->ll, Nc * 8); e->fetch_after.push_back([=] { memcpy(out, s, Nc * 8); }); break;
    case PTM_ARR_LPRIOR: FETCH(s, e->lp, Nc * 8); e->fetch_after.push_back([=] { memcpy(out, s, Nc * 8); }); break;
    case PTM_ARR_LPOST: {
      if (!e->have_ladder) return fail(PTM_ERR_INVALID, "no ladder set");
      const unsigned char *sl, *sp, *sb = nullptr;
      FETCH(sl, e->ll, Nc * 8);
      FETCH(sp, e->lp, Nc * 8);
      { int rc2 = ensure_betaC(e); if (rc2) return rc2; }
      if (e->betaC) FETCH(sb, e->betaC, Nc * 8);
      e->fetch_after.push_back([=] {
        const double *ll = (const double*)sl, *lp = (const double*)sp, *bc = (const double*)sb;
        double* o = (double*)out;
        for (size_t c = 0; c < Nc; ++c) {
          volatile double t = (bc ? bc[c] : e->h_beta[e->r0 + c / e->W]) * ll[c];  // product rounded before the sum (chain.cc:928)
          o[c] = lp[c] + t;
        }
      });
      break;
    }
    case PTM_ARR_NTRIES: FETCH(s, e->ntries, Nc * 4); e->fetch_after.push_back([=] { memcpy(out, s, Nc * 4); }); break;
    case PTM_ARR_NACCEPT: FETCH(s, e->naccept, Nc * 4); e->fetch_after.push_back([=] { memcpy(out, s, Nc * 4); }); break;
    case PTM_ARR_LAST_TYPE: FETCH(s, e->last_type, Nc * 4); e->fetch_after.push_back([=] { memcpy(out, s, Nc * 4); }); break;
    case PTM_ARR_NHIST:
    case PTM_ARR_NSIZE: {
      { int rc = flush_nhist(e); if (rc) return rc; }
      FETCH(s, e->nhist, Nc * 4);
      const int64_t N = e->cfg.add_every_n;
      // add k (k = 0,1,..) appends a row iff k % N == 0 (chain.cc:935-946); one row exists after initialize(1)
      e->fetch_after.push_back([=] {
        const unsigned int* h = (const unsigned int*)s;
        int64_t* o = (int64_t*)out;
        for (size_t c = 0; c < Nc; ++c) o[c] = which == PTM_ARR_NHIST ? (int64_t)h[c] : 1 + ((int64_t)h[c] + N - 1) / N;
      });
      break;
    }
    default: return fail(PTM_ERR_INVALID, "unknown array id %d", which);
  }
  return fetch_done(e);
}

extern "C" int ptm_get_swap_counts(ptm_engine* e, int64_t* tries, int64_t* accepts) {
  SETTLE(e);
  if (!e) return fail(PTM_ERR_INVALID, "null engine");
  const size_t np = (size_t)e->W * (e->Nt > 1 ? e->Nt - 1 : 1);
  { int rc = fold_swap_log(e); if (rc) return rc; }
  const unsigned char* s;
  FETCH(s, e->swap_cnt, 2 * np * 8);
  e->fetch_after.push_back([=] {
    const long long* both = (const long long*)s;
    for (size_t i = 0; i < np; ++i) {
      if (tries) tries[i] = both[2 * i];
      if (accepts) accepts[i] = both[2 * i + 1];
    }
  });
  return fetch_done(e);
}

extern "C" int ptm_get_last_swaps(ptm_engine* e, int32_t* pairs, int32_t* accepted) {
  SETTLE(e);
  if (!e) return fail(PTM_ERR_INVALID, "null engine");
  const size_t n = (size_t)e->W * e->ms;
  const int newest = (e->log_head + PTM_LOG_RING - 1) % PTM_LOG_RING;
  const unsigned char* s;
  FETCH(s, e->swap_log + (size_t)newest * n, n * 4);
  e->fetch_after.push_back([=] {
    const int32_t* log = (const int32_t*)s;
    for (size_t i = 0; i < n; ++i) {
      const int32_t v = log[i];
      if (pairs) pairs[i] = v < 0 ? (v == -3 ? -3 : -2) : (v & 0x3fffffff);
      if (accepted) accepted[i] = (v >= 0 && (v & 0x40000000)) ? 1 : 0;
    }
  });
  return fetch_done(e);
}

extern "C" int ptm_get_map(ptm_engine* e, double* X, double* lpost, double* llike, double* lprior) {
  SETTLE(e);
  if (!e) return fail(PTM_ERR_INVALID, "null engine");
  if (!e->map.rungs) return fail(PTM_ERR_INVALID, "MAP tracking is off (ptm_config.map_rungs)");
  const size_t n = (size_t)e->map.MC, D = e->D, DP = e->DP;
  const unsigned char* s;
  if (X) { FETCH(s, e->map.x, n * DP * 8); e->fetch_after.push_back([=] { unpad_rows((const double*)s, n, D, DP, X); }); }
  if (lpost) { FETCH(s, e->map.lpost, n * 8); e->fetch_after.push_back([=] { memcpy(lpost, s, n * 8); }); }
  if (llike) { FETCH(s, e->map.ll, n * 8); e->fetch_after.push_back([=] { memcpy(llike, s, n * 8); }); }
  if (lprior) { FETCH(s, e->map.lp, n * 8); e->fetch_after.push_back([=] { memcpy(lprior, s, n * 8); }); }
  return fetch_done(e);
}

extern "C" int ptm_restore(ptm_engine* e, const double* X, const double* llike, const int32_t* ntries, const int32_t* naccept,
                           const int32_t* last_type, const int64_t* nhist, uint64_t step_count, const int64_t* swap_tries,
                           const int64_t* swap_accepts) {
  SETTLE(e);
  NO_BATCH(e, "ptm_restore");
  if (!e || !X || !llike || !ntries || !naccept || !last_type || !nhist) return fail(PTM_ERR_INVALID, "null argument");
  // (a history ring / MAP restart from the restored state here; ptm_set_history / ptm_set_map put saved ones back)
  int rc = ptm_set_states(e, X, llike);   // enforces (a no-op on saved states), recomputes lprior, resets counters
  if (rc) return rc;
  const size_t Nc = e->Nc;
  std::vector<unsigned int> nh(Nc);
  for (size_t c = 0; c < Nc; ++c) {
    if (nhist[c] < 0 || nhist[c] > 0xFFFFFFFFll) return fail(PTM_ERR_INVALID, "nhist out of range");
    nh[c] = (unsigned int)nhist[c];
  }
  if ((rc = upload(e->ntries, ntries, Nc, e->stream)) || (rc = upload(e->naccept, naccept, Nc, e->stream)) ||
      (rc = upload(e->last_type, last_type, Nc, e->stream)) || (rc = upload(e->nhist, nh.data(), Nc, e->stream)))
    return rc;
  const size_t np = (size_t)e->W * (e->Nt > 1 ? e->Nt - 1 : 1);
  {
    std::vector<long long> both(2 * np, 0);
    for (size_t i = 0; i < np; ++i) {
      if (swap_tries) both[2 * i] = swap_tries[i];
      if (swap_accepts) both[2 * i + 1] = swap_accepts[i];
    }
    e->log_pending = 0;   // (the counters are replaced: steps logged before belong to the run that is left)
    if ((rc = upload(e->swap_cnt, both.data(), 2 * np, e->stream))) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));   // `both` leaves scope
  }
  e->step = step_count;
  HIPCHK(hipStreamSynchronize(e->stream));
  return PTM_OK;
}

extern "C" int ptm_set_map(ptm_engine* e, const double* X, const double* lpost, const double* llike, const double* lprior) {
  SETTLE(e);
  NO_BATCH(e, "ptm_set_map");
  if (!e || !X || !lpost || !llike || !lprior) return fail(PTM_ERR_INVALID, "null argument");
  if (!e->map.rungs) return fail(PTM_ERR_INVALID, "MAP tracking is off (ptm_config.map_rungs)");
  const size_t n = (size_t)e->map.MC;
  const std::vector<double> rows = pad_rows(X, n, e->D, e->DP);
  int rc;
  if ((rc = upload(e->map.x, rows.data(), rows.size(), e->stream)) || (rc = upload(e->map.lpost, lpost, n, e->stream)) ||
      (rc = upload(e->map.ll, llike, n, e->stream)) || (rc = upload(e->map.lp, lprior, n, e->stream)))
    return rc;
  HIPCHK(hipStreamSynchronize(e->stream));
  return PTM_OK;
}

extern "C" int ptm_set_history(ptm_engine* e, const double* X, const double* llike, const double* lprior, const int32_t* meta,
                               const double* invtemps) {
  SETTLE(e);
  NO_BATCH(e, "ptm_set_history");
  if (!e || !X || !llike || !lprior || !meta) return fail(PTM_ERR_INVALID, "null argument");
  if (!e->hist.rungs) return fail(PTM_ERR_INVALID, "history is off (ptm_config.history_rungs)");
  if (e->hist.beta && !invtemps) return fail(PTM_ERR_INVALID, "an evolving run's history needs the rows' temperatures (ptm_get_history_invtemps)");
  const size_t n = (size_t)e->hist.cap * e->hist.HC;
  const std::vector<double> rows = pad_rows(X, n, e->D, e->DP);
  int rc;
  if ((rc = upload(e->hist.x, rows.data(), rows.size(), e->stream)) || (rc = upload(e->hist.ll, llike, n, e->stream)) ||
      (rc = upload(e->hist.lp, lprior, n, e->stream)) || (rc = upload((int*)e->hist.meta, (const int*)meta, 4 * n, e->stream)))
    return rc;
  if (e->hist.beta && (rc = upload(e->hist.beta, invtemps, n, e->stream))) return rc;
  HIPCHK(hipStreamSynchronize(e->stream));
  return PTM_OK;
}

extern "C" int ptm_max_swaps_per_step(ptm_engine* e) { return e ? e->ms : 0; }

extern "C" int ptm_get_history(ptm_engine* e, double* X, double* llike, double* lprior, int32_t* meta) {
  SETTLE(e);
  if (!e) return fail(PTM_ERR_INVALID, "null engine");
  if (!e->hist.rungs) return fail(PTM_ERR_INVALID, "history is off (ptm_config.history_rungs)");
  const size_t n = (size_t)e->hist.cap * e->hist.HC, D = e->D, DP = e->DP;
  const unsigned char* s;
  if (X) { FETCH(s, e->hist.x, n * DP * 8); e->fetch_after.push_back([=] { unpad_rows((const double*)s, n, D, DP, X); }); }
  if (llike) { FETCH(s, e->hist.ll, n * 8); e->fetch_after.push_back([=] { memcpy(llike, s, n * 8); }); }
  if (lprior) { FETCH(s, e->hist.lp, n * 8); e->fetch_after.push_back([=] { memcpy(lprior, s, n * 8); }); }
  if (meta) { FETCH(s, e->hist.meta, n * 16); e->fetch_after.push_back([=] { memcpy(meta, s, n * 16); }); }
  return fetch_done(e);
}
// The history ring of chains [chain_begin, chain_begin + chain_count) only, into host arrays of the FULL layout of ptm_get_history
// ([capacity][history chains][..]: the other chains' entries stay as they are): a chain-file writer reads the two or three rungs it
// dumps, not every rung's ring -- with differential evolution on the device the ring holds every rung for the whole run.
extern "C" int ptm_get_history_chains(ptm_engine* e, int chain_begin, int chain_count, double* X, double* llike, double* lprior, int32_t* meta, double* beta) {
  SETTLE(e);
  NO_BATCH(e, "ptm_get_history_chains");
  if (!e) return fail(PTM_ERR_INVALID, "null engine");
  if (!e->hist.rungs) return fail(PTM_ERR_INVALID, "history is off (ptm_config.history_rungs)");
  const size_t cap = (size_t)e->hist.cap, HC = (size_t)e->hist.HC, D = e->D, DP = e->DP;
  if (chain_begin < 0 || chain_count < 0 || (size_t)chain_begin + (size_t)chain_count > HC) return fail(PTM_ERR_INVALID, "chain range outside the history's chains");
  if (!chain_count) return PTM_OK;
  const size_t n = (size_t)chain_count, b0 = (size_t)chain_begin;
  HIPCHK(hipStreamSynchronize(e->stream));
  std::vector<double> tmp;
  if (X) {
    tmp.resize(cap * n * DP);
    HIPCHK(hipMemcpy2D(tmp.data(), n * DP * 8, e->hist.x + b0 * DP, HC * DP * 8, n * DP * 8, cap, hipMemcpyDeviceToHost));
    for (size_t r = 0; r < cap; ++r) unpad_rows(tmp.data() + r * n * DP, n, D, DP, X + (r * HC + b0) * D);
  }
  auto scalars = [&](const double* dev, double* out) -> hipError_t { return hipMemcpy2D(out + b0, HC * 8, dev + b0, HC * 8, n * 8, cap, hipMemcpyDeviceToHost); };
  if (llike) HIPCHK(scalars(e->hist.ll, llike));
  if (lprior) HIPCHK(scalars(e->hist.lp, lprior));
  if (meta) HIPCHK(hipMemcpy2D(meta + 4 * b0, HC * 16, reinterpret_cast<const char*>(e->hist.meta) + b0 * 16, HC * 16, n * 16, cap, hipMemcpyDeviceToHost));
  if (beta) {
    if (e->hist.beta) HIPCHK(scalars(e->hist.beta, beta));
    else {
      if (!e->have_ladder) return fail(PTM_ERR_INVALID, "no ladder set");
      for (size_t r = 0; r < cap; ++r)
        for (size_t c = b0; c < b0 + n; ++c) beta[r * HC + c] = e->h_beta[e->r0 + (int)c / e->W];
    }
  }
  return PTM_OK;
}
extern "C" uint64_t ptm_step_count(ptm_engine* e) {
  if (e) (void)ladder_settle(e);
  return e ? e->step : 0;
}

// ---- measurement ------------------------------------------------------------------------------------------------------
extern "C" int ptm_timer_start(ptm_engine* e) {
  SETTLE(e);
  if (!e) return fail(PTM_ERR_INVALID, "null engine");
  HIPCHK(hipEventRecord(e->t0, e->stream));
  return PTM_OK;
}
extern "C" int ptm_timer_stop(ptm_engine* e, float* ms) {
  SETTLE(e);
  if (!e || !ms) return fail(PTM_ERR_INVALID, "null argument");
  HIPCHK(hipEventRecord(e->t1, e->stream));
  HIPCHK(hipEventSynchronize(e->t1));
  HIPCHK(hipEventElapsedTime(ms, e->t0, e->t1));
  return PTM_OK;
}
extern "C" int ptm_get_kernel_times(ptm_engine* e, float* ms, int capacity, int* count) {
  SETTLE(e);
  if (!e || !count) return fail(PTM_ERR_INVALID, "null argument");
  HIPCHK(hipStreamSynchronize(e->stream));
  int n = 0;
  for (size_t k = 0; k + 1 < e->kev_used + 1 && k + 1 < e->kev.size() + 1 && k < e->kev_used; k += 2) {
    float t = 0;
    if (ms && n < capacity) {   // (ms == NULL: the records are only dropped -- hundreds of event queries are a gap the device idles in)
      HIPCHK(hipEventElapsedTime(&t, e->kev[k], e->kev[k + 1]));
      ms[n] = t;
    }
    n++;
  }
  *count = n;
  e->kev_used = 0;
  return PTM_OK;
}

extern "C" int ptm_get_ladder_stats(ptm_engine* e, int64_t out[4]) {
  if (!e || !out) return fail(PTM_ERR_INVALID, "null argument");
  SETTLE(e);
  out[0] = e->ladder_launches; out[1] = e->ladder_fallbacks; out[2] = e->ladder_whole_steps; out[3] = e->lad_disabled ? 1 : 0;
  return PTM_OK;
}

extern "C" int ptm_get_counter_sums(ptm_engine* e, int64_t* ntries_sum, int64_t* naccept_sum) {
  if (!e) return fail(PTM_ERR_INVALID, "null engine");
  NO_BATCH(e, "ptm_get_counter_sums");
  int rc = ladder_settle(e);
  if (rc) return rc;
  if (!e->sums) {
    HIPCHK(hipMalloc((void**)&e->sums, 16));
    HIPCHK(hipHostMalloc((void**)&e->h_sums, 16, hipHostMallocDefault));
  }
  HIPCHK(hipMemsetAsync(e->sums, 0, 16, e->stream));
  hipLaunchKernelGGL(counter_sums_kernel, dim3(1024), dim3(256), 0, e->stream, e->ntries, e->naccept, (size_t)e->Nc, e->sums);
  HIPCHK(hipMemcpyAsync(e->h_sums, e->sums, 16, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  if (ntries_sum) *ntries_sum = (int64_t)e->h_sums[0];
  if (naccept_sum) *naccept_sum = (int64_t)e->h_sums[1];
  return PTM_OK;
}

extern "C" int ptm_calibrate(ptm_engine* e, ptm_calibration* out) {
  SETTLE(e);
  if (!e || !out) return fail(PTM_ERR_INVALID, "null argument");
  NO_BATCH(e, "ptm_calibrate");
  memset(out, 0, sizeof *out);
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, e->device));
  const int cus = prop.multiProcessorCount;
  out->compute_units = cus;
  hipEvent_t a = nullptr, b = nullptr;
  HIPCHK(hipEventCreate(&a));
  HIPCHK(hipEventCreate(&b));
  // (1) the copy: 2.4 GB read + 2.4 GB written (the bench's state is 4.8 GB), far beyond the 256 MiB Infinity Cache
  const size_t half = (size_t)2400 << 20;
  void *src = nullptr, *dst = nullptr;
  HIPCHK(hipMalloc(&src, half));
  if (hipMalloc(&dst, half) != hipSuccess) { (void)hipFree(src); return fail(PTM_ERR_HIP, "ptm_calibrate: no room for its 4.8 GB of scratch"); }
  HIPCHK(hipMemsetAsync(src, 0x3c, half, e->stream));
  HIPCHK(hipMemsetAsync(dst, 0, half, e->stream));
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    HIPCHK(hipEventRecord(a, e->stream));
    // (a block per 4 KB: tools/probes/copy_probe.hip -- 6.2 TB/s; a persistent grid-stride copy reaches 4.7-5.5, hipMemcpyAsync 4.7)
    hipLaunchKernelGGL(calib_copy_kernel, dim3((unsigned)(half / 16 / 256)), dim3(256), 0, e->stream, (const uint4*)src, (uint4*)dst, half / 16);
    HIPCHK(hipEventRecord(b, e->stream));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    if (rep > 0 && ms < best) best = ms;   // (the first pass warms the clocks and the page tables)
  }
  HIPCHK(hipFree(src));
  HIPCHK(hipFree(dst));
  out->copy_bytes = 2.0 * (double)half;
  out->copy_ms = best;
  out->copy_GBs = 2.0 * (double)half / (best * 1e-3) / 1e9;
  // (2) the f64 issue loop: 4 blocks of 256 threads per CU (four waves per SIMD), 8 chains x iters fma per lane
  const int blocks = cus * 4, iters = 1 << 15;
  double* sink = nullptr;
  long long* clk = nullptr;
  HIPCHK(hipMalloc((void**)&sink, 64));
  HIPCHK(hipMalloc((void**)&clk, (size_t)blocks * 16));
  best = 1e30f;
  std::vector<long long> hc((size_t)blocks * 2);
  for (int rep = 0; rep < 4; ++rep) {
    HIPCHK(hipEventRecord(a, e->stream));
    hipLaunchKernelGGL(calib_fma_kernel, dim3(blocks), dim3(256), 0, e->stream, sink, clk, iters);
    HIPCHK(hipEventRecord(b, e->stream));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    if (rep > 0 && ms < best) { best = ms; HIPCHK(hipMemcpy(hc.data(), clk, hc.size() * 8, hipMemcpyDeviceToHost)); }
  }
  HIPCHK(hipFree(sink));
  HIPCHK(hipFree(clk));
  HIPCHK(hipEventDestroy(a));
  HIPCHK(hipEventDestroy(b));
  std::vector<double> mhz;
  for (int i = 0; i < blocks; ++i)
    if (hc[2 * i + 1] > 0) mhz.push_back(100.0 * (double)hc[2 * i] / (double)hc[2 * i + 1]);
  std::sort(mhz.begin(), mhz.end());
  out->sclk_MHz = mhz.empty() ? 0.0 : mhz[mhz.size() / 2];
  out->fma_ms = best;
  out->f64_fma_TFs = 2.0 * 8.0 * (double)iters * 256.0 * (double)blocks / (best * 1e-3) / 1e12;
  return PTM_OK;
}

extern "C" const char* ptm_sweep_kernel_name(ptm_engine* e) {
  if (!e) return "";
  char b[96];
  const SweepSel s = sweep_sel(e);
  const char* fv = getenv("PTM_FORCE_VALU");
  if (e->DP == 32 && s.uni && !s.callback && !s.host_prop && !s.de && !(fv && *fv && *fv != '0')) {
    const char* cv = getenv("PTM_COMPACT");
    const bool g1 = !s.simple && e->all_uniform && (!e->has_bounds || e->bounds_box);
    const bool cpt = !(cv && *cv == '0') && (s.simple || g1) && !e->hist.rungs && !e->map.rungs && e->W >= 1024 && e->nloc <= 4096;   // (in PT steps; plain sweeps visit every chain)
    snprintf(b, sizeof b, "sweep_mfma32_kernel<%d, %s, %d%s, %s>", s.kind == KIND_DIAG ? KIND_LOWER : s.kind, (e->hist.rungs || e->map.rungs) ? "true" : "false",
             (s.simple || s.lean_ev) ? 0 : ((e->all_uniform && (!e->has_bounds || e->bounds_box)) ? ((cpt && !e->has_mean && !e->any_oned && e->mix_K == 0) ? 3 : 1) : 2),
             (!s.simple && e->all_uniform && (!e->has_bounds || e->bounds_box) && e->betaC) ? ", true" : ", false", cpt ? "true" : "false");   // as rocprofv3 prints it
  }
  else if ((e->DP == 64 || e->DP == 128) && s.uni && e->all_uniform && (!e->has_bounds || e->bounds_box) && !e->has_mean && !e->any_oned && e->mix_K == 0 && !s.callback &&
           !s.host_prop && !e->hist.rungs && !e->map.rungs && !(fv && *fv && *fv != '0'))
    snprintf(b, sizeof b, "sweep_mfma%d_kernel<%d, %s, %s>", e->DP, s.kind == KIND_DIAG ? KIND_LOWER : s.kind, e->has_bounds ? "true" : "false", e->betaC ? "true" : "false");
  else if (e->DP >= 64 || s.host_prop || (!getenv("PTM_FORCE_VALU") && ((!s.uni && (long long)e->Nc * e->DP <= (e->DP >= 16 ? PTM_LANES_MAX : 4096ll * e->DP)) ||
                                                                           (s.uni && s.de && e->DP >= 16 && (long long)e->Nc * e->DP <= (e->DP >= 32 ? (1ll << 21) : (1ll << 19))))))   // (launch_kind's rule)
    snprintf(b, sizeof b, "sweep_lanes_kernel<%d, %d, %s>", e->DP, s.kind, s.plain ? "false" : "true");
  else snprintf(b, sizeof b, "sweep_kernel<%d, %d, %s, %s>", e->DP, s.kind, s.uni ? "true" : "false", s.simple ? "true" : "false");
  e->kname = b;
  return e->kname.c_str();
}

extern "C" const char* ptm_step_kernel_name(ptm_engine* e) {
  if (!e) return "";
  static thread_local std::string name;
  char b[160];
  const bool fused = e->DP <= 16 && (long long)e->Nt * e->DP <= 256 && !e->cb && !e->pcb && !e->cfg.time_kernels && !(getenv("PTM_FUSED") && *getenv("PTM_FUSED") == '0') &&
                     !(e->evolve_rate > 0 && (e->W > 64 || e->evolve_cut >= 0));
  if (e->nloc != e->Nt) snprintf(b, sizeof b, "(sharded: ptm_exchange_* / ptm_shard_step) decide_kernel + %s", ptm_sweep_kernel_name(e));
  else if (ladder_applies(e)) snprintf(b, sizeof b, "ladder_persistent_kernel<%d, %d, %d>", e->DP, e->prop_kind == PTM_PROP_DIAG ? KIND_DIAG : KIND_DENSE, ladder_flavour(e));
  else if (fused) snprintf(b, sizeof b, "ladder_steps_kernel<%d, %d, %d>", e->DP, e->prop_kind == PTM_PROP_DIAG ? KIND_DIAG : KIND_DENSE, (long long)e->Nt * e->DP <= 64 ? 64 : 256);
  else snprintf(b, sizeof b, "decide_kernel + %s", ptm_sweep_kernel_name(e));
  name = b;
  return name.c_str();
}

// ---- verification hooks ------------------------------------------------------------------------------------------------
extern "C" int ptm_debug_eval(int device, int fn, const double* a, const double* b, double* out, int n) {
  int rc = need_device();
  if (rc) return rc;
  if (device >= 0) HIPCHK(hipSetDevice(device));
  double *da = nullptr, *db = nullptr, *dout = nullptr;
  HIPCHK(hipMalloc((void**)&da, (size_t)n * 8));
  HIPCHK(hipMalloc((void**)&db, (size_t)n * 8));
  HIPCHK(hipMalloc((void**)&dout, (size_t)n * 8));
  HIPCHK(hipMemcpy(da, a, (size_t)n * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(db, b ? b : a, (size_t)n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(debug_eval_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, fn, da, db, dout, n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, dout, (size_t)n * 8, hipMemcpyDeviceToHost));
  (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
  return PTM_OK;
}

extern "C" int ptm_debug_philox(int device, uint64_t seed, int tag, uint32_t stream, uint64_t step, uint32_t block, uint32_t out[4]) {
  int rc = need_device();
  if (rc) return rc;
  if (device >= 0) HIPCHK(hipSetDevice(device));
  uint32_t* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, 16));
  hipLaunchKernelGGL(debug_philox_kernel, dim3(1), dim3(1), 0, 0, seed, tag, stream, step, block, d);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, d, 16, hipMemcpyDeviceToHost));
  (void)hipFree(d);
  return PTM_OK;
}

extern "C" int ptm_debug_boxmuller(int device, const uint32_t* k1, const uint32_t* k2, double* z0, double* z1, int n) {
  int rc = need_device();
  if (rc) return rc;
  if (device >= 0) HIPCHK(hipSetDevice(device));
  uint32_t *d1 = nullptr, *d2 = nullptr;
  double *o0 = nullptr, *o1 = nullptr;
  HIPCHK(hipMalloc((void**)&d1, (size_t)n * 4));
  HIPCHK(hipMalloc((void**)&d2, (size_t)n * 4));
  HIPCHK(hipMalloc((void**)&o0, (size_t)n * 8));
  HIPCHK(hipMalloc((void**)&o1, (size_t)n * 8));
  HIPCHK(hipMemcpy(d1, k1, (size_t)n * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d2, k2, (size_t)n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(debug_boxmuller_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, d1, d2, o0, o1, n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(z0, o0, (size_t)n * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(z1, o1, (size_t)n * 8, hipMemcpyDeviceToHost));
  (void)hipFree(d1); (void)hipFree(d2); (void)hipFree(o0); (void)hipFree(o1);
  return PTM_OK;
}

extern "C" int ptm_debug_sqrt_scan(int device, uint64_t* mismatches) {
  int rc = need_device();
  if (rc) return rc;
  if (!mismatches) return fail(PTM_ERR_INVALID, "null argument");
  if (device >= 0) HIPCHK(hipSetDevice(device));
  unsigned long long* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, 16));
  HIPCHK(hipMemset(d, 0, 16));
  hipLaunchKernelGGL(debug_sqrt_scan_kernel, dim3(256 * 16), dim3(256), 0, 0, d);
  HIPCHK(hipGetLastError());
  unsigned long long v[2] = {0, 0};
  HIPCHK(hipMemcpy(v, d, 16, hipMemcpyDeviceToHost));
  (void)hipFree(d);
  *mismatches = v[0] + v[1];   // arguments where bm_sqrt is not the correctly rounded root + arguments outside its domain
  return PTM_OK;
}

extern "C" int ptm_debug_evaluate(ptm_engine* e, const double* X, int n, int32_t* valid, double* Xe, double* lprior, double* llike) {
  if (!e || !X || n < 1) return fail(PTM_ERR_INVALID, "bad argument");
  const size_t D = e->D, DP = e->DP;
  std::vector<double> rows = pad_rows(X, (size_t)n, D, DP);
  double *dx = nullptr, *dlp = nullptr, *dll = nullptr;
  int* dv = nullptr;
  HIPCHK(hipMalloc((void**)&dx, (size_t)n * DP * 8));
  HIPCHK(hipMalloc((void**)&dlp, (size_t)n * 8));
  HIPCHK(hipMalloc((void**)&dll, (size_t)n * 8));
  HIPCHK(hipMalloc((void**)&dv, (size_t)n * 4));
  HIPCHK(hipMemcpy(dx, rows.data(), (size_t)n * DP * 8, hipMemcpyHostToDevice));
  int rc = run_eval(e, n, dx, dv, dlp, dll, (e->have_target && !e->cb) ? 1 : 0);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpy(rows.data(), dx, (size_t)n * DP * 8, hipMemcpyDeviceToHost));
  if (Xe) unpad_rows(rows, (size_t)n, D, DP, Xe);
  if (valid) HIPCHK(hipMemcpy(valid, dv, (size_t)n * 4, hipMemcpyDeviceToHost));
  if (lprior) HIPCHK(hipMemcpy(lprior, dlp, (size_t)n * 8, hipMemcpyDeviceToHost));
  if (llike && e->have_target && !e->cb) HIPCHK(hipMemcpy(llike, dll, (size_t)n * 8, hipMemcpyDeviceToHost));
  (void)hipFree(dx); (void)hipFree(dlp); (void)hipFree(dll); (void)hipFree(dv);
  return PTM_OK;
}
