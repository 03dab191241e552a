"""ctypes binding of libptm_engine.so (C ABI: include/ptm_engine.h).

Thin by design: numpy arrays in, numpy arrays out; every call goes straight through the C ABI
into the HIP engine.  There is no CPU fallback -- without the built library or without an
MI355X the constructor raises.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PTM_ENGINE_LIB") or os.path.join(_HERE, "libptm_engine.so")   # override: A/B builds

# enums of include/ptm_engine.h
BOUND_OPEN, BOUND_LIMIT, BOUND_REFLECT, BOUND_WRAP = 0, 1, 2, 3
PRIOR_FLAT, PRIOR_UNIFORM, PRIOR_GAUSSIAN, PRIOR_POLAR, PRIOR_COPOLAR, PRIOR_LOG = 0, 1, 2, 3, 4, 5
PROP_DENSE, PROP_DIAG, PROP_LOWER = 0, 1, 2
ARR_LLIKE, ARR_LPRIOR, ARR_LPOST, ARR_NTRIES, ARR_NACCEPT, ARR_LAST_TYPE, ARR_NHIST, ARR_NSIZE = range(8)
FN_LOG, FN_EXP, FN_SIN_0_PI, FN_COS_HPI, FN_SQRT, FN_DIV, FN_SQRT_RAW = range(7)
PRIOR_NAMES = {"uni": 1, "uniform": 1, "gauss": 2, "gaussian": 2, "pol": 3, "polar": 3, "cpol": 4, "copol": 4, "log": 5}

EXPORTS = [
    "ptm_last_error", "ptm_abi_version", "ptm_device_count", "ptm_engine_create", "ptm_engine_destroy",
    "ptm_set_bounds", "ptm_set_prior", "ptm_set_target_gaussian", "ptm_set_target_callback", "ptm_set_prior_callback", "ptm_set_ladder", "ptm_set_evolve_temps", "ptm_get_invtemps", "ptm_set_invtemps",
    "ptm_set_proposals", "ptm_set_proposal_rung", "ptm_set_proposal_mixture", "ptm_set_proposal_callback", "ptm_set_proposal_de", "ptm_set_states", "ptm_init_from_prior", "ptm_init_from_prior_k", "ptm_draw_prior_rows", "ptm_get_history_chains", "ptm_sweep", "ptm_step", "ptm_sync",
    "ptm_copy_llike", "ptm_copy_lprior", "ptm_llike_device_ptr", "ptm_exchange_decide", "ptm_exchange_decide_gathered", "ptm_set_shard_map", "ptm_exchange_redo_count", "ptm_exchange_redo", "ptm_exchange_finish_and_sweep", "ptm_exchange_install", "ptm_sweep_rungs", "ptm_exchange_buffer_doubles", "ptm_exchange_row_capacity", "ptm_shard_unique_id", "ptm_shard_init", "ptm_shard_step", "ptm_shard_finalize", "ptm_get_states", "ptm_batch_begin", "ptm_batch_end",
    "ptm_get_array", "ptm_get_swap_counts", "ptm_get_last_swaps", "ptm_max_swaps_per_step", "ptm_get_history", "ptm_get_history_invtemps", "ptm_set_history", "ptm_set_map", "ptm_get_map", "ptm_restore", "ptm_step_count",
    "ptm_timer_start", "ptm_timer_stop", "ptm_get_kernel_times", "ptm_calibrate", "ptm_get_counter_sums", "ptm_get_ladder_stats", "ptm_sweep_kernel_name", "ptm_step_kernel_name", "ptm_debug_eval",
    "ptm_debug_philox", "ptm_debug_boxmuller", "ptm_debug_sqrt_scan", "ptm_debug_evaluate",
    "ptm_dev_alloc", "ptm_dev_free", "ptm_dev_copy",
]


class PtmConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("dim", C.c_int32), ("n_rungs", C.c_int32), ("rung_begin", C.c_int32),
                ("rung_count", C.c_int32), ("n_walkers", C.c_int32), ("seed", C.c_uint64), ("swap_rate", C.c_double),
                ("add_every_n", C.c_int32), ("min_prior", C.c_double), ("device", C.c_int32), ("stream", C.c_void_p),
                ("time_kernels", C.c_int32), ("swap_log_steps", C.c_int32), ("exchange_row_capacity", C.c_int32), ("history_rungs", C.c_int32),
                ("history_capacity", C.c_int32), ("map_rungs", C.c_int32), ("walker_begin", C.c_int32)]


class PtmCalibration(C.Structure):
    _fields_ = [("copy_GBs", C.c_double), ("copy_bytes", C.c_double), ("copy_ms", C.c_double), ("f64_fma_TFs", C.c_double),
                ("fma_ms", C.c_double), ("sclk_MHz", C.c_double), ("compute_units", C.c_int32), ("reserved", C.c_int32)]


class PtmDeParams(C.Structure):
    _fields_ = [("snooker", C.c_double), ("gamma_one_frac", C.c_double), ("reduce_gamma", C.c_double), ("ignore_frac", C.c_double)]


class PtmError(RuntimeError):
    pass


_lib = None
_dp = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_u32p = C.POINTER(C.c_uint32)

LOGLIKE_BATCH_FN = C.CFUNCTYPE(None, C.c_void_p, _dp, C.c_int, C.c_int, _dp)
# ptm_propose_batch_fn / ptm_proposal_result_fn (include/ptm_engine.h)
PROPOSE_BATCH_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, _dp, _i32p, _i32p, C.c_uint64, _dp, _dp, _i32p, _i32p)
PROPOSAL_RESULT_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, _i32p, _i32p, _i32p)


TORCH_LOADED_FIRST = None


def load():
    """dlopen the engine; raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PtmError("HIP engine not built: %s is missing (run `python -c 'import __graft_entry__ as g; g.build()'`)" % LIB_PATH)
    global TORCH_LOADED_FIRST
    TORCH_LOADED_FIRST = "torch" in sys.modules      # (see ptmcmc_amd.parallel.EngineShard: one HIP runtime per process)
    L = C.CDLL(LIB_PATH)
    L.ptm_last_error.restype = C.c_char_p
    L.ptm_sweep_kernel_name.restype = C.c_char_p
    L.ptm_sweep_kernel_name.argtypes = [C.c_void_p]
    if hasattr(L, "ptm_step_kernel_name"):
        L.ptm_step_kernel_name.restype = C.c_char_p
        L.ptm_step_kernel_name.argtypes = [C.c_void_p]
    L.ptm_step_count.restype = C.c_uint64
    L.ptm_step_count.argtypes = [C.c_void_p]
    L.ptm_engine_create.argtypes = [C.POINTER(PtmConfig), C.POINTER(C.c_void_p)]
    L.ptm_engine_destroy.argtypes = [C.c_void_p]
    L.ptm_set_bounds.argtypes = [C.c_void_p, _i32p, _i32p, _dp, _dp]
    L.ptm_set_prior.argtypes = [C.c_void_p, _i32p, _dp, _dp]
    L.ptm_set_target_gaussian.argtypes = [C.c_void_p, _dp, _dp, C.c_double]
    L.ptm_set_target_callback.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    if hasattr(L, "ptm_set_prior_callback"):   # (absent from older builds handed over through PTM_ENGINE_LIB for A/B timing)
        L.ptm_set_prior_callback.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.ptm_set_ladder.argtypes = [C.c_void_p, _dp]
    L.ptm_set_evolve_temps.argtypes = [C.c_void_p, C.c_double, C.c_double]
    L.ptm_get_invtemps.argtypes = [C.c_void_p, _dp]
    L.ptm_set_invtemps.argtypes = [C.c_void_p, _dp]
    L.ptm_set_proposals.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
    L.ptm_set_states.argtypes = [C.c_void_p, _dp, _dp]
    L.ptm_init_from_prior.argtypes = [C.c_void_p]
    L.ptm_init_from_prior_k.argtypes = [C.c_void_p, C.c_int]
    if hasattr(L, "ptm_draw_prior_rows"):
        L.ptm_draw_prior_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp]
    L.ptm_sweep.argtypes = [C.c_void_p, C.c_int]
    L.ptm_step.argtypes = [C.c_void_p, C.c_int]
    L.ptm_sync.argtypes = [C.c_void_p]
    L.ptm_llike_device_ptr.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.ptm_exchange_decide.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.ptm_copy_llike.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    if hasattr(L, "ptm_copy_lprior"):
        L.ptm_copy_lprior.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.ptm_exchange_decide_gathered.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    if hasattr(L, "ptm_set_shard_map"):
        L.ptm_set_shard_map.argtypes = [C.c_void_p, C.c_int, _i32p, C.c_int]
        L.ptm_exchange_redo_count.argtypes = [C.c_void_p, _i32p]
        L.ptm_exchange_redo.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ptm_dev_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
    L.ptm_dev_free.argtypes = [C.c_void_p]
    L.ptm_dev_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.ptm_exchange_finish_and_sweep.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.ptm_exchange_install.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.ptm_sweep_rungs.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.ptm_get_states.argtypes = [C.c_void_p, _dp]
    if hasattr(L, "ptm_batch_begin"):
        L.ptm_batch_begin.argtypes = [C.c_void_p]
        L.ptm_batch_end.argtypes = [C.c_void_p]
    L.ptm_get_array.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.ptm_get_swap_counts.argtypes = [C.c_void_p, _i64p, _i64p]
    L.ptm_get_last_swaps.argtypes = [C.c_void_p, _i32p, _i32p]
    L.ptm_max_swaps_per_step.argtypes = [C.c_void_p]
    L.ptm_timer_start.argtypes = [C.c_void_p]
    L.ptm_timer_stop.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.ptm_get_kernel_times.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]
    L.ptm_calibrate.argtypes = [C.c_void_p, C.POINTER(PtmCalibration)]
    L.ptm_get_counter_sums.argtypes = [C.c_void_p, _i64p, _i64p]
    L.ptm_get_ladder_stats.argtypes = [C.c_void_p, _i64p]
    L.ptm_debug_eval.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, C.c_int]
    L.ptm_debug_philox.argtypes = [C.c_int, C.c_uint64, C.c_int, C.c_uint32, C.c_uint64, C.c_uint32, _u32p]
    L.ptm_debug_boxmuller.argtypes = [C.c_int, _u32p, _u32p, _dp, _dp, C.c_int]
    L.ptm_exchange_buffer_doubles.argtypes = [C.c_void_p]
    L.ptm_exchange_row_capacity.argtypes = [C.c_void_p]
    L.ptm_get_history.argtypes = [C.c_void_p, _dp, _dp, _dp, _i32p]
    L.ptm_get_history_invtemps.argtypes = [C.c_void_p, _dp]
    L.ptm_set_history.argtypes = [C.c_void_p, _dp, _dp, _dp, _i32p, _dp]
    L.ptm_set_map.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
    L.ptm_get_map.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
    L.ptm_set_proposal_rung.argtypes = [C.c_void_p, C.c_int, _dp, C.c_double]
    L.ptm_set_proposal_mixture.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp]
    L.ptm_set_proposal_de.argtypes = [C.c_void_p, C.POINTER(PtmDeParams), C.c_int, _dp]
    L.ptm_set_proposal_callback.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ptm_shard_unique_id.argtypes = [C.c_void_p]
    L.ptm_shard_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _i32p, C.c_int]
    L.ptm_shard_step.argtypes = [C.c_void_p, C.c_int]
    L.ptm_shard_finalize.argtypes = [C.c_void_p]
    L.ptm_restore.argtypes = [C.c_void_p, _dp, _dp, _i32p, _i32p, _i32p, C.POINTER(C.c_int64), C.c_uint64, C.POINTER(C.c_int64),
                              C.POINTER(C.c_int64)]
    L.ptm_debug_sqrt_scan.argtypes = [C.c_int, C.POINTER(C.c_uint64)]
    L.ptm_debug_evaluate.argtypes = [C.c_void_p, _dp, C.c_int, _i32p, _dp, _dp, _dp]
    _lib = L
    return L


def _chk(rc):
    if rc != 0:
        raise PtmError("ptm error %d: %s" % (rc, load().ptm_last_error().decode()))


def _d(a):
    return a.ctypes.data_as(_dp)


def device_count():
    return load().ptm_device_count()


def debug_eval(fn, a, b=None, device=-1):
    a = np.ascontiguousarray(a, dtype=np.float64)
    bb = a if b is None else np.ascontiguousarray(b, dtype=np.float64)
    out = np.empty_like(a)
    _chk(load().ptm_debug_eval(device, fn, _d(a), _d(bb), _d(out), a.size))
    return out


def debug_philox(seed, tag, stream, step, block, device=-1):
    o = (C.c_uint32 * 4)()
    _chk(load().ptm_debug_philox(device, seed, tag, stream, step, block, o))
    return [int(v) for v in o]


def debug_boxmuller(k1, k2, device=-1):
    k1 = np.ascontiguousarray(k1, dtype=np.uint32)
    k2 = np.ascontiguousarray(k2, dtype=np.uint32)
    z0, z1 = np.empty(k1.size), np.empty(k1.size)
    _chk(load().ptm_debug_boxmuller(device, k1.ctypes.data_as(_u32p), k2.ctypes.data_as(_u32p), _d(z0), _d(z1), k1.size))
    return z0, z1


def debug_sqrt_scan(device=-1):
    n = C.c_uint64()
    _chk(load().ptm_debug_sqrt_scan(device, C.byref(n)))
    return n.value


class DeviceBuffer:
    """A raw device allocation (for tests and tools that have no GPU array library at hand)."""

    def __init__(self, nbytes):
        p = C.c_void_p()
        _chk(load().ptm_dev_alloc(nbytes, C.byref(p)))
        self.ptr, self.nbytes = p.value, nbytes

    def data_ptr(self):
        return self.ptr

    def copy_from(self, src_ptr, nbytes=None):
        _chk(load().ptm_dev_copy(self.ptr, src_ptr, self.nbytes if nbytes is None else nbytes))

    def to_numpy(self, dtype=np.float64):
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        _chk(load().ptm_dev_copy(out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes))
        return out

    def free(self):
        if self.ptr:
            load().ptm_dev_free(self.ptr)
            self.ptr = None

    __del__ = free

    def slice(self, offset_bytes, nbytes):
        """a view of part of the allocation (the parent keeps the memory)"""
        assert 0 <= offset_bytes and offset_bytes + nbytes <= self.nbytes
        return _DeviceView(self, offset_bytes, nbytes)


class _DeviceView:
    def __init__(self, parent, offset, nbytes):
        self.parent, self.ptr, self.nbytes = parent, parent.ptr + offset, nbytes

    def data_ptr(self):
        return self.ptr

    def copy_from(self, src_ptr, nbytes=None):
        _chk(load().ptm_dev_copy(self.ptr, src_ptr, self.nbytes if nbytes is None else nbytes))

    def to_numpy(self, dtype=np.float64):
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        _chk(load().ptm_dev_copy(out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes))
        return out


def geometric_ladder(n_rungs, tmax):
    """beta of parallel_tempering_chains' constructor + initialize (chain.cc:1181-1183,1340): host constants."""
    import math
    tratio = math.exp(math.log(tmax) / (n_rungs - 1)) if n_rungs > 1 else 1.0
    t, beta = 1.0, [1.0]
    for _ in range(1, n_rungs):
        t = t * tratio
        beta.append(1 / t)
    return np.array(beta)


class Engine:
    """One GPU shard of a parallel-tempering ladder: rungs [rung_begin, rung_begin+rung_count) x W walkers."""

    def __init__(self, dim, n_rungs, n_walkers=1, seed=0x5EED0001, swap_rate=0.1, add_every_n=1, min_prior=-30.0,
                 rung_begin=0, rung_count=None, device=-1, stream=None, time_kernels=False, exchange_row_capacity=0, history_rungs=0,
                 history_capacity=0, map_rungs=0, walker_begin=0):
        L = load()
        cfg = PtmConfig()
        cfg.struct_size = C.sizeof(PtmConfig)
        cfg.dim, cfg.n_rungs, cfg.rung_begin = dim, n_rungs, rung_begin
        cfg.rung_count = n_rungs if rung_count is None else rung_count
        cfg.n_walkers, cfg.seed, cfg.swap_rate = n_walkers, seed, swap_rate
        cfg.add_every_n, cfg.min_prior, cfg.device = add_every_n, min_prior, device
        cfg.stream = stream
        cfg.time_kernels = 1 if time_kernels else 0
        cfg.swap_log_steps = 0
        cfg.exchange_row_capacity = exchange_row_capacity
        cfg.history_rungs, cfg.history_capacity = history_rungs, history_capacity
        self.hist_rungs, self.hist_cap = history_rungs, history_capacity
        cfg.map_rungs = map_rungs
        self.map_rungs = map_rungs
        cfg.walker_begin = walker_begin
        self.walker_begin = walker_begin
        h = C.c_void_p()
        _chk(L.ptm_engine_create(C.byref(cfg), C.byref(h)))
        self.h, self.L = h, L
        self.D, self.Nt, self.W = dim, n_rungs, n_walkers
        self.r0, self.nloc = rung_begin, cfg.rung_count
        self.Nc = self.nloc * n_walkers
        self._evolving = False
        self._evolve_cut = -1.0
        self._keep = []
        self._batch_depth, self._batch_keep = 0, []

    def close(self):
        if getattr(self, "h", None):
            self.L.ptm_engine_destroy(self.h)
            self.h = None

    __del__ = close

    # -- problem description
    def set_bounds(self, lo, hi, xmin, xmax):
        lo = np.ascontiguousarray(lo, dtype=np.int32); hi = np.ascontiguousarray(hi, dtype=np.int32)
        xmin = np.ascontiguousarray(xmin, dtype=np.float64); xmax = np.ascontiguousarray(xmax, dtype=np.float64)
        _chk(self.L.ptm_set_bounds(self.h, lo.ctypes.data_as(_i32p), hi.ctypes.data_as(_i32p), _d(xmin), _d(xmax)))

    def set_prior(self, types, centers, halfwidths):
        t = np.ascontiguousarray([PRIOR_NAMES[v] if isinstance(v, str) else int(v) for v in types], dtype=np.int32)
        c = np.ascontiguousarray(centers, dtype=np.float64); h = np.ascontiguousarray(halfwidths, dtype=np.float64)
        _chk(self.L.ptm_set_prior(self.h, t.ctypes.data_as(_i32p), _d(c), _d(h)))

    def set_target_gaussian(self, precision, like0, mean=None):
        P = np.ascontiguousarray(precision, dtype=np.float64).reshape(self.D, self.D)
        m = None if mean is None else np.ascontiguousarray(mean, dtype=np.float64)
        _chk(self.L.ptm_set_target_gaussian(self.h, None if m is None else _d(m), _d(P), like0))

    def set_target_callback(self, fn, batched=False):
        """user plug-in likelihood (the C shape of bayes_likelihood::register_evaluate_log, bayesian.hh:536-552).
        fn(x[D]) -> float, or with batched=True fn(X[n][D]) -> array of n."""
        def tramp(user, Xp, n, dim, outp):
            X = np.ctypeslib.as_array(Xp, shape=(n, dim))
            out = np.ctypeslib.as_array(outp, shape=(n,))
            if batched:
                out[:] = fn(X)
            else:
                for k in range(n):
                    out[k] = fn(X[k])
        cb = LOGLIKE_BATCH_FN(tramp)
        self._keep.append(cb)
        _chk(self.L.ptm_set_target_callback(self.h, C.cast(cb, C.c_void_p), None))

    def set_prior_callback(self, fn, batched=False):
        """a prior evaluated on the host (ptm_set_prior_callback): the C shape of probability_function::evaluate_log
        (probability_function.hh:31-44) for priors ptm_set_prior cannot describe.  fn(x[D]) -> log-prior (-inf outside the
        support), or with batched=True fn(X[n][D]) -> array of n.  Needs set_target_callback; start states come from set_states."""
        if fn is None:
            _chk(self.L.ptm_set_prior_callback(self.h, None, None))
            return
        def tramp(user, Xp, n, dim, outp):
            X = np.ctypeslib.as_array(Xp, shape=(n, dim))
            out = np.ctypeslib.as_array(outp, shape=(n,))
            if batched:
                out[:] = fn(X)
            else:
                for k in range(n):
                    out[k] = fn(X[k])
        cb = LOGLIKE_BATCH_FN(tramp)
        self._keep.append(cb)
        _chk(self.L.ptm_set_prior_callback(self.h, C.cast(cb, C.c_void_p), None))

    def set_proposal_callback(self, propose, result=None):
        """host-side proposals (ptm_set_proposal_callback): the C shape of proposal_distribution::draw / log_hastings_ratio /
        type and accept / reject (proposal_distribution.hh:65-87).
        propose: a C callback (PROPOSE_BATCH_FN or any ctypes function pointer of that signature), or a Python function
          propose(X_cur[n][D], rung[n], walker[n], step) -> (X_prop[n][D], log_hastings[n], type[n], valid[n]);
        result:  optional Python function result(rung[n], walker[n], accepted[n]) called after every sweep."""
        if propose is None:
            _chk(self.L.ptm_set_proposal_callback(self.h, None, None, None))
            return
        if not isinstance(propose, C._CFuncPtr):
            fn = propose

            def tramp(user, n, dim, xc, rung, walker, step, xp, lh, ty, va):
                P, H, T, V = fn(np.ctypeslib.as_array(xc, shape=(n, dim)).copy(), np.ctypeslib.as_array(rung, shape=(n,)).copy(),
                                np.ctypeslib.as_array(walker, shape=(n,)).copy(), int(step))
                np.ctypeslib.as_array(xp, shape=(n, dim))[:] = P
                np.ctypeslib.as_array(lh, shape=(n,))[:] = H
                np.ctypeslib.as_array(ty, shape=(n,))[:] = T
                np.ctypeslib.as_array(va, shape=(n,))[:] = V
            propose = PROPOSE_BATCH_FN(tramp)
        rcb = None
        if result is not None:
            rf = result

            def rtramp(user, n, rung, walker, acc):
                rf(np.ctypeslib.as_array(rung, shape=(n,)).copy(), np.ctypeslib.as_array(walker, shape=(n,)).copy(),
                   np.ctypeslib.as_array(acc, shape=(n,)).copy())
            rcb = PROPOSAL_RESULT_FN(rtramp)
        self._keep += [propose, rcb]
        _chk(self.L.ptm_set_proposal_callback(self.h, C.cast(propose, C.c_void_p), None if rcb is None else C.cast(rcb, C.c_void_p), None))

    def set_ladder(self, beta):
        b = np.ascontiguousarray(beta, dtype=np.float64)
        assert b.size == self.Nt
        _chk(self.L.ptm_set_ladder(self.h, _d(b)))

    def set_evolve_temps(self, rate, lpost_cut=-1.0):
        """parallel_tempering_chains::evolve_temps (chain.hh:302-307): every accepted exchange pries its gap apart"""
        _chk(self.L.ptm_set_evolve_temps(self.h, float(rate), float(lpost_cut)))
        self._evolving = self._evolving or rate > 0
        self._evolve_cut = lpost_cut if lpost_cut >= 0 else -1.0

    def invtemps(self):
        """[W][Nt] inverse temperatures of every ladder"""
        b = self._out(np.empty((self.W, self.Nt)))
        _chk(self.L.ptm_get_invtemps(self.h, b.ctypes.data_as(_dp)))
        return b

    def set_invtemps(self, beta):
        b = np.ascontiguousarray(beta, dtype=np.float64)
        assert b.size == self.W * self.Nt
        _chk(self.L.ptm_set_invtemps(self.h, b.ctypes.data_as(_dp)))

    def set_proposals(self, kind, factors, one_d_frac=None):
        f = np.ascontiguousarray(factors, dtype=np.float64)
        per = self.D if kind == PROP_DIAG else self.D * self.D
        assert f.size == self.nloc * per, (f.size, self.nloc, per)
        o = None if one_d_frac is None else np.ascontiguousarray(one_d_frac, dtype=np.float64)
        _chk(self.L.ptm_set_proposals(self.h, kind, _d(f), None if o is None else _d(o)))

    # -- state
    def set_states(self, X, llike=None):
        X = np.ascontiguousarray(X, dtype=np.float64).reshape(self.Nc, self.D)
        ll = None if llike is None else np.ascontiguousarray(llike, dtype=np.float64)
        _chk(self.L.ptm_set_states(self.h, _d(X), None if ll is None else _d(ll)))

    # -- native RCCL sharding (ptm_shard_*): the C/C++ host's form of ptmcmc_amd.parallel.ShardedLadder
    @staticmethod
    def shard_unique_id():
        buf = C.create_string_buffer(128)
        _chk(load().ptm_shard_unique_id(buf))
        return buf.raw

    def shard_init(self, uid, rank, world, rung_counts, halo=0):
        rc = np.ascontiguousarray(rung_counts, dtype=np.int32)
        _chk(self.L.ptm_shard_init(self.h, C.c_char_p(uid), rank, world, rc.ctypes.data_as(_i32p), halo))

    def shard_step(self, n=1):
        _chk(self.L.ptm_shard_step(self.h, n))

    def shard_finalize(self):
        _chk(self.L.ptm_shard_finalize(self.h))

    def init_from_prior(self, k=0):
        """the k-th initial draw of every chain (k = 0: MH_chain::initialize(1)'s; ptm_init_from_prior_k)"""
        _chk(self.L.ptm_init_from_prior_k(self.h, int(k)))

    def history_chains(self, chain_begin, chain_count):
        """the ring entries of a range of history chains, in arrays of history()'s full layout whose other entries are NaN / -1 (ptm_get_history_chains)"""
        cap, HC, D = self.hist_cap, self.hist_rungs * self.W, self.D
        x = np.full((cap, HC, D), np.nan); ll = np.full((cap, HC), np.nan); lp = np.full((cap, HC), np.nan); beta = np.full((cap, HC), np.nan)
        meta = np.full((cap, HC, 4), -1, dtype=np.int32)
        self.L.ptm_get_history_chains.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp, C.c_void_p, _dp]
        _chk(self.L.ptm_get_history_chains(self.h, int(chain_begin), int(chain_count), x.ctypes.data_as(_dp), ll.ctypes.data_as(_dp), lp.ctypes.data_as(_dp),
                                           meta.ctypes.data, beta.ctypes.data_as(_dp)))
        return dict(x=x, llike=ll, lprior=lp, meta=meta, invtemp=beta)

    def draw_prior_rows(self, k_begin, n):
        """draws k_begin .. k_begin + n - 1 of every chain, the engine's state untouched (ptm_draw_prior_rows): x [n][Nc][D], llike, lprior [n][Nc]"""
        x = np.empty((n, self.Nc, self.D)); ll = np.empty((n, self.Nc)); lp = np.empty((n, self.Nc))
        _chk(self.L.ptm_draw_prior_rows(self.h, int(k_begin), int(n), x.ctypes.data_as(_dp), ll.ctypes.data_as(_dp), lp.ctypes.data_as(_dp)))
        return x, ll, lp

    # -- hot path
    def sweep(self, n=1):
        _chk(self.L.ptm_sweep(self.h, n))

    def step(self, n=1):
        _chk(self.L.ptm_step(self.h, n))

    def sync(self):
        _chk(self.L.ptm_sync(self.h))

    def llike_device_ptr(self):
        p = C.c_void_p()
        _chk(self.L.ptm_llike_device_ptr(self.h, C.byref(p)))
        return p.value

    def copy_llike(self, first_local_rung, n_rungs, dst_dev):
        _chk(self.L.ptm_copy_llike(self.h, first_local_rung, n_rungs, dst_dev))

    def copy_lprior(self, first_local_rung, n_rungs, dst_dev):
        _chk(self.L.ptm_copy_lprior(self.h, first_local_rung, n_rungs, dst_dev))

    def exchange_decide_gathered(self, ll_all_dev, lp_all_dev, send_up_dev, send_down_dev):
        """the exchange phase from the whole ladder's llikes [Nt][W] (and lpriors, with a posterior-ordering cut)"""
        _chk(self.L.ptm_exchange_decide_gathered(self.h, ll_all_dev, lp_all_dev, send_up_dev, send_down_dev))

    def set_shard_map(self, rung_counts, halo):
        """recovery of runs longer than a halo (ptm_set_shard_map): every shard's rung count, the halo depth all ask for"""
        rc = np.ascontiguousarray(rung_counts, dtype=np.int32)
        _chk(self.L.ptm_set_shard_map(self.h, rc.size, rc.ctypes.data_as(_i32p), int(halo)))

    def exchange_redo_count(self):
        n = C.c_int32()
        _chk(self.L.ptm_exchange_redo_count(self.h, C.byref(n)))
        return n.value

    def exchange_redo(self, ll_all_dev, lp_all_dev, send_up_dev, send_down_dev):
        _chk(self.L.ptm_exchange_redo(self.h, ll_all_dev, lp_all_dev, send_up_dev, send_down_dev))

    def exchange_decide(self, ll_below_dev, ll_above_dev, halo_rungs, send_up_dev, send_down_dev):
        _chk(self.L.ptm_exchange_decide(self.h, ll_below_dev, ll_above_dev, halo_rungs, send_up_dev, send_down_dev))

    def exchange_install(self, recv_below_dev, recv_above_dev):
        _chk(self.L.ptm_exchange_install(self.h, recv_below_dev, recv_above_dev))

    def sweep_rungs(self, first_local_rung, n_rungs, closes_step):
        _chk(self.L.ptm_sweep_rungs(self.h, first_local_rung, n_rungs, 1 if closes_step else 0))

    @property
    def exchange_buffer_doubles(self):
        return self.L.ptm_exchange_buffer_doubles(self.h)

    def set_proposal_mixture(self, cum_shares, scales, one_d_fracs):
        """arrays [rung_count][K]; K = 0 (empty arrays) removes the mixture"""
        cs, sc, od = (np.ascontiguousarray(a, dtype=np.float64) for a in (cum_shares, scales, one_d_fracs))
        K = 0 if cs.size == 0 else cs.shape[1]
        _chk(self.L.ptm_set_proposal_mixture(self.h, K, cs.ctypes.data_as(_dp), sc.ctypes.data_as(_dp), od.ctypes.data_as(_dp)))

    def set_proposal_de(self, snooker=0.1, gamma_one_frac=0.3, reduce_gamma=4.0, ignore_frac=0.0, init_rows=None, off=False):
        """differential evolution on the device as the member of negative scale of the proposal sets (ptm_set_proposal_de);
        init_rows [n_extra][Nc][D] in the engine's chain order"""
        if off:
            _chk(self.L.ptm_set_proposal_de(self.h, None, 0, None))
            return
        q = PtmDeParams(snooker, gamma_one_frac, reduce_gamma, ignore_frac)
        ir = None if init_rows is None else np.ascontiguousarray(init_rows, dtype=np.float64).reshape(-1, self.Nc, self.D)
        _chk(self.L.ptm_set_proposal_de(self.h, C.byref(q), 0 if ir is None else ir.shape[0], None if ir is None else ir.ctypes.data_as(_dp)))

    def set_proposal_rung(self, local_rung, factor, one_d_frac=-1.0):
        f = np.ascontiguousarray(factor, dtype=np.float64)
        _chk(self.L.ptm_set_proposal_rung(self.h, local_rung, f.ctypes.data_as(_dp), float(one_d_frac)))

    def map(self):
        """MAP of the tracked rungs: dict x [map_rungs*W][D], lpost, llike, lprior [map_rungs*W]"""
        n = self.map_rungs * self.W
        X = self._out(np.empty((n, self.D))); lpo = self._out(np.empty(n)); ll = self._out(np.empty(n)); lp = self._out(np.empty(n))
        _chk(self.L.ptm_get_map(self.h, X.ctypes.data_as(_dp), lpo.ctypes.data_as(_dp), ll.ctypes.data_as(_dp), lp.ctypes.data_as(_dp)))
        return dict(x=X, lpost=lpo, llike=ll, lprior=lp)

    def checkpoint(self):
        """everything the run's future depends on (ptm_restore)"""
        t, a = self.swap_counts()
        return dict(invtemps=self.invtemps() if self._evolving else None, history=self.history() if self.hist_rungs else None,
                    map=self.map() if self.map_rungs else None, x=self.states(), llike=self.llike, ntries=self.ntries.astype(np.int32), naccept=self.naccept.astype(np.int32),
                    last_type=self.last_type.astype(np.int32), nhist=self.nhist.astype(np.int64), step=self.step_count,
                    swap_tries=np.ascontiguousarray(t, dtype=np.int64), swap_accepts=np.ascontiguousarray(a, dtype=np.int64))

    def restore(self, ck):
        i64 = lambda v: np.ascontiguousarray(v, dtype=np.int64).ctypes.data_as(C.POINTER(C.c_int64))
        i32 = lambda v: np.ascontiguousarray(v, dtype=np.int32).ctypes.data_as(_i32p)
        f64 = lambda v: np.ascontiguousarray(v, dtype=np.float64).ctypes.data_as(_dp)
        _chk(self.L.ptm_restore(self.h, f64(ck["x"]), f64(ck["llike"]), i32(ck["ntries"]), i32(ck["naccept"]), i32(ck["last_type"]),
                                i64(ck["nhist"]), int(ck["step"]), i64(ck["swap_tries"]), i64(ck["swap_accepts"])))
        if ck.get("invtemps") is not None:   # an evolving run: set_evolve_temps first, then the ladders as they were
            self.set_invtemps(ck["invtemps"])
        h = ck.get("history")
        if h is not None and self.hist_rungs:
            meta = np.ascontiguousarray(np.stack([h["naccept"], h["ntries"], h["last_type"], h["row"]], axis=-1), dtype=np.int32)
            _chk(self.L.ptm_set_history(self.h, f64(h["x"]), f64(h["llike"]), f64(h["lprior"]), meta.ctypes.data_as(_i32p),
                                        f64(h["invtemp"])))
        m = ck.get("map")
        if m is not None and self.map_rungs:
            _chk(self.L.ptm_set_map(self.h, f64(m["x"]), f64(m["lpost"]), f64(m["llike"]), f64(m["lprior"])))

    def history(self):
        """dict of arrays [cap][history_rungs*W](,D): x, llike, lprior, naccept, ntries, last_type, row (saved row number,
        -1 = empty slot), invtemp (the temperature the row was saved at); saved row s of a chain sits in slot s % cap"""
        n = self.hist_cap * self.hist_rungs * self.W
        X = self._out(np.empty((n, self.D))); ll = self._out(np.empty(n)); lp = self._out(np.empty(n)); meta = self._out(np.empty((n, 4), dtype=np.int32))
        _chk(self.L.ptm_get_history(self.h, X.ctypes.data_as(_dp), ll.ctypes.data_as(_dp), lp.ctypes.data_as(_dp),
                                    meta.ctypes.data_as(_i32p)))
        sh = (self.hist_cap, self.hist_rungs * self.W)
        b = self._out(np.empty(n))
        _chk(self.L.ptm_get_history_invtemps(self.h, b.ctypes.data_as(_dp)))
        m = meta.reshape(sh + (4,))   # (views, not copies: inside a batch() the arrays are filled later)
        return dict(x=X.reshape(sh + (self.D,)), llike=ll.reshape(sh), lprior=lp.reshape(sh), naccept=m[..., 0],
                    ntries=m[..., 1], last_type=m[..., 2], row=m[..., 3], invtemp=b.reshape(sh))

    @property
    def exchange_row_capacity(self):
        return self.L.ptm_exchange_row_capacity(self.h)

    def exchange_finish_and_sweep(self, recv_below_dev, recv_above_dev):
        _chk(self.L.ptm_exchange_finish_and_sweep(self.h, recv_below_dev, recv_above_dev))

    # -- read-back
    def batch(self):
        """context manager: the reads inside only queue their copies; the arrays they returned are filled when the block ends
        (ptm_batch_begin / ptm_batch_end: one wait on the device for all of them).  Keep the returned arrays alive until then."""
        eng = self
        class _Batch:
            def __enter__(self):
                _chk(eng.L.ptm_batch_begin(eng.h))
                eng._batch_depth += 1
            def __exit__(self, *exc):
                try:
                    _chk(eng.L.ptm_batch_end(eng.h))
                finally:
                    eng._batch_depth -= 1
                    if eng._batch_depth == 0:
                        eng._batch_keep.clear()     # the outputs are filled: temporaries may go now
                return False
        return _Batch()

    def _out(self, arr):
        """inside a batch() the C side fills an output when the bracket ends: hold a reference to every array handed out until
        then, so that a temporary (eng.naccept.sum()) or a discarded return value is not freed under the pending write"""
        if self._batch_depth > 0:
            self._batch_keep.append(arr)
        return arr

    def states(self):
        X = self._out(np.empty((self.Nc, self.D)))
        _chk(self.L.ptm_get_states(self.h, _d(X)))
        return X

    def array(self, which):
        dt = {ARR_LLIKE: np.float64, ARR_LPRIOR: np.float64, ARR_LPOST: np.float64, ARR_NTRIES: np.int32,
              ARR_NACCEPT: np.int32, ARR_LAST_TYPE: np.int32, ARR_NHIST: np.int64, ARR_NSIZE: np.int64}[which]
        out = self._out(np.empty(self.Nc, dtype=dt))
        _chk(self.L.ptm_get_array(self.h, which, out.ctypes.data_as(C.c_void_p)))
        return out

    llike = property(lambda s: s.array(ARR_LLIKE))
    lprior = property(lambda s: s.array(ARR_LPRIOR))
    lpost = property(lambda s: s.array(ARR_LPOST))
    ntries = property(lambda s: s.array(ARR_NTRIES))
    naccept = property(lambda s: s.array(ARR_NACCEPT))
    last_type = property(lambda s: s.array(ARR_LAST_TYPE))
    nhist = property(lambda s: s.array(ARR_NHIST))
    nsize = property(lambda s: s.array(ARR_NSIZE))

    def swap_counts(self):
        n = self.W * max(self.Nt - 1, 1)
        t, a = self._out(np.empty(n, dtype=np.int64)), self._out(np.empty(n, dtype=np.int64))
        _chk(self.L.ptm_get_swap_counts(self.h, t.ctypes.data_as(_i64p), a.ctypes.data_as(_i64p)))
        return t.reshape(self.W, -1), a.reshape(self.W, -1)

    def last_swaps(self):
        ms = self.max_swaps
        p, a = self._out(np.empty(self.W * ms, dtype=np.int32)), self._out(np.empty(self.W * ms, dtype=np.int32))
        _chk(self.L.ptm_get_last_swaps(self.h, p.ctypes.data_as(_i32p), a.ctypes.data_as(_i32p)))
        return p.reshape(self.W, ms), a.reshape(self.W, ms)

    @property
    def max_swaps(self):
        return self.L.ptm_max_swaps_per_step(self.h)

    @property
    def step_count(self):
        return int(self.L.ptm_step_count(self.h))

    def debug_evaluate(self, X):
        X = np.ascontiguousarray(X, dtype=np.float64).reshape(-1, self.D)
        n = X.shape[0]
        valid = np.empty(n, dtype=np.int32); Xe = np.empty_like(X); lp = np.empty(n); ll = np.full(n, np.nan)
        _chk(self.L.ptm_debug_evaluate(self.h, _d(X), n, valid.ctypes.data_as(_i32p), _d(Xe), _d(lp), _d(ll)))
        return valid, Xe, lp, ll

    # -- measurement
    def timer_start(self):
        _chk(self.L.ptm_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        _chk(self.L.ptm_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def kernel_times(self, capacity=1 << 16, drop=False):
        """per-launch durations (ms) of the sweep kernel since the last call; drop=True: forget them unread (no event queries)"""
        n = C.c_int()
        if drop:
            _chk(self.L.ptm_get_kernel_times(self.h, None, 0, C.byref(n)))
            return np.zeros(0)
        buf = (C.c_float * capacity)()
        _chk(self.L.ptm_get_kernel_times(self.h, buf, capacity, C.byref(n)))
        return np.array(buf[:min(n.value, capacity)], dtype=np.float64)

    def ladder_stats(self):
        """persistent ladder kernel: launches, launches that gave up (repeated on the two-launch path), whole-ladder steps, switched off"""
        o = (C.c_int64 * 4)()
        _chk(self.L.ptm_get_ladder_stats(self.h, o))
        return dict(launches=o[0], fallbacks=o[1], whole_steps=o[2], disabled=bool(o[3]))

    def counter_sums(self):
        """(sum of MH_chain::Ntries, sum of ::Naccept) over the engine's chains, reduced on the device (ptm_get_counter_sums)"""
        t, a = C.c_int64(), C.c_int64()
        _chk(self.L.ptm_get_counter_sums(self.h, C.byref(t), C.byref(a)))
        return t.value, a.value

    def calibrate(self):
        """what this device gives a streaming copy and an f64 fma loop right now (ptm_calibrate)"""
        c = PtmCalibration()
        _chk(self.L.ptm_calibrate(self.h, C.byref(c)))
        return {"copy_GBs": c.copy_GBs, "copy_bytes": c.copy_bytes, "copy_ms": c.copy_ms, "f64_fma_TFs": c.f64_fma_TFs,
                "fma_ms": c.fma_ms, "sclk_MHz": c.sclk_MHz, "compute_units": c.compute_units}

    @property
    def sweep_kernel_name(self):
        return self.L.ptm_sweep_kernel_name(self.h).decode()

    @property
    def step_kernel_name(self):
        """what step() launches (ptm_step_kernel_name)"""
        return self.L.ptm_step_kernel_name(self.h).decode()
