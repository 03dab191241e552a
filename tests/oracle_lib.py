"""ctypes binding of oracle/libptm_oracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is the plain-C CPU restatement of the reference's chain::step() path
(oracle/ptm_oracle.c).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product (ptmcmc_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_SO = os.path.join(ORACLE_DIR, "libptm_oracle.so")

OPEN, LIMIT, REFLECT, WRAP = 0, 1, 2, 3
FLAT, UNIFORM, GAUSSIAN, POLAR, COPOLAR, LOG = 0, 1, 2, 3, 4, 5
PROP_DENSE, PROP_DIAG = 0, 1
TYPE_NAMES = {"uni": UNIFORM, "uniform": UNIFORM, "gauss": GAUSSIAN, "gaussian": GAUSSIAN, "pol": POLAR,
              "polar": POLAR, "cpol": COPOLAR, "copol": COPOLAR, "log": LOG}

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_u32p = C.POINTER(C.c_uint32)


def build(force=False):
    if os.environ.get("PTM_ORACLE_LIB"):     # e.g. a sanitizer build of the same source
        return os.environ["PTM_ORACLE_LIB"]
    src = [os.path.join(ORACLE_DIR, f) for f in ("ptm_oracle.c", "ptm_oracle.h", "ptm_tables.inc")]
    stale = force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if stale:
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])
    return _SO


class _Problem(C.Structure):
    _fields_ = [("D", C.c_int), ("blo", _ip), ("bhi", _ip), ("bmin", _dp), ("bmax", _dp), ("origin_valid", C.c_int),
                ("ptype", _ip), ("plo", _dp), ("phi", _dp), ("pcoef", _dp), ("all_uniform", C.c_int),
                ("lprior_const", C.c_double), ("have_gauss", C.c_int), ("mean", _dp), ("P2", _dp),
                ("like0", C.c_double), ("user_fn", C.c_void_p), ("user", C.c_void_p), ("minPrior", C.c_double),
                ("prior_fn", C.c_void_p), ("prior_user", C.c_void_p)]


class _Proposal(C.Structure):  # (fields below; K, mix appended for scale mixtures)
    _fields_ = [("kind", C.c_int), ("M", _dp), ("oneDfrac", C.c_double), ("K", C.c_int), ("mix", _dp)]


class _PT(C.Structure):
    _fields_ = [("D", C.c_int), ("Nt", C.c_int), ("W", C.c_int), ("beta", _dp), ("swap_rate", C.c_double),
                ("maxswaps", C.c_int), ("add_every_N", C.c_int), ("step", C.c_uint64), ("x", _dp), ("llike", _dp),
                ("lprior", _dp), ("ntries", C.POINTER(C.c_int32)), ("naccept", C.POINTER(C.c_int32)),
                ("last_type", C.POINTER(C.c_int32)), ("nhist", C.POINTER(C.c_int64)), ("nsize", C.POINTER(C.c_int64)),
                ("swap_count", C.POINTER(C.c_int64)), ("swap_accept_count", C.POINTER(C.c_int64)),
                ("last_pairs", _ip), ("last_accept", _ip), ("touched", C.POINTER(C.c_uint8)),
                ("hist_cap", C.c_int), ("hist_x", _dp), ("hist_ll", _dp), ("hist_lp", _dp),
                ("hist_nacc", C.POINTER(C.c_int32)), ("hist_ntry", C.POINTER(C.c_int32)), ("hist_type", C.POINTER(C.c_int32)),
                ("map_lpost", _dp), ("map_x", _dp), ("evolve_rate", C.c_double), ("betaw", _dp), ("hist_beta", _dp),
                ("host_prop", C.c_void_p), ("host_prop_user", C.c_void_p), ("last_accept_mh", C.POINTER(C.c_uint8)),
                ("evolve_cut", C.c_double), ("de_on", C.c_int), ("de", C.c_double * 4), ("de_init_extra", C.c_int), ("de_init", _dp)]


class _DeParams(C.Structure):
    _fields_ = [("snooker", C.c_double), ("gamma_one_frac", C.c_double), ("reduce_gamma", C.c_double), ("ignore_frac", C.c_double)]


DE_UNIFORM_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_int)
DE_ROW_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_long)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    L.ptmo_philox4x32_10.argtypes = [_u32p, _u32p, _u32p]
    L.ptmo_u01.restype = C.c_double
    L.ptmo_u01.argtypes = [C.c_uint32]
    L.ptmo_bm_neg2log.restype = C.c_double
    L.ptmo_bm_neg2log.argtypes = [C.c_uint32]
    L.ptmo_draw_block.argtypes = [C.c_uint64, C.c_int, C.c_uint32, C.c_uint64, C.c_uint32, _u32p]
    L.ptmo_boxmuller.argtypes = [C.c_uint32, C.c_uint32, _dp, _dp]
    for f in ("ptmo_log", "ptmo_exp", "ptmo_sin_0_pi", "ptmo_cos_hpi"):
        getattr(L, f).restype = C.c_double
        getattr(L, f).argtypes = [C.c_double]
    L.ptmo_boundary_enforce.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, _dp]
    L.ptmo_enforce.argtypes = [C.POINTER(_Problem), _dp]
    L.ptmo_lprior.restype = C.c_double
    L.ptmo_lprior.argtypes = [C.POINTER(_Problem), _dp, C.c_int]
    L.ptmo_llike.restype = C.c_double
    L.ptmo_llike.argtypes = [C.POINTER(_Problem), _dp]
    L.ptmo_lpost.restype = C.c_double
    L.ptmo_lpost.argtypes = [C.c_double] * 3
    L.ptmo_ladder.argtypes = [C.c_int, C.c_double, _dp]
    L.ptmo_problem_create.restype = C.POINTER(_Problem)
    L.ptmo_problem_create.argtypes = [C.c_int]
    L.ptmo_problem_free.argtypes = [C.POINTER(_Problem)]
    L.ptmo_problem_set_bounds.argtypes = [C.POINTER(_Problem), _ip, _ip, _dp, _dp]
    L.ptmo_problem_set_prior.argtypes = [C.POINTER(_Problem), _ip, _dp, _dp]
    L.ptmo_problem_set_gauss.argtypes = [C.POINTER(_Problem), _dp, _dp, C.c_double]
    L.ptmo_problem_set_user.argtypes = [C.POINTER(_Problem), C.c_void_p, C.c_void_p]
    L.ptmo_problem_set_user_prior.argtypes = [C.POINTER(_Problem), C.c_void_p, C.c_void_p]
    L.ptmo_pt_create.restype = C.POINTER(_PT)
    L.ptmo_pt_create.argtypes = [C.c_int, C.c_int, C.c_int, _dp, C.c_double, C.c_int]
    L.ptmo_pt_free.argtypes = [C.POINTER(_PT)]
    L.ptmo_pt_enable_history.argtypes = [C.POINTER(_PT), C.c_int]
    L.ptmo_pt_evolve_temps.argtypes = [C.POINTER(_PT), C.c_double]
    L.ptmo_pt_evolve_lpost_cut.argtypes = [C.POINTER(_PT), C.c_double]
    L.ptmo_chunk_prefix.restype = C.c_double
    L.ptmo_chunk_prefix.argtypes = [_dp, C.c_int, _dp]
    L.ptmo_pt_set_states.argtypes = [C.POINTER(_PT), C.POINTER(_Problem), _dp, _dp]
    L.ptmo_mh_step.argtypes = [C.POINTER(_PT), C.POINTER(_Problem), C.POINTER(_Proposal), C.c_void_p, C.c_int, C.c_int]
    L.ptmo_pt_step.argtypes = [C.POINTER(_PT), C.POINTER(_Problem), C.POINTER(_Proposal), C.c_void_p, C.c_int]
    L.ptmo_sweep.argtypes = [C.POINTER(_PT), C.POINTER(_Problem), C.POINTER(_Proposal), C.c_void_p, C.c_int]
    L.ptmo_exchange_phase.argtypes = [C.POINTER(_PT), C.c_void_p]
    L.ptmo_rng_philox.restype = C.c_void_p
    L.ptmo_rng_philox.argtypes = [C.c_uint64, C.c_int]
    L.ptmo_rng_tape.restype = C.c_void_p
    L.ptmo_rng_tape.argtypes = [C.c_int, C.c_int, C.c_int, _dp, C.c_int, _dp, C.c_int, _dp, C.c_int]
    L.ptmo_rng_free.argtypes = [C.c_void_p]
    L.ptmo_rng_tape_hastings.argtypes = [C.c_void_p, _dp, C.POINTER(C.c_int32)]
    L.ptmo_pt_set_host_proposal.argtypes = [C.POINTER(_PT), C.c_void_p, C.c_void_p]
    L.ptmo_pt_set_de.argtypes = [C.POINTER(_PT), C.POINTER(_DeParams), C.c_int, _dp]
    L.ptmo_de_draw.argtypes = [C.c_int, _dp, C.c_long, C.POINTER(_DeParams), DE_UNIFORM_FN, C.c_void_p, DE_ROW_FN, C.c_void_p, _dp, _dp]
    L.ptmo_de_ready.argtypes = [C.c_int, C.c_long]
    L.ptmo_init_from_prior.argtypes = [C.POINTER(_PT), C.POINTER(_Problem), C.c_uint64]
    L.ptmo_selection_run_census.argtypes = [C.c_uint64, C.c_int, C.c_double, C.c_int, C.c_uint64, C.c_int, _ip, C.c_int, C.c_int,
                                            C.POINTER(C.c_int64), C.c_int]
    _lib = L
    return L


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


LOGLIKE_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, _dp, C.c_int)
_i32p = C.POINTER(C.c_int32)
# the engine's ptm_propose_batch_fn (include/ptm_engine.h) == the oracle's ptmo_propose_fn
PROPOSE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, _dp, _i32p, _i32p, C.c_uint64, _dp, _dp, _i32p, _i32p)


def de_draw(x, history, uniforms, snooker, gamma_one_frac, reduce_gamma, ignore_frac):
    """one differential_evolution::draw of the oracle (ptmo_de_draw) for the state x of a chain with the saved history `history`
    [rows][D], fed the uniforms of a tape in sequence: (type, proposed state, log-Hastings ratio, uniforms used)"""
    x = np.ascontiguousarray(x, dtype=np.float64)
    H = np.ascontiguousarray(history, dtype=np.float64)
    D = x.size
    used = [0]

    def uni(ctx, slot):
        v = uniforms[used[0]]
        used[0] += 1
        return v

    def row(ctx, r):
        return H[r].ctypes.data

    q = _DeParams(snooker, gamma_one_frac, reduce_gamma, ignore_frac)
    xn = np.zeros(D)
    lh = C.c_double(0.0)
    cu, cr = DE_UNIFORM_FN(uni), DE_ROW_FN(row)
    t = lib().ptmo_de_draw(D, _d(x), H.shape[0], C.byref(q), cu, None, cr, None, _d(xn), C.byref(lh))
    return t, xn, lh.value, used[0]


def make_propose_fn(pyfunc):
    """pyfunc(X_cur[n][dim], rung[n], walker[n], step) -> (X_prop[n][dim], log_hastings[n], type[n], valid[n]); returns the
    C callback both the oracle and the engine take (keep a reference to it while it is registered)"""
    def tramp(user, n, dim, xc, rung, walker, step, xp, lh, ty, va):
        X = np.ctypeslib.as_array(xc, shape=(n, dim)).copy()
        r = np.ctypeslib.as_array(rung, shape=(n,)).copy()
        w = np.ctypeslib.as_array(walker, shape=(n,)).copy()
        P, H, T, V = pyfunc(X, r, w, int(step))
        np.ctypeslib.as_array(xp, shape=(n, dim))[:] = P
        np.ctypeslib.as_array(lh, shape=(n,))[:] = H
        np.ctypeslib.as_array(ty, shape=(n,))[:] = T
        np.ctypeslib.as_array(va, shape=(n,))[:] = V
    return PROPOSE_FN(tramp)


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().ptmo_philox4x32_10(c, k, o)
    return [int(v) for v in o]


def draw_block(seed, tag, stream, step, block):
    o = (C.c_uint32 * 4)()
    lib().ptmo_draw_block(seed, tag, stream, step, block, o)
    return [int(v) for v in o]


def bm_neg2log(k):
    return lib().ptmo_bm_neg2log(k)


def boxmuller(k1, k2):
    a, b = C.c_double(), C.c_double()
    lib().ptmo_boxmuller(k1, k2, C.byref(a), C.byref(b))
    return a.value, b.value


def boundary_enforce(lo, hi, xmin, xmax, x):
    v = C.c_double(x)
    ok = lib().ptmo_boundary_enforce(lo, hi, xmin, xmax, C.byref(v))
    return ok, v.value


def ladder(nt, tmax):
    b = np.zeros(nt)
    lib().ptmo_ladder(nt, tmax, _d(b))
    return b


def selection_run_census(seed, Nt, swap_rate, W, nsteps, bounds, Lmax=12, step0=0, nthreads=8):
    """hist[boundary][L] of the runs of consecutive surviving exchange picks that start at rung b-1 (see ptm_oracle.c)"""
    b = np.ascontiguousarray(bounds, dtype=np.int32)
    h = np.zeros((len(b), Lmax + 1), dtype=np.int64)
    lib().ptmo_selection_run_census(seed, Nt, swap_rate, W, step0, nsteps, _i(b), len(b), Lmax,
                                    h.ctypes.data_as(C.POINTER(C.c_int64)), nthreads)
    return h


class Problem:
    """Target + prior + boundaries (what bayes_likelihood::basic_setup describes)."""

    def __init__(self, D, min_prior=-30.0):
        self.D = D
        self.p = lib().ptmo_problem_create(D)
        self.p.contents.minPrior = min_prior
        self._keep = []

    def set_bounds(self, lo, hi, xmin, xmax):
        lo = np.ascontiguousarray(lo, dtype=np.int32)
        hi = np.ascontiguousarray(hi, dtype=np.int32)
        xmin = np.ascontiguousarray(xmin, dtype=np.float64)
        xmax = np.ascontiguousarray(xmax, dtype=np.float64)
        lib().ptmo_problem_set_bounds(self.p, _i(lo), _i(hi), _d(xmin), _d(xmax))

    def set_prior(self, types, centers, halfwidths):
        t = np.ascontiguousarray([TYPE_NAMES[v] if isinstance(v, str) else int(v) for v in types], dtype=np.int32)
        c = np.ascontiguousarray(centers, dtype=np.float64)
        h = np.ascontiguousarray(halfwidths, dtype=np.float64)
        lib().ptmo_problem_set_prior(self.p, _i(t), _d(c), _d(h))

    def set_gauss(self, P, like0, mean=None):
        P = np.ascontiguousarray(P, dtype=np.float64).reshape(self.D, self.D)
        m = None if mean is None else np.ascontiguousarray(mean, dtype=np.float64)
        lib().ptmo_problem_set_gauss(self.p, None if m is None else _d(m), _d(P), like0)

    def set_user(self, pyfunc):
        def tramp(user, xp, dim):
            return float(pyfunc(np.ctypeslib.as_array(xp, shape=(dim,)).copy()))
        cb = LOGLIKE_FN(tramp)
        self._keep.append(cb)
        lib().ptmo_problem_set_user(self.p, C.cast(cb, C.c_void_p), None)

    def set_user_prior(self, pyfunc):
        """a prior handed over as a function of the (valid) state: log-prior, -inf outside the support"""
        def tramp(user, xp, dim):
            return float(pyfunc(np.ctypeslib.as_array(xp, shape=(dim,)).copy()))
        cb = LOGLIKE_FN(tramp)
        self._keep.append(cb)
        lib().ptmo_problem_set_user_prior(self.p, C.cast(cb, C.c_void_p), None)

    def enforce(self, x):
        x = np.array(x, dtype=np.float64)
        ok = lib().ptmo_enforce(self.p, _d(x))
        return ok, x

    def lprior(self, x, valid=1):
        x = np.ascontiguousarray(x, dtype=np.float64)
        return lib().ptmo_lprior(self.p, _d(x), valid)

    def llike(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        return lib().ptmo_llike(self.p, _d(x))

    @property
    def origin_valid(self):
        return self.p.contents.origin_valid

    def __del__(self):
        try:
            lib().ptmo_problem_free(self.p)
        except Exception:
            pass


class Ladder:
    """W independent ladders of Nt rungs (chain (w,r) at index w*Nt+r): the oracle's PT state."""

    def __init__(self, problem, beta, W=1, swap_rate=0.1, add_every_N=1):
        self.pb = problem
        self.beta = np.ascontiguousarray(beta, dtype=np.float64)
        self.Nt, self.W, self.D = len(self.beta), W, problem.D
        self.s = lib().ptmo_pt_create(self.D, self.Nt, W, _d(self.beta), swap_rate, add_every_N)
        self.N = self.Nt * W
        self._props = None
        self._keep = []
        self.rng = None

    # proposals: list (len Nt) of (kind, M, oneDfrac)
    def set_proposals(self, props):
        arr = (_Proposal * self.Nt)()
        self._keep = []
        self._prop_specs = [tuple(p) for p in props]
        for r, (kind, M, f) in enumerate(props):
            M = np.ascontiguousarray(M, dtype=np.float64)
            self._keep.append(M)
            arr[r].kind, arr[r].M, arr[r].oneDfrac = kind, _d(M), f
            arr[r].K = 0
        self._props = arr

    def set_mixture(self, cum_shares, scales, one_d_fracs):
        """scale mixture on top of set_proposals: arrays [Nt][K] (cumulative shares, scales, oneDfracs)"""
        cs, sc, od = (np.asarray(a, dtype=np.float64) for a in (cum_shares, scales, one_d_fracs))
        K = cs.shape[1]
        for r in range(self.Nt):
            m = np.ascontiguousarray(np.stack([cs[r], sc[r], od[r]], axis=1).ravel())
            self._keep.append(m)
            self._props[r].K = K
            self._props[r].mix = _d(m)

    def set_de(self, snooker=0.1, gamma_one_frac=0.3, reduce_gamma=4.0, ignore_frac=0.0, init_rows=None):
        """differential evolution as the member of negative scale of the proposal sets (set_mixture); init_rows [n_extra][N][D] in the
        oracle's chain order: what MH_chain::initialize(n) saved in front of the start state"""
        q = _DeParams(snooker, gamma_one_frac, reduce_gamma, ignore_frac)
        ir = None if init_rows is None else np.ascontiguousarray(init_rows, dtype=np.float64).reshape(-1, self.N, self.D)
        lib().ptmo_pt_set_de(self.s, C.byref(q), 0 if ir is None else ir.shape[0], None if ir is None else _d(ir))

    def use_philox(self, seed):
        self.rng = lib().ptmo_rng_philox(seed, self.Nt)

    def use_tape(self, chain_tapes, pt_tapes, deltas):
        ct = np.ascontiguousarray(chain_tapes, dtype=np.float64).reshape(self.N, -1)
        pt = np.ascontiguousarray(pt_tapes, dtype=np.float64).reshape(self.W, -1)
        dl = np.ascontiguousarray(deltas, dtype=np.float64).reshape(self.N, -1, self.D)
        self._keep += [ct, pt, dl]
        self.rng = lib().ptmo_rng_tape(self.W, self.Nt, self.D, _d(ct), ct.shape[1], _d(pt), pt.shape[1], _d(dl), dl.shape[1])

    def tape_hastings(self, log_hastings, types=None):
        """scripted log-Hastings ratios / type codes of the tape's offsets, [N][nsteps]"""
        h = np.ascontiguousarray(log_hastings, dtype=np.float64).reshape(self.N, -1)
        t = None if types is None else np.ascontiguousarray(types, dtype=np.int32).reshape(self.N, -1)
        self._keep += [h, t]
        lib().ptmo_rng_tape_hastings(self.rng, _d(h), None if t is None else t.ctypes.data_as(_i32p))

    def set_host_proposal(self, cfn):
        """cfn: a PROPOSE_FN (make_propose_fn) -- replaces the proposals of set_proposals in every MH step"""
        self._keep.append(cfn)
        lib().ptmo_pt_set_host_proposal(self.s, C.cast(cfn, C.c_void_p), None)

    @property
    def last_accept_mh(self):
        return self._arr(self.s.contents.last_accept_mh, (self.N,), np.int64)

    def set_states(self, x, llike=None):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(self.N, self.D)
        ll = None if llike is None else np.ascontiguousarray(llike, dtype=np.float64)
        lib().ptmo_pt_set_states(self.s, self.pb.p, _d(x), None if ll is None else _d(ll))

    def init_from_prior(self, seed):
        lib().ptmo_init_from_prior(self.s, self.pb.p, seed)

    def pt_step(self, n=1, nthreads=1):
        for _ in range(n):
            lib().ptmo_pt_step(self.s, self.pb.p, self._props, self.rng, nthreads)

    def sweep(self, n=1, nthreads=1):
        for _ in range(n):
            lib().ptmo_sweep(self.s, self.pb.p, self._props, self.rng, nthreads)

    def _arr(self, ptr, shape, dtype):
        return np.ctypeslib.as_array(ptr, shape=shape).astype(dtype, copy=True)

    @property
    def x(self):
        return self._arr(self.s.contents.x, (self.N, self.D), np.float64)

    @property
    def llike(self):
        return self._arr(self.s.contents.llike, (self.N,), np.float64)

    @property
    def lprior(self):
        return self._arr(self.s.contents.lprior, (self.N,), np.float64)

    def evolve_temps(self, rate, lpost_cut=-1.0):
        lib().ptmo_pt_evolve_temps(self.s, float(rate))
        lib().ptmo_pt_evolve_lpost_cut(self.s, float(lpost_cut))

    @property
    def betaw(self):
        """[W][Nt] inverse temperatures (per walker once the ladders evolve)"""
        if not self.s.contents.betaw:
            return np.tile(self.beta, (self.W, 1))
        return self._arr(self.s.contents.betaw, (self.W, self.Nt), np.float64)

    @property
    def lpost(self):
        b = self.betaw.ravel()
        ll, lp = self.llike, self.lprior
        return np.array([lib().ptmo_lpost(lp[c], b[c], ll[c]) for c in range(self.N)])

    @property
    def ntries(self):
        return self._arr(self.s.contents.ntries, (self.N,), np.int64)

    @property
    def naccept(self):
        return self._arr(self.s.contents.naccept, (self.N,), np.int64)

    @property
    def last_type(self):
        return self._arr(self.s.contents.last_type, (self.N,), np.int64)

    @property
    def map_lpost(self):
        return self._arr(self.s.contents.map_lpost, (self.N,), np.float64)

    @property
    def map_x(self):
        return self._arr(self.s.contents.map_x, (self.N, self.D), np.float64)

    def enable_history(self, rows_per_chain):
        lib().ptmo_pt_enable_history(self.s, rows_per_chain)

    def history(self):
        """dict of arrays [N][cap](,D): x, llike, lprior, naccept, ntries, last_type (oracle chain order w*Nt + r)"""
        c, cap = self.s.contents, self.s.contents.hist_cap
        return dict(invtemp=self._arr(c.hist_beta, (self.N, cap), np.float64),
                    x=self._arr(c.hist_x, (self.N, cap, self.D), np.float64), llike=self._arr(c.hist_ll, (self.N, cap), np.float64),
                    lprior=self._arr(c.hist_lp, (self.N, cap), np.float64), naccept=self._arr(c.hist_nacc, (self.N, cap), np.int64),
                    ntries=self._arr(c.hist_ntry, (self.N, cap), np.int64), last_type=self._arr(c.hist_type, (self.N, cap), np.int64))

    @property
    def nhist(self):
        return self._arr(self.s.contents.nhist, (self.N,), np.int64)

    @property
    def nsize(self):
        return self._arr(self.s.contents.nsize, (self.N,), np.int64)

    @property
    def swap_count(self):
        return self._arr(self.s.contents.swap_count, (self.W, max(self.Nt - 1, 1)), np.int64)

    @property
    def swap_accept_count(self):
        return self._arr(self.s.contents.swap_accept_count, (self.W, max(self.Nt - 1, 1)), np.int64)

    @property
    def last_pairs(self):
        ms = self.s.contents.maxswaps
        return self._arr(self.s.contents.last_pairs, (self.W, ms), np.int64)

    @property
    def last_accept(self):
        ms = self.s.contents.maxswaps
        return self._arr(self.s.contents.last_accept, (self.W, ms), np.int64)

    @property
    def step(self):
        return int(self.s.contents.step)

    def __del__(self):
        try:
            if self.rng:
                lib().ptmo_rng_free(self.rng)
            lib().ptmo_pt_free(self.s)
        except Exception:
            pass
