"""Helpers shared by the GPU parity tests: build the oracle twin of an engine configuration."""
import numpy as np

import oracle_lib as O
from ptmcmc_amd import engine as E
from ptmcmc_amd.problems import GaussianProblem

KIND_TO_ORACLE = {E.PROP_DENSE: O.PROP_DENSE, E.PROP_LOWER: O.PROP_DENSE, E.PROP_DIAG: O.PROP_DIAG}


def to_oracle_order(a, Nt, W):
    """engine order (rung-major: c = r*W + w)  ->  oracle order (walker-major: c = w*Nt + r)"""
    a = np.asarray(a)
    return a.reshape((Nt, W) + a.shape[1:]).swapaxes(0, 1).reshape((Nt * W,) + a.shape[1:])


def to_engine_order(a, Nt, W):
    a = np.asarray(a)
    return a.reshape((W, Nt) + a.shape[1:]).swapaxes(0, 1).reshape((Nt * W,) + a.shape[1:])


def oracle_problem(pr, bounds=None, prior=None, mean=None, min_prior=-30.0):
    D = pr.D
    pb = O.Problem(D, min_prior=min_prior)
    if bounds is None:
        pb.set_bounds([0] * D, [0] * D, [0.0] * D, [0.0] * D)
    else:
        pb.set_bounds(*bounds)
    if prior is None:
        pb.set_prior(pr.types, pr.centers, pr.halfwidths)
    else:
        pb.set_prior(*prior)
    pb.set_gauss(pr.P, pr.like0, mean)
    return pb


def problem_for(D, Nt, tmax):
    """The synthetic Gaussian problem of the parity tests.  Up to 128 dimensions the reference example's box (100 sigma,
    exampleGaussian.py:71-76).  Beyond, the PRODUCT of such uniform densities underflows to 0 -- log-prior -inf, every move
    accepted, as the reference would do (probability_function.cc:281-304 multiplies, then takes the log) -- so the
    high-dimensional cases use a small target (sigma ~ 0.01-0.04) in a box of 20 sigma, whose densities are of order one."""
    if D <= 128:
        return GaussianProblem(D, Nt, tmax)
    return GaussianProblem(D, Nt, tmax, prior_scale=20.0, cov_scale=1e-3)


def make_pair(D, Nt, W, tmax, kind=E.PROP_LOWER, seed=0x5EED0001, swap_rate=0.1, one_d_frac=None, add_every_n=1,
              bounds=None, prior=None, mean=None, min_prior=-30.0, init="prior", x0=None, history_cap=0):
    """An engine and its oracle twin on the same synthetic Gaussian problem and the same start states (history_cap > 0: both record
    every rung's history -- every add_every_n-th add -- and MAP)."""
    pr = problem_for(D, Nt, tmax)
    eng = E.Engine(D, Nt, W, seed=seed, swap_rate=swap_rate, add_every_n=add_every_n, min_prior=min_prior,
                   history_rungs=Nt if history_cap else 0, history_capacity=history_cap, map_rungs=Nt if history_cap else 0)
    odf = None if one_d_frac is None else np.full(Nt, one_d_frac)
    fac = pr.configure(eng, kind, odf)
    if bounds is not None:
        eng.set_bounds(*bounds)
    if prior is not None:
        eng.set_prior(*prior)
    if mean is not None:
        eng.set_target_gaussian(pr.P, pr.like0, mean)
    pb = oracle_problem(pr, bounds, prior, mean, min_prior)
    lad = O.Ladder(pb, pr.beta, W=W, swap_rate=swap_rate, add_every_N=add_every_n)
    f = 0.0 if one_d_frac is None else one_d_frac
    lad.set_proposals([(KIND_TO_ORACLE[kind], fac[r], f) for r in range(Nt)])
    lad.use_philox(seed)
    if history_cap:
        lad.enable_history(history_cap)
    if x0 is None:
        eng.init_from_prior()
        x0 = eng.states()
    else:
        eng.set_states(x0)
    lad.set_states(to_oracle_order(x0, Nt, W))
    return pr, eng, lad


def assert_same_state(eng, lad, what=""):
    Nt, W = eng.Nt, eng.W
    xe = eng.states()
    xo = to_engine_order(lad.x, Nt, W)
    assert np.array_equal(xe, xo), "%s states differ: max|d|=%g at %s" % (
        what, np.nanmax(np.abs(xe - xo)), np.argwhere(xe != xo)[:4].tolist())
    for name in ("llike", "lprior", "ntries", "naccept", "last_type", "nhist", "nsize"):
        a = getattr(eng, name)
        b = to_engine_order(getattr(lad, name), Nt, W)
        assert np.array_equal(a, b, equal_nan=True) if a.dtype.kind == "f" else np.array_equal(a, b), \
            "%s %s differ at %s" % (what, name, np.argwhere(a != b)[:4].tolist())


def assert_same_history_and_map(eng, lad, cap):
    """every saved row (states, scalars, counters, the temperature it was saved at) and every rung's MAP; the run must fit the ring"""
    Nt, W = eng.Nt, eng.W
    he, ho = eng.history(), lad.history()
    nsize = eng.nsize
    assert nsize.max() <= cap
    for name in ("x", "llike", "lprior", "naccept", "ntries", "last_type", "invtemp"):
        for s_ in range(int(nsize.max())):
            have = nsize > s_
            got, want = he[name][s_ % cap][have], to_engine_order(ho[name][:, s_], Nt, W)[have]
            assert np.array_equal(got, want), (name, s_, np.argwhere(got != want)[:3].tolist())
    m = eng.map()
    assert np.array_equal(m["lpost"], to_engine_order(lad.map_lpost, Nt, W))
    assert np.array_equal(m["x"], to_engine_order(lad.map_x, Nt, W))
