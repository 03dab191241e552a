"""Host-side proposals (ptm_set_proposal_callback): any proposal_distribution::draw / log_hastings_ratio / type /
accept / reject (proposal_distribution.hh:65-87) evaluated by the caller, the rest of MH_chain::step (chain.cc:976-1018)
on the device.  The engine and the oracle are handed THE SAME scripted C callback -- deterministic in (rung, walker, step),
with non-zero log-Hastings ratios, NaN ratios, invalid states, jumps out of the prior and varying type codes -- and must
produce the same chains bit for bit.  (The oracle's handling of log-Hastings ratios and types is pinned against the real
reference by golden trace 10, tests/test_oracle_golden.py.)"""
import numpy as np
import pytest

import oracle_lib as O
import parity_util as PU
from ptmcmc_amd import engine as E
from ptmcmc_amd.problems import GaussianProblem

pytestmark = pytest.mark.gpu
M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(z):
    """splitmix64 finaliser on uint64 arrays (exact integer arithmetic: the same numbers whatever the batch)"""
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _uniforms(rung, walker, step, k):
    """u[n][k] in [0, 1), a function of (rung, walker, step, column) only"""
    with np.errstate(over="ignore"):
        base = (rung.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) + walker.astype(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F)
                + np.uint64(step) * np.uint64(0x165667B19E3779F9))
        cols = np.arange(1, k + 1, dtype=np.uint64) * np.uint64(0xD6E8FEB86659FD93)
        z = _mix(base[:, None] + cols[None, :])
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def scripted_proposal(scale_of_rung):
    def propose(X, rung, walker, step):
        n, D = X.shape
        u = _uniforms(rung, walker, step, D + 4)
        P = X + scale_of_rung[rung][:, None] * (2 * u[:, :D] - 1)
        big = u[:, D] < 0.04                               # now and then far out of the prior's support
        P[big] = P[big] * 40.0 + 7.0
        H = 1.6 * (u[:, D + 1] - 0.5)                      # a log-Hastings ratio of either sign ...
        H[u[:, D + 2] < 0.05] = 0.0
        H[u[:, D + 2] > 0.97] = np.nan                     # ... and the reference's NaN rule (chain.cc:990-993)
        T = (u[:, D + 3] * 5).astype(np.int32)             # proposal_distribution::type()
        V = (u[:, D + 3] > 0.03).astype(np.int32)          # state::invalid()
        return P, H, T, V
    return propose


CASES = [
    # D, Nt, W, general state space?, callback likelihood?, evolve, add_every_n
    (3, 6, 1, False, False, 0.0, 1),
    (6, 8, 4, True, False, 0.0, 2),
    (32, 5, 64, False, False, 0.0, 1),      # a population the MFMA kernel would take: host-side proposals keep the lanes kernel
    (40, 4, 2, False, False, 0.0, 1),       # 33..64 dimensions
    (5, 7, 3, True, True, 0.0, 1),          # ... around a host-callback likelihood: propose -> enforce/prior -> host llike -> accept
    (12, 6, 2, False, True, 0.02, 1),
    (16, 9, 5, True, False, 0.03, 3),       # evolving ladder
]


@pytest.mark.parametrize("D,Nt,W,general,cb,ev,N", CASES)
def test_scripted_host_proposals_bit_exact(D, Nt, W, general, cb, ev, N):
    sr, cap, seed = 0.35, 96, 0x5EED0001
    pr = GaussianProblem(D, Nt, 1e3)
    bounds = prior = mean = None
    if general:
        blo, bhi, bmin, bmax = [0] * D, [0] * D, [0.0] * D, [0.0] * D
        blo[0], bhi[0], bmin[0], bmax[0] = 3, 3, -3.0, 2.5                    # wrap
        blo[1], bhi[1], bmin[1], bmax[1] = 2, 2, -4.0, 4.0                    # reflect
        blo[2], bhi[2], bmin[2], bmax[2] = 1, 1, -6.0, 6.0                    # limit
        bounds = (blo, bhi, bmin, bmax)
        types, cen, hw = [1] * D, [0.0] * D, [8.0] * D
        types[1], cen[1], hw[1] = 2, 0.1, 2.0                                  # gaussian
        prior = (types, cen, hw)
        mean = np.linspace(-0.3, 0.3, D)
    scale = 2.0 / np.sqrt(D) / np.sqrt(np.maximum(pr.beta, 0.02)) * np.sqrt(np.diag(pr.cov).mean())
    cfn = O.make_propose_fn(scripted_proposal(scale))
    eng = E.Engine(D, Nt, W, seed=seed, swap_rate=sr, add_every_n=N, history_rungs=Nt, history_capacity=cap, map_rungs=Nt)
    pr.configure(eng, E.PROP_DIAG)      # (device proposals set first: the callback must replace them)
    if bounds: eng.set_bounds(*bounds)
    if prior: eng.set_prior(*prior)
    if mean is not None: eng.set_target_gaussian(pr.P, pr.like0, mean)
    pb = PU.oracle_problem(pr, bounds, prior, mean)
    if cb:
        sc = np.linspace(0.7, 1.4, D)
        loglike = lambda x: float(-0.5 * np.sum((np.asarray(x) * sc) ** 2))
        eng.set_target_callback(loglike)
        pb.set_user(loglike)
    results = []
    eng.set_proposal_callback(cfn, result=lambda r, w, a: results.append((r, w, a)))
    assert eng.sweep_kernel_name.startswith("sweep_lanes_kernel<%d" % (4 if D <= 4 else 8 if D <= 8 else 16 if D <= 16 else 32 if D <= 32 else 64))
    rng = np.random.default_rng(5)
    x0 = rng.uniform(-1.0, 1.0, size=(Nt * W, D)) * np.sqrt(np.diag(pr.cov))
    eng.set_states(x0)
    lad = O.Ladder(pb, pr.beta, W=W, swap_rate=sr, add_every_N=N)
    lad.set_proposals([(O.PROP_DIAG, np.ones(D), 0.0)] * Nt)       # unused
    lad.use_philox(seed)
    lad.enable_history(cap)
    lad.set_host_proposal(cfn)
    lad.set_states(PU.to_oracle_order(x0, Nt, W))
    if ev:
        eng.set_evolve_temps(ev); lad.evolve_temps(ev)
    PU.assert_same_state(eng, lad, "start")
    nsteps = 30
    for k in range(nsteps):
        results.clear()
        eng.step(1); eng.sync(); lad.pt_step(1)
        PU.assert_same_state(eng, lad, "after step %d" % (k + 1))
        # accept() / reject() notifications: one per moving chain, the oracle's outcomes
        assert len(results) == 1
        r, w, a = results[0]
        want = PU.to_engine_order(lad.last_accept_mh, Nt, W)
        moving = np.flatnonzero(want != 2)
        assert np.array_equal(r * W + w, moving) and np.array_equal(a, want[moving])
    if ev:
        assert np.array_equal(eng.invtemps(), lad.betaw)
    acc = eng.naccept.sum() - eng.Nc
    assert 10 < acc < 0.95 * eng.Nc * nsteps, acc
    assert len(set(he_types := eng.history()["last_type"].ravel().tolist())) >= 5   # every type code 0..4 (and -1: the initial rows)
    he, ho = eng.history(), lad.history()
    nsize = eng.nsize
    for name in ("x", "llike", "lprior", "naccept", "ntries", "last_type", "invtemp"):
        for s_ in range(int(nsize.max())):
            have = nsize > s_
            got, want = he[name][s_ % cap][have], PU.to_engine_order(ho[name][:, s_], Nt, W)[have]
            assert np.array_equal(got, want), (name, s_)
    m = eng.map()
    assert np.array_equal(m["lpost"], PU.to_engine_order(lad.map_lpost, Nt, W))
    assert np.array_equal(m["x"], PU.to_engine_order(lad.map_x, Nt, W))
    # back to the device proposals
    eng.set_proposal_callback(None)
    assert not eng.sweep_kernel_name.startswith("sweep_lanes_kernel") or W % 64 != 0 or D > 32
    eng.step(2); eng.sync()
    eng.close()


def test_host_proposals_refuse_partial_sweeps():
    pr = GaussianProblem(4, 8, 1e2)
    eng = E.Engine(4, 8, 2, rung_begin=0, rung_count=4)
    pr.configure(eng, E.PROP_DIAG)
    eng.init_from_prior()
    eng.set_proposal_callback(lambda X, r, w, s: (X, np.zeros(len(X)), np.zeros(len(X), dtype=np.int32), np.ones(len(X), dtype=np.int32)))
    with pytest.raises(E.PtmError, match="partial sweeps"):
        eng.sweep_rungs(0, 2, True)
    eng.close()


def test_host_proposals_name_the_global_walker_in_a_population_split():
    """ptm_propose_batch_fn / ptm_proposal_result_fn are handed each chain's GLOBAL rung and walker (include/ptm_engine.h): a
    population split by walkers (ptm_config.walker_begin > 0) with a proposal keyed by (rung, walker, step) then walks the very
    chains of one engine holding every ladder."""
    D, Nt, W, sr, seed = 6, 7, 5, 0.3, 0x5EED0001
    pr = GaussianProblem(D, Nt, 1e3)
    scale = 2.0 / np.sqrt(D) / np.sqrt(np.maximum(pr.beta, 0.02)) * np.sqrt(np.diag(pr.cov).mean())
    rng = np.random.default_rng(11)
    x0 = (rng.uniform(-1.0, 1.0, size=(Nt * W, D)) * np.sqrt(np.diag(pr.cov))).reshape(Nt, W, D)

    def run(w0, n):
        seen = []
        eng = E.Engine(D, Nt, n, seed=seed, swap_rate=sr, walker_begin=w0)
        pr.configure(eng, E.PROP_DIAG)
        fn = scripted_proposal(scale)

        def propose(X, rung, walker, step):
            seen.append(walker.copy())
            return fn(X, rung, walker, step)
        eng.set_proposal_callback(propose, result=lambda r, w, a: seen.append(w.copy()))
        eng.set_states(x0[:, w0:w0 + n].reshape(Nt * n, D))
        eng.step(12); eng.sync()
        out = eng.states().reshape(Nt, n, D), eng.llike.reshape(Nt, n), eng.naccept.reshape(Nt, n)
        eng.close()
        allw = np.concatenate(seen)
        assert allw.min() >= w0 and allw.max() < w0 + n and (n == 1 or allw.max() > w0)
        return out

    whole = run(0, W)
    lo, hi = run(0, 3), run(3, 2)
    for k in range(3):
        assert np.array_equal(whole[k][:, :3], lo[k]) and np.array_equal(whole[k][:, 3:], hi[k]), k
