"""Differential evolution drawn ON THE DEVICE (ptm_set_proposal_de) against the CPU oracle's restatement of
differential_evolution::draw (ptmo_de_draw; pinned against the real reference draw by draw in
tests/test_oracle_golden.py::test_oracle_differential_evolution_matches_the_reference_draw_by_draw).  Bit for bit."""
import numpy as np
import pytest

import oracle_lib as O
import parity_util as PU
from ptmcmc_amd import engine as E

pytestmark = pytest.mark.gpu


def _recipe(Nt, K, de_share, odf):
    """the reference sampler's default set (ptmcmc.cc:60-143): differential evolution first, then K Gaussians of doubling shares"""
    g = 2.0 ** np.arange(1, K + 1)
    shares = np.concatenate([[de_share], (1 - de_share) * g / g.sum()])
    cum = np.tile(np.cumsum(shares), (Nt, 1)); cum[:, -1] = 1.0
    scales = np.tile(np.concatenate([[-1.0], 2.0 ** -np.arange(K)[::-1]]), (Nt, 1))
    odfs = np.tile(np.concatenate([[0.0], np.full(K, odf)]), (Nt, 1))
    return cum, scales, odfs


DE_CASES = [
    # D, Nt, W, kind, add_every_N, snooker, n_init_extra (in units of D), K Gaussians, steps
    (6, 20, 1, E.PROP_DIAG, 1, 0.1, 50, 6, 60),      # the LISA example's shape with the sampler's defaults (de_ni = 50 per dimension, six Gaussians)
    (6, 12, 3, E.PROP_DIAG, 1, 0.3, 0, 2, 90),       # no initial rows: the member is passed over until 10 D rows are saved, then drawn
    (16, 8, 64, E.PROP_LOWER, 2, 0.5, 10, 3, 40),    # whole waves per rung, every second add saved
    (32, 6, 2, E.PROP_DENSE, 1, 0.2, 10, 1, 30),     # 32 dimensions (rows in the matrix cores' layout)
    (3, 5, 7, E.PROP_DIAG, 3, 1.0, 40, 1, 60),       # snooker moves only
    (20, 4, 5, E.PROP_LOWER, 1, 0.0, 12, 2, 30),     # parallel moves only, padded dimensions
    (32, 2, 64, E.PROP_DENSE, 1, 0.3, 10, 2, 30),    # 32 dimensions, whole waves per rung (a lane per dimension all the same: no matrix cores with differential evolution)
    (7, 3, 64, E.PROP_LOWER, 2, 0.3, 12, 2, 40),     # a lane per chain
    (20, 40, 64, E.PROP_LOWER, 1, 0.2, 10, 2, 24),   # whole waves per rung, too many workgroups for the persistent kernel: exchange kernel + lanes kernel
    (3, 3, 128, E.PROP_DIAG, 1, 1.0, 40, 1, 40),     # snooker moves only, a lane per chain
    (48, 4, 3, E.PROP_LOWER, 1, 0.3, 10, 2, 30),     # 33..64 dimensions: a wave per chain
    (100, 3, 2, E.PROP_DENSE, 2, 0.4, 10, 1, 24),    # 65..128 dimensions: two dimensions per lane
    (12, 24, 4, E.PROP_DIAG, 1, 0.2, 10, 6, 40),     # four chains per wave: snooker and parallel moves and Gaussians side by side in a wave
]


def _pair(D, Nt, W, kind, N, snooker, ninit, K, cap, de_share=0.7, ignore=0.0, seed=0x5EED0001):
    pr = PU.problem_for(D, Nt, 1e3)
    eng = E.Engine(D, Nt, W, seed=seed, swap_rate=0.3, add_every_n=N, history_rungs=Nt, history_capacity=cap, map_rungs=Nt)
    fac = pr.configure(eng, kind)
    eng.init_from_prior()
    x0 = eng.states()
    lad = O.Ladder(PU.oracle_problem(pr), pr.beta, W=W, swap_rate=0.3, add_every_N=N)
    lad.set_proposals([(PU.KIND_TO_ORACLE[kind], fac[r], 0.0) for r in range(Nt)])
    lad.use_philox(seed)
    lad.enable_history(cap)
    lad.set_states(PU.to_oracle_order(x0, Nt, W))
    cum, scales, odfs = _recipe(Nt, K, de_share, 0.5)
    eng.set_proposal_mixture(cum, scales, odfs); lad.set_mixture(cum, scales, odfs)
    rng = np.random.default_rng(D * 1000 + Nt)
    init = None
    if ninit:
        init = rng.uniform(-1.0, 1.0, size=(ninit * D, Nt * W, D)) * np.asarray(pr.halfwidths)[None, None, :] * 0.02
    eng.set_proposal_de(snooker, 0.3, 4.0, ignore, init_rows=init)
    lad.set_de(snooker, 0.3, 4.0, ignore, init_rows=None if init is None else np.stack([PU.to_oracle_order(init[k], Nt, W) for k in range(init.shape[0])]))
    return pr, eng, lad


@pytest.mark.parametrize("D,Nt,W,kind,N,snooker,ninit,K,steps", DE_CASES)
def test_differential_evolution_on_the_device_matches_the_oracle(D, Nt, W, kind, N, snooker, ninit, K, steps):
    """A proposal set of differential evolution and K Gaussians (the reference sampler's default recipe, ptmcmc.cc:60-143) drawn
    entirely on the device: member choice, the passing-over of a differential evolution that is not ready, row picks from the
    initial rows and from the history ring, parallel and snooker moves, the snooker move's log-Hastings ratio in the Metropolis test,
    type codes -- states, counters, every saved row and every rung's MAP bit for bit the oracle's."""
    cap = 2 * steps + 8
    pr, eng, lad = _pair(D, Nt, W, kind, N, snooker, ninit, K, cap)
    # whole waves per rung of up to 8 dimensions (and big populations): the general kernel draws a chain's move on the chain's lane;
    # everything else: a lane per dimension
    assert eng.sweep_kernel_name.startswith("sweep_kernel<" if (W % 64 == 0 and D <= 8) else "sweep_lanes_kernel<"), eng.sweep_kernel_name
    # ... and 9..32 dimensions of them step in the persistent ladder kernel's build with differential evolution (FL = 11)
    # (the persistent kernel: up to 32 dimensions, any population whose grid -- walkers x workgroups per ladder -- is resident at once:
    #  a workgroup per CU; else, a ladder whose rungs x padded dimensions fit 256 lanes: the fused small-ladder kernel)
    DPad = 4 if D <= 4 else 8 if D <= 8 else 16 if D <= 16 else 32
    on_ladder = D <= 32 and W * -(-Nt // (256 // DPad)) <= 256
    fused = not on_ladder and D <= 16 and Nt * DPad <= 256
    assert eng.step_kernel_name.startswith("ladder_persistent_kernel<") == on_ladder, eng.step_kernel_name
    assert eng.step_kernel_name.startswith("ladder_steps_kernel<") == fused, eng.step_kernel_name
    if on_ladder:
        assert eng.step_kernel_name.endswith(", 11>"), eng.step_kernel_name
    done = 0
    for n in (1, 4, steps - 5):
        eng.step(n); eng.sync(); lad.pt_step(n)
        done += n
        PU.assert_same_state(eng, lad, "after %d PT steps" % done)
    he, ho = eng.history(), lad.history()
    nsize = eng.nsize
    assert nsize.max() <= cap
    for name in ("x", "llike", "lprior", "naccept", "ntries", "last_type"):
        for s_ in range(int(nsize.max())):
            have = nsize > s_
            got, want = he[name][s_ % cap][have], PU.to_engine_order(ho[name][:, s_], Nt, W)[have]
            assert np.array_equal(got, want), (name, s_)
    m = eng.map()
    assert np.array_equal(m["lpost"], PU.to_engine_order(lad.map_lpost, Nt, W))
    # the moves were made: types of differential evolution (member 0: 0 parallel, 10 snooker) and of the Gaussians were accepted
    seen = set(int(v) for row in he["last_type"][:int(nsize.min())] for v in np.unique(row))
    if snooker < 1.0:
        assert 0 in seen, seen
    if snooker > 0.0:
        assert 10 in seen, seen
    assert any(1 <= (v % 10) <= K for v in seen), seen
    if on_ladder:
        st = eng.ladder_stats()
        assert st["launches"] > 0 and st["fallbacks"] == 0 and not st["disabled"], st
    eng.close()


@pytest.mark.parametrize("D,Nt,W,kind,snooker,K,steps", [
    (12, 20, 4, E.PROP_DIAG, 0.2, 6, 60),
    (12, 16, 2, E.PROP_LOWER, 0.2, 3, 40),
    (32, 24, 2, E.PROP_DENSE, 0.3, 2, 40),
    (20, 40, 3, E.PROP_LOWER, 0.1, 3, 40),
    (6, 12, 3, E.PROP_DIAG, 0.3, 2, 60),       # (8 padded dimensions: 32 rungs per workgroup)
    (3, 10, 300, E.PROP_DIAG, 0.3, 2, 40),     # (more ladders than workgroups fit, evolving: exchange kernel + lanes kernel)
])
def test_differential_evolution_on_evolving_ladders(D, Nt, W, kind, snooker, K, steps):
    """The reference sampler's defaults together: the default proposal set drawn on the device AND pry_temps after every accepted
    exchange (chain.cc:1809-1846) -- 9..32 dimensions in one persistent launch per call (the ladder kernel's build FL = 15)."""
    cap = 2 * steps + 8
    pr, eng, lad = _pair(D, Nt, W, kind, 1, snooker, 12, K, cap)
    eng.set_evolve_temps(0.01); lad.evolve_temps(0.01)
    DPad = 4 if D <= 4 else 8 if D <= 8 else 16 if D <= 16 else 32
    on_ladder = D <= 32 and W * -(-Nt // (256 // DPad)) <= 256
    fused = not on_ladder and D <= 16 and Nt * DPad <= 256 and W <= 64   # (an evolving population beyond 64 ladders: two launches)
    assert eng.step_kernel_name.startswith("ladder_persistent_kernel<") == on_ladder, eng.step_kernel_name
    assert eng.step_kernel_name.startswith("ladder_steps_kernel<") == fused, eng.step_kernel_name
    if on_ladder:
        assert eng.step_kernel_name.endswith(", 15>"), eng.step_kernel_name
    done = 0
    for n in (1, 7, steps - 8):
        eng.step(n); eng.sync(); lad.pt_step(n)
        done += n
        PU.assert_same_state(eng, lad, "after %d PT steps" % done)
        assert np.array_equal(eng.invtemps(), lad.betaw), done
    PU.assert_same_history_and_map(eng, lad, cap)
    lt = set(int(v) for v in np.unique(eng.last_type))
    assert (0 in lt or 10 in lt) and any(1 <= v % 10 <= K for v in lt), lt
    if on_ladder:
        st = eng.ladder_stats()
        assert st["launches"] > 0 and st["fallbacks"] == 0 and not st["disabled"], st
    eng.close()


def test_differential_evolution_with_a_plugin_likelihood():
    """BASELINE configs[4]'s shape of problem -- the toy LISA likelihood through the C-ABI callback, mixed prior, wrap and limit
    boundaries -- with the sampler's default proposal recipe on the device: the propose pass draws the differential-evolution
    moves from the device history and hands ratio and type to the accept pass around the host's likelihood call."""
    import lisa_toy
    D, Nt, W, steps = 6, 24, 2, 50
    cap = 2 * steps + 8
    beta = E.geometric_ladder(Nt, 1e6)
    rng = np.random.default_rng(11)
    lo = np.array(lisa_toy.CENTERS) - np.array(lisa_toy.SCALES)
    hi = np.array(lisa_toy.CENTERS) + np.array(lisa_toy.SCALES)
    x0 = rng.uniform(lo + 0.05, hi - 0.05, size=(Nt * W, D))
    init = rng.uniform(lo + 0.05, hi - 0.05, size=(50 * D, Nt * W, D))
    sig = np.array(lisa_toy.SCALES) / 100.0
    fac = np.tile(sig, (Nt, 1))
    eng = E.Engine(D, Nt, W, swap_rate=0.1, history_rungs=Nt, history_capacity=cap, map_rungs=Nt)
    eng.set_bounds(lisa_toy.BLO, lisa_toy.BHI, lisa_toy.BMIN, lisa_toy.BMAX)
    eng.set_prior(lisa_toy.TYPES, lisa_toy.CENTERS, lisa_toy.SCALES)
    eng.set_target_callback(lisa_toy.loglike)
    eng.set_ladder(beta)
    eng.set_proposals(E.PROP_DIAG, fac)
    eng.set_states(x0)
    pb = O.Problem(D)
    pb.set_bounds(lisa_toy.BLO, lisa_toy.BHI, lisa_toy.BMIN, lisa_toy.BMAX)
    pb.set_prior(lisa_toy.TYPES, lisa_toy.CENTERS, lisa_toy.SCALES)
    pb.set_user(lisa_toy.loglike)
    lad = O.Ladder(pb, beta, W=W, swap_rate=0.1)
    lad.set_proposals([(O.PROP_DIAG, fac[r], 0.0) for r in range(Nt)])
    lad.use_philox(0x5EED0001)
    lad.enable_history(cap)
    lad.set_states(PU.to_oracle_order(x0, Nt, W))
    cum, scales, odfs = _recipe(Nt, 6, 0.8, 0.5)
    eng.set_proposal_mixture(cum, scales, odfs); lad.set_mixture(cum, scales, odfs)
    eng.set_proposal_de(0.1, 0.3, 4.0, 0.0, init_rows=init)
    lad.set_de(0.1, 0.3, 4.0, 0.0, init_rows=np.stack([PU.to_oracle_order(init[k], Nt, W) for k in range(init.shape[0])]))
    eng.set_evolve_temps(0.01); lad.evolve_temps(0.01)
    for k in range(5):
        eng.step(10); eng.sync(); lad.pt_step(10)
        PU.assert_same_state(eng, lad, "after %d steps" % (10 * (k + 1)))
        assert np.array_equal(eng.invtemps(), lad.betaw)
    lt = set(int(v) for v in np.unique(eng.last_type))
    assert (0 in lt or 10 in lt) and any(1 <= v % 10 <= 6 for v in lt), lt
    eng.close()


def test_a_history_ring_too_short_for_differential_evolution_is_reported():
    """The device draws from the WHOLE saved history (ignore_frac = 0, the sampler's default): a ring that has lost a row the draw asks
    for is an error ptm_sync reports, never a silently different chain."""
    pr, eng, lad = _pair(6, 6, 2, E.PROP_DIAG, 1, 0.1, 10, 2, cap=16)
    eng.step(80)
    with pytest.raises(E.PtmError, match="history"):
        eng.sync()
    eng.close()


def test_differential_evolution_is_refused_on_a_rung_shard():
    """Every rung's history is what the draw reads; a rung shard cannot record its top rung's completely (the row held between two
    exchanges of one step sits in the shard above): refused when it is set, not when the first such row is missed."""
    D, Nt, W = 6, 12, 2
    pr = PU.problem_for(D, Nt, 1e3)
    # (a shard below the ladder's top is not even created with history on all its rungs; the top shard is, and is refused here)
    with pytest.raises(E.PtmError, match="top rung of a shard"):
        E.Engine(D, Nt, W, rung_begin=0, rung_count=6, history_rungs=6, history_capacity=64)
    eng = E.Engine(D, Nt, W, rung_begin=6, rung_count=6, history_rungs=6, history_capacity=64)
    pr.configure(eng, E.PROP_DIAG)
    cum, scales, odfs = _recipe(Nt, 2, 0.7, 0.5)
    eng.set_proposal_mixture(cum[:6], scales[:6], odfs[:6])
    with pytest.raises(E.PtmError, match="rung shard"):
        eng.set_proposal_de(0.1, 0.3, 4.0, 0.0)
    eng.close()


@pytest.mark.parametrize("plugin", [False, True])
def test_initial_rows_drawn_in_one_call_are_the_rows_of_one_initialisation_each(plugin):
    """ptm_draw_prior_rows: MH_chain::initialize(n)'s rows in front of the start state (chain.cc:846-876) in one call, bit for bit what
    ptm_init_from_prior_k leaves in the engine draw by draw -- device target and plug-in likelihood (with its redraw loop) -- and the
    engine's own state untouched."""
    if plugin:
        import lisa_toy
        D, Nt, W = 6, 10, 3
        eng = E.Engine(D, Nt, W, swap_rate=0.1)
        eng.set_bounds(lisa_toy.BLO, lisa_toy.BHI, lisa_toy.BMIN, lisa_toy.BMAX)
        eng.set_prior(lisa_toy.TYPES, lisa_toy.CENTERS, lisa_toy.SCALES)
        eng.set_target_callback(lisa_toy.loglike)
        eng.set_ladder(E.geometric_ladder(Nt, 1e4))
        eng.set_proposals(E.PROP_DIAG, np.tile(np.array(lisa_toy.SCALES) / 100.0, (Nt, 1)))
    else:
        D, Nt, W = 20, 7, 5
        pr = PU.problem_for(D, Nt, 1e3)
        eng = E.Engine(D, Nt, W)
        pr.configure(eng, E.PROP_LOWER)
    eng.init_from_prior()
    x0, ll0 = eng.states(), eng.llike.copy()
    x, ll, lp = eng.draw_prior_rows(2, 5)
    assert np.array_equal(eng.states(), x0) and np.array_equal(eng.llike, ll0)
    for j, k in enumerate(range(2, 7)):
        eng.init_from_prior(k)
        assert np.array_equal(eng.states(), x[j]) and np.array_equal(eng.llike, ll[j]) and np.array_equal(eng.lprior, lp[j]), k
    assert len({tuple(r) for r in x[:, 0, :]}) == 5
    eng.close()
