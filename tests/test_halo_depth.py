"""How deep must the llike halo between two ladder shards be?  (CPU; the oracle replays the exchange candidate draws.)

A shard decides the exchanges that reach it from above out of the llikes of the H rungs above its top rung.  It cannot
when the SURVIVING picks of a step (chain.cc:1410-1420: accepted or not) cover H+1 consecutive pairs upwards from its top
rung -- ptm_exchange_decide then raises PTM_ERR_FAR_MOVE.  This test measures how often that happens on the engine's own
ladder streams at the benchmark's shape (1024 rungs, swap_rate 0.1, 8 shards) and pins the bound the default halo rests on."""
import math

import numpy as np

import oracle_lib as O
from ptmcmc_amd.parallel import DEFAULT_HALO


def test_runs_of_surviving_picks_above_a_shard_boundary():
    Nt, sr, G = 1024, 0.1, 8
    bounds = np.arange(Nt // G, Nt, Nt // G)                  # the 7 boundaries of bench.py --gpus 8
    W, nsteps = 32768, 440
    h = O.selection_run_census(0x5EED0001, Nt, sr, W, nsteps, bounds, Lmax=12).sum(axis=0)
    n = W * nsteps * len(bounds)
    assert n >= 10 ** 8 and h.sum() == n
    ge = h[::-1].cumsum()[::-1]                               # ge[k] = events with a run of >= k picks
    # a run of k surviving picks needs k picked rungs (each with probability <= swap_rate) whose FIRST picks come in
    # descending rung order (alive[n] = !(alive[n-1] and first[n-1] < first[n])): <= sr^k / k!
    for k in range(1, 5):
        bound = sr ** k / math.factorial(k)
        assert ge[k] / n <= bound * (1 + 4 / math.sqrt(bound * n)), (k, ge[k] / n, bound)   # (bound + 4 sigma)
        assert ge[k] / n >= 0.5 * bound, (k, ge[k] / n, bound)                              # ... and it is not slack
    # the first round's default halo of 4 fails on a run of 5: ~2-6e-8 per boundary-ladder-step, i.e. several times in one
    # 8-GPU bench run -- too rare for this sample to pin (a handful of events), never seen beyond 6
    assert ge[5] <= 30 and ge[7] == 0
    # what the default halo assumes
    assert DEFAULT_HALO >= 12
    p_fail = sr ** (DEFAULT_HALO + 1) / math.factorial(DEFAULT_HALO + 1)
    bench_bws = 16384 * 8 * (G - 1) * (300 + 100 + 10)        # ladders x boundaries x steps of bench.py --gpus 8
    assert p_fail * bench_bws < 2e-6
    assert p_fail * 16384 * 8 * (G - 1) * 1e6 < 1e-9          # ... and a 10^6-step production run of that size: never
    old = sr ** 5 / math.factorial(5)
    assert old * bench_bws > 5                                # ... and what the old default of 4 meant


def test_census_agrees_with_the_oracle_swap_phase():
    """the census' per-rung form of the drop rule == the surviving picks ptmo_pt_step logs (its loop over earlier picks)"""
    from ptmcmc_amd.problems import GaussianProblem
    D, Nt, W, sr, seed = 2, 24, 50, 0.45, 77
    pr = GaussianProblem(D, Nt, 1e3)
    pb = O.Problem(D)
    pb.set_bounds([0] * D, [0] * D, [0.0] * D, [0.0] * D)
    pb.set_prior(pr.types, pr.centers, pr.halfwidths)
    pb.set_gauss(pr.P, pr.like0)
    lad = O.Ladder(pb, pr.beta, W=W, swap_rate=sr)
    fac = pr.proposal_factors()
    lad.set_proposals([(O.PROP_DENSE, fac[r], 0.0) for r in range(Nt)])
    lad.use_philox(seed)
    lad.init_from_prior(seed)
    bounds = np.array([6, 12, 18])
    Lmax = 8
    want = np.zeros((len(bounds), Lmax + 1), dtype=np.int64)
    nsteps = 12
    for _ in range(nsteps):
        lad.pt_step(1)
        for pairs in lad.last_pairs:
            alive = set(int(v) for v in pairs if v >= 0)
            for i, b in enumerate(bounds):
                L = 0
                while (b - 1 + L) in alive:
                    L += 1
                want[i, min(L, Lmax)] += 1
    got = O.selection_run_census(seed, Nt, sr, W, nsteps, bounds, Lmax=Lmax, nthreads=2)
    assert np.array_equal(got, want)
    assert want[:, 3:].sum() > 0
