"""Randomised bit-exact comparison of the engine with the CPU checker (test infrastructure: imports tests/): shapes, proposal
kinds, boundaries, priors, a mean, one-dimensional moves, evolving ladders with and without the posterior-ordering cut, drawn
scale mixtures, the history ring with MAP tracking, differential evolution from the device history -- drawn from a seeded generator; a few PT steps and plain sweeps each.  Stops at the first difference with the case printed.
usage (GPU box): python tools/fuzz_parity.py [seconds] [seed]"""
import math
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import parity_util as PU
from ptmcmc_amd import engine as E

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t0 = time.time()
n = 0
kernels = {}
while time.time() - t0 < budget:
    D = int(rng.choice([1, 2, 3, 5, 8, 9, 13, 16, 17, 24, 31, 32, 33, 48, 64, 65, 100, 128, 130, 200, 300, 600]))
    W = int(rng.choice([1, 2, 3, 5, 64, 64, 70, 128, 192, 1024]))
    Nt = int(rng.choice([1, 2, 3, 5, 8, 13, 40, 200] if W < 1024 else [3, 5, 8]))
    if D > 32 and W >= 1024:
        W = 64
    if Nt * W * D > 1.5e6:           # (the checker runs on one core)
        Nt = max(2, int(1.5e6 / (W * D)))
    kind = [E.PROP_LOWER, E.PROP_DENSE, E.PROP_DIAG][int(rng.integers(3))]
    sr = float(rng.choice([0.05, 0.1, 0.3, 0.45, 0.7]))
    odf = None if rng.random() < 0.5 else float(rng.choice([0.2, 0.5, 1.0]))
    bounds = prior = mean = None
    flav = int(rng.integers(4))      # 0 plain, 1 limit / open bounds + uniform prior, 2 everything, 3 mean only
    if flav in (1, 2):
        kinds = [0, 1] if flav == 1 else [0, 1, 2, 3]      # open, limit, reflect, wrap
        blo = [int(rng.choice(kinds)) for _ in range(D)]
        bhi = [b if b in (2, 3) else int(rng.choice([0, 1])) for b in blo]
        bmin = list(rng.uniform(-3.0, -1.5, D)); bmax = list(rng.uniform(1.5, 3.0, D))
        bounds = (blo, bhi, bmin, bmax)
        types = [1] * D
        cen, hw = [0.0] * D, list(rng.uniform(3.5, 6.0, D))
        if flav == 2:
            for d in range(D):
                if rng.random() < 0.3:
                    types[d] = 2; cen[d] = float(rng.normal() * 0.2); hw[d] = float(rng.uniform(0.8, 2.0))
        prior = (types, cen, hw)
    if flav in (2, 3) and rng.random() < 0.7:
        mean = rng.normal(size=D) * 0.1
    x0 = rng.uniform(-1.2, 1.2, size=(Nt * W, D)) if bounds is not None else None
    ev = 0.0 if (rng.random() < 0.5 or Nt < 3) else float(rng.choice([0.01, 0.05]))
    cut = -1.0 if rng.random() < 0.6 else float(rng.choice([0.0, 1.0, 3.0]))
    K = 0 if rng.random() < 0.65 else int(rng.choice([1, 2, 4]))                         # a scale mixture (the sampler's default Gaussian recipe)
    hist = 0 if (rng.random() < 0.65 or Nt * W * D > 3e5 or D > 128) else int(rng.choice([1, 2, 3]))   # history + MAP, every hist-th add saved
    # differential evolution as the first member of the set (needs every rung's history; 12 D initial rows make it ready, none: passed over)
    de = bool(hist) and D <= 128 and rng.random() < 0.6
    if de and not K:
        K = int(rng.choice([1, 2, 6]))
    de_init = int(rng.choice([0, 12])) * D if de else 0
    de_snk = float(rng.choice([0.0, 0.1, 0.5, 1.0])) if de else 0.0
    case = dict(D=D, Nt=Nt, W=W, kind=kind, sr=sr, odf=odf, flav=flav, mean=mean is not None, ev=ev, cut=cut, K=K, hist=hist, de=de, de_init=de_init, de_snk=de_snk)
    try:
        pr, eng, lad = PU.make_pair(D, Nt, W, 1e3, kind=kind, swap_rate=sr, one_d_frac=odf, bounds=bounds, prior=prior, mean=mean, x0=x0,
                                    add_every_n=max(1, hist), history_cap=32 if hist else 0)
        if K:
            shares = 2.0 ** np.arange(1, K + 1)
            cumk = np.cumsum(shares) / shares.sum()
            sck = 2.0 ** -np.arange(K)[::-1]
            odk = np.where(np.arange(K) % 2 == 0, odf or 0.0, 0.0)
            if de:   # the sampler's default set: differential evolution first (negative scale), then the Gaussians
                cumk, sck, odk = np.concatenate([[0.7], 0.7 + 0.3 * cumk]), np.concatenate([[-1.0], sck]), np.concatenate([[0.0], odk])
            cum = np.tile(cumk, (Nt, 1)); cum[:, -1] = 1.0
            scales, odfs = np.tile(sck, (Nt, 1)), np.tile(odk, (Nt, 1))
            eng.set_proposal_mixture(cum, scales, odfs); lad.set_mixture(cum, scales, odfs)
        if de:
            init = None
            if de_init:
                init = rng.uniform(-1.0, 1.0, size=(de_init, Nt * W, D)) * 0.3
            eng.set_proposal_de(de_snk, 0.3, 4.0, 0.0, init_rows=init)
            lad.set_de(de_snk, 0.3, 4.0, 0.0, init_rows=None if init is None else np.stack([PU.to_oracle_order(init[k], Nt, W) for k in range(de_init)]))
        if ev:
            eng.set_evolve_temps(ev, lpost_cut=cut); lad.evolve_temps(ev, cut)
        for k in range(2):
            eng.step(3); eng.sync(); lad.pt_step(3)
            PU.assert_same_state(eng, lad, "step %d" % (3 * k + 3))
            if ev:
                assert np.array_equal(eng.invtemps(), lad.betaw)
            if k == 0:
                eng.sweep(2); eng.sync(); lad.sweep(2)
                PU.assert_same_state(eng, lad, "plain sweeps")
        t, a = eng.swap_counts()
        assert np.array_equal(t, lad.swap_count) and np.array_equal(a, lad.swap_accept_count)
        if hist:
            PU.assert_same_history_and_map(eng, lad, 32)
        for nm in {eng.sweep_kernel_name, eng.step_kernel_name}:      # (the step of a long ladder of few walkers is ONE kernel)
            kernels[nm] = kernels.get(nm, 0) + 1
        eng.close()
    except Exception:
        print("FAILED case", case, flush=True)
        raise
    n += 1
    if n % 20 == 0:
        print("  %d cases ok (%.0f s)" % (n, time.time() - t0), flush=True)
print("%d random cases bit-identical in %.0f s (seed %d); kernels exercised:" % (n, time.time() - t0, seed))
for k in sorted(kernels):
    print("   %4d  %s" % (kernels[k], k))
