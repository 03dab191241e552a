"""Independent of the CPU checker: does the engine SAMPLE the right distributions?  Many independent ladders run from
prior draws; across walkers the cold rung must show the target's covariance, rung r the covariance cov / beta_r (its
tempered target, the prior box being 100 sigma wide), fixed and evolving ladder alike, on every kernel family.
(The pass / fail form of this, with thresholds derived from the sample counts and chains started from exact samples, is
tests/test_gpu_statistics.py since round 4; this tool starts from prior draws 100 sigma out and prints what it finds.)
usage (GPU box): python tools/stat_check.py"""
import os
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ptmcmc_amd import engine as E
from ptmcmc_amd.problems import GaussianProblem

worst = 0.0
for D, Nt, W, kind, ev, burn in ((8, 12, 8192, E.PROP_LOWER, 0.0, 1500), (32, 8, 4096, E.PROP_LOWER, 0.0, 2500), (32, 8, 4096, E.PROP_LOWER, 0.01, 2500),
                               (12, 10, 60, E.PROP_DENSE, 0.0, 3000),
                               (64, 6, 2048, E.PROP_LOWER, 0.0, 4000), (128, 5, 1024, E.PROP_LOWER, 0.01, 60000),  # the 64- / 128-dimension MFMA kernels (from 100 sigma out, 128 dimensions take their time)
                               (32, 40, 48, E.PROP_LOWER, 0.0, 3000)):                                             # the persistent ladder kernel (40 x 48: 7.2e4 samples per rung)

    pr = GaussianProblem(D, Nt, 1e2)
    eng = E.Engine(D, Nt, W, swap_rate=0.2)
    pr.configure(eng, kind)
    if ev:
        eng.set_evolve_temps(ev)
    eng.init_from_prior()
    eng.step(burn); eng.sync()
    acc = np.zeros((Nt, D, D)); n = 0
    reps = (40 if D <= 64 else 120) if W >= 1000 else 1500
    for k in range(reps):
        eng.step(25 if W >= 1000 else 10); eng.sync()
        X = eng.states().reshape(Nt, W, D)
        acc += np.einsum("rwi,rwj->rij", X, X); n += W
    beta = eng.invtemps().mean(axis=0)
    errs = []
    for r in range(Nt):
        C = acc[r] / n
        want = pr.cov / beta[r]
        s = np.sqrt(np.diag(want))
        errs.append(np.abs((C - want) / np.outer(s, s)).max())
    worst = max(worst, max(errs[:max(1, Nt // 2)]))
    print("D=%d %dx%d %s ladder, kernel %s: max |C - cov/beta| / (sigma_i sigma_j): cold rung %.4f, all rungs %.4f  (samples per rung %d: 5 sigma of the estimator at n / 2 = %.4f)"
          % (D, Nt, W, "evolving" if ev else "fixed", eng.step_kernel_name, errs[0], max(errs), n, 5 * np.sqrt(4.0 / n)), flush=True)
    eng.close()
print("worst (colder half):", worst)
