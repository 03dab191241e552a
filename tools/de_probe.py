"""The reference sampler's default configuration on a device target -- 80 % differential evolution from the chain's saved history + six
Gaussians with gauss_1d_frac 0.5 (ptmcmc.cc:60-143), evolving ladder, history every second add, MAP tracking -- bare engine, us per PT
step.  PTM_LADDER=0: two launches per step (lanes kernel); PTM_LADDER=0 PTM_FORCE_VALU=1: the general kernel.
A fifth argument `general`: wrap / reflect boundaries and Gaussian priors on half of the dimensions (the general state space).
usage: python tools/de_probe.py [D] [Nt] [W] [steps] [general]"""
import os
import sys
import time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ptmcmc_amd import engine as E
from ptmcmc_amd.problems import GaussianProblem

D = int(sys.argv[1]) if len(sys.argv) > 1 else 12
Nt = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n = int(sys.argv[4]) if len(sys.argv) > 4 else 4000
reps, warm, every = 3, 200, 2
cap = (warm + reps * n) * 2 // every + 64
pr = GaussianProblem(D, Nt, 1e6)
e = E.Engine(D, Nt, W, add_every_n=every, history_rungs=Nt, history_capacity=cap, map_rungs=Nt)
pr.configure(e, E.PROP_DIAG)
general = len(sys.argv) > 5
if general:
    hw = np.asarray(pr.halfwidths)
    blo = [3 if d % 3 == 0 else (2 if d % 3 == 1 else 0) for d in range(D)]   # wrap, reflect, open
    e.set_bounds(blo, blo, list(-hw), list(hw))
    types = [2 if d % 2 else 1 for d in range(D)]                              # gaussian / uniform
    e.set_prior(types, [0.0] * D, list(hw))
K = 6
g = 2.0 ** np.arange(1, K + 1)
shares = np.concatenate([[0.8], 0.2 * g / g.sum()])
cum = np.tile(np.cumsum(shares), (Nt, 1)); cum[:, -1] = 1.0
scales = np.tile(np.concatenate([[-1.0], 4.0 ** -np.arange(K)[::-1]]), (Nt, 1))
odfs = np.tile(np.concatenate([[0.0], np.full(K, 0.5)]), (Nt, 1))
e.set_proposal_mixture(cum, scales, odfs)
e.init_from_prior()
rng = np.random.default_rng(1)
init = rng.uniform(-1.0, 1.0, size=(50 * D, Nt * W, D)) * np.asarray(pr.halfwidths)[None, None, :] * 0.02
e.set_proposal_de(0.1, 0.3, 4.0, 0.0, init_rows=init)
e.set_evolve_temps(0.01)
e.step(warm); e.sync()
best = 1e9
for rep in range(reps):
    t0 = time.perf_counter()
    e.step(n); e.sync()
    best = min(best, (time.perf_counter() - t0) / n)
t, a = e.counter_sums()
print("D=%d, %d rungs x %d, default recipe + evolving ladder + history + MAP%s: %.2f us per PT step   [%s]  MH acceptance %.3f" %
      (D, Nt, W, ", wrap / reflect bounds + Gaussian priors" if general else "", best * 1e6, e.step_kernel_name, a / max(1, t)), flush=True)
e.close()
