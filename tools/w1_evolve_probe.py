"""The bare 1024-rung ladder with evolving temperatures (the reference sampler's default ladder), us per PT step; with
PTM_LADDER_PROF=1 / 2 / 3 the persistent ladder kernel prints its phase clocks (a chains' wave, a replay wave, a window wave).
usage: python tools/w1_evolve_probe.py [steps] [everything]"""
import os
import sys
import time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ptmcmc_amd import engine as E
from ptmcmc_amd.problems import GaussianProblem

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
full = len(sys.argv) > 2
D, Nt, W = 32, 1024, 1
pr = GaussianProblem(D, Nt, 1e9)
e = E.Engine(D, Nt, W, add_every_n=100, history_rungs=Nt if full else 0, history_capacity=64 if full else 0, map_rungs=Nt if full else 0)
pr.configure(e, E.PROP_LOWER)
if full:
    K = 4
    sh = np.cumsum([2.0 ** (k + 1) for k in range(K)]); sh /= sh[-1]
    e.set_proposal_mixture(np.tile(sh, (Nt, 1)), np.tile([2.0 ** -k for k in range(K)], (Nt, 1)), np.full((Nt, K), 0.5))
e.set_evolve_temps(0.01)
e.init_from_prior()
e.step(200); e.sync()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    e.step(n); e.sync()
    best = min(best, (time.perf_counter() - t0) / n)
print("evolving%s: %.2f us per PT step   [%s]" % (", everything" if full else "", best * 1e6, e.step_kernel_name), flush=True)
e.close()
