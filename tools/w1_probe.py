"""The bare 1024-rung ladder (W = 1: the reference's own shape), us per PT step; also C2 / C3 of BASELINE.json.
usage: python tools/w1_probe.py [steps]      (PTM_LADDER=0: the two-launch path)"""
import os
import sys
import time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ptmcmc_amd import engine as E
from ptmcmc_amd.problems import GaussianProblem

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for D, Nt, W, tmax in ((32, 1024, 1, 1e9), (32, 256, 4, 1e6), (16, 64, 1, 1e4)):
    pr = GaussianProblem(D, Nt, tmax)
    e = E.Engine(D, Nt, W, add_every_n=100)
    pr.configure(e, E.PROP_LOWER)
    e.init_from_prior()
    e.step(200); e.sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        e.step(n); e.sync()
        best = min(best, (time.perf_counter() - t0) / n)
    print("D=%d %d rungs x %d: %.2f us per PT step = %.3g MH steps/s   [%s]" % (D, Nt, W, best * 1e6, Nt * W / best, e.step_kernel_name), flush=True)
    e.close()
# the same ladder with what the reference sampler switches on by default (ptmcmc.cc:117-139,601-616): a 4-member scale mixture with
# one-dimensional moves, the cold rungs' history (every 100th add saved) and MAP tracking on every rung
import numpy as np
for what in ("recipe", "history", "recipe + history", "evolving", "recipe + history + evolving"):
    D, Nt, W = 32, 1024, 1
    pr = GaussianProblem(D, Nt, 1e9)
    hist = "history" in what
    e = E.Engine(D, Nt, W, add_every_n=100, history_rungs=Nt if hist else 0, history_capacity=64 if hist else 0, map_rungs=Nt if hist else 0)
    pr.configure(e, E.PROP_LOWER)
    if "recipe" in what:
        K = 4
        sh = np.cumsum([2.0 ** (k + 1) for k in range(K)]); sh /= sh[-1]
        e.set_proposal_mixture(np.tile(sh, (Nt, 1)), np.tile([2.0 ** -k for k in range(K)], (Nt, 1)), np.full((Nt, K), 0.5))
    if "evolving" in what:
        e.set_evolve_temps(0.01)
    e.init_from_prior()
    e.step(200); e.sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        e.step(n); e.sync()
        best = min(best, (time.perf_counter() - t0) / n)
    print("D=%d %d rungs x %d, %s: %.2f us per PT step   [%s] %s" % (D, Nt, W, what, best * 1e6, e.step_kernel_name, e.ladder_stats()), flush=True)
    e.close()
