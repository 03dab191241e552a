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
