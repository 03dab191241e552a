"""ptm_step(n) called with small n, as a host loop that steps one at a time does: where does the persistent ladder kernel (one
synchronous launch per call) stop paying against the two-launch path (asynchronous launches)?  usage (GPU box): python tools/step1_probe.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptmcmc_amd import engine as E
from ptmcmc_amd.problems import GaussianProblem

D, Nt, W = 32, 1024, 1
pr = GaussianProblem(D, Nt, 1e9)
eng = E.Engine(D, Nt, W, swap_rate=0.1)
pr.configure(eng, E.PROP_LOWER)
eng.init_from_prior()
eng.step(200); eng.sync()
for n in (1, 2, 4, 8, 16, 64):
    calls = max(50, 2000 // n)
    t0 = time.perf_counter()
    for _ in range(calls):
        eng.step(n)
    eng.sync()
    dt = time.perf_counter() - t0
    print("ptm_step(%d) x %d: %.2f us per PT step   [%s]" % (n, calls, dt / (calls * n) * 1e6, eng.step_kernel_name), flush=True)
eng.close()
