"""Step times of the small BASELINE configurations (latency regime): C1 2-D 8 rungs, C2 16-D 64 rungs x 1 walker,
C3 32-D 256 rungs x 4 walkers, and the bare 1024-rung ladder.  usage (GPU box): python tools/small_probe.py"""
import os
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptmcmc_amd import engine as E
from ptmcmc_amd.problems import GaussianProblem
for name, D, Nt, W, tmax in (("C1", 2, 8, 1, 1e2), ("C2", 16, 64, 1, 1e4), ("C3", 32, 256, 4, 1e6), ("W=1", 32, 1024, 1, 1e9)):
    pr = GaussianProblem(D, Nt, tmax)
    e = E.Engine(D, Nt, W, add_every_n=100)
    pr.configure(e, E.PROP_LOWER)
    e.init_from_prior()
    e.step(300); e.sync()
    e.timer_start(); e.step(1000); ms = e.timer_stop() / 1000
    print("%-4s D=%-2d %4d rungs x %d: %.1f us per PT step, %.3e MH steps/s  (%s)" % (name, D, Nt, W, ms * 1e3, Nt * W / (ms * 1e-3), e.sweep_kernel_name), flush=True)
    e.close()
