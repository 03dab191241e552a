#!/usr/bin/env python3
"""Latency regime: time per PT step of the small BASELINE configurations (and a sampler-sized ladder), default build vs
PTM_FUSED=0 (exchange kernel + lanes kernel per step instead of the fused small-ladder kernel)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ptmcmc_amd import engine as E
from ptmcmc_amd.problems import GaussianProblem

for name, D, Nt, W, tmax in (("C1", 2, 8, 1, 1e2), ("C2", 16, 64, 1, 1e4), ("C3", 32, 256, 4, 1e6), ("20x6", 6, 20, 1, 1e9), ("20x6 x64 replicas", 6, 20, 64, 1e9)):
    pr = GaussianProblem(D, Nt, tmax)
    eng = E.Engine(D, Nt, W, add_every_n=100)
    pr.configure(eng, E.PROP_LOWER)
    eng.init_from_prior()
    eng.step(200); eng.sync()
    n = 4000
    t0 = time.perf_counter(); eng.step(n); eng.sync(); dt = time.perf_counter() - t0
    print("%-18s D=%d rungs=%d walkers=%d  fused=%s: %.2f us per PT step = %.3e MH steps/s  (%s)"
          % (name, D, Nt, W, os.environ.get("PTM_FUSED", "1"), dt / n * 1e6, Nt * W * n / dt, eng.sweep_kernel_name), flush=True)
    eng.close()
