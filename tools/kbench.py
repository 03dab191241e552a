#!/usr/bin/env python3
"""Kernel A/B harness: time the fused sweep kernel (and the exchange kernel) of a given engine build.
usage: PTM_ENGINE_LIB=path/to/lib.so python tools/kbench.py [--walkers W] [--dim D] [--rungs N] [--kind lower|dense|diag]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ptmcmc_amd import engine as E
from ptmcmc_amd.problems import GaussianProblem

ap = argparse.ArgumentParser()
ap.add_argument("--walkers", type=int, default=4096)
ap.add_argument("--dim", type=int, default=32)
ap.add_argument("--rungs", type=int, default=1024)
ap.add_argument("--tmax", type=float, default=1e9)
ap.add_argument("--kind", default="lower")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--tag", default="")
ap.add_argument("--bounds", action="store_true", help="limit bounds on every dimension, uniform prior: the general build's cheap case")
ap.add_argument("--general", action="store_true", help="limit bounds on every dimension + one gaussian prior factor: the general build")
ap.add_argument("--recipe", action="store_true", help="the reference sampler's default Gaussian recipe (ptmcmc.cc:117-139) on limit bounds: a 4-member scale mixture of the rung's factor with gauss_1d_frac = 0.5")
ap.add_argument("--recipe-null", action="store_true", help="the recipe's CODE without its physics: four members of scale 1, one-dimensional moves switched on but never drawn (gauss_1d_frac 1e-300) -- the chains move as in --bounds")
ap.add_argument("--evolve", type=float, default=0.0, help="evolve_temps rate (per-ladder temperatures, sequential exchange decisions)")
ap.add_argument("--history", type=int, default=0, help="rungs with a saved history (and MAP tracking on every rung): the sampler's engine")
a = ap.parse_args()
kind = {"lower": E.PROP_LOWER, "dense": E.PROP_DENSE, "diag": E.PROP_DIAG}[a.kind]
pr = GaussianProblem(a.dim, a.rungs, a.tmax)
eng = E.Engine(a.dim, a.rungs, a.walkers, add_every_n=100, time_kernels=True, history_rungs=a.history, history_capacity=64 if a.history else 0,
               map_rungs=a.rungs if a.history else 0)
pr.configure(eng, kind)
if a.bounds:
    eng.set_bounds([1] * a.dim, [1] * a.dim, [-1e3] * a.dim, [1e3] * a.dim)
if a.general:
    D = a.dim
    eng.set_bounds([1] * D, [1] * D, [-1e3] * D, [1e3] * D)
    eng.set_prior([2] + [1] * (D - 1), list(pr.centers), [30.0] + list(pr.halfwidths[1:]))
if a.recipe_null:
    D, K = a.dim, 4
    eng.set_bounds([1] * D, [1] * D, [-1e3] * D, [1e3] * D)
    eng.set_proposal_mixture(np.tile([0.25, 0.5, 0.75, 1.0], (a.rungs, 1)), np.ones((a.rungs, K)), np.full((a.rungs, K), 1e-300))
if a.recipe:
    D, K = a.dim, 4
    eng.set_bounds([1] * D, [1] * D, [-1e3] * D, [1e3] * D)
    sh = np.cumsum([2.0 ** (k + 1) for k in range(K)]); sh /= sh[-1]
    eng.set_proposal_mixture(np.tile(sh, (a.rungs, 1)), np.tile([2.0 ** -k for k in range(K)], (a.rungs, 1)), np.full((a.rungs, K), 0.5))
if a.evolve > 0:
    eng.set_evolve_temps(a.evolve)
eng.init_from_prior()
eng.sweep(3); eng.sync(); eng.kernel_times()
eng.timer_start(); eng.sweep(a.reps); ms_sweep = eng.timer_stop() / a.reps
kt = eng.kernel_times()
eng.timer_start(); eng.step(a.reps); ms_step = eng.timer_stop() / a.reps
kt2 = eng.kernel_times()
nch = a.rungs * a.walkers
bytes_alg = (16 * a.dim + 44) * nch
print("%-28s %-34s sweep %.4f ms (min %.4f)  step %.4f ms [sweep-in-step %.4f, exchange %.4f]  -> %.3e MH steps/s (sweep only), %.0f GB/s algorithmic = %.1f%% of 8 TB/s"
      % (a.tag or os.path.basename(E.LIB_PATH), eng.sweep_kernel_name, ms_sweep, kt.min(), ms_step, kt2.mean(), ms_step - kt2.mean(), nch / (ms_sweep * 1e-3),
         bytes_alg / (ms_sweep * 1e-3) / 1e9, bytes_alg / (ms_sweep * 1e-3) / 8e12 * 100), flush=True)
