#!/usr/bin/env python3
"""Feasibility probe: does the exchange phase hide behind the sweep when the population runs as TWO halves on two streams?
Two engines (walkers [0, W/2) and [W/2, W): the same chains as one engine of W walkers, ptm_config.walker_begin) stepped
alternately against one engine of W walkers.  usage: [PTM_PERSIST=2] python tools/overlap_probe.py [--walkers W] [--steps K]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ptmcmc_amd import engine as E
from ptmcmc_amd.problems import GaussianProblem

ap = argparse.ArgumentParser()
ap.add_argument("--walkers", type=int, default=16384)
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--parts", type=int, default=2)
a = ap.parse_args()
D, NT = 32, 1024
pr = GaussianProblem(D, NT, 1e9)
dev = torch.device("cuda", 0)

def make(W, wb, stream):
    e = E.Engine(D, NT, W, seed=1234, swap_rate=0.1, add_every_n=100, stream=stream.cuda_stream, walker_begin=wb)
    pr.configure(e, E.PROP_LOWER)
    e.init_from_prior()
    return e

def run_alternating(engs, K):
    """sweeps strictly one after the other (events), each part's exchange kernel free to run beside the OTHER part's sweep"""
    for e in engs:
        e.step(150)
    for e in engs:
        e.sync()
    torch.cuda.synchronize()
    last = None
    t0 = time.perf_counter()
    for k in range(K):
        for i, e in enumerate(engs):
            e.exchange_decide(None, None, 0, None, None)
            if last is not None:
                s[i].wait_event(last)
            e.sweep_rungs(0, NT, True)
            last = torch.cuda.Event()
            last.record(s[i])
    for e in engs:
        e.sync()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


def run(engs, K):
    for e in engs:
        e.step(150)
    for e in engs:
        e.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # out of phase: the first half a step ahead
    for k in range(K):
        for e in engs:
            e.step(1)
    for e in engs:
        e.sync()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3

s = [torch.cuda.Stream(device=dev) for _ in range(a.parts)]
one = make(a.walkers, 0, s[0])
ms1 = run([one], a.steps)
x1 = one.states()
one.close()
Wh = a.walkers // a.parts
parts = [make(Wh, k * Wh, s[k]) for k in range(a.parts)]
msn = run(parts, a.steps)
msa = run_alternating(parts, a.steps)
xs = np.concatenate([p.states().reshape(NT, Wh, D) for p in parts], axis=1).reshape(-1, D)
print("PTM_PERSIST=%s  one engine of %d walkers: %.3f ms/step   %d engines of %d on %d streams: free-running %.3f, sweeps alternating %.3f ms/step  (same chains: %s)"
      % (os.environ.get("PTM_PERSIST", "-"), a.walkers, ms1, a.parts, Wh, a.parts, msn, msa, np.array_equal(x1, xs)), flush=True)
