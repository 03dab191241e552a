"""Runs a population too big for the CPU checker in a process of its own (the engine's environment switches are read once per
process) and prints a digest of everything a chain is -- states, llikes, lpriors, counters, swap bookkeeping -- so that two builds
/ two kernels can be compared bit for bit at full size.  TEST INFRASTRUCTURE.
usage: python hash_worker.py D Nt W kind nsteps [evolve_rate]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from ptmcmc_amd import engine as E
from ptmcmc_amd.problems import GaussianProblem

if __name__ == "__main__":
    D, Nt, W = (int(v) for v in sys.argv[1:4])
    kind = {"lower": E.PROP_LOWER, "dense": E.PROP_DENSE, "diag": E.PROP_DIAG}[sys.argv[4]]
    nsteps = int(sys.argv[5])
    ev = float(sys.argv[6]) if len(sys.argv) > 6 else 0.0
    pr = GaussianProblem(D, Nt, 1e4)
    eng = E.Engine(D, Nt, W, swap_rate=0.2)
    pr.configure(eng, kind)
    if ev > 0:
        eng.set_evolve_temps(ev)
    eng.init_from_prior()
    h = hashlib.sha256()
    for k in range(2):
        eng.step(nsteps // 2); eng.sync()
        for a in (eng.states(), eng.llike, eng.lprior, eng.ntries, eng.naccept, eng.nhist, eng.last_type) + tuple(eng.swap_counts()):
            h.update(np.ascontiguousarray(a).tobytes())
    eng.sweep(2); eng.sync()
    h.update(np.ascontiguousarray(eng.states()).tobytes())
    acc = int(eng.naccept.sum() - eng.Nc)
    print("ok %s %s | %s accepts %d swaps %d" % (h.hexdigest(), eng.sweep_kernel_name, eng.step_kernel_name, acc, int(eng.swap_counts()[1].sum())))
    eng.close()
