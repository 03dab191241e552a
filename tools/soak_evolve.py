import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from ptmcmc_amd import engine as E
from ptmcmc_amd.problems import GaussianProblem
for (D, Nt, W) in ((32, 64, 64), (16, 128, 3)):
    pr = GaussianProblem(D, Nt, 1e6)
    e = E.Engine(D, Nt, W, swap_rate=0.1, add_every_n=10, history_rungs=2, history_capacity=64, map_rungs=1)
    pr.configure(e, E.PROP_LOWER)
    e.set_evolve_temps(0.01)
    e.init_from_prior()
    t0 = time.time()
    for k in range(20):
        e.step(1000); e.sync()
    b = e.invtemps()
    t, a = e.swap_counts()
    print(D, Nt, W, "20000 steps in %.1fs" % (time.time() - t0), "ordered", bool((np.diff(b, axis=1) < 0).all()), "finite", bool(np.isfinite(b).all()),
          "ends", b[:, 0].min(), b[:, -1].max(), "swap acc rate min/mean/max over pairs (ladder 0): %.3f %.3f %.3f" % tuple(f((a[0] / np.maximum(t[0], 1))) for f in (np.min, np.mean, np.max)),
          "sum gaps", float((b[0, :-1] - b[0, 1:]).sum()), 1 - b[0, -1], "kernel", e.sweep_kernel_name)
    acc0 = a[0] / np.maximum(t[0], 1)
    print("  initial geometric ladder vs evolved, rung 1,2,Nt/2:", pr.beta[[1, 2, Nt // 2]], b[0, [1, 2, Nt // 2]])
    e.close()
