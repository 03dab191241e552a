import sys, os
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import parity_util as PU
from ptmcmc_amd import engine as E
cases = [  # D, Nt, W, kind, swap_rate, evolve, odf
    (3, 6, 2, E.PROP_DENSE, 0.0, 0.0, None), (3, 6, 64, E.PROP_DIAG, 0.0, 0.01, None), (5, 4, 3, E.PROP_LOWER, 0.75, 0.0, 0.3),
    (5, 4, 64, E.PROP_LOWER, 0.9, 0.05, None), (32, 3, 64, E.PROP_LOWER, 0.6, 0.02, None), (17, 2, 5, E.PROP_DENSE, 0.5, 0.05, None),
    (1, 3, 64, E.PROP_DIAG, 0.4, 0.02, None), (64, 3, 2, E.PROP_DIAG, 0.5, 0.02, 0.5), (8, 1, 3, E.PROP_DENSE, 0.3, 0.0, None),
    (2, 1500, 1, E.PROP_DIAG, 0.3, 0.01, None), (4, 2000, 64, E.PROP_DIAG, 0.05, 0.0, None)]
for D, Nt, W, kind, sr, ev, odf in cases:
    pr, eng, lad = PU.make_pair(D, Nt, W, 1e3, kind=kind, swap_rate=sr, one_d_frac=odf)
    if ev and Nt > 1:
        eng.set_evolve_temps(ev); lad.evolve_temps(ev)
    for k in range(3):
        eng.step(7); eng.sync(); lad.pt_step(7)
        PU.assert_same_state(eng, lad, "D=%d Nt=%d W=%d sr=%g ev=%g" % (D, Nt, W, sr, ev))
        assert np.array_equal(eng.invtemps(), lad.betaw)
    t, a = eng.swap_counts()
    assert np.array_equal(t, lad.swap_count) and np.array_equal(a, lad.swap_accept_count)
    print("ok D=%d Nt=%d W=%d sr=%g ev=%g ms=%d kernel=%s tries=%d" % (D, Nt, W, sr, ev, eng.max_swaps, eng.sweep_kernel_name, t.sum()), flush=True)
    eng.close()
