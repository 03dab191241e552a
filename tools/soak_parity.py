"""Long bit-exact runs of the engine against the CPU checker (test infrastructure: uses tests/oracle_lib.py), one per
kernel family, ladders evolving: rare branches (exchange overflow paths, table edges, twice-touched rungs) get their turn.
usage (GPU box): python tools/soak_parity.py [steps]"""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import parity_util as PU
from ptmcmc_amd import engine as E

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
cases = [(6, 40, 64, E.PROP_DENSE, 0.3, 0.02, "general kernel"), (5, 9, 5, E.PROP_DENSE, 0.3, 0.02, "lanes kernel (8)"),
         (14, 24, 3, E.PROP_LOWER, 0.2, 0.01, "lanes kernel"),
         (32, 10, 320, E.PROP_LOWER, 0.3, 0.01, "MFMA kernel"), (40, 8, 2, E.PROP_DIAG, 0.4, 0.0, "lanes kernel, 64-dim rows"),
         (90, 6, 2, E.PROP_LOWER, 0.4, 0.01, "lanes kernel, 128-dim rows"),
         (32, 6, 1024, E.PROP_LOWER, 0.3, 0.0, "MFMA kernel, compacted"), (28, 6, 1024, E.PROP_DENSE, 0.3, 0.01, "MFMA kernel, box-bounds build, compacted")]
# the persistent ladder kernel (round 4): its plain build, the evolving one, and the one with everything the sampler switches on
cases += [(32, 96, 2, E.PROP_LOWER, 0.25, 0.0, "persistent ladder kernel"), (32, 96, 2, E.PROP_LOWER, 0.25, 0.02, "persistent ladder kernel, evolving"),
          (20, 60, 3, E.PROP_DENSE, 0.3, 0.02, "persistent ladder kernel, everything")]
for D, Nt, W, kind, sr, ev, what in cases:
    lean = what == "MFMA kernel, compacted" or what.startswith("persistent ladder kernel") and "everything" not in what
    hist = "everything" in what
    pr, eng, lad = PU.make_pair(D, Nt, W, 1e4, kind=kind, swap_rate=sr, one_d_frac=None if lean else 0.2, add_every_n=50 if hist else 1,
                                history_cap=(2 * steps // 50 + 8) if hist else 0)
    if what.startswith("persistent ladder kernel"):
        assert eng.step_kernel_name.startswith("ladder_persistent_kernel"), eng.step_kernel_name
    if W >= 1024:
        steps_case = min(steps, 600)   # (the checker walks 6144 chains per step on one core)
    else:
        steps_case = steps
    if ev:
        eng.set_evolve_temps(ev); lad.evolve_temps(ev)
    t0 = time.time()
    done = 0
    while done < steps_case:
        n = min(500 if W < 1024 else 100, steps_case - done)
        eng.step(n); eng.sync(); lad.pt_step(n)
        done += n
        PU.assert_same_state(eng, lad, "%s after %d steps" % (what, done))
        assert np.array_equal(eng.invtemps(), lad.betaw)
        print("  %-28s %6d steps ok (%.0fs)" % (what, done, time.time() - t0), flush=True)
    if hist:
        PU.assert_same_history_and_map(eng, lad, 2 * steps // 50 + 8)
    t, a = eng.swap_counts()
    print("%s: D=%d %dx%d, %d steps bit-identical; kernel %s; MH accept %.3f, swap accept %.3f" %
          (what, D, Nt, W, steps_case, eng.sweep_kernel_name, (eng.naccept.sum() - eng.Nc) / max(1, eng.ntries.sum() - eng.Nc), a.sum() / max(1, t.sum())) + "  [step kernel %s]" % eng.step_kernel_name, flush=True)
    eng.close()

# differential evolution from the device history (round 4): the sampler's default set on an evolving ladder with history and MAP, in the
# persistent ladder kernel (FL 15), the lanes kernel (64-dimensional rows) and the general kernel; long runs fill the ring, reach the
# snooker move's retries and the early rows' picks
import test_gpu_de as TD
de_steps = min(steps, 1500)
for D, Nt, W, kind, snk, ign, what in [(20, 48, 2, E.PROP_DENSE, 0.1, 0.0, "differential evolution, persistent ladder kernel"),
                                       (40, 8, 2, E.PROP_DIAG, 0.3, 0.2, "differential evolution, lanes kernel, 64-dim rows"),
                                       (6, 40, 64, E.PROP_LOWER, 0.1, 0.0, "differential evolution, general kernel")]:
    cap = 2 * de_steps + 8
    pr, eng, lad = TD._pair(D, Nt, W, kind, 1, snk, 12, 6, cap, de_share=0.8, ignore=ign)
    eng.set_evolve_temps(0.01); lad.evolve_temps(0.01)
    t0 = time.time()
    done = 0
    while done < de_steps:
        n = min(250, de_steps - done)
        eng.step(n); eng.sync(); lad.pt_step(n)
        done += n
        PU.assert_same_state(eng, lad, "%s after %d steps" % (what, done))
        assert np.array_equal(eng.invtemps(), lad.betaw)
        print("  %-28s %6d steps ok (%.0fs)" % (what, done, time.time() - t0), flush=True)
    PU.assert_same_history_and_map(eng, lad, cap)
    lt = np.unique(eng.last_type)
    print("%s: D=%d %dx%d, %d steps bit-identical (states, every saved row, MAPs, temperatures); step kernel %s; accepted types %s" %
          (what, D, Nt, W, de_steps, eng.step_kernel_name, lt.tolist()), flush=True)
    eng.close()
