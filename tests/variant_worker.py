"""Runs one small bit-exact comparison of the engine against the CPU checker in a process of its own, so that an environment
switch of the engine (read once per process: PTM_LEAN_PIPE, PTM_COMPACT, PTM_FORCE_VALU, PTM_FUSED ...) can be exercised by
the test suite.  usage: python variant_worker.py D Nt W kind nsteps expected_kernel_substring"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parity_util as PU
from ptmcmc_amd import engine as E

if __name__ == "__main__":
    D, Nt, W = (int(v) for v in sys.argv[1:4])
    kind = {"lower": E.PROP_LOWER, "dense": E.PROP_DENSE, "diag": E.PROP_DIAG}[sys.argv[4]]
    if sys.argv[5] == "giveup":
        # the persistent ladder kernel under PTM_LADDER_SPIN_US=0: some launch gives up, commits nothing, and its steps are repeated
        # on the two-launch path (tests/test_gpu_parity.py::test_persistent_ladder_kernel_that_gives_up_leaves_nothing_behind)
        pr, eng, lad = PU.make_pair(D, Nt, W, 1e6, kind=kind, swap_rate=0.1)
        assert eng.step_kernel_name.startswith("ladder_persistent_kernel"), eng.step_kernel_name
        done = 0
        for n in (300, 1, 200, 40):                # asynchronous launches behind each other, then a look
            eng.step(n); eng.step(n); done += 2 * n
            eng.sync(); lad.pt_step(2 * n)
            PU.assert_same_state(eng, lad, "after %d steps" % done)
            assert eng.step_count == done
        t, a = eng.swap_counts()
        assert (t == lad.swap_count).all() and (a == lad.swap_accept_count).all()
        pairs, acc = eng.last_swaps()
        assert (pairs == lad.last_pairs).all() and (acc == lad.last_accept).all()
        st = eng.ladder_stats()
        assert st["disabled"] == (st["fallbacks"] > 0) and st["fallbacks"] <= 1, st
        assert not eng.step_kernel_name.startswith("ladder_persistent_kernel") or st["fallbacks"] == 0
        print("ok fallbacks=%d launches=%d" % (st["fallbacks"], st["launches"]))
        eng.close()
        sys.exit(0)
    nsteps, want = int(sys.argv[5]), sys.argv[6]
    pr, eng, lad = PU.make_pair(D, Nt, W, 1e3, kind=kind, swap_rate=0.3)
    assert want in eng.sweep_kernel_name or want in eng.step_kernel_name, (eng.sweep_kernel_name, eng.step_kernel_name)
    for k in range(3):
        eng.step(nsteps); eng.sync(); lad.pt_step(nsteps)
        PU.assert_same_state(eng, lad, "after %d steps" % (nsteps * (k + 1)))
    eng.sweep(2); eng.sync(); lad.sweep(2)
    PU.assert_same_state(eng, lad, "after plain sweeps")
    t, a = eng.swap_counts()
    assert (t == lad.swap_count).all() and (a == lad.swap_accept_count).all()
    print("ok %s | %s accepts %d" % (eng.sweep_kernel_name, eng.step_kernel_name, int(eng.naccept.sum() - eng.Nc)))
    eng.close()
