import os, sys
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np
import parity_util as PU
from ptmcmc_amd import engine as E
D, Nt, W = 32, 64, 2
pr, eng, lad = PU.make_pair(D, Nt, W, 1e3, kind=E.PROP_LOWER, swap_rate=0.3)
print(eng.step_kernel_name)
for k in range(12):
    n = 1 if k < 8 else 4
    eng.step(n); eng.sync(); lad.pt_step(n)
    xe = eng.states(); xo = PU.to_engine_order(lad.x, Nt, W)
    bad = np.argwhere((xe != xo).any(axis=1)).ravel()
    pairs, acc = eng.last_swaps()
    print("step", k, "n", n, "bad chains", bad[:12].tolist(), "pairs eq", np.array_equal(pairs, lad.last_pairs), "acc eq", np.array_equal(acc, lad.last_accept))
    if len(bad):
        w = bad[0] % W
        print(" oracle pairs w:", [int(v) for v in lad.last_pairs[w] if v >= 0], "acc", [int(a) for v, a in zip(lad.last_pairs[w], lad.last_accept[w]) if v >= 0])
        print(" engine pairs w:", [int(v) for v in pairs[w] if v >= 0])
        break
