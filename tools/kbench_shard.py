#!/usr/bin/env python3
"""Per-rank cost of one sharded step at the weak-scaling bench sizes, measured on ONE GPU: a middle shard (128 of 1024
rungs) with walkers x G ladders and dummy halo / row buffers (timing only -- the neighbours' data is zeros)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ptmcmc_amd import engine as E
from ptmcmc_amd.problems import GaussianProblem

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=8)
ap.add_argument("--walkers", type=int, default=4096)
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
D, Nt, G = 32, 1024, a.gpus
W = a.walkers * G
nloc = Nt // G
r0 = nloc * (G // 2) if G > 1 else 0
pr = GaussianProblem(D, Nt, 1e9)
eng = E.Engine(D, Nt, W, rung_begin=r0, rung_count=nloc, add_every_n=100, time_kernels=True)
pr.configure(eng, E.PROP_LOWER)
eng.init_from_prior()
from ptmcmc_amd.parallel import DEFAULT_HALO
H = DEFAULT_HALO
first, last = r0 == 0, r0 + nloc == Nt
bufs = {k: E.DeviceBuffer(8 * n) for k, n in dict(lb=W, la=H * W, su=eng.exchange_buffer_doubles, sd=eng.exchange_buffer_doubles,
                                                   rb=eng.exchange_buffer_doubles, ra=eng.exchange_buffer_doubles).items()}
zero = np.zeros(eng.exchange_buffer_doubles)
for k in ("rb", "ra"):   # empty boundary messages
    bufs[k].copy_from(zero.ctypes.data, bufs[k].nbytes)
ll = eng.llike
fill = np.full(H * W, float(np.median(ll)))
for k in ("lb", "la"):
    bufs[k].copy_from(fill.ctypes.data, bufs[k].nbytes)
# the launch sequence of ShardedLadder.step (interior A | install | boundary rungs | interior B), without the messages
nloc_ = nloc
nb = 0 if first else min(H, nloc_)
nt = 0 if last or nloc_ <= nb else 1
lo, hi = nb, nloc_ - nt
mid = lo + (hi - lo) // 2
def step():
    eng.exchange_decide(None if first else bufs["lb"].ptr, None if last else bufs["la"].ptr, H, None if last else bufs["su"].ptr,
                        None if first else bufs["sd"].ptr)
    if G == 1:
        eng.exchange_finish_and_sweep(None, None)
        return
    eng.sweep_rungs(lo, mid - lo, False)
    eng.exchange_install(None if first else bufs["rb"].ptr, None if last else bufs["ra"].ptr)
    eng.sweep_rungs(0, nb, False)
    eng.sweep_rungs(nloc_ - nt, nt, False)
    eng.sweep_rungs(mid, hi - mid, True)
for _ in range(3):
    step()
eng.sync()
eng.kernel_times()
eng.timer_start()
for _ in range(a.reps):
    step()
ms = eng.timer_stop() / a.reps
kt = eng.kernel_times()
ksum = kt.sum() / a.reps
print("G=%d shard %d..%d x %d walkers (%d chains): step %.4f ms, sweep launches %.4f ms, exchange kernels %.4f ms"
      % (G, r0, r0 + nloc, W, nloc * W, ms, ksum, ms - ksum), flush=True)
