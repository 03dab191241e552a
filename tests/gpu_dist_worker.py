"""One RANK of the multi-process GPU test (tests/test_gpu_sharding.py): its block of the ladder on a real engine shard
(ptmcmc_amd.parallel.EngineShard, torch tensors, an explicit torch stream), the sharded step of ptmcmc_amd.parallel.ShardedLadder
with its messages between PROCESSES -- over gloo, because a one-GPU box cannot give RCCL two ranks (PTM_BENCH_REHEARSAL=1: every
rank on device 0, the engine's stream drained before a message starts, since gloo moves device buffers from the host without
regard for streams).  Everything but the transport is what bench.py --gpus N runs.  Saves its block for the parent to compare
with one engine holding the whole ladder."""
import os
import sys

import torch  # noqa: E402  (before anything loads libptm_engine.so: tests/torch_shard_worker.py says why)
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np

from ptmcmc_amd import engine as E
from ptmcmc_amd.parallel import EngineShard, ShardedLadder, shard_bounds
from ptmcmc_amd.problems import GaussianProblem

if __name__ == "__main__":
    D, Nt, W, nsteps, halo = (int(v) for v in sys.argv[1:6])
    sr, out = float(sys.argv[6]), sys.argv[7]
    evolve = float(sys.argv[8]) if len(sys.argv) > 8 else 0.0
    hist_every = int(sys.argv[9]) if len(sys.argv) > 9 else 0      # > 0: record the history (every hist_every-th add) and the MAP of the shard's rungs
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ["PTM_BENCH_REHEARSAL"] = "1"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    pr = GaussianProblem(D, Nt, 1e6)
    ref = E.Engine(D, Nt, W, swap_rate=sr)            # every rank draws the same global start, keeps its block
    pr.configure(ref, E.PROP_LOWER)
    ref.init_from_prior()
    x0 = ref.states()
    ref.close()
    r0, n = shard_bounds(Nt, world, rank)
    stream = torch.cuda.Stream(device=dev)
    # (a shard below the ladder's top records its LOWER rungs: the row a rung holds between its two exchanges of a step comes from
    #  the rung above -- through a run of accepted picks possibly from several rungs above --, and a row that still sits in the
    #  neighbour shard at that moment is refused loudly, error bit 16; half a shard is a run no step ever sees)
    hr = 0 if not hist_every else (n if rank == world - 1 else max(1, n // 2))
    eng = E.Engine(D, Nt, W, swap_rate=sr, rung_begin=r0, rung_count=n, stream=stream.cuda_stream, add_every_n=max(1, hist_every),
                   history_rungs=hr, history_capacity=(2 * nsteps // max(1, hist_every) + 8) if hr else 0, map_rungs=hr)
    pr.configure(eng, E.PROP_LOWER)
    if evolve > 0:
        eng.set_evolve_temps(evolve)      # evolving ladders: ShardedLadder.step_gathered (an all-gather of the llikes per step)
    eng.set_states(x0[r0 * W:(r0 + n) * W])
    lad = ShardedLadder(EngineShard(eng, torch, dev, stream), dist, rank, world, halo=halo)
    lad.step(7)              # as bench.py drives it: several calls, halos left in flight between them, drained at the end
    lad.drain()
    lad.step(nsteps - 7)
    lad.drain()
    eng.sync()
    t, a = eng.swap_counts()
    extra = {}
    if hr:
        h, m = eng.history(), eng.map()
        extra = dict(hist_rungs=hr, **{"h_" + k: v for k, v in h.items()}, **{"m_" + k: v for k, v in m.items()})
    np.savez(out % rank, x=eng.states(), ll=eng.llike, nacc=eng.naccept, nhist=eng.nhist, st=t, sa=a, invtemps=eng.invtemps(), recovered=lad.recovered, **extra)
    dist.barrier()
    eng.close()
    dist.destroy_process_group()
