"""Child process of tests/test_gpu_sharding.py: 1024 rungs in 8 EngineShards on torch streams against one engine.

torch is imported FIRST: the torch wheel carries its own HIP runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7) and
asks for it by the name "libamdhip64.so", so a process that loaded libptm_engine.so (which needs /opt/rocm's libamdhip64.so.7)
before torch ends up with two HIP runtimes, and the second one finds no GPU.  With torch first, the engine library binds
to the runtime torch loaded (same soname) and both share streams and memory -- the order bench.py's N-GPU path uses too."""
import os
import sys

import torch  # noqa: E402  (before anything loads libptm_engine.so)

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np

import shard_sim
from ptmcmc_amd import engine as E
from ptmcmc_amd.parallel import DEFAULT_HALO, EngineShard, shard_bounds
from ptmcmc_amd.problems import GaussianProblem


def main(overlap):
    D, Nt, W, G, sr = 32, 1024, 64, 8, 0.1
    dev = torch.device("cuda", 0)
    torch.cuda.init()
    pr = GaussianProblem(D, Nt, 1e9)
    ref = E.Engine(D, Nt, W, swap_rate=sr)
    pr.configure(ref, E.PROP_LOWER)
    ref.init_from_prior()
    x0 = ref.states()
    shards, backs = [], []
    for g in range(G):
        r0, n = shard_bounds(Nt, G, g)
        s = torch.cuda.Stream(device=dev)
        e = E.Engine(D, Nt, W, swap_rate=sr, rung_begin=r0, rung_count=n, stream=s.cuda_stream)
        pr.configure(e, E.PROP_LOWER)
        e.set_states(x0[r0 * W:(r0 + n) * W])
        shards.append(e)
        backs.append(EngineShard(e, torch, dev, s))
    lads = shard_sim.build(backs, halo=DEFAULT_HALO)

    def copy(dst, src):                 # the wire: every shard was synchronised before, and is again before it reads
        dst.copy_(src)
        torch.cuda.synchronize()
    nsteps = 30
    for k in range(0, nsteps, 5):
        ref.step(5)
        (shard_sim.step_overlapped if overlap else shard_sim.step)(lads, copy, 5)
        xs = np.concatenate([e.states() for e in shards])
        assert np.array_equal(xs, ref.states()), "states differ after step %d" % (k + 5)
    for name in ("llike", "lprior", "ntries", "naccept", "nhist", "last_type"):
        assert np.array_equal(np.concatenate([getattr(e, name) for e in shards]), getattr(ref, name)), name
    t = sum(e.swap_counts()[0] for e in shards)
    a = sum(e.swap_counts()[1] for e in shards)
    rt, ra = ref.swap_counts()
    assert np.array_equal(t, rt) and np.array_equal(a, ra) and a.sum() > 0
    for g in range(G - 1):              # rows did cross every boundary
        b = shard_bounds(Nt, G, g + 1)[0]
        assert ra[:, b - 1].sum() > 0, b
    for e in shards + [ref]:
        e.close()
    print("OK overlap=%d: %d accepted exchanges, %d across shard boundaries" % (
        overlap, int(ra.sum()), int(sum(ra[:, shard_bounds(Nt, G, g + 1)[0] - 1].sum() for g in range(G - 1)))))


if __name__ == "__main__":
    main(int(sys.argv[1]))
