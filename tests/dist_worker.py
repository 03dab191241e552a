"""One rank of the CPU (gloo) sharding test: runs its block of the ladder on the oracle stand-in and saves the result."""
import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle_lib as O
from oracle_shard import OracleShard
from ptmcmc_amd.parallel import ShardedLadder, shard_bounds
from ptmcmc_amd.problems import GaussianProblem


def make_ladder(D, Nt, W, sr, seed, evolve=0.0, cut=-1.0):
    pr = GaussianProblem(D, Nt, 1e3)
    pb = O.Problem(D)
    pb.set_bounds([0] * D, [0] * D, [0.0] * D, [0.0] * D)
    pb.set_prior(pr.types, pr.centers, pr.halfwidths)
    pb.set_gauss(pr.P, pr.like0)
    lad = O.Ladder(pb, pr.beta, W=W, swap_rate=sr)
    fac = pr.proposal_factors()
    lad.set_proposals([(O.PROP_DENSE, fac[r], 0.0) for r in range(Nt)])
    lad.use_philox(seed)
    lad.init_from_prior(seed)
    if evolve > 0:
        lad.evolve_temps(evolve, cut)
    return lad


if __name__ == "__main__":
    D, Nt, W, nsteps, halo = (int(v) for v in sys.argv[1:6])
    sr, out = float(sys.argv[6]), sys.argv[7]
    evolve, cut = (float(sys.argv[8]), float(sys.argv[9])) if len(sys.argv) > 9 else (0.0, -1.0)
    recover = None if len(sys.argv) <= 10 else bool(int(sys.argv[10]))
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    seed = 0x5EED0001
    lad = make_ladder(D, Nt, W, sr, seed, evolve, cut)   # every rank draws the same global start, keeps its block
    r0, nloc = shard_bounds(Nt, world, rank)
    sh = OracleShard(lad, r0, nloc, seed)
    sl = ShardedLadder(sh, dist, rank, world, halo=halo, recover=recover)
    sl.step(10)            # as bench.py drives it: several calls, halos left in flight between them, drained at the end
    sl.drain()
    sl.step(nsteps - 10)
    sl.drain()
    sh.sync()
    np.savez(out % rank, x=sh.local(sh.x), ll=sh.local(sh.ll), nhist=sh.local(sh.nhist), nacc=sh.local(lad.naccept),
             st=sh.swap_try, sa=sh.swap_acc, r0=r0, nloc=nloc, betaw=lad.betaw, recovered=sl.recovered)
    dist.barrier()
    dist.destroy_process_group()
