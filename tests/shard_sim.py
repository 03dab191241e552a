"""In-process simulator of the sharded step: G shards of one ladder driven in lockstep through the very phases
ptmcmc_amd.parallel.ShardedLadder runs under torch.distributed, with the messages delivered by plain copies."""
from ptmcmc_amd.parallel import ShardedLadder, shard_bounds


class _NoDist:
    pass


def build(backends, halo=4):
    world = len(backends)
    sizes = [b.nloc for b in backends]
    return [ShardedLadder(b, _NoDist(), r, world, halo=halo, sizes=sizes) for r, b in enumerate(backends)]


def _deliver(ladders, kind, copy):
    for r, lad in enumerate(ladders):
        msgs = lad.halo_messages() if kind == "halo" else lad.row_messages()
        for idx, (send, _recv, peer) in enumerate(msgs):
            if peer is None:
                continue
            # my message idx 0 goes up and lands in the peer's "from below" buffer (its message idx 1's recv), and v.v.
            pm = ladders[peer].halo_messages() if kind == "halo" else ladders[peer].row_messages()
            copy(pm[1 - idx][1], send)


def step(ladders, copy, n=1):
    for _ in range(n):
        for lad in ladders:
            lad.stage_halos()
        for lad in ladders:
            lad.b.sync()
        _deliver(ladders, "halo", copy)
        for lad in ladders:
            lad.decide()
        for lad in ladders:
            lad.b.sync()
        _deliver(ladders, "rows", copy)
        for lad in ladders:
            lad.finish()
        for lad in ladders:
            lad.b.sync()
