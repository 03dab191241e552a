"""In-process simulator of the sharded step: G shards of one ladder driven in lockstep through the very phases
ptmcmc_amd.parallel.ShardedLadder runs under torch.distributed, with the messages delivered by plain copies."""
from ptmcmc_amd.parallel import ShardedLadder, shard_bounds


class _NoDist:
    pass


def build(backends, halo=4, recover=False):
    world = len(backends)
    sizes = [b.nloc for b in backends]
    return [ShardedLadder(b, _NoDist(), r, world, halo=halo, sizes=sizes, recover=recover) for r, b in enumerate(backends)]


def _deliver(ladders, kind, copy):
    for r, lad in enumerate(ladders):
        msgs = lad.halo_messages() if kind == "halo" else lad.row_messages()
        for idx, (send, _recv, peer) in enumerate(msgs):
            if peer is None:
                continue
            # my message idx 0 goes up and lands in the peer's "from below" buffer (its message idx 1's recv), and v.v.
            pm = ladders[peer].halo_messages() if kind == "halo" else ladders[peer].row_messages()
            copy(pm[1 - idx][1], send)


def step_gathered(ladders, copy, n=1):
    """evolving ladders: every shard's llikes / lpriors to every shard (the all-gather, by plain copies), then the usual rounds"""
    for _ in range(n):
        for lad in ladders:
            lad.stage_gather()
        for lad in ladders:
            lad.b.sync()
        for lad in ladders:
            slab2 = 2 * lad.maxn * lad.b.W
            for r, src in enumerate(ladders):
                copy(lad.b.sub(lad.g_recv, r * slab2, slab2), src.g_send)
            lad.assemble_gathered()
        for lad in ladders:
            lad.decide_gathered()
        for lad in ladders:
            lad.b.sync()
        _deliver(ladders, "rows", copy)
        for lad in ladders:
            lad.finish()
        for lad in ladders:
            lad.b.sync()


def _recover(ladders, copy):
    """the second pass of ShardedLadder._recover_step in lockstep: every shard must have left the SAME ladders alone"""
    counts = [lad.redo_pending() for lad in ladders]
    assert len(set(counts)) == 1, counts
    if not counts[0]:
        return 0
    for lad in ladders:
        lad.recovered += counts[0]
        lad.stage_gather()
    for lad in ladders:
        lad.b.sync()
    for lad in ladders:
        slab2 = 2 * lad.maxn * lad.b.W
        for r, src in enumerate(ladders):
            copy(lad.b.sub(lad.g_recv, r * slab2, slab2), src.g_send)
        lad.assemble_gathered()
    for lad in ladders:
        lad.redo()
    return counts[0]


def step(ladders, copy, n=1):
    if ladders and ladders[0].gathered:
        return step_gathered(ladders, copy, n)
    for _ in range(n):
        for lad in ladders:
            lad.stage_halos()
        for lad in ladders:
            lad.b.sync()
        _deliver(ladders, "halo", copy)
        for lad in ladders:
            lad.decide()
        _recover(ladders, copy)
        for lad in ladders:
            lad.b.sync()
        _deliver(ladders, "rows", copy)
        for lad in ladders:
            lad.finish()
        for lad in ladders:
            lad.b.sync()


def step_overlapped(ladders, copy, n=1):
    """the order ShardedLadder.step uses to hide both message rounds (interior A | install | boundary | halos | interior B),
    driven in lockstep with the messages delivered by plain copies"""
    for lad in ladders:
        if lad._halo_reqs is None:
            lad.stage_halos()
            lad._halo_reqs = []
    for lad in ladders:
        lad.b.sync()
    if not getattr(ladders[0], "_sim_halos_delivered", False):
        _deliver(ladders, "halo", copy)
        for lad in ladders:
            lad._sim_halos_delivered = True
    for _ in range(n):
        plans = [lad.sweep_plan() for lad in ladders]
        for lad in ladders:
            lad.decide()
        _recover(ladders, copy)
        for lad, (bottom, int_a, int_b, top) in zip(ladders, plans):
            lad.b.sweep_rungs(int_a[0], int_a[1], False)
        for lad in ladders:
            lad.b.sync()
        _deliver(ladders, "rows", copy)
        for lad, (bottom, int_a, int_b, top) in zip(ladders, plans):
            lad.b.install(lad.recv_below, lad.recv_above)
            lad.b.sweep_rungs(bottom[0], bottom[1], False)
            lad.b.sweep_rungs(top[0], top[1], False)
            lad.stage_halos()
        for lad in ladders:
            lad.b.sync()
        _deliver(ladders, "halo", copy)
        for lad, (bottom, int_a, int_b, top) in zip(ladders, plans):
            lad.b.sweep_rungs(int_b[0], int_b[1], True)
        for lad in ladders:
            lad.b.sync()
