"""Ladder sharding across the GPUs of one node: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" on CPU for the protocol tests).

The reference distributes rungs cyclically over MPI ranks and all-gathers every state, llike and invtemp each step
(chain.cc:1298-1309,1433-1435,1905-1967).  Here the ladder is cut into CONTIGUOUS rung blocks, so only the G-1 rung
pairs that straddle two GPUs ever move data, and -- because exchanges propagate only downwards within one step
(chain.cc:1417-1418) -- every shard can replay the decisions that concern it from a small llike halo:

  per step and per shard boundary
    1. llike halos, point-to-point:   top rung's llike      -> upper neighbour   (W doubles)
                                      bottom H rungs' llike -> lower neighbour   (H*W doubles)
    2. exchange kernel (ptm_exchange_decide): replays the candidate draws of every ladder (replicated counter RNG),
       decides, names the row moves inside the shard and packs the rows that leave it
    3. boundary rows, point-to-point: one (state, llike, lprior) row per walker and direction, only neighbours talk
    4. fused MH sweep (ptm_exchange_finish_and_sweep) installs arrivals and advances every untouched rung

Both message rounds are hidden behind arithmetic (ShardedLadder.step): the Metropolis moves of a step do not depend on
the rows in flight (their landing slots are exchanged rungs, which make no move), so half of the interior rungs are swept
while the boundary messages travel; then arrivals are installed, the boundary rungs swept, their llikes sent off as the
NEXT step's halos, and the other half of the interior swept while those travel.

There is no collective on the data path; swap bookkeeping stays where the reference keeps it (host side, from the
per-pair counters each shard owns).  Chains are bit-identical for any number of shards: all random streams are keyed by
global (seed, walker, rung, step).
"""
import contextlib
import os
import sys
import time

import numpy as np

# Depth of the llike halo a shard receives from the shard above it.  The exchange kernel cannot decide a shard's top
# pairs when the step's SURVIVING picks (chain.cc:1417-1418; accepted or not) cover H+1 consecutive pairs starting at the
# shard's top rung: it then raises PTM_ERR_FAR_MOVE (loud, never silently wrong).  A run of k consecutive surviving picks
# needs k picked rungs whose first picks come in descending rung order: probability <= swap_rate^k / k! per boundary,
# ladder and step (measured on the ladder streams themselves by tests/test_halo_depth.py: 9.1e-2, 4.2e-3, 1.3e-4, 3.1e-6,
# 2e-8 for k = 1..5 at swap_rate 0.1).  H = 4 -- the first round's default -- fails every ~2e7 boundary-ladder-steps, i.e.
# several times in ONE 8-GPU bench run (131072 ladders x 7 boundaries x 400 steps = 3.7e8); H = 8 (round 2) makes it 0.1^9 / 9! =
# 2.8e-15: < 1e-6 per bench run but still ~3e-3 per 10^6-step production run of that size -- a run that dies after days.
# H = 12: 0.1^13 / 13! = 1.6e-23 per boundary, ladder and step, 1.5e-11 per such production run -- never, for any run anybody
# will make (and every shard needs 12 rungs at most from the shard above, so shards of 12 rungs and more never look past their
# neighbour).  Cost: H x W doubles per boundary and step (12.6 MB at W = 131072: ~0.25 ms of xGMI), sent while the interior
# rungs are swept.  The condition is a property of the candidate draws alone (every shard replays them): a step that would
# need more is still detected and reported loudly (PTM_ERR_FAR_MOVE), never decided blindly.
DEFAULT_HALO = 12


def shard_bounds(n_rungs, world, rank):
    """contiguous block of rungs of `rank`: sizes differ by at most one, lower ranks take the remainder"""
    base, rem = divmod(n_rungs, world)
    lo = rank * base + min(rank, rem)
    return lo, base + (1 if rank < rem else 0)


class EngineShard:
    """Backend of ShardedLadder on a GPU: a ptmcmc_amd Engine + torch tensors for the message buffers."""

    def __init__(self, engine, torch, device, stream=None):
        """`stream`: the torch.cuda.Stream the engine was created on (Engine(stream=s.cuda_stream)).  Every message of the
        sharded step is issued with that stream current, so the order engine kernel -> RCCL send / RCCL receive -> engine
        kernel is the order of ONE stream (c10d's NCCL operations wait for the current stream when they are issued and
        req.wait() makes the current stream wait for them) -- not a side effect of the legacy NULL stream."""
        from . import engine as _E
        if _E.TORCH_LOADED_FIRST is False:
            # the torch wheel carries its own HIP runtime and loads it by a name the dynamic loader does not match with the
            # /opt/rocm one libptm_engine.so already brought in: the process would hold two runtimes, the second without a GPU
            raise RuntimeError("import torch before the first ptmcmc_amd engine call in a process that uses both "
                               "(libptm_engine.so then binds to the HIP runtime torch loaded)")
        self.e, self.torch, self.device, self.stream = engine, torch, device, stream
        self.W, self.nloc, self.r0, self.Nt = engine.W, engine.nloc, engine.r0, engine.Nt
        self.row_doubles = engine.exchange_buffer_doubles

    def stream_context(self):
        return self.torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def alloc(self, n):
        with self.stream_context():     # (the caching allocator ties a block to the stream it was allocated on)
            return self.torch.zeros(n, dtype=self.torch.float64, device=self.device)

    def copy_llike(self, first_rung, n_rungs, dst):
        self.e.copy_llike(first_rung, n_rungs, dst.data_ptr())

    def exchange_decide(self, ll_below, ll_above, halo, send_up, send_down):
        p = lambda t: None if t is None else t.data_ptr()
        self.e.exchange_decide(p(ll_below), p(ll_above), halo, p(send_up), p(send_down))

    def finish_and_sweep(self, recv_below, recv_above):
        p = lambda t: None if t is None else t.data_ptr()
        self.e.exchange_finish_and_sweep(p(recv_below), p(recv_above))

    def install(self, recv_below, recv_above):
        p = lambda t: None if t is None else t.data_ptr()
        self.e.exchange_install(p(recv_below), p(recv_above))

    def sweep_rungs(self, first, n, closes_step):
        self.e.sweep_rungs(first, n, closes_step)

    @property
    def can_overlap(self):
        return self.e.hist_rungs == 0 and self.e.map_rungs == 0   # (a recorded exchanged rung reads its final row: needs the arrivals first)

    @property
    def gathered(self):
        """evolving ladders: the exchange phase needs the whole ladder's llikes (ShardedLadder.step_gathered)"""
        return bool(getattr(self.e, "_evolving", False))

    @property
    def needs_lprior(self):
        return getattr(self.e, "_evolve_cut", -1.0) >= 0

    def copy_lprior(self, first_rung, n_rungs, dst):
        self.e.copy_lprior(first_rung, n_rungs, dst.data_ptr())

    @staticmethod
    def sub(buf, off, n):
        return buf[off:off + n]

    @staticmethod
    def dcopy(dst, src):
        dst.copy_(src)

    def set_shard_map(self, sizes, halo):
        self.e.set_shard_map(sizes, halo)

    def exchange_redo_count(self):
        return self.e.exchange_redo_count()

    def exchange_redo(self, ll_all, lp_all, send_up, send_down):
        p = lambda t: None if t is None else t.data_ptr()
        self.e.exchange_redo(p(ll_all), p(lp_all), p(send_up), p(send_down))

    def exchange_decide_gathered(self, ll_all, lp_all, send_up, send_down):
        p = lambda t: None if t is None else t.data_ptr()
        self.e.exchange_decide_gathered(p(ll_all), p(lp_all), p(send_up), p(send_down))

    def sync(self):
        self.e.sync()

    def before_messages(self):
        """Rehearsal over gloo only (PTM_BENCH_REHEARSAL=1, bench_main): gloo reads and writes device buffers from the host with
        no regard for streams, so everything the engine has queued must have run before a message starts.  RCCL needs none of
        this: its operations are ordered with the engine's stream (see __init__)."""
        if os.environ.get("PTM_BENCH_REHEARSAL", "") == "1":
            self.e.sync_stream() if hasattr(self.e, "sync_stream") else self.torch.cuda.synchronize()


class ShardedLadder:
    """Drives one shard of a ladder that is spread over `world` ranks.  `backend` supplies the local compute
    (EngineShard on a GPU; the tests plug a CPU stand-in with the same five methods)."""

    def __init__(self, backend, dist, rank, world, halo=DEFAULT_HALO, sizes=None, recover=None):
        """recover: RECOVER from a run of surviving picks longer than the halo instead of failing on it (PTM_ERR_FAR_MOVE).  Every
        shard replays the step's candidate draws, so -- told the whole shard map -- every shard finds the same ladders that some
        shard could not decide and leaves them alone; the step then asks for their number (one wait on the device per step: what
        this costs), and if there are any, every rank gathers the whole ladder's llikes and decides those ladders from the full view
        (ptm_exchange_redo) before the boundary rows travel.  None: on when the halo is shallower than DEFAULT_HALO (at the default
        depth a blind shard is a 1e-23 event per boundary, ladder and step)."""
        self.b, self.dist, self.rank, self.world = backend, dist, rank, world
        W = backend.W
        sizes = sizes or [shard_bounds(backend.Nt, world, r)[1] for r in range(world)]
        self.up = rank + 1 if rank + 1 < world else None
        self.down = rank - 1 if rank > 0 else None
        # halo depth: what I receive from above is limited by the upper neighbour's size, what I send down by mine
        self.h_recv = min(halo, sizes[rank + 1]) if self.up is not None else 0
        self.h_send = min(halo, sizes[rank]) if self.down is not None else 0
        a = backend.alloc
        self.ll_top = a(W) if self.up is not None else None            # my top rung's llike  -> up
        self.ll_bottom = a(self.h_send * W) if self.down is not None else None  # my bottom rungs -> down
        self.ll_below = a(W) if self.down is not None else None        # <- from below
        self.ll_above = a(self.h_recv * W) if self.up is not None else None     # <- from above
        n = backend.row_doubles
        self.send_up = a(n) if self.up is not None else None
        self.recv_above = a(n) if self.up is not None else None
        self.send_down = a(n) if self.down is not None else None
        self.recv_below = a(n) if self.down is not None else None
        self._halo_reqs = None
        # evolving ladders (gathered form): this shard's llikes | lpriors padded to the largest shard, everybody's, and the
        # whole ladder's [Nt][W] views the exchange kernel reads
        self.sizes = list(sizes)
        self.gathered = bool(getattr(backend, "gathered", False))
        self.recover = (halo < DEFAULT_HALO) if recover is None else bool(recover)
        if self.gathered or world == 1 or not hasattr(backend, "set_shard_map"):
            self.recover = False
        self.recovered = 0            # ladders decided by the second pass so far
        if self.recover:
            backend.set_shard_map(self.sizes, halo)
        if self.gathered or self.recover:
            self.maxn = max(self.sizes)
            self.g_send = a(2 * self.maxn * W)
            self.g_recv = a(world * 2 * self.maxn * W)
            self.ll_all = a(backend.Nt * W)
            self.lp_all = a(backend.Nt * W)

    # -- evolving ladders: the exchange phase from the whole ladder's llikes (the reference's gather_llikes / gather_lposts,
    #    chain.cc:1433-1435,1950-1972: an all-gather per step), then the boundary rows between neighbours as ever
    def stage_gather(self):
        b, slab = self.b, self.maxn * self.b.W
        b.copy_llike(0, b.nloc, b.sub(self.g_send, 0, b.nloc * b.W))
        b.copy_lprior(0, b.nloc, b.sub(self.g_send, slab, b.nloc * b.W))

    def assemble_gathered(self):
        """the shards' slabs, unpadded, in rung order -> ll_all / lp_all"""
        b, W, slab, at = self.b, self.b.W, self.maxn * self.b.W, 0
        for r, n in enumerate(self.sizes):
            b.dcopy(b.sub(self.ll_all, at, n * W), b.sub(self.g_recv, r * 2 * slab, n * W))
            b.dcopy(b.sub(self.lp_all, at, n * W), b.sub(self.g_recv, r * 2 * slab + slab, n * W))
            at += n * W

    def decide_gathered(self):
        self.b.exchange_decide_gathered(self.ll_all, self.lp_all if getattr(self.b, "needs_lprior", True) else None, self.send_up, self.send_down)

    def step_gathered(self, n=1):
        with self._ctx():
            for _ in range(n):
                self.stage_gather()
                self._before_messages()
                if self.world > 1:
                    self.dist.all_gather_into_tensor(self.g_recv, self.g_send)
                else:
                    self.b.dcopy(self.g_recv, self.g_send)
                self.assemble_gathered()
                self.decide_gathered()
                self._exchange(self.row_messages())
                self.finish()

    # -- the step, in phases (the in-process shard simulator of the tests drives the same phases in lockstep)
    def stage_halos(self):
        b = self.b
        if self.up is not None:
            b.copy_llike(b.nloc - 1, 1, self.ll_top)
        if self.down is not None:
            b.copy_llike(0, self.h_send, self.ll_bottom)

    def halo_messages(self):
        """(send buffer, receive buffer, peer rank): my top rung's llike goes up, my bottom rungs' llikes go down"""
        return [(self.ll_top, self.ll_above, self.up), (self.ll_bottom, self.ll_below, self.down)]

    def decide(self):
        self.b.exchange_decide(self.ll_below, self.ll_above, self.h_recv, self.send_up, self.send_down)

    # -- recovery (see __init__): the ladders this step's halo pass left alone, on every rank the same
    def redo_pending(self):
        return self.b.exchange_redo_count() if self.recover else 0

    def redo(self):
        self.b.exchange_redo(self.ll_all, self.lp_all, self.send_up, self.send_down)

    def _recover_step(self):
        n = self.redo_pending()
        if not n:
            return
        self.recovered += n
        self.stage_gather()          # (the llikes of the ladders left alone are untouched: only their columns matter)
        self._before_messages()
        self.dist.all_gather_into_tensor(self.g_recv, self.g_send)
        self.assemble_gathered()
        self.redo()

    def row_messages(self):
        return [(self.send_up, self.recv_above, self.up), (self.send_down, self.recv_below, self.down)]

    def finish(self):
        self.b.finish_and_sweep(self.recv_below, self.recv_above)

    def _before_messages(self):
        f = getattr(self.b, "before_messages", None)
        if f:
            f()

    def _exchange(self, msgs):
        self._before_messages()
        ops = []
        for send, recv, peer in msgs:
            if peer is None:
                continue
            ops.append(self.dist.P2POp(self.dist.isend, send, peer))
            ops.append(self.dist.P2POp(self.dist.irecv, recv, peer))
        if ops:
            for r in self.dist.batch_isend_irecv(ops):
                r.wait()

    def _start(self, msgs):
        self._before_messages()
        ops = []
        for send, recv, peer in msgs:
            if peer is None:
                continue
            ops.append(self.dist.P2POp(self.dist.isend, send, peer))
            ops.append(self.dist.P2POp(self.dist.irecv, recv, peer))
        return self.dist.batch_isend_irecv(ops) if ops else []

    @staticmethod
    def _wait(reqs):
        for r in reqs:
            r.wait()

    def sweep_plan(self):
        """(bottom, interior A, interior B, top) local rung ranges: the boundary rungs are the ones whose llikes the
        neighbours need as halos (my bottom h_send rungs, my top rung)"""
        n = self.b.nloc
        nb = min(self.h_send, n)
        nt = 1 if (self.up is not None and n > nb) else 0
        lo, hi = nb, n - nt
        mid = lo + (hi - lo) // 2
        return (0, nb), (lo, mid - lo), (mid, hi - mid), (n - nt, nt)

    def _ctx(self):
        f = getattr(self.b, "stream_context", None)
        return f() if f else contextlib.nullcontext()

    def step_simple(self, n=1):
        """the four phases one after the other (what the in-process simulator of the tests drives in lockstep)"""
        if self.gathered:
            return self.step_gathered(n)
        with self._ctx():
            for _ in range(n):
                self.stage_halos()
                self._exchange(self.halo_messages())
                self.decide()
                self._recover_step()
                self._exchange(self.row_messages())
                self.finish()

    def step(self, n=1):
        """both message rounds behind arithmetic; the next step's halos are left in flight between calls"""
        if self.gathered:
            return self.step_gathered(n)
        if self.world == 1 or not getattr(self.b, "can_overlap", False):
            return self.step_simple(n)
        with self._ctx():
            self._step_overlapped(n)

    def _step_overlapped(self, n):
        bottom, int_a, int_b, top = self.sweep_plan()
        for _ in range(n):
            if self._halo_reqs is None:                       # first step: nothing in flight yet
                self.stage_halos()
                self._halo_reqs = self._start(self.halo_messages())
            self._wait(self._halo_reqs)
            self.decide()
            self._recover_step()
            reqs = self._start(self.row_messages())
            self.b.sweep_rungs(int_a[0], int_a[1], False)     # ... while the boundary messages travel
            self._wait(reqs)
            self.b.install(self.recv_below, self.recv_above)
            self.b.sweep_rungs(bottom[0], bottom[1], False)
            self.b.sweep_rungs(top[0], top[1], False)
            self.stage_halos()
            self._halo_reqs = self._start(self.halo_messages())
            self.b.sweep_rungs(int_b[0], int_b[1], True)      # ... while the next step's halos travel

    def drain(self):
        """wait for the halos left in flight by step() (call before reading results or tearing down)"""
        if self._halo_reqs is not None:
            with self._ctx():
                self._wait(self._halo_reqs)
            self._halo_reqs = []      # delivered and still valid for the next step


def walker_bounds(n_walkers, world, rank):
    """block of independent ladders (walkers) of `rank`: sizes differ by at most one"""
    base, rem = divmod(n_walkers, world)
    lo = rank * base + min(rank, rem)
    return lo, base + (1 if rank < rem else 0)


class WalkerShardedLadders:
    """The other way to spread a population over GPUs: whole ladders per GPU.  Walkers are independent ladders, so a rank
    that holds walkers [w0, w0 + n) of the population steps them with no message at all -- exchange phases, evolving ladders
    (evolve_temps, the sampler's default, which rung sharding cannot do: every accepted exchange renormalises the whole
    ladder within the step), histories and MAPs included.  The engine keys every random stream by the GLOBAL walker
    (ptm_config.walker_begin), so the chains are those of one engine holding the whole population.
    Rung sharding (ShardedLadder) is for ladders too long for one GPU, or a single ladder (W = 1) as the reference runs it."""

    def __init__(self, make_engine, n_walkers, rank, world):
        """make_engine(walker_begin, n_walkers) -> a configured ptmcmc_amd.engine.Engine holding the whole ladder"""
        self.w0, self.n = walker_bounds(n_walkers, world, rank)
        self.e = make_engine(self.w0, self.n)

    def step(self, n=1):
        self.e.step(n)

    def sync(self):
        self.e.sync()


# ----------------------------------------------------------------------------------------------------------------------
# bench.py --gpus N  (launched by torch.distributed.run, one rank per GPU)
# ----------------------------------------------------------------------------------------------------------------------
def preflight_neighbour_messages(dist, torch, rank, world, device, stream=None, sabotage=False, wait_s=60.0, sizes=(256, 1 << 20, 1 << 20)):
    """The step's message pattern once, with stamped records: the very calls of ShardedLadder._exchange (a batch of
    isend / irecv to both neighbours), on `stream` when given; `sizes`: doubles per message and round.  Returns 0 if every record arrived from the right rank, 1 if
    not (an exception included), 2 if the messages had not completed on the device after `wait_s` seconds -- the stream is
    then still busy with them and must not be used again (the host polls it, it never blocks on it: with RCCL a wait on a
    stuck message would last until the watchdog ends the process).  `sabotage` makes this rank fail on purpose before it
    sends anything (tests): its neighbours then time out instead of waiting for ever."""
    import datetime
    bad = 0
    deadline = time.monotonic() + wait_s
    try:
        if sabotage:
            raise RuntimeError("sabotaged on purpose")
        ctx = torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()
        with ctx:
            up = rank + 1 if rank + 1 < world else None
            down = rank - 1 if rank > 0 else None
            for it, n in enumerate(sizes):   # (the step's messages are megabytes: RCCL picks its protocol by size)
                mine = torch.full((n,), float(1000 * it + rank), dtype=torch.float64, device=device)
                got_up = torch.zeros(n, dtype=torch.float64, device=device)
                got_down = torch.zeros(n, dtype=torch.float64, device=device)
                ops = []
                for send, recv, peer in ((mine, got_up, up), (mine, got_down, down)):
                    if peer is None:
                        continue
                    ops.append(dist.P2POp(dist.isend, send, peer))
                    ops.append(dist.P2POp(dist.irecv, recv, peer))
                if ops:
                    for r in dist.batch_isend_irecv(ops):
                        r.wait(timeout=datetime.timedelta(seconds=wait_s))
                if stream is not None:
                    while not stream.query():
                        if time.monotonic() > deadline:
                            sys.stderr.write("[rank %d] pre-flight: the neighbour messages did not complete within %.0f s\n" % (rank, wait_s))
                            return 2
                        time.sleep(0.0005)
                if up is not None and float(got_up[0]) != 1000 * it + up:
                    bad = 1
                if down is not None and float(got_down[n - 1]) != 1000 * it + down:
                    bad = 1
    except Exception as ex:   # noqa: BLE001
        sys.stderr.write("[rank %d] pre-flight of the neighbour messages failed: %s: %s\n" % (rank, type(ex).__name__, ex))
        bad = 1
    return bad


def bench_main(args):
    import json
    import torch
    import torch.distributed as dist
    from . import engine as E
    from .problems import GaussianProblem
    import bench as B

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    # rehearsal on a box with ONE GPU (tests): every rank on device 0 and the messages over gloo, which takes device tensors --
    # the ranks' engines, torch tensors and message pattern are the real ones, only the transport is not RCCL
    rehearsal = os.environ.get("PTM_BENCH_REHEARSAL", "") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import datetime
    if rehearsal:
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=300))
    else:
        dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=300))   # (a message that never arrives ends the run, not the node's patience)
    D, NT = B.D, B.NT
    W = args.walkers * world            # weak scaling: chains per GPU stay 1024 * walkers
    r0, nloc = shard_bounds(NT, world, rank)
    pr = GaussianProblem(D, NT, B.TMAX)
    # the engine works on an explicit torch stream, and every RCCL message of the step is issued with that stream current
    # (EngineShard.stream_context): kernel -> send / receive -> kernel is then the order of one stream by construction
    stream = torch.cuda.Stream(device=dev)
    by_walkers = getattr(args, "shard", "rungs") == "walkers"
    fallback = ""
    stalled = False # ... because a message never completed (the record is printed without the CPU leg: nothing that takes time happens
                    #     between a message that hangs and the exit)
    stuck = False   # the pre-flight did not pass: RCCL is not touched again, the run ends through os._exit
    ctl = None      # host-side group (gloo) for what the ranks must agree on whatever state RCCL is in
    if not by_walkers and world > 1 and not getattr(args, "native_rccl", False):
        # pre-flight of the step's message pattern (the very calls of ShardedLadder._exchange, on the engine's stream): every
        # rank sends a stamped record to both neighbours and checks what it receives.  If torch.distributed cannot do that here
        # every rank learns it through one all-reduce and the run continues with the population split by WALKERS (whole ladders
        # per GPU, no message in a step) -- said in config.sharding and on stderr -- instead of dying inside the timed region.
        # The verdict is agreed over a HOST-side group: it must get through even when a device message is stuck.
        ctl = dist.group.WORLD if rehearsal else dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=300))
        sab = os.environ.get("PTM_PREFLIGHT_SABOTAGE", "")   # rehearsals: "fail" = every rank's pre-flight fails, "stall" = ... never completes
        bad = 2 if sab == "stall" else preflight_neighbour_messages(dist, torch, rank, world, dev, stream, sabotage=(sab == "fail"))
        flag = torch.tensor([bad], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=ctl)
        if int(flag.item()):
            # Whatever went wrong, RCCL is not touched again: a failed message may have left the communicator (and the stream it
            # was issued on) waiting for ever, so the engine gets a fresh stream, the ranks meet over the host-side group, and
            # the processes leave through os._exit once the record is printed.
            stuck = True
            stalled = int(flag.item()) >= 2
            fallback = "FALLBACK (the pre-flight of the neighbour messages %s on some rank): " % ("never completed" if int(flag.item()) >= 2 else "failed")
            by_walkers = True
            stream = torch.cuda.Stream(device=dev)
            if rank == 0:
                sys.stderr.write("[bench] rung sharding is not available here; the population is split by walkers instead\n")
    try:
        if by_walkers:
            # whole ladders per GPU: rank r holds walkers [r * walkers, (r + 1) * walkers) of the population; no message in a step
            class _Whole:
                def __init__(self, e): self.e = e
                def step(self, n): self.e.step(n)
                def drain(self): pass
            r0, nloc = 0, NT
            eng = E.Engine(D, NT, args.walkers, seed=B.SEED, swap_rate=B.SWAP_RATE, add_every_n=100, device=local, stream=stream.cuda_stream,
                           time_kernels=True, walker_begin=rank * args.walkers)
            pr.configure(eng, E.PROP_LOWER)
            eng.init_from_prior()
            lad = _Whole(eng)
        else:
            eng = E.Engine(D, NT, W, seed=B.SEED, swap_rate=B.SWAP_RATE, add_every_n=100, rung_begin=r0, rung_count=nloc,
                           device=local, stream=stream.cuda_stream, time_kernels=True)
            pr.configure(eng, E.PROP_LOWER)
            eng.init_from_prior()
            if getattr(args, "native_rccl", False):
                class _Native:   # the engine library's own RCCL step (ptm_shard_*)
                    def __init__(self, e): self.e = e
                    def step(self, n): self.e.shard_step(n)
                    def drain(self): pass
                ids = [E.Engine.shard_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(ids, src=0)
                eng.shard_init(ids[0], rank, world, [shard_bounds(NT, world, r)[1] for r in range(world)], args.halo)
                lad = _Native(eng)
            else:
                lad = ShardedLadder(EngineShard(eng, torch, dev, stream), dist, rank, world, halo=args.halo)
        # after a fallback the ranks meet over the host-side group and wait for the ENGINE's stream only (eng.sync): a device-wide
        # wait would include whatever RCCL left behind
        def meet():
            if fallback:
                eng.sync()
                dist.barrier(group=ctl)
                eng.sync()
            else:
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()
        lad.step(300)             # set-up (untimed, uncounted): clocks ramped, chains off their prior draws, RCCL channels open
        lad.drain()
        eng.sync()
        # what each rank's device gives a streaming copy and an f64 fma loop right now (ptm_calibrate: devices differ by several per cent)
        cal = None if getattr(args, "no_calibration", False) else eng.calibrate()
        # evidence of the transport, outside the timed region: what RCCL itself makes of "one from every rank"
        evidence = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}
        if not fallback:
            ones = torch.ones(1, dtype=torch.float64, device=dev)
            dist.all_reduce(ones)
            evidence["all_reduce_of_ones"] = float(ones.item())
            props = torch.cuda.get_device_properties(local)
            mine = {"rank": rank, "device": local, "name": props.name, "uuid": str(getattr(props, "uuid", "")),
                    "rungs": [int(r0), int(r0 + nloc)], "walkers": int(eng.W), "walker_begin": int(eng.walker_begin), "calibration": cal}
            seen = [None] * world
            try:      # (over the host-side group where there is one: a record of who ran must not be what stops the run)
                dist.all_gather_object(seen, mine, group=ctl)
                evidence["ranks"] = seen
            except Exception as ex:   # noqa: BLE001
                evidence["ranks"] = "unavailable (%s: %s)" % (type(ex).__name__, ex)
        # warm again and go straight into the timed steps: a device that idles for ~20 ms drops its clocks (bench.py has the story)
        lad.step(150)
        lad.step(args.warmup)
        lad.drain()
        eng.sync()
        eng.kernel_times(drop=True)
        tries0 = eng.counter_sums()[0]   # (two sums reduced on the device: no idle gap for the clocks to drop in -- bench.py has the story)
        meet()
        t0 = time.perf_counter()
        lad.step(args.steps)
        lad.drain()
        eng.sync()
        meet()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=None if fallback else dev)
    except Exception as ex:
        # a rank that fails (e.g. PTM_ERR_FAR_MOVE out of eng.sync()) must not leave its peers waiting in a receive or a
        # barrier: leave at once with a non-zero code, the launcher then ends the other ranks
        sys.stderr.write("[bench rank %d] %s: %s\n" % (rank, type(ex).__name__, ex))
        sys.stderr.flush()
        os._exit(17)
    grp = ctl if fallback else None
    dist.all_reduce(dt, op=dist.ReduceOp.MAX, group=grp)
    wall = float(dt.item())
    kt = eng.kernel_times()              # one entry per sweep launch; a step's sweep is up to four launches
    moved = (eng.counter_sums()[0] - tries0) / float(args.steps)   # chains this rank's sweeps worked on, per step
    kavg = torch.tensor([float(kt.sum()) / args.steps], dtype=torch.float64, device=None if fallback else dev)
    dist.all_reduce(kavg, op=dist.ReduceOp.MAX, group=grp)
    nchains = NT * W
    if rank == 0:
        kavg_ms = float(kavg.item())
        value = nchains * args.steps / wall
        roof = B.roofline_record(eng.sweep_kernel_name, kavg_ms, int(kt.size), moved, eng.Nc, value, world, None, calibration=cal)
        roof["per_gpu"] = True
        roof["kernel_avg_ms_is"] = "sum of the rank's sweep launches of a step (up to four partial sweeps), max over ranks; chains_processed of rank 0"
        how = "walkers" if by_walkers else "rungs"
        out = {
            "metric": "ladder-wide MH steps/sec (D=32 Gaussian, 1024 temps)",
            "value": value, "unit": "MH steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "D=32 correlated Gaussian, 1024-rung ladder (Tmax=1e9, swap_rate=0.1) x %d walkers; "
                                   "per-rung Cholesky proposal factors; uniform box prior" % W,
                       "layout": B.LAYOUT_NOTE,
                       "dim": D, "rungs": NT, "walkers": W, "chains": nchains,
                       # what was measured, unmistakably: "rung-sharded" is BASELINE's configuration (neighbour exchanges over
                       # RCCL); "fallback" means the pre-flight of those messages did not pass and this line is NOT that measurement
                       "measured": "fallback" if fallback else ("rung-sharded" if not by_walkers else "walker-split (asked for: --shard walkers)"),
                       "data_path_messages": "none" if by_walkers else "llike halos + boundary rows, point-to-point between neighbour ranks",
                       "transport": evidence,
                       "sharding": (fallback + "%d blocks of %d whole ladders (walkers), no message in a step" % (world, args.walkers)) if by_walkers else
                                   ("%d contiguous rung blocks of %d rungs; llike halo %d rungs; neighbour p2p over RCCL (%s)"
                                    % (world, nloc, args.halo, "ptm_shard_*: native ncclSend/ncclRecv" if getattr(args, "native_rccl", False) else "torch.distributed"))},
            "roofline": roof,
            "calibration": cal,
        }
        if not getattr(args, "no_cpu", False) and not stalled:
            # the CPU baseline of the same workload on this box's host cores, rank 0 only, after the timed region (the other ranks
            # wait at the final barrier; bounded: ~10-30 s)
            try:
                out["cpu_baseline"] = B.cpu_baseline(pr)
            except Exception as ex:   # noqa: BLE001
                sys.stderr.write("[bench] cpu_baseline failed: %s\n" % ex)
        print(json.dumps(out), flush=True)
    if stuck:   # RCCL may still hold a message that will never complete: freeing device memory or closing the group would wait for it
        dist.barrier(group=ctl)
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)
    eng.close()
    dist.destroy_process_group()
