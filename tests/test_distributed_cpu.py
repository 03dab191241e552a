"""N>1 path on CPU: the sharding protocol of ptmcmc_amd.parallel.ShardedLadder over torch.distributed/gloo
(world_size 2 and 3), with the oracle as each rank's local compute, against the single-process oracle."""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import parity_util as PU

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,D,Nt,W,halo,sr", [(2, 4, 10, 3, 4, 0.4), (3, 3, 12, 2, 3, 0.45)])
def test_sharded_ladder_over_gloo_matches_single_process(world, D, Nt, W, halo, sr):
    sys.path.insert(0, HERE)
    import dist_worker
    nsteps = 25
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "rank%d.npz")
        port = free_port()
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       OMP_NUM_THREADS="1")
            procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(D), str(Nt), str(W),
                                           str(nsteps), str(halo), str(sr), out], env=env))
        for p in procs:
            assert p.wait(timeout=300) == 0
        parts = [np.load(out % r) for r in range(world)]
    ref = dist_worker.make_ladder(D, Nt, W, sr, 0x5EED0001)
    ref.pt_step(nsteps)
    x = np.concatenate([p["x"] for p in parts])
    assert np.array_equal(x, PU.to_engine_order(ref.x, Nt, W))
    assert np.array_equal(np.concatenate([p["ll"] for p in parts]), PU.to_engine_order(ref.llike, Nt, W))
    assert np.array_equal(np.concatenate([p["nhist"] for p in parts]), PU.to_engine_order(ref.nhist, Nt, W))
    assert np.array_equal(np.concatenate([p["nacc"] for p in parts]), PU.to_engine_order(ref.naccept, Nt, W))
    assert np.array_equal(sum(p["st"] for p in parts), ref.swap_count)
    assert np.array_equal(sum(p["sa"] for p in parts), ref.swap_accept_count)
    assert ref.swap_accept_count.sum() > 0


def _run_ranks(world, args, timeout=300):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "rank%d.npz")
        port = free_port()
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       OMP_NUM_THREADS="1")
            a = [str(v) for v in args]
            procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py")] + a[:6] + [out] + a[6:], env=env,
                                          stderr=subprocess.DEVNULL if "quiet" in os.environ.get("PTM_TEST_WORKER", "") else None))
        codes = [p.wait(timeout=timeout) for p in procs]
        parts = [np.load(out % r) for r in range(world)] if not any(codes) else None
    return codes, parts


@pytest.mark.parametrize("world,D,Nt,W,sr", [(2, 4, 12, 3, 0.5), (3, 3, 12, 4, 0.5)])
def test_a_run_longer_than_the_halo_is_recovered_over_gloo(world, D, Nt, W, sr):
    """Halo of ONE rung on a 12-rung ladder, many exchange attempts per step: within the first steps some shard meets a run of
    surviving picks it cannot decide from its halo.  (1) Without recovery the ranks stop loudly -- the run provably comes.  (2) With
    it (ShardedLadder(recover=True), what a halo below the default depth turns on) every rank finds the same ladders from the
    replayed draws, leaves them alone, all-gathers the ladder's llikes and decides them from the full view: the sharded run equals
    the single-process oracle bit for bit, and the second pass was taken."""
    sys.path.insert(0, HERE)
    import dist_worker
    nsteps = 60
    os.environ["PTM_TEST_WORKER"] = "quiet"
    try:
        codes, _ = _run_ranks(world, [D, Nt, W, nsteps, 1, sr, 0.0, -1.0, 0])
    finally:
        del os.environ["PTM_TEST_WORKER"]
    assert any(codes), "the configuration does not meet a run longer than the halo: the test below would prove nothing"
    codes, parts = _run_ranks(world, [D, Nt, W, nsteps, 1, sr, 0.0, -1.0, 1])
    assert not any(codes)
    rec = [int(p["recovered"]) for p in parts]
    assert rec[0] > 0 and len(set(rec)) == 1
    ref = dist_worker.make_ladder(D, Nt, W, sr, 0x5EED0001)
    ref.pt_step(nsteps)
    assert np.array_equal(np.concatenate([p["x"] for p in parts]), PU.to_engine_order(ref.x, Nt, W))
    assert np.array_equal(np.concatenate([p["ll"] for p in parts]), PU.to_engine_order(ref.llike, Nt, W))
    assert np.array_equal(np.concatenate([p["nhist"] for p in parts]), PU.to_engine_order(ref.nhist, Nt, W))
    assert np.array_equal(np.concatenate([p["nacc"] for p in parts]), PU.to_engine_order(ref.naccept, Nt, W))
    assert np.array_equal(sum(p["st"] for p in parts), ref.swap_count)
    assert np.array_equal(sum(p["sa"] for p in parts), ref.swap_accept_count)


@pytest.mark.parametrize("world,D,Nt,W,sr,rate,cut", [(2, 4, 10, 3, 0.4, 0.05, -1.0), (3, 3, 13, 2, 0.45, 0.02, 0.0)])
def test_evolving_sharded_ladder_over_gloo_matches_single_process(world, D, Nt, W, sr, rate, cut):
    """evolve_temps on a rung-sharded ladder over torch.distributed (gloo): ShardedLadder.step_gathered -- an all-gather of every
    shard's llikes / lpriors per step (the reference's gather_llikes / gather_lposts, chain.cc:1433-1435,1950-1972), every rank
    replaying the whole ladder's trials, boundary rows between neighbours -- against the single-process oracle: states, llikes,
    counters, swap counts and the evolved temperatures (every rank holds them all)."""
    sys.path.insert(0, HERE)
    import dist_worker
    nsteps = 25
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "rank%d.npz")
        port = free_port()
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       OMP_NUM_THREADS="1")
            procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(D), str(Nt), str(W),
                                           str(nsteps), "4", str(sr), out, str(rate), str(cut)], env=env))
        for p in procs:
            assert p.wait(timeout=300) == 0
        parts = [np.load(out % r) for r in range(world)]
    ref = dist_worker.make_ladder(D, Nt, W, sr, 0x5EED0001, rate, cut)
    ref.pt_step(nsteps)
    assert np.array_equal(np.concatenate([p["x"] for p in parts]), PU.to_engine_order(ref.x, Nt, W))
    assert np.array_equal(np.concatenate([p["ll"] for p in parts]), PU.to_engine_order(ref.llike, Nt, W))
    assert np.array_equal(np.concatenate([p["nhist"] for p in parts]), PU.to_engine_order(ref.nhist, Nt, W))
    assert np.array_equal(np.concatenate([p["nacc"] for p in parts]), PU.to_engine_order(ref.naccept, Nt, W))
    assert np.array_equal(sum(p["st"] for p in parts), ref.swap_count)
    assert np.array_equal(sum(p["sa"] for p in parts), ref.swap_accept_count)
    for p in parts:
        assert np.array_equal(p["betaw"], ref.betaw)
    assert ref.swap_accept_count.sum() > 0 and not np.array_equal(ref.betaw[0], ref.beta)


def test_shard_bounds_cover_the_ladder():
    from ptmcmc_amd.parallel import shard_bounds
    for nt in (1024, 10, 7):
        for g in (1, 2, 3, 4, 8):
            if g > nt:
                continue
            blocks = [shard_bounds(nt, g, r) for r in range(g)]
            assert blocks[0][0] == 0 and sum(n for _, n in blocks) == nt
            for (a, n), (b, _) in zip(blocks, blocks[1:]):
                assert a + n == b


@pytest.mark.parametrize("world,sabotage", [(3, -1), (3, -2), (2, -1)])
def test_bench_preflight_of_the_neighbour_messages(world, sabotage):
    """bench.py --gpus N checks the step's message pattern before its timed region (stamped records to both neighbours through
    the calls ShardedLadder._exchange makes) and all ranks agree on the verdict through one all-reduce; if the calls fail
    (every rank alike: the failure such a check can survive -- a rank that dies alone takes the process group with it, and the
    run ends on the group's timeout) the bench goes on with the walker split.  Here over gloo: a clean ring agrees on 0,
    ranks that all fail before sending (sabotage -2) agree on 1."""
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "preflight_worker.py"), str(sabotage)], env=env, stdout=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    want = 1 if sabotage == -2 else 0
    for r, o in enumerate(outs):
        assert "rank %d bad %d agreed %d" % (r, want, want) in o, outs
