"""Ladder sharding at the benchmark's shape, on the ONE GPU a test box has (-m gpu):
  - the exchange kernel alone on a middle 128-rung shard of the 1024-rung ladder at the 8-GPU bench's 131072 ladders, with
    the default halo (the first round's halo of 4 trips PTM_ERR_FAR_MOVE at this scale, tests/test_halo_depth.py);
  - BASELINE configs[3] in process: 1024 rungs in 8 shards of 128, every shard an EngineShard on its own torch stream with
    torch tensors as message buffers (the objects bench.py --gpus 8 uses), plain and overlapped order, against one engine;
  - bench.py's torch.distributed / RCCL path with one rank."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from ptmcmc_amd import engine as E
from ptmcmc_amd.parallel import DEFAULT_HALO
from ptmcmc_amd.problems import GaussianProblem

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _decide_only(H, W, nsteps):
    """ptm_exchange_decide alone, `nsteps` steps, on the shard [512, 640) of the 1024-rung ladder; returns None or the error"""
    D, Nt, G = 32, 1024, 8
    nloc = Nt // G
    r0 = nloc * (G // 2)
    pr = GaussianProblem(D, Nt, 1e9)
    eng = E.Engine(D, Nt, W, rung_begin=r0, rung_count=nloc, swap_rate=0.1)
    pr.configure(eng, E.PROP_LOWER)
    eng.init_from_prior()
    n = eng.exchange_buffer_doubles
    ll = eng.llike
    lb, la = E.DeviceBuffer(8 * W), E.DeviceBuffer(8 * H * W)
    su, sd = E.DeviceBuffer(8 * n), E.DeviceBuffer(8 * n)
    fill = np.ascontiguousarray(np.resize(ll, H * W))        # plausible neighbour llikes
    lb.copy_from(fill.ctypes.data, lb.nbytes)
    la.copy_from(fill.ctypes.data, la.nbytes)
    err = None
    try:
        for k in range(nsteps):
            eng.exchange_decide(lb.ptr, la.ptr, H, su.ptr, sd.ptr)
            eng.sweep_rungs(0, 0, True)                       # closes the step: the next one draws new candidates
            if k % 50 == 49:
                eng.sync()
        eng.sync()
    except E.PtmError as ex:
        err = str(ex)
    assert eng.step_count == nsteps or err
    eng.close()
    return err


def test_exchange_decide_at_bench_scale_is_clean_with_the_default_halo():
    W, nsteps = 131072, 400          # bench.py --gpus 8: 16384 x 8 ladders, 300 set-up + warm-up + timed steps
    assert _decide_only(DEFAULT_HALO, W, nsteps) is None
    err = _decide_only(4, W, nsteps)  # ~2-6e-8 per ladder-step => a few events expected in 5.2e7; not required to fire
    if err is not None:
        assert "halo" in err


@pytest.mark.parametrize("overlap", [0, 1])
def test_1024_rungs_in_8_engine_shards_on_torch_streams(overlap):
    """BASELINE configs[3] in process (D=32, 1024 rungs, 8 shards of 128, 64 ladders), through ptmcmc_amd.parallel.EngineShard
    with torch tensors and one explicit torch stream per shard -- what bench.py --gpus 8 runs, minus the wire.  In a child
    process that imports torch BEFORE the engine library loads (tests/torch_shard_worker.py says why)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "torch_shard_worker.py"), str(overlap)], capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "OK" in out.stdout


@pytest.mark.parametrize("world,D,Nt,W,halo,sr,evolve", [(2, 32, 16, 64, 4, 0.3, 0.0), (3, 32, 48, 128, 8, 0.1, 0.0), (4, 8, 24, 5, 4, 0.45, 0.0),
                                                         (3, 32, 25, 64, 4, 0.3, 0.01), (2, 6, 11, 3, 4, 0.45, 0.05),
                                                         (3, 8, 12, 64, 1, 0.5, 0.0)])
def test_sharded_ladder_between_processes_on_the_gpu(world, D, Nt, W, halo, sr, evolve):
    """The N > 1 path with real engines in separate PROCESSES: every rank an EngineShard on its own torch stream, the sharded
    (overlapped) step of ShardedLadder, its messages between the processes -- over gloo on the one GPU this box has (RCCL refuses
    two ranks on one device; tests/gpu_dist_worker.py says what differs).  The blocks put together are, bit for bit, the ladder
    of one engine: states, llikes, counters, swap bookkeeping -- for fixed ladders (llike halos between neighbours) and evolving
    ones (an all-gather of the llikes per step, ShardedLadder.step_gathered).  The last case has a halo of ONE rung: runs of
    surviving picks longer than that come every few steps and are recovered (every rank leaves the same ladders alone, gathers
    the ladder's llikes and decides them from the full view -- ShardedLadder._recover_step)."""
    import socket
    import tempfile
    from ptmcmc_amd.parallel import shard_bounds
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    nsteps = 25
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "rank%d.npz")
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "gpu_dist_worker.py"), str(D), str(Nt), str(W), str(nsteps),
                                           str(halo), str(sr), out, str(evolve)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        res = [p.communicate(timeout=600) for p in procs]
        assert all(p.returncode == 0 for p in procs), [r[1][-1500:] for r in res]
        parts = [np.load(out % r) for r in range(world)]
    if halo == 1:
        rec = [int(p["recovered"]) for p in parts]
        assert rec[0] > 0 and len(set(rec)) == 1, rec
    pr = GaussianProblem(D, Nt, 1e6)
    ref = E.Engine(D, Nt, W, swap_rate=sr)
    pr.configure(ref, E.PROP_LOWER)
    if evolve > 0:     # (evolving ladders: the gathered form of the sharded step, every rank keeps the whole ladders' temperatures)
        ref.set_evolve_temps(evolve)
    ref.init_from_prior()
    ref.step(nsteps); ref.sync()
    if evolve > 0:
        for p in parts:
            assert np.array_equal(p["invtemps"], ref.invtemps())
        assert not np.array_equal(ref.invtemps()[0], pr.beta)
    assert np.array_equal(np.concatenate([p["x"] for p in parts]), ref.states())
    assert np.array_equal(np.concatenate([p["ll"] for p in parts]), ref.llike)
    assert np.array_equal(np.concatenate([p["nacc"] for p in parts]), ref.naccept)
    assert np.array_equal(np.concatenate([p["nhist"] for p in parts]), ref.nhist)
    t, a = ref.swap_counts()
    assert np.array_equal(sum(p["st"] for p in parts), t) and np.array_equal(sum(p["sa"] for p in parts), a)
    for g in range(world - 1):              # rows did cross every boundary
        b = shard_bounds(Nt, world, g + 1)[0]
        assert a[:, b - 1].sum() > 0, b
    ref.close()


@pytest.mark.parametrize("world,D,Nt,W,sr,evolve,every,nsteps", [(2, 6, 20, 1, 0.1, 0.01, 40, 400),     # the reference's MPI regression: 20 rungs over 2 ranks, evolving, one ladder
                                                                 (2, 6, 20, 3, 0.4, 0.02, 3, 60), (3, 32, 25, 64, 0.3, 0.01, 2, 30)])
def test_evolving_sharded_ladder_records_history_and_map_like_one_engine(world, D, Nt, W, sr, evolve, every, nsteps):
    """History and MAP tracking of an EVOLVING ladder on rung shards -- the reference's own multi-rank regression is that shape:
    20 rungs over 2 ranks with --pt_evolve_rate=0.01, the cold chain's file compared with the 1-rank file
    (test/exampleLISA/Makefile:7,29-31; the rows MH_chain::add_state saves, chain.cc:935-946, gathered :1905-1972).  Every shard
    replays every pick of the ladder and so knows the temperature each of its rungs had at each add_state of the exchange phase:
    the rows each rank recorded -- states, llikes, lpriors, counters, the temperatures they were saved at, in-between rows of rungs
    exchanged twice included -- and its rungs' MAPs are, bit for bit, those of one engine holding the whole ladder."""
    import socket
    import tempfile
    from ptmcmc_amd.parallel import shard_bounds
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "rank%d.npz")
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "gpu_dist_worker.py"), str(D), str(Nt), str(W), str(nsteps),
                                           "4", str(sr), out, str(evolve), str(every)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        res = [p.communicate(timeout=900) for p in procs]
        assert all(p.returncode == 0 for p in procs), [r[1][-1500:] for r in res]
        parts = [dict(np.load(out % r)) for r in range(world)]
    pr = GaussianProblem(D, Nt, 1e6)
    cap = 2 * nsteps // every + 8
    ref = E.Engine(D, Nt, W, swap_rate=sr, add_every_n=every, history_rungs=Nt, history_capacity=cap, map_rungs=Nt)
    pr.configure(ref, E.PROP_LOWER)
    ref.set_evolve_temps(evolve)
    ref.init_from_prior()
    ref.step(nsteps); ref.sync()
    assert np.array_equal(np.concatenate([p["x"] for p in parts]), ref.states())
    assert np.array_equal(np.concatenate([p["nhist"] for p in parts]), ref.nhist)
    assert not np.array_equal(ref.invtemps()[0], pr.beta)
    hr_, mr = ref.history(), ref.map()
    nsize = ref.nsize.reshape(Nt, W)
    rows_checked = double_adds = 0
    for g, p in enumerate(parts):
        r0, n = shard_bounds(Nt, world, g)
        hr = int(p["hist_rungs"])
        assert hr == (n if g == world - 1 else max(1, n // 2))
        sel = slice(r0 * W, (r0 + hr) * W)
        for name in ("x", "llike", "lprior", "naccept", "ntries", "last_type", "invtemp", "row"):
            a, b = p["h_" + name], hr_[name][:, sel]
            for s_ in range(int(nsize[r0:r0 + hr].max())):
                have = (nsize[r0:r0 + hr] > s_).ravel()
                assert np.array_equal(a[s_ % cap][have], b[s_ % cap][have]), (g, name, s_)
                rows_checked += int(have.sum()) if name == "x" else 0
        for name in ("x", "lpost", "llike", "lprior"):
            assert np.array_equal(p["m_" + name], mr[name][sel]), (g, name)
        double_adds += int((ref.nhist.reshape(Nt, W)[r0:r0 + hr] > nsteps).sum())
    assert rows_checked > world * W and (every > 3 or double_adds > 0)
    # the temperatures saved with the rows are not the common ladder's: the rows of the exchange phases were saved between two pries
    assert (np.abs(hr_["invtemp"][1:int(nsize.min())] - np.repeat(pr.beta, W)) > 0).any()
    ref.close()


def test_bench_distributed_path_with_one_rank():
    """bench.py --force-dist: init_process_group("nccl"), the engine on an explicit torch stream, EngineShard on torch
    tensors, the all_reduces of the record -- the N > 1 code path with world size 1"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "3", "--warmup", "1",
                          "--walkers", "1024"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["n_gpus"] == 1 and rec["steps"] == 3 and rec["value"] > 1e8
    assert rec["config"]["chains"] == 1024 * 1024 and rec["roofline"]["kernel"].startswith("sweep_mfma32_kernel")


@pytest.mark.parametrize("sabotage,expect", [("", "contiguous rung blocks"), ("fail", "FALLBACK"), ("stall", "never completed")])
def test_bench_with_two_ranks_rehearsed_on_one_gpu(sabotage, expect):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), rehearsed on the one GPU of
    this box: both ranks on device 0, messages over gloo (PTM_BENCH_REHEARSAL=1 -- RCCL refuses two ranks on one device).  The
    rung-sharded step; a pre-flight of the neighbour messages that fails on every rank; one whose messages never complete: the
    last two must end in the walker split, with ONE record and exit code 0."""
    import socket
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    env = dict(os.environ, PTM_BENCH_REHEARSAL="1", PTM_PREFLIGHT_SABOTAGE=sabotage, OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--walkers", "1024", "--steps", "6", "--warmup", "2"],
                         env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    recs = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(recs) == 1
    rec = recs[0]
    assert rec["n_gpus"] == 2 and rec["steps"] == 6 and rec["scaling"] == "weak" and rec["value"] > 1e7
    assert rec["config"]["walkers"] == 2048 and rec["config"]["chains"] == 2 * 1024 * 1024
    assert expect in rec["config"]["sharding"], rec["config"]["sharding"]
    # what was measured is said unmistakably, with the transport's own evidence, the CPU baseline and an honest roofline
    assert rec["config"]["measured"] == ("rung-sharded" if sabotage == "" else "fallback")
    ev = rec["config"]["transport"]
    assert ev["world_size"] == 2
    if sabotage == "":
        assert ev["all_reduce_of_ones"] == 2.0 and [r["rank"] for r in ev["ranks"]] == [0, 1]
        assert [r["rungs"] for r in ev["ranks"]] == [[0, 512], [512, 1024]]
    if sabotage != "stall":
        assert "cpu_baseline" in rec, out.stderr[-3000:]
        assert rec["cpu_baseline"]["value"] > 0 and rec["cpu_baseline"]["kind"] in ("reference", "port")
    rf = rec["roofline"]
    assert 0.5 < rf["moving_fraction"] < 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert abs(rf["step_frac"] - rec["value"] * rf["bytes_per_mh_step"] / (2 * rf["peak"] * 1e9)) < 1e-9


@pytest.mark.parametrize("D,Nt,W,ev,hist", [(32, 16, 256, 0.02, False), (6, 12, 10, 0.03, True), (32, 8, 2048, 0.0, False)])
def test_population_split_by_walkers_gives_the_single_engine_chains(D, Nt, W, ev, hist):
    """ptm_config.walker_begin: engines that hold whole ladders for blocks of the walkers (no message between them) reproduce
    the one-engine population bit for bit -- evolving ladders (what rung sharding refuses), histories and MAPs included; the
    third case goes through the compacted sweep."""
    from ptmcmc_amd.parallel import walker_bounds
    G, sr = 3, 0.3
    pr = GaussianProblem(D, Nt, 1e3)
    kw = dict(history_rungs=Nt, history_capacity=40, map_rungs=Nt) if hist else {}
    ref = E.Engine(D, Nt, W, swap_rate=sr, **kw)
    pr.configure(ref, E.PROP_LOWER)
    if ev: ref.set_evolve_temps(ev)
    ref.init_from_prior()
    parts = []
    for g in range(G):
        w0, n = walker_bounds(W, G, g)
        e = E.Engine(D, Nt, n, swap_rate=sr, walker_begin=w0, **kw)
        pr.configure(e, E.PROP_LOWER)
        if ev: e.set_evolve_temps(ev)
        e.init_from_prior()               # the prior draws are keyed by the global walker too
        parts.append((w0, n, e))
    def gather(name):
        full = np.array(getattr(ref, name)) if name != "x" else ref.states()
        got = np.empty_like(full)
        g3 = got.reshape((Nt, W) + full.shape[1:])
        for w0, n, e in parts:
            a = e.states() if name == "x" else getattr(e, name)
            g3[:, w0:w0 + n] = a.reshape((Nt, n) + full.shape[1:])
        return got, full
    got, full = gather("x")
    assert np.array_equal(got, full), "prior draws"
    for k in range(4):
        ref.step(5)
        for _, _, e in parts: e.step(5)
        got, full = gather("x")
        assert np.array_equal(got, full), "states after %d steps" % (5 * k + 5)
    for name in ("llike", "lprior", "ntries", "naccept", "nhist", "last_type"):
        got, full = gather(name)
        assert np.array_equal(got, full), name
    if ev:
        b = ref.invtemps()
        for w0, n, e in parts:
            assert np.array_equal(e.invtemps(), b[w0:w0 + n])
        assert (np.abs(b[:, 1:-1] - pr.beta[1:-1]) > 0).any()
    if hist:
        hr, mr = ref.history(), ref.map()
        for w0, n, e in parts:
            he, me = e.history(), e.map()
            used = hr["row"].reshape(40, Nt, W)[:, :, w0:w0 + n] >= 0          # (ring slots never written hold no row)
            assert np.array_equal(he["row"].reshape(40, Nt, n) >= 0, used) and used.sum() > 20 * Nt * n
            for key in ("x", "llike", "naccept", "last_type", "invtemp", "row"):
                a = hr[key].reshape((40, Nt, W) + hr[key].shape[2:])[:, :, w0:w0 + n]
                assert np.array_equal(he[key].reshape(a.shape)[used], a[used]), key
            assert np.array_equal(me["lpost"].reshape(Nt, n), mr["lpost"].reshape(Nt, W)[:, w0:w0 + n])
    for _, _, e in parts: e.close()
    ref.close()


def test_bench_distributed_path_by_walkers_with_one_rank():
    """bench.py --shard walkers: whole ladders per rank, streams keyed by the global walker -- world size 1 over nccl"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29542", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--shard", "walkers", "--steps", "3",
                          "--warmup", "1", "--walkers", "1024"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["n_gpus"] == 1 and rec["value"] > 1e8 and "whole ladders" in rec["config"]["sharding"]


def test_native_rccl_sharded_step_with_one_rank():
    """ptm_shard_*: the sharded step driven by the engine library itself over RCCL (dlopen'ed at the first call).  With one
    rank there is nobody to talk to, but everything else runs: communicator, side stream and events, the overlapped launch
    order over interior / boundary rung ranges (through the compacted sweep at this size) -- and must give ptm_step's chains."""
    D, Nt, W, sr = 32, 24, 1024, 0.3
    pr = GaussianProblem(D, Nt, 1e3)
    ref = E.Engine(D, Nt, W, swap_rate=sr)
    pr.configure(ref, E.PROP_LOWER)
    ref.init_from_prior()
    eng = E.Engine(D, Nt, W, swap_rate=sr)
    pr.configure(eng, E.PROP_LOWER)
    eng.set_states(ref.states())
    uid = E.Engine.shard_unique_id()
    assert len(uid) == 128 and any(uid)
    with pytest.raises(E.PtmError, match="rung_counts"):
        eng.shard_init(uid, 0, 1, [Nt - 1])
    eng.shard_init(uid, 0, 1, [Nt])
    for k in range(3):
        ref.step(4); eng.shard_step(4)
        eng.sync()
        assert np.array_equal(eng.states(), ref.states()), k
    for name in ("llike", "ntries", "naccept", "nhist", "last_type"):
        assert np.array_equal(getattr(eng, name), getattr(ref, name)), name
    eng.shard_finalize()
    eng.step(2); ref.step(2)                       # back to the plain step
    assert np.array_equal(eng.states(), ref.states())
    eng.close(); ref.close()


def test_bench_native_rccl_path_with_one_rank():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--native-rccl", "--steps", "3",
                          "--warmup", "1", "--walkers", "1024"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["n_gpus"] == 1 and rec["value"] > 1e8 and "native ncclSend" in rec["config"]["sharding"]
