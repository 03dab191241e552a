#!/usr/bin/env python3
"""bench.py -- ladder-wide MH steps/s of the fused parallel-tempering step on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json): D=32 correlated Gaussian, 1024-rung geometric ladder (Tmax=1e9), swap_rate 0.1,
per-rung Cholesky proposal factors, uniform box prior; `walkers` independent ladders are batched per GPU so the
chip is full (1024 chains alone are 16 wavefronts on a 256-CU part; that latency-bound case is reported beside the
headline as "w1").  A step = one parallel_tempering_chains::step for every ladder: exchange phase + MH sweep.
With N GPUs the ladder is sharded in contiguous rung blocks (1024/N rungs per GPU) and the walker count grows
with N so that per-GPU work is fixed (weak scaling); accepted exchanges across a shard boundary travel as
point-to-point messages between neighbouring ranks.

Prints ONE JSON line (rank 0).  roofline.achieved = algorithmic bytes per launch (16*D+44 B per MH step, SURVEY.md
section 8(d), x the chains the sweep kernel works on in that launch -- the compacted sweep skips the rungs an exchange
attempt touched; counted from MH_chain::Ntries) / mean duration of that kernel from HIP events recorded around each of
its launches on the engine's stream inside the timed region (roofline_record() has the definitions of frac, step_frac,
traffic_frac).  cpu_baseline = the real reference's own parallel_tempering_chains::step
(oracle/_ref/ptm_ref_driver, built from /root/reference) on this host, 1 thread, bounded sample.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D, NT, TMAX, SWAP_RATE, SEED = 32, 1024, 1e9, 0.1, 0x5EED0001
LAYOUT_NOTE = ("states are contiguous rows per chain (AoS, in MFMA accumulator order), updated in place; the rung's proposal factor "
               "and the precision matrix are read as MFMA operand tiles from L2 / LDS (north_star names SoA planes and an LDS-staged "
               "factor: both were measured and lost, DESIGN.md section 2 / 3.1)")
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
NOMINAL_F64_FMA_TFS = 70.0   # what ptm_calibrate's f64 fma loop reaches on the usual device of this pool (profiles/r04_calibration.json)


def algorithmic_bytes(dim):
    return 16 * dim + 44   # r+w state, beta, r+w lpost & llike, accept/type flag


def roofline_record(kernel, kernel_avg_ms, launches, moved_per_launch, chains_per_gpu, value, n_gpus, traffic, kt=None, calibration=None):
    """The roofline object of the JSON line, for the dominant kernel (the fused MH sweep).

    achieved   = B(D) x (chains the kernel actually worked on per launch) / (its mean launch duration, HIP events on the engine's
                 stream around that kernel alone -- the same kernel, by name, whose average rocprofv3 --kernel-trace --stats
                 reports in profiles/).  The compacted sweep visits only the chains that make a Metropolis move this step (a rung
                 touched by an exchange attempt makes none, chain.cc:1553-1557): they are counted from MH_chain::Ntries, not
                 assumed.  frac = achieved / peak.
    step_frac  = value x B(D) / (n_gpus x peak): SURVEY 8(d)'s own step-level formula (every chain of the ladder, whole step).
    traffic    = HBM bytes per launch of that kernel from the committed PMC passes (profiles/*_pmc_summary.json), or null;
                 traffic_frac = traffic / kernel time / peak: the real HBM rate."""
    B = algorithmic_bytes(D)
    achieved = B * moved_per_launch / (kernel_avg_ms * 1e-3) / 1e9
    spread = {}
    if kt is not None and len(kt):
        # the launches one by one: a box that throttles shows as a max far from the median, a slow box as all three moved together
        spread = {"kernel_min_ms": float(np.min(kt)), "kernel_median_ms": float(np.median(kt)), "kernel_max_ms": float(np.max(kt)),
                  "kernel_ms_series": [round(float(v), 4) for v in kt[:200]]}
    if calibration:
        # the same bytes against what a plain copy kernel reached on THIS device minutes before (calibration.copy_GBs), and the
        # kernel's time scaled to a nominal device: it is bound by f64 issue, so its time goes with the f64 fma
        # rate the device holds (calibration.f64_fma_TFs; NOMINAL_F64_FMA_TFS is the pool's usual figure)
        spread["frac_of_measured_copy"] = achieved / calibration["copy_GBs"]
        spread["kernel_avg_ms_at_nominal_f64_rate"] = kernel_avg_ms * calibration["f64_fma_TFs"] / NOMINAL_F64_FMA_TFS
    return {**spread, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": (traffic["bytes"] if traffic else None),
            "traffic_frac": (traffic["bytes"] / (kernel_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic else None),
            "traffic_detail": traffic,
            "definition": "frac = bytes_per_mh_step x chains_processed_per_launch / kernel_avg_ms / peak (the chains the kernel "
                          "visits, from Ntries); step_frac = value x bytes_per_mh_step / (n_gpus x peak)",
            "step_frac": value * B / (n_gpus * HBM_PEAK_GBS * 1e9),
            "kernel": kernel, "kernel_avg_ms": kernel_avg_ms, "launches": launches,
            "timed_bracket": [kernel], "outside_the_bracket": ["partition_kernel (the compacted sweep's list build)",
                                                               "decide_kernel (exchange phase)", "fold_swap_log_kernel"],
            "chains_processed_per_launch": moved_per_launch, "chains_per_gpu": chains_per_gpu,
            "moving_fraction": moved_per_launch / float(chains_per_gpu),
            "algorithmic_bytes": B * moved_per_launch, "bytes_per_mh_step": B}


def measured_traffic(kernel_name):
    """HBM bytes per launch of the sweep kernel from the newest committed rocprofv3 --pmc summary (profiles/*_pmc_summary.json,
    written by tools/summarize_profile.py from separate FETCH_SIZE / WRITE_SIZE passes with the gfx950 corrections of
    MI355X_MICROARCH.md).  Counters cannot be read from inside this process; None if no matching profile is committed."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))):
        try:
            ks = json.load(open(f))["kernels"]
        except Exception:
            continue
        for k, v in ks.items():
            # (only summaries of the HBM traffic passes carry these keys: other counter summaries under profiles/ are not for here)
            if kernel_name.replace(" ", "") in k.replace(" ", "") and isinstance(v, dict) and all(q in v for q in ("hbm_bytes", "hbm_read_bytes", "hbm_write_bytes")):
                best = {"bytes": v["hbm_bytes"], "read": v["hbm_read_bytes"], "write": v["hbm_write_bytes"], "source": os.path.basename(f)}
    return best


def cpu_baseline(problem, budget_s=15.0):
    """Reference CPU path on this box's host cores.  Bounded: ~10-30 s of CPU work."""
    drv = os.path.join(ROOT, "oracle", "_ref", "ptm_ref_driver")
    nsteps = 300   # 1024 rungs x 300 steps ~ 3e5 reference MH steps ~ 10 s at the ~3e4 steps/s measured in BASELINE.md
    if os.path.exists(drv):
        with tempfile.NamedTemporaryFile("w", suffix=".spec", delete=False) as f:
            f.write("%d %d %d %.17g %.17g %.17g %.17g\n" % (D, NT, nsteps, TMAX, SWAP_RATE, 0.012556, problem.basescale_fac))
            for row in problem.cov:
                f.write(" ".join("%.17g" % v for v in row) + "\n")
            f.write(" ".join("%.17g" % v for v in problem.halfwidths) + "\n")
            spec = f.name
        try:
            out = subprocess.check_output([drv, "bench", spec], timeout=600).decode().strip().splitlines()[-1]
            r = json.loads(out)
            return {"value": r["steps_per_s"], "unit": "MH steps/s", "cores": 1, "kind": "reference",
                    "sample": "parallel_tempering_chains::step, D=%d, %d rungs x 1 ladder, %d steps after %d warm-up, one "
                              "gaussian_prop(cov_r) per rung (the GPU workload's per-rung covariances), OpenMP inert as written "
                              "(1 thread)" % (D, NT, nsteps, nsteps // 10 + 1)}
        except Exception as ex:  # fall through to the port
            sys.stderr.write("[bench] reference driver failed (%s); timing the C port instead\n" % ex)
        finally:
            os.unlink(spec)
    return cpu_port(problem, budget_s)   # fallback: the C restatement, all host cores


def usable_cpus():
    """the processors this process may really use: its affinity mask cut by a cgroup CPU quota (a GPU box shows 256 hardware
    threads to a container that may use 16 of them: more OpenMP threads than that only take turns)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if q != "max" and int(per) > 0:
            n = min(n, max(1, -(-int(q) // int(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())    # cgroup v1
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, -(-q // per)))
        except (OSError, ValueError):
            pass
    return n


def cpu_port(problem, budget_s=8.0):
    """the oracle's C restatement of the same step (oracle/ptm_oracle.c), OpenMP over chains on ALL host cores: what a
    straightforward multi-core CPU implementation with a counter-based RNG reaches on this box"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    ncore = max(1, min(usable_cpus(), 64))
    W = 4
    pb = O.Problem(D)
    pb.set_bounds([0] * D, [0] * D, [0.0] * D, [0.0] * D)
    pb.set_prior(problem.types, problem.centers, problem.halfwidths)
    pb.set_gauss(problem.P, problem.like0)
    lad = O.Ladder(pb, problem.beta, W=W, swap_rate=SWAP_RATE)
    fac = problem.proposal_factors()
    lad.set_proposals([(O.PROP_DENSE, fac[r], 0.0) for r in range(NT)])
    lad.use_philox(SEED)
    lad.init_from_prior(SEED)
    lad.pt_step(3, ncore)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < budget_s:
        lad.pt_step(5, ncore)
        n += 5
    dt = time.perf_counter() - t0
    return {"value": n * NT * W / dt, "unit": "MH steps/s", "cores": ncore, "kind": "port",
            "sample": "oracle/ptm_oracle.c pt_step, D=%d, %d rungs x %d ladders, %d steps, OpenMP over chains" % (D, NT, W, n)}


# Independent ladders batched per GPU.  Throughput still grows a little with the batch (launch tails amortise: 6.7e9 steps/s
# at 4096, 7.0e9 at 8192, 7.1e9 at 16384 on chains fresh from the prior); 16384 x 1024 rungs = 16.8 M chains = 4.8 GB of state.
DEFAULT_WALKERS = 16384


def settle(eng, seconds=0.25):
    """set-up, before the W warm-up steps: run the ladder for a quarter of a second so that the GPU's clocks have ramped
    and the chains have left their prior draws (the acceptance pattern, and with it the row traffic, is then the
    stationary one).  Not timed, not counted."""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        eng.step(20)
        eng.sync()


def run_single(args):
    from ptmcmc_amd import engine as E
    from ptmcmc_amd.problems import GaussianProblem
    if E.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X (no gfx950 device visible); there is no CPU fallback")
    pr = GaussianProblem(D, NT, TMAX)
    W = args.walkers
    eng = E.Engine(D, NT, W, seed=SEED, swap_rate=SWAP_RATE, add_every_n=100, time_kernels=True)
    pr.configure(eng, E.PROP_LOWER)
    eng.init_from_prior()
    settle(eng)
    # what THIS device gives a streaming copy and an f64 fma loop right now (a few hundred ms, before anything is timed): boxes of
    # the pool differ by several per cent in the clock they hold under an f64 load, and the sweep kernel is bound by f64 issue
    calibration = None if args.no_calibration else eng.calibrate()
    # The GPU must not idle between the warm-up and the timed steps: after ~20 ms without work the device drops its clocks and the
    # next ~100 ms of kernels run up to 40 % slower (profiles/r04_sweep_duration_series.txt: 1.67, 2.14, 1.91, 1.75 ... 1.42 ms
    # after every gap).  Round 3's line read 67 MB of counters back right here and so timed exactly that ramp.  So: everything the
    # host wants to know first (two sums reduced on the device: 16 bytes), then clocks up again, warm-up, and straight into the
    # timed steps with nothing but the contract's synchronisation in between.
    settle(eng, 0.25)
    eng.step(args.warmup)            # (queued behind the settling steps: the device never idles from here to the end of the timed steps)
    tries0 = eng.counter_sums()[0]   # MH_chain::Ntries counts the Metropolis moves made (chain.cc:1005); exchanged rungs make none
    eng.kernel_times(drop=True)      # forget the records so far, unread (a few hundred event queries would be a gap of their own)
    eng.sync()
    eng.timer_start()
    t0 = time.perf_counter()
    eng.step(args.steps)
    ms_dev = eng.timer_stop()
    eng.sync()
    wall = time.perf_counter() - t0
    kt = eng.kernel_times()
    moved = (eng.counter_sums()[0] - tries0) / float(args.steps)   # chains the sweep kernel worked on, per launch
    nchains = NT * W
    value = nchains * args.steps / wall
    roof = roofline_record(eng.sweep_kernel_name, float(kt.mean()), int(kt.size), moved, nchains, value, 1,
                           measured_traffic(eng.sweep_kernel_name) if W == DEFAULT_WALKERS else None, kt=kt, calibration=calibration)
    acc = float((eng.naccept.sum() - eng.Nc)) / max(1, float((eng.ntries.sum() - eng.Nc)))
    t, a = eng.swap_counts()
    # latency-bound companion: the bare 1024-chain ladder (W = 1, the reference's own shape and BASELINE's literal one)
    w1 = None
    if not args.no_w1:
        e1 = E.Engine(D, NT, 1, seed=SEED, swap_rate=SWAP_RATE, add_every_n=100)
        pr.configure(e1, E.PROP_LOWER)
        e1.init_from_prior()
        e1.step(200); e1.sync()
        n1, best = 2000, None
        for _ in range(3):
            t1 = time.perf_counter()
            e1.step(n1); e1.sync()
            d1 = time.perf_counter() - t1
            best = d1 if best is None else min(best, d1)
        us = best / n1 * 1e6
        w1 = {"chains": NT, "value": NT * n1 / best, "ms_per_step": us * 1e-3, "us_per_step": us, "steps_per_launch": n1,
              "kernel": e1.step_kernel_name,
              "roofline": {"bound": "latency", "note": "1024 chains are 512 wavefronts on a 256-CU part: neither HBM nor the f64 pipes are near "
                           "a limit; what bounds a step is the chain of dependent instructions of a chains' wave (one issued per 8-11 cycles "
                           "per wave; the exchange phase's replay runs beside it on bookkeeper waves) and the neighbour hand-over through "
                           "memory: flag and window round trips of ~0.8 us each (DESIGN.md section 3.8)",
                           "hbm_frac_if_it_were_streaming": NT * n1 / best * algorithmic_bytes(D) / (HBM_PEAK_GBS * 1e9)}}
        e1.close()
    # ... and the same ladder with EVERYTHING ptmcmc_sampler switches on by default (ptmcmc.cc:60-143,389,512,601-616): 80 % differential
    # evolution from the chain's saved history + six Gaussians with one-dimensional moves, pry_temps after every accepted exchange,
    # the history ring (every second add) and MAP tracking -- the persistent ladder kernel's build FL = 15
    w1_defaults = None
    if not args.no_w1:
        n1, warm, every = 1000, 200, 2
        e3 = E.Engine(D, NT, 1, seed=SEED, swap_rate=SWAP_RATE, add_every_n=every, history_rungs=NT, history_capacity=(warm + 3 * n1) * 2 // every + 64, map_rungs=NT)
        pr.configure(e3, E.PROP_DIAG)
        K = 6
        g = 2.0 ** np.arange(1, K + 1)
        cum = np.tile(np.cumsum(np.concatenate([[0.8], 0.2 * g / g.sum()])), (NT, 1)); cum[:, -1] = 1.0
        e3.set_proposal_mixture(cum, np.tile(np.concatenate([[-1.0], 4.0 ** -np.arange(K)[::-1]]), (NT, 1)), np.tile(np.concatenate([[0.0], np.full(K, 0.5)]), (NT, 1)))
        e3.init_from_prior()
        init = np.random.default_rng(1).uniform(-1.0, 1.0, size=(50 * D, NT, D)) * np.asarray(pr.halfwidths)[None, None, :] * 0.02
        e3.set_proposal_de(0.1, 0.3, 4.0, 0.0, init_rows=init)
        e3.set_evolve_temps(0.01)
        e3.step(warm); e3.sync()
        best = None
        for _ in range(3):
            t1 = time.perf_counter()
            e3.step(n1); e3.sync()
            d1 = time.perf_counter() - t1
            best = d1 if best is None else min(best, d1)
        w1_defaults = {"chains": NT, "value": NT * n1 / best, "us_per_step": best / n1 * 1e6, "steps_per_launch": n1, "kernel": e3.step_kernel_name,
                       "what": "default proposal set (differential evolution 0.8 + six Gaussians, gauss_1d_frac 0.5), evolving ladder (rate 0.01), "
                               "history every 2nd add, MAP tracking"}
        e3.close()
    # ... and the headline workload with EVOLVING ladders (pry_temps after every accepted exchange: the reference sampler's
    # default, ptmcmc.cc:389,512): the same population, per-ladder temperatures
    evolving = None
    if not args.no_w1 and W == DEFAULT_WALKERS:
        e2 = E.Engine(D, NT, W, seed=SEED, swap_rate=SWAP_RATE, add_every_n=100)
        pr.configure(e2, E.PROP_LOWER)
        e2.set_evolve_temps(0.01)
        e2.init_from_prior()
        e2.step(120); e2.sync()
        t2 = time.perf_counter()
        e2.step(20); e2.sync()
        d2 = time.perf_counter() - t2
        evolving = {"evolve_rate": 0.01, "value": nchains * 20 / d2, "ms_per_step": d2 / 20 * 1e3, "kernel": e2.sweep_kernel_name}
        e2.close()
    out = {
        "metric": "ladder-wide MH steps/sec (D=32 Gaussian, 1024 temps)",
        "value": value, "unit": "MH steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "D=32 correlated Gaussian, 1024-rung ladder (Tmax=1e9, swap_rate=0.1) x %d walkers; "
                               "per-rung Cholesky proposal factors; uniform box prior" % W,
                   "layout": LAYOUT_NOTE,
                   "dim": D, "rungs": NT, "walkers": W, "chains": nchains, "sharding": "1 GPU holds the whole ladder"},
        "roofline": roof,
        "calibration": calibration,
        "device_ms_per_step": ms_dev / args.steps,
        "mh_accept_rate": acc, "swap_accept_rate": float(a.sum()) / max(1, float(t.sum())),
        "w1": w1,
        "w1_sampler_defaults": w1_defaults,
        "evolving_ladders": evolving,
    }
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(pr)
        if out["cpu_baseline"]["kind"] == "reference":
            out["cpu_port_all_cores"] = cpu_port(pr)
    eng.close()
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--walkers", type=int, default=DEFAULT_WALKERS, help="independent ladders batched per GPU")
    ap.add_argument("--halo", type=int, default=None, help="llike halo depth (rungs) between shards (default: ptmcmc_amd.parallel.DEFAULT_HALO)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-w1", action="store_true", help="skip the 1024-chain latency companion")
    ap.add_argument("--no-calibration", action="store_true", help="skip the copy / f64-fma calibration of the device (ptm_calibrate)")
    ap.add_argument("--shard", choices=("rungs", "walkers"), default="rungs",
                    help="N > 1: how the population is spread -- contiguous rung blocks with neighbour exchanges over RCCL (BASELINE's "
                         "configuration, the default), or whole ladders per GPU (no message at all; the form evolving ladders need)")
    ap.add_argument("--native-rccl", action="store_true",
                    help="N > 1, --shard rungs: drive the sharded step through the engine library's own RCCL calls (ptm_shard_*: the C/C++ "
                         "host's path; torch.distributed only hands the communicator id round and times the run) instead of "
                         "ptmcmc_amd.parallel.ShardedLadder over torch.distributed")
    ap.add_argument("--force-dist", action="store_true", help="take the torch.distributed path even with one rank (smoke test)")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started by hand without the launcher: run the ranks as child processes (nothing has touched the GPU yet)
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", os.environ.get("MASTER_PORT", "29533"), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    if args.gpus > 1 or world > 1 or args.force_dist:
        from ptmcmc_amd import parallel
        if args.halo is None:
            args.halo = parallel.DEFAULT_HALO
        return parallel.bench_main(args)
    run_single(args)


if __name__ == "__main__":
    main()
