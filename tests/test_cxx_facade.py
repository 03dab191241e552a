"""The host-side C++ mirror of the reference's plug-in surface (ptmcmc_amd/host/ptmcmc_gpu.hh): it must compile as
plain C++11 against the C ABI alone (CPU test) and drive the engine on the GPU (gpu test)."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(out, src="example_gaussian_pt.cc"):
    cmd = ["g++", "-std=c++11", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "ptmcmc_amd", "host"),
           os.path.join(ROOT, "examples", src), "-L", os.path.join(ROOT, "ptmcmc_amd"), "-lptm_engine",
           "-Wl,-rpath," + os.path.join(ROOT, "ptmcmc_amd"), "-pthread", "-o", out]
    subprocess.check_call(cmd)


def test_facade_compiles_as_cxx11_against_the_c_abi():
    with tempfile.TemporaryDirectory() as d:
        build(os.path.join(d, "ex2"), "example_sampler.cc")
        build(os.path.join(d, "ex3"), "example_lisa.cc")
        build(os.path.join(d, "ex"))
        # without a GPU the program must fail loudly, not compute on the host
        if not os.path.exists("/dev/kfd"):
            r = subprocess.run([os.path.join(d, "ex")], capture_output=True, text=True)
            assert r.returncode != 0 and "no gfx950" in r.stdout


@pytest.mark.gpu
def test_facade_runs_pt_on_device_and_with_user_callback():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "ex")
        build(exe)
        outs = {}
        for mode in ("device", "callback"):
            r = subprocess.run([exe, mode, "4", "8", "3000"], capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stdout + r.stderr
            outs[mode] = r.stdout.splitlines()[0]
        f = lambda line, key: float(line.split(key + "=")[1].split()[0])
        # tridiag(-0.4,1,-0.4)^-1 for D=4: var(x0) = 1.2613, var(x3) the same by symmetry
        P = np.eye(4) + np.diag([-0.4] * 3, 1) + np.diag([-0.4] * 3, -1)
        v0 = np.linalg.inv(P)[0, 0]
        for mode, line in outs.items():
            assert abs(f(line, "var(x0)") - v0) < 0.25 * v0, line
            assert abs(f(line, "var(x3)") - v0) < 0.25 * v0, line
            assert abs(f(line, "beta_top") - 0.01) < 1e-12
        # the device target and the user callback compute the same likelihood => the same chain
        assert outs["device"].split("var")[1:] == outs["callback"].split("var")[1:] or \
            abs(f(outs["device"], "var(x0)") - f(outs["callback"], "var(x0)")) < 1e-6
        assert int(outs["callback"].split("likelihood_calls=")[1]) > 1000
        assert int(outs["device"].split("likelihood_calls=")[1]) == 0


@pytest.mark.gpu
def test_facade_writes_the_cold_chain_file_from_the_device_history():
    """MH_chain::dumpChain's format (chain.cc:1112-1135) from the history ring: header, then
    'i lpost llike acceptance_ratio type: p0 .. pD-1 invtemp' for every saved step after burn-in."""
    with tempfile.TemporaryDirectory() as d:
        exe, out = os.path.join(d, "ex"), os.path.join(d, "chain.dat")
        build(exe)
        r = subprocess.run([exe, "device", "4", "8", "4000", out], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "instances are a permutation of the rungs" in r.stdout and " 0 displaced" not in r.stdout
        mp = [l for l in r.stdout.splitlines() if l.startswith("MAP lpost=")][0]
        assert float(mp.split("MAP lpost=")[1].split()[0]) >= float(mp.split("lpost now ")[1].rstrip(")"))
        ts = [l.split() for l in open(out + ".tempstats").read().splitlines() if l and not l.startswith("#")]
        assert len(ts) == 8 and float(ts[0][1]) == 1.0 and float(ts[7][1]) == 0.0
        assert all(0.0 <= float(t[1]) <= 1.0 for t in ts) and all(0.0 < float(t[2].rstrip(":")) <= 1.0 for t in ts[:7])
        lines = open(out).read().splitlines()
        assert lines[0] == "#Ninit=1, Nburn=1000"
        assert lines[1].startswith("#eval: log(posterior) log(likelihood) acceptance_ratio prop_type: x0 x1 x2 x3")
        rows = [l for l in lines[2:] if l]
        idx = np.array([int(l.split()[0]) for l in rows])
        assert idx[0] == 1000 and np.all(np.diff(idx) == 10) and len(rows) >= 395   # >= 4000 adds after burn-in, every 10th
        head, pars = zip(*(l.split(": ") for l in rows))
        H = np.array([[float(v) for v in h.split()] for h in head])
        X = np.array([[float(v) for v in p.split()] for p in pars])
        assert X.shape[1] == 5 and np.all(X[:, 4] == 1.0)                  # D parameters + invtemp of the cold rung
        assert np.all((H[:, 3] > 0) & (H[:, 3] <= 1)) and set(H[:, 4]) <= {-1.0, 0.0}
        P = np.eye(4) + np.diag([-0.4] * 3, 1) + np.diag([-0.4] * 3, -1)
        # flat box prior => lpost - llike is the prior constant; llike = -x'Px/2
        assert np.allclose(H[:, 2], -0.5 * np.einsum("ni,ij,nj->n", X[:, :4], P, X[:, :4]), atol=1e-9)
        assert np.allclose(H[:, 1] - H[:, 2], (H[:, 1] - H[:, 2])[0], atol=1e-9)
        v0 = np.linalg.inv(P)[0, 0]
        assert abs(X[:, 0].var() - v0) < 0.35 * v0


@pytest.mark.gpu
def test_sampler_driver_writes_the_reference_chain_files():
    """ptmcmc_sampler::run (ptmcmc.cc:563-607): every nevery steps dumpChain(ich, out, istep-nevery+1, nskip) for the
    pt_dump_n coldest chains into <base>_t<ich>.dat -- here from the device's history ring."""
    with tempfile.TemporaryDirectory() as d:
        exe, base = os.path.join(d, "ex2"), os.path.join(d, "run")
        build(exe, "example_sampler.cc")
        r = subprocess.run([exe, base, "--pt_evolve_rate=0"], capture_output=True, text=True, timeout=600)   # fixed ladder
        assert r.returncode == 0, r.stdout + r.stderr
        for ich, beta_expect in ((0, 1.0), (1, 50.0 ** (-1 / 5))):
            lines = open("%s_t%d.dat" % (base, ich)).read().splitlines()
            assert sum(l.startswith("#Ninit=1, Nburn=") for l in lines) == 5     # istep = 0, 500, ..., 2000
            rows = [l for l in lines if l and not l.startswith("#")]
            idx = np.array([int(l.split()[0]) for l in rows])
            assert idx[0] == -1 and idx[1] == 1 and np.mean(np.diff(idx) == 4) > 0.9   # the reference's loop: i = Nburn, Nburn + nskip, ...; -1 = the
                                                                   # initial state (chain.cc:1122-1125); a report restarts at istep-nevery+1 whatever the last one reached
            assert len(rows) > 400                                                # ~ >= 2000 adds / nskip 4
            X = np.array([[float(v) for v in l.split(": ")[1].split()] for l in rows])
            assert X.shape[1] == 4 and np.allclose(X[:, 3], beta_expect, rtol=1e-10)
            H = np.array([[float(v) for v in l.split(": ")[0].split()] for l in rows])
            P = np.array([[2.0, 0.6, 0.0], [0.6, 1.0, -0.3], [0.0, -0.3, 1.5]])
            assert np.allclose(H[:, 2], -0.5 * np.einsum("ni,ij,nj->n", X[:, :3], P, X[:, :3]), atol=1e-9)
            types = set(H[:, 4])                                  # member + 10 * (one-dimensional move), proposal_distribution.cc:117
            assert types <= {-1.0} | {float(k) for k in range(6)} | {float(10 + k) for k in range(6)} and len(types) >= 4


@pytest.mark.gpu
def test_sampler_replicas_run_side_by_side_and_do_not_change_each_other():
    """--nchains=N runs the reference's Nchain repeats as N replicas in one engine: replica 0's chain files are, byte
    for byte, those of a run with a single replica; the other replicas are different, valid chains."""
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "ex2")
        build(exe, "example_sampler.cc")
        one, many = os.path.join(d, "one"), os.path.join(d, "many")
        for base, n in ((one, 1), (many, 64)):   # (the sampler's default: the ladders evolve, pt_evolve_rate = 0.01)
            r = subprocess.run([exe, base, "--nchains=%d" % n, "--nsteps=600", "--nevery=200"], capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stdout + r.stderr
        for ich in (0, 1):
            assert open("%s_t%d.dat" % (one, ich)).read() == open("%s_t%d.dat" % (many, ich)).read()
        a = open(many + "_t0.dat").read()
        for w in (1, 17, 63):
            b = open("%s_c%d_t0.dat" % (many, w)).read()
            assert b != a and len(b.splitlines()) > 100


@pytest.mark.gpu
def test_sampler_default_run_evolves_the_ladder():
    """The sampler's defaults switch temperature evolution on (pt_evolve_rate 0.01, ptmcmc.cc:389,512): every row of a
    report carries the chain's temperature at the time of the report (chain.cc:1131), the cold chain stays at 1, the
    next rung's temperature drifts from report to report, and the log-posterior column is the one add_state computed
    at the temperature the rung had at that very add (lprior + beta*llike with beta between the reports' values)."""
    with tempfile.TemporaryDirectory() as d:
        exe, base = os.path.join(d, "ex2"), os.path.join(d, "run")
        build(exe, "example_sampler.cc")
        r = subprocess.run([exe, base], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        blocks = {}
        for ich in (0, 1):
            cur, out = None, []
            for l in open("%s_t%d.dat" % (base, ich)).read().splitlines():
                if l.startswith("#Ninit"):
                    cur = []
                    out.append(cur)
                elif l and not l.startswith("#"):
                    cur.append([float(v) for v in l.replace(":", " ").split()])
            blocks[ich] = [np.array(b) for b in out if len(b)]
        assert all(np.all(b[:, -1] == 1.0) for b in blocks[0])             # the cold chain never moves (chain.cc:1836)
        b1 = [b[0, -1] for b in blocks[1]]
        assert all(np.all(b[:, -1] == b[0, -1]) for b in blocks[1])        # one temperature per report ...
        assert len(set(b1)) == len(b1) and all(0.2 < v < 0.8 for v in b1)  # ... that drifts (start: 50^(-1/5) = 0.457)
        # lpost - llike*beta_row = lprior (a constant: uniform box) for SOME beta_row between the extremes seen
        P = np.array([[2.0, 0.6, 0.0], [0.6, 1.0, -0.3], [0.0, -0.3, 1.5]])
        allrows = np.concatenate(blocks[1])
        ll = -0.5 * np.einsum("ni,ij,nj->n", allrows[:, 5:8], P, allrows[:, 5:8])
        assert np.allclose(allrows[:, 2], ll, atol=1e-9)
        lp0 = blocks[0][0][0, 1] - blocks[0][0][0, 2]                      # cold chain: lpost - llike = lprior
        big = ll < -0.5
        brow = (allrows[big, 1] - lp0) / ll[big]
        assert brow.min() > 0.2 and brow.max() < 0.8 and brow.std() > 1e-4


@pytest.mark.gpu
def test_sampler_checkpoint_and_restart_give_the_uninterrupted_chain_files():
    """The reference's restart-identity test (test/exampleLISA/Makefile:14-17, cp-test): a run stopped by
    --checkp_at_step (ptmcmc.cc:567,593-596; files under ./step_<n>-cp/) and continued with --restart_dir writes, byte
    for byte, the chain files of the run that never stopped -- evolving ladder (the default), history ring, MAPs and all."""
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "ex2")
        build(exe, "example_sampler.cc")
        common = ["--nsteps=1200", "--nevery=300", "--nchains=2"]
        def run(*args):
            r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=600, cwd=d)
            assert r.returncode == 0, r.stdout + r.stderr
            return r.stdout
        run("whole", *common)
        out1 = run("parts", *common, "--checkp_at_step=500")
        assert "Checkpointing triggered." in out1 and os.path.exists(os.path.join(d, "step_500-cp", "ptmcmc.cp"))
        assert os.path.exists(os.path.join(d, "step_500-cp", "chain0-cp", "PTchain.cp"))
        part1 = open(os.path.join(d, "parts_t0.dat")).read()
        out2 = run("parts", *common, "--restart_dir=step_500-cp")
        assert "Restarting from checkpoint files" in out2
        for name in ("_t0.dat", "_t1.dat", "_c1_t0.dat", "_c1_t1.dat"):
            a = open(os.path.join(d, "whole" + name)).read()
            b = open(os.path.join(d, "parts" + name)).read()
            assert a == b, name
        whole = open(os.path.join(d, "whole_t0.dat")).read()
        assert whole.startswith(part1) and len(part1) < len(whole)


@pytest.mark.gpu
def test_lisa_plugin_through_the_sampler():
    """BASELINE configs[4] end to end in C++: the toy LISA plug-in likelihood registered through
    bayes_likelihood::register_evaluate_log, the mixed uniform / polar / co-polar prior with limit and wrap boundaries,
    prior draws on the device, the sampler's default evolving ladder and chain files.  Every row's log-likelihood is the
    plug-in's value at the row's parameters (recomputed here), the parameters respect the state space, and the cold
    chain finds the injected signal (log-likelihood near 0 from ~ -1e4 at the prior draws)."""
    import lisa_toy
    with tempfile.TemporaryDirectory() as d:
        exe, base = os.path.join(d, "ex3"), os.path.join(d, "lisa")
        build(exe, "example_lisa.cc")
        r = subprocess.run([exe, base, "--nsteps=3000", "--nevery=1000", "--nchains=4"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "MAP: lpost" in r.stdout
        best = []
        for name in ("_t0.dat", "_c3_t0.dat"):
            rows = [l for l in open(base + name).read().splitlines() if l and not l.startswith("#")]
            assert len(rows) > 150
            R = np.array([[float(v) for v in l.replace(":", " ").split()] for l in rows])
            X, ll = R[:, 5:11], R[:, 2]
            want = np.array([lisa_toy.loglike(x) for x in X])
            assert np.allclose(ll, want, rtol=1e-8, atol=1e-5), np.abs(ll - want).max()
            lo = np.array(lisa_toy.BMIN); hi = np.array(lisa_toy.BMAX)
            assert (X >= lo - 1e-12).all() and (X <= hi + 1e-12).all()
            assert (R[:, -1] == 1.0).all()                       # the cold chain's temperature
            best.append(ll.max())
            assert ll[0] < -100 and ll[-50:].max() > -30, (ll[0], ll[-50:].max())
        assert max(best) > -10
