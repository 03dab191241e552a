"""The oracle (oracle/ptm_oracle.c) against fixtures produced by the REAL reference
(tests/golden/make_golden.py) and against the reference's own golden file.  CPU only."""
import math

import numpy as np
import pytest

import golden_io
import lisa_toy
import oracle_lib as O

TOL = 1e-10   # |delta log-posterior| bar of BASELINE.json north_star


def close(a, b, tol=TOL):
    if math.isinf(a) or math.isinf(b) or math.isnan(a) or math.isnan(b):
        return (a == b) or (math.isnan(a) and math.isnan(b))
    return abs(a - b) <= tol * max(1.0, abs(b))


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    assert O.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert O.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert O.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_uniform_map_matches_newran():
    # newran1.cxx:432: ((double)seed + 0.5) / 4294967296.0 -- open interval, log() always finite
    L = O.lib()
    assert L.ptmo_u01(0) == 0.5 / 4294967296.0
    assert L.ptmo_u01(0xffffffff) == (4294967295.0 + 0.5) / 4294967296.0 < 1.0


def test_elementary_functions_vs_libm():
    L = O.lib()
    rng = np.random.default_rng(7)
    for x in np.concatenate([rng.uniform(1e-300, 1, 3000), 10 ** rng.uniform(-320, 300, 3000), rng.uniform(.5, 2, 3000)]):
        assert abs(L.ptmo_log(float(x)) - math.log(x)) <= 4e-16 * max(1, abs(math.log(x)))
    for x in rng.uniform(-740, 700, 5000):
        assert abs(L.ptmo_exp(float(x)) - math.exp(x)) <= 4e-16 * math.exp(x) + 1e-320
    for x in rng.uniform(0, math.pi, 5000):
        assert abs(L.ptmo_sin_0_pi(float(x)) - math.sin(x)) <= 4e-16
    for x in rng.uniform(-math.pi / 2, math.pi / 2, 5000):
        assert abs(L.ptmo_cos_hpi(float(x)) - math.cos(x)) <= 4e-16
    assert L.ptmo_log(0.0) == -math.inf and math.isnan(L.ptmo_log(-1.0)) and L.ptmo_exp(-800.0) == 0.0


def test_boxmuller_matches_formula_and_moments():
    rng = np.random.default_rng(11)
    ks = rng.integers(0, 2 ** 32, size=(40000, 2), dtype=np.uint64)
    z = np.array([O.boxmuller(int(a), int(b)) for a, b in ks])
    u1 = (ks[:, 0].astype(float) + .5) / 2 ** 32
    th = 2 * np.pi * (ks[:, 1].astype(float) + .5) / 2 ** 32
    ref = np.stack([np.sqrt(-2 * np.log(u1)) * np.cos(th), np.sqrt(-2 * np.log(u1)) * np.sin(th)], 1)
    assert np.abs(z - ref).max() < 1e-13
    assert abs(z.mean()) < 0.02 and abs(z.var() - 1) < 0.02
    # extreme draws stay finite
    for k1, k2 in [(0, 0), (0xffffffff, 0xffffffff), (0, 0x80000000), (1, 0x1fffffff), (5, 0x20000000)]:
        a, b = O.boxmuller(k1, k2)
        assert math.isfinite(a) and math.isfinite(b)


def test_boxmuller_normals_are_normal():
    """distribution checks of the table-driven Box-Muller on 4e5 draws of a Philox stream: moments up to the 6th,
    a Kolmogorov-Smirnov distance, tail frequencies, independence of the pair, and symmetry."""
    from scipy import stats
    n = 200000
    z = np.empty((n, 2))
    for i in range(0, n, 2):
        o = O.draw_block(12345, 0, 7, i // 2, 1)
        z[i] = O.boxmuller(o[0], o[1])
        z[i + 1] = O.boxmuller(o[2], o[3])
    a = z.ravel()
    m = a.size
    assert abs(a.mean()) < 4 / np.sqrt(m)
    assert abs(a.var() - 1) < 4 * np.sqrt(2 / m)
    assert abs(stats.skew(a)) < 4 * np.sqrt(6 / m)
    assert abs(stats.kurtosis(a)) < 4 * np.sqrt(24 / m)                    # excess kurtosis
    assert abs((a ** 6).mean() - 15) < 4 * np.sqrt((10395 - 225) / m)      # E z^6 = 15, Var z^6 = 10395 - 15^2
    assert stats.kstest(a, "norm").statistic < 1.63 / np.sqrt(m)            # 1 % critical value
    for t in (1.0, 2.0, 3.0, 4.0):
        p = 2 * stats.norm.sf(t)
        assert abs((np.abs(a) > t).mean() - p) < 4.5 * np.sqrt(p / m), t
    assert abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 4 / np.sqrt(n)
    assert abs(np.corrcoef(z[:, 0] ** 2, z[:, 1] ** 2)[0, 1]) < 4 / np.sqrt(n)
    # exact symmetry: flipping the half-turn bit of the angle draw flips both normals
    for k1, k2 in [(5, 77), (0xdeadbeef, 0x12345678), (0xffffffff, 0x7fffffff)]:
        p0, p1 = O.boxmuller(k1, k2), O.boxmuller(k1, k2 ^ 0x80000000)
        assert p0[0] == -p1[0] and p0[1] == -p1[1]


def test_boundary_enforce_table():
    g = golden_io.load("basic.json.gz")["boundary"]
    assert len(g) > 300
    for c in g:
        ok, y = O.boundary_enforce(c["lo"], c["hi"], c["xmin"], c["xmax"], c["x"])
        assert ok == c["ok"], c
        if ok:
            assert close(y, c["y"], 1e-12), (c, y)


def _problem_from(cfg):
    D = len(cfg["types"])
    pb = O.Problem(D)
    pb.set_bounds(cfg["blo"], cfg["bhi"], cfg["bmin"], cfg["bmax"])
    pb.set_prior(cfg["types"], cfg["centers"], cfg["halfwidths"])
    return pb


def test_prior_tables():
    n = 0
    for cfg in golden_io.load("basic.json.gz")["priors"]:
        pb = _problem_from(cfg)
        for case in cfg["cases"]:
            ok, xe = pb.enforce(case["x"])
            assert ok == case["valid"], (cfg["name"], case)
            if ok:
                assert np.allclose(xe, case["xe"], rtol=0, atol=1e-12)
            lp = pb.lprior(xe, ok)
            exp = case["lprior"]
            # Q2: log(prod pdf) loses precision once the product is subnormal (< ~1e-308, lprior < -709)
            tol = TOL if not (math.isfinite(exp) and exp < -700) else 1e-6
            assert close(lp, exp, tol), (cfg["name"], case, lp)
            n += 1
    assert n > 150


def test_ladder():
    for l in golden_io.load("basic.json.gz")["ladders"]:
        b = O.ladder(l["ntemps"], l["tmax"])
        assert np.allclose(b, l["invtemps"], rtol=1e-14, atol=0)
        assert b[0] == 1.0


def test_gaussian_target_values():
    for g in golden_io.load("gauss_target.json"):
        D = g["D"]
        pb = O.Problem(D)
        pb.set_gauss(np.array(g["invcov"]).reshape(D, D), g["like0"])
        for x, ll in zip(g["x"], g["llike"]):
            assert close(pb.llike(x), ll), (D, ll)


def test_reference_exampleLISA_golden_rows():
    """The 31 prior-draw rows of test/exampleLISA/exampleLISA_test_0_t0.dat: (lpost, llike, params) printed with
    13 significant digits.  Pins the mixed prior (uniform/polar/copolar) and the plug-in likelihood."""
    rows = golden_io.load("lisa_init_rows.json")
    assert len(rows) == 31
    pb = O.Problem(6)
    pb.set_bounds(lisa_toy.BLO, lisa_toy.BHI, lisa_toy.BMIN, lisa_toy.BMAX)
    pb.set_prior(lisa_toy.TYPES, lisa_toy.CENTERS, lisa_toy.SCALES)
    pb.set_user(lisa_toy.loglike)
    for r in rows:
        ok, xe = pb.enforce(r["x"])
        assert ok
        ll = pb.llike(xe)
        lp = pb.lprior(xe, 1)
        # parameters carry 13 digits; llike ~ 1e4..5e5 with gradient ~1e5 => absolute agreement ~1e-6 relative
        assert abs(ll - r["llike"]) <= 2e-9 * abs(r["llike"]) + 1e-6, (r, ll)
        assert abs((lp + ll) - r["lpost"]) <= 2e-9 * abs(r["lpost"]) + 1e-6, (r, lp)
        assert abs(lp - (r["lpost"] - r["llike"])) < 5e-7


@pytest.mark.parametrize("tid", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
def test_reference_pt_trace(tid):
    """Replay a real parallel_tempering_chains run: same initial states, same uniforms (recorded tapes of the
    reference's MotherOfAll generators), same scripted proposal offsets => the restatement must hold the same
    state on every rung after every PT step (chain.cc:1393-1571 + 966-1022 semantics incl. quirks Q1,Q5,Q6,Q7)."""
    g = golden_io.load("trace%d.json.gz" % tid)
    D, Nt, ns = g["D"], g["Nt"], g["nsteps"]
    pb = O.Problem(D, min_prior=g["minPrior"])
    pb.set_bounds(g["blo"], g["bhi"], g["bmin"], g["bmax"])
    pb.set_prior(g["types"], g["centers"], g["scales"])
    pb.set_gauss(np.array(g["P"]).reshape(D, D), g["like0"])
    beta = O.ladder(Nt, g["Tmax"])
    assert np.allclose(beta, g["invtemps"], rtol=1e-14)
    lad = O.Ladder(pb, g["invtemps"], W=1, swap_rate=g["swap_rate"], add_every_N=g["add_every_N"])
    assert lad.s.contents.maxswaps == g["maxswaps"]
    lad.set_proposals([(O.PROP_DIAG, np.ones(D), 0.0)] * Nt)   # unused: offsets come from the tape
    lad.use_tape(np.array(g["chain_tapes"]), np.array(g["pt_tape"])[None, :], golden_io.trace_deltas(g))
    if "log_hastings" in g:
        # trace 10: every scripted offset carries a log-Hastings ratio (and a type code k % 3), MH_chain::step chain.cc:989-994
        lad.tape_hastings(np.array(g["log_hastings"]), np.tile(np.arange(ns) % 3, (Nt, 1)))
    lad.enable_history(2 * ns + 4)
    lad.set_states(np.array([c["x"] for c in g["init"]]))
    evolve = g.get("evolve_rate", 0.0) > 0
    if evolve:
        # traces 5, 6: evolve_temps(rate) -- pry_temps after every accepted exchange (chain.cc:1501-1518,1809-1846).  The
        # restatement keeps the gaps lazily normalised, so temperatures agree to rounding, not to the bit.
        # traces 11, 12: with evolve_temp_lpost_cut >= 0 every pry also widens the gaps whose chains' posteriors are out of
        # order by more than cut * invtemp (chain.cc:1819-1827)
        lad.evolve_temps(g["evolve_rate"], g.get("evolve_lpost_cut", -1.0))
    for r, c in enumerate(g["init"]):
        assert close(lad.llike[r], c["llike"]) and close(lad.lpost[r], c["lpost"])
    nswapped = 0
    for k in range(ns):
        lad.pt_step()
        x, ll, lp, nsz = lad.x, lad.llike, lad.lpost, lad.nsize
        nswapped += int(lad.last_accept.sum())
        for r, c in enumerate(g["steps"][k]):
            if "x" in c:    # (compact fixtures, traces 7-9: every 5th step; llike pins the state on the others)
                assert np.array_equal(x[r], np.array(c["x"])), (tid, k, r, x[r], c["x"])
            assert close(ll[r], c["llike"]), (tid, k, r)
            assert close(lp[r], c["lpost"]), (tid, k, r)
            assert nsz[r] == c["size"], (tid, k, r, nsz[r], c["size"])
        if evolve:
            want = np.array([c["invtemp"] for c in g["steps"][k]])
            assert np.allclose(lad.betaw[0], want, rtol=1e-12, atol=0), (tid, k, lad.betaw[0] - want)
    # what every add_state call pushed (MH_chain::lposts / llikes, chain.cc:935-946) -- with an evolving ladder the rows of
    # an exchange phase carry the log-posterior at the temperature the rung had between two pries of that step
    h = lad.history()
    for r in range(Nt):
        n = len(g["hist_lpost"][r])
        assert n == lad.nsize[r]
        for e in range(n):
            assert close(h["llike"][r, e], g["hist_llike"][r][e]), (tid, r, e)
            lpo = O.lib().ptmo_lpost(h["lprior"][r, e], h["invtemp"][r, e], h["llike"][r, e])
            assert close(lpo, g["hist_lpost"][r][e]), (tid, r, e, lpo, g["hist_lpost"][r][e])
    if "hist_type" in g:         # the type MH_chain::add_state pushed with every row (rows 1.. of the raw history)
        for r in range(Nt):
            assert list(h["last_type"][r, 1:lad.nsize[r]]) == g["hist_type"][r], (tid, r)
        assert len(set(sum(g["hist_type"], []))) >= 3
        hs = np.array(g["log_hastings"])
        assert (hs != 0).mean() > 0.8 and (hs > 0).any() and (hs < 0).any()
    assert nswapped > 5          # the trace really exercised accepted exchanges
    if evolve:
        moved = np.abs(lad.betaw[0] - np.array(g["invtemps"]))[1:-1]
        assert (moved > 1e-4).all() and lad.betaw[0][0] == 1.0 and lad.betaw[0][-1] == g["invtemps"][-1]
    assert (lad.ntries > (20 if ns > 100 else 3)).all()
    if tid == 9:
        assert lad.s.contents.maxswaps == 205 and nswapped > 400
    if tid == 3:
        # quirk Q9 (states.cc:183-192,205-214): the origin violates a `limit` bound, so every state::add() result is
        # born invalid and the reference rejects every MH move; only exchanges move states.
        assert pb.origin_valid == 0 and lad.naccept.sum() == Nt
    else:
        assert pb.origin_valid == 1 and lad.naccept.sum() > Nt + 20


def test_prior_given_as_a_function_equals_the_described_prior():
    """ptmo_problem_set_user_prior (the checker's side of ptm_set_prior_callback): a prior handed over as a function of the valid
    state replaces the per-dimension description.  Handing over the mixed uniform / polar / copolar prior of exampleLISA as
    a function (the checker's own ptmo_lprior of a second, described problem) must give the very chain the description gives --
    the reference's prior gate (chain.cc:980) and exchange phase see the same numbers either way."""
    D, Nt, W = 6, 5, 2
    beta = O.geometric_ladder(Nt, 1e4) if hasattr(O, "geometric_ladder") else np.exp(-np.log(1e4) * np.arange(Nt) / (Nt - 1))
    rng = np.random.default_rng(3)
    lo = np.array(lisa_toy.CENTERS) - np.array(lisa_toy.SCALES)
    hi = np.array(lisa_toy.CENTERS) + np.array(lisa_toy.SCALES)
    x0 = rng.uniform(lo + 0.05, hi - 0.05, size=(Nt * W, D))
    sig = np.array(lisa_toy.SCALES) / 10.0
    fac = np.tile(sig, (Nt, 1)) / np.sqrt(beta)[:, None].clip(1e-3)
    described = O.Problem(D)
    described.set_bounds(lisa_toy.BLO, lisa_toy.BHI, lisa_toy.BMIN, lisa_toy.BMAX)
    described.set_prior(lisa_toy.TYPES, lisa_toy.CENTERS, lisa_toy.SCALES)
    described.set_user(lisa_toy.loglike)
    handed = O.Problem(D)
    handed.set_bounds(lisa_toy.BLO, lisa_toy.BHI, lisa_toy.BMIN, lisa_toy.BMAX)
    handed.set_user(lisa_toy.loglike)
    asked = []
    def prior(x):
        asked.append(x.copy())
        return described.lprior(x, 1)
    handed.set_user_prior(prior)
    lads = []
    for pb in (described, handed):
        lad = O.Ladder(pb, beta, W=W, swap_rate=0.3)
        lad.set_proposals([(O.PROP_DIAG, fac[r], 0.3) for r in range(Nt)])
        lad.use_philox(0x5EED0001)
        lad.set_states(x0)
        lad.pt_step(40)
        lads.append(lad)
    a, b = lads
    assert np.array_equal(a.x, b.x) and np.array_equal(a.lprior, b.lprior) and np.array_equal(a.llike, b.llike)
    assert np.array_equal(a.naccept, b.naccept) and a.naccept.sum() > Nt * W
    assert len(asked) > Nt * W     # start states, then the valid proposals
    X = np.array(asked)
    assert (X[:, 0] >= lisa_toy.BMIN[0]).all() and (X[:, 0] <= lisa_toy.BMAX[0]).all()   # never asked about an invalid state


def test_oracle_differential_evolution_matches_the_reference_draw_by_draw():
    """The oracle's restatement of differential_evolution::draw (ptmo_de_draw; proposal_distribution.cc:476-592,745-801) against the
    REAL reference: tests/golden/trace13.json.gz holds, per draw, the rung's saved history, the current state, the uniforms the
    reference's generator delivered and what the reference proposed.  Fed those uniforms in sequence the oracle proposes the same
    state (1e-12 relative), the same log-Hastings ratio and type, and asks for no more uniforms than the reference drew -- every draw
    without temperature mixing and with unlikely_alpha = 0 (the sampler's defaults, ptmcmc.cc:81-91: what the engine draws on the
    device), parallel and snooker moves, short histories and ones long enough for the ignored early fraction."""
    g = golden_io.load("trace13.json.gz")
    seen = set()
    for c in g["cases"]:
        D = c["D"]
        for q in c["draws"]:
            if q["mixing"] or q["unlikely_alpha"] > 0:
                continue
            r = c["rungs"][q["rung"]]
            u = [(k + 0.5) / 4294967296.0 for k in q["uniform_k"]]     # MotherOfAll::Next (newran1.cxx:432)
            t, xn, lh, used = O.de_draw(q["x"], r["x"], u, q["snooker"], q["gamma_one_frac"], q["reduce_gamma"], q["ignore_frac"])
            what = (q["variant"], q["rung"])
            assert t == q["type"] and q["valid"] == 1, what
            want = np.array(q["proposed"])
            assert np.allclose(xn, want, rtol=1e-12, atol=1e-12 * np.abs(np.array(q["x"])).max()), (what, xn, want)
            assert abs(lh - q["log_hastings"]) <= 1e-12 * max(1.0, abs(q["log_hastings"])), (what, lh, q["log_hastings"])
            # a snooker move uses every uniform the reference drew; a parallel move leaves the D normals of the small jump it discards
            assert used == len(u) if t == 1 else used == 4 and len(u) - used >= D, (what, used, len(u))
            seen.add((t, r["size"] > 110 * D, q["ignore_frac"] > 0))
    assert {k[0] for k in seen} == {0, 1} and len({k[1] for k in seen}) == 2, seen
