"""Parity tests proper: the HIP engine (through the C ABI) against the CPU oracle on the same seeded inputs.
Bar: bit-exact for the accept/reject index stream and everything integer; floating-point state bit-exact too
(same IEEE op sequence on both sides); |delta log-posterior| < 1e-10 against the reference-derived fixtures."""
import math

import numpy as np
import pytest

import golden_io
import oracle_lib as O
import parity_util as PU
from ptmcmc_amd import engine as E

pytestmark = pytest.mark.gpu
TOL = 1e-10


def test_device_present_and_abi():
    assert E.device_count() >= 1
    assert E.load().ptm_abi_version() == 3


def test_philox_on_device_matches_known_answers_and_oracle():
    # the engine's counter layout on (seed=0,tag=0,stream=0,step=0,block=0) is the all-zero Random123 vector
    assert E.debug_philox(0, 0, 0, 0, 0) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    rng = np.random.default_rng(3)
    for _ in range(20):
        seed = int(rng.integers(0, 2 ** 63)); tag = int(rng.integers(0, 3)); stream = int(rng.integers(0, 2 ** 32))
        step = int(rng.integers(0, 2 ** 40)); block = int(rng.integers(0, 300))
        assert E.debug_philox(seed, tag, stream, step, block) == O.draw_block(seed, tag, stream, step, block)


def test_elementary_functions_bit_exact():
    L = O.lib()
    rng = np.random.default_rng(5)
    n = 200000
    cases = {
        E.FN_LOG: (np.concatenate([rng.uniform(1e-300, 1, n), 10 ** rng.uniform(-320, 300, n), rng.uniform(.5, 2, n),
                                   (rng.integers(0, 2 ** 32, n) + .5) / 2 ** 32]), L.ptmo_log),
        E.FN_EXP: (np.concatenate([rng.uniform(-750, 710, n), rng.uniform(-2, 2, n)]), L.ptmo_exp),
        E.FN_SIN_0_PI: (rng.uniform(0, math.pi, n), L.ptmo_sin_0_pi),
        E.FN_COS_HPI: (rng.uniform(-math.pi / 2, math.pi / 2, n), L.ptmo_cos_hpi),
    }
    for fn, (xs, ofn) in cases.items():
        got = E.debug_eval(fn, xs)
        sub = rng.choice(xs.size, 20000, replace=False)
        exp = np.array([ofn(float(xs[i])) for i in sub])
        assert np.array_equal(got[sub], exp, equal_nan=True), fn
    # sqrt and division must be the IEEE correctly rounded results (what the CPU checker computes)
    xs = np.concatenate([rng.uniform(0, 50, n), 10 ** rng.uniform(-300, 300, n), -2 * np.log((rng.integers(0, 2 ** 32, n) + .5) / 2 ** 32)])
    assert np.array_equal(E.debug_eval(E.FN_SQRT, xs), np.sqrt(xs))
    a, b = rng.normal(size=n) * 10 ** rng.uniform(-50, 50, n), rng.normal(size=n) * 10 ** rng.uniform(-50, 50, n)
    assert np.array_equal(E.debug_eval(E.FN_DIV, a, b), a / b)


def test_boxmuller_bit_exact():
    rng = np.random.default_rng(9)
    k1 = rng.integers(0, 2 ** 32, 100000, dtype=np.uint64).astype(np.uint32)
    k2 = rng.integers(0, 2 ** 32, 100000, dtype=np.uint64).astype(np.uint32)
    k1[:6] = [0, 0xffffffff, 0, 1, 5, 0x80000000]; k2[:6] = [0, 0xffffffff, 0x80000000, 0x1fffffff, 0x20000000, 0x7fffffff]
    z0, z1 = E.debug_boxmuller(k1, k2)
    for i in list(range(6)) + list(rng.choice(100000, 5000, replace=False)):
        a, b = O.boxmuller(int(k1[i]), int(k2[i]))
        assert z0[i] == a and z1[i] == b, i
    assert abs(z0.mean()) < 0.02 and abs(z0.var() - 1) < 0.02 and abs(z1.var() - 1) < 0.02


def test_boxmuller_radius_sqrt_is_correctly_rounded_for_all_arguments():
    """The hot path takes sqrt(-2 ln u) with an unscaled v_rsq_f64 + Newton sequence; the CPU checker calls sqrt().
    All 2^32 possible arguments are scanned on the GPU against the exact-residual corrected root."""
    assert E.debug_sqrt_scan() == 0


def test_prior_and_boundary_tables_from_reference():
    """mixed_dist_product::evaluate_log and stateSpace::enforce on the device vs the real reference's values."""
    for cfg in golden_io.load("basic.json.gz")["priors"]:
        D = len(cfg["types"])
        eng = E.Engine(D, 2, 1)
        eng.set_bounds(cfg["blo"], cfg["bhi"], cfg["bmin"], cfg["bmax"])
        eng.set_prior(cfg["types"], cfg["centers"], cfg["halfwidths"])
        X = np.array([c["x"] for c in cfg["cases"]])
        valid, Xe, lp, _ = eng.debug_evaluate(X)
        pb = O.Problem(D)
        pb.set_bounds(cfg["blo"], cfg["bhi"], cfg["bmin"], cfg["bmax"])
        pb.set_prior(cfg["types"], cfg["centers"], cfg["halfwidths"])
        for k, c in enumerate(cfg["cases"]):
            assert valid[k] == c["valid"], (cfg["name"], k)
            ok, xo = pb.enforce(c["x"])
            if ok:
                assert np.array_equal(Xe[k], xo), (cfg["name"], k)          # bit-exact vs oracle
                assert np.allclose(Xe[k], c["xe"], rtol=0, atol=1e-12)       # and the reference's value
            lo = pb.lprior(xo, ok)
            assert (lp[k] == lo) or (math.isnan(lp[k]) and math.isnan(lo)), (cfg["name"], k, lp[k], lo)
            exp = c["lprior"]
            if math.isinf(exp):
                assert lp[k] == exp
            elif exp > -700:   # Q2: beyond that the reference's product is subnormal
                assert abs(lp[k] - exp) <= TOL * max(1, abs(exp)), (cfg["name"], k, lp[k], exp)
        eng.close()


def test_gaussian_target_values_from_reference_formula():
    for g in golden_io.load("gauss_target.json"):
        D = g["D"]
        eng = E.Engine(D, 2, 1)
        eng.set_target_gaussian(np.array(g["invcov"]).reshape(D, D), g["like0"])
        _, _, _, ll = eng.debug_evaluate(np.array(g["x"]))
        for a, b in zip(ll, g["llike"]):
            assert abs(a - b) <= TOL * max(1, abs(b)), (D, a, b)
        pb = O.Problem(D)
        pb.set_gauss(np.array(g["invcov"]).reshape(D, D), g["like0"])
        assert np.array_equal(ll, np.array([pb.llike(x) for x in g["x"]]))
        eng.close()


def test_init_from_prior_matches_oracle():
    pr, eng, lad = PU.make_pair(5, 6, 3, 1e3)
    lad2 = O.Ladder(lad.pb, pr.beta, W=3)
    lad2.init_from_prior(0x5EED0001)
    # same draws: oracle INIT stream is keyed by the oracle chain index w*Nt+r, as is the engine's
    assert np.array_equal(PU.to_engine_order(lad2.x, 6, 3), eng.states())
    assert np.array_equal(PU.to_engine_order(lad2.llike, 6, 3), eng.llike)
    eng.close()


def test_init_from_prior_draws_every_support_type():
    """MH_chain::initialize (chain.cc:846-876) with a mixed prior: uniform, gaussian, polar, copolar and log dimensions
    (inverse cdfs of ProbabilityDist.h:32-34,108-110,149-151) drawn on the device, bit-identical to the oracle's
    restatement of the same procedure, inside their supports and with the right marginals."""
    import math
    D, Nt, W = 5, 4, 4096
    pi = math.pi
    types = [1, 2, 3, 4, 5]
    cen = [0.5, 0.3, pi / 2, 0.1, 3.0]
    hw = [2.0, 1.5, pi / 2 - 0.2, 1.2, 2.5]          # polar on (0.2, pi - 0.2), copolar on (-1.1, 1.3), log on (1.2, 7.5)
    pr, eng, lad = PU.make_pair(D, Nt, W, 1e2, kind=E.PROP_DIAG, prior=(types, cen, hw), x0=np.tile(cen, (Nt * W, 1)))
    eng.init_from_prior()
    lad2 = O.Ladder(lad.pb, pr.beta, W=W)
    lad2.init_from_prior(0x5EED0001)
    x = eng.states()
    assert np.array_equal(PU.to_engine_order(lad2.x, Nt, W), x)
    assert np.array_equal(PU.to_engine_order(lad2.llike, Nt, W), eng.llike)
    assert np.isfinite(eng.lprior).all()
    lo, hi = np.array(cen) - np.array(hw), np.array(cen) + np.array(hw)
    lo[4], hi[4] = cen[4] / hw[4], cen[4] * hw[4]     # the log type's limits (probability_function.cc:219-262)
    for d in (0, 2, 3, 4):
        assert (x[:, d] >= lo[d]).all() and (x[:, d] <= hi[d]).all()
    n = len(x)
    # marginals: cdf of the draws at the support's midpoint against the analytic value (4 sigma of a binomial)
    def check(d, cdf_mid):
        got = (x[:, d] < cen[d]).mean()
        assert abs(got - cdf_mid) < 4 * math.sqrt(cdf_mid * (1 - cdf_mid) / n), (d, got, cdf_mid)
    check(0, 0.5)
    check(2, (math.cos(lo[2]) - math.cos(cen[2])) / (math.cos(lo[2]) - math.cos(hi[2])))
    check(3, (math.sin(cen[3]) - math.sin(lo[3])) / (math.sin(hi[3]) - math.sin(lo[3])))
    check(4, (math.log(cen[4]) - math.log(lo[4])) / (math.log(hi[4]) - math.log(lo[4])))
    eng.close()


SWEEP_CASES = [
    # (D, Nt, W, Tmax, kind, oneDfrac)      BASELINE configs first
    (2, 8, 1, 1e2, E.PROP_LOWER, None),      # C1
    (16, 64, 1, 1e4, E.PROP_LOWER, None),    # C2
    (32, 256, 4, 1e6, E.PROP_LOWER, None),   # C3
    (32, 16, 64, 1e6, E.PROP_LOWER, None),   # wave-uniform rung path
    (32, 8, 128, 1e3, E.PROP_DENSE, None),   # MFMA kernel, dense factor tiles
    (21, 6, 64, 1e3, E.PROP_LOWER, None),    # MFMA kernel with 11 pad dimensions (21 -> 32)
    (32, 7, 64, 1e3, E.PROP_DIAG, None),     # diagonal sigmas through the MFMA kernel (as the diagonal matrix they are)
    (19, 5, 128, 1e2, E.PROP_DIAG, 0.5),     # ... with one-dimensional moves: the sampler's default Gaussian flavour
    (27, 5, 192, 1e2, E.PROP_DENSE, None),
    # (populations small enough for one workgroup per CU step in the persistent ladder kernel; these keep the MFMA kernel in PT steps)
    (32, 16, 320, 1e6, E.PROP_LOWER, None),
    (21, 6, 384, 1e3, E.PROP_LOWER, None),
    (32, 8, 320, 1e3, E.PROP_DENSE, None),
    (19, 5, 320, 1e2, E.PROP_DIAG, 0.5),
    # (... and these the general kernel -- whole waves per rung, a lane per chain -- and the fused small-ladder kernel)
    (6, 40, 320, 1e3, E.PROP_LOWER, None),
    (3, 12, 320, 1e2, E.PROP_DENSE, None),
    (5, 9, 300, 1e2, E.PROP_DENSE, 0.3),
    (16, 12, 64, 1e3, E.PROP_DIAG, 0.5),     # the sampler's default Gaussian flavour: diagonal + 1-D moves
    (32, 9, 3, 1e3, E.PROP_DENSE, None),     # lanes kernel (a lane per dimension: fewer than 64 walkers per rung), dense
    (13, 11, 5, 1e2, E.PROP_DIAG, None),     # lanes kernel, 13 -> 16 dimensions, diagonal; 55 chains: a ragged last wave
    (29, 6, 70, 1e2, E.PROP_LOWER, None),    # lanes kernel, walkers not a multiple of 64
    (40, 6, 3, 1e2, E.PROP_DENSE, None),     # 33..64 dimensions: the lanes kernel with one chain per wave
    (64, 5, 64, 1e2, E.PROP_LOWER, None),    # ... and with whole waves per rung: the 64-dimensional MFMA kernel (4 x 4 tiles of 16)
    (64, 6, 128, 1e2, E.PROP_DENSE, None),   # dense factor tiles
    (40, 5, 64, 1e2, E.PROP_DIAG, None),     # 24 pad dimensions, diagonal sigmas as the diagonal matrix they are
    (33, 4, 192, 1e3, E.PROP_LOWER, None),
    (50, 3, 64, 1e2, E.PROP_DENSE, None),
    (50, 4, 70, 1e2, E.PROP_DIAG, 0.3),
    (100, 5, 3, 1e2, E.PROP_LOWER, None),    # 65..128 dimensions: the lanes kernel, two dimensions per lane
    (128, 4, 64, 1e2, E.PROP_DENSE, 0.3),
    (65, 6, 5, 1e2, E.PROP_DIAG, 0.4),
    (128, 4, 64, 1e2, E.PROP_LOWER, None),   # ... and with whole waves per rung, the plain workload: the 128-dimensional MFMA kernel (8 x 8 tiles)
    (128, 3, 128, 1e2, E.PROP_DENSE, None),  # dense factor tiles
    (100, 5, 64, 1e2, E.PROP_LOWER, None),   # 28 pad dimensions
    (65, 3, 192, 1e2, E.PROP_DIAG, None),    # diagonal sigmas as the diagonal matrix they are
    (90, 4, 64, 1e2, E.PROP_DENSE, None),
    (129, 4, 3, 1e2, E.PROP_LOWER, None),    # 129..256 dimensions: four dimensions per lane, the precision matrix read from memory
    (200, 3, 5, 1e2, E.PROP_DENSE, 0.3),
    (256, 4, 2, 1e2, E.PROP_DIAG, 0.4),
    (300, 3, 2, 1e2, E.PROP_LOWER, None),    # 257..512: eight per lane
    (512, 2, 3, 1e2, E.PROP_DENSE, 0.3),
    (600, 3, 2, 1e2, E.PROP_DIAG, None),     # 513..1024: sixteen per lane
    (1024, 2, 2, 1e2, E.PROP_LOWER, 0.3),
    (18, 7, 3, 1e2, E.PROP_LOWER, 0.4),      # lanes kernel, general build: one-dimensional moves
    (32, 5, 2, 1e2, E.PROP_DIAG, 0.5),
    (5, 7, 3, 1e2, E.PROP_DENSE, 0.3),       # padded dimension (5 -> 8), ragged sizes
    (3, 5, 70, 1e2, E.PROP_DIAG, None),      # W not a multiple of 64
    (3, 70, 70, 1e2, E.PROP_DENSE, 0.2),     # ... and past the lanes kernel's population limit: the general kernel, a rung per lane
    (12, 260, 260, 1e3, E.PROP_LOWER, None), # (67600 chains x 16 lanes > 2^20)
    (1, 4, 2, 1e1, E.PROP_DIAG, 1.0),
    (9, 3, 64, 1e2, E.PROP_LOWER, 0.25),
    (3, 1, 64, 1.0, E.PROP_DENSE, None),     # a single rung: no exchange phase at all
    (3, 2, 5, 4.0, E.PROP_LOWER, 0.5),       # two rungs: one pair
    (4, 400, 64, 1e3, E.PROP_DIAG, None),    # ~64 moved rows per ladder and step: both sides of the 64-thread exchange
                                             # block's register capacity (in-block moves / list handed to move_kernel)
]


@pytest.mark.parametrize("D,Nt,W,tmax,kind,odf", SWEEP_CASES)
def test_mh_sweeps_bit_exact(D, Nt, W, tmax, kind, odf):
    pr, eng, lad = PU.make_pair(D, Nt, W, tmax, kind=kind, one_d_frac=odf)
    PU.assert_same_state(eng, lad, "start")
    for k in range(3):
        eng.sweep(7); eng.sync()
        lad.sweep(7)
        PU.assert_same_state(eng, lad, "after %d sweeps" % (7 * (k + 1)))
    acc = eng.naccept.sum() - eng.Nc
    assert 0 < acc < 21 * eng.Nc
    eng.close()


@pytest.mark.parametrize("D,Nt,W,tmax,kind,odf", SWEEP_CASES)
def test_pt_steps_bit_exact(D, Nt, W, tmax, kind, odf):
    """parallel_tempering_chains::step: exchange phase + MH sweep, incl. the candidate log and per-pair counters."""
    sr = 0.1 if Nt >= 16 else 0.35
    pr, eng, lad = PU.make_pair(D, Nt, W, tmax, kind=kind, one_d_frac=odf, swap_rate=sr)
    assert eng.max_swaps == lad.s.contents.maxswaps
    nacc = 0
    for k in range(12):
        eng.step(1); eng.sync()
        lad.pt_step(1)
        PU.assert_same_state(eng, lad, "after PT step %d" % (k + 1))
        pairs, acc = eng.last_swaps()
        assert np.array_equal(pairs, lad.last_pairs) and np.array_equal(acc, lad.last_accept), k
        nacc += int(acc.sum())
    eng.step(20); eng.sync(); lad.pt_step(20)
    PU.assert_same_state(eng, lad, "after 32 PT steps")
    t, a = eng.swap_counts()
    assert np.array_equal(t, lad.swap_count) and np.array_equal(a, lad.swap_accept_count)
    assert nacc > 0 or Nt == 1
    eng.close()


@pytest.mark.parametrize("D", [32, 64, 128])
def test_mfma_kernel_with_a_tight_prior_box(D):
    """The MFMA kernels' box test (ballots over the four lanes of a chain): a uniform prior so narrow that a large share
    of the proposals leaves it on some dimension; accept stream, states and counters must still match the oracle."""
    Nt, W = (6, 320) if D == 32 else (4, 64)   # (320 walkers: too many workgroups for the persistent ladder kernel, the MFMA kernel sweeps)
    rng = np.random.default_rng(12)
    prior = ([1] * D, list(rng.uniform(-0.2, 0.2, D)), list(rng.uniform(0.8, 1.6, D)))   # uniform: centers, halfwidths
    pr, eng, lad = PU.make_pair(D, Nt, W, 1e4, kind=E.PROP_LOWER, prior=prior, swap_rate=0.3)
    assert "mfma%d" % D in eng.sweep_kernel_name
    for k in range(10):
        eng.step(2); eng.sync()
        lad.pt_step(2)
        PU.assert_same_state(eng, lad, "after PT step %d" % (2 * (k + 1)))
    tries = eng.ntries.sum() - eng.Nc
    acc = eng.naccept.sum() - eng.Nc
    assert acc < 0.8 * tries and (acc > 0 or D > 32)   # the box (and the target) reject a good share (at 128 dimensions nearly everything)
    eng.close()


@pytest.mark.parametrize("D,Nt,W,kind,bounds,ev", [(64, 5, 64, E.PROP_LOWER, True, 0.0), (50, 6, 128, E.PROP_DENSE, False, 0.03), (40, 5, 64, E.PROP_DIAG, True, 0.02),
                                                   (128, 4, 64, E.PROP_LOWER, True, 0.0), (100, 5, 64, E.PROP_DENSE, False, 0.03), (70, 4, 128, E.PROP_LOWER, True, 0.02)])
def test_high_dimension_mfma_kernels_with_bounds_and_evolving_ladders(D, Nt, W, kind, bounds, ev):
    """The 64- and 128-dimension MFMA kernels' other builds: open / `limit` boundaries on top of the prior's box (narrow limits, so
    that a good share of the proposals is invalid) and the per-chain temperatures of evolving ladders -- the sampler's default --
    bit for bit the oracle's chains, counters and temperatures."""
    rng = np.random.default_rng(77)
    bnd = prior = x0 = None
    if bounds:
        blo = [1 if d % 3 else 0 for d in range(D)]
        bhi = [1 if d % 2 else 0 for d in range(D)]
        bnd = (blo, bhi, list(rng.uniform(-2.5, -1.5, D)), list(rng.uniform(1.5, 2.5, D)))
        prior = ([1] * D, [0.0] * D, list(rng.uniform(3.0, 6.0, D)))
        x0 = rng.uniform(-1.2, 1.2, size=(Nt * W, D))
    pr, eng, lad = PU.make_pair(D, Nt, W, 1e2, kind=kind, bounds=bnd, prior=prior, swap_rate=0.3, x0=x0)
    if ev:
        eng.set_evolve_temps(ev); lad.evolve_temps(ev)
    assert eng.sweep_kernel_name.startswith("sweep_mfma%d_kernel<" % (64 if D <= 64 else 128))
    assert eng.sweep_kernel_name.endswith("%s, %s>" % ("true" if bounds else "false", "true" if ev else "false")), eng.sweep_kernel_name
    for k in range(4):
        eng.step(3); eng.sync(); lad.pt_step(3)
        PU.assert_same_state(eng, lad, "after %d PT steps" % (3 * (k + 1)))
        if ev:
            assert np.array_equal(eng.invtemps(), lad.betaw)
        if k == 1:
            eng.sweep(2); eng.sync(); lad.sweep(2)
            PU.assert_same_state(eng, lad, "after plain sweeps")
    tries, acc = eng.ntries.sum() - eng.Nc, eng.naccept.sum() - eng.Nc
    assert acc < 0.9 * tries
    t, a = eng.swap_counts()
    assert np.array_equal(t, lad.swap_count) and np.array_equal(a, lad.swap_accept_count)
    eng.close()


def test_add_every_n_history_counters():
    pr, eng, lad = PU.make_pair(4, 9, 64, 1e2, swap_rate=0.4, add_every_n=3)
    eng.step(40); eng.sync(); lad.pt_step(40)
    PU.assert_same_state(eng, lad)
    assert (eng.nhist >= 40).all() and (eng.nhist > 40).any()   # Q6: a rung in two attempts gets two rows
    eng.close()


@pytest.mark.parametrize("D,Nt,W,kind,N,sr,ev", [(32, 8, 64, E.PROP_LOWER, 1, 0.45, 0),   # MFMA kernel, every add saved
                                                 (32, 6, 128, E.PROP_DENSE, 3, 0.45, 0),
                                                 (5, 7, 3, E.PROP_DENSE, 2, 0.45, 0),    # general kernel, ragged sizes
                                                 (20, 7, 3, E.PROP_DENSE, 2, 0.45, 0),   # lanes kernel (a lane per dimension)
                                                 (40, 7, 3, E.PROP_DENSE, 2, 0.45, 0),   # 64-dimension rows
                                                 (33, 300, 64, E.PROP_DIAG, 2, 0.45, 0.01),   # ... in every exchange path, evolving
                                                 (100, 6, 3, E.PROP_LOWER, 2, 0.45, 0),   # 128-dimension rows (two dimensions per lane)
                                                 (150, 5, 3, E.PROP_LOWER, 2, 0.45, 0),   # 256-dimension rows
                                                 (260, 40, 2, E.PROP_DIAG, 2, 0.45, 0.01),  # 512-dimension rows, evolving
                                                 (700, 4, 2, E.PROP_LOWER, 1, 0.45, 0),     # 1024-dimension rows
                                                 (70, 300, 64, E.PROP_DIAG, 2, 0.45, 0.01),
                                                 (16, 9, 5, E.PROP_LOWER, 1, 0.3, 0),
                                                 (16, 12, 64, E.PROP_DIAG, 4, 0.3, 0),
                                                 (4, 900, 64, E.PROP_DIAG, 2, 0.45, 0),  # > 256 moved rows: the slow exchange path
                                                 # evolving ladders (evolve_temps): rows saved in an exchange phase carry the
                                                 # temperature their rung had between two pries of that step
                                                 (32, 8, 64, E.PROP_LOWER, 1, 0.45, 0.03),
                                                 (5, 7, 3, E.PROP_DENSE, 2, 0.45, 0.05),
                                                 (18, 8, 5, E.PROP_LOWER, 2, 0.45, 0.03),   # lanes kernel
                                                 (16, 40, 64, E.PROP_DIAG, 3, 0.3, 0.01),
                                                 (4, 900, 64, E.PROP_DIAG, 2, 0.45, 0.01),
                                                 # ... with a posterior-ordering cut: (rate, lpost_cut)
                                                 (32, 8, 64, E.PROP_LOWER, 1, 0.45, (0.03, 0.5)),
                                                 (5, 7, 3, E.PROP_DENSE, 2, 0.45, (0.05, 0.0)),
                                                 (18, 8, 5, E.PROP_LOWER, 2, 0.45, (0.03, 1.0)),
                                                 (4, 900, 64, E.PROP_DIAG, 2, 0.45, (0.01, 0.0))])
def test_history_rows_match_the_oracle(D, Nt, W, kind, N, sr, ev):
    """What MH_chain::add_state pushes every add_every_N-th call (chain.cc:935-946): state, llike, lprior, Naccept,
    Ntries, last_type -- including the rows a rung holds BETWEEN two exchanges of one step (quirk Q6)."""
    from ptmcmc_amd.problems import GaussianProblem
    steps, cap = (10 if Nt > 100 else 24), 64
    cut = -1.0
    if isinstance(ev, tuple):
        ev, cut = ev
    pr = PU.problem_for(D, Nt, 1e3)
    eng = E.Engine(D, Nt, W, swap_rate=sr, add_every_n=N, history_rungs=Nt, history_capacity=cap, map_rungs=Nt)
    fac = pr.configure(eng, kind)
    eng.init_from_prior()
    x0 = eng.states()
    pb = PU.oracle_problem(pr)
    lad = O.Ladder(pb, pr.beta, W=W, swap_rate=sr, add_every_N=N)
    lad.set_proposals([(PU.KIND_TO_ORACLE[kind], fac[r], 0.0) for r in range(Nt)])
    lad.use_philox(0x5EED0001)
    lad.enable_history(cap)
    lad.set_states(PU.to_oracle_order(x0, Nt, W))
    if ev:
        eng.step(2); lad.pt_step(2)          # (rows saved before the ladders evolve keep the common ladder's temperatures)
        eng.set_evolve_temps(ev, lpost_cut=cut); lad.evolve_temps(ev, cut)
        steps -= 2
    eng.step(steps); eng.sync()
    lad.pt_step(steps)
    if ev:
        steps += 2
        assert np.array_equal(eng.invtemps(), lad.betaw)
    PU.assert_same_state(eng, lad, "after %d steps" % steps)
    he, ho = eng.history(), lad.history()
    nsize = eng.nsize
    assert nsize.max() <= cap and nsize.min() >= 2
    double_adds = int((eng.nhist > steps).sum())
    assert double_adds > 0                                  # some rungs made two add_state calls in one step
    for name in ("x", "llike", "lprior", "naccept", "ntries", "last_type", "invtemp"):
        a = he[name]                                        # [cap][Nt*W] engine order
        b = ho[name]                                        # [N][cap] oracle order
        for s_ in range(int(nsize.max())):
            have = nsize > s_
            got = a[s_ % cap][have]
            want = PU.to_engine_order(b[:, s_], Nt, W)[have]
            assert np.array_equal(got, want), (name, s_, np.argwhere(got != want)[:3].tolist())
    assert np.array_equal(he["row"][0], np.zeros(Nt * W))
    # MAP tracking (chain.cc:931-934), every rung: the in-between rows of twice-exchanged rungs are candidates too
    m = eng.map()
    assert np.array_equal(m["lpost"], PU.to_engine_order(lad.map_lpost, Nt, W))
    assert np.array_equal(m["x"], PU.to_engine_order(lad.map_x, Nt, W))
    if not ev:
        assert np.all(m["lpost"] >= eng.lpost)
    else:   # the rows of the exchange phases were saved at temperatures no ladder holds before or after the step
        hb = he["invtemp"][:int(nsize.min())]
        assert (np.abs(hb - np.repeat(pr.beta, W)) > 0).any()
    eng.close()


@pytest.mark.parametrize("halo", [4, 1])
def test_history_ring_wraps_and_sharded_cold_chain(halo):
    """(1) a ring shorter than the run keeps the newest rows; (2) the cold rung recorded on the first of two shards equals
    the single-engine history (the shard's top rung cannot be recorded: loud) -- also with a halo of ONE rung, where the ladders
    some shard cannot decide are left to the gathered second pass (ptm_exchange_redo), saved rows included."""
    import shard_sim
    from ptmcmc_amd.parallel import shard_bounds
    from ptmcmc_amd.problems import GaussianProblem
    D, Nt, W, sr, steps = 8, 8, 64, 0.45, 30
    pr = GaussianProblem(D, Nt, 1e3)
    ref = E.Engine(D, Nt, W, swap_rate=sr, history_rungs=Nt, history_capacity=64)
    small = E.Engine(D, Nt, W, swap_rate=sr, history_rungs=2, history_capacity=5)
    for e in (ref, small):
        pr.configure(e, E.PROP_LOWER)
    ref.init_from_prior()
    x0 = ref.states()
    small.set_states(x0)
    shards = []
    for g in range(2):
        r0, n = shard_bounds(Nt, 2, g)
        e = E.Engine(D, Nt, W, swap_rate=sr, rung_begin=r0, rung_count=n, history_rungs=(1 if g == 0 else 0), history_capacity=64)
        pr.configure(e, E.PROP_LOWER)
        e.set_states(x0[r0 * W:(r0 + n) * W])
        shards.append(e)
    with pytest.raises(E.PtmError, match="top rung"):
        E.Engine(D, Nt, W, rung_begin=0, rung_count=4, history_rungs=4, history_capacity=8)
    lads = shard_sim.build([_DevShard(e) for e in shards], halo=halo, recover=halo < 4)
    copy = lambda dst, src: dst.copy_from(src.ptr, min(dst.nbytes, src.nbytes))
    ref.step(steps); small.step(steps); shard_sim.step(lads, copy, steps)
    ref.sync(); small.sync()
    for e in shards:
        e.sync()
    assert (lads[0].recovered > 0) == (halo < 4)
    assert np.array_equal(np.concatenate([e.states() for e in shards]), ref.states())
    hr, hs, h0 = ref.history(), small.history(), shards[0].history()
    ns = ref.nsize
    for name in ("x", "llike", "lprior", "naccept", "ntries", "last_type", "row"):
        # the ring of 5 holds saved rows nsize-5 .. nsize-1 of the two recorded rungs
        for c in range(2 * W):
            for s_ in range(int(ns[c]) - 5, int(ns[c])):
                assert np.array_equal(hs[name][s_ % 5][c], hr[name][s_][c]), (name, c, s_)
        for c in range(W):                       # cold rung on shard 0
            n = int(ns[c])
            assert np.array_equal(h0[name][:n, c], hr[name][:n, c]), (name, c)
    for e in shards + [ref, small]:
        e.close()


@pytest.mark.parametrize("D,Nt,W,kind", [(32, 16, 64, E.PROP_LOWER), (6, 9, 5, E.PROP_DENSE)])
def test_checkpoint_and_resume_continue_bit_for_bit(D, Nt, W, kind):
    """ptm_restore: states, llikes, counters, step count and swap counters put into a fresh engine => the same future."""
    from ptmcmc_amd.problems import GaussianProblem
    pr = GaussianProblem(D, Nt, 1e3)
    a = E.Engine(D, Nt, W, swap_rate=0.3, add_every_n=3)
    pr.configure(a, kind)
    a.init_from_prior()
    a.step(17); a.sync()
    ck = a.checkpoint()
    a.step(23); a.sync()
    b = E.Engine(D, Nt, W, swap_rate=0.3, add_every_n=3)
    pr.configure(b, kind)
    b.restore(ck)
    assert b.step_count == 17
    b.step(23); b.sync()
    assert np.array_equal(a.states(), b.states())
    for name in ("llike", "lprior", "ntries", "naccept", "nhist", "nsize", "last_type"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    ta, aa = a.swap_counts(); tb, ab = b.swap_counts()
    assert np.array_equal(ta, tb) and np.array_equal(aa, ab)
    a.close(); b.close()


@pytest.mark.parametrize("D,Nt,W,kind,ev", [(32, 12, 64, E.PROP_LOWER, 0.0), (6, 9, 5, E.PROP_DENSE, 0.02), (32, 10, 64, E.PROP_DIAG, 0.05)])
def test_checkpoint_and_resume_with_history_ring_and_map(D, Nt, W, kind, ev):
    """ptm_restore + ptm_set_history + ptm_set_map (+ ptm_set_invtemps for evolving ladders): the resumed engine's history
    ring (wrapping), MAP and temperatures equal those of the run that never stopped."""
    from ptmcmc_amd.problems import GaussianProblem
    pr = GaussianProblem(D, Nt, 1e3)
    kw = dict(swap_rate=0.4, add_every_n=2, history_rungs=Nt, history_capacity=12, map_rungs=Nt)
    def make():
        e = E.Engine(D, Nt, W, **kw)
        pr.configure(e, kind)
        if ev:
            e.set_evolve_temps(ev)
        return e
    a = make()
    a.init_from_prior()
    a.step(19); a.sync()
    ck = a.checkpoint()
    a.step(21); a.sync()
    b = make()
    b.restore(ck)
    hb0, ha0 = b.history(), ck["history"]
    assert all(np.array_equal(hb0[k], ha0[k]) for k in ha0)
    b.step(21); b.sync()
    assert np.array_equal(a.states(), b.states()) and np.array_equal(a.nhist, b.nhist)
    ha, hb = a.history(), b.history()
    for k in ha:
        assert np.array_equal(ha[k], hb[k]), k
    assert ha["row"].max() >= 12                       # the ring wrapped
    ma, mb = a.map(), b.map()
    for k in ma:
        assert np.array_equal(ma[k], mb[k]), k
    assert np.array_equal(a.invtemps(), b.invtemps())
    a.close(); b.close()


@pytest.mark.parametrize("D,W,kind", [(32, 64, E.PROP_LOWER), (6, 5, E.PROP_DENSE), (16, 64, E.PROP_DIAG)])
def test_one_rung_gets_a_new_proposal_factor_mid_run(D, W, kind):
    """ptm_set_proposal_rung: what user_gaussian_prop::check_update does for one chain (proposal_distribution.cc:406-441);
    the run continues bit-identical to an oracle whose rung got the same new factor."""
    Nt = 7
    pr, eng, lad = PU.make_pair(D, Nt, W, 1e3, kind=kind, swap_rate=0.3)
    fac = pr.proposal_factors() if kind != E.PROP_DIAG else None
    eng.step(6); eng.sync(); lad.pt_step(6)
    rng = np.random.default_rng(3)
    if kind == E.PROP_DIAG:
        new = rng.uniform(0.05, 0.3, D)
    elif kind == E.PROP_LOWER:
        new = np.tril(rng.normal(size=(D, D)) * 0.05) + np.eye(D) * 0.2
    else:
        new = rng.normal(size=(D, D)) * 0.05 + np.eye(D) * 0.2
    eng.set_proposal_rung(3, new)
    props = lad._prop_specs
    props[3] = (PU.KIND_TO_ORACLE[kind], new, props[3][2])
    lad.set_proposals(props)
    eng.step(9); eng.sync(); lad.pt_step(9)
    PU.assert_same_state(eng, lad, "after the factor change")
    eng.close()


@pytest.mark.parametrize("D,Nt,W,kind,K", [(32, 6, 64, E.PROP_DIAG, 6), (24, 5, 128, E.PROP_LOWER, 3), (5, 7, 3, E.PROP_DENSE, 4),
                                          (20, 6, 5, E.PROP_DENSE, 3), (11, 6, 2, E.PROP_DIAG, 6),   # lanes kernel
                                          (16, 6, 64, E.PROP_DIAG, 1)])
def test_scale_mixture_proposals(D, Nt, W, kind, K):
    """A proposal_distribution_set of Gaussian members that are scalar multiples of the rung's factor (the sampler's
    default Gaussian recipe, ptmcmc.cc:117-139): member choice, scales, per-member one-dimensional moves and the
    reference's type code (member + 10 * type, proposal_distribution.cc:117) -- bit-identical to the oracle."""
    pr, eng, lad = PU.make_pair(D, Nt, W, 1e3, kind=kind, swap_rate=0.3)
    rng = np.random.default_rng(8)
    shares = 2.0 ** np.arange(1, K + 1)                     # doubling shares (ptmcmc.cc:121-136)
    cum = np.tile(np.cumsum(shares) / shares.sum(), (Nt, 1))
    cum[:, -1] = 1.0
    scales = np.tile(4.0 ** -np.arange(K)[::-1] * 1.5, (Nt, 1)) * rng.uniform(0.8, 1.2, (Nt, 1))
    odf = np.tile(np.where(np.arange(K) % 2 == 0, 0.5, 0.0), (Nt, 1))
    eng.set_proposal_mixture(cum, scales, odf)
    lad.set_mixture(cum, scales, odf)
    for k in range(5):
        eng.step(5); eng.sync(); lad.pt_step(5)
        PU.assert_same_state(eng, lad, "after %d steps" % (5 * (k + 1)))
    lt = eng.last_type
    seen = set(int(v) for v in np.unique(lt))
    assert seen <= set(range(K)) | set(10 + k for k in range(K)) | {-1}
    if K > 1:
        assert len(seen - {-1}) >= 3 and any(v >= 10 for v in seen)
    eng.set_proposal_mixture(np.zeros((Nt, 0)), np.zeros((Nt, 0)), np.zeros((Nt, 0)))   # back to the plain proposal
    lad.set_proposals(lad._prop_specs)
    eng.step(5); eng.sync(); lad.pt_step(5)
    PU.assert_same_state(eng, lad, "mixture removed")
    eng.close()


@pytest.mark.parametrize("D,Nt,W,kind,sr,rate", [(4, 8, 3, E.PROP_DENSE, 0.3, 0.05), (32, 40, 64, E.PROP_LOWER, 0.45, 0.01),
                                                 (16, 70, 64, E.PROP_DIAG, 0.2, 0.02), (32, 1024, 64, E.PROP_LOWER, 0.1, 0.01),
                                                 (5, 2, 64, E.PROP_DENSE, 0.4, 0.05), (3, 3, 7, E.PROP_DIAG, 0.5, 0.1),
                                                 (20, 9, 3, E.PROP_DENSE, 0.3, 0.05)])   # lanes kernel
def test_evolving_ladders_match_the_oracle(D, Nt, W, kind, sr, rate):
    """parallel_tempering_chains::evolve_temps (chain.hh:302-307): every accepted exchange pries its temperature gap apart
    and renormalises the ladder (pry_temps, chain.cc:1501-1518,1809-1846), so each walker's ladder owns its temperatures.
    States, llikes, counters, exchange decisions AND the temperatures are bit-identical to the oracle (which
    tests/test_oracle_golden.py pins against the real reference with evolve_temps on, traces 5 and 6)."""
    pr, eng, lad = PU.make_pair(D, Nt, W, 1e3, kind=kind, swap_rate=sr)
    b0 = eng.invtemps()
    assert b0.shape == (W, Nt) and np.array_equal(b0, np.tile(pr.beta, (W, 1)))
    eng.set_evolve_temps(rate)
    lad.evolve_temps(rate)
    nsteps = 4 if Nt >= 1024 else 8
    for k in range(3):
        eng.step(nsteps); eng.sync(); lad.pt_step(nsteps)
        PU.assert_same_state(eng, lad, "after %d steps" % (nsteps * (k + 1)))
        be, bo = eng.invtemps(), lad.betaw
        assert np.array_equal(be, bo), (np.abs(be - bo).max(), np.argwhere(be != bo)[:4].tolist())
        assert np.array_equal(eng.lpost, PU.to_engine_order(lad.lpost, Nt, W))
    t, a = eng.swap_counts()
    assert np.array_equal(a, np.asarray(lad._arr(lad.s.contents.swap_accept_count, (W, max(Nt - 1, 1)), np.int64)))
    be = eng.invtemps()
    assert (be[:, 0] == pr.beta[0]).all() and (be[:, -1] == pr.beta[-1]).all()      # the ends stay (chain.cc:1836-1843)
    if Nt > 2:
        assert a.sum() > 0 and (np.abs(be - pr.beta)[:, 1:-1].max(axis=1) > 0).mean() > 0.5   # the ladders moved ...
        assert (np.diff(be, axis=1) < 0).all()                                                 # ... and stay ordered
        if W > 1:
            assert not np.array_equal(be[0], be[1])                                            # each on its own
    eng.close()


@pytest.mark.parametrize("D,Nt,W,kind,sr,rate,cut", [(4, 8, 3, E.PROP_DENSE, 0.3, 0.05, 0.0), (32, 40, 64, E.PROP_LOWER, 0.45, 0.01, 2.0),
                                                     (3, 7, 5, E.PROP_DIAG, 0.45, 0.02, 0.0), (20, 9, 3, E.PROP_DENSE, 0.3, 0.05, 1.0),
                                                     (2, 300, 2, E.PROP_DIAG, 0.3, 0.01, 0.0), (6, 1100, 1, E.PROP_DIAG, 0.1, 0.01, 3.0)])
def test_evolving_ladders_with_a_posterior_ordering_cut_match_the_oracle(D, Nt, W, kind, sr, rate, cut):
    """evolve_temps(rate, lpost_cut >= 0) (chain.hh:302-307; pry_temps chain.cc:1809-1846): every pry ALSO widens each gap whose
    two chains' current log-posteriors are out of order by more than cut * invtemp (:1819-1827) -- any gap of the ladder, so
    the exchange kernel decides the picks one after the other and goes over all the gaps after every accepted exchange.
    States, counters, decisions and temperatures are bit-identical to the oracle, which tests/test_oracle_golden.py pins
    against the real reference with a cut (traces 11, 12); test_history_rows_match_the_oracle has the saved rows' temperatures."""
    pr, eng, lad = PU.make_pair(D, Nt, W, 1e3, kind=kind, swap_rate=sr)
    eng.set_evolve_temps(rate, lpost_cut=cut)
    lad.evolve_temps(rate, cut)
    nsteps = 3 if Nt >= 300 else 8
    for k in range(3):
        eng.step(nsteps); eng.sync(); lad.pt_step(nsteps)
        PU.assert_same_state(eng, lad, "after %d steps" % (nsteps * (k + 1)))
        be, bo = eng.invtemps(), lad.betaw
        assert np.array_equal(be, bo), (np.abs(be - bo).max(), np.argwhere(be != bo)[:4].tolist())
        assert np.array_equal(eng.lpost, PU.to_engine_order(lad.lpost, Nt, W))
    t, a = eng.swap_counts()
    assert np.array_equal(a, np.asarray(lad._arr(lad.s.contents.swap_accept_count, (W, max(Nt - 1, 1)), np.int64))) and a.sum() > 0
    be = eng.invtemps()
    assert (be[:, 0] == pr.beta[0]).all() and (be[:, -1] == pr.beta[-1]).all() and (np.diff(be, axis=1) < 0).all()
    # the cut did something: without it the same run ends on other ladders
    pr2, eng2, lad2 = PU.make_pair(D, Nt, W, 1e3, kind=kind, swap_rate=sr)
    eng2.set_evolve_temps(rate)
    eng2.step(3 * nsteps); eng2.sync()
    assert not np.array_equal(eng2.invtemps(), be)
    eng.close(); eng2.close()


def test_evolving_ladders_checkpoint_resume_and_refusals():
    """checkpoint / resume of an evolving run carries the ladders (ptm_get_invtemps / ptm_set_invtemps); the combinations
    that are not built are refused loudly."""
    from ptmcmc_amd.problems import GaussianProblem
    D, Nt, W = 6, 9, 5
    pr = GaussianProblem(D, Nt, 1e3)
    a = E.Engine(D, Nt, W, swap_rate=0.4)
    pr.configure(a, E.PROP_DENSE)
    a.set_evolve_temps(0.03)
    a.init_from_prior()
    a.step(15); a.sync()
    ck = a.checkpoint()
    assert ck["invtemps"] is not None and not np.array_equal(ck["invtemps"], np.tile(pr.beta, (W, 1)))
    a.step(20); a.sync()
    b = E.Engine(D, Nt, W, swap_rate=0.4)
    pr.configure(b, E.PROP_DENSE)
    b.set_evolve_temps(0.03)
    b.restore(ck)
    b.step(20); b.sync()
    assert np.array_equal(a.states(), b.states()) and np.array_equal(a.invtemps(), b.invtemps())
    assert np.array_equal(a.lpost, b.lpost)
    with pytest.raises(E.PtmError):
        a.set_evolve_temps(0.0)          # no way back (nor in the reference)
    a.close(); b.close()
    d = E.Engine(D, Nt, W, swap_rate=0.4)
    pr.configure(d, E.PROP_DENSE)
    with pytest.raises(E.PtmError):
        d.set_invtemps(np.tile(pr.beta, (W, 1)))   # only once the ladders evolve
    d.close()
    s = E.Engine(D, Nt, W, swap_rate=0.4, rung_begin=0, rung_count=4, history_rungs=2, history_capacity=8)
    s.set_ladder(pr.beta)
    s.set_evolve_temps(0.01)             # a shard of the ladder that records a history evolves too since round 4
    s.close()                            #  (tests/test_gpu_sharding.py::test_evolving_sharded_ladder_records_history_and_map_like_one_engine)


def test_bounds_and_mixed_prior_path_bit_exact():
    """wrap / limit / reflect boundaries and a gaussian+log+uniform+polar+copolar prior on the device path."""
    D, Nt, W = 5, 6, 64
    pi = math.pi
    bounds = ([0, 1, 3, 1, 2], [0, 0, 3, 1, 2], [0, -0.5, -1.5, 0, -pi / 2], [0, 0, 3.5, pi, pi / 2])
    prior = ([2, 5, 1, 3, 4], [0.5, 3.0, 1.0, pi / 2, 0.0], [2.0, 4.0, 2.5, pi / 2, pi / 2])
    rng = np.random.default_rng(1)
    x0 = np.stack([rng.normal(0.5, 1, Nt * W), rng.uniform(0.8, 11, Nt * W), rng.uniform(-1.4, 3.4, Nt * W),
                   rng.uniform(0.1, 3.0, Nt * W), rng.uniform(-1.5, 1.5, Nt * W)], 1)
    pr, eng, lad = PU.make_pair(D, Nt, W, 1e3, kind=E.PROP_DENSE, bounds=bounds, prior=prior, swap_rate=0.3, x0=x0,
                                mean=np.array([0.5, 3.0, 1.0, 1.5, 0.0]))
    PU.assert_same_state(eng, lad, "start")
    eng.step(30); eng.sync(); lad.pt_step(30)
    PU.assert_same_state(eng, lad, "after 30 steps")
    acc = eng.naccept.sum() - eng.Nc
    assert acc > 100
    assert np.isfinite(eng.lprior).all()
    eng.close()


@pytest.mark.parametrize("W", [320, 64, 3])   # (320 walkers: exchange kernel + the MFMA kernel's general build; 64 and 3: the persistent ladder kernel's)
@pytest.mark.parametrize("kind,odf,with_mean,all_uniform", [(E.PROP_DENSE, 0.0, True, False), (E.PROP_LOWER, 0.3, False, False),
                                                            (E.PROP_LOWER, 0.0, False, True), (E.PROP_DENSE, 1.0, True, True)])
def test_mfma_kernel_general_state_space_and_priors(kind, odf, with_mean, all_uniform, W):
    """The MFMA kernel's general build (27 dimensions -> 32, 64 walkers) and the lanes kernel's (3 walkers): wrap / limit /
    reflect boundaries, a gaussian + log + uniform + polar + copolar + flat prior (or an all-uniform box with limit
    bounds), a mean, and one-dimensional moves -- states, llike, lprior and counters bit-identical to the oracle."""
    D, Nt = 27, 5
    pi = math.pi
    rng = np.random.default_rng(21)
    blo, bhi, bmin, bmax = [0] * D, [0] * D, [0.0] * D, [0.0] * D
    types, cen, hw = [1] * D, [0.0] * D, [30.0] * D
    lo_x, hi_x = np.full(D, -2.0), np.full(D, 2.0)
    def setb(d, lo, hi, mn, mx):
        blo[d], bhi[d], bmin[d], bmax[d] = lo, hi, mn, mx
    setb(1, 1, 0, -6.0, 0.0)           # limit below only
    setb(2, 3, 3, -3.0, 3.0)           # wrap
    setb(7, 2, 2, -2.5, 2.5)           # reflect both
    setb(12, 1, 1, -4.0, 4.0)          # limit both
    setb(20, 0, 2, 0.0, 3.5)           # reflect above only
    setb(26, 3, 3, 0.0, 2 * pi)        # wrap on the last (padded-tile) dimension
    lo_x[26], hi_x[26] = 0.5, 5.5
    if not all_uniform:
        types[0], cen[0], hw[0] = 2, 0.3, 1.5                  # gaussian
        types[3], cen[3], hw[3] = 5, 3.0, 4.0                  # log on (0.75, 12)
        lo_x[3], hi_x[3] = 1.0, 9.0
        types[5], cen[5], hw[5] = 3, pi / 2, pi / 2            # polar on (0, pi)
        lo_x[5], hi_x[5] = 0.3, 2.8
        types[9], cen[9], hw[9] = 4, 0.0, pi / 2               # copolar
        lo_x[9], hi_x[9] = -1.3, 1.3
        types[13] = 0                                          # flat
    else:
        hw = [5.0 + 0.1 * d for d in range(D)]
        cen[26], hw[26] = pi, pi
    x0 = rng.uniform(lo_x, hi_x, size=(Nt * W, D))
    mean = rng.normal(size=D) * 0.2 if with_mean else None
    pr, eng, lad = PU.make_pair(D, Nt, W, 50.0, kind=kind, bounds=(blo, bhi, bmin, bmax), prior=(types, cen, hw), swap_rate=0.3,
                                x0=x0, mean=mean, one_d_frac=(odf if odf > 0 else None))
    if W % 64 == 0:
        assert "mfma32_kernel" in eng.sweep_kernel_name and ", 2, " in eng.sweep_kernel_name
    else:
        assert eng.sweep_kernel_name.startswith("sweep_lanes_kernel<32") and eng.sweep_kernel_name.endswith("true>")
    assert eng.step_kernel_name.startswith("decide_kernel + sweep_mfma32" if W == 320 else "ladder_persistent_kernel<32"), eng.step_kernel_name
    PU.assert_same_state(eng, lad, "start")
    for k in range(6):
        eng.step(5); eng.sync(); lad.pt_step(5)
        PU.assert_same_state(eng, lad, "after %d steps" % (5 * (k + 1)))
    acc = eng.naccept.sum() - eng.Nc
    assert acc > (100 if W >= 64 else 10) and np.isfinite(eng.lprior).all()
    if odf > 0:
        assert (eng.last_type == 1).any()
    eng.close()


@pytest.mark.parametrize("kind,odf,with_mean", [(E.PROP_LOWER, 0.0, False), (E.PROP_DENSE, 0.4, True)])
def test_mfma_kernel_limit_bounds_with_uniform_prior(kind, odf, with_mean):
    """The usual real-world state space -- uniform priors, `limit` / open boundaries -- has its own build of the MFMA
    kernel (enforcing is a box test); narrow limits so that a good share of the proposals is invalid."""
    D, Nt, W = 30, 6, 320   # (more workgroups than the persistent ladder kernel holds: the MFMA kernel's box-bounds build sweeps)
    rng = np.random.default_rng(33)
    blo = [1 if d % 3 else 0 for d in range(D)]
    bhi = [1 if d % 2 else 0 for d in range(D)]
    bmin = list(rng.uniform(-2.5, -1.5, D)); bmax = list(rng.uniform(1.5, 2.5, D))
    prior = ([1] * D, [0.0] * D, list(rng.uniform(3.0, 6.0, D)))
    x0 = rng.uniform(-1.4, 1.4, size=(Nt * W, D))
    mean = rng.normal(size=D) * 0.1 if with_mean else None
    pr, eng, lad = PU.make_pair(D, Nt, W, 1e3, kind=kind, bounds=(blo, bhi, bmin, bmax), prior=prior, swap_rate=0.3, x0=x0, mean=mean,
                                one_d_frac=(odf if odf > 0 else None))
    assert eng.sweep_kernel_name.endswith(", 1, false, false>")
    for k in range(5):
        eng.step(4); eng.sync(); lad.pt_step(4)
        PU.assert_same_state(eng, lad, "after %d steps" % (4 * (k + 1)))
    tries, acc = eng.ntries.sum() - eng.Nc, eng.naccept.sum() - eng.Nc
    assert 0 < acc < 0.8 * tries
    eng.close()


def test_origin_outside_limit_bound_rejects_everything():
    """quirk Q9 (states.cc:183-192,205-214), pinned against the reference by golden trace 3."""
    D, Nt, W = 2, 4, 64
    bounds = ([1, 0], [0, 0], [0.75, 0], [0, 0])
    prior = ([1, 1], [3.0, 0.0], [2.0, 50.0])
    rng = np.random.default_rng(2)
    x0 = np.stack([rng.uniform(1.1, 4.9, Nt * W), rng.normal(size=Nt * W)], 1)
    pr, eng, lad = PU.make_pair(D, Nt, W, 10.0, kind=E.PROP_DIAG, bounds=bounds, prior=prior, x0=x0)
    eng.sweep(10); eng.sync(); lad.sweep(10)
    PU.assert_same_state(eng, lad)
    assert (eng.naccept == 1).all() and (eng.ntries == 11).all()
    eng.close()


def test_streams_do_not_depend_on_batch_composition():
    """Size-independent property: a walker's chain depends only on (seed, walker, rung, step) -- batching it with
    64 or 192 walkers, or launching with the per-lane or the wave-uniform kernel, gives the same chain."""
    D, Nt = 8, 16
    pr, e1, lad = PU.make_pair(D, Nt, 64, 1e3)
    x0 = e1.states().reshape(Nt, 64, D)
    e2 = E.Engine(D, Nt, 192)
    pr.configure(e2, E.PROP_LOWER)
    x2 = np.concatenate([x0, x0[:, ::-1], x0], axis=1)          # walkers 0..63 identical, the rest arbitrary
    e2.set_states(x2.reshape(-1, D))
    e3 = E.Engine(D, Nt, 3)                                      # non-uniform kernel variant
    pr.configure(e3, E.PROP_LOWER)
    e3.set_states(x0[:, :3].reshape(-1, D))
    for e in (e1, e2, e3):
        e.step(25); e.sync()
    a = e1.states().reshape(Nt, 64, D)
    assert np.array_equal(a, e2.states().reshape(Nt, 192, D)[:, :64])
    assert np.array_equal(a[:, :3], e3.states().reshape(Nt, 3, D))
    for e in (e1, e2, e3):
        e.close()


def test_full_size_baseline_config_through_size_independent_properties():
    """BASELINE configs[1] at FULL size (D=32, 1024 rungs x 4096 walkers = 4.19 M chains, the benchmark workload), checked
    through properties that do not need a 4-million-chain oracle run:
      (1) a walker's chain does not depend on the batch: walkers 0..63 of the full run == a 64-walker run, bit for bit;
      (2) that 64-walker run (65 536 chains) == the CPU oracle, bit for bit;
      (3) conservation: every chain made exactly one add_state-or-MH per step and exchange (Ntries + exchanged adds ==
          steps + 1 + adds), accepts <= tries, every swap attempt touched two rungs, all values finite."""
    from ptmcmc_amd.problems import GaussianProblem
    D, Nt, Wfull, Wsmall, steps, sr = 32, 1024, 4096, 64, 6, 0.1
    pr = GaussianProblem(D, Nt, 1e9)
    big = E.Engine(D, Nt, Wfull, swap_rate=sr)
    fac = pr.configure(big, E.PROP_LOWER)
    assert "mfma" in big.sweep_kernel_name
    big.init_from_prior()
    x0 = big.states().reshape(Nt, Wfull, D)[:, :Wsmall].reshape(-1, D).copy()
    small = E.Engine(D, Nt, Wsmall, swap_rate=sr)
    pr.configure(small, E.PROP_LOWER)
    small.set_states(x0)
    big.step(steps); small.step(steps)
    big.sync(); small.sync()
    xb = big.states().reshape(Nt, Wfull, D)
    assert np.array_equal(xb[:, :Wsmall].reshape(-1, D), small.states())                       # (1)
    for name in ("llike", "ntries", "naccept", "nhist"):
        assert np.array_equal(getattr(big, name).reshape(Nt, Wfull)[:, :Wsmall].ravel(), getattr(small, name)), name
    pb = PU.oracle_problem(pr)                                                                  # (2)
    lad = O.Ladder(pb, pr.beta, W=Wsmall, swap_rate=sr)
    lad.set_proposals([(O.PROP_DENSE, fac[r], 0.0) for r in range(Nt)])
    lad.use_philox(0x5EED0001)
    lad.set_states(PU.to_oracle_order(x0, Nt, Wsmall))
    lad.pt_step(steps, nthreads=8)
    PU.assert_same_state(small, lad, "full-length ladder, 64 walkers, %d steps" % steps)
    nt, na, nh = big.ntries, big.naccept, big.nhist                                             # (3)
    t, a = big.swap_counts()
    assert np.all(np.isfinite(xb)) and np.all(np.isfinite(big.llike))
    assert np.all(na <= nt) and np.all(nt >= 1)
    assert int((nt - 1).sum()) + 2 * int(t.sum()) == int(nh.sum())      # MH adds + two adds per exchange attempt
    assert np.all(nh >= steps) and np.all(nh <= 2 * steps)
    assert 0 < a.sum() <= t.sum()
    for e in (big, small):
        e.close()


# ---------------------------------------------------------------------------------------------------------------
# ladder sharding: G engines (one per "GPU") must reproduce the single-engine chains bit for bit
# ---------------------------------------------------------------------------------------------------------------
class _DevShard:
    """EngineShard with raw device buffers instead of torch tensors (same five methods)."""

    def __init__(self, eng):
        self.e = eng
        self.W, self.nloc, self.r0, self.Nt = eng.W, eng.nloc, eng.r0, eng.Nt
        self.row_doubles = eng.exchange_buffer_doubles

    def alloc(self, n):
        return E.DeviceBuffer(n * 8)

    def copy_llike(self, first, n, dst):
        self.e.copy_llike(first, n, dst.ptr)

    def exchange_decide(self, lb, la, halo, su, sd):
        p = lambda b: None if b is None else b.ptr
        self.e.exchange_decide(p(lb), p(la), halo, p(su), p(sd))

    def finish_and_sweep(self, rb, ra):
        p = lambda b: None if b is None else b.ptr
        self.e.exchange_finish_and_sweep(p(rb), p(ra))

    def install(self, rb, ra):
        p = lambda b: None if b is None else b.ptr
        self.e.exchange_install(p(rb), p(ra))

    def sweep_rungs(self, first, n, closes_step):
        self.e.sweep_rungs(first, n, closes_step)

    can_overlap = True

    def sync(self):
        self.e.sync()

    # evolving ladders: the gathered form (ShardedLadder.step_gathered)
    @property
    def gathered(self):
        return bool(self.e._evolving)

    @property
    def needs_lprior(self):
        return self.e._evolve_cut >= 0

    def copy_lprior(self, first, n, dst):
        self.e.copy_lprior(first, n, dst.ptr)

    def exchange_decide_gathered(self, ll_all, lp_all, su, sd):
        p = lambda b: None if b is None else b.ptr
        self.e.exchange_decide_gathered(p(ll_all), p(lp_all), p(su), p(sd))

    def set_shard_map(self, sizes, halo):
        self.e.set_shard_map(sizes, halo)

    def exchange_redo_count(self):
        return self.e.exchange_redo_count()

    def exchange_redo(self, ll_all, lp_all, su, sd):
        p = lambda b: None if b is None else b.ptr
        self.e.exchange_redo(p(ll_all), p(lp_all), p(su), p(sd))

    @staticmethod
    def sub(buf, off, n):
        return buf.slice(off * 8, n * 8)

    def dcopy(self, dst, src):
        self.e.sync()
        dst.copy_from(src.ptr, min(dst.nbytes, src.nbytes))


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("D,Nt,W,G,halo,sr", [(32, 16, 64, 2, 4, 0.3), (8, 12, 64, 3, 4, 0.45), (5, 9, 3, 4, 3, 0.45),
                                              (32, 64, 64, 8, 4, 0.1), (4, 6, 64, 6, 2, 0.5),
                                              (40, 12, 5, 3, 3, 0.45)])   # 64-dimension rows through the messages
def test_sharded_engines_match_single_engine(D, Nt, W, G, halo, sr, overlap):
    import shard_sim
    from ptmcmc_amd.parallel import shard_bounds
    from ptmcmc_amd.problems import GaussianProblem
    pr = GaussianProblem(D, Nt, 1e3)
    ref = E.Engine(D, Nt, W, swap_rate=sr)
    pr.configure(ref, E.PROP_LOWER)
    ref.init_from_prior()
    x0 = ref.states()
    shards = []
    for g in range(G):
        r0, n = shard_bounds(Nt, G, g)
        e = E.Engine(D, Nt, W, swap_rate=sr, rung_begin=r0, rung_count=n)
        pr.configure(e, E.PROP_LOWER)
        e.set_states(x0[r0 * W:(r0 + n) * W])
        shards.append(e)
    lads = shard_sim.build([_DevShard(e) for e in shards], halo=halo)
    copy = lambda dst, src: dst.copy_from(src.ptr)
    nsteps, far = 40, False
    for k in range(nsteps):
        ref.step(1)
        try:
            (shard_sim.step_overlapped if overlap else shard_sim.step)(lads, copy, 1)
        except E.PtmError as ex:         # a chain longer than the halo / a row through a whole shard: must be LOUD
            assert "halo" in str(ex) or "crossed" in str(ex)   # (never "overflowed": capacity = W at these sizes)
            far = True
            break
        xs = np.concatenate([e.states() for e in shards])
        assert np.array_equal(xs, ref.states()), "states differ after step %d" % (k + 1)
    if not far:
        for name in ("llike", "lprior", "ntries", "naccept", "nhist", "nsize", "last_type"):
            assert np.array_equal(np.concatenate([getattr(e, name) for e in shards]), getattr(ref, name)), name
        t = sum(e.swap_counts()[0] for e in shards); a = sum(e.swap_counts()[1] for e in shards)
        rt, ra = ref.swap_counts()
        assert np.array_equal(t, rt) and np.array_equal(a, ra)
        assert a.sum() > 0
    else:
        assert min(e.nloc for e in shards) <= 2   # only tiny shards may trip the halo / far-move guard in 40 steps
    for e in shards + [ref]:
        e.close()


@pytest.mark.parametrize("D,Nt,W,G,sr,rate,cut", [(6, 12, 3, 2, 0.4, 0.05, -1.0), (32, 24, 64, 3, 0.3, 0.01, -1.0), (5, 17, 4, 4, 0.45, 0.03, 0.0),
                                                  (16, 40, 2, 5, 0.2, 0.01, 1.5), (40, 9, 5, 3, 0.45, 0.02, -1.0)])
def test_evolving_ladders_on_rung_shards_match_the_single_engine(D, Nt, W, G, sr, rate, cut):
    """evolve_temps on a ladder sharded by rungs: every accepted exchange renormalises all gaps and every later trial of the step
    sees it, so every shard replays the whole ladder's trials -- from the whole ladder's llikes (and lpriors, with a
    posterior-ordering cut), gathered each step as the reference's MPI ranks gather them (gather_llikes / gather_lposts,
    chain.cc:1433-1435,1950-1972; ptm_exchange_decide_gathered).  G shards, driven in lockstep with the all-gather done by plain
    copies, must walk the single engine's chains bit for bit, temperatures included."""
    import shard_sim
    from ptmcmc_amd.parallel import shard_bounds
    from ptmcmc_amd.problems import GaussianProblem
    pr = GaussianProblem(D, Nt, 1e3)
    ref = E.Engine(D, Nt, W, swap_rate=sr)
    pr.configure(ref, E.PROP_LOWER)
    ref.set_evolve_temps(rate, cut)
    ref.init_from_prior()
    x0 = ref.states()
    shards = []
    for g in range(G):
        r0, n = shard_bounds(Nt, G, g)
        e = E.Engine(D, Nt, W, swap_rate=sr, rung_begin=r0, rung_count=n)
        pr.configure(e, E.PROP_LOWER)
        e.set_evolve_temps(rate, cut)
        e.set_states(x0[r0 * W:(r0 + n) * W])
        with pytest.raises(E.PtmError, match="gathered"):      # the halo form cannot serve an evolving ladder
            e.exchange_decide(None, None, 1, None, None)
        shards.append(e)
    lads = shard_sim.build([_DevShard(e) for e in shards], halo=4)
    assert all(l.gathered for l in lads)
    copy = lambda dst, src: dst.copy_from(src.ptr, min(dst.nbytes, src.nbytes))
    for k in range(40):
        ref.step(1)
        shard_sim.step(lads, copy, 1)
        xs = np.concatenate([e.states() for e in shards])
        assert np.array_equal(xs, ref.states()), "states differ after step %d" % (k + 1)
        for e in shards:                                        # every shard keeps the whole ladders' temperatures
            assert np.array_equal(e.invtemps(), ref.invtemps()), "temperatures differ after step %d" % (k + 1)
    assert not np.array_equal(ref.invtemps()[0], pr.beta)       # ... and they did evolve
    for name in ("llike", "lprior", "lpost", "ntries", "naccept", "nhist", "nsize", "last_type"):
        assert np.array_equal(np.concatenate([getattr(e, name) for e in shards]), getattr(ref, name)), name
    t = sum(e.swap_counts()[0] for e in shards); a = sum(e.swap_counts()[1] for e in shards)
    rt, ra = ref.swap_counts()
    assert np.array_equal(t, rt) and np.array_equal(a, ra) and a.sum() > 0
    for e in shards + [ref]:
        e.close()


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("D,Nt,W,G,halo,sr", [(4, 12, 64, 3, 1, 0.5), (32, 16, 64, 2, 1, 0.45), (5, 12, 3, 4, 1, 0.5), (8, 24, 70, 3, 2, 0.45)])
def test_runs_longer_than_the_halo_are_recovered_not_fatal(D, Nt, W, G, halo, sr, overlap):
    """With a halo of ONE rung and many exchange attempts per step, shards meet runs of surviving picks they cannot decide every
    few steps.  Without recovery that is PTM_ERR_FAR_MOVE (loud, fatal: test_sharded_engines_match_single_engine).  With the shard
    map (ptm_set_shard_map; ShardedLadder(recover=True)) every shard finds the same ladders -- the condition is a property of the
    replayed candidate draws -- leaves them alone, and they are decided from the gathered llikes by a second pass
    (ptm_exchange_redo): the sharded ladder walks the single engine's chains bit for bit, and the path was provably taken."""
    import shard_sim
    from ptmcmc_amd.parallel import shard_bounds
    from ptmcmc_amd.problems import GaussianProblem
    pr = GaussianProblem(D, Nt, 1e3)
    ref = E.Engine(D, Nt, W, swap_rate=sr)
    pr.configure(ref, E.PROP_LOWER)
    ref.init_from_prior()
    x0 = ref.states()
    shards = []
    for g in range(G):
        r0, n = shard_bounds(Nt, G, g)
        e = E.Engine(D, Nt, W, swap_rate=sr, rung_begin=r0, rung_count=n)
        pr.configure(e, E.PROP_LOWER)
        e.set_states(x0[r0 * W:(r0 + n) * W])
        shards.append(e)
    lads = shard_sim.build([_DevShard(e) for e in shards], halo=halo, recover=True)
    assert all(l.recover for l in lads)
    copy = lambda dst, src: dst.copy_from(src.ptr, min(dst.nbytes, src.nbytes))
    for k in range(40):
        ref.step(1)
        (shard_sim.step_overlapped if overlap else shard_sim.step)(lads, copy, 1)
        xs = np.concatenate([e.states() for e in shards])
        assert np.array_equal(xs, ref.states()), "states differ after step %d" % (k + 1)
        for e in shards:
            e.sync()                     # no deferred error: nothing was decided blindly
    assert lads[0].recovered > 0 and len({l.recovered for l in lads}) == 1   # the path was taken, by all shards alike
    for name in ("llike", "lprior", "ntries", "naccept", "nhist", "nsize", "last_type"):
        assert np.array_equal(np.concatenate([getattr(e, name) for e in shards]), getattr(ref, name)), name
    t = sum(e.swap_counts()[0] for e in shards); a = sum(e.swap_counts()[1] for e in shards)
    rt, ra = ref.swap_counts()
    assert np.array_equal(t, rt) and np.array_equal(a, ra)
    pairs = [e.last_swaps() for e in shards]                # the last step's candidate log: a ladder of the halo pass has every pick
    rp, racc = ref.last_swaps()                             # logged by its one owner, a recovered ladder by every shard (the counters
    for pk, ac in pairs:                                    # above take a pick from the shard that owns its lower rung either way)
        m = pk >= 0
        assert np.array_equal(pk[m], rp[m]) and np.array_equal(ac[m], racc[m])
    assert np.array_equal(np.stack([p[0] >= 0 for p in pairs]).any(axis=0), rp >= 0)
    for e in shards + [ref]:
        e.close()


def test_boundary_message_overflow_is_loud():
    """A boundary message with fewer row slots than rows crossing must raise at the next sync, never drop rows silently;
    with the automatic capacity the same run is clean."""
    import shard_sim
    from ptmcmc_amd.parallel import shard_bounds
    from ptmcmc_amd.problems import GaussianProblem
    D, Nt, W, G, sr = 8, 8, 256, 2, 0.5
    pr = GaussianProblem(D, Nt, 1e3)
    for cap, loud in ((1, True), (0, False)):
        shards = []
        for g in range(G):
            r0, n = shard_bounds(Nt, G, g)
            e = E.Engine(D, Nt, W, swap_rate=sr, rung_begin=r0, rung_count=n, exchange_row_capacity=cap)
            pr.configure(e, E.PROP_LOWER)
            e.init_from_prior()
            shards.append(e)
        assert shards[0].exchange_row_capacity == (1 if cap else W)    # W*0.5 + 8 sigma + 64 > W here
        assert shards[0].exchange_buffer_doubles == 2 + shards[0].exchange_row_capacity * (D + 4)
        lads = shard_sim.build([_DevShard(e) for e in shards], halo=4)
        copy = lambda dst, src: dst.copy_from(src.ptr)
        if loud:
            with pytest.raises(E.PtmError, match="overflowed"):
                shard_sim.step(lads, copy, 5)
        else:
            shard_sim.step(lads, copy, 5)
        for e in shards:
            e.close()


def test_default_row_capacity_tracks_swap_rate():
    e = E.Engine(4, 8, 100000, swap_rate=0.1, rung_begin=0, rung_count=4)
    want = int(100000 * 0.1 + 8 * np.sqrt(100000 * 0.1) + 64)
    assert e.exchange_row_capacity == want
    e.close()


@pytest.mark.parametrize("Nt,W", [(16, 4), (128, 1)])
@pytest.mark.parametrize("ev", [0.0, 0.02])
def test_host_callback_likelihood_C5_exampleLISA(ev, Nt, W):
    """BASELINE configs[4] (ev > 0: with the ladder evolving, the sampler's default; Nt = 128, W = 1: the configuration's own
    shape, 128 temperatures): a user plug-in likelihood (the reference's toy LISA likelihood, exampleLISA.cc:59-72,130-142)
    through the C-ABI callback, mixed uniform/polar/copolar prior with wrap + limit boundaries (exampleLISA.cc:528-593).
    The propose kernel, the host call and the accept kernel must reproduce the oracle's chain bit for bit."""
    import lisa_toy
    D = 6
    beta = E.geometric_ladder(Nt, 1e9)
    rng = np.random.default_rng(4)
    lo = np.array(lisa_toy.CENTERS) - np.array(lisa_toy.SCALES)
    hi = np.array(lisa_toy.CENTERS) + np.array(lisa_toy.SCALES)
    x0 = rng.uniform(lo + 0.05, hi - 0.05, size=(Nt * W, D))
    sig = np.array(lisa_toy.SCALES) / 20.0
    eng = E.Engine(D, Nt, W, swap_rate=0.3)
    eng.set_bounds(lisa_toy.BLO, lisa_toy.BHI, lisa_toy.BMIN, lisa_toy.BMAX)
    eng.set_prior(lisa_toy.TYPES, lisa_toy.CENTERS, lisa_toy.SCALES)
    eng.set_target_callback(lisa_toy.loglike)
    eng.set_ladder(beta)
    eng.set_proposals(E.PROP_DIAG, np.tile(sig, (Nt, 1)) / np.sqrt(beta)[:, None].clip(1e-3), np.full(Nt, 0.5))
    eng.set_states(x0)
    pb = O.Problem(D)
    pb.set_bounds(lisa_toy.BLO, lisa_toy.BHI, lisa_toy.BMIN, lisa_toy.BMAX)
    pb.set_prior(lisa_toy.TYPES, lisa_toy.CENTERS, lisa_toy.SCALES)
    pb.set_user(lisa_toy.loglike)
    lad = O.Ladder(pb, beta, W=W, swap_rate=0.3)
    fac = np.tile(sig, (Nt, 1)) / np.sqrt(beta)[:, None].clip(1e-3)
    lad.set_proposals([(O.PROP_DIAG, fac[r], 0.5) for r in range(Nt)])
    lad.use_philox(0x5EED0001)
    lad.set_states(PU.to_oracle_order(x0, Nt, W))
    if ev:
        eng.set_evolve_temps(ev); lad.evolve_temps(ev)
    PU.assert_same_state(eng, lad, "start")
    for k in range(6):
        eng.step(5); eng.sync(); lad.pt_step(5)
        PU.assert_same_state(eng, lad, "after %d steps" % (5 * (k + 1)))
        assert np.array_equal(eng.invtemps(), lad.betaw)
    assert eng.naccept.sum() - eng.Nc > 50
    assert (eng.last_type >= 0).any()
    eng.close()


@pytest.mark.parametrize("D,Nt,W,ev", [(12, 8, 3, 0.0), (20, 6, 5, 0.03), (40, 5, 2, 0.0), (70, 4, 3, 0.02), (140, 4, 2, 0.0)])
def test_host_callback_likelihood_through_the_lanes_kernel(D, Nt, W, ev):
    """A plug-in likelihood on a small population with more than 8 dimensions: the propose and accept passes of the lanes
    kernel (a lane per dimension) around the host call -- bit for bit the oracle's chain, with a gaussian + uniform prior,
    a wrapped dimension, one-dimensional moves and (ev > 0) an evolving ladder."""
    import math
    rng = np.random.default_rng(7)
    sc = rng.uniform(0.5, 2.0, D)
    def loglike(x):
        x = np.asarray(x, dtype=np.float64)
        return float(-0.5 * np.sum((x * sc) ** 2) - 0.1 * math.cos(3.0 * x[0]))
    beta = E.geometric_ladder(Nt, 1e3)
    blo, bhi, bmin, bmax = [0] * D, [0] * D, [0.0] * D, [0.0] * D
    blo[1], bhi[1], bmin[1], bmax[1] = 3, 3, -2.0, 2.0                     # wrap
    types, cen, hw = [1] * D, [0.0] * D, [4.0] * D
    types[2], cen[2], hw[2] = 2, 0.2, 1.5                                   # gaussian
    x0 = rng.uniform(-1.5, 1.5, size=(Nt * W, D))
    sig = np.full(D, 0.4)
    fac = np.tile(sig, (Nt, 1)) / np.sqrt(beta)[:, None].clip(1e-2)
    eng = E.Engine(D, Nt, W, swap_rate=0.3)
    eng.set_bounds(blo, bhi, bmin, bmax)
    eng.set_prior(types, cen, hw)
    eng.set_target_callback(loglike)
    eng.set_ladder(beta)
    eng.set_proposals(E.PROP_DIAG, fac, np.full(Nt, 0.3))
    eng.set_states(x0)
    assert eng.sweep_kernel_name.startswith("sweep_lanes_kernel<%d" % (16 if D <= 16 else 32 if D <= 32 else 64 if D <= 64 else 128 if D <= 128 else 256))
    pb = O.Problem(D)
    pb.set_bounds(blo, bhi, bmin, bmax)
    pb.set_prior(types, cen, hw)
    pb.set_user(loglike)
    lad = O.Ladder(pb, beta, W=W, swap_rate=0.3)
    lad.set_proposals([(O.PROP_DIAG, fac[r], 0.3) for r in range(Nt)])
    lad.use_philox(0x5EED0001)
    lad.set_states(PU.to_oracle_order(x0, Nt, W))
    if ev:
        eng.set_evolve_temps(ev); lad.evolve_temps(ev)
    PU.assert_same_state(eng, lad, "start")
    for k in range(5):
        eng.step(6); eng.sync(); lad.pt_step(6)
        PU.assert_same_state(eng, lad, "after %d steps" % (6 * (k + 1)))
    assert eng.naccept.sum() - eng.Nc > 10 and (eng.last_type == 1).any()
    eng.close()


@pytest.mark.parametrize("D,Nt,W,ev", [(3, 6, 4, 0.0), (12, 8, 3, 0.02), (20, 5, 2, 0.0)])
def test_host_evaluated_prior_matches_the_oracle(D, Nt, W, ev):
    """A prior ptm_set_prior cannot describe (here: a correlated Gaussian times a hard disc in the first two dimensions -- what
    an independent_dist_product / transformed_dist / user subclass of probability_function may be) through
    ptm_set_prior_callback: propose kernel -> host prior -> prior gate (chain.cc:980) -> host likelihood -> accept kernel.
    Bit for bit the chain of the oracle handed the same function as its prior; wrap + limit boundaries stay on the device
    (an invalid state is never shown to the prior); the exchange phase carries the log-priors with the rows."""
    import math
    rng = np.random.default_rng(11)
    sc = rng.uniform(0.5, 2.0, D)
    def loglike(x):
        x = np.asarray(x, dtype=np.float64)
        return float(-0.5 * np.sum((x * sc) ** 2) - 0.1 * math.cos(3.0 * x[0]))
    calls = {"n": 0, "bad": 0}
    def logprior(x):
        x = np.asarray(x, dtype=np.float64)
        calls["n"] += 1
        if abs(x[1]) > 2.0 + 1e-12 or x[2] < -3.0 - 1e-12 or x[2] > 3.0 + 1e-12:
            calls["bad"] += 1               # never asked about a state outside the boundaries
        if x[0] * x[0] + x[1] * x[1] > 6.0:
            return -math.inf                # hard support
        return float(-0.5 * (x[0] - 0.3 * x[1]) ** 2 - 0.05 * np.sum(x[2:] ** 2) - 1.234)
    beta = E.geometric_ladder(Nt, 1e3)
    blo, bhi, bmin, bmax = [0] * D, [0] * D, [0.0] * D, [0.0] * D
    blo[1], bhi[1], bmin[1], bmax[1] = 3, 3, -2.0, 2.0                     # wrap
    blo[2], bhi[2], bmin[2], bmax[2] = 1, 1, -3.0, 3.0                     # limit: proposals beyond it are invalid
    x0 = rng.uniform(-1.2, 1.2, size=(Nt * W, D))
    fac = np.tile(np.full(D, 0.6), (Nt, 1)) / np.sqrt(beta)[:, None].clip(1e-2)
    eng = E.Engine(D, Nt, W, swap_rate=0.3)
    eng.set_bounds(blo, bhi, bmin, bmax)
    eng.set_target_callback(loglike)
    eng.set_prior_callback(logprior)
    eng.set_ladder(beta)
    eng.set_proposals(E.PROP_DIAG, fac, np.full(Nt, 0.2))
    with pytest.raises(E.PtmError):
        eng.init_from_prior()               # a host prior is drawn from on the host
    eng.set_states(x0)
    pb = O.Problem(D)
    pb.set_bounds(blo, bhi, bmin, bmax)
    pb.set_user(loglike)
    pb.set_user_prior(logprior)
    lad = O.Ladder(pb, beta, W=W, swap_rate=0.3)
    lad.set_proposals([(O.PROP_DIAG, fac[r], 0.2) for r in range(Nt)])
    lad.use_philox(0x5EED0001)
    lad.set_states(PU.to_oracle_order(x0, Nt, W))
    if ev:
        eng.set_evolve_temps(ev); lad.evolve_temps(ev)
    PU.assert_same_state(eng, lad, "start")
    for k in range(6):
        eng.step(5); eng.sync(); lad.pt_step(5)
        PU.assert_same_state(eng, lad, "after %d steps" % (5 * (k + 1)))
    assert calls["bad"] == 0 and calls["n"] > Nt * W
    lp = eng.lprior
    assert np.isfinite(lp).all() and len(np.unique(lp)) > Nt * W // 2      # the host's values, not a constant
    assert eng.naccept.sum() - eng.Nc > 10
    eng.close()


def test_exchange_overflow_path_many_moved_rows():
    """More than 256 rows of one ladder move in one step (high swap rate on a long ladder): the exchange kernel's
    in-kernel cycle walk must give the same chain as the register gather/scatter kernel does for smaller counts."""
    pr, eng, lad = PU.make_pair(4, 900, 64, 1e3, kind=E.PROP_DIAG, swap_rate=0.45)
    moved_max = 0
    for k in range(6):
        eng.step(1); eng.sync(); lad.pt_step(1)
        PU.assert_same_state(eng, lad, "after step %d" % (k + 1))
        moved_max = max(moved_max, int(2 * eng.last_swaps()[1].sum(axis=1).max()))
    assert moved_max > 256, moved_max
    eng.close()


@pytest.mark.parametrize("kind", [E.PROP_LOWER, E.PROP_DENSE])
def test_compacted_sweep_of_big_populations_matches_the_oracle(kind):
    """From 1024 walkers per rung on, the lean MFMA build visits only the chains that make a move after an exchange phase
    (partition_kernel + the compacted tile walk), and counts every chain's one add_state of the step lazily.  Same chains as
    the oracle bit for bit -- with plain sweeps (every chain visited in place, the pending counts flushed first) in between,
    counters read back mid-run, and a checkpoint / resume across the lazy count."""
    D, Nt, W, sr = 32, 12, 1024, 0.3
    pr, eng, lad = PU.make_pair(D, Nt, W, 1e3, kind=kind, swap_rate=sr)
    assert eng.sweep_kernel_name.endswith(", 0, false, true>")
    eng.step(3); eng.sync(); lad.pt_step(3)
    PU.assert_same_state(eng, lad, "after 3 compacted steps")
    eng.sweep(2); eng.sync(); lad.sweep(2)                  # no exchange phase: nothing to compact
    PU.assert_same_state(eng, lad, "after plain sweeps")
    eng.step(4); lad.pt_step(4)
    ck = eng.checkpoint()                                    # (reads nhist: flushes the lazy count)
    PU.assert_same_state(eng, lad, "after 7 steps")
    assert int((eng.nhist > 9).sum()) > 0                    # rungs exchanged twice in one step got their extra add
    e2 = E.Engine(D, Nt, W, swap_rate=sr)
    pr.configure(e2, kind)
    e2.restore(ck)
    for e in (eng, e2):
        e.step(5); e.sync()
    lad.pt_step(5)
    PU.assert_same_state(eng, lad, "after 12 steps")
    PU.assert_same_state(e2, lad, "resumed engine after 12 steps")
    t, a = eng.swap_counts()
    assert np.array_equal(t, lad.swap_count) and np.array_equal(a, lad.swap_accept_count)
    eng.close(); e2.close()


@pytest.mark.parametrize("kind,odf,with_mean,ev", [(E.PROP_LOWER, 0.0, False, 0.0), (E.PROP_DENSE, 0.4, True, 0.0), (E.PROP_LOWER, 0.3, False, 0.02),
                                                   (E.PROP_DENSE, 0.0, False, 0.02)])
def test_compacted_sweep_of_the_box_bounds_build_matches_the_oracle(kind, odf, with_mean, ev):
    """The usual real-world state space (uniform priors, `limit` / open boundaries, a mean, one-dimensional moves, an evolving
    ladder) on a big population without history: its MFMA build walks the moving chains only, like the lean one.  Narrow
    limits, so that a good share of the proposals is invalid; bit for bit the oracle's chains, counters and temperatures."""
    D, Nt, W = 30, 8, 1024
    rng = np.random.default_rng(35)
    blo = [1 if d % 3 else 0 for d in range(D)]
    bhi = [1 if d % 2 else 0 for d in range(D)]
    bmin = list(rng.uniform(-2.5, -1.5, D)); bmax = list(rng.uniform(1.5, 2.5, D))
    prior = ([1] * D, [0.0] * D, list(rng.uniform(3.0, 6.0, D)))
    x0 = rng.uniform(-1.4, 1.4, size=(Nt * W, D))
    mean = rng.normal(size=D) * 0.1 if with_mean else None
    pr, eng, lad = PU.make_pair(D, Nt, W, 1e3, kind=kind, bounds=(blo, bhi, bmin, bmax), prior=prior, swap_rate=0.3, x0=x0, mean=mean,
                                one_d_frac=(odf if odf > 0 else None))
    if ev:
        eng.set_evolve_temps(ev); lad.evolve_temps(ev)
    # bounds and nothing else: the build that carries no mean / one-dimensional moves / mixtures (GEN 3); else the box-bounds build
    assert eng.sweep_kernel_name.endswith(", %d, %s, true>" % (1 if (odf > 0 or with_mean) else 3, "true" if ev else "false"))
    for k in range(3):
        eng.step(3); eng.sync(); lad.pt_step(3)
        PU.assert_same_state(eng, lad, "after %d compacted steps" % (3 * (k + 1)))
        if ev:
            assert np.array_equal(eng.invtemps(), lad.betaw)
        if k == 0:
            eng.sweep(2); eng.sync(); lad.sweep(2)          # plain sweeps in between: every chain visited in place
            PU.assert_same_state(eng, lad, "after plain sweeps")
    tries, acc = eng.ntries.sum() - eng.Nc, eng.naccept.sum() - eng.Nc
    assert 0 < acc < 0.8 * tries
    if odf > 0:
        assert (eng.last_type == 1).any()
    t, a = eng.swap_counts()
    assert np.array_equal(t, lad.swap_count) and np.array_equal(a, lad.swap_accept_count)
    eng.close()


@pytest.mark.parametrize("overlap", [False, True])
def test_compacted_sweep_in_sharded_engines(overlap):
    """the compacted sweep through the sharded step's partial sweeps (interior / boundary rungs separately)"""
    import shard_sim
    from ptmcmc_amd.parallel import shard_bounds
    from ptmcmc_amd.problems import GaussianProblem
    D, Nt, W, G, sr = 32, 24, 1024, 3, 0.3
    pr = GaussianProblem(D, Nt, 1e3)
    ref = E.Engine(D, Nt, W, swap_rate=sr)
    pr.configure(ref, E.PROP_LOWER)
    ref.init_from_prior()
    x0 = ref.states()
    shards = []
    for g in range(G):
        r0, n = shard_bounds(Nt, G, g)
        e = E.Engine(D, Nt, W, swap_rate=sr, rung_begin=r0, rung_count=n)
        pr.configure(e, E.PROP_LOWER)
        e.set_states(x0[r0 * W:(r0 + n) * W])
        shards.append(e)
    lads = shard_sim.build([_DevShard(e) for e in shards], halo=4)
    copy = lambda dst, src: dst.copy_from(src.ptr)
    for k in range(3):
        ref.step(4)
        (shard_sim.step_overlapped if overlap else shard_sim.step)(lads, copy, 4)
        assert np.array_equal(np.concatenate([e.states() for e in shards]), ref.states()), "states differ after step %d" % (4 * k + 4)
    for name in ("llike", "ntries", "naccept", "nhist", "last_type"):
        assert np.array_equal(np.concatenate([getattr(e, name) for e in shards]), getattr(ref, name)), name
    for e in shards + [ref]:
        e.close()


@pytest.mark.parametrize("env,D,Nt,W,kind,want", [({"PTM_COMPACT": "0"}, 32, 6, 1024, "lower", "sweep_mfma32_kernel<2, false, 0, false, false>"),
                                                 ({"PTM_FORCE_VALU": "1"}, 32, 5, 64, "lower", "sweep_kernel<32"),
                                                 ({"PTM_FUSED": "0"}, 6, 12, 3, "dense", "sweep_lanes_kernel<8"),
                                                 ({"PTM_LADDER": "0"}, 32, 40, 2, "lower", "decide_kernel + sweep_lanes_kernel<32"),
                                                 ({"PTM_LADDER": "0"}, 12, 70, 3, "diag", "decide_kernel + sweep_lanes_kernel<16"),
                                                 ({"PTM_LADDER_MAXRUN": "1"}, 32, 64, 2, "lower", "ladder_persistent_kernel<32"),
                                                 ({"PTM_LADDER_MAXRUN": "2"}, 10, 90, 3, "dense", "ladder_persistent_kernel<16")])
def test_switched_off_variants_stay_bit_exact(env, D, Nt, W, kind, want):
    """The engine's environment switches select code that the default run never reaches: the un-compacted sweep of a big population, the general VALU
    kernel on the MFMA workload, the two-launch step of small and of long ladders, and the persistent ladder kernel with its longest
    admissible run of surviving picks lowered to 1 / 2 -- every few steps the ladder's workgroups then take the exchange phase
    from the whole ladder's publications (with the halo's 8 that path is a once-in-10^12-steps event).  Each in a process of its
    own (the switches are read once), each bit for bit the oracle's chains."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "variant_worker.py"), str(D), str(Nt), str(W), kind, "4", want],
                       env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("D,Nt,W,kind,kernel", [(128, 64, 1024, "lower", "sweep_mfma128_kernel<2, false, false>"), (100, 48, 512, "dense", "sweep_mfma128_kernel<0, false, false>"),
                                                (64, 64, 2048, "lower", "sweep_mfma64_kernel<2, false, false>"), (32, 128, 2048, "lower", "sweep_mfma32_kernel<2, false, 0, false, true>")])
def test_matrix_core_kernels_equal_the_vector_kernels_at_sizes_the_checker_cannot_walk(D, Nt, W, kind, kernel):
    """A size-independent property: at 65536+ chains the MFMA kernels (32 / 64 / 128 dimensions; the 32-dimensional one compacted)
    and the vector kernels that PTM_FORCE_VALU=1 selects walk the SAME chains -- digest of states, llikes, lpriors, counters and swap
    bookkeeping after 12 PT steps and two plain sweeps, each engine in a process of its own (the switch is read once).  The vector
    kernels are the ones the CPU checker pins at small sizes (SWEEP_CASES), so this carries that pin to full size."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    outs = []
    for env in ({}, {"PTM_FORCE_VALU": "1"}):
        r = subprocess.run([sys.executable, os.path.join(here, "hash_worker.py"), str(D), str(Nt), str(W), kind, "12"],
                           env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout[-2000:] + r.stderr[-3000:]
        outs.append(r.stdout.split())
    assert kernel in " ".join(outs[0][2:]) and "mfma" not in " ".join(outs[1][2:]), (outs[0], outs[1])
    assert outs[0][1] == outs[1][1], (outs[0], outs[1])
    assert int(outs[0][outs[0].index("accepts") + 1]) > 0 and int(outs[0][outs[0].index("swaps") + 1]) > 0


@pytest.mark.parametrize("D,Nt,W,kind,nsteps", [(32, 1024, 1, "lower", 20000), (16, 300, 3, "diag", 20000), (24, 90, 5, "dense", 20000)])
def test_persistent_ladder_kernel_equals_the_two_launch_path_over_many_steps(D, Nt, W, kind, nsteps):
    """The persistent ladder kernel's workgroups hand their rungs to each other through flags and counters, thousands of times per
    launch: a race there would show up rarely.  20 000 steps in two launches against the two-launch path (PTM_LADDER=0, which the
    CPU checker pins step by step) -- digest of states, llikes, counters and swap bookkeeping, each engine in a process of its own."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    outs = []
    for env in ({}, {"PTM_LADDER": "0"}):
        r = subprocess.run([sys.executable, os.path.join(here, "hash_worker.py"), str(D), str(Nt), str(W), kind, str(nsteps)],
                           env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout[-2000:] + r.stderr[-3000:]
        outs.append(r.stdout.split())
    assert outs[0][1] == outs[1][1], (outs[0], outs[1])
    assert int(outs[0][outs[0].index("swaps") + 1]) > 0


@pytest.mark.parametrize("W,cap", [(3, 6), (4096, 4)])
def test_reads_in_one_batch_equal_the_same_reads_one_by_one(W, cap):
    """ptm_batch_begin / ptm_batch_end: the reads between them return what the same calls return outside a bracket (small
    population: history ring in mapped host memory, read in place; large: ring on the device, the states through the big-array
    path), a step inside a bracket is refused, and so is an end without a begin."""
    from ptmcmc_amd.problems import GaussianProblem
    D, Nt = 7, 9
    pr = GaussianProblem(D, Nt, 1e3)
    e = E.Engine(D, Nt, W, swap_rate=0.4, history_rungs=Nt, history_capacity=cap, map_rungs=Nt)
    pr.configure(e, E.PROP_DENSE)
    e.set_evolve_temps(0.02)
    e.init_from_prior()
    e.step(11); e.sync()
    one = dict(x=e.states(), ll=e.llike, lpost=e.array(E.ARR_LPOST), nsize=e.nsize, hist=e.history(), swaps=e.last_swaps(),
               counts=e.swap_counts(), beta=e.invtemps(), map=e.map())
    with e.batch():
        got = dict(x=e.states(), ll=e.llike, lpost=e.array(E.ARR_LPOST), nsize=e.nsize, hist=e.history(), swaps=e.last_swaps(),
                   counts=e.swap_counts(), beta=e.invtemps(), map=e.map())
        with pytest.raises(E.PtmError):
            e.step(1)
    def same(a, b):
        if isinstance(a, dict):
            return all(same(a[k], b[k]) for k in a)
        if isinstance(a, tuple):
            return all(same(u, v) for u, v in zip(a, b))
        return np.array_equal(a, b)
    for k in one:
        assert same(one[k], got[k]), k
    with pytest.raises(E.PtmError):
        E._chk(e.L.ptm_batch_end(e.h))
    e.step(3); e.sync()                                  # the engine goes on as if nothing happened
    assert e.history()["row"].max() >= cap


LADDER_CASES = [
    # D, Nt, W, kind, swap_rate: long ladders of few walkers -> ladder_persistent_kernel (ptm_ladder_kernel.hpp)
    (32, 1024, 1, E.PROP_LOWER, 0.1),     # the reference's own shape: BASELINE's D = 32, 1024 temperatures, one ladder
    (32, 100, 2, E.PROP_LOWER, 0.1),      # 13 workgroups per ladder, the last one ragged (4 rungs)
    (32, 256, 4, E.PROP_DENSE, 0.25),     # C3's population
    (20, 37, 1, E.PROP_DENSE, 0.35),      # padded dimensions, a ragged last workgroup, many exchanges per step
    (16, 70, 3, E.PROP_DIAG, 0.3),        # 16 rungs per workgroup, diagonal proposals
    (9, 41, 2, E.PROP_LOWER, 0.2),        # 9 -> 16 dimensions
    (32, 8, 5, E.PROP_LOWER, 0.35),       # one workgroup per ladder: no neighbour at all
    (32, 9, 1, E.PROP_DIAG, 0.35),        # two workgroups, the second holds one rung
]


@pytest.mark.parametrize("D,Nt,W,kind", [(32, 200, 1, E.PROP_LOWER), (12, 90, 3, E.PROP_DIAG), (24, 40, 2, E.PROP_DENSE)])
def test_persistent_ladder_kernel_with_limit_bounds(D, Nt, W, kind):
    """The persistent ladder kernel with open / `limit` boundaries (a second box test per lane: boundary::enforce for those sides,
    states.cc:53-55): narrow limits, so that a good share of the proposals is invalid -- which is not the same as outside the prior
    (chain.cc:976-986) --, many steps per launch, bit for bit the oracle's chains."""
    rng = np.random.default_rng(5)
    blo = [1 if d % 3 else 0 for d in range(D)]
    bhi = [1 if d % 2 else 0 for d in range(D)]
    bnd = (blo, bhi, list(rng.uniform(-2.5, -1.5, D)), list(rng.uniform(1.5, 2.5, D)))
    prior = ([1] * D, [0.0] * D, list(rng.uniform(3.0, 6.0, D)))
    x0 = rng.uniform(-1.2, 1.2, size=(Nt * W, D))
    pr, eng, lad = PU.make_pair(D, Nt, W, 1e4, kind=kind, bounds=bnd, prior=prior, swap_rate=0.2, x0=x0)
    assert eng.step_kernel_name.startswith("ladder_persistent_kernel<%d" % (16 if D <= 16 else 32)), eng.step_kernel_name
    for n in (1, 2, 40, 100):
        eng.step(n); eng.sync(); lad.pt_step(n)
        PU.assert_same_state(eng, lad, "after %d more PT steps in one launch" % n)
    t, a = eng.swap_counts()
    assert np.array_equal(t, lad.swap_count) and np.array_equal(a, lad.swap_accept_count)
    tries, acc = eng.ntries.sum() - eng.Nc, eng.naccept.sum() - eng.Nc
    assert 0 < acc < 0.9 * tries
    eng.close()


@pytest.mark.parametrize("D,Nt,W,kind,sr", LADDER_CASES)
def test_persistent_ladder_kernel_matches_the_oracle(D, Nt, W, kind, sr):
    """Long ladders of few walkers step through ONE launch per ptm_step(n) call: resident workgroups of 8 (16) rungs keep their
    chains in registers, replay the whole ladder's candidate draws and talk to their two neighbours through flags
    (ptm_ladder_kernel.hpp).  Bit for bit the oracle's chains -- states, llikes, MH_chain counters, swap counters, the last
    step's candidate log -- step by step and over many steps per launch."""
    pr, eng, lad = PU.make_pair(D, Nt, W, 1e6, kind=kind, swap_rate=sr)
    assert eng.step_kernel_name.startswith("ladder_persistent_kernel<%d" % (16 if D <= 16 else 32)), eng.step_kernel_name
    for k in range(4):
        eng.step(1); eng.sync(); lad.pt_step(1)
        PU.assert_same_state(eng, lad, "after PT step %d" % (k + 1))
        pairs, acc = eng.last_swaps()
        assert np.array_equal(pairs, lad.last_pairs) and np.array_equal(acc, lad.last_accept), k
    for n in (2, 3, 150):
        eng.step(n); eng.sync(); lad.pt_step(n)
        PU.assert_same_state(eng, lad, "after %d more PT steps in one launch" % n)
        pairs, acc = eng.last_swaps()
        assert np.array_equal(pairs, lad.last_pairs) and np.array_equal(acc, lad.last_accept), n
    t, a = eng.swap_counts()
    assert np.array_equal(t, lad.swap_count) and np.array_equal(a, lad.swap_accept_count)
    assert a.sum() > 0 and eng.naccept.sum() > eng.Nc
    # plain sweeps and the two-launch building blocks still work on the state the kernel left
    eng.sweep(3); eng.sync(); lad.sweep(3)
    PU.assert_same_state(eng, lad, "after plain sweeps")
    eng.step(5); eng.sync(); lad.pt_step(5)
    PU.assert_same_state(eng, lad, "after PT steps again")
    eng.close()


LADDER_EVOLVING = [
    # D, Nt, W, kind, swap_rate, evolve rate, gauss_1d_frac, mixture members K, add_every_N (0: no history / MAP)
    (32, 1024, 1, E.PROP_LOWER, 0.1, 0.01, 0.0, 0, 0),      # the reference's own shape with the sampler's default ladder (ptmcmc.cc:389,512)
    (32, 1024, 1, E.PROP_LOWER, 0.1, 0.01, 0.5, 4, 2),      # ... and everything else the sampler switches on: recipe, history, MAP
    (32, 100, 2, E.PROP_DENSE, 0.3, 0.03, 0.0, 0, 0),       # ragged last workgroup, many pries per step
    (16, 70, 3, E.PROP_DIAG, 0.3, 0.02, 0.3, 2, 1),         # 16 rungs per workgroup, every add saved: every in-between row with its own temperature
    (20, 37, 1, E.PROP_DENSE, 0.45, 0.05, 0.0, 0, 1),       # history alone
    (32, 8, 5, E.PROP_LOWER, 0.35, 0.02, 0.4, 3, 3),        # one workgroup per ladder
    (32, 1500, 1, E.PROP_DIAG, 0.1, 0.01, 0.0, 0, 0),       # more than 1024 rungs: 47 chunks of gaps
]


@pytest.mark.parametrize("D,Nt,W,kind,sr,rate,odf,K,N", LADDER_EVOLVING)
def test_persistent_ladder_kernel_with_evolving_ladders(D, Nt, W, kind, sr, rate, odf, K, N):
    """parallel_tempering_chains::evolve_temps (the reference sampler's default: pry_temps after every accepted exchange,
    chain.cc:1501-1518,1809-1846) inside the persistent ladder kernel: every workgroup replays every trial of the ladder in pick order
    from the whole ladder's published llikes and keeps the ladder's temperatures; the Metropolis tests take the step's new
    temperatures.  States, counters, the temperatures themselves, and -- in the build with everything -- every saved row with the
    temperature it was saved at and every rung's MAP: bit for bit the oracle's."""
    pr, eng, lad = _ladder_flavour_pair(D, Nt, W, kind, sr, odf, K, N)
    eng.set_evolve_temps(rate); lad.evolve_temps(rate)
    fl = 7 if (odf > 0 or K > 0 or N) else 4
    want = "ladder_persistent_kernel<%d, %d, %d>" % (16 if D <= 16 else 32, 1 if kind == E.PROP_DIAG else 0, fl)
    assert eng.step_kernel_name == want, (eng.step_kernel_name, want)
    done = 0
    for n in (1, 1, 2, 30):
        eng.step(n); eng.sync(); lad.pt_step(n)
        done += n
        PU.assert_same_state(eng, lad, "after %d PT steps" % done)
        assert np.array_equal(eng.invtemps(), lad.betaw), "temperatures differ after %d steps" % done
        if N:
            _assert_same_history_and_map(eng, lad)
            he, ho = eng.history(), lad.history()
            nsize = eng.nsize
            for s_ in range(int(nsize.max())):
                have = nsize > s_
                assert np.array_equal(he["invtemp"][s_ % 64][have], PU.to_engine_order(ho["invtemp"][:, s_], Nt, W)[have]), ("invtemp", s_)
    assert not np.array_equal(eng.invtemps()[0], pr.beta)
    t, a = eng.swap_counts()
    assert np.array_equal(t, lad.swap_count) and np.array_equal(a, lad.swap_accept_count)
    pairs, acc = eng.last_swaps()
    assert np.array_equal(pairs, lad.last_pairs) and np.array_equal(acc, lad.last_accept)
    # the two-launch path takes over the ladders the kernel left (temperatures, their chain-indexed image)
    eng.sweep(2); eng.sync(); lad.sweep(2)
    PU.assert_same_state(eng, lad, "after plain sweeps")
    st = eng.ladder_stats()
    assert st["launches"] == 4 and st["fallbacks"] == 0, st
    eng.close()


LADDER_FLAVOURS = [
    # D, Nt, W, kind, swap_rate, gauss_1d_frac, mixture members K, add_every_N (0: no history / MAP)
    (32, 1024, 1, E.PROP_LOWER, 0.1, 0.3, 3, 3),     # the reference's own shape with the sampler's default Gaussian recipe, saved every 3rd add
    (32, 100, 2, E.PROP_DENSE, 0.25, 0.5, 0, 2),     # one-dimensional moves + history, no mixture
    (16, 70, 3, E.PROP_DIAG, 0.3, 0.0, 4, 0),        # mixture alone
    (20, 37, 1, E.PROP_DENSE, 0.35, 0.0, 0, 1),      # history alone, every add saved: every in-between row of a rung exchanged twice is one
    (32, 8, 5, E.PROP_LOWER, 0.35, 0.4, 2, 5),       # one workgroup per ladder
    (9, 41, 2, E.PROP_LOWER, 0.45, 0.25, 1, 1),      # a set of one (draws nothing), many exchanges per step
]


def _ladder_flavour_pair(D, Nt, W, kind, sr, odf, K, N, cap=64):
    pr = PU.problem_for(D, Nt, 1e4)
    eng = E.Engine(D, Nt, W, swap_rate=sr, add_every_n=max(N, 1), history_rungs=Nt if N else 0, history_capacity=cap if N else 0, map_rungs=Nt if N else 0)
    fac = pr.configure(eng, kind, np.full(Nt, odf) if odf > 0 else None)
    eng.init_from_prior()
    x0 = eng.states()
    lad = O.Ladder(PU.oracle_problem(pr), pr.beta, W=W, swap_rate=sr, add_every_N=max(N, 1))
    lad.set_proposals([(PU.KIND_TO_ORACLE[kind], fac[r], odf) for r in range(Nt)])
    lad.use_philox(0x5EED0001)
    if N:
        lad.enable_history(cap)
    lad.set_states(PU.to_oracle_order(x0, Nt, W))
    if K:
        rng = np.random.default_rng(K)
        shares = 2.0 ** np.arange(1, K + 1)
        cum = np.tile(np.cumsum(shares) / shares.sum(), (Nt, 1)); cum[:, -1] = 1.0
        scales = np.tile(3.0 ** -np.arange(K)[::-1] * 1.2, (Nt, 1)) * rng.uniform(0.8, 1.2, (Nt, 1))
        odfs = np.tile(np.where(np.arange(K) % 2 == 0, odf, 0.0), (Nt, 1))
        eng.set_proposal_mixture(cum, scales, odfs); lad.set_mixture(cum, scales, odfs)
    return pr, eng, lad


def _assert_same_history_and_map(eng, lad, cap=64):
    Nt, W = eng.Nt, eng.W
    he, ho = eng.history(), lad.history()
    nsize = eng.nsize
    for name in ("x", "llike", "lprior", "naccept", "ntries", "last_type"):
        a, b = he[name], ho[name]
        for s_ in range(max(0, int(nsize.max()) - cap), int(nsize.max())):
            have = (nsize > s_) & (nsize - s_ <= cap)
            got, want = a[s_ % cap][have], PU.to_engine_order(b[:, s_ % cap] if b.shape[1] == cap and nsize.max() > cap else b[:, s_], Nt, W)[have]
            assert np.array_equal(got, want), (name, s_, np.argwhere(got != want)[:3].tolist())
    m = eng.map()
    assert np.array_equal(m["lpost"], PU.to_engine_order(lad.map_lpost, Nt, W))
    assert np.array_equal(m["x"], PU.to_engine_order(lad.map_x, Nt, W))


@pytest.mark.parametrize("D,Nt,W,kind,sr,odf,K,N", LADDER_FLAVOURS)
def test_persistent_ladder_kernel_runs_what_the_sampler_runs(D, Nt, W, kind, sr, odf, K, N):
    """The persistent ladder kernel's builds for the reference sampler's defaults (ptmcmc.cc:117-139,601-616): one-dimensional moves and
    scale mixtures (FL bit 0), the history ring and MAP tracking of MH_chain::add_state (chain.cc:931-946; FL bit 1) -- the rows a rung
    holds BETWEEN its two exchanges of one step included (quirk Q6).  States, counters, type codes, every saved row and every rung's
    MAP bit for bit the oracle's, step by step and over many steps per launch."""
    pr, eng, lad = _ladder_flavour_pair(D, Nt, W, kind, sr, odf, K, N)
    fl = (1 if (odf > 0 or K > 0) else 0) | (2 if N else 0)
    want = "ladder_persistent_kernel<%d, %d, %d>" % (16 if D <= 16 else 32, 1 if kind == E.PROP_DIAG else 0, fl)
    assert eng.step_kernel_name == want, (eng.step_kernel_name, want)
    done = 0
    for n in (1, 1, 2, 40):
        eng.step(n); eng.sync(); lad.pt_step(n)
        done += n
        PU.assert_same_state(eng, lad, "after %d PT steps" % done)
        if N:
            _assert_same_history_and_map(eng, lad)
    t, a = eng.swap_counts()
    assert np.array_equal(t, lad.swap_count) and np.array_equal(a, lad.swap_accept_count)
    if N:
        assert int((eng.nhist > done).sum()) > 0          # some rungs made two add_state calls in one step
    if odf > 0 or K > 1:
        assert len(set(int(v) for v in np.unique(eng.last_type)) - {-1}) >= 2
    st = eng.ladder_stats()
    assert st["launches"] == 4 and st["fallbacks"] == 0 and not st["disabled"], st
    eng.close()


def test_persistent_ladder_kernel_that_gives_up_leaves_nothing_behind():
    """A workgroup of the persistent ladder kernel that waits in vain for a neighbour gives up (a grid that is not resident: a shared
    device).  PTM_LADDER_SPIN_US=0 makes every flag that is not up at the first look such a case.  The launch then commits NOTHING (the
    chains live in registers until every workgroup has finished every step), the engine repeats its steps on the two-launch path at
    its next look and keeps that path: the run equals the oracle's bit for bit, the event is counted, no error is raised."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "variant_worker.py"), "32", "600", "1", "lower", "giveup"],
                       env=dict(os.environ, PTM_LADDER_SPIN_US="0", PTM_QUIET="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "fallbacks=" in r.stdout and "fallbacks=0" not in r.stdout, r.stdout[-500:]


def test_two_engines_with_persistent_ladder_kernels_take_turns():
    """Two engines of one process, each with a grid of the persistent ladder kernel as large as the device holds alone: their launches
    are put behind each other (an event of the last launch of the OTHER engine), so neither waits for workgroups that cannot become
    resident; both equal their oracles and nobody gives up."""
    pairs = [PU.make_pair(32, 1024, 1, 1e6, kind=E.PROP_LOWER, seed=0x5EED0001 + k) for k in range(2)]
    for _, eng, _ in pairs:
        assert eng.step_kernel_name.startswith("ladder_persistent_kernel<32")
    for rep in range(3):
        for _, eng, _ in pairs:
            eng.step(200)                      # (asynchronous: the second engine's launch is queued while the first one's runs)
    for _, eng, lad in pairs:
        eng.sync(); lad.pt_step(600)
        PU.assert_same_state(eng, lad, "after 600 PT steps beside another engine")
        st = eng.ladder_stats()
        assert st["fallbacks"] == 0 and st["launches"] == 3, st
        eng.close()


def test_persistent_ladder_kernel_after_the_proposal_kind_changes():
    """The residency figure and the LDS attribute of the persistent ladder kernel are kept per build: a long ladder (more than 64 KB of
    LDS per workgroup) that changes from diagonal to full factors between two ptm_step calls launches the other build with its own
    figures, and stays bit-identical to the oracle."""
    D, Nt, W = 32, 1500, 1
    pr, eng, lad = PU.make_pair(D, Nt, W, 1e6, kind=E.PROP_DIAG, swap_rate=0.1)
    name = eng.step_kernel_name
    eng.step(30); eng.sync(); lad.pt_step(30)
    PU.assert_same_state(eng, lad, "diagonal factors")
    fac = pr.proposal_factors(range(Nt), lower=True)
    eng.set_proposals(E.PROP_LOWER, fac)
    lad.set_proposals([(O.PROP_DENSE, fac[r], 0.0) for r in range(Nt)])
    assert eng.step_kernel_name != name or not name.startswith("ladder_persistent")
    eng.step(30); eng.sync(); lad.pt_step(30)
    PU.assert_same_state(eng, lad, "full factors")
    assert eng.ladder_stats()["fallbacks"] == 0
    eng.close()


@pytest.mark.parametrize("D,Nt,W,kind,ev,hist", [(32, 40, 3, E.PROP_LOWER, 0.0, 0), (12, 24, 5, E.PROP_DENSE, 0.02, 2), (5, 40, 2, E.PROP_DIAG, 0.0, 1), (20, 16, 64, E.PROP_LOWER, 0.01, 0)])
def test_persistent_ladder_kernel_with_a_target_mean(D, Nt, W, kind, ev, hist):
    """A Gaussian target that is not centred on the origin (gaussian_likelihood's mean: cython/exampleGaussian.py:46-109) in the
    persistent ladder kernel -- one subtraction on the dimension's lane in front of the precision product, as in every other kernel
    -- plain, evolving, with history and MAP, 4..32 padded dimensions, whole waves per rung."""
    rng = np.random.default_rng(D + Nt)
    mean = rng.normal(size=D) * 0.3
    pr, eng, lad = PU.make_pair(D, Nt, W, 1e3, kind=kind, swap_rate=0.3, one_d_frac=0.3 if hist else None, mean=mean, add_every_n=max(1, hist),
                                history_cap=64 if hist else 0)
    if ev:
        eng.set_evolve_temps(ev); lad.evolve_temps(ev)
    assert eng.step_kernel_name.startswith("ladder_persistent_kernel<"), eng.step_kernel_name
    for n in (1, 5, 24):
        eng.step(n); eng.sync(); lad.pt_step(n)
        PU.assert_same_state(eng, lad, "after %d more steps" % n)
        if ev:
            assert np.array_equal(eng.invtemps(), lad.betaw)
    if hist:
        PU.assert_same_history_and_map(eng, lad, 64)
    st = eng.ladder_stats()
    assert st["launches"] > 0 and st["fallbacks"] == 0
    eng.close()


def test_history_of_a_range_of_chains_is_the_whole_history_s_slice():
    """ptm_get_history_chains: the ring entries of some chains only -- what a chain-file writer reads for the rungs it dumps -- are,
    entry for entry, those of ptm_get_history; the rest of the (full-layout) arrays is left alone."""
    D, Nt, W = 7, 9, 5
    pr, eng, lad = PU.make_pair(D, Nt, W, 1e3, kind=E.PROP_DENSE, swap_rate=0.4, add_every_n=2, history_cap=16)
    eng.set_evolve_temps(0.02)
    eng.step(40); eng.sync()
    full = eng.history()
    for b, n in ((0, W), (2 * W + 1, 3 * W), (Nt * W - 4, 4), (0, Nt * W)):
        part = eng.history_chains(b, n)
        sl = slice(b, b + n)
        assert np.array_equal(part["x"][:, sl], full["x"][:, sl]) and np.array_equal(part["llike"][:, sl], full["llike"][:, sl])
        assert np.array_equal(part["lprior"][:, sl], full["lprior"][:, sl]) and np.array_equal(part["invtemp"][:, sl], full["invtemp"][:, sl])
        assert np.array_equal(part["meta"][:, sl, 3], full["row"][:, sl]) and np.array_equal(part["meta"][:, sl, 0], full["naccept"][:, sl])
        rest = np.ones(Nt * W, bool); rest[sl] = False
        assert np.isnan(part["x"][:, rest]).all() and (part["meta"][:, rest] == -1).all()
    with pytest.raises(E.PtmError):
        eng.history_chains(Nt * W - 2, 3)
    eng.close()


@pytest.mark.parametrize("D,Nt,W,kind,ev,hist,K,want", [(12, 40, 3, E.PROP_DENSE, 0.0, 0, 0, 19), (32, 24, 2, E.PROP_LOWER, 0.02, 2, 3, 23), (6, 30, 5, E.PROP_DIAG, 0.02, 1, 2, 23),
                                                      (3, 12, 64, E.PROP_DENSE, 0.0, 1, 0, 19)])
def test_persistent_ladder_kernel_with_any_boundary_and_any_per_dimension_prior(D, Nt, W, kind, ev, hist, K, want):
    """The general state space in the persistent ladder kernel (its builds FL = 19 / 23 / 27 / 31): wrap, reflect, limit and open
    boundaries (boundary::enforce, states.cc:11-58), a mixed prior -- uniform and Gaussian factors, four interleaved partial products
    (mixed_dist_product::evaluate, probability_function.cc:219-262) --, a target mean; plain, evolving, with history, MAP and a scale
    mixture: bit for bit the checker's chains, temperatures, saved rows."""
    rng = np.random.default_rng(D * 7 + Nt)
    blo = [int(rng.choice([0, 1, 2, 3])) for _ in range(D)]
    bhi = [b if b in (2, 3) else int(rng.choice([0, 1])) for b in blo]
    bounds = (blo, bhi, list(rng.uniform(-3.0, -1.5, D)), list(rng.uniform(1.5, 3.0, D)))
    types, cen, hw = [1] * D, [0.0] * D, list(rng.uniform(3.5, 6.0, D))
    for d in range(0, D, 2):
        types[d] = 2; cen[d] = float(rng.normal() * 0.2); hw[d] = float(rng.uniform(0.8, 2.0))
    mean = rng.normal(size=D) * 0.1
    x0 = rng.uniform(-1.2, 1.2, size=(Nt * W, D))
    pr, eng, lad = PU.make_pair(D, Nt, W, 1e3, kind=kind, swap_rate=0.3, one_d_frac=0.3 if K else None, bounds=bounds, prior=(types, cen, hw), mean=mean, x0=x0,
                                add_every_n=max(1, hist), history_cap=64 if hist else 0)
    if K:
        shares = 2.0 ** np.arange(1, K + 1)
        cum = np.tile(np.cumsum(shares) / shares.sum(), (Nt, 1)); cum[:, -1] = 1.0
        scales = np.tile(2.0 ** -np.arange(K)[::-1], (Nt, 1)); odfs = np.full((Nt, K), 0.3)
        eng.set_proposal_mixture(cum, scales, odfs); lad.set_mixture(cum, scales, odfs)
    if ev:
        eng.set_evolve_temps(ev); lad.evolve_temps(ev)
    assert eng.step_kernel_name.startswith("ladder_persistent_kernel<") and eng.step_kernel_name.endswith(", %d>" % want), eng.step_kernel_name
    for n in (1, 6, 23):
        eng.step(n); eng.sync(); lad.pt_step(n)
        PU.assert_same_state(eng, lad, "after %d more steps" % n)
        if ev:
            assert np.array_equal(eng.invtemps(), lad.betaw)
    if hist:
        PU.assert_same_history_and_map(eng, lad, 64)
    tries, acc = eng.ntries.sum() - eng.Nc, eng.naccept.sum() - eng.Nc
    assert 0 < acc < tries
    st = eng.ladder_stats()
    assert st["launches"] > 0 and st["fallbacks"] == 0
    eng.close()
