"""Does the engine SAMPLE the right distributions?  Independent of the CPU checker (the parity tests compare bits; these compare
moments).  The reference's own tests of this kind: testGaussian.cc:62-73 (a chain's covariance against the target's) and
test_proposal.hh:122-184 (a proposal leaves the target invariant: start from exact samples, step, compare).

Every case starts MANY independent ladders from EXACT samples of every rung's tempered target N(0, cov / beta_r) (the prior box is
100 sigma wide: no truncation), runs PT steps -- exchanges and Metropolis moves -- and takes snapshots far apart.  Across walkers and
snapshots rung r must still show the covariance cov / beta_r.  Threshold: the estimator's own noise.  An entry of the sample
covariance of n independent Gaussian samples, in units of sigma_i sigma_j, has a standard deviation of at most sqrt(2 / n); the
walkers of a snapshot are independent, snapshots of one walker are not quite (n_eff = n / 2 is assumed: the spacing is about two
autocorrelation times), and the largest of the D (D + 1) / 2 entries of the colder rungs is compared with 5 standard deviations
(the largest of 528 standard normals exceeds 5 with probability 3e-4).  A kernel that sampled a covariance wrong by a few per cent,
or the wrong temperature, fails by a wide margin at these sample counts; one that did not leave the target invariant drifts within
the first snapshots."""
import numpy as np
import pytest

from ptmcmc_amd import engine as E
from ptmcmc_amd.problems import GaussianProblem

pytestmark = pytest.mark.gpu


def _exact_start(pr, Nt, W, rng):
    L = np.linalg.cholesky(pr.cov)
    z = rng.standard_normal((Nt, W, pr.D))
    return (z @ L.T) / np.sqrt(np.asarray(pr.beta))[:, None, None]


def _worst_error(acc, n, pr, beta, rungs):
    errs = []
    for r in rungs:
        C = acc[r] / n
        want = pr.cov / beta[r]
        s = np.sqrt(np.diag(want))
        errs.append(np.abs((C - want) / np.outer(s, s)).max())
    return max(errs)


STAT_CASES = [
    # name, D, Nt, W, kind, evolve rate, snapshots, spacing (steps), what must be in the kernel name
    ("general / fused kernel", 8, 12, 8192, E.PROP_LOWER, 0.0, 10, 100, "ladder_steps_kernel<8"),
    ("f64 matrix cores, 32 dimensions", 32, 8, 4096, E.PROP_LOWER, 0.0, 10, 150, "sweep_mfma32_kernel"),
    ("... with evolving ladders", 32, 8, 4096, E.PROP_LOWER, 0.01, 10, 150, "sweep_mfma32_kernel"),
    # (300 walkers: more ladders than the persistent ladder kernel's grid holds -- the fused small-ladder kernel / exchange + lanes kernel)
    ("small ladder in one workgroup", 12, 10, 300, E.PROP_DENSE, 0.0, 120, 60, "ladder_steps_kernel<16"),
    ("lanes kernel", 12, 40, 100, E.PROP_DENSE, 0.0, 360, 60, "decide_kernel + sweep_lanes_kernel<16"),
    ("persistent ladder kernel, 8 padded dimensions", 8, 40, 60, E.PROP_DENSE, 0.0, 600, 60, "ladder_persistent_kernel<8"),
    ("f64 matrix cores, 64 dimensions", 64, 6, 2048, E.PROP_LOWER, 0.0, 8, 300, "sweep_mfma64_kernel"),
    ("f64 matrix cores, 128 dimensions, evolving", 128, 5, 1024, E.PROP_LOWER, 0.01, 8, 600, "sweep_mfma128_kernel"),
    ("persistent ladder kernel", 32, 40, 48, E.PROP_LOWER, 0.0, 1500, 60, "ladder_persistent_kernel<32"),
    ("persistent ladder kernel, the sampler's recipe", 32, 40, 48, E.PROP_LOWER, -1.0, 1500, 60, "ladder_persistent_kernel<32, 0, 1>"),
]


@pytest.mark.parametrize("name,D,Nt,W,kind,ev,nsnap,spacing,kernel", STAT_CASES, ids=[c[0] for c in STAT_CASES])
def test_every_rung_keeps_the_covariance_of_its_tempered_target(name, D, Nt, W, kind, ev, nsnap, spacing, kernel):
    rng = np.random.default_rng(D * 100 + Nt)
    pr = GaussianProblem(D, Nt, 1e2)
    eng = E.Engine(D, Nt, W, swap_rate=0.2, seed=0xC0FFEE + D)
    pr.configure(eng, kind)
    if ev > 0:
        eng.set_evolve_temps(ev)
    if ev < 0:   # the reference sampler's default Gaussian recipe (ptmcmc.cc:117-139): four scales, half the moves one-dimensional
        K = 4
        sh = np.cumsum([2.0 ** (k + 1) for k in range(K)]); sh /= sh[-1]
        eng.set_proposal_mixture(np.tile(sh, (Nt, 1)), np.tile([2.0 ** -k for k in range(K)], (Nt, 1)), np.full((Nt, K), 0.5))
    eng.set_states(_exact_start(pr, Nt, W, rng).reshape(Nt * W, D))
    if kernel:
        assert kernel in eng.step_kernel_name or kernel in eng.sweep_kernel_name, (eng.step_kernel_name, eng.sweep_kernel_name)
    acc = np.zeros((Nt, D, D))
    for k in range(nsnap):
        eng.step(spacing); eng.sync()
        X = eng.states().reshape(Nt, W, D)
        acc += np.einsum("rwi,rwj->rij", X, X)
    n = nsnap * W
    beta = eng.invtemps().mean(axis=0)
    bound = 5.0 * np.sqrt(2.0 / (n / 2.0))
    # an evolving ladder's interior rungs have a different temperature in every walker: its cold rung (beta = 1 always) is the test
    rungs = [0] if ev > 0 else list(range(max(1, Nt // 2)))
    err = _worst_error(acc, n, pr, beta, rungs)
    assert err < bound, "%s: max |C - cov / beta| / (sigma_i sigma_j) = %.4f over rungs %s, bound %.4f (%d samples per rung)" % (name, err, rungs, bound, n)
    # ... and the chains did move (a kernel that never accepted would pass the first test from an exact start)
    tries, acc_ = eng.ntries.sum() - eng.Nc, eng.naccept.sum() - eng.Nc
    assert 0.05 * tries < acc_ < 0.95 * tries
    t, a = eng.swap_counts()
    if D <= 32:   # (six rungs over a factor 100 in temperature exchange next to nothing at 64 and more dimensions: the Metropolis moves carry those cases)
        assert a.sum() > 0.05 * t.sum()
    print("%s: %.4f (bound %.4f, %d samples per rung; exchanges accepted %.3f), kernel %s" % (name, err, bound, n, a.sum() / max(1, t.sum()), eng.step_kernel_name))
    eng.close()


def test_differential_evolution_on_the_device_keeps_the_target():
    """The same for the default proposal recipe drawn on the device -- 70 % differential evolution from the device's own history,
    the rest Gaussians --: started from exact samples with a history of exact samples, the cold rungs keep their covariance."""
    D, Nt, W, N, nsnap, spacing = 6, 10, 512, 10, 200, 60
    rng = np.random.default_rng(77)
    pr = GaussianProblem(D, Nt, 1e2)
    steps = nsnap * spacing
    eng = E.Engine(D, Nt, W, swap_rate=0.2, add_every_n=N, history_rungs=Nt, history_capacity=2 * steps // N + 8, seed=0xDE)
    pr.configure(eng, E.PROP_DIAG)
    g = 2.0 ** np.arange(1, 4)
    shares = np.concatenate([[0.7], 0.3 * g / g.sum()])
    cum = np.tile(np.cumsum(shares), (Nt, 1)); cum[:, -1] = 1.0
    eng.set_proposal_mixture(cum, np.tile([-1.0, 0.25, 0.5, 1.0], (Nt, 1)), np.tile([0.0, 0.5, 0.5, 0.5], (Nt, 1)))
    init = np.stack([_exact_start(pr, Nt, W, rng).reshape(Nt * W, D) for _ in range(10 * D)])
    eng.set_proposal_de(0.1, 0.3, 4.0, 0.0, init_rows=init)
    eng.set_states(_exact_start(pr, Nt, W, rng).reshape(Nt * W, D))
    assert eng.sweep_kernel_name.startswith("sweep_kernel<")
    acc = np.zeros((Nt, D, D))
    for k in range(nsnap):
        eng.step(spacing); eng.sync()
        X = eng.states().reshape(Nt, W, D)
        acc += np.einsum("rwi,rwj->rij", X, X)
    n = nsnap * W
    bound = 5.0 * np.sqrt(2.0 / (n / 2.0))
    err = _worst_error(acc, n, pr, np.asarray(pr.beta), list(range(Nt // 2)))
    assert err < bound, (err, bound)
    print("differential evolution on the device: %.4f (bound %.4f, %d samples per rung)" % (err, bound, n))
    lt = eng.last_type
    assert ((lt == 0) | (lt == 10)).mean() > 0.3      # differential-evolution moves were accepted
    eng.close()
