"""Synthetic workloads of BASELINE.json / SURVEY.md section 8(d): a zero-mean correlated Gaussian target
(cython/exampleGaussian.py:46-109), uniform box prior, geometric ladder, per-rung Gaussian step proposals."""
import math

import numpy as np

from . import engine as E


def covariance(D, key=0xC0FFEE):
    """Sigma = A^T A / D + 0.1 I, A_ij ~ U(-1,1) from a fixed Philox key."""
    g = np.random.Generator(np.random.Philox(key=key))
    A = g.uniform(-1.0, 1.0, size=(D, D))
    return A.T @ A / D + 0.1 * np.eye(D)


class GaussianProblem:
    """Everything the engine needs for the correlated-Gaussian ladder, as plain arrays."""

    def __init__(self, D, n_rungs, tmax, prior_scale=100.0, basescale_fac=0.5, key=0xC0FFEE, cov_scale=1.0):
        self.D, self.Nt, self.tmax = D, n_rungs, tmax
        self.cov = covariance(D, key) * cov_scale
        self.P = np.linalg.inv(self.cov)
        self.P = 0.5 * (self.P + self.P.T)
        # exampleGaussian.py:53-54: like0 = -0.5*(npar*log(2 pi) + ln det cov)
        self.like0 = -0.5 * (D * math.log(2 * math.pi) + np.linalg.slogdet(self.cov)[1])
        # exampleGaussian.py:71-76: uniform prior, half-width 100 sigma, trivial (open) boundaries
        self.halfwidths = prior_scale * np.sqrt(np.diag(self.cov))
        self.centers = np.zeros(D)
        self.types = [E.PRIOR_UNIFORM] * D
        self.beta = E.geometric_ladder(n_rungs, tmax)
        self.basescale_fac = basescale_fac

    def proposal_factors(self, rungs=None, lower=True):
        """exampleGaussian.py:88-95,145: Sigma_c = inv(beta_c * invcov + diag((f h)^-2)) * 2.38^2/D; Cholesky factor."""
        D = self.D
        rungs = range(self.Nt) if rungs is None else rungs
        out = np.empty((len(rungs), D, D))
        base = np.diag((self.basescale_fac * self.halfwidths) ** -2.0) if self.basescale_fac > 0 else 0.0
        for k, r in enumerate(rungs):
            S = np.linalg.inv(self.beta[r] * self.P + base) * (2.38 ** 2 / D)
            S = 0.5 * (S + S.T)
            if lower:
                out[k] = np.linalg.cholesky(S)
            else:  # gaussian_prop(cov) form: eigenvectors * sqrt(eigenvalues) (proposal_distribution.hh:173-176,207-213)
                lam, V = np.linalg.eigh(S)
                out[k] = V * np.sqrt(np.maximum(lam, 0.0))
        return out

    def configure(self, eng, kind=E.PROP_LOWER, one_d_frac=None):
        D = self.D
        eng.set_bounds([E.BOUND_OPEN] * D, [E.BOUND_OPEN] * D, np.zeros(D), np.zeros(D))
        eng.set_prior(self.types, self.centers, self.halfwidths)
        eng.set_target_gaussian(self.P, self.like0)
        eng.set_ladder(self.beta)
        rungs = range(eng.r0, eng.r0 + eng.nloc)
        if kind == E.PROP_DIAG:
            f = np.stack([np.sqrt(np.diag(T @ T.T)) for T in self.proposal_factors(rungs)])
        else:
            f = self.proposal_factors(rungs, lower=(kind == E.PROP_LOWER))
        eng.set_proposals(kind, f, one_d_frac)
        return f


# BASELINE.json configs: name -> (D, rungs, walkers, Tmax)
CONFIGS = {
    "C1": (2, 8, 1, 1e2),
    "C2": (16, 64, 1, 1e4),
    "C3": (32, 256, 4, 1e6),
    "C4": (32, 1024, 1, 1e9),
}
