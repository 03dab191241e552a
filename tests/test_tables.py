"""The Box-Muller radius table (generated data shared by the kernels and the CPU checker) and the function built on it."""
import math
import os
import sys
from decimal import Decimal, getcontext

import numpy as np

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_table_files_are_the_generators_output():
    import gen_tables
    txt = gen_tables.render()
    for rel in ("ptmcmc_amd/csrc/ptm_tables.inc", "oracle/ptm_tables.inc"):
        assert open(os.path.join(ROOT, rel)).read() == txt, rel


def test_table_entries_rederived():
    """independent derivation: mpmath-free, via math.log1p on the exactly known rounding residual of 1/c"""
    import gen_tables
    ent = gen_tables.entries()
    assert len(ent) == 256 and ent[255] == (0.5, 0.0)
    for i, (rc, a) in enumerate(ent[:255]):
        c = 1.0 + (i + 0.5) / 256.0
        assert rc == 1.0 / c
        # ln(rc) = -ln(c) + log1p(rc*c - 1), with rc*c - 1 exact in rational arithmetic
        from fractions import Fraction
        resid = float(Fraction(rc) * Fraction(c) - 1)
        want = 2 * (-math.log(c) + math.log1p(resid)) + (2 * math.log(2.0) if i >= gen_tables.SPLIT else 0.0)
        assert abs(a - want) <= 7e-16, (i, a, want)     # (the float64 check itself cancels ~1 + 1.4)
        assert abs(a) < 0.71


def _exact(k):
    getcontext().prec = 50
    return -2 * (Decimal(2 * k + 1) / Decimal(2 ** 33)).ln()


def test_neg2log_accuracy_and_domain():
    rng = np.random.default_rng(5)
    ks = [0, 1, 2, 3, 2 ** 32 - 1, 2 ** 32 - 2, 2 ** 31, 2 ** 31 - 1, 2 ** 31 + 1]
    # both sides of every table boundary in a few binades, and of the binade boundaries themselves
    for E in (31, 30, 20, 9):
        for i in (0, 1, 105, 106, 107, 254, 255):
            base = (1 << E) + (i << (E - 8)) if E >= 8 else (1 << E)
            ks += [base - 1, base, base + 1]
    ks += [int(v) for v in rng.integers(0, 2 ** 32, 3000, dtype=np.uint64)]
    ks += [int(v) for v in 2 ** 32 - 1 - rng.integers(0, 2 ** 12, 200, dtype=np.uint64)]   # u -> 1: no cancellation
    worst = 0.0
    for k in ks:
        k = min(max(k, 0), 2 ** 32 - 1)
        a = O.bm_neg2log(k)
        ex = _exact(k)
        ulp = math.ulp(float(ex))
        err = abs(float(Decimal(a) - ex)) / ulp
        worst = max(worst, err)
        assert 0.0 < a < 64.0
    assert worst < 1.5, worst      # measured ~0.9 ulp
    assert O.bm_neg2log(2 ** 32 - 1) == float(_exact(2 ** 32 - 1))   # the u -> 1 end is exact to the last bit
