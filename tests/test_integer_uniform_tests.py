"""The exchange kernels test their candidate uniforms in integers (ptm_device_math.hpp: u01_below_bound, u01_times).  The two
identities behind that, checked against the floating-point forms the CPU checker keeps (oracle/ptm_oracle.c: u < thresh with
u = (k + 0.5) / 2^32, (int)(u * (Nt - 1))) -- at random and at every boundary a threshold or a rung count has."""
import numpy as np


def u01(k):
    return (k.astype(np.float64) + 0.5) / 4294967296.0


def below_bound(t):
    T = t * 4294967296.0 - 0.5
    if not T > 0.0:
        return 0
    if T >= 4294967296.0:
        return 4294967296
    return int(np.ceil(T))


def test_threshold_test_in_integers_is_the_floating_point_test():
    rng = np.random.default_rng(5)
    ts = np.concatenate([rng.uniform(0, 1.2, 400), [0.0, 1.0, 0.5, 2.0 ** -32, 2.0 ** -33, 1 - 2.0 ** -33, 1 - 2.0 ** -32, 0.49755859375, 1e-300, -0.3]])
    for t in ts:
        kt = below_bound(float(t))
        near = np.clip(np.arange(kt - 4, kt + 5), 0, 2 ** 32 - 1).astype(np.uint64)
        k = np.concatenate([rng.integers(0, 2 ** 32, 2000, dtype=np.uint64), near, np.array([0, 1, 2 ** 31, 2 ** 32 - 2, 2 ** 32 - 1], dtype=np.uint64)])
        assert np.array_equal(u01(k) < t, k < np.uint64(kt) if kt < 2 ** 64 else np.ones(k.shape, bool)), t


def test_rung_pick_in_integers_is_the_floating_point_pick():
    rng = np.random.default_rng(6)
    for m in [1, 2, 3, 7, 19, 63, 127, 1023, 4095, 65534] + list(rng.integers(1, 65535, 40)):
        m = int(m)
        # every k where the pick changes value: (2k + 1) m crosses a multiple of 2^33
        edges = (np.arange(1, min(m, 3000) + 1, dtype=np.float64) * 2.0 ** 33 / m - 1.0) / 2.0
        near = np.clip(np.concatenate([np.floor(edges) + d for d in (-1, 0, 1, 2)]), 0, 2 ** 32 - 1).astype(np.uint64)
        k = np.concatenate([rng.integers(0, 2 ** 32, 5000, dtype=np.uint64), near, np.array([0, 2 ** 32 - 1], dtype=np.uint64)])
        want = (u01(k) * float(m)).astype(np.int64)            # (int)(u * (Nt - 1)): truncation of a non-negative number
        got = ((2 * k.astype(object) + 1) * m) >> 33           # exact integers
        assert np.array_equal(want, np.array(got, dtype=np.int64)), m
        assert want.max() <= m - 1 or m == 1
