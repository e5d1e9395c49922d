"""The device-side build of the rating orders (pmf_index.hip: stable radix sort of positions by
row id) against the host counting sort it replaces: same within-row order, hence bit-identical
sweeps (fp32 sums are order-sensitive, so equality of the results pins the order inside every
row to the input order the reference keeps; hpf_cavi.py:97-107)."""
import os

import numpy as np
import pytest

from helpers import skewed_problem

pytestmark = pytest.mark.gpu


def _fit_state(u, i, x, U, I, K, dtype, host):
    import pmf_hip
    from pmf_hip import ARR_BIAS, ARR_COV, ARR_FACTOR, ARR_RATE, ARR_SHAPE, ITEM, USER
    old = os.environ.pop("PMF_INDEX_HOST", None)
    if host:
        os.environ["PMF_INDEX_HOST"] = "1"
    try:
        out = []
        rng = np.random.default_rng(5)
        c = pmf_hip.Context(U, I, K, dtype=dtype)
        c.set_ratings(u, i, x)
        c.set_array(USER, ARR_FACTOR, rng.gamma(1.0, 0.5, (U, K)) + 0.05)
        c.set_array(ITEM, ARR_FACTOR, rng.gamma(1.0, 0.5, (I, K)) + 0.05)
        for _ in range(2):
            c.gamma_sweep(USER, 0.3, 0.3)
            c.gamma_sweep(ITEM, 0.3, 0.3)
        out += [c.get_array(s, a) for s in (USER, ITEM) for a in (ARR_FACTOR, ARR_SHAPE, ARR_RATE)]
        c.close()
        g = pmf_hip.Context(U, I, K, dtype=dtype)
        g.set_ratings(u, i, x - 3.0)
        g.set_array(USER, ARR_FACTOR, 0.1 * rng.standard_normal((U, K)))
        g.set_array(ITEM, ARR_FACTOR, 0.1 * rng.standard_normal((I, K)))
        g.set_cov_identity(USER); g.set_cov_identity(ITEM)
        g.set_array(USER, ARR_BIAS, np.zeros(U)); g.set_array(ITEM, ARR_BIAS, np.zeros(I))
        for _ in range(2):
            for s in (USER, ITEM):
                g.gauss_factor_sweep(s, 0.4, 0.8)
            for s in (USER, ITEM):
                g.gauss_bias_sweep(s, 0.4, 1.0)
        out += [g.get_array(s, a) for s in (USER, ITEM) for a in (ARR_FACTOR, ARR_COV, ARR_BIAS)]
        g.close()
        return out
    finally:
        os.environ.pop("PMF_INDEX_HOST", None)
        if old is not None:
            os.environ["PMF_INDEX_HOST"] = old


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("shape", [(2500, 130, 70000, 12), (37, 5000, 9000, 8), (1, 1, 3, 4), (300, 300, 0, 4)])
def test_device_index_build_equals_host_build(shape, dtype):
    U, I, N, K = shape
    if N >= 1000:
        u, i, x = skewed_problem(9, U, I, N)            # skewed, duplicates, rows without ratings
    else:
        u, i = np.zeros(N, np.int64), np.zeros(N, np.int64)
        x = np.arange(1, N + 1, dtype=np.float64)
    dev = _fit_state(u, i, x, U, I, K, dtype, host=False)
    host = _fit_state(u, i, x, U, I, K, dtype, host=True)
    for a, b in zip(dev, host):
        assert np.array_equal(a, b)

