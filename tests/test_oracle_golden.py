"""Pin the CPU oracle (oracle/oracle_np.py) to the reference's own outputs (tests/golden) and to
numpy's histogram family, which is the oracle of the reference's own tests (test_core.py:25-228)."""
import numpy as np
import pytest

from conftest import MANIFEST, assert_hist_equal
from oracle import oracle_np as onp


@pytest.mark.parametrize("name", sorted(MANIFEST["hotpath"]))
def test_hotpath_matches_reference(golden, name):
    samples, edges, w, want = golden.hotpath_case(name)
    got = onp.bincount_rows(samples, edges, w)
    assert got.dtype == want.dtype
    # the restatement uses the same numpy primitives in the same order: exact, weighted too
    np.testing.assert_array_equal(got, want)


SMALL = [n for n in sorted(MANIFEST["hotpath"]) if n not in ("wide_bins_1k", "f64_uniform_1row")]


@pytest.mark.parametrize("name", SMALL)
def test_definitional_matches_reference(golden, name):
    samples, edges, w, want = golden.hotpath_case(name)
    if samples[0].size > 5000:
        samples = [s[:, :300] for s in samples]
        w = None if w is None else w[:, :300]
        want = onp.bincount_rows(samples, edges, w)
    got = onp.bincount_rows_definitional(samples, edges, w)
    assert_hist_equal(got, want, weighted=w is not None)


@pytest.mark.parametrize("name", sorted(MANIFEST["core"]))
def test_public_api_matches_reference(golden, name):
    args, kw, want, meta = golden.core_case(name)
    got, edges = onp.histogram(*args, **kw)
    assert got.shape == tuple(meta["h_shape"])
    assert str(got.dtype) == meta["h_dtype"]
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=0, equal_nan=True)
    for i, e in enumerate(edges):
        np.testing.assert_array_equal(e, golden.core["%s/edges%d" % (name, i)])


@pytest.mark.parametrize("name", sorted(MANIFEST["dask_cases"]))
def test_dask_cases_equal_unchunked(golden, name):
    """The reference's dask branch (blockwise + sum, core.py:429-439) gives the unchunked result."""
    args, kw, want, meta = golden.core_case(name, "dask_cases")
    got, _ = onp.histogram(*args, **kw)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=0, equal_nan=True)


# ---- the reference's own known-answer / relational tests, restated against the oracle -------
def test_right_edge_known_answer():  # test_core.py:95-113
    data = np.ones((5, 20))
    bins = np.array([0, 0.5, 1])
    h, _ = onp.histogram(data, bins=bins, axis=1)
    np.testing.assert_array_equal(h.sum(axis=0), [0, 100])
    np.testing.assert_array_equal(onp.histogram(data, bins=bins)[0], np.histogram(data, bins=bins)[0])


def test_specials_known_answer():  # 12-element vector -> [3 1 3] for edges [0,1,2,4]
    x = np.array([[np.nan, -np.inf, np.inf, -0.0, 0.0, 1.0, 2.0, 4.0, 3.9999999999999996, 4.000000000000001, -1e-300, 0.5]])
    got = onp.bincount_rows([x], [np.array([0.0, 1.0, 2.0, 4.0])])
    np.testing.assert_array_equal(got, [[3, 1, 3]])


@pytest.mark.parametrize("density", [False, True])
@pytest.mark.parametrize("add_nans", [False, True])
def test_vs_numpy_1d(density, add_nans):  # test_core.py:25-69
    rng = np.random.default_rng(2)
    data = rng.standard_normal((5, 20))
    if add_nans:
        data.ravel()[rng.choice(data.size, 20, replace=False)] = np.nan
    bins = np.linspace(-4, 4, 10)
    h, _ = onp.histogram(data, bins=bins, axis=1, density=density)
    want = np.stack([np.histogram(data[i], bins=bins, density=density)[0] for i in range(5)])
    np.testing.assert_allclose(h, want)


def test_vs_numpy_dd():  # test_core.py:116-228
    rng = np.random.default_rng(3)
    a, b, c = (rng.standard_normal((5, 20)) for _ in range(3))
    ba, bb, bc = np.linspace(-4, 4, 10), np.linspace(-4, 4, 11), np.linspace(-4, 4, 10)
    h, _ = onp.histogram(a, b, bins=[ba, bb])
    np.testing.assert_array_equal(h, np.histogram2d(a.ravel(), b.ravel(), bins=[ba, bb])[0])
    h, _ = onp.histogram(a, b, c, bins=[ba, bb, bc], density=True)
    want = np.histogramdd((a.ravel(), b.ravel(), c.ravel()), bins=[ba, bb, bc], density=True)[0]
    np.testing.assert_allclose(h, want)
