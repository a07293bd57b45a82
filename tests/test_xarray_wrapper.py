"""Label handling of xhistogram_amd.xarray.histogram, restating the reference's test_xarray.py
known answers (dims, coords = bin centres, name, keep_coords, weights broadcast, errors).

CPU tests swap the compute for the oracle (the wrapper does no arithmetic besides bin centres);
the gpu-marked test runs the same wrapper over the HIP path.  Uses the real xarray if present,
else the small double in tests/doubles/."""
import importlib
import os
import sys

import numpy as np
import pytest

try:
    import xarray as xr  # noqa: F401
except ImportError:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "doubles"))
    import xarray as xr  # the double

from oracle import oracle_np as onp

xhx = importlib.import_module("xhistogram_amd.xarray")


def _two_passes(*args, weights=None, **kw):
    ha, edges = onp.histogram(*args, weights=weights[0], **kw)
    return ha, onp.histogram(*args, weights=weights[1], **kw)[0], edges


@pytest.fixture
def cpu_compute(monkeypatch):
    monkeypatch.setattr(xhx, "_core_histogram", onp.histogram)
    monkeypatch.setattr(xhx, "_core_histogram_two_weights", _two_passes)


def _ones(dims, shape, name="T", with_coords=True):
    coords = {d: np.arange(n) * 10.0 for d, n in zip(dims, shape)} if with_coords else None
    return xr.DataArray(np.ones(shape), dims=dims, coords=coords, name=name, attrs={"units": "K"})


@pytest.mark.parametrize("ndims", [1, 2, 3, 4])
def test_ones_every_dim_combination(cpu_compute, ndims):  # test_xarray.py:38-67
    from itertools import combinations

    dims = ["x", "y", "z", "t"][:ndims]
    shape = (3, 4, 5, 6)[:ndims]
    da = _ones(dims, shape)
    bins = np.array([0.0, 0.9, 1.1, 2.0])
    centres = 0.5 * (bins[:-1] + bins[1:])
    h = xhx.histogram(da, bins=[bins])
    assert h.dims == ("T_bin",) and h.name == "histogram_T"
    np.testing.assert_array_equal(h.values, [0, da.values.size, 0])
    np.testing.assert_array_equal(h["T_bin"].values, centres)
    assert h["T_bin"].attrs == {"units": "K"}
    for k in range(1, ndims + 1):
        for red in combinations(dims, k):
            h = xhx.histogram(da, bins=[bins], dim=red)
            kept = [d for d in dims if d not in red]
            assert list(h.dims) == kept + ["T_bin"]
            n_red = int(np.prod([s for d, s in zip(dims, shape) if d in red]))
            want = np.zeros([s for d, s in zip(dims, shape) if d not in red] + [3])
            want[..., 1] = n_red
            np.testing.assert_array_equal(h.values, want)
            for d in kept:  # dimension coordinates of kept dims are carried over
                np.testing.assert_array_equal(h[d].values, da[d].values)


@pytest.mark.parametrize("ndims", [1, 2, 3, 4])
def test_ones_density_integrates_to_one_over_every_dim_combination(cpu_compute, ndims):  # test_xarray.py:70-94
    """density=True: at every kept location the pdf integrates to 1 over the bins — for every combination of reduced dims
    (the reference checks (h * bin_area).sum("ones_bin") == 1 with everything in the middle bin of width 0.2)"""
    from itertools import combinations

    dims = ["x", "y", "z", "t"]
    shape = (3, 4, 5, 6)
    da = _ones(dims, shape, name="ones")
    bins = np.array([0.0, 0.9, 1.1, 2.0])
    widths = np.diff(bins)
    for red in combinations(dims, ndims):
        h = xhx.histogram(da, bins=[bins], dim=red, density=True)
        other = [d for d in dims if d not in red]
        assert set(other) <= set(h.dims) and h.dims[-1] == "ones_bin" and h.name == "histogram_ones"
        vals = h.values
        assert vals.shape == tuple(s for d, s in zip(dims, shape) if d not in red) + (3,)
        np.testing.assert_allclose((vals * widths).sum(axis=-1), 1.0)           # the integral over the bins, everywhere
        np.testing.assert_allclose(vals[..., 1] * 0.2, 1.0)                     # the reference's own form: all mass in the middle bin
        np.testing.assert_array_equal(vals[..., 0], 0.0)
        for d in other:
            np.testing.assert_array_equal(h[d].values, da[d].values)
    # weighted density: the same integral (core.py:444-462 normalises by the in-range weight sum)
    w = xr.DataArray(np.arange(1.0, 5.0), dims=["y"], name="w")
    h = xhx.histogram(da, bins=[bins], dim=["y", "t"], weights=w, density=True)
    np.testing.assert_allclose((h.values * widths).sum(axis=-1), 1.0)


def test_weights_of_every_sub_dimensionality(cpu_compute):  # test_xarray.py:99-135
    da = _ones(["x", "y", "z"], (3, 4, 5))
    bins = np.array([0.0, 0.9, 1.1, 2.0])
    for wdims, wshape in ((["x"], (3,)), (["y", "z"], (4, 5)), (["z", "x"], (5, 3)), (["x", "y", "z"], (3, 4, 5))):
        w = xr.DataArray(0.5 * np.ones(wshape), dims=wdims, name="w")
        h = xhx.histogram(da, bins=[bins], weights=w)
        np.testing.assert_allclose(h.values, [0, 0.5 * 60, 0])
        h = xhx.histogram(da, bins=[bins], weights=w, dim=["y", "z"])
        assert list(h.dims) == ["x", "T_bin"]
        np.testing.assert_allclose(h.values[:, 1], 0.5 * 20)


def test_two_args_dims_order_and_name(cpu_compute):  # test_xarray.py:139-173 (issue #5)
    rng = np.random.default_rng(0)
    a = xr.DataArray(rng.standard_normal((4, 5, 6)), dims=["t", "y", "x"], name="a")
    b = xr.DataArray(rng.standard_normal((5, 6)), dims=["y", "x"], name="b")  # broadcast over t
    ba, bb = np.linspace(-3, 3, 7), np.linspace(-3, 3, 5)
    h = xhx.histogram(a, b, bins=[ba, bb], dim=["y", "x"])
    assert list(h.dims) == ["t", "a_bin", "b_bin"] and h.name == "histogram_a_b"
    want = np.stack([np.histogram2d(a.values[i].ravel(), b.values.ravel(), bins=[ba, bb])[0] for i in range(4)])
    np.testing.assert_array_equal(h.values, want)
    h2 = xhx.histogram(a, b, bins=[ba, bb], bin_dim_suffix="_edges")
    assert list(h2.dims) == ["a_edges", "b_edges"]


def test_keep_coords(cpu_compute):  # test_xarray.py:176-211
    da = xr.DataArray(np.ones((3, 4)), dims=["x", "y"], name="T",
                      coords={"x": np.arange(3.0), "y": np.arange(4.0), "lon": (("x",), np.array([7.0, 8.0, 9.0])),
                              "area": (("x", "y"), np.ones((3, 4)))})
    bins = np.array([0.0, 2.0])
    h = xhx.histogram(da, bins=[bins], dim=["y"])
    assert "lon" not in h.coords and "area" not in h.coords
    h = xhx.histogram(da, bins=[bins], dim=["y"], keep_coords=True)
    assert "lon" in h.coords and "area" not in h.coords  # area spans the reduced dim
    np.testing.assert_array_equal(h["lon"].values, [7.0, 8.0, 9.0])


def test_errors(cpu_compute):  # test_xarray.py:215-218, xarray.py:116-117, 126
    with pytest.raises(TypeError):
        xhx.histogram(np.ones(3), bins=[np.linspace(0, 1, 3)])
    with pytest.raises(AssertionError):
        xhx.histogram(xr.DataArray(np.ones(3), dims=["x"]), bins=[np.linspace(0, 1, 3)])
    a = xr.DataArray(np.ones(3), dims=["x"], coords={"x": [0.0, 1.0, 2.0]}, name="a")
    b = xr.DataArray(np.ones(3), dims=["x"], coords={"x": [0.0, 1.0, 5.0]}, name="b")
    with pytest.raises(ValueError):
        xhx.histogram(a, b, bins=[np.linspace(0, 2, 3)] * 2)


def test_pair_of_weights_gives_pair_of_histograms(cpu_compute):  # the TODO at xarray.py:106
    rng = np.random.default_rng(8)
    t = xr.DataArray(rng.standard_normal((5, 6, 7)), dims=["time", "lat", "lon"], name="T")
    area = xr.DataArray(rng.uniform(1, 2, (6, 7)), dims=["lat", "lon"], name="area")
    salt = xr.DataArray(rng.uniform(30, 40, (5, 6, 7)), dims=["time", "lat", "lon"], name="S")
    bins = np.linspace(-3, 3, 13)
    flux = xr.DataArray(salt.values * area.values, dims=["time", "lat", "lon"], name="flux")
    num, den = xhx.histogram(t, bins=[bins], dim=["lat", "lon"], weights=(flux, area))
    assert list(num.dims) == list(den.dims) == ["time", "T_bin"] and num.name == den.name == "histogram_T"
    np.testing.assert_allclose(num.values, xhx.histogram(t, bins=[bins], dim=["lat", "lon"], weights=flux).values, rtol=1e-12)
    np.testing.assert_allclose(den.values, xhx.histogram(t, bins=[bins], dim=["lat", "lon"], weights=area).values, rtol=1e-12)
    with pytest.raises(ValueError):
        xhx.histogram(t, bins=[bins], weights=(area, area), density=True)
    with pytest.raises(ValueError):
        xhx.histogram(t, bins=[bins], weights=(area,))


@pytest.mark.gpu
def test_wrapper_over_hip_path():
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(3)
    t = rng.standard_normal((6, 40, 50)).astype(np.float32)
    bins = np.linspace(-4, 4, 51)
    da = xr.DataArray(t, dims=["time", "lat", "lon"], name="T", coords={"time": np.arange(6.0)})
    h = xhx.histogram(da, bins=[bins], dim=["lat", "lon"])
    assert list(h.dims) == ["time", "T_bin"]
    np.testing.assert_array_equal(h.values, onp.histogram(t, bins=bins, axis=(1, 2))[0])
    dg = xr.DataArray(torch.as_tensor(t).cuda(), dims=["time", "lat", "lon"], name="T")  # GPU-resident data
    hg = xhx.histogram(dg, bins=[bins], dim=["lat", "lon"])
    np.testing.assert_array_equal(hg.values, h.values)
    w = xr.DataArray(rng.uniform(0, 1, (40, 50)), dims=["lat", "lon"], name="w")
    tw = xr.DataArray(t * w.values, dims=["time", "lat", "lon"], name="Tw")
    num, den = xhx.histogram(da, bins=[bins], dim=["lat", "lon"], weights=(tw, w))  # one pass, two weights
    np.testing.assert_allclose(num.values, onp.histogram(t, bins=bins, axis=(1, 2), weights=t * w.values)[0], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(den.values, onp.histogram(t, bins=bins, axis=(1, 2), weights=np.broadcast_to(w.values, t.shape))[0], rtol=1e-6)


@pytest.mark.gpu
def test_dims_and_coords_over_hip_path():  # test_xarray.py:139-173 (issue #5), the HIP path under the wrapper — not the patched-in oracle
    rng = np.random.default_rng(5)
    time_axis, depth_axis, x_axis, y_axis = np.arange(4), np.arange(10), np.arange(30), np.arange(30)
    dat1 = rng.integers(0, 100, size=(4, 10, 30, 30))
    dat2 = rng.integers(0, 50, size=(4, 10, 30, 30))
    coords = {"time": time_axis, "depth": depth_axis, "X": x_axis, "Y": y_axis}
    one = xr.DataArray(dat1, dims=["time", "depth", "X", "Y"], coords=coords, name="one")
    two = xr.DataArray(dat2, dims=["time", "depth", "X", "Y"], coords=coords, name="two")
    bins1, bins2 = np.linspace(0, 100, 50), np.linspace(0, 50, 25)
    h = xhx.histogram(one, two, dim=["X", "Y"], bins=[bins1, bins2])
    assert tuple(h.dims) == ("time", "depth", "one_bin", "two_bin") and h.name == "histogram_one_two"
    np.testing.assert_array_equal(h["time"].values, time_axis)
    np.testing.assert_array_equal(h["depth"].values, depth_axis)
    np.testing.assert_allclose(h["one_bin"].values, 0.5 * (bins1[:-1] + bins1[1:]))
    np.testing.assert_allclose(h["two_bin"].values, 0.5 * (bins2[:-1] + bins2[1:]))
    want = onp.histogram(dat1, dat2, bins=[bins1, bins2], axis=(2, 3))[0]
    np.testing.assert_array_equal(h.values, want)
    assert h.values.sum() == dat1.size - np.count_nonzero(dat1 > 100)  # (integers on [0, 100): every pair is counted once)


@pytest.mark.gpu
@pytest.mark.parametrize("number_of_inputs", [1, 2])
@pytest.mark.parametrize("keep_coords", [True, False])
@pytest.mark.parametrize("include_weights", [True, False])
def test_carry_coords_over_hip_path(keep_coords, number_of_inputs, include_weights):  # test_xarray.py:176-211
    rng = np.random.default_rng(6)
    time_axis, x_axis, y_axis = np.arange(40), np.arange(10), np.arange(10)
    data = rng.integers(0, 100, size=(40, 10, 10))
    lon = x_axis[:, None] ** 2 + y_axis[None, :] ** 2  # "faking coordinates": a two-dimensional non-index coordinate
    da = xr.DataArray(data, dims=["time", "X", "Y"], name="one",
                      coords={"time": time_axis, "X": x_axis, "Y": y_axis, "lon": (("X", "Y"), lon)})
    assert "lon" in da.coords
    weights = xr.DataArray(np.full(data.shape, 0.5), dims=["time", "X", "Y"], name="w") if include_weights else None
    bins = np.linspace(0, 100, 10)
    h = xhx.histogram(*[da] * number_of_inputs, bins=[bins] * number_of_inputs, dim=["time"], weights=weights, keep_coords=keep_coords)
    assert ("lon" in h.coords) == keep_coords
    if keep_coords:
        np.testing.assert_array_equal(h["lon"].values, lon)
    assert tuple(h.dims) == ("X", "Y") + ("one_bin",) * number_of_inputs
    want = onp.histogram(*[data] * number_of_inputs, bins=[bins] * number_of_inputs, axis=0,
                         weights=None if weights is None else weights.values)[0]
    if include_weights:
        np.testing.assert_allclose(h.values, want, rtol=1e-6)
    else:
        np.testing.assert_array_equal(h.values, want)
