"""bench.py certifies every timed leg against `torch_reference` — an independent restatement in torch ops.  That checker is
itself held to the pinned oracle here (CPU tensors; no GPU, no library), edge cases included, so that "verified" in the
driver's line means what it says."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import oracle_np as onp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _edges(rng, k):
    e = np.sort(rng.uniform(-4, 4, k))
    e[0], e[-1] = -4.0, 4.0
    return e


@pytest.mark.parametrize("case", ["c2", "c2u", "c3", "c4", "c5"])
def test_torch_reference_equals_the_oracle_on_every_bench_shape(case):
    rng = np.random.default_rng(5)
    n = 20_011
    specials = np.array([np.nan, -np.inf, np.inf, -0.0, 0.0, 4.0, -4.0, 3.9999999999999996, 4.000000000000001, -4.000000000000001])
    if case in ("c2", "c2u"):
        xs, edges, rows = [np.concatenate([rng.standard_normal(n) * 2, specials])], [np.linspace(-4, 4, 101)], 1
    elif case == "c3":
        e = [_edges(rng, 257), _edges(rng, 257)]
        xs = [np.concatenate([rng.standard_normal(n) * 2, specials, e[0][:10]]), np.concatenate([rng.standard_normal(n) * 2, specials[::-1], e[1][-10:]])]
        edges, rows = e, 1
    elif case == "c4":
        rows = 7
        xs, edges = [(rng.standard_normal((rows, 3001)) * 2).astype(np.float32)], [np.linspace(-4, 4, 51)]
        xs[0][3, :5] = [4.0, -4.0, np.nan, np.inf, 3.9999998]
    else:
        xs = [np.concatenate([rng.standard_normal(n) * 2, specials]), np.concatenate([rng.standard_normal(n) * 2, specials[::-1]])]
        edges, rows = [np.linspace(-4, 4, 1025)] * 2, 1
    weighted = case in ("c2", "c5")
    cols = xs[0].size // rows
    w = rng.uniform(0, 1, xs[0].shape) if weighted else None
    got = bench.torch_reference(torch, [torch.as_tensor(x) for x in xs], torch.as_tensor(w) if weighted else None, edges, rows, cols, weighted, chunk=4096)
    want = onp.bincount_rows([x.reshape(rows, cols) for x in xs], edges, None if w is None else w.reshape(rows, cols))
    want = np.asarray(want).reshape(rows, -1)
    if weighted:
        np.testing.assert_allclose(got.numpy(), want, rtol=1e-12, atol=0)
    else:
        np.testing.assert_array_equal(got.numpy(), want)
    # a prefix of the columns (the strong-scaling leg reads n / N samples of the same buffers)
    half = cols // 2
    got_h = bench.torch_reference(torch, [torch.as_tensor(x) for x in xs], torch.as_tensor(w) if weighted else None, edges, rows, half, weighted)
    want_h = np.asarray(onp.bincount_rows([x.reshape(rows, cols)[:, :half] for x in xs], edges, None if w is None else w.reshape(rows, cols)[:, :half])).reshape(rows, -1)
    np.testing.assert_allclose(got_h.numpy(), want_h, rtol=1e-12, atol=0)


def test_compare_with_reference_flags_what_it_must():
    ref = torch.tensor([[0, 5, 7]])
    assert bench.compare_with_reference(torch, torch.tensor([[0, 5, 7]]), ref)["ok"]
    assert not bench.compare_with_reference(torch, torch.tensor([[0, 5, 8]]), ref)["ok"]
    reff = torch.tensor([[0.0, 5.0, 7.0]], dtype=torch.float64)
    assert bench.compare_with_reference(torch, reff * (1 + 5e-7), reff)["ok"]
    assert not bench.compare_with_reference(torch, reff * (1 + 5e-6), reff)["ok"]
    assert not bench.compare_with_reference(torch, torch.tensor([[1e-30, 5.0, 7.0]], dtype=torch.float64), reff)["ok"]  # an empty bin must stay empty
    assert not bench.compare_with_reference(torch, torch.tensor([[0.0, float("nan"), 7.0]], dtype=torch.float64), reff)["ok"]
