"""Pin the C restatement (oracle/oracle_c.c) to the reference's golden vectors (float64 domain)."""
import numpy as np
import pytest

from conftest import MANIFEST, assert_hist_equal
from oracle import oracle_c

F64_DOMAIN = [n for n in sorted(MANIFEST["hotpath"]) if n not in ("i64_datetime_like", "big_int64_vs_int_edges")]


@pytest.mark.parametrize("name", F64_DOMAIN)
def test_c_oracle_matches_reference(golden, name):
    samples, edges, w, want = golden.hotpath_case(name)
    if samples[0].shape[1] == 0:
        pytest.skip("empty")
    got = oracle_c.bincount_rows(samples, edges, w)
    assert_hist_equal(got, want, weighted=w is not None)
