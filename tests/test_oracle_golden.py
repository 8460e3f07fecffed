"""Pins the CPU oracle against the reference's own known-answer tests (tests/golden/reference_known_answers.json)."""
import numpy as np
import pytest

from polars_amd import _ffi
from tests import util

KA = util.load_known_answers()


@pytest.mark.parametrize("case", KA["hash"], ids=lambda c: c["name"])
def test_hash_known_answers(oracle, case):
    keys = [util.make_column(k) for k in case["keys"]]
    got = oracle.vec_hash(keys, case["null_hash"])
    assert got.tolist() == case["expected"]


def test_folded_multiply_matches_python_bigint(oracle):
    M = 6364136223846793005
    rng = np.random.default_rng(1)
    for v in [0, 1, 1234, 5678, 2**63, 2**64 - 1] + rng.integers(0, 2**63, 50).tolist():
        r = v * M
        assert oracle.folded_multiply(v, M) == ((r & (2**64 - 1)) ^ (r >> 64))


@pytest.mark.parametrize("threads", [1, 8])
@pytest.mark.parametrize("partitioned", [0, 4])
@pytest.mark.parametrize("case", KA["group_by"], ids=lambda c: c["name"])
def test_group_by_known_answers(oracle, case, threads, partitioned):
    keys = [util.make_column(k) for k in case["keys"]]
    values = [util.make_column(v) for v in case["values"]]
    aggs = [(ci, kind) for ci, kind in case["aggs"]]
    out, _ = oracle.group_by_agg(keys, values, aggs, maintain_order=case["maintain_order"], n_threads=threads,
                                 partitioned=partitioned)
    cols = [util.to_py(v, m) for _, v, m in out]
    dts = [_ffi.DTYPE_NAMES[_ffi.np_dtype_code(v.dtype)] for _, v, _ in out]
    util.check_known_answer(case, cols, dts)
