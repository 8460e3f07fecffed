"""Shared helpers of the parity tests: fixture loading, column building, canonical comparison."""
from __future__ import annotations

import json
import math
from pathlib import Path

import numpy as np

from polars_amd.column import Column

GOLDEN = Path(__file__).resolve().parent / "golden"
NP = {"i8": np.int8, "i16": np.int16, "i32": np.int32, "i64": np.int64, "u8": np.uint8, "u16": np.uint16,
      "u32": np.uint32, "u64": np.uint64, "f32": np.float32, "f64": np.float64}


def load_known_answers():
    return json.loads((GOLDEN / "reference_known_answers.json").read_text())


def expand(data):
    if isinstance(data, dict):
        return list(data["repeat"]) * int(data["times"])
    return data


def make_column(spec) -> Column:
    data = expand(spec["data"])
    mask = np.array([x is not None for x in data], dtype=bool)
    vals = np.array([0 if x is None else x for x in data], dtype=NP[spec["dtype"]])
    return Column(spec["name"], vals, None if mask.all() else mask)


def to_py(values, valid):
    """numpy values + bool valid -> python list with None for nulls."""
    out = values.tolist()
    return [v if ok else None for v, ok in zip(out, valid)]


def _sort_key(x):
    # nulls first, NaN last, then by value
    if x is None:
        return (0, 0)
    if isinstance(x, float) and math.isnan(x):
        return (2, 0)
    return (1, x)


def canonical_rows(columns: list[list], n_keys: int) -> list[tuple]:
    """Rows sorted by the key tuple (the 'canonical sort on the keys' of the north star)."""
    rows = list(zip(*columns)) if columns else []
    return sorted(rows, key=lambda r: tuple(_sort_key(v) for v in r[:n_keys]))


def values_equal(a, b, rel=0.0) -> bool:
    if a is None or b is None:
        return a is None and b is None
    if isinstance(a, float) or isinstance(b, float):
        a, b = float(a), float(b)
        if math.isnan(a) or math.isnan(b):
            return math.isnan(a) and math.isnan(b)
        if a == b:
            return True
        return rel > 0 and abs(a - b) <= rel * max(abs(a), abs(b))
    return a == b


def assert_rows_equal(got: list[tuple], want: list[tuple], rel=0.0, what=""):
    assert len(got) == len(want), f"{what}: {len(got)} rows, expected {len(want)}"
    for i, (g, w) in enumerate(zip(got, want)):
        assert len(g) == len(w), f"{what}: row {i} width {len(g)} != {len(w)}"
        for j, (x, y) in enumerate(zip(g, w)):
            assert values_equal(x, y, rel), f"{what}: row {i} col {j}: got {x!r}, expected {y!r}"


def check_known_answer(case, columns: list[list], dtypes: list[str] | None = None, rel=0.0):
    """columns = key columns then agg columns, python lists with None for nulls, in engine output order."""
    n_keys = len(case["keys"])
    mode = case["compare"]
    if mode == "n_groups":
        assert len(columns[0]) == case["expected_n_groups"], case["name"]
        return
    if dtypes is not None:
        want_dt = case["expected_dtypes"]
        got_dt = dtypes if mode != "sorted_column" else dtypes[n_keys:]
        assert got_dt == want_dt, f"{case['name']}: dtypes {got_dt} != {want_dt}"
    exp = case["expected"]
    if mode == "ordered":
        assert_rows_equal(list(zip(*columns)), list(zip(*exp)), rel, case["name"])
    elif mode == "sorted_keys":
        assert_rows_equal(canonical_rows(columns, n_keys), canonical_rows(exp, n_keys), rel, case["name"])
    elif mode == "sorted_column":
        got = sorted(columns[n_keys])
        assert_rows_equal([(x,) for x in got], [(x,) for x in exp[0]], rel, case["name"])
    else:
        raise AssertionError(mode)


# ---------------------------------------------------------------------------------------------------
# numpy-level comparison for large frames
# ---------------------------------------------------------------------------------------------------
def _bits(v: np.ndarray) -> np.ndarray:
    if v.dtype.kind == "f":
        return v.view(np.uint32 if v.dtype.itemsize == 4 else np.uint64)
    return v


def canonical_order(key_cols) -> np.ndarray:
    """Permutation that sorts rows by the key tuple (nulls first, floats by bit pattern)."""
    ks = []
    for v, m in reversed(list(key_cols)):
        b = _bits(v)
        if m is not None:
            b = np.where(m, b, 0)
            ks.append(b)
            ks.append(m.astype(np.uint8))
        else:
            ks.append(b)
    return np.lexsort(ks)


def assert_frames_match(got, want, n_keys: int, rtol_f64=1e-12, rtol_f32=1e-5, what=""):
    """got / want: lists of (values ndarray, valid mask or None).  Rows are compared after the canonical
    sort on the keys.  Integers, counts, keys and validity must be bit-exact; floats within rtol."""
    assert len(got) == len(want), f"{what}: {len(got)} columns vs {len(want)}"
    ng, nw = len(got[0][0]), len(want[0][0])
    assert ng == nw, f"{what}: {ng} groups, expected {nw}"
    og, ow = canonical_order(got[:n_keys]), canonical_order(want[:n_keys])
    for j, ((gv, gm), (wv, wm)) in enumerate(zip(got, want)):
        assert gv.dtype == wv.dtype, f"{what}: column {j} dtype {gv.dtype} != {wv.dtype}"
        gv, wv = gv[og], wv[ow]
        gm = np.ones(ng, bool) if gm is None else gm[og]
        wm = np.ones(nw, bool) if wm is None else wm[ow]
        assert np.array_equal(gm, wm), f"{what}: column {j} validity differs at {np.flatnonzero(gm != wm)[:5]}"
        a, b = gv[gm], wv[wm]
        if gv.dtype.kind == "f" and j >= n_keys:
            rtol = rtol_f32 if gv.dtype.itemsize == 4 else rtol_f64
            nan_a, nan_b = np.isnan(a), np.isnan(b)
            assert np.array_equal(nan_a, nan_b), f"{what}: column {j} NaN pattern differs"
            a, b = a[~nan_a].astype(np.float64), b[~nan_b].astype(np.float64)
            ok = (a == b) | (np.abs(a - b) <= rtol * np.maximum(np.abs(a), np.abs(b)))
            assert ok.all(), (f"{what}: column {j}: {np.count_nonzero(~ok)} values beyond rtol={rtol}; first "
                              f"{a[~ok][:3]} vs {b[~ok][:3]}")
        else:
            assert np.array_equal(_bits(a), _bits(b)), (
                f"{what}: column {j} differs at {np.flatnonzero(_bits(a) != _bits(b))[:5]}: "
                f"{a[_bits(a) != _bits(b)][:3]} vs {b[_bits(a) != _bits(b)][:3]}")


def frame_to_np(df):
    """polars_amd.DataFrame -> list of (values, valid mask or None)."""
    return [(df[c].to_numpy(), df[c].valid_mask()) for c in df.columns]


def oracle_to_np(out):
    return [(v, None if m.all() else m) for _, v, m in out]


# test_quantile (crates/polars-core/src/chunked_array/ops/aggregate/mod.rs:789-944): exact expectations
QUANTILE_KNOWN = [
    ([2, 1, None, 3, 5, None, 4], {
        "nearest": {0.1: 1.0, 0.9: 5.0, 0.6: 4.0}, "lower": {0.1: 1.0, 0.9: 4.0, 0.6: 3.0}, "higher": {0.1: 2.0, 0.9: 5.0, 0.6: 4.0},
        "midpoint": {0.1: 1.5, 0.9: 4.5, 0.6: 3.5}, "linear": {0.1: 1.4, 0.9: 4.6}}),
    ([None, 7, 6, 2, 1, None, 3, 5, None, 4], {
        "nearest": {0.1: 1.0, 0.9: 7.0, 0.6: 5.0}, "lower": {0.1: 1.0, 0.9: 6.0, 0.6: 4.0}, "higher": {0.1: 2.0, 0.9: 7.0, 0.6: 5.0},
        "midpoint": {0.1: 1.5, 0.9: 6.5, 0.6: 4.5}, "linear": {0.1: 1.6, 0.9: 6.4, 0.6: 4.6}}),
]


def quantile_known_column(data):
    from polars_amd.column import Column
    return Column("a", np.array([0 if x is None else x for x in data], dtype=np.uint32), np.array([x is not None for x in data]))
