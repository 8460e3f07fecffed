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
