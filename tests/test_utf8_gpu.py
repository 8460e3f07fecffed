"""String keys: plhip_utf8_encode against the oracle's restatement of IntoGroupsProxy for Utf8Chunked
(crates/polars-core/src/frame/group_by/into_groups.rs:232-290; BytesHash equality, vector_hasher.rs:422-446), and the group-by
over the codes against the oracle's group-by over its own codes.  The two number their groups differently (the oracle by
first occurrence, the device in table order), so everything is compared through the strings."""
import numpy as np
import pytest

import polars_amd as pa
from polars_amd import _ffi
from polars_amd.column import Column
from polars_amd.strings import Utf8Column, encode
from tests import util

pytestmark = pytest.mark.gpu


def make_strings(rng, n, n_distinct, null_frac, kind):
    ids = rng.integers(0, n_distinct, n)
    if kind == "db-benchmark":  # id1 .. id3 of groupby-datagen.R: "id" + zero-padded number
        pool = [f"id{v:010d}" for v in range(n_distinct)]
    elif kind == "ragged":  # lengths 0 .. 40, shared prefixes, multi-byte characters, an empty string
        pool = [("" if v == 0 else ("prefix-" * (v % 5)) + "é" * (v % 3) + str(v * 7919)) for v in range(n_distinct)]
    else:  # "long": strings beyond several 8-byte words that differ only in their last byte
        pool = [("x" * 61) + chr(33 + v % 90) + ("y" * (v // 90)) for v in range(n_distinct)]
    vals = [pool[i] for i in ids]
    if null_frac:
        for i in np.flatnonzero(rng.random(n) < null_frac):
            vals[i] = None
    return vals


@pytest.mark.parametrize("kind", ["db-benchmark", "ragged", "long"])
@pytest.mark.parametrize("n,n_distinct,null_frac", [(1, 1, 0.0), (1000, 7, 0.1), (200_000, 5_000, 0.02), (300_000, 290_000, 0.0)])
def test_utf8_encode_matches_the_oracle(oracle, kind, n, n_distinct, null_frac):
    rng = np.random.default_rng(n + n_distinct)
    vals = make_strings(rng, n, n_distinct, null_frac, kind)
    col = Utf8Column.from_list("s", vals)
    codes = encode(col)
    got_codes, got_valid = codes.to_numpy(), codes.valid_mask()
    d = codes.dictionary.strings()
    want_codes, want_first = oracle.utf8_group_codes(col.offsets, col.data, col.valid_mask())
    assert len(d) == len(want_first), (len(d), len(want_first))
    # the dictionary holds every group's string once (the null group's entry is None)
    assert sorted(x for x in d if x is not None) == sorted(vals[f] for f in want_first if vals[f] is not None)
    assert sum(x is None for x in d) == (1 if any(v is None for v in vals) else 0)
    # every row decodes to itself; null rows keep the column's validity
    if got_valid is None:
        assert all(v is not None for v in vals)
    else:
        assert got_valid.tolist() == [v is not None for v in vals]
    dec = [d[c] if (got_valid is None or got_valid[i]) else None for i, c in enumerate(got_codes)]
    assert dec == vals
    # same partition of the rows as the reference's groups: a bijection between the two numberings
    pairs = set(zip(got_codes[:n].tolist() if got_valid is None else np.where(got_valid, got_codes, 2**32 - 1).tolist(), want_codes.tolist()))
    assert len(pairs) == len(want_first)


def test_utf8_slice_and_empty(oracle):
    vals = ["b", "a", None, "b", "", "a", "ccc", None, ""]
    full = Utf8Column.from_list("s", vals)
    part = Utf8Column("s", full.offsets, full.data, full.validity, offset=2, length=6)  # an Arrow slice: offset into all three buffers
    codes = encode(part)
    d = codes.dictionary.strings()
    m = codes.valid_mask()
    assert [d[c] if m[i] else None for i, c in enumerate(codes.to_numpy())] == vals[2:8]
    assert sorted(x for x in d if x is not None) == ["", "a", "b", "ccc"] and d.count(None) == 1
    empty = encode(Utf8Column.from_list("s", []))
    assert empty.length == 0 and empty.dictionary.n_categories == 0 and empty.dictionary.strings() == []


@pytest.mark.parametrize("n,n_distinct", [(50_000, 100), (2_000_000, 100_000)])
def test_group_by_string_key_vs_oracle(oracle, n, n_distinct):
    """q1 / q3 of db-benchmark with the id column as strings (`id3` is a string column in the generated CSV): the frame API
    end to end, checked against the oracle's group-by over ITS codes, both sides keyed by the strings."""
    rng = np.random.default_rng(n)
    vals = make_strings(rng, n, n_distinct, 0.01, "db-benchmark")
    v1 = rng.integers(1, 6, n).astype(np.int32)
    v3 = np.round(rng.uniform(0, 100, n), 6)
    v3m = rng.random(n) >= 0.05
    df = pa.DataFrame({"id3": vals, "v1": v1, "v3": (v3, v3m)})
    out = df.group_by("id3").agg(pa.col("v1").sum(), pa.col("v3").mean(), pa.col("v3").max().alias("v3_max"), pa.count())
    if n_distinct > 4096:  # the codes are dense: a query whose state fits the dense-code path's ranges takes it
        df.group_by("id3").agg(pa.col("v1").sum(), pa.col("v3").mean())
        assert pa.last_strategy() == "dense"
    got = {k: row for k, *row in zip(*[out[c].to_list() for c in out.columns])}
    col = Utf8Column.from_list("id3", vals)
    ocodes, ofirst = oracle.utf8_group_codes(col.offsets, col.data, col.valid_mask())
    want, _ = oracle.group_by_agg([Column("c", ocodes)], [Column("v1", v1), Column("v3", v3, v3m)], [(0, "sum"), (1, "mean"), (1, "max"), (0, "count")])
    wcols = util.oracle_to_np(want)
    assert len(got) == len(wcols[0][0])
    for j, c in enumerate(wcols[0][0]):
        key = vals[ofirst[c]]
        g = got[key]
        assert g[0] == wcols[1][0][j] and g[3] == wcols[4][0][j], key
        for a in (1, 2):
            wv, wm = wcols[a + 1]
            if wm is not None and not wm[j]:
                assert g[a] is None, key
            else:
                assert g[a] is not None and abs(g[a] - wv[j]) <= 1e-12 * max(1.0, abs(wv[j])), (key, g[a], wv[j])


def test_group_by_two_keys_one_string(oracle):
    rng = np.random.default_rng(3)
    n = 100_000
    vals = make_strings(rng, n, 50, 0.05, "ragged")
    k2 = rng.integers(0, 4, n).astype(np.int64)
    v = rng.integers(-100, 100, n).astype(np.int64)
    df = pa.DataFrame({"s": vals, "k2": k2, "v": v})
    out = df.group_by("s", "k2").agg(pa.col("v").sum(), pa.count())
    got = {(a, b): (c, d) for a, b, c, d in zip(*[out[c].to_list() for c in out.columns])}
    want = {}
    for s, b, x in zip(vals, k2.tolist(), v.tolist()):
        t = want.setdefault((s, b), [0, 0])
        t[0] += x
        t[1] += 1
    assert got == {k: tuple(t) for k, t in want.items()}


def test_reference_docstring_answers_on_the_gpu():
    """py-polars/polars/dataframe/frame.py:4989-5036: the printed results of the reference's own group_by examples."""
    import json
    import os

    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_utf8_known_answers.json")))
    fr = gold["frame"]
    df = pa.DataFrame({"a": fr["a"], "b": np.array(fr["b"], np.int64), "c": np.array(fr["c"], np.int64)})
    for case in gold["cases"]:
        vcol, kind = case["agg"]
        out = df.group_by(*case["by"]).agg(getattr(pa.col(vcol), kind)())
        rows = sorted(list(r) for r in zip(*[out[c].to_list() for c in out.columns]))
        assert rows == sorted(case["rows"]), case
