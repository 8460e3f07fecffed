"""Arrow ingest at the boundary: an IPC / Parquet file read with the chunks left as the reader delivered them
(polars_amd.io), grouped through plhip_group_by_agg_chunked -- the reference reads the same files into ChunkedArrays and rechunks
on the host first (executors/group_by.rs:66).  The result must be the in-memory query's: checked against the oracle on the
concatenated columns.  pyarrow writes the files (several record batches / row groups, nulls, sliced arrays, a dictionary-encoded
string column, a plain string column, a Boolean column)."""
import numpy as np
import pytest

import polars_amd as pa
from polars_amd import _ffi
from polars_amd.column import Column
from tests import util

pytestmark = pytest.mark.gpu


def _table(n, seed, batches):
    import pyarrow as pyar

    rng = np.random.default_rng(seed)
    k = rng.integers(-40, 40, n).astype(np.int32)
    km = rng.random(n) >= 0.05
    k2 = rng.integers(0, 7, n).astype(np.int64)
    v = rng.integers(-1000, 1000, n).astype(np.int64)
    w = np.round(rng.uniform(0, 100, n), 6)
    wm = rng.random(n) >= 0.1
    b = rng.random(n) < 0.3
    cats = np.array([f"id{i:03d}" for i in range(50)])
    s = cats[rng.integers(0, 50, n)]
    arrays = {
        "k": pyar.array(k, mask=~km), "k2": pyar.array(k2), "v": pyar.array(v), "w": pyar.array(w, mask=~wm), "b": pyar.array(b),
        "cat": pyar.array(s).dictionary_encode(), "s": pyar.array(s),
    }
    t = pyar.table(arrays)
    # several batches of uneven sizes
    cuts = sorted(set(rng.integers(1, n - 1, batches - 1).tolist()))
    bounds = [0] + cuts + [n]
    parts = [t.slice(a, z - a) for a, z in zip(bounds[:-1], bounds[1:])]
    host = {"k": (k, km), "k2": (k2, None), "v": (v, None), "w": (w, wm), "b": (b, None), "s": s}
    return t, parts, host


def _check(oracle, df, host, what):
    out = df.group_by("k", "k2").agg(pa.col("v").sum().alias("vs"), pa.col("w").mean().alias("wm"), pa.col("w").min().alias("wlo"),
                                     pa.col("v").max().alias("vhi"), pa.count().alias("n"), pa.col("b").sum().alias("bs"))
    keys = [Column("k", *host["k"]), Column("k2", host["k2"][0])]
    vals = [Column("v", host["v"][0]), Column("w", *host["w"]), Column("b", host["b"][0].astype(np.uint32))]
    aggs = [(0, "sum"), (1, "mean"), (1, "min"), (0, "max"), (0, "count"), (2, "sum")]
    want, _ = oracle.group_by_agg(keys, vals, aggs)
    got = [(out[c].to_numpy(), out[c].valid_mask()) for c in out.columns]
    util.assert_frames_match(got, util.oracle_to_np(want), 2, what=what)


def test_read_ipc_keeps_the_record_batches_as_chunks(oracle, tmp_path):
    import pyarrow as pyar
    import pyarrow.ipc as ipc

    t, parts, host = _table(300_000, 3, 7)
    path = tmp_path / "g.arrow"
    with ipc.new_file(str(path), t.schema) as w:
        for p in parts:
            for batch in p.to_batches():
                w.write_batch(batch)
    df = pa.read_ipc(path)
    assert isinstance(df["k"], pa.ChunkedColumn) and df["k"].n_chunks >= 7 and df.height == 300_000
    assert df["k"].chunks[0].location == _ffi.HOST
    _check(oracle, df, host, "IPC file, memory-mapped, 7 record batches")
    _check(oracle, pa.read_ipc(path, memory_map=False, columns=["k", "k2", "v", "w", "b"]), host, "IPC file, read into memory, projected")
    # a dictionary-encoded string key (a Categorical on the reference side): the codes take the dense path, the result shows the strings
    out = df.group_by("cat").agg(pa.col("v").sum().alias("vs"), pa.count().alias("n"))
    assert pa.last_strategy() in ("dense", "direct"), pa.last_strategy()
    want = {}
    for name, val in zip(host["s"].tolist(), host["v"][0].tolist()):
        a = want.setdefault(name, [0, 0])
        a[0] += val; a[1] += 1
    got = {k: [s_, n_] for k, s_, n_ in zip(out["cat"].to_list(), out["vs"].to_list(), out["n"].to_list())}
    assert got == want
    # a plain string key goes through the string encoder (one LargeUtf8 chunk)
    out = df.group_by("s").agg(pa.col("v").sum().alias("vs"), pa.count().alias("n"))
    got = {k: [s_, n_] for k, s_, n_ in zip(out["s"].to_list(), out["vs"].to_list(), out["n"].to_list())}
    assert got == want


def test_read_parquet_row_groups_and_sliced_arrays(oracle, tmp_path):
    import pyarrow as pyar
    import pyarrow.parquet as pq

    t, parts, host = _table(200_000, 5, 4)
    path = tmp_path / "g.parquet"
    pq.write_table(t, str(path), row_group_size=37_000)
    df = pa.read_parquet(path)
    assert df.height == 200_000
    _check(oracle, df, host, "Parquet file, row groups of 37 000 rows")
    # chunks that are SLICES of larger arrays (Arrow offsets into values and validity, not multiples of 8)
    sl = pyar.concat_tables([p.slice(3, max(len(p) - 5, 0)) for p in parts])
    lens = [len(q) for q in parts]
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    keep = np.concatenate([np.arange(st + 3, st + 3 + max(ln - 5, 0)) for st, ln in zip(starts.tolist(), lens)])
    h2 = {k_: (v_[0][keep], None if v_[1] is None else v_[1][keep]) if isinstance(v_, tuple) else v_[keep] for k_, v_ in host.items()}
    df2 = pa.from_arrow(sl)
    assert any(c.offset % 8 for c in df2["k"].chunks)
    _check(oracle, df2, h2, "sliced chunks")
    # the two-step seams take a reader's column too (rechunked for that call)
    g = df2.group_by("k", "k2").groups()
    assert g.n_rows == df2.height


def test_scan_reads_only_the_columns_the_query_names(oracle, tmp_path):
    """pl.scan_ipc / pl.scan_parquet + group_by + agg + collect: the lazy query touches the key and value columns it names and
    nothing else (projection pushdown for this query shape); slice pushdown rides along."""
    import pyarrow.ipc as ipc
    import pyarrow.parquet as pq

    t, parts, host = _table(120_000, 9, 3)
    p_ipc, p_pq = tmp_path / "s.arrow", tmp_path / "s.parquet"
    with ipc.new_file(str(p_ipc), t.schema) as w:
        for part in parts:
            for batch in part.to_batches():
                w.write_batch(batch)
    pq.write_table(t, str(p_pq), row_group_size=50_000)
    keys = [Column("k2", host["k2"][0])]
    vals = [Column("w", *host["w"])]
    want, _ = oracle.group_by_agg(keys, vals, [(0, "mean"), (0, "count")])
    for lf in (pa.scan_ipc(p_ipc), pa.scan_parquet(p_pq)):
        out = lf.group_by("k2").agg(pa.col("w").mean().alias("m"), pa.count().alias("n")).collect()
        assert sorted(lf._df.loaded) == ["k2", "w"], lf._df.loaded
        got = [(out[c].to_numpy(), out[c].valid_mask()) for c in out.columns]
        util.assert_frames_match(got, util.oracle_to_np(want), 1, what="scan + group_by + agg")
        head = lf.group_by("k2", maintain_order=True).agg(pa.col("w").mean().alias("m")).head(3).collect()
        assert head.height == 3
    assert pa.scan_parquet(p_pq).collect().height == 120_000


def test_read_csv_as_the_harness_starts(oracle, tmp_path):
    """run_h2oai_benchmark.py:21-35: read_csv with dtype overrides, id1 .. id3 cast to Categorical, then the queries (q1, q3, q6 here) --
    on a small table of the generator's shape with NAs (empty fields), written as CSV."""
    rng = np.random.default_rng(8)
    n, K = 60_000, 20
    ids = {c: rng.integers(1, K + 1, n) for c in ("id1", "id2", "id4", "id5")}
    ids["id3"], ids["id6"] = rng.integers(1, n // K + 1, n), rng.integers(1, n // K + 1, n)
    v1, v2, v3 = rng.integers(1, 6, n), rng.integers(1, 16, n), np.round(rng.uniform(0, 100, n), 6)
    na = {c: rng.random(n) < 0.05 for c in ("id1", "id3", "id4", "v1", "v3")}
    path = tmp_path / "G1.csv"
    with open(path, "w") as f:
        f.write("id1,id2,id3,id4,id5,id6,v1,v2,v3\n")
        cell = lambda c, i, txt: "" if c in na and na[c][i] else txt
        for i in range(n):
            f.write(",".join([cell("id1", i, f"id{ids['id1'][i]:03d}"), f"id{ids['id2'][i]:03d}", cell("id3", i, f"id{ids['id3'][i]:010d}"),
                              cell("id4", i, str(ids["id4"][i])), str(ids["id5"][i]), str(ids["id6"][i]),
                              cell("v1", i, str(v1[i])), str(v2[i]), cell("v3", i, repr(float(v3[i])))]) + "\n")
    df = pa.read_csv(path, dtypes={"id4": np.int32, "id5": np.int32, "id6": np.int32, "v1": np.int32, "v2": np.int32, "v3": np.float64},
                     categorical=["id1", "id2", "id3"])
    assert df.height == n and df["id1"].dictionary is not None and df["v1"].dtype == _ffi.I32
    x = df.lazy()
    col = lambda name, arr, dt: Column(name, arr.astype(dt), None if name not in na else ~na[name])
    # q1: sum(v1) by id1 -- the result shows the categories, the null key is one group
    out = x.group_by("id1").agg(pa.col("v1").sum().alias("v1_sum")).collect()
    want, _ = oracle.group_by_agg([col("id1", ids["id1"], np.int32)], [col("v1", v1, np.int32)], [(0, "sum")])
    got = dict(zip(out["id1"].to_list(), out["v1_sum"].to_list()))
    exp = {(f"id{int(k):03d}" if ok else None): int(s) for k, ok, s in zip(want[0][1], want[0][2], want[1][1])}
    assert got == exp
    # q3: sum(v1), mean(v3) by id3
    out = x.group_by("id3").agg([pa.col("v1").sum().alias("v1_sum"), pa.col("v3").mean().alias("v3_mean")]).collect()
    want, _ = oracle.group_by_agg([col("id3", ids["id3"], np.int64)], [col("v1", v1, np.int32), col("v3", v3, np.float64)], [(0, "sum"), (1, "mean")])
    exp = {(f"id{int(k):010d}" if ok else None): (int(s), (float(m) if mok else None))
           for k, ok, s, m, mok in zip(want[0][1], want[0][2], want[1][1], want[2][1], want[2][2])}
    got = dict(zip(out["id3"].to_list(), zip(out["v1_sum"].to_list(), out["v3_mean"].to_list())))
    assert got.keys() == exp.keys()
    for k_, (s_, m_) in got.items():
        assert s_ == exp[k_][0] and ((m_ is None) == (exp[k_][1] is None)) and (m_ is None or abs(m_ - exp[k_][1]) <= 1e-12 * abs(exp[k_][1])), k_
    # q6: median(v3), std(v3) by id4, id5 -- over the CSR seams
    out = x.group_by(["id4", "id5"], maintain_order=True).agg([pa.col("v3").median().alias("v3_median"), pa.col("v3").std().alias("v3_std")]).collect()
    g = oracle.Groups([col("id4", ids["id4"], np.int32), col("id5", ids["id5"], np.int32)], True)
    for name, kind in (("v3_median", "median"), ("v3_std", "std")):
        wv, wm = oracle.agg_ext(col("v3", v3, np.float64), kind, g, 1)
        gv, gm = out[name].to_numpy(), out[name].valid_mask()
        gm = np.ones(len(gv), bool) if gm is None else gm
        assert gm.tolist() == wm.tolist() and np.allclose(gv[gm], wv[wm], rtol=1e-11, atol=0), kind
