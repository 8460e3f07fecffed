"""Parity edges of the group-by path, through the C ABI:
  - mixed-sign f64 sums with groups that nearly cancel, for every strategy, at the bound the header states
    (|err| <= 1e-12 * sum|x| of the group: the reference's left fold and these re-associated sums are both
    within that of the exact sum, aggregations/mod.rs:622-651 for the fold being matched);
  - a wide key tuple (more than 64 key bits) holding a NULLABLE column that spans all 2^64 values
    (the reference hashes rows and compares tuples: hashing.rs:334-378, null is a value of its own);
  - the C ABI called from four host threads at once (the reference is called from a rayon pool: lib.rs:49-62);
  - the result-buffer pool is keyed by device, and a thread's workspace can be given back."""
import threading

import numpy as np
import pytest

import polars_amd as pa
from polars_amd import _ffi
from polars_amd.column import Column
from tests import util
from tests.test_parity_gpu import STRATEGIES, check_parity, rand_col, run_product

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _reset_strategy():
    yield
    pa.set_strategy("auto")


def cancelling_values(rng, kk, n):
    """Values in +-1e6; every second group is made of (x, -x) pairs so that its sum nearly cancels."""
    x = rng.uniform(-1e6, 1e6, n)
    order = np.argsort(kk, kind="stable")
    ks, xs = kk[order], x[order]
    starts = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
    for gi, (a, b) in enumerate(zip(starts, np.r_[starts[1:], n])):
        if gi % 2 == 0 and b - a >= 2:
            h = (b - a) // 2
            xs[a + h:a + 2 * h] = -xs[a:a + h]
    x[order] = xs
    return x


@pytest.mark.parametrize("strategy", STRATEGIES)
@pytest.mark.parametrize("card", [300, 40_000])
def test_mixed_sign_f64_sums_every_strategy(oracle, strategy, card):
    rng = np.random.default_rng(17 + card)
    n = 400_000
    kk = rng.integers(-card // 2, card - card // 2, n).astype(np.int32)
    x = cancelling_values(rng, kk, n)
    vm = rng.random(n) >= 0.05
    key, v = Column("k", kk), Column("v", x, vm)
    aggs = [(0, "sum"), (0, "mean"), (0, "count")]
    want, _ = oracle.group_by_agg([key], [v], aggs)
    got = run_product([key], [v], aggs, False, strategy)
    g, w = util.frame_to_np(got), util.oracle_to_np(want)
    og, ow = util.canonical_order(g[:1]), util.canonical_order(w[:1])
    assert np.array_equal(g[0][0][og], w[0][0][ow]), f"{strategy}: keys differ"
    assert np.array_equal(g[3][0][og], w[3][0][ow]), f"{strategy}: counts differ"
    keys_sorted = g[0][0][og].astype(np.int64) + card // 2
    mag = np.zeros(card)
    np.add.at(mag, kk.astype(np.int64)[vm] + card // 2, np.abs(x[vm]))
    nvalid = np.bincount(kk.astype(np.int64)[vm] + card // 2, minlength=card)
    for j, scale in ((1, mag[keys_sorted]), (2, mag[keys_sorted] / np.maximum(nvalid[keys_sorted], 1))):
        gv, wv = g[j][0][og], w[j][0][ow]
        gm = np.ones(len(gv), bool) if g[j][1] is None else g[j][1][og]
        wm = np.ones(len(wv), bool) if w[j][1] is None else w[j][1][ow]
        assert np.array_equal(gm, wm), f"{strategy}: validity of column {j} differs"
        err = np.abs(gv - wv)[gm]
        assert np.all(err <= 1e-12 * scale[gm] + 1e-300), \
            f"{strategy} ({pa.last_strategy()}): column {j}: max err / sum|x| = {np.max(err / np.maximum(scale[gm], 1e-300)):.3e}"


@pytest.mark.parametrize("strategy", ["auto", "sort", "global"])
@pytest.mark.parametrize("other", ["u64_full", "two_i64"])
def test_wide_tuple_with_nullable_full_range_key(oracle, strategy, other):
    """The tuple needs more than 64 bits AND one member covers every 64-bit value plus null: no spare code for the
    null inside that member's bit range.  It is encoded alone (null -> the reserved slot), then packed with the rest."""
    rng = np.random.default_rng(23)
    n = 60_000
    pool = np.array([-(2 ** 63), 2 ** 63 - 1, -1, 0, 1, 12345, -(2 ** 62)], dtype=np.int64)
    pool = np.r_[pool, rng.integers(-(2 ** 63), 2 ** 63 - 1, 200, dtype=np.int64)]
    a = pool[rng.integers(0, len(pool), n)]
    am = rng.random(n) >= 0.07
    keys = [Column("a", a, am)]
    if other == "u64_full":
        upool = np.array([0, 2 ** 64 - 1, 2 ** 63, 7], dtype=np.uint64)
        keys.append(Column("b", upool[rng.integers(0, len(upool), n)], rng.random(n) >= 0.1))
    else:
        keys.append(rand_col(rng, "b", np.int64, n, card=5))
        keys.append(rand_col(rng, "c", np.int64, n, card=4, null_frac=0.2))
    values = [rand_col(rng, "v", np.int64, n, lo=-1000, hi=1000), rand_col(rng, "w", np.float64, n, null_frac=0.1)]
    aggs = [(0, "sum"), (0, "min"), (0, "max"), (1, "mean"), (0, "count")]
    check_parity(oracle, keys, values, aggs, strategy, what=f"wide tuple, nullable full-range member ({other})")
    # the same tuple with that member first / last in the key list (the packing order is the library's choice)
    check_parity(oracle, keys[::-1], values, aggs, strategy, what=f"wide tuple reversed ({other})")


def _thread_queries():
    """Four different queries (strategy, columns, aggs): each thread owns one."""
    rng = np.random.default_rng(99)
    out = []
    n = 300_000
    out.append(("auto", [rand_col(rng, "k", np.int32, n, card=100)], [rand_col(rng, "v", np.int32, n, lo=1, hi=5)], [(0, "sum"), (0, "count")]))
    out.append(("tilesort", [rand_col(rng, "k", np.int64, n, card=50_000, null_frac=0.02)],
                [rand_col(rng, "v", np.float64, n, null_frac=0.05)], [(0, "sum"), (0, "mean"), (0, "max")]))
    out.append(("auto", [rand_col(rng, "a", np.uint32, n, lo=1, hi=100), rand_col(rng, "b", np.int16, n, card=30)],
                [rand_col(rng, "v", np.int64, n, lo=-50, hi=50)], [(0, "sum"), (0, "min"), (0, "count")]))
    out.append(("sort", [rand_col(rng, "k", np.uint64, n, card=200_000)], [rand_col(rng, "v", np.uint32, n)], [(0, "sum"), (0, "max")]))
    return out


def test_c_abi_from_four_host_threads(oracle):
    """Every thread has its own context (stream, workspace, strategy, last error): results must not depend on what the
    other threads run at the same time.  ctypes releases the GIL for the duration of a call, so the calls do overlap."""
    queries = _thread_queries()
    wants = [util.oracle_to_np(oracle.group_by_agg(k, v, a)[0]) for _, k, v, a in queries]
    errors: list = []
    start = threading.Barrier(len(queries))

    def worker(i):
        try:
            strategy, keys, values, aggs = queries[i]
            start.wait()
            for it in range(6):
                got = run_product(keys, values, aggs, False, strategy)
                util.assert_frames_match(util.frame_to_np(got), wants[i], len(keys), what=f"thread {i} iteration {it}")
            # errors are per thread as well: a failing call here must not disturb the others
            from polars_amd.frame import group_by_agg_raw
            with pytest.raises(_ffi.PlHipError):  # key / value length mismatch: raised inside the library (group_by/mod.rs:59-73)
                group_by_agg_raw([Column("a", np.zeros(4, np.int32))], [Column("b", np.zeros(5 + i, np.int32))], [(0, _ffi.AGG_NAMES["sum"])])
            assert _ffi.lib().plhip_last_error() != b""
            got = run_product(keys, values, aggs, False, strategy)
            util.assert_frames_match(util.frame_to_np(got), wants[i], len(keys), what=f"thread {i} after an error")
            _ffi.check(_ffi.lib().plhip_release_workspace())  # the thread ends: give the workspace back first
        except BaseException as e:  # noqa: BLE001 - reported by the main thread
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(queries))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a worker thread hangs"
    assert not errors, errors
    # the main thread's context is untouched by the workers' lives and deaths
    strategy, keys, values, aggs = queries[0]
    got = run_product(keys, values, aggs, False, strategy)
    util.assert_frames_match(util.frame_to_np(got), wants[0], len(keys), what="main thread afterwards")


def test_release_workspace_then_query_again(oracle):
    rng = np.random.default_rng(4)
    n = 500_000
    keys = [rand_col(rng, "k", np.int64, n, card=120_000)]
    values = [rand_col(rng, "v", np.float64, n)]
    aggs = [(0, "sum"), (0, "count")]
    for strategy in ("auto", "tilesort2", "auto"):
        check_parity(oracle, keys, values, aggs, strategy, what=f"before/after release ({strategy})")
        _ffi.check(_ffi.lib().plhip_release_workspace())
        assert _ffi.lib().plhip_pooled_buffers(0) == 0


def test_large_host_results_live_in_recycled_pinned_buffers(oracle):
    """Host-resident results of 32 KiB and more come from the page-locked pool (api.hip HostPool): same frames as ever, across
    reuse of the buffers by results of other sizes, a frame kept alive while later queries run, plhip_release_workspace in between,
    and with the pool switched off (PLHIP_NO_PINNED_RESULTS=1, a fresh process)."""
    import os
    import subprocess
    import sys

    rng = np.random.default_rng(12)
    n = 1_500_000
    k = rand_col(rng, "k", np.int64, n, card=400_000)  # ~4e5 groups x 8 B = 3 MB per column
    v = rand_col(rng, "v", np.float64, n)
    df = pa.DataFrame({"k": k, "v": v})
    want, _ = oracle.group_by_agg([k], [v], [(0, "sum"), (0, "count")])
    want = util.oracle_to_np(want)
    kept = []
    for rep in range(3):
        out = df.group_by("k").agg([pa.col("v").sum().alias("s"), pa.col("v").count().alias("c")])
        util.assert_frames_match(util.frame_to_np(out), want, 1, what=f"pass {rep}")
        kept.append(out)  # (the frame aliases the library's buffers: it must stay intact while later results are made)
        small = pa.DataFrame({"k": Column("k", k.to_numpy()[:300_000] % 7), "v": Column("v", v.to_numpy()[:300_000])}).group_by("k").agg(pa.col("v").count())
        assert small.height == 7
        if rep == 1:
            _ffi.check(_ffi.lib().plhip_release_workspace())
    for rep, out in enumerate(kept):
        util.assert_frames_match(util.frame_to_np(out), want, 1, what=f"kept frame {rep}")
    # the CSR in host memory: 6 MB of row indices from the same pool
    g = pa.group_tuples([k], True)
    g_o = oracle.Groups([k], True)
    first, offsets, indices = g.buffers()
    assert np.array_equal(first, g_o.first) and np.array_equal(offsets, g_o.offsets) and np.array_equal(indices, g_o.indices)
    g.close()
    del kept, out
    code = ("import numpy as np, polars_amd as pa\n"
            "rng = np.random.default_rng(1); n = 1_000_000\n"
            "k = rng.integers(0, 300_000, n); v = rng.integers(0, 9, n)\n"
            "out = pa.DataFrame({'k': k, 'v': v}).group_by('k').agg(pa.col('v').sum())\n"
            "u, inv = np.unique(k, return_inverse=True); s = np.bincount(inv, weights=v).astype(np.int64)\n"
            "o = np.argsort(out['k'].to_numpy()); assert np.array_equal(out['k'].to_numpy()[o], u) and np.array_equal(out['v'].to_numpy()[o], s)\n"
            "print('ok')")
    env = dict(os.environ, PLHIP_NO_PINNED_RESULTS="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_result_pool_is_keyed_by_device():
    """A freed device-resident frame goes to the pool of ITS device; another device's pool never sees it."""
    import torch

    from polars_amd.frame import group_by_agg_raw

    L = _ffi.lib()
    _ffi.check(L.plhip_release_workspace())
    rng = np.random.default_rng(8)
    n = 100_000
    k = Column("k", rng.integers(0, 1000, n).astype(np.int32))
    v = Column("v", rng.integers(0, 9, n).astype(np.int64))
    handle, _ = group_by_agg_raw([k], [v], [(0, _ffi.AGG_NAMES["sum"])], _ffi.FLAG_OUTPUT_DEVICE)
    before0, before1 = L.plhip_pooled_buffers(0), L.plhip_pooled_buffers(1)
    L.plhip_result_free(handle)
    assert L.plhip_pooled_buffers(0) > before0, "the frame's buffers did not return to device 0's pool"
    assert L.plhip_pooled_buffers(1) == before1 == 0
    if torch.cuda.device_count() < 2:
        return
    # two GPUs: a thread bound to device 1 must get memory of device 1 (a device-0 buffer would fault or cross xGMI)
    res: list = []

    def on_dev1():
        try:
            _ffi.check(L.plhip_init(1))
            kt = torch.randint(0, 1000, (n,), device="cuda:1", dtype=torch.int32)
            vt = torch.ones(n, device="cuda:1", dtype=torch.int64)
            h, cols = group_by_agg_raw([Column("k", kt)], [Column("v", vt)], [(0, _ffi.AGG_NAMES["sum"])], _ffi.FLAG_OUTPUT_DEVICE)
            res.append(int(L.plhip_result_num_rows(h)))
            res.append(int(L.plhip_pooled_buffers(0)))  # device 0's pool was not drawn from
            L.plhip_result_free(h)
            res.append(int(L.plhip_pooled_buffers(1)))
        except BaseException as e:  # noqa: BLE001
            res.append(repr(e))

    in_pool0 = L.plhip_pooled_buffers(0)
    t = threading.Thread(target=on_dev1)
    t.start()
    t.join(timeout=120)
    assert len(res) == 3 and res[0] == 1000 and res[1] == in_pool0 and res[2] > 0, res


@pytest.mark.parametrize("case", ["all_ones_u32", "all_null", "i32_extremes", "u32_compact_high", "nullable_compact", "f32_key"])
def test_key_range_comes_back_through_one_pinned_word(oracle, case):
    """One key column of at most 32 bits: its min / max reach the host as ONE pinned 64-bit word (min << 32 | max) that starts
    as all ones.  Edges: the answer IS all ones (every key 0xFFFFFFFF: the stream going idle settles it); no valid row; the
    dtype's extremes in one column; a compact range high up (AUTO then takes the dense path with a bias); a nullable compact
    range; an f32 key (bit patterns).  Sizes above 2^20 rows so that AUTO measures the range."""
    rng = np.random.default_rng(5)
    n = (1 << 20) + 12_345
    km = None
    if case == "all_ones_u32":
        k = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
    elif case == "all_null":
        k, km = rng.integers(0, 9, n).astype(np.int32), np.zeros(n, bool)
    elif case == "i32_extremes":
        k = rng.choice(np.array([-2**31, 2**31 - 1, 0, -1, 7], dtype=np.int32), n)
    elif case == "u32_compact_high":
        k = (0xFFFF0000 + rng.integers(0, 40_000, n)).astype(np.uint32)
    elif case == "nullable_compact":
        k, km = rng.integers(-20_000, 20_000, n).astype(np.int32), rng.random(n) >= 0.03
    else:
        k = rng.choice(np.array([0.0, -0.0, 1.5, -1.5, np.inf, np.nan], dtype=np.float32), n)
    keys = [Column("k", k, km)]
    values = [Column("v", rng.integers(-100, 100, n).astype(np.int64)), Column("w", np.round(rng.uniform(0, 1, n), 6))]
    aggs = [(0, "sum"), (1, "mean"), (0, "min"), (0, "count")]
    for strategy in ("auto", "sort"):  # (the sort strategy measures a nullable single key too)
        check_parity(oracle, keys, values, aggs, strategy, what=f"pinned-word key range, {case}, {strategy}")
    pa.set_strategy("auto")
    check_parity(oracle, keys, values, aggs, "auto", what=f"pinned-word key range, {case}, again")
    if case in ("u32_compact_high", "nullable_compact"):
        # (the i64 sum and the min above have no straight-line dense kind: AUTO leaves such queries to the hashed pipelines, with
        # the measured range as their hint; q5's shape -- 32-bit integer sums, f64 sums -- takes the dense path with the bias)
        assert pa.last_strategy() != "dense", pa.last_strategy()
        q5 = [Column("u", rng.integers(0, 100, n).astype(np.int32)), values[1]]
        check_parity(oracle, keys, q5, [(0, "sum"), (1, "sum")], "auto", what=f"pinned-word key range, {case}, dense kinds")
        assert pa.last_strategy() == "dense", pa.last_strategy()
