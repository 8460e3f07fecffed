"""ctypes wrapper of the CPU oracle (oracle/groupby_oracle.c).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product (polars_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import math

import numpy as np

from polars_amd import _ffi
from polars_amd.column import Column

HERE = Path(__file__).resolve().parent
LIB = HERE / "_build" / "libgroupby_oracle.so"


class _Groups(C.Structure):
    _fields_ = [("n_groups", C.c_uint32), ("n_rows", C.c_uint32), ("first", C.POINTER(C.c_uint32)),
                ("offsets", C.POINTER(C.c_uint32)), ("indices", C.POINTER(C.c_uint32))]


_lib = None


def build() -> Path:
    subprocess.run(["make", "-C", str(HERE)], check=True, capture_output=True)
    return LIB


def lib():
    global _lib
    if _lib is None:
        if not LIB.exists():
            build()
        L = C.CDLL(str(LIB))
        P = C.POINTER
        L.orc_folded_multiply.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_folded_multiply.restype = C.c_uint64
        L.orc_vec_hash.argtypes = [P(_ffi.ColumnStruct), C.c_int, C.c_uint64, C.c_void_p]
        L.orc_vec_hash_seeded.argtypes = [P(_ffi.ColumnStruct), C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
        L.orc_group_tuples.argtypes = [P(_ffi.ColumnStruct), C.c_int, C.c_int, C.c_int, P(_Groups)]
        L.orc_groups_free.argtypes = [P(_Groups)]
        L.orc_groups_free.restype = None
        L.orc_agg_out_dtype.argtypes = [C.c_int, C.c_int]
        L.orc_agg.argtypes = [P(_ffi.ColumnStruct), C.c_int, P(_Groups), C.c_void_p, C.c_void_p, C.c_int]
        L.orc_agg_partitioned.argtypes = [P(_ffi.ColumnStruct), C.c_int, P(_Groups), C.c_int, C.c_void_p, C.c_void_p]
        L.orc_utf8_group_codes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
        L.orc_utf8_group_codes.restype = C.c_int64
        L.orc_hash_join.argtypes = [P(_ffi.ColumnStruct), P(_ffi.ColumnStruct), C.c_int, C.c_void_p, C.c_void_p]
        L.orc_hash_join.restype = C.c_int64
        L.orc_hash_join_multi.argtypes = [P(_ffi.ColumnStruct), P(_ffi.ColumnStruct), C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_hash_join_multi.restype = C.c_int64
        L.orc_take_keys.argtypes = [P(_ffi.ColumnStruct), P(_Groups), C.c_void_p, C.c_void_p]
        L.orc_sink_group_by.argtypes = [P(_ffi.ColumnStruct), P(_ffi.ColumnStruct), C.c_int, P(C.c_int), P(C.c_int), C.c_int, C.c_int,
                                        C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, P(C.c_int64)]
        _lib = L
    return _lib


def _structs(cols):
    arr = (_ffi.ColumnStruct * max(1, len(cols)))()
    for i, c in enumerate(cols):
        if c.location != _ffi.HOST:
            raise ValueError("the oracle reads host columns only")
        arr[i] = c.struct()
    return arr


def _check(rc: int):
    if rc != 0:
        raise _ffi._ERR_CLASS.get(rc, _ffi.PlHipError)(rc, "oracle")


def folded_multiply(s: int, by: int) -> int:
    return lib().orc_folded_multiply(s, by)


def vec_hash(keys: list[Column], null_h: int) -> np.ndarray:
    out = np.empty(len(keys[0]), dtype=np.uint64)
    _check(lib().orc_vec_hash(_structs(keys), len(keys), null_h, out.ctypes.data))
    return out


def vec_hash_seeded(keys: list[Column], null_h: int, true_h: int, false_h: int) -> np.ndarray:
    """series_to_hashes with Boolean columns (dtype BOOL: values = a packed bitmap): orc_vec_hash_seeded."""
    out = np.empty(len(keys[0]), dtype=np.uint64)
    _check(lib().orc_vec_hash_seeded(_structs(keys), len(keys), null_h, true_h, false_h, out.ctypes.data))
    return out


class Groups:
    """GroupsProxy::Idx as CSR numpy arrays (copies; the C buffers are freed)."""

    def __init__(self, keys: list[Column], sorted_: bool = False, n_threads: int = 1):
        g = _Groups()
        _check(lib().orc_group_tuples(_structs(keys), len(keys), int(sorted_), n_threads, C.byref(g)))
        G, n = g.n_groups, g.n_rows
        self.first = np.ctypeslib.as_array(g.first, shape=(max(G, 1),))[:G].copy()
        self.offsets = np.ctypeslib.as_array(g.offsets, shape=(G + 1,)).copy()
        self.indices = np.ctypeslib.as_array(g.indices, shape=(max(n, 1),))[:n].copy()
        lib().orc_groups_free(C.byref(g))
        self.n_groups, self.n_rows = G, n

    def _c(self) -> _Groups:
        g = _Groups()
        g.n_groups, g.n_rows = self.n_groups, self.n_rows
        g.first = self.first.ctypes.data_as(C.POINTER(C.c_uint32))
        g.offsets = self.offsets.ctypes.data_as(C.POINTER(C.c_uint32))
        g.indices = self.indices.ctypes.data_as(C.POINTER(C.c_uint32))
        return g

    def lists(self) -> list[list[int]]:
        return [self.indices[self.offsets[i]:self.offsets[i + 1]].tolist() for i in range(self.n_groups)]


def agg(values: Column | None, kind: str | int, groups: Groups, n_threads: int = 1, partitioned: int = 0):
    """Returns (values ndarray, valid bool ndarray) for one aggregation over `groups`."""
    k = _ffi.AGG_NAMES[kind] if isinstance(kind, str) else kind
    src_dt = _ffi.U32 if values is None else values.dtype
    out_dt = _ffi.code_np_dtype(lib().orc_agg_out_dtype(src_dt, k))
    out = np.zeros(groups.n_groups, dtype=out_dt)
    valid = np.zeros(groups.n_groups, dtype=np.uint8)
    g = groups._c()
    vs = _structs([values]) if values is not None else None
    if partitioned:
        _check(lib().orc_agg_partitioned(vs, k, C.byref(g), partitioned, out.ctypes.data, valid.ctypes.data))
    else:
        _check(lib().orc_agg(vs, k, C.byref(g), out.ctypes.data, valid.ctypes.data, n_threads))
    return out, valid.astype(bool)


def take_keys(key: Column, groups: Groups):
    out = np.zeros(groups.n_groups, dtype=_ffi.code_np_dtype(key.dtype))
    valid = np.zeros(groups.n_groups, dtype=np.uint8)
    g = groups._c()
    _check(lib().orc_take_keys(_structs([key]), C.byref(g), out.ctypes.data, valid.ctypes.data))
    return out, valid.astype(bool)


def group_by_agg(keys: list[Column], values: list[Column], aggs: list[tuple[int, str]], *,
                 maintain_order: bool = False, n_threads: int = 1, partitioned: int = 0):
    """group_by_helper (crates/polars-lazy/src/physical_plan/executors/group_by.rs:57-95) on the oracle:
    returns a list of (name, values, valid) = key columns then one column per agg, in group order."""
    groups = Groups(keys, maintain_order, n_threads)
    out = []
    for kc in keys:
        v, m = take_keys(kc, groups)
        out.append((kc.name, v, m))
    for ci, kind in aggs:
        col = None if kind == "count" else values[ci]
        v, m = agg(col, kind, groups, n_threads, partitioned)
        out.append((f"{values[ci].name if col is not None else 'count'}_{kind}", v, m))
    return out, groups


def sink_group_by(key: Column, values: list[Column], aggs: list[tuple[int, str]], n_threads: int = 1, morsel: int = 50_000,
                  max_groups: int | None = None, outputs: bool = True):
    """Path B, the streaming engine's PrimitiveGroupbySink (crates/polars-pipe/src/executors/sinks/group_by/primitive/
    mod.rs:359-454): per-thread sinks of partitioned hash maps with one accumulator per (key, aggregation), morsel by morsel,
    combined at the end.  One primitive key.  Returns (key bits u64, key valid, [f64 values per agg], [valid per agg]) in
    sink order (partition by partition); with outputs=False only the group count (timing runs)."""
    L = lib()
    n = len(key)
    cap = int(max_groups if max_groups is not None else max(n, 1))
    kinds = (C.c_int * len(aggs))(*[_ffi.AGG_NAMES[k] for _, k in aggs])
    cols = (C.c_int * len(aggs))(*[max(ci, 0) for ci, _ in aggs])
    varr = (_ffi.ColumnStruct * max(len(values), 1))(*[v.struct() for v in values])
    ks = key.struct()
    ng = C.c_int64(0)
    if not outputs:
        _check(L.orc_sink_group_by(C.byref(ks), varr, len(values), cols, kinds, len(aggs), n_threads, morsel, 0, None, None, None, None, C.byref(ng)))
        return int(ng.value)
    ok, okv = np.empty(cap, np.uint64), np.empty(cap, np.uint8)
    ov, ovv = np.empty((len(aggs), cap), np.float64), np.empty((len(aggs), cap), np.uint8)
    _check(L.orc_sink_group_by(C.byref(ks), varr, len(values), cols, kinds, len(aggs), n_threads, morsel, cap, ok.ctypes.data, okv.ctypes.data,
                               ov.ctypes.data, ovv.ctypes.data, C.byref(ng)))
    g = int(ng.value)
    return ok[:g], okv[:g].astype(bool), [ov[i, :g] for i in range(len(aggs))], [ovv[i, :g].astype(bool) for i in range(len(aggs))]


def sorted_key_groups(key: Column) -> list[list[int]]:
    """The sorted-key fast path: GroupsProxy::Slice as [[first, len], ...] in row order.  Restates
    create_groups_from_sorted (crates/polars-core/src/frame/group_by/into_groups.rs:58-127) over
    partition_to_groups_amortized (crates/polars-arrow/src/kernels/sort_partition.rs:71-116): the null_count
    nulls are taken to sit contiguously at the front (if element 0 is null) or at the back and form one group;
    among the values a new group starts wherever `val != first` (VALUE comparison: every NaN starts a group,
    -0.0 == 0.0).  The flag is trusted: unsorted input simply yields one group per run."""
    vals = key.to_numpy()
    n = len(vals)
    if n == 0:
        return []
    mask = key.valid_mask()
    null_count = 0 if mask is None else int((~mask).sum())
    if null_count == n:
        return [[0, n]]
    nulls_first = null_count > 0 and not mask[0]
    v = vals[null_count:] if nulls_first else vals[:n - null_count]
    out = []
    if nulls_first:
        out.append([0, null_count])
    base = null_count if nulls_first else 0
    first_idx, first = 0, v[0]
    for i in range(len(v)):  # the reference's loop, literally
        if v[i] != first:
            out.append([base + first_idx, i - first_idx])
            first_idx, first = i, v[i]
    out.append([base + first_idx, len(v) - first_idx])
    if not nulls_first and null_count > 0:
        out.append([len(v), null_count])
    return out


INTERPOLATIONS = ("nearest", "lower", "higher", "midpoint", "linear")


def agg_quantile(values: Column, quantile: float, interpolation: str, groups: Groups):
    """agg_quantile over GroupsProxy::Idx (dispatch.rs:115-140 -> agg_quantile_generic, aggregations/mod.rs:333-362):
    per group generic_quantile / quantile_slice (chunked_array/ops/aggregate/quantile.rs:19-61, 90-179) of the taken
    values: quantile_idx counts in the sorted column WITH its nulls in front -- base_idx = trunc(((len - nulls) [- 1]) * q
    + nulls) (ceil for Higher; Nearest without the - 1), clamped to len - 1; float_idx = (len - nulls - 1) * q + nulls;
    top_idx = ceil(float_idx) -- Midpoint / Linear interpolate towards sorted[idx + 1] when top_idx != idx; all in f64,
    an f32 column's answer is that cast (quantile.rs:231-246).  A quantile outside [0, 1] -> all-null column of the input
    dtype (mod.rs:346-349).  Returns (values ndarray, valid bool ndarray)."""
    vals = values.to_numpy()
    mask = values.valid_mask()
    G = groups.n_groups
    if not 0.0 <= quantile <= 1.0:
        return np.zeros(G, dtype=vals.dtype), np.zeros(G, dtype=bool)
    out = np.zeros(G, dtype=np.float32 if vals.dtype == np.float32 else np.float64)
    valid = np.zeros(G, dtype=bool)
    for g, idx in enumerate(groups.lists()):
        x = np.sort(np.array([vals[r] for r in idx if mask is None or mask[r]], dtype=vals.dtype))  # NaN last
        length, m = len(idx), len(x)
        if m == 0:
            continue
        nc = length - m
        float_idx = (float(m) - 1.0) * quantile + float(nc)
        if interpolation == "nearest":
            base = int(float(m) * quantile + float(nc))
        elif interpolation == "higher":
            base = int(math.ceil(float_idx))
        else:
            base = int(float_idx)
        base = min(max(base, 0), length - 1)
        top = int(math.ceil(float_idx))
        i = base - nc
        lower = float(x[i])
        r = lower
        if interpolation in ("midpoint", "linear") and top != base:
            upper = float(x[i + 1])
            if lower != upper:
                r = (lower + upper) / 2.0 if interpolation == "midpoint" else (float_idx - float(base)) * (upper - lower) + lower
        with np.errstate(over="ignore"):
            out[g] = r
        valid[g] = True
    return out, valid


def agg_ext(values: Column, kind: str, groups: Groups, ddof: int = 1):
    """first / last / var / std / median over GroupsProxy::Idx, restated in plain Python (small inputs only):
    agg_first / agg_last (crates/polars-core/src/frame/group_by/aggregations/dispatch.rs:47-76: the value, null
    or not, at the group's first / last index; empty group -> null) and agg_var / agg_std (aggregations/mod.rs:
    781-803, 841-862, 1018-1077) over online_variance (crates/polars-arrow/src/kernels/take_agg/var.rs:11-40):
    valid values as f64 in index order, None when count <= ddof; f32 columns answer in f32, the rest in f64.
    Returns (values ndarray, valid bool ndarray)."""
    vals = values.to_numpy()
    mask = values.valid_mask()
    G = groups.n_groups
    if kind in ("first", "last"):
        out = np.zeros(G, dtype=vals.dtype)
        valid = np.zeros(G, dtype=bool)
        for g, idx in enumerate(groups.lists()):
            if idx:
                r = idx[0] if kind == "first" else idx[-1]
                if mask is None or mask[r]:
                    out[g], valid[g] = vals[r], True
        return out, valid
    if kind == "n_unique":
        # agg_n_unique (dispatch.rs:82-107): empty group -> None, else ChunkUnique::n_unique of the taken values
        # (chunked_array/ops/unique/mod.rs:136-171): sorted, then one per element that differs from its predecessor
        # as Option<T> (nulls equal each other; floats compare by IEEE: -0.0 == 0.0, NaN != NaN), as IdxSize
        out = np.zeros(G, dtype=np.uint32)
        valid = np.zeros(G, dtype=bool)
        for g, idx in enumerate(groups.lists()):
            if not idx:
                continue
            x = np.sort(np.array([vals[r] for r in idx if mask is None or mask[r]], dtype=vals.dtype))  # NaN last
            cnt = 1 if len(x) < len(idx) else 0  # the nulls
            for i in range(len(x)):
                if i == 0 or x[i] != x[i - 1]:
                    cnt += 1
            out[g], valid[g] = cnt, True
        return out, valid
    out = np.zeros(G, dtype=np.float32 if vals.dtype == np.float32 else np.float64)
    valid = np.zeros(G, dtype=bool)
    if kind == "median":
        # agg_median = quantile(0.5, Linear) per group (dispatch.rs:98-113; quantile_idx / linear_interpol /
        # generic_quantile, chunked_array/ops/aggregate/quantile.rs:19-54,133-175): valid values sorted ascending
        # (NaN last), float_idx = (m - 1) * 0.5, lower + (float_idx - idx) * (upper - lower) in f64, cast for an f32 column
        ft = out.dtype.type
        for g, idx in enumerate(groups.lists()):
            x = np.array([vals[r] for r in idx if mask is None or mask[r]], dtype=vals.dtype)
            if len(x) == 0:
                continue
            x = np.sort(x)
            float_idx = (len(x) - 1.0) * 0.5
            i, top = int(float_idx), int(np.ceil(float_idx))
            lower, upper = float(x[i]), float(x[top])  # f64 throughout; an f32 column's answer is the cast (quantile.rs:231-246)
            out[g] = ft(lower if lower == upper else (float_idx - float(i)) * (upper - lower) + lower)
            valid[g] = True
        return out, valid
    for g, idx in enumerate(groups.lists()):
        m2 = mean = 0.0
        count = 0
        for r in idx:
            if mask is not None and not mask[r]:
                continue
            value = float(vals[r])
            count += 1
            delta_1 = value - mean
            mean = delta_1 / count + mean
            m2 += delta_1 * (value - mean)
        if idx and count > ddof:
            v = m2 / (count - ddof)
            out[g], valid[g] = (v ** 0.5 if kind == "std" else v), True
    return out, valid


def utf8_group_codes(offsets: np.ndarray, data: np.ndarray, valid: np.ndarray | None):
    """orc_utf8_group_codes: (codes[n] u32, first[G] u32) of a LargeUtf8 column (into_groups.rs:232-290), groups numbered by
    first occurrence; `valid` is a bool mask or None."""
    n = len(offsets) - 1
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    data = np.ascontiguousarray(data, dtype=np.uint8)
    bits = None if valid is None else np.packbits(np.asarray(valid, dtype=bool), bitorder="little")
    codes, first = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint32)
    g = lib().orc_utf8_group_codes(offsets.ctypes.data, data.ctypes.data if len(data) else None, None if bits is None else bits.ctypes.data, 0, n,
                                   codes.ctypes.data, first.ctypes.data)
    if g < 0:
        raise MemoryError("orc_utf8_group_codes")
    return codes[:n], first[:g]


def agg_utf8(strings: list, kind: str, groups: Groups) -> list:
    """A string column over GroupsProxy::Idx, restated in plain Python (small cases).  min / max: Utf8Chunked::agg_min / agg_max
    (aggregations/utf8.rs:21-142): an empty group -> None, a one-row group -> that row (null stays null), otherwise the fold
    `if acc < v { acc } else { v }` over the group's non-null rows in Rust's `str` order (= byte order), None when all are null.
    first / last: Series::agg_first / agg_last (aggregations/dispatch.rs:48-78,182-212): the value at the group's first / last row.
    n_unique: distinct values of the group, a null counting as one (dispatch.rs:82-105).  sum / mean: full-null columns
    (dispatch.rs:126,177)."""
    as_b = lambda v: v if isinstance(v, bytes) else v.encode()
    out = []
    for g in range(groups.n_groups):
        idx = groups.indices[groups.offsets[g]:groups.offsets[g + 1]].tolist()
        if not idx or kind in ("sum", "mean"):
            out.append(None)
        elif kind == "first":
            out.append(strings[int(groups.first[g])])
        elif kind == "last":
            out.append(strings[idx[-1]])
        elif kind == "n_unique":
            out.append(len({strings[i] for i in idx}))
        elif kind in ("min", "max"):
            acc = None
            for i in idx:
                v = strings[i]
                if v is None:
                    continue
                if acc is None:
                    acc = v
                elif kind == "min":
                    acc = acc if as_b(acc) < as_b(v) else v
                else:
                    acc = acc if as_b(acc) > as_b(v) else v
            out.append(acc)
        else:
            raise ValueError(kind)
    return out


def hash_join(left: Column, right: Column, how: str = "inner"):
    """orc_hash_join: (left_idx, right_idx) u32 arrays in the reference's pair order; right_idx 0xFFFFFFFF = no match (left join)."""
    h = {"inner": 0, "left": 1, "semi": 2, "anti": 3, "outer": 4}[how]
    ls, rs = left.struct(), right.struct()
    m = lib().orc_hash_join(C.byref(ls), C.byref(rs), h, None, None)
    if m < 0:
        raise ValueError("orc_hash_join: unsupported key columns")
    li, ri = np.zeros(max(m, 1), np.uint32), np.zeros(max(m, 1), np.uint32)
    lib().orc_hash_join(C.byref(ls), C.byref(rs), h, li.ctypes.data, ri.ctypes.data)
    return li[:m], ri[:m]


def hash_join_multi(left: list[Column], right: list[Column], how: str = "inner"):
    """orc_hash_join_multi: the pairs of a join on several key columns (multiple_keys.rs), same conventions as hash_join."""
    h = {"inner": 0, "left": 1, "semi": 2, "anti": 3, "outer": 4}[how]
    la, ra = _structs(left), _structs(right)
    m = lib().orc_hash_join_multi(la, ra, len(left), h, None, None)
    if m < 0:
        raise ValueError("orc_hash_join_multi: unsupported key columns")
    li, ri = np.zeros(max(m, 1), np.uint32), np.zeros(max(m, 1), np.uint32)
    lib().orc_hash_join_multi(la, ra, len(left), h, li.ctypes.data, ri.ctypes.data)
    return li[:m], ri[:m]
