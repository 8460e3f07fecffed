"""Host-side mirror of the reference's group-by API for the accelerated path.

    df.group_by("id3", maintain_order=False).agg([col("v1").sum(), col("v3").mean()])
    df.lazy().group_by("id3").agg(...).collect()

mirrors py-polars `DataFrame.group_by(...).agg(...)` (py-polars/polars/dataframe/group_by.py:241-244, which
itself is `self.df.lazy().group_by(by, maintain_order).agg(aggs).collect(no_optimization=True)`), and the
eager Rust `GroupBy::{sum,mean,min,max,count}` helpers with their `{name}_{agg}` output names
(crates/polars-core/src/frame/group_by/mod.rs:348-699, 889-909).  Only what the hash group-by hot path
needs exists here: plain-column keys, and sum/mean/min/max/count of plain columns.  All compute happens in
libplhip.so (HIP); nothing in this file computes an aggregate.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Sequence

import numpy as np

from . import _ffi
from .column import Column, unpack_validity


# ---------------------------------------------------------------------------------------------------
# expressions: col("x").sum().alias("y"), count()
# ---------------------------------------------------------------------------------------------------
class Expr:
    """A plain column, optionally with one aggregation and an alias (AggregationExpr,
    crates/polars-lazy/src/physical_plan/expressions/aggregation.rs:19-33)."""

    def __init__(self, column: str | None, agg: str | None = None, alias: str | None = None):
        self.column, self.agg, self._alias = column, agg, alias

    def _with(self, agg: str) -> "Expr":
        if self.agg is not None:
            raise _ffi.ComputeError(1, "nested aggregations are outside the accelerated path")
        return Expr(self.column, agg, self._alias)

    def sum(self) -> "Expr":
        return self._with("sum")

    def mean(self) -> "Expr":
        return self._with("mean")

    def min(self) -> "Expr":
        return self._with("min")

    def max(self) -> "Expr":
        return self._with("max")

    def count(self) -> "Expr":
        return self._with("count")

    # aggregations the reference evaluates over the GroupsProxy (AggregationExpr::evaluate_on_groups, aggregation.rs:44-258): they
    # run through the CSR seams (plhip_group_tuples + plhip_agg_groups), as db-benchmark's q6 does (median(v3), std(v3) by id4, id5)
    def first(self) -> "Expr":
        return self._with("first")

    def last(self) -> "Expr":
        return self._with("last")

    def median(self) -> "Expr":
        return self._with("median")

    def n_unique(self) -> "Expr":
        return self._with("n_unique")

    def std(self, ddof: int = 1) -> "Expr":
        e = self._with("std")
        e.ddof = int(ddof)
        return e

    def var(self, ddof: int = 1) -> "Expr":
        e = self._with("var")
        e.ddof = int(ddof)
        return e

    def quantile(self, quantile: float, interpolation: str = "nearest") -> "Expr":
        e = self._with("quantile")
        e.q, e.interpolation = float(quantile), interpolation
        return e

    ddof, q, interpolation = 1, 0.5, "nearest"

    def alias(self, name: str) -> "Expr":
        e = Expr(self.column, self.agg, name)
        e.ddof, e.q, e.interpolation = self.ddof, self.q, self.interpolation
        return e

    @property
    def output_name(self) -> str:
        # lazy aggregations keep the input column name unless aliased (aggregation.rs:51-52)
        return self._alias or (self.column if self.column is not None else "count")


def col(name: str) -> Expr:
    return Expr(name)


def count() -> Expr:
    """pl.count(): group length including nulls (CountExpr, expressions/count.rs:28-37)."""
    return Expr(None, "count", "count")


def sum_(name: str) -> Expr:  # exported as polars_amd.sum (mirrors pl.sum)
    return col(name).sum()


def mean_(name: str) -> Expr:
    return col(name).mean()


def min_(name: str) -> Expr:
    return col(name).min()


def max_(name: str) -> Expr:
    return col(name).max()


# ---------------------------------------------------------------------------------------------------
# frames
# ---------------------------------------------------------------------------------------------------
class DataFrame:
    def __init__(self, data=None):
        self._cols: dict[str, Column] = {}
        if data is None:
            data = {}
        if isinstance(data, dict):
            for name, v in data.items():
                self._add(v if isinstance(v, Column) or hasattr(v, "struct") or hasattr(v, "chunks") else self._make(name, v), name)
        else:
            for c in data:
                self._add(c, c.name)

    @staticmethod
    def _make(name, v) -> Column:
        if isinstance(v, tuple):  # (values, valid_mask)
            return Column(name, v[0], v[1])
        if isinstance(v, list):
            if any(isinstance(x, (str, bytes)) for x in v):  # a Utf8 column (strings.py: a key, or a value under min / max / first / last)
                from .strings import Utf8Column

                return Utf8Column.from_list(name, v)
            if any(x is None for x in v):
                mask = np.array([x is not None for x in v])
                vals = np.array([0 if x is None else x for x in v])
                return Column(name, vals, mask)
            v = np.array(v)
        return Column(name, v)

    def _add(self, c: Column, name: str):
        if name in self._cols:
            raise _ffi.PlHipError(6, f"duplicate column {name!r}")
        if self._cols and len(c) != self.height:
            raise _ffi.ShapeMismatch(2, f"column {name!r} has {len(c)} rows, frame has {self.height}")
        c.name = name
        self._cols[name] = c

    @property
    def columns(self) -> list[str]:
        return list(self._cols)

    @property
    def height(self) -> int:
        return len(next(iter(self._cols.values()))) if self._cols else 0

    @property
    def shape(self) -> tuple[int, int]:
        return (self.height, len(self._cols))

    def __getitem__(self, name: str) -> Column:
        return self._cols[name]

    def get_column(self, name: str) -> Column:
        if name not in self._cols:
            raise _ffi.PlHipError(6, f"column {name!r} not found")
        return self._cols[name]

    def to_dict(self) -> dict[str, list]:
        return {n: c.to_list() for n, c in self._cols.items()}

    def rows(self) -> list[tuple]:
        cols = [c.to_list() for c in self._cols.values()]
        return list(zip(*cols))

    def lazy(self) -> "LazyFrame":
        return LazyFrame(self)

    def group_by(self, *by, maintain_order: bool = False) -> "GroupBy":
        return GroupBy(self, _flatten(by), maintain_order)

    def join(self, other: "DataFrame", on: str | None = None, how: str = "inner", *, left_on: str | None = None,
             right_on: str | None = None, suffix: str = "_right") -> "DataFrame":
        """DataFrame.join on one integer key, `how` = 'inner' | 'left' | 'semi' | 'anti' | 'outer' (py-polars/polars/dataframe/frame.py:5856): the index
        pairs come from plhip_hash_join in the reference's order and stay in HBM; the payload gather behind `_finish_join` (left
        columns, then the right ones without the key, clashing names suffixed) is plhip_take on the device for fixed-width columns,
        on the host for strings."""
        from .join import join_indices, take

        if on is not None:
            left_on = right_on = on
        if left_on is None or right_on is None:
            raise ValueError("you should pass the column to join on as an argument")
        lo = [left_on] if isinstance(left_on, str) else list(left_on)
        ro = [right_on] if isinstance(right_on, str) else list(right_on)
        lk, rk = [self.get_column(n) for n in lo], [other.get_column(n) for n in ro]
        for i in range(len(lk)):  # string keys: both sides' codes under one dictionary, built on the device
            if type(lk[i]).__name__ == "Utf8Column" and type(rk[i]).__name__ == "Utf8Column":
                from .strings import encode_together

                lk[i], rk[i] = encode_together(lk[i], rk[i])
        from .join import Join

        for c in lk + rk:  # (device-resident key columns: torch's queued work must have landed before the library reads them)
            if c.location == _ffi.DEVICE and type(c.values).__module__.startswith("torch"):
                import torch

                torch.cuda.current_stream(c.values.device).synchronize()
                break
        jn = Join(lk, rk, how)  # the pairs stay in HBM: fixed-width payload columns are gathered there (plhip_take)
        li, ri = jn.indices()   # (host copies: string payloads and the outer join's key coalescing)
        right_on = ro
        fixed = lambda c: type(c).__name__ == "Column" and c.dtype != _ffi.BOOL and len(c) > 0

        def gather(frame_cols, side, idx, rename):
            items = list(frame_cols.items())
            dev = jn.take([c for _, c in items if fixed(c)], side, [rename(n) for n, c in items if fixed(c)])
            it = iter(dev)
            return [next(it) if fixed(c) else take(c, idx, rename(n)) for n, c in items]

        out = gather(self._cols, "left", li, lambda n: n)
        if how in ("semi", "anti"):  # only the left frame's rows
            jn.close()
            return DataFrame(out)
        if how == "outer":  # the key columns are coalesced: the right frame's key where the left side has no row
            from .join import NONE

            for ln, rn in zip(lo, ro):
                i = list(self._cols).index(ln)
                lcol, rcol = out[i], take(other.get_column(rn), ri, ln)
                lv, rv = lcol.to_list(), rcol.to_list()
                merged = [r if a == NONE else l for l, r, a in zip(lv, rv, li.tolist())]
                if type(self.get_column(ln)).__name__ == "Utf8Column":
                    from .strings import DecodedColumn

                    out[i] = DecodedColumn(ln, merged)
                    continue
                dt = _ffi.code_np_dtype(self.get_column(ln).dtype)
                mask = np.array([x is not None for x in merged], dtype=bool)
                out[i] = Column(ln, np.array([0 if x is None else x for x in merged], dtype=dt), None if mask.all() else mask)
        taken = set(self._cols)
        out += gather({n: c for n, c in other._cols.items() if n not in right_on}, "right", ri, lambda n: n + suffix if n in taken else n)
        jn.close()
        return DataFrame(out)

    def group_by_stable(self, *by) -> "GroupBy":
        """DataFrame::group_by_stable (group_by/mod.rs:122-129)."""
        return GroupBy(self, _flatten(by), True)

    def __repr__(self) -> str:
        return f"DataFrame(shape={self.shape}, columns={self.columns})"


def _flatten(by) -> list[str]:
    out: list[str] = []
    for b in by:
        if isinstance(b, (list, tuple)):
            out.extend(_flatten(b))
        elif isinstance(b, Expr):
            if b.agg is not None:
                raise _ffi.ComputeError(1, "aggregations are not valid group-by keys")
            out.append(b.column)
        else:
            out.append(b)
    return out


class GroupBy:
    """Mirrors py-polars GroupBy / Rust GroupBy{df, selected_keys, groups, selected_agg}
    (group_by/mod.rs:182-189)."""

    def __init__(self, df: DataFrame, by: list[str], maintain_order: bool, selection: list[str] | None = None):
        self.df, self.by, self.maintain_order, self.selection = df, by, maintain_order, selection

    def select(self, names: Iterable[str]) -> "GroupBy":
        return GroupBy(self.df, self.by, self.maintain_order, list(names))

    def agg(self, *aggs) -> DataFrame:
        exprs: list[Expr] = []
        for a in aggs:
            exprs.extend(a if isinstance(a, (list, tuple)) else [a])
        return _execute(self.df, self.by, exprs, self.maintain_order)

    def groups(self) -> "Groups":
        """GroupBy::get_groups: the GroupsProxy of the key columns (group_by/mod.rs:182-189)."""
        return group_tuples([self.df.get_column(n) for n in self.by], self.maintain_order)

    def _eager(self, kind: str) -> DataFrame:
        # GroupBy::sum()/mean()/... aggregate every non-key column (or the selection) and name the
        # outputs `{name}_{agg}` (fmt_group_by_column, group_by/mod.rs:889-909)
        names = self.selection if self.selection is not None else [c for c in self.df.columns if c not in self.by]
        return _execute(self.df, self.by, [Expr(n, kind, f"{n}_{kind}") for n in names], self.maintain_order)

    def sum(self) -> DataFrame:
        return self._eager("sum")

    def mean(self) -> DataFrame:
        return self._eager("mean")

    def min(self) -> DataFrame:
        return self._eager("min")

    def max(self) -> DataFrame:
        return self._eager("max")

    def count(self) -> DataFrame:
        return self._eager("count")

    def _two_step(self, kind: str, suffix: str, ddof: int = 1, quantile: float = 0.5, interpolation: str = "nearest") -> DataFrame:
        """The eager GroupBy's non-fused aggregations (group_by/mod.rs:498-662): groups first (GroupsProxy), then one
        agg_* per selected column over them; keys = each key column taken at the groups' first rows (keys_sliced,
        mod.rs:250-297); output names as fmt_group_by_column (mod.rs:889-909)."""
        names = self.selection if self.selection is not None else [c for c in self.df.columns if c not in self.by]
        g = self.groups()
        try:
            cols = []
            for k in self.by:
                c = agg_groups(self.df.get_column(k), "first", g)
                c.name = k
                cols.append(c)
            for n in names:
                if kind == "quantile":
                    c = agg_groups_quantile(self.df.get_column(n), quantile, g, interpolation)
                else:
                    c = agg_groups(self.df.get_column(n), kind, g, ddof=ddof)
                c.name = f"{n}_{suffix}"
                cols.append(c)
        finally:
            g.close()
        return DataFrame(cols)

    def first(self) -> DataFrame:
        return self._two_step("first", "first")

    def last(self) -> DataFrame:
        return self._two_step("last", "last")

    def var(self, ddof: int = 1) -> DataFrame:
        return self._two_step("var", "agg_var", ddof)

    def std(self, ddof: int = 1) -> DataFrame:
        return self._two_step("std", "agg_std", ddof)

    def median(self) -> DataFrame:
        return self._two_step("median", "median")

    def n_unique(self) -> DataFrame:
        return self._two_step("n_unique", "n_unique")

    def quantile(self, quantile: float, interpolation: str = "nearest") -> DataFrame:
        # GroupBy::quantile (group_by/mod.rs:602-630): errors outside [0, 1]; names `{name}_quantile_{q:.2}` (mod.rs:905)
        if not 0.0 <= quantile <= 1.0:
            raise _ffi.PlHipError(1, "ComputeError: `quantile` should be within 0.0 and 1.0")
        return self._two_step("quantile", f"quantile_{quantile:.2f}", quantile=quantile, interpolation=interpolation)


class LazyFrame:
    def __init__(self, df: DataFrame):
        self._df = df

    def group_by(self, *by, maintain_order: bool = False) -> "LazyGroupBy":
        return LazyGroupBy(self._df, _flatten(by), maintain_order)

    def collect(self) -> DataFrame:
        return self._df.materialize() if hasattr(self._df, "materialize") else self._df  # (a scanned file: read it whole)


class LazyGroupBy:
    def __init__(self, df, by, maintain_order):
        self._df, self._by, self._mo = df, by, maintain_order

    def agg(self, *aggs) -> "_LazyAgg":
        exprs: list[Expr] = []
        for a in aggs:
            exprs.extend(a if isinstance(a, (list, tuple)) else [a])
        return _LazyAgg(self._df, self._by, exprs, self._mo)


class _LazyAgg:
    def __init__(self, df, by, exprs, mo, slice_=None):
        self._args = (df, by, exprs, mo)
        self._slice = slice_

    def slice(self, offset: int, length: int | None = None) -> "_LazyAgg":
        """`.slice()` right after the aggregation: the optimizer pushes it into GroupByExec
        (crates/polars-lazy/src/physical_plan/executors/group_by.rs:79-82); here into plhip_options_t."""
        if self._slice is not None:
            raise _ffi.PlHipError(6, "one slice per query on the accelerated path")
        return _LazyAgg(*self._args, slice_=(int(offset), (1 << 63) - 1 if length is None else int(length)))

    def head(self, n: int = 5) -> "_LazyAgg":
        return self.slice(0, n)

    def tail(self, n: int = 5) -> "_LazyAgg":
        return self.slice(-n, n)

    def collect(self) -> DataFrame:
        return _execute(*self._args, slice_=self._slice)


# ---------------------------------------------------------------------------------------------------
# the one place that calls the C ABI
# ---------------------------------------------------------------------------------------------------
def set_strategy(name: str) -> None:
    """'auto' | 'global' | 'lds' | 'partitioned' | 'direct' | 'tilesort' | 'tilesort2' | 'sort' | 'dense'
    (POLARS_FORCE_PARTITION / POLARS_NO_PARTITION analogue)."""
    _ffi.check(_ffi.lib().plhip_set_strategy(_ffi.STRATEGIES[name]))


def last_strategy() -> str:
    return _ffi.lib().plhip_last_strategy().decode()


def last_kernel_ms() -> tuple[float, float, str]:
    t, d, k = C.c_float(), C.c_float(), C.c_char_p()
    _ffi.check(_ffi.lib().plhip_last_kernel_ms(C.byref(t), C.byref(d), C.byref(k)))
    return t.value, d.value, (k.value or b"").decode()


def last_kernel_breakdown() -> tuple[float, float, float]:
    a, b, c = C.c_float(), C.c_float(), C.c_float()
    _ffi.check(_ffi.lib().plhip_last_kernel_breakdown(C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


def group_by_agg_raw(keys: Sequence[Column], values: Sequence[Column], aggs: Sequence[tuple[int, int]],
                     flags: int = 0, n_groups_hint: int = 0, slice_: tuple[int, int] | None = None):
    """Thin call of plhip_group_by_agg.  Returns (handle, [ColumnStruct...]); the caller must
    plhip_result_free(handle).  Used by bench.py (device-resident outputs) and by _execute."""
    L = _ffi.lib()
    # libplhip.so runs on its own stream: device-resident columns must be complete when it reads them, so whatever torch
    # still has queued on its current stream (the kernels that produce these tensors) is waited for first
    for c in list(keys) + list(values):
        if c.location == _ffi.DEVICE and type(c.values).__module__.startswith("torch"):
            import torch

            torch.cuda.current_stream(c.values.device).synchronize()
            break
    karr = (_ffi.ColumnStruct * max(1, len(keys)))(*[k.struct() for k in keys])
    varr = (_ffi.ColumnStruct * max(1, len(values)))(*[v.struct() for v in values])
    aarr = (_ffi.AggStruct * max(1, len(aggs)))(*[_ffi.AggStruct(c, k) for c, k in aggs])
    handle = C.c_void_p()
    opts = _ffi.OptionsStruct(C.sizeof(_ffi.OptionsStruct), int(n_groups_hint))
    if slice_ is not None:
        opts.slice_offset, opts.slice_len, opts.has_slice = int(slice_[0]), int(slice_[1]), 1
    _ffi.check(L.plhip_group_by_agg_opts(karr, len(keys), varr, len(values), aarr, len(aggs), flags, C.byref(opts),
                                         C.byref(handle)))
    cols = []
    for i in range(L.plhip_result_num_columns(handle)):
        cs = _ffi.ColumnStruct()
        _ffi.check(L.plhip_result_column(handle, i, C.byref(cs)))
        cols.append(cs)
    return handle, cols


def group_by_agg_chunked_raw(keys: Sequence[Sequence[Column]], values: Sequence[Sequence[Column]], aggs: Sequence[tuple[int, int]],
                             flags: int = 0, n_groups_hint: int = 0, slice_: tuple[int, int] | None = None):
    """plhip_group_by_agg_chunked: every column is a list of chunks (Columns of one dtype, host or device, each with its own
    offset / validity); the library lays them end to end in HBM as the reference's rechunk does (executors/group_by.rs:66).
    Returns (handle, [ColumnStruct ...]) like group_by_agg_raw."""
    L = _ffi.lib()
    for col in list(keys) + list(values):
        for c in col:
            if c.location == _ffi.DEVICE and type(c.values).__module__.startswith("torch"):
                import torch

                torch.cuda.current_stream(c.values.device).synchronize()
                break
    keep = []

    def chunked(cols):
        arr = (_ffi.ChunkedColumnStruct * max(1, len(cols)))()
        for i, chunks in enumerate(cols):
            ca = (_ffi.ColumnStruct * max(1, len(chunks)))(*[c.struct() for c in chunks])
            keep.append(ca)
            arr[i].chunks, arr[i].n_chunks = ca, len(chunks)
        return arr

    karr, varr = chunked(keys), chunked(values)
    aarr = (_ffi.AggStruct * max(1, len(aggs)))(*[_ffi.AggStruct(c, k) for c, k in aggs])
    handle = C.c_void_p()
    opts = _ffi.OptionsStruct(C.sizeof(_ffi.OptionsStruct), int(n_groups_hint))
    if slice_ is not None:
        opts.slice_offset, opts.slice_len, opts.has_slice = int(slice_[0]), int(slice_[1]), 1
    _ffi.check(L.plhip_group_by_agg_chunked(karr, len(keys), varr, len(values), aarr, len(aggs), flags, C.byref(opts), C.byref(handle)))
    cols = []
    for i in range(L.plhip_result_num_columns(handle)):
        cs = _ffi.ColumnStruct()
        _ffi.check(L.plhip_result_column(handle, i, C.byref(cs)))
        cols.append(cs)
    return handle, cols


class _ResultOwner:
    """Keeps a host-resident plhip_result_t alive for as long as any array that aliases its buffers: the frame `collect()` returns
    IS the library's result (a DataFrame owns its Series' buffers the same way); plhip_result_free runs when the last view is gone."""

    def __init__(self, handle):
        self.h = handle

    def __del__(self):
        try:
            if self.h:
                _ffi.lib().plhip_result_free(self.h)
                self.h = None
        except Exception:
            pass


def _bytes_view(ptr, nbytes: int, owner) -> np.ndarray:
    buf = (C.c_uint8 * nbytes).from_address(ptr)
    buf._owner = owner  # numpy's base chain holds `buf`, `buf` holds the result
    return np.frombuffer(buf, dtype=np.uint8)


def _host_column(name: str, cs: _ffi.ColumnStruct, owner=None) -> Column:
    """A host-resident result column.  Without `owner` the buffers are copied (the caller frees the result right away); with one
    the column aliases them (no second copy of a large frame: q10's 1e7 groups x 8 columns)."""
    n, off = cs.length, cs.offset  # a sliced frame's columns carry an Arrow offset
    dt = _ffi.code_np_dtype(cs.dtype)
    if n == 0:
        return Column(name, np.empty(0, dtype=dt))
    if cs.dtype == _ffi.BOOL:  # a bitmap: one bit per row
        bits = np.ctypeslib.as_array(C.cast(cs.values, C.POINTER(C.c_uint8)), shape=((off + n + 7) // 8,))
        vals = np.unpackbits(bits, bitorder="little")[off:off + n].astype(np.bool_)
    elif owner is not None:
        vals = _bytes_view(cs.values, (off + n) * dt.itemsize, owner).view(dt)[off:]
    else:
        vals = np.ctypeslib.as_array(C.cast(cs.values, C.POINTER(C.c_uint8)), shape=((off + n) * dt.itemsize,)).view(dt)[off:].copy()
    validity = None
    if cs.validity:
        nb = (off + n + 7) // 8
        bits = np.ctypeslib.as_array(C.cast(cs.validity, C.POINTER(C.c_uint8)), shape=(nb,))
        validity = unpack_validity(bits, n, off) if off else bits.copy()
        return Column(name, vals, validity, packed=not off)
    return Column(name, vals)


def _execute(df: DataFrame, by: list[str], exprs: list[Expr], maintain_order: bool, extra_flags: int = 0,
             slice_: tuple[int, int] | None = None) -> DataFrame:
    """group_by_helper (crates/polars-lazy/src/physical_plan/executors/group_by.rs:57-95): keys then aggs."""
    if not by:
        raise _ffi.ComputeError(1, "at least one key is required in a group_by operation")
    if any(e.agg in _TWO_STEP_AGGS for e in exprs):
        return _execute_over_groups(df, by, exprs, maintain_order, slice_)
    keys = [df.get_column(n) for n in by]
    dicts = {}
    for i, k in enumerate(keys):  # Utf8 keys: dense codes + dictionary on the device (into_groups.rs:232-290), encoded once per column
        if type(k).__name__ == "Utf8Column":
            from .strings import codes_of

            keys[i] = codes_of(k)
            dicts[i] = keys[i].dictionary
    vals: list[Column] = []
    index: dict[str, int] = {}
    aggs: list[tuple[int, int]] = []
    names = list(by)
    str_outs: dict[int, object] = {}  # output position -> how a string column's aggregate is decoded
    for e in exprs:
        if e.agg is None:
            raise _ffi.ComputeError(1, "only col(x).sum/mean/min/max/count and count() run on the accelerated path")
        if e.column is None:
            aggs.append((-1, _ffi.AGG_COUNT))
        else:
            c = df.get_column(e.column)
            slot, kind = e.column, e.agg
            if type(c).__name__ == "Utf8Column":
                # a string VALUE column (Utf8Chunked::agg_min / agg_max, aggregations/utf8.rs:21-142): the strings are ranked in byte
                # order once (strings.ranks_of: encode on the device, order the dictionary, code -> rank on the device) and the fused
                # query takes min / max of the u32 ranks; nulls are skipped, an all-null group yields null.  sum / mean of strings
                # are full-null columns in the reference (dispatch.rs:126,177; its test_streaming_group_by_types: str_sum = [None])
                from .strings import codes_of, ranks_of

                if e.agg in ("min", "max"):
                    c, slot = ranks_of(c), e.column + "\0rank"
                    str_outs[len(names)] = c
                else:
                    c, slot = codes_of(c), e.column + "\0code"
                    if e.agg in ("sum", "mean"):
                        kind = "count"
                        str_outs[len(names)] = None
            if slot not in index:
                index[slot] = len(vals)
                vals.append(c)
            aggs.append((index[slot], _ffi.AGG_NAMES[kind]))
        if e.output_name in names:
            raise _ffi.PlHipError(6, f"column with name {e.output_name!r} has more than one occurrence")  # DuplicateError
        names.append(e.output_name)
    flags = (_ffi.FLAG_MAINTAIN_ORDER if maintain_order else 0) | extra_flags
    if len(keys) == 1 and keys[0].sorted_flag is not None:  # IsSorted flag on the key: the sorted fast path
        flags |= _ffi.FLAG_KEYS_SORTED
    hint = 0
    if len(keys) == 1 and keys[0].n_categories is not None:  # a Categorical's codes: the dense-code path (perfect.rs:190-227)
        flags |= _ffi.FLAG_KEYS_DENSE
        hint = keys[0].n_categories
    from .io import ChunkedColumn

    if any(isinstance(c, ChunkedColumn) for c in keys + vals):
        # columns as a reader delivered them (io.py): the chunks go to the library as they are and are laid end to end in HBM
        # (plhip_group_by_agg_chunked) -- the reference's df.as_single_chunk_par(), executors/group_by.rs:66, without the host copy
        as_chunks = lambda c: c.chunks if isinstance(c, ChunkedColumn) else [c]
        handle, cols = group_by_agg_chunked_raw([as_chunks(k) for k in keys], [as_chunks(v) for v in vals], aggs, flags, hint, slice_=slice_)
    else:
        handle, cols = group_by_agg_raw(keys, vals, aggs, flags, hint, slice_=slice_)
    owner = _ResultOwner(handle)  # (freed with the last column that aliases it, or right here when a column cannot be made)
    host = [_host_column(nm, cs, owner) for nm, cs in zip(names, cols)]
    del owner
    for i, r in str_outs.items():  # string aggregates: ranks back to strings; sum / mean of strings: all null
        from .strings import DecodedColumn

        host[i] = DecodedColumn(names[i], [None] * len(host[i]) if r is None else r.decode(host[i].to_numpy(), host[i].valid_mask()))
    for i, k in enumerate(keys):  # a dictionary-encoded key read from a file: its codes back to the categories
        cats = getattr(k, "dictionary", None)
        if isinstance(cats, list) and i not in dicts:
            from .strings import DecodedColumn

            codes, m = host[i].to_numpy(), host[i].valid_mask()
            host[i] = DecodedColumn(names[i], [cats[int(c)] if (m is None or m[j]) else None for j, c in enumerate(codes)])
    for i, d in dicts.items():  # the key codes back to their strings, as a Categorical's rev-map does at output
        from .strings import DecodedColumn

        host[i] = DecodedColumn(names[i], d.decode(host[i].to_numpy(), host[i].valid_mask()))
    return DataFrame(host)


_TWO_STEP_AGGS = ("first", "last", "median", "n_unique", "std", "var", "quantile")


def _slice_rows(c, offset: int, length: int):
    """A result column's rows [offset, offset + length) with the reference's slice rules (negative offsets count from the end,
    out-of-range parts are dropped: slice_offsets, crates/polars-core/src/utils/mod.rs)."""
    n = len(c)
    if offset < 0:
        back = min(-offset, n)
        start, ln = n - back, min(length, back)
    else:
        start = min(offset, n)
        ln = min(length, n - start)
    if type(c).__name__ == "DecodedColumn":
        from .strings import DecodedColumn

        return DecodedColumn(c.name, c.to_list()[start:start + ln])
    m = c.valid_mask()
    return Column(c.name, c.to_numpy()[start:start + ln].copy(), None if m is None else m[start:start + ln].copy())


def _execute_over_groups(df: DataFrame, by: list[str], exprs: list[Expr], maintain_order: bool, slice_=None) -> DataFrame:
    """A query with an aggregation that has no fused accumulator (first / last / median / n_unique / std / var / quantile): the
    reference's general route -- the GroupsProxy once, then every expression over it (group_by_helper, executors/group_by.rs:57-95
    with AggregationExpr::evaluate_on_groups, aggregation.rs:44-258).  Keys are the key columns at the groups' first rows
    (keys_sliced, group_by/mod.rs:250-297).  The CSR stays in HBM when the key columns are there."""
    keys = [df.get_column(n) for n in by]
    on_dev = all(getattr(k, "location", _ffi.HOST) == _ffi.DEVICE for k in keys)
    g = group_tuples(keys, maintain_order, _ffi.DEVICE if on_dev else _ffi.HOST)
    try:
        cols, names = [], list(by)
        for k, n in zip(keys, by):
            c = agg_groups(k, "first", g)
            c.name = n
            cols.append(c)
        for e in exprs:
            if e.agg is None:
                raise _ffi.ComputeError(1, "every expression of agg() needs an aggregation on the accelerated path")
            if e.output_name in names:
                raise _ffi.PlHipError(6, f"column with name {e.output_name!r} has more than one occurrence")  # DuplicateError
            names.append(e.output_name)
            v = None if e.column is None else df.get_column(e.column)
            if e.agg == "quantile":
                c = agg_groups_quantile(v, e.q, g, e.interpolation)
            else:
                c = agg_groups(v, e.agg, g, ddof=e.ddof)
            c.name = e.output_name
            cols.append(c)
    finally:
        g.close()
    if slice_ is not None:
        cols = [_slice_rows(c, int(slice_[0]), int(slice_[1])) for c in cols]
    return DataFrame(cols)


# ---------------------------------------------------------------------------------------------------
# two-step API: GroupsProxy + Series.agg_* (the seams a Rust host layer would rebind)
# ---------------------------------------------------------------------------------------------------
class Groups:
    """GroupsProxy::Idx (crates/polars-core/src/frame/group_by/proxy.rs:16-20) as CSR: `first[G]`,
    `offsets[G + 1]`, `indices[N]` (u32 = IdxSize).  Wraps a plhip_groups_t handle."""

    def __init__(self, handle):
        self._h = handle
        L = _ffi.lib()
        self.n_groups = L.plhip_groups_num_groups(handle)
        self.n_rows = L.plhip_groups_num_rows(handle)

    def buffers(self):
        """(first, offsets, indices) as numpy arrays (host handles) or raw device pointers (device handles)."""
        L = _ffi.lib()
        f, o, i, loc = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int32()
        _ffi.check(L.plhip_groups_buffers(self._h, C.byref(f), C.byref(o), C.byref(i), C.byref(loc)))
        if loc.value == _ffi.DEVICE:
            return f.value, o.value, i.value
        as_np = lambda p, n: np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(max(n, 1),))[:n].copy()
        return as_np(f, self.n_groups), as_np(o, self.n_groups + 1), as_np(i, self.n_rows)

    def lists(self) -> list[list[int]]:
        _, off, idx = self.buffers()
        return [idx[off[g]:off[g + 1]].tolist() for g in range(self.n_groups)]

    def close(self):
        if self._h is not None:
            _ffi.lib().plhip_groups_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def group_tuples(keys: Sequence[Column], sorted_: bool = False, location: int = _ffi.HOST) -> Groups:
    """IntoGroupsProxy::group_tuples (into_groups.rs:12-19) / group_by_threaded_multiple_keys_flat
    (hashing.rs:380-468) through plhip_group_tuples."""
    keys = [k.rechunk() if hasattr(k, "rechunk") else k for k in keys]  # (a reader's chunked column: one chunk for this seam)
    if any(type(k).__name__ == "Utf8Column" for k in keys):  # string keys group by their codes (into_groups.rs:232-290)
        from .strings import codes_of

        keys = [codes_of(k) if type(k).__name__ == "Utf8Column" else k for k in keys]
    karr = (_ffi.ColumnStruct * max(1, len(keys)))(*[k.struct() for k in keys])
    h = C.c_void_p()
    _ffi.check(_ffi.lib().plhip_group_tuples(karr, len(keys), int(sorted_), location, C.byref(h)))
    return Groups(h)


INTERPOLATIONS = {"nearest": 0, "lower": 1, "higher": 2, "midpoint": 3, "linear": 4}


def agg_groups_quantile(values: Column, quantile: float, groups: Groups, interpolation: str = "nearest") -> Column:
    """Series::agg_quantile on GroupsProxy::Idx (dispatch.rs:115-140) through plhip_agg_groups_quantile."""
    L = _ffi.lib()
    h = C.c_void_p()
    _ffi.check(L.plhip_agg_groups_quantile(C.byref(values.struct()), groups._h, float(quantile), INTERPOLATIONS[interpolation], 0, C.byref(h)))
    try:
        cs = _ffi.ColumnStruct()
        _ffi.check(L.plhip_result_column(h, 0, C.byref(cs)))
        return _host_column(f"{values.name}_quantile_{quantile:.2f}", cs)
    finally:
        L.plhip_result_free(h)


def agg_groups(values: Column | None, kind: str, groups: Groups, ddof: int = 1) -> Column:
    """Series::agg_sum/mean/min/max/first/last/var/std/median/n_unique on GroupsProxy::Idx and GroupsProxy::group_count
    (kind='count') through plhip_agg_groups.  Returns the one output column (host).  `ddof`: var / std."""
    L = _ffi.lib()
    h = C.c_void_p()
    if values is not None and hasattr(values, "rechunk"):
        values = values.rechunk()
    if type(values).__name__ == "Utf8Column":
        # a string column over the groups: first / last are the codes at the groups' first / last rows (agg_first / agg_last,
        # dispatch.rs:48-78,182-212), min / max those of the byte-order ranks (utf8.rs:21-142), decoded on the way out; count and
        # n_unique are numbers and run on the codes (equal strings <=> equal codes)
        from .strings import DecodedColumn, codes_of, ranks_of

        if kind in ("first", "last"):
            codes = codes_of(values)
            out = agg_groups(codes, kind, groups)
            return DecodedColumn(f"{values.name}_{kind}", codes.dictionary.decode(out.to_numpy(), out.valid_mask()))
        if kind in ("min", "max"):
            ranks = ranks_of(values)
            out = agg_groups(ranks, kind, groups)
            return DecodedColumn(f"{values.name}_{kind}", ranks.decode(out.to_numpy(), out.valid_mask()))
        if kind in ("count", "n_unique"):
            out = agg_groups(codes_of(values), kind, groups)
            out.name = f"{values.name}_{kind}"
            return out
        if kind in ("sum", "mean"):  # full-null columns in the reference (dispatch.rs:126,177)
            return DecodedColumn(f"{values.name}_{kind}", [None] * groups.n_groups)
        raise _ffi.PlHipError(4, f"{kind} of the string column {values.name!r} is not on the accelerated path")
    vs = C.byref(values.struct()) if values is not None else None
    code = _ffi.AGG_NAMES[kind] if kind in _ffi.AGG_NAMES else _ffi.AGG_GROUPS_ONLY[kind]
    _ffi.check(L.plhip_agg_groups(vs, code, groups._h, (int(ddof) & 0xFF) << 8, C.byref(h)))
    try:
        cs = _ffi.ColumnStruct()
        _ffi.check(L.plhip_result_column(h, 0, C.byref(cs)))
        return _host_column(f"{values.name if values is not None else 'count'}_{kind}", cs)
    finally:
        L.plhip_result_free(h)
