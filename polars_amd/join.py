"""Hash join: the row-index pairs (plhip_hash_join / plhip_hash_join_multi, include/plhip.h) and the payload gather behind them
(plhip_take: the `take_unchecked` / `_take_opt_chunked_unchecked` calls that feed DataFrame::_finish_join in the reference,
crates/polars-ops/src/frame/join/hash_join/mod.rs:105-201).  `Join` keeps the pairs in HBM between the two steps; string payload
columns and the outer join's key coalescing are gathered on the host.

Mirrors crates/polars-ops/src/frame/join/hash_join/single_keys_inner.rs / single_keys_left.rs: pair order, build side
and null == null semantics are the reference's."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi
from .column import Column

NONE = 0xFFFFFFFF


def join_indices(left, right, how: str = "inner") -> tuple[np.ndarray, np.ndarray]:
    """(left_idx, right_idx) as host u32 arrays; right_idx == NONE where a left-join row has no match; semi / anti: the kept left
    rows and an empty right array."""
    L = _ffi.lib()
    h = {"inner": 0, "left": 1, "semi": 2, "anti": 3, "outer": 4}.get(how)
    if h is None:
        raise _ffi.PlHipError(6, f"join kind {how!r} is not on the accelerated path (inner, left, semi, anti, outer)")
    lefts = list(left) if isinstance(left, (list, tuple)) else [left]
    rights = list(right) if isinstance(right, (list, tuple)) else [right]
    if len(lefts) != len(rights) or not lefts:
        raise _ffi.PlHipError(6, "the two sides of a join need the same number of key columns")
    for c in lefts + rights:
        if c.location == _ffi.DEVICE and type(c.values).__module__.startswith("torch"):
            import torch

            torch.cuda.current_stream(c.values.device).synchronize()
            break
    handle = C.c_void_p()
    if len(lefts) == 1:
        ls, rs = lefts[0].struct(), rights[0].struct()
        _ffi.check(L.plhip_hash_join(C.byref(ls), C.byref(rs), h, _ffi.HOST, C.byref(handle)))
    else:  # several key columns: hash_join/multiple_keys.rs
        la = (_ffi.ColumnStruct * len(lefts))(*[c.struct() for c in lefts])
        ra = (_ffi.ColumnStruct * len(rights))(*[c.struct() for c in rights])
        _ffi.check(L.plhip_hash_join_multi(la, ra, len(lefts), h, _ffi.HOST, C.byref(handle)))
    try:
        m = L.plhip_join_num_rows(handle)
        lp, rp, loc = C.c_void_p(), C.c_void_p(), C.c_int32()
        _ffi.check(L.plhip_join_buffers(handle, C.byref(lp), C.byref(rp), C.byref(loc)))
        if m == 0:
            return np.zeros(0, np.uint32), np.zeros(0, np.uint32)
        li = np.ctypeslib.as_array(C.cast(lp, C.POINTER(C.c_uint32)), shape=(m,)).copy()
        ri = np.ctypeslib.as_array(C.cast(rp, C.POINTER(C.c_uint32)), shape=(m,)).copy() if rp.value else np.zeros(0, np.uint32)
        return li, ri
    finally:
        L.plhip_join_free(handle)


class Join:
    """The pairs of one join, device-resident (plhip_hash_join(..., PLHIP_DEVICE)); `take` gathers payload columns with them on the
    device, `indices()` copies them to the host (strings, tests)."""

    def __init__(self, left, right, how: str = "inner"):
        L = _ffi.lib()
        h = {"inner": 0, "left": 1, "semi": 2, "anti": 3, "outer": 4}.get(how)
        if h is None:
            raise _ffi.PlHipError(6, f"join kind {how!r} is not on the accelerated path (inner, left, semi, anti, outer)")
        lefts = list(left) if isinstance(left, (list, tuple)) else [left]
        rights = list(right) if isinstance(right, (list, tuple)) else [right]
        if len(lefts) != len(rights) or not lefts:
            raise _ffi.PlHipError(6, "the two sides of a join need the same number of key columns")
        self.how = how
        self._h = C.c_void_p()
        if len(lefts) == 1:
            ls, rs = lefts[0].struct(), rights[0].struct()
            _ffi.check(L.plhip_hash_join(C.byref(ls), C.byref(rs), h, _ffi.DEVICE, C.byref(self._h)))
        else:
            la = (_ffi.ColumnStruct * len(lefts))(*[c.struct() for c in lefts])
            ra = (_ffi.ColumnStruct * len(rights))(*[c.struct() for c in rights])
            _ffi.check(L.plhip_hash_join_multi(la, ra, len(lefts), h, _ffi.DEVICE, C.byref(self._h)))
        self.m = int(L.plhip_join_num_rows(self._h))
        lp, rp, loc = C.c_void_p(), C.c_void_p(), C.c_int32()
        _ffi.check(L.plhip_join_buffers(self._h, C.byref(lp), C.byref(rp), C.byref(loc)))
        self._lp, self._rp = lp.value, rp.value

    def close(self):
        if self._h:
            _ffi.lib().plhip_join_free(self._h)
            self._h = None

    __del__ = close

    def indices(self) -> tuple[np.ndarray, np.ndarray]:
        out = []
        for p_ in (self._lp, self._rp):
            a = np.zeros(self.m if p_ else 0, np.uint32)
            if len(a):
                _ffi.check(_ffi.lib().plhip_memcpy(a.ctypes.data, _ffi.HOST, p_, _ffi.DEVICE, a.nbytes))
            out.append(a)
        return out[0], out[1]

    def take(self, cols: list, side: str, names: list[str] | None = None) -> list:
        """Fixed-width columns of one side gathered on the device in ONE plhip_take call (rows with no partner come back null)."""
        names = names or [c.name for c in cols]
        ptr = self._lp if side == "left" else self._rp
        if not cols:
            return []
        L = _ffi.lib()
        # which side may lack a partner: the right one of a left join, either one of an outer join
        nullable = self.how == "outer" or (self.how == "left" and side == "right")
        arr = (_ffi.ColumnStruct * len(cols))(*[c.struct() for c in cols])
        h = C.c_void_p()
        _ffi.check(L.plhip_take(arr, len(cols), ptr, self.m, _ffi.DEVICE, _ffi.TAKE_NULLABLE_IDX if nullable else 0, C.byref(h)))
        try:
            from .frame import _host_column

            out = []
            for i, n in enumerate(names):
                cs = _ffi.ColumnStruct()
                _ffi.check(L.plhip_result_column(h, i, C.byref(cs)))
                c = _host_column(n, cs)
                m = c.valid_mask()
                out.append(Column(n, c.to_numpy().copy(), None if m is None or m.all() else m.copy()))
            return out
        finally:
            L.plhip_result_free(h)


def take(col, idx: np.ndarray, name: str | None = None):
    """Gathers a column's rows; idx == NONE yields null (the right side of a left join)."""
    name = name or col.name
    miss = idx == NONE
    safe = np.where(miss, 0, idx).astype(np.int64)
    if type(col).__name__ in ("Utf8Column", "DecodedColumn"):
        from .strings import DecodedColumn

        vals = col.to_list()
        return DecodedColumn(name, [None if m else vals[i] for i, m in zip(safe.tolist(), miss.tolist())])
    if len(col) == 0:
        return Column(name, np.zeros(len(idx), _ffi.code_np_dtype(col.dtype)), ~miss if len(idx) else None)
    v = col.to_numpy()[safe]
    m = col.valid_mask()
    valid = ~miss if m is None else (m[safe] & ~miss)
    return Column(name, v, None if valid.all() else valid)
