"""String key columns: Arrow LargeUtf8 on the host side, dense u32 codes + a dictionary on the device side.

Mirrors Utf8Chunked as a group-by key (crates/polars-core/src/frame/group_by/into_groups.rs:232-290): the library hashes
every string, groups (hash, bytes) pairs exactly and returns what a Categorical holds -- physical codes and the strings they
stand for (plhip_utf8_encode, include/plhip.h).  The group-by then runs on the codes (a single string key takes the
dense-code path, perfect.rs:190-227) and the result's key codes are decoded through the dictionary, as a Categorical's
rev-map would.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi
from .column import Column, pack_validity, unpack_validity


class Utf8Column:
    """One single-chunk LargeUtf8 array: int64 offsets (length + 1), uint8 data, optional validity."""

    def __init__(self, name: str, offsets, data, validity=None, *, offset: int = 0, length: int | None = None):
        self.name = name
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        self.data = np.ascontiguousarray(data, dtype=np.uint8)
        self.offset = int(offset)
        self.length = int(len(self.offsets) - 1 - offset if length is None else length)
        if self.offset < 0 or self.length < 0 or self.offset + self.length + 1 > len(self.offsets):
            raise _ffi.PlHipError(6, "offset/length outside the offsets buffer")
        self.validity = None
        if validity is not None:
            v = np.asarray(validity)
            if v.dtype == np.bool_:
                if len(v) != len(self.offsets) - 1:
                    raise _ffi.PlHipError(2, "validity mask length differs from the column's")
                v = pack_validity(v)
            self.validity = np.ascontiguousarray(v, dtype=np.uint8)

    @classmethod
    def from_list(cls, name: str, values) -> "Utf8Column":
        enc = [b"" if v is None else (v if isinstance(v, bytes) else str(v).encode()) for v in values]
        offsets = np.zeros(len(enc) + 1, np.int64)
        np.cumsum([len(b) for b in enc], out=offsets[1:])
        data = np.frombuffer(b"".join(enc), dtype=np.uint8)
        mask = np.array([v is not None for v in values], dtype=bool)
        return cls(name, offsets, data, None if mask.all() else mask)

    def __len__(self) -> int:
        return self.length

    def struct(self) -> _ffi.Utf8ColumnStruct:
        return _ffi.Utf8ColumnStruct(self.offsets.ctypes.data, self.data.ctypes.data if len(self.data) else self.offsets.ctypes.data,
                                     None if self.validity is None else self.validity.ctypes.data, self.offset, self.length, _ffi.HOST, 0)

    def valid_mask(self):
        return None if self.validity is None else unpack_validity(self.validity, self.length, self.offset)

    def to_list(self) -> list:
        m = self.valid_mask()
        o, d = self.offsets[self.offset:self.offset + self.length + 1], self.data.tobytes()
        return [d[o[i]:o[i + 1]].decode() if m is None or m[i] else None for i in range(self.length)]

    def __repr__(self) -> str:
        return f"Utf8Column({self.name!r}, len={self.length})"


class Dictionary:
    """plhip_dictionary_t: owns the device-resident codes and strings of one encoded column."""

    def __init__(self, handle):
        self._h = handle
        self.n_categories = int(_ffi.lib().plhip_dictionary_len(handle))
        self._strings = None

    def strings(self) -> list:
        """The dictionary on the host: entry g is the string all rows with code g hold (None: the null group's entry)."""
        if self._strings is None:
            cs = _ffi.Utf8ColumnStruct()
            _ffi.check(_ffi.lib().plhip_dictionary_strings(self._h, _ffi.HOST, C.byref(cs)))
            g = self.n_categories
            offs = np.ctypeslib.as_array(C.cast(cs.offsets, C.POINTER(C.c_int64)), shape=(g + 1,)).copy()
            total = int(offs[g]) if g else 0
            data = bytes(np.ctypeslib.as_array(C.cast(cs.data, C.POINTER(C.c_uint8)), shape=(max(total, 1),))[:total])
            valid = unpack_validity(np.ctypeslib.as_array(C.cast(cs.validity, C.POINTER(C.c_uint8)), shape=((g + 7) // 8 + 1,)), g, 0)
            self._strings = [data[offs[i]:offs[i + 1]].decode() if valid[i] else None for i in range(g)]
        return self._strings

    def decode(self, codes: np.ndarray, valid: np.ndarray | None = None) -> list:
        s = self.strings()
        return [s[int(c)] if valid is None or valid[i] else None for i, c in enumerate(codes)]

    def __del__(self):
        try:
            if self._h:
                _ffi.lib().plhip_dictionary_free(self._h)
                self._h = None
        except Exception:
            pass


class CodesColumn(Column):
    """The u32 codes of an encoded string column: device memory owned by the Dictionary (kept alive through `dictionary`)."""

    def __init__(self, name: str, cs: _ffi.ColumnStruct, dictionary: Dictionary):
        self.name = name
        self.sorted_flag = None
        self.dictionary = dictionary
        self.n_categories = dictionary.n_categories if dictionary.n_categories > 0 else None
        self.location = _ffi.DEVICE
        self.dtype = _ffi.U32
        self._ptr, self._vptr = cs.values, cs.validity
        self.values = None
        self.validity = None if not cs.validity else True
        self.offset, self.length = 0, int(cs.length)

    def to_numpy(self) -> np.ndarray:
        out = np.empty(self.length, np.uint32)
        if self.length:
            _ffi.check(_ffi.lib().plhip_memcpy(out.ctypes.data, _ffi.HOST, self._ptr, _ffi.DEVICE, out.nbytes))
        return out

    def valid_mask(self):
        if not self._vptr:
            return None
        bits = np.empty((self.length + 7) // 8 + 1, np.uint8)
        _ffi.check(_ffi.lib().plhip_memcpy(bits.ctypes.data, _ffi.HOST, self._vptr, _ffi.DEVICE, (self.length + 7) // 8))
        return unpack_validity(bits, self.length, 0)


def encode(col: Utf8Column) -> CodesColumn:
    """plhip_utf8_encode: the column's dense codes (device-resident, `.dictionary` holds the strings)."""
    L = _ffi.lib()
    handle, cs = C.c_void_p(), _ffi.ColumnStruct()
    st = col.struct()
    _ffi.check(L.plhip_utf8_encode(C.byref(st), C.byref(handle), C.byref(cs)))
    return CodesColumn(col.name, cs, Dictionary(handle))


def codes_of(col: Utf8Column) -> CodesColumn:
    """The column's codes, encoded once per column object (a frame that aggregates a string column twice pays for one encode)."""
    if getattr(col, "_encoded", None) is None:
        col._encoded = encode(col)
    return col._encoded


class RanksColumn(Column):
    """u32 per row: the position of the row's string among the column's DISTINCT strings in byte order (Rust's `str` Ord, which
    Utf8Chunked::agg_min / agg_max fold with, aggregations/utf8.rs:48,110) -- min / max of the strings are min / max of the ranks.
    Device memory of a plhip_take result; the validity is the codes' (owned by the Dictionary, kept alive through `codes`)."""

    def __init__(self, name: str, handle, cs: _ffi.ColumnStruct, codes: CodesColumn, ordered: list):
        self.name = name
        self.sorted_flag = None
        self.n_categories = None
        self._h, self.codes, self.ordered = handle, codes, ordered
        self.location = _ffi.DEVICE
        self.dtype = _ffi.U32
        self._ptr, self._vptr = cs.values, codes._vptr
        self.values = None
        self.validity = None if not codes._vptr else True
        self.offset, self.length = 0, int(codes.length)

    def decode(self, ranks: np.ndarray, valid: np.ndarray | None = None) -> list:
        return [self.ordered[int(r)] if valid is None or valid[i] else None for i, r in enumerate(ranks)]

    def __del__(self):
        try:
            if self._h:
                _ffi.lib().plhip_result_free(self._h)
                self._h = None
        except Exception:
            pass


def ranks_of(col: Utf8Column) -> RanksColumn:
    """Encodes the column (plhip_utf8_encode), orders its dictionary -- G distinct strings, on the host -- and maps every row's code to
    its string's rank ON THE DEVICE (plhip_take of the G-entry rank table with the device-resident codes as indices: the codes are the
    library's own group numbers, all < G, null rows included).  Cached on the column object."""
    if getattr(col, "_ranks", None) is not None:
        return col._ranks
    codes = codes_of(col)
    strs = codes.dictionary.strings()
    order = sorted((i for i, s in enumerate(strs) if s is not None), key=lambda i: strs[i].encode())
    table = np.zeros(max(len(strs), 1), np.uint32)
    table[order] = np.arange(len(order), dtype=np.uint32)
    ordered = [strs[i] for i in order]
    L = _ffi.lib()
    h, cs = C.c_void_p(), _ffi.ColumnStruct()
    if codes.length == 0:  # nothing to gather
        col._ranks = Column(col.name, np.empty(0, np.uint32))
        col._ranks.ordered, col._ranks.decode = ordered, (lambda r, v=None: [])
        return col._ranks
    src = (_ffi.ColumnStruct * 1)(Column("rank", table).struct())
    _ffi.check(L.plhip_take(src, 1, codes._ptr, codes.length, _ffi.DEVICE, _ffi.FLAG_OUTPUT_DEVICE, C.byref(h)))
    _ffi.check(L.plhip_result_column(h, 0, C.byref(cs)))
    col._ranks = RanksColumn(col.name, h, cs, codes, ordered)
    return col._ranks


class DecodedColumn:
    """A result key column of strings (what the frame shows for a Utf8 key)."""

    def __init__(self, name: str, values: list):
        self.name, self._values = name, values

    def __len__(self) -> int:
        return len(self._values)

    def to_list(self) -> list:
        return list(self._values)

    def __repr__(self) -> str:
        return f"DecodedColumn({self.name!r}, utf8, len={len(self._values)})"


def concat(cols: list) -> Utf8Column:
    """The columns' logical slices end to end as one host LargeUtf8 column (the join's key columns share one dictionary)."""
    offs, datas, masks, base, any_null = [np.zeros(1, np.int64)], [], [], 0, False
    for c in cols:
        o = c.offsets[c.offset:c.offset + c.length + 1]
        datas.append(c.data[o[0]:o[-1]])
        offs.append(o[1:] - o[0] + base)
        base += int(o[-1] - o[0])
        m = c.valid_mask()
        any_null |= m is not None
        masks.append(np.ones(c.length, bool) if m is None else m)
    return Utf8Column(cols[0].name, np.concatenate(offs), np.concatenate(datas) if datas else np.zeros(0, np.uint8),
                      np.concatenate(masks) if any_null else None)


def encode_together(left: Utf8Column, right: Utf8Column):
    """Codes of two string columns under ONE dictionary (equal strings of either column get equal codes): views of the encoded
    concatenation.  What a join on string keys needs (the reference hashes both sides' bytes with one RandomState)."""
    both = encode(concat([left, right]))
    import copy

    lc, rc = copy.copy(both), copy.copy(both)
    lc.offset, lc.length = 0, left.length
    rc.offset, rc.length = left.length, right.length
    return lc, rc
