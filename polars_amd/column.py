"""Arrow-layout primitive column: values buffer + optional LSB-first validity bitmap + offset.

Mirrors nano-arrow's PrimitiveArray{values, validity} (crates/nano-arrow/src/array/primitive/mod.rs:48-52)
for the dtypes on the group-by path.  Buffers may be host numpy arrays or device-resident torch tensors
(`torch` is only used for device memory, never for compute).
"""
from __future__ import annotations

import numpy as np

from . import _ffi


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


_TORCH_CODES = None


def _torch_code(t) -> int:
    global _TORCH_CODES
    import torch

    if _TORCH_CODES is None:
        _TORCH_CODES = {
            torch.int8: _ffi.I8, torch.int16: _ffi.I16, torch.int32: _ffi.I32, torch.int64: _ffi.I64,
            torch.uint8: _ffi.U8, torch.float32: _ffi.F32, torch.float64: _ffi.F64,
        }
        for name, code in (("uint16", _ffi.U16), ("uint32", _ffi.U32), ("uint64", _ffi.U64)):
            if hasattr(torch, name):
                _TORCH_CODES[getattr(torch, name)] = code
    if t.dtype not in _TORCH_CODES:
        raise _ffi.PlHipError(4, f"torch dtype {t.dtype} is not supported on the group-by path")
    return _TORCH_CODES[t.dtype]


def pack_validity(mask) -> np.ndarray:
    """bool mask (True = valid) -> Arrow LSB-first bitmap (uint8)."""
    return np.packbits(np.asarray(mask, dtype=bool), bitorder="little")


def unpack_validity(bitmap, length: int, offset: int = 0) -> np.ndarray:
    bits = np.unpackbits(np.asarray(bitmap, dtype=np.uint8), bitorder="little")
    return bits[offset:offset + length].astype(bool)


class Column:
    """One named column.  `values`: 1-D numpy array or torch tensor (host or cuda), contiguous.
    `validity`: None, a bool mask of the same length (host) or an already packed uint8 bitmap
    (numpy or torch, same location as values).  `dtype` overrides the physical dtype code, e.g. to pass a
    torch int32 buffer that holds u32 categorical codes."""

    def __init__(self, name: str, values, validity=None, *, offset: int = 0, length: int | None = None,
                 dtype: int | None = None, packed: bool = False):
        self.name = name
        self.sorted_flag = None  # 'ascending' | 'descending': see set_sorted
        self.n_categories = None  # see set_categories
        if _is_torch(values):
            if values.dim() != 1 or not values.is_contiguous():
                raise _ffi.PlHipError(6, "column buffers must be 1-D and contiguous")
            self.location = _ffi.DEVICE if values.is_cuda else _ffi.HOST
            self.dtype = _torch_code(values) if dtype is None else dtype
            self._ptr = values.data_ptr()
            buf_len = values.numel() * (8 if self.dtype == _ffi.BOOL else 1)  # (a boolean column's buffer is a bitmap: uint8 tensor, dtype=BOOL)
        else:
            values = np.ascontiguousarray(values)
            if values.ndim != 1:
                raise _ffi.PlHipError(6, "column buffers must be 1-D")
            self.location = _ffi.HOST
            buf_len = values.shape[0]
            if values.dtype == np.bool_:  # an Arrow boolean array: the values buffer is a bitmap (PLHIP_BOOL)
                self._bool = values
                values = np.concatenate([np.packbits(values, bitorder="little"), np.zeros(8, np.uint8)])
                self.dtype = _ffi.BOOL
            else:
                self.dtype = _ffi.np_dtype_code(values.dtype) if dtype is None else dtype
            self._ptr = values.ctypes.data
        self.values = values
        self.offset = int(offset)
        self.length = int(buf_len - offset if length is None else length)
        if self.offset < 0 or self.length < 0 or self.offset + self.length > buf_len:
            raise _ffi.PlHipError(6, "offset/length outside the values buffer")
        self._vptr = None
        self.validity = None
        if validity is not None:
            if _is_torch(validity):
                if not packed:
                    validity = validity.to(dtype=__import__("torch").bool).cpu().numpy()
                else:
                    self.validity = validity
                    self._vptr = validity.data_ptr()
            if self.validity is None:
                v = np.asarray(validity)
                if not packed:
                    if v.shape[0] != buf_len:
                        raise _ffi.PlHipError(2, "validity mask length differs from the values buffer")
                    v = pack_validity(v)
                v = np.ascontiguousarray(v, dtype=np.uint8)
                if self.location == _ffi.DEVICE:
                    import torch

                    v = torch.from_numpy(v).to(values.device)
                    self._vptr = v.data_ptr()
                else:
                    self._vptr = v.ctypes.data
                self.validity = v

    # ------------------------------------------------------------------
    def __len__(self) -> int:
        return self.length

    def set_sorted(self, descending: bool = False) -> "Column":
        """Series.set_sorted: declares the column sorted (IsSorted flag, crates/polars-core/src/series/mod.rs); a
        single group-by key with the flag takes the sorted fast path (into_groups.rs:142-150).  Like the
        reference, nobody checks the claim."""
        self.sorted_flag = "descending" if descending else "ascending"
        return self

    def set_categories(self, n_categories: int) -> "Column":
        """Declares the column the physical u32 codes of a Categorical with a local rev-map of `n_categories` entries:
        every valid code is < n_categories.  A single group-by key with this knowledge takes the dense-code path, the
        counterpart of group_tuples_perfect (crates/polars-core/src/frame/group_by/perfect.rs:190-227).  A code outside
        the promise costs time, never correctness."""
        if self.dtype != _ffi.U32:
            raise _ffi.PlHipError(4, "categorical codes are u32")
        self.n_categories = int(n_categories)
        return self

    def struct(self) -> _ffi.ColumnStruct:
        return _ffi.ColumnStruct(self._ptr, self._vptr, self.offset, self.length, self.dtype, self.location)

    def to_numpy(self) -> np.ndarray:
        """Values of the logical slice as a host numpy array (nulls keep whatever the buffer holds)."""
        if self.dtype == _ffi.BOOL:
            return self._bool[self.offset:self.offset + self.length]
        v = self.values
        if _is_torch(v):
            v = v.cpu().numpy()
        v = v[self.offset:self.offset + self.length]
        want = _ffi.code_np_dtype(self.dtype)
        return v.view(want) if v.dtype != want and v.dtype.itemsize == want.itemsize else v

    def valid_mask(self) -> np.ndarray | None:
        if self.validity is None:
            return None
        b = self.validity
        if _is_torch(b):
            b = b.cpu().numpy()
        return unpack_validity(b, self.length, self.offset)

    def to_list(self) -> list:
        vals = self.to_numpy().tolist()
        m = self.valid_mask()
        if m is None:
            return vals
        return [v if ok else None for v, ok in zip(vals, m)]

    def __repr__(self) -> str:
        return f"Column({self.name!r}, {_ffi.DTYPE_NAMES[self.dtype]}, len={self.length})"


def column_from_list(name: str, data: list, dtype) -> Column:
    """Builds a host column from a Python list that may contain None (null)."""
    mask = np.array([x is not None for x in data], dtype=bool)
    vals = np.array([0 if x is None else x for x in data], dtype=dtype)
    return Column(name, vals, None if mask.all() else mask)
