"""Arrow ingest at the boundary: Arrow IPC / Parquet files -> the chunked columns plhip_group_by_agg_chunked takes.

The reference reads a file into a DataFrame whose columns are ChunkedArrays -- one chunk per record batch / row group
(crates/polars-io/src/ipc/ipc_file.rs, parquet/read.rs) -- and `group_by_helper` rechunks them on the host before it groups
(`df.as_single_chunk_par()`, crates/polars-lazy/src/physical_plan/executors/group_by.rs:66).  Here the chunks stay where the
reader put them (for a memory-mapped IPC file: the file's own pages) and travel to HBM one by one inside the query
(plhip_group_by_agg_chunked lays them end to end on the device); nothing is concatenated on the host.

pyarrow is the Arrow implementation on the host side (as in the plugin boundary's test): an Arrow array IS the boundary's column
-- values buffer, LSB-first validity bitmap, offset, length -- so the mapping below copies nothing for primitive columns.
"""
from __future__ import annotations

import numpy as np

from . import _ffi
from .column import Column


class ChunkedColumn:
    """A column as the reader delivered it: a list of single-chunk Columns of one dtype (ChunkedArray, chunked_array/mod.rs)."""

    def __init__(self, name: str, chunks: list, keep=None):
        if not chunks:
            raise _ffi.PlHipError(6, "a chunked column needs at least one chunk")
        if any(c.dtype != chunks[0].dtype for c in chunks):
            raise _ffi.PlHipError(6, "the chunks of a column differ in dtype")
        self.name = name
        self.chunks = chunks
        self.dtype = chunks[0].dtype
        self.sorted_flag = None
        self.n_categories = None
        self.dictionary = None  # list of str: a dictionary-encoded (Categorical) column's categories
        self._keep = keep       # the Arrow objects whose buffers the chunks alias

    def __len__(self) -> int:
        return sum(len(c) for c in self.chunks)

    @property
    def n_chunks(self) -> int:
        return len(self.chunks)

    def set_categories(self, n: int):
        self.n_categories = int(n)
        return self

    def to_numpy(self):
        return np.concatenate([c.to_numpy() for c in self.chunks])

    def valid_mask(self):
        if all(c.valid_mask() is None for c in self.chunks):
            return None
        return np.concatenate([np.ones(len(c), bool) if c.valid_mask() is None else c.valid_mask() for c in self.chunks])

    def to_list(self) -> list:
        m, v = self.valid_mask(), self.to_numpy().tolist()
        return v if m is None else [x if ok else None for x, ok in zip(v, m.tolist())]

    def rechunk(self) -> Column:
        """One host-side chunk (ChunkedArray::rechunk): for the seams that take single-chunk columns (plhip_group_tuples,
        plhip_agg_groups); the fused query never needs it."""
        if len(self.chunks) == 1:
            c = self.chunks[0]
        else:
            c = Column(self.name, self.to_numpy(), self.valid_mask(), dtype=None if self.dtype == _ffi.BOOL else self.dtype)
        if self.n_categories is not None:
            c.n_categories = self.n_categories
        c.name = self.name
        return c

    def __repr__(self) -> str:
        return f"ChunkedColumn({self.name!r}, len={len(self)}, chunks={self.n_chunks})"


def _np_dtype_of(pa_type):
    import pyarrow as pa

    table = {pa.int8(): np.int8, pa.int16(): np.int16, pa.int32(): np.int32, pa.int64(): np.int64, pa.uint8(): np.uint8,
             pa.uint16(): np.uint16, pa.uint32(): np.uint32, pa.uint64(): np.uint64, pa.float32(): np.float32, pa.float64(): np.float64}
    return table.get(pa_type)


def _primitive_chunk(name: str, arr) -> Column:
    """One Arrow primitive array -> one Column over the SAME buffers (values, validity bitmap, Arrow offset)."""
    import pyarrow as pa

    dt = _np_dtype_of(arr.type)
    n = len(arr)
    if pa.types.is_boolean(arr.type):  # (the Python mirror packs Boolean columns itself: one copy, bits -> bool -> bits)
        vals = np.asarray(arr.fill_null(False).to_numpy(zero_copy_only=False), dtype=bool)
        mask = None if arr.null_count == 0 else ~np.asarray(arr.is_null().to_numpy(zero_copy_only=False), dtype=bool)
        return Column(name, vals, mask)
    if dt is None:
        raise _ffi.PlHipError(4, f"column {name!r}: Arrow type {arr.type} is not on the accelerated path")
    vbuf, dbuf = arr.buffers()[0], arr.buffers()[1]
    if n == 0 or dbuf is None:
        return Column(name, np.empty(0, dtype=dt))
    values = np.frombuffer(dbuf, dtype=dt, count=arr.offset + n)  # the buffer from its start: rows [offset, offset + n) are the array's
    validity = None
    if vbuf is not None and arr.null_count:
        validity = np.frombuffer(vbuf, dtype=np.uint8)
    return Column(name, values, validity, offset=arr.offset, length=n, packed=True)


def column_from_arrow(name: str, chunked):
    """pyarrow.ChunkedArray (or Array) -> ChunkedColumn / Utf8Column.  Primitive and Boolean chunks keep their buffers;
    dictionary<string> columns (a Categorical on the reference side) become their u32 codes under ONE unified dictionary;
    plain string columns are concatenated into one LargeUtf8 column (the string encoder takes a single chunk)."""
    import pyarrow as pa

    if isinstance(chunked, pa.Array):
        chunked = pa.chunked_array([chunked])
    t = chunked.type
    if pa.types.is_dictionary(t):
        if not (pa.types.is_string(t.value_type) or pa.types.is_large_string(t.value_type)):
            raise _ffi.PlHipError(4, f"column {name!r}: only dictionaries of strings are on the accelerated path")
        uni = chunked.unify_dictionaries()
        cats = uni.chunk(0).dictionary.to_pylist() if uni.num_chunks else []
        chunks = [_primitive_chunk(name, c.indices.cast(pa.uint32())) for c in uni.chunks if len(c)]
        col = ChunkedColumn(name, chunks or [Column(name, np.empty(0, np.uint32))], keep=uni)
        col.set_categories(max(len(cats), 1))
        col.dictionary = cats
        return col
    if pa.types.is_string(t) or pa.types.is_large_string(t):
        from .strings import Utf8Column

        arr = pa.concat_arrays([c.cast(pa.large_string()) for c in chunked.chunks]) if chunked.num_chunks else pa.array([], pa.large_string())
        vbuf, obuf, dbuf = arr.buffers()
        offsets = np.frombuffer(obuf, dtype=np.int64) if obuf is not None else np.zeros(1, np.int64)
        data = np.frombuffer(dbuf, dtype=np.uint8) if dbuf is not None else np.empty(0, np.uint8)
        validity = np.frombuffer(vbuf, dtype=np.uint8) if (vbuf is not None and arr.null_count) else None
        col = Utf8Column(name, offsets, data, validity, offset=arr.offset, length=len(arr))
        col._keep = arr
        return col
    chunks = [_primitive_chunk(name, c) for c in chunked.chunks if len(c)]
    if not chunks:
        dt = np.bool_ if pa.types.is_boolean(t) else _np_dtype_of(t)
        if dt is None:
            raise _ffi.PlHipError(4, f"column {name!r}: Arrow type {t} is not on the accelerated path")
        chunks = [Column(name, np.empty(0, dtype=dt))]
    return ChunkedColumn(name, chunks, keep=chunked)


def from_arrow(table):
    """pyarrow.Table / RecordBatch -> DataFrame of chunked columns (no host-side rechunk)."""
    import pyarrow as pa

    from .frame import DataFrame

    if isinstance(table, pa.RecordBatch):
        table = pa.Table.from_batches([table])
    return DataFrame([column_from_arrow(n, table.column(n)) for n in table.column_names])


def read_ipc(path, columns=None, memory_map: bool = True):
    """Arrow IPC file (Feather v2): memory-mapped by default, so a record batch's buffers are the file's pages and the only copy
    of a column is the one into HBM.  Mirrors `pl.read_ipc(path, columns=..., memory_map=True)`."""
    import pyarrow as pa
    import pyarrow.ipc as ipc

    src = pa.memory_map(str(path), "r") if memory_map else pa.OSFile(str(path), "rb")
    table = ipc.open_file(src).read_all()
    if columns is not None:
        table = table.select(list(columns))
    return from_arrow(table)


def read_parquet(path, columns=None):
    """Parquet file: one chunk per row group, as `pl.read_parquet` delivers them (decoding is pyarrow's, on the host)."""
    import pyarrow.parquet as pq

    return from_arrow(pq.read_table(str(path), columns=None if columns is None else list(columns)))


def read_csv(path, columns=None, dtypes=None, categorical=None, null_values=("", "NA")):
    """CSV file -> DataFrame of chunked columns, as the reference's harness starts (`pl.read_csv("G1_1e7_1e2_5_0.csv", dtypes={..})`
    followed by `.cast(pl.Categorical)` of id1 .. id3, run_h2oai_benchmark.py:21-35).  Parsing is pyarrow's, on the host (one chunk
    per block it read); `dtypes` maps column names to numpy dtypes (the harness's Int32 / Float64 overrides), `categorical` names
    the string columns to dictionary-encode -- they reach the group-by as u32 codes under one dictionary and take the dense-code
    path, as a Categorical does."""
    import pyarrow as pa
    import pyarrow.compute as pc
    import pyarrow.csv as pcsv

    types = {n: pa.from_numpy_dtype(np.dtype(t)) for n, t in (dtypes or {}).items()}
    conv = pcsv.ConvertOptions(column_types=types, null_values=list(null_values), strings_can_be_null=True,
                               include_columns=None if columns is None else list(columns))
    table = pcsv.read_csv(str(path), convert_options=conv)
    for n in categorical or []:
        i = table.column_names.index(n)
        table = table.set_column(i, n, pc.dictionary_encode(table.column(n)))
    return from_arrow(table)


class _ScanFrame:
    """What `scan_ipc` / `scan_parquet` hold until `collect()`: the file's schema and a loader.  A column is read when the query asks
    for it (`get_column`), so only the columns a group-by names ever leave the file -- the effect of the reference's projection
    pushdown (crates/polars-plan/src/logical_plan/optimizer/projection_pushdown) for this one query shape."""

    def __init__(self, names, loader):
        self._names, self._loader, self._cols = list(names), loader, {}

    @property
    def columns(self) -> list:
        return list(self._names)

    def get_column(self, name: str):
        if name not in self._names:
            raise _ffi.PlHipError(6, f"column {name!r} not found")
        if name not in self._cols:
            self._cols[name] = self._loader([name]).get_column(name)
        return self._cols[name]

    __getitem__ = get_column

    @property
    def loaded(self) -> list:
        return list(self._cols)

    def materialize(self):
        from .frame import DataFrame

        return DataFrame([self.get_column(n) for n in self._names])


def scan_ipc(path, memory_map: bool = True):
    """`pl.scan_ipc(path)`: a LazyFrame over the file; `.group_by(..).agg(..).collect()` reads the named columns only."""
    import pyarrow as pa
    import pyarrow.ipc as ipc

    from .frame import LazyFrame

    src = pa.memory_map(str(path), "r") if memory_map else pa.OSFile(str(path), "rb")
    names = ipc.open_file(src).schema.names
    return LazyFrame(_ScanFrame(names, lambda cols: read_ipc(path, columns=cols, memory_map=memory_map)))


def scan_parquet(path):
    """`pl.scan_parquet(path)`: as scan_ipc; pyarrow decodes only the requested column chunks of every row group."""
    import pyarrow.parquet as pq

    from .frame import LazyFrame

    names = pq.read_schema(str(path)).names
    return LazyFrame(_ScanFrame(names, lambda cols: read_parquet(path, columns=cols)))
