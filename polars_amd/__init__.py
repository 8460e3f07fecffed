"""polars_amd — MI355X-native hash group-by aggregation (the Polars `group_by().agg()` hot path).

The compute lives in libplhip.so (hand-written gfx950 HIP kernels behind the C ABI of include/plhip.h);
this package is the thin host-side mirror of the reference API.  There is no CPU fallback.
"""
from ._ffi import ComputeError, PlHipError, ShapeMismatch  # noqa: F401
from .column import Column, column_from_list  # noqa: F401
from .frame import (DataFrame, Expr, GroupBy, Groups, LazyFrame, agg_groups, col, count, group_by_agg_raw,  # noqa: F401
                    group_tuples,
                    last_kernel_breakdown, last_kernel_ms, last_strategy, set_strategy)
from .frame import max_ as max, mean_ as mean, min_ as min, sum_ as sum  # noqa: F401,A001
from .io import ChunkedColumn, from_arrow, read_csv, read_ipc, read_parquet, scan_ipc, scan_parquet  # noqa: F401

__version__ = "0.1.0"
