"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/plhip.h declares, and
refuses to compute without a GPU (there is no CPU fallback)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

from polars_amd import _ffi

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols(header="plhip.h"):
    text = (ROOT / "include" / header).read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(plhip_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built():
    assert _ffi.LIB_PATH.exists(), "build with __graft_entry__.build()"


def test_exports_every_declared_symbol():
    lib = _ffi.lib()
    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in plhip.h but not exported"
    assert sorted(_ffi.EXPORTED_SYMBOLS) == declared


def test_exports_the_reference_plugin_symbols():
    # include/plhip_plugin.h: the FfiPlugin boundary (crates/polars-plan/src/dsl/function_expr/plugin.rs:33-75)
    lib = _ffi.lib()
    declared = _declared_symbols("plhip_plugin.h")
    assert declared == ["plhip_plugin_group_by", "plhip_plugin_group_by_field"]
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in plhip_plugin.h but not exported"


def test_abi_version():
    assert _ffi.lib().plhip_abi_version() == 1


def test_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from polars_amd import DataFrame, col

    df = DataFrame({"k": np.arange(10, dtype=np.int32), "v": np.ones(10)})
    with pytest.raises(_ffi.PlHipError) as ei:
        df.group_by("k").agg(col("v").sum())
    assert ei.value.status == 7  # PLHIP_ERR_NO_DEVICE


def test_argument_errors_match_reference_kinds():
    # group_by/mod.rs:59-62: zero keys -> ComputeError, raised before any device work
    L = _ffi.lib()
    out = C.c_void_p()
    rc = L.plhip_group_by_agg(None, 0, None, 0, None, 0, 0, C.byref(out))
    assert rc == 1 and b"at least one key" in L.plhip_last_error()
    # group_by/mod.rs:67-71: key length mismatch -> ShapeMismatch
    a = np.zeros(4, dtype=np.int32)
    b = np.zeros(5, dtype=np.int32)
    keys = (_ffi.ColumnStruct * 2)(
        _ffi.ColumnStruct(a.ctypes.data, None, 0, 4, _ffi.I32, 0), _ffi.ColumnStruct(b.ctypes.data, None, 0, 5, _ffi.I32, 0))
    rc = L.plhip_group_by_agg(keys, 2, None, 0, None, 0, 0, C.byref(out))
    assert rc == 2


def test_missing_rccl_library_is_a_clean_error():
    """with no loadable librccl.so the communicator entry points must answer PLHIP_ERR_NO_DEVICE with a
    message (what HipEngine.communicator falls back on), not crash while building that message.  A fresh process: the
    library resolves RCCL once."""
    import subprocess
    import sys

    code = (
        "import ctypes as C, sys\n"
        f"sys.path.insert(0, {str(ROOT)!r})\n"
        "from polars_amd import _ffi\n"
        "L = _ffi.lib()\n"
        "buf = (C.c_uint8 * _ffi.COMM_ID_BYTES)()\n"
        "rc = L.plhip_comm_unique_id(buf)\n"
        "msg = L.plhip_last_error().decode()\n"
        "print(rc, '|', msg)\n"
        "rc2 = L.plhip_comm_unique_id(buf)\n"
        "assert rc == 7 and rc2 == 7, (rc, rc2)\n"
        "assert 'librccl.so not found' in msg and 'no-such-dir' in msg, msg\n")
    env = dict(__import__("os").environ, PLHIP_RCCL_LIBRARY="/no-such-dir/librccl.so.1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
