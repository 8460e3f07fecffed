"""The sharded path behind the C ABI (plhip_group_by_agg_sharded): local pass -> pack -> exchange -> regroup -> finalize in
ONE library call, the counterpart of PartitionGroupByExec (group_by_partitioned.rs:273-361) across GPUs.
  - RCCL transport on the one GPU of the box: a communicator of world size 1 (the sends and receives go to the rank
    itself: ncclGroupStart; ncclSend; ncclRecv; ncclGroupEnd run for real), both protocols, against the oracle;
  - two and three ranks on ONE GPU over the host transport (gloo carries the bytes, everything else is the library):
    scripts/rehearse_ranks.py under torch.distributed.run, whole result against the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from polars_amd import _ffi
from polars_amd.column import Column
from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def rccl_world1():
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", device_id=torch.device("cuda:0"), rank=0, world_size=1)
    yield
    dist.destroy_process_group()


def _frame(res):
    return [(c.t.cpu().numpy().view(_ffi.code_np_dtype(c.code)), None if c.mask is None else c.mask.cpu().numpy()) for c in res]


@pytest.mark.parametrize("protocol", ["slots", "exact", "hint_too_small", "no_hint"])
def test_sharded_rccl_world1_vs_oracle(oracle, rccl_world1, protocol):
    import torch

    from polars_amd import distributed as pd_

    rng = np.random.default_rng(31)
    n, g = 1_000_000, 20_000
    k = rng.integers(-g // 2, g // 2, n).astype(np.int32)
    km = rng.random(n) >= 0.02
    v1 = rng.integers(1, 6, n).astype(np.int32)
    v3 = np.round(rng.uniform(0, 100, n), 6)  # (mixed signs: tests/test_edges_gpu.py, with the bound that applies there)
    v3m = rng.random(n) >= 0.05
    keys = [Column("k", k, km)]
    vals = [Column("v1", v1), Column("v3", v3, v3m)]
    aggs = [(0, "sum"), (1, "mean"), (1, "min"), (1, "max"), (0, "count"), (1, "sum"), (0, "mean")]
    eng = pd_.HipEngine(torch.device("cuda:0"))
    eng.use_slots = protocol != "exact"
    hint = {"slots": g + 1, "exact": g + 1, "hint_too_small": 50, "no_hint": 0}[protocol]
    res = pd_.group_by_agg_sharded(keys, vals, aggs, device=torch.device("cuda:0"), engine=eng, n_groups_hint=hint, force_exchange=True)
    want, _ = oracle.group_by_agg(keys, vals, aggs)
    util.assert_frames_match(_frame(res), util.oracle_to_np(want), 1, what=f"sharded, RCCL world 1, {protocol}")
    assert eng.last_protocol.startswith("slots" if protocol == "slots" else "exact"), eng.last_protocol
    assert eng.last_shuffle_bytes > 0 and eng.last_partial_groups == len(want[0][1])
    assert eng.last_host_syncs <= (4 if protocol == "slots" else 8)
    assert set(eng.last_phase_ms) == {"partial_pass", "pack", "exchange", "merge", "finalize"}


def test_sharded_rccl_world1_dense_codes_and_device_inputs(oracle, rccl_world1):
    """The bench's shape: a Categorical key's codes (dense-code local pass), device-resident torch columns."""
    import torch

    from polars_amd import distributed as pd_

    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    n, ncat = 4_000_000, 40_000
    kt = torch.randint(0, ncat, (n,), generator=gen, device=dev, dtype=torch.int32)
    v1 = torch.randint(1, 6, (n,), generator=gen, device=dev, dtype=torch.int32)
    v3 = torch.rand((n,), generator=gen, device=dev, dtype=torch.float64) * 100
    keys = [Column("id3", kt, dtype=_ffi.U32).set_categories(ncat)]
    vals = [Column("v1", v1), Column("v3", v3)]
    aggs = [(0, "sum"), (1, "mean")]
    eng = pd_.HipEngine(dev)
    eng.dense_keys = True
    res = pd_.group_by_agg_sharded(keys, vals, aggs, device=dev, engine=eng, n_groups_hint=ncat, force_exchange=True)
    assert eng.last_partial_strategy == "dense", eng.last_partial_strategy
    assert eng.last_protocol.startswith("dense") and eng.last_host_syncs == 1, (eng.last_protocol, eng.last_host_syncs)
    hk = [Column("id3", kt.cpu().numpy().view(np.uint32))]
    hv = [Column("v1", v1.cpu().numpy()), Column("v3", v3.cpu().numpy())]
    want, _ = oracle.group_by_agg(hk, hv, aggs)
    util.assert_frames_match(_frame(res), util.oracle_to_np(want), 1, what="sharded, dense local pass")


@pytest.mark.parametrize("case", ["plain", "nullable_key"])
def test_sharded_dense_exchange_beyond_eight_ranges(oracle, rccl_world1, case):
    """More codes than eight LDS ranges hold (config 4's shards: 1e7 ids): the local pass range-partitions the rows first
    (strategy dense_part) and the exchange stays the dense one; a nullable key is what this mode does not take: the rank flags
    its slot headers and every rank falls back to the hashed protocol together."""
    import torch

    from polars_amd import distributed as pd_

    rng = np.random.default_rng(43)
    n, ncat = 3_000_000, 600_000
    k = rng.integers(0, ncat, n).astype(np.uint32)
    km = rng.random(n) >= 0.02 if case == "nullable_key" else None
    v1 = rng.integers(-5, 6, n).astype(np.int32)
    v3 = np.round(rng.uniform(0, 100, n), 6)
    v3m = rng.random(n) >= 0.05
    keys = [Column("k", k, km).set_categories(ncat)]
    vals = [Column("v1", v1), Column("v3", v3, v3m)]
    aggs = [(0, "sum"), (1, "mean"), (1, "min"), (1, "max"), (0, "count"), (1, "sum")]
    eng = pd_.HipEngine(torch.device("cuda:0"))
    eng.dense_keys = True
    res = pd_.group_by_agg_sharded(keys, vals, aggs, device=torch.device("cuda:0"), engine=eng, n_groups_hint=ncat, force_exchange=True)
    want, _ = oracle.group_by_agg(keys, vals, aggs)
    util.assert_frames_match(_frame(res), util.oracle_to_np(want), 1, what=f"sharded dense exchange, partition mode, {case}")
    if case == "plain":
        assert eng.last_partial_strategy == "dense_part", eng.last_partial_strategy
        assert eng.last_protocol.startswith("dense") and eng.last_host_syncs == 1, (eng.last_protocol, eng.last_host_syncs)
    else:
        assert not eng.last_protocol.startswith("dense"), eng.last_protocol


@pytest.mark.parametrize("case", ["nullable", "code_beyond_hint", "misaligned", "tiny", "hashed_forced"])
def test_sharded_dense_exchange_world1(oracle, rccl_world1, case, monkeypatch):
    """Protocol 3 (owners by code range, raw partial arrays through ncclSend / ncclRecv) against the oracle: null key, nullable
    values, every aggregation; what breaks the dense promise on this rank falls back to the hashed protocols."""
    import torch

    from polars_amd import distributed as pd_

    rng = np.random.default_rng(41)
    n, ncat = (700, 50) if case == "tiny" else (1_500_000, 30_011)
    k = rng.integers(0, ncat, n + 1).astype(np.uint32)
    km = rng.random(n + 1) >= 0.02
    v1 = rng.integers(-5, 6, n + 1).astype(np.int32)
    v3 = np.round(rng.uniform(0, 100, n + 1), 6)
    v3m = rng.random(n + 1) >= 0.05
    if case == "code_beyond_hint":
        k[n // 2] = ncat + 5
    sl = slice(1, n + 1) if case == "misaligned" else slice(0, n)
    if case == "hashed_forced":
        monkeypatch.setenv("PLHIP_NO_DENSE_EXCHANGE", "1")
    keys = [Column("k", k[sl], km[sl]).set_categories(ncat)]
    vals = [Column("v1", v1[sl]), Column("v3", v3[sl], v3m[sl])]
    if case == "misaligned":  # (host columns are staged into fresh allocations: only device-resident buffers can be misaligned)
        dev = torch.device("cuda:0")
        dk = [Column("k", torch.from_numpy(k.view(np.int32)).to(dev)[1:], dtype=_ffi.U32).set_categories(ncat)]
        dv = [Column("v1", torch.from_numpy(v1).to(dev)[1:]), Column("v3", torch.from_numpy(v3).to(dev)[1:])]
        keys, vals = [Column("k", k[sl])], [Column("v1", v1[sl]), Column("v3", v3[sl])]
    else:
        dk, dv = keys, vals
    aggs = [(0, "sum"), (1, "mean"), (1, "min"), (1, "max"), (0, "count"), (1, "sum"), (0, "mean"), (0, "min")]
    eng = pd_.HipEngine(torch.device("cuda:0"))
    eng.dense_keys = True
    res = pd_.group_by_agg_sharded(dk, dv, aggs, device=torch.device("cuda:0"), engine=eng, n_groups_hint=ncat, force_exchange=True)
    want, _ = oracle.group_by_agg(keys, vals, aggs)
    util.assert_frames_match(_frame(res), util.oracle_to_np(want), 1, what=f"sharded dense exchange, {case}")
    if case in ("code_beyond_hint", "misaligned", "hashed_forced"):
        assert not eng.last_protocol.startswith("dense"), eng.last_protocol
    else:
        assert eng.last_protocol.startswith("dense") and eng.last_host_syncs == 1, (eng.last_protocol, eng.last_host_syncs)


@pytest.mark.parametrize("hint", [0, 4000])
def test_sharded_boolean_columns(oracle, rccl_world1, hint):
    """Boolean key and value columns through plhip_group_by_agg_sharded: lowered with the reference's casts on every rank (key ->
    UInt8, into_groups.rs:215-230; values -> IdxSize, boolean.rs:107-109), partials exchanged and finalized as for any integer
    column, Boolean outputs (the key, min, max) packed back into bitmaps.  Against the oracle on the cast columns."""
    import torch

    from polars_amd import distributed as pd_

    rng = np.random.default_rng(61 + hint)
    n = 400_000
    kb, kbm = rng.random(n) < 0.4, rng.random(n) >= 0.05
    k2 = rng.integers(0, 900, n).astype(np.int32)
    vb, vbm = rng.random(n) < 0.7, rng.random(n) >= 0.2
    w = rng.integers(-50, 50, n).astype(np.int64)
    keys = [Column("kb", kb, kbm), Column("k2", k2)]
    vals = [Column("vb", vb, vbm), Column("w", w)]
    aggs = [(0, "sum"), (0, "mean"), (0, "min"), (0, "max"), (0, "count"), (1, "sum"), (1, "max")]
    eng = pd_.HipEngine(torch.device("cuda:0"))
    res = pd_.group_by_agg_sharded(keys, vals, aggs, device=torch.device("cuda:0"), engine=eng, n_groups_hint=hint, force_exchange=True)
    assert [c.code for c in res] == [_ffi.BOOL, _ffi.I32, _ffi.U32, _ffi.F64, _ffi.BOOL, _ffi.BOOL, _ffi.U32, _ffi.I64, _ffi.I64]
    ok = [Column("kb", kb.astype(np.uint8), kbm), Column("k2", k2)]
    ov = [Column("vb", vb.astype(np.uint32), vbm), Column("w", w)]
    want, _ = oracle.group_by_agg(ok, ov, aggs)
    got = []
    for i, c in enumerate(res):
        v = c.t.cpu().numpy()
        v = v.astype(np.uint8 if i == 0 else np.uint32) if c.code == _ffi.BOOL else v.view(_ffi.code_np_dtype(c.code))
        got.append((v, None if c.mask is None else c.mask.cpu().numpy()))
    util.assert_frames_match(got, util.oracle_to_np(want), 2, what=f"sharded, Boolean columns, hint {hint}")


def test_sharded_torch_transport_fallback(oracle, rccl_world1, monkeypatch):
    """When the library's own RCCL communicator cannot be made, every rank falls back to the group's all_to_all_single
    (device tensors) behind the same entry point: forced here."""
    import torch

    from polars_amd import distributed as pd_

    monkeypatch.setenv("PLHIP_EXCHANGE_TRANSPORT", "torch")
    monkeypatch.setattr(pd_.HipEngine, "_comms", {})
    rng = np.random.default_rng(77)
    n = 300_000
    keys = [Column("k", rng.integers(0, 5000, n).astype(np.int64))]
    vals = [Column("v", rng.integers(-9, 9, n).astype(np.int32)), Column("w", np.round(rng.uniform(0, 1, n), 6))]
    aggs = [(0, "sum"), (1, "mean"), (0, "max"), (0, "count")]
    eng = pd_.HipEngine(torch.device("cuda:0"))
    for hint in (5000, 0):
        res = pd_.group_by_agg_sharded(keys, vals, aggs, device=torch.device("cuda:0"), engine=eng, n_groups_hint=hint, force_exchange=True)
        want, _ = oracle.group_by_agg(keys, vals, aggs)
        util.assert_frames_match(_frame(res), util.oracle_to_np(want), 1, what=f"sharded, torch transport, hint {hint}")
    assert eng.transport.startswith("torch.distributed")
    monkeypatch.setattr(pd_.HipEngine, "_comms", {})


def test_sharded_argument_errors(rccl_world1):
    import ctypes as C

    L = _ffi.lib()
    k = Column("k", np.zeros(8, np.int32))
    karr = (_ffi.ColumnStruct * 1)(k.struct())
    aarr = (_ffi.AggStruct * 1)(_ffi.AggStruct(-1, _ffi.AGG_COUNT))
    out = C.c_void_p()
    assert L.plhip_group_by_agg_sharded(None, karr, 1, karr, 0, aarr, 1, 0, None, C.byref(out)) == 6  # PLHIP_ERR_INVALID: no communicator
    assert b"comm" in L.plhip_last_error()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_ranks_on_one_gpu(world):
    """`world` processes share the GPU; the exchange's bytes travel over gloo through the library's host transport."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "scripts", "rehearse_ranks.py")]
    env = dict(os.environ)
    env.pop("MASTER_PORT", None)
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert p.stdout.count("ok:") == 12, p.stdout[-3000:]
