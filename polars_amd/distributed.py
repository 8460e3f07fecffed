"""Row-sharded group-by over several GPUs: one process per GPU, one exchange step.

The reference already does this across threads (PartitionGroupByExec::run_partitions,
crates/polars-lazy/src/physical_plan/executors/group_by_partitioned.rs:66-113, 320-360): contiguous row
split -> local group-by with mergeable partials (AggregationExpr::evaluate_partitioned,
expressions/aggregation.rs:283-365: mean = (sum, valid count)) -> concat -> regroup -> finalize
(aggregation.rs:367-483: sum of sums / counts, min of mins, mean = sum / count).  Here the "concat" is ONE
all-to-all (RCCL over xGMI; torch.distributed backend "nccl") of partial records bucketed by
owner(key) = high bits of the key hash % world, so every rank regroups a disjoint share of the keys and the
final result is the concatenation of the ranks' frames.  A heavy key costs one record per rank, whatever its
row count, so the exchange stays balanced under skew.

torch is plumbing here (device buffers, sort / gather of the partial records, the collective); both
group-by passes run in libplhip.so.  `engine` exists so that the CPU tests can drive the same protocol over
gloo with a stand-in for the two passes.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import _ffi
from .column import Column

NULL_HASH = 0x9E3779B97F4A7C15  # any constant: null keys only need to agree on an owner


@dataclass
class Col:
    """One column of a (partial) frame: `t` holds the bits (u32/u64 data sit in int32/int64 tensors),
    `code` the plhip dtype, `mask` a bool tensor (True = valid) or None."""
    name: str
    t: object
    code: int
    mask: object = None


_TORCH_OF_CODE = None


def _torch_dtype(code: int):
    global _TORCH_OF_CODE
    import torch

    if _TORCH_OF_CODE is None:
        _TORCH_OF_CODE = {_ffi.I8: torch.int8, _ffi.I16: torch.int16, _ffi.I32: torch.int32, _ffi.I64: torch.int64,
                          _ffi.U8: torch.uint8, _ffi.U16: torch.int16, _ffi.U32: torch.int32, _ffi.U64: torch.int64,
                          _ffi.F32: torch.float32, _ffi.F64: torch.float64}
    return _TORCH_OF_CODE[code]


def sum_out_code(code: int) -> int:
    """Output dtype of SUM (series/implementations/mod.rs:167-174): narrow ints widen to i64."""
    return _ffi.I64 if code in (_ffi.I8, _ffi.I16, _ffi.U8, _ffi.U16) else code


def mean_out_code(code: int) -> int:
    return _ffi.F32 if code == _ffi.F32 else _ffi.F64


# ---------------------------------------------------------------------------------------------------
# the two local passes on libplhip.so
# ---------------------------------------------------------------------------------------------------
class _FrameOwner:
    """Keeps a plhip_result_t alive for the tensors that alias its buffers."""

    def __init__(self, handle):
        self.handle = handle

    def __del__(self):
        try:
            _ffi.lib().plhip_result_free(self.handle)
        except Exception:
            pass


class _DevBuf:
    def __init__(self, ptr: int, nbytes: int, owner):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
        self._owner = owner


class _HostTransport:
    """plhip_alltoall_fn / plhip_alltoallv_fn over a torch.distributed group that cannot move device memory (gloo): the
    buffers are copied to the host with plhip_memcpy (the library has drained its stream before it calls), exchanged
    with dist.all_to_all_single, and copied back.  Rehearsal transport: several ranks on one GPU."""

    def __init__(self, group, world: int, device=None):
        # device = None: stage through host memory (gloo); a torch device: the group moves device memory itself (NCCL) and
        # the buffers are copied device-to-device into torch-owned tensors -- the fallback when the library's own RCCL
        # communicator cannot be made
        self.group, self.world, self.device = group, world, device
        self.a2a = _ffi.ALLTOALL_FN(self._a2a)
        self.a2av = _ffi.ALLTOALLV_FN(self._a2av)

    def _buf(self, n):
        import torch

        return torch.empty(max(n, 1), dtype=torch.uint8, device=self.device if self.device is not None else "cpu")[:n]

    def _loc(self):
        return _ffi.DEVICE if self.device is not None else _ffi.HOST

    def _done(self):
        if self.device is not None:
            import torch

            torch.cuda.current_stream(self.device).synchronize()

    def _a2a(self, user, send, recv, bytes_per_peer):
        try:
            import torch
            import torch.distributed as dist

            n = int(bytes_per_peer) * self.world
            hs, hr = self._buf(n), self._buf(n)
            _ffi.check(_ffi.lib().plhip_memcpy(hs.data_ptr(), self._loc(), send, _ffi.DEVICE, n))
            dist.all_to_all_single(hr, hs, group=self.group)
            self._done()
            _ffi.check(_ffi.lib().plhip_memcpy(recv, _ffi.DEVICE, hr.data_ptr(), self._loc(), n))
            return 0
        except Exception:  # never unwind into C
            import traceback

            traceback.print_exc()
            return 1

    def _a2av(self, user, send, send_bytes, recv, recv_bytes):
        try:
            import torch
            import torch.distributed as dist

            sb = [int(send_bytes[p]) for p in range(self.world)]
            rb = [int(recv_bytes[p]) for p in range(self.world)]
            hs, hr = self._buf(sum(sb)), self._buf(sum(rb))
            if sum(sb):
                _ffi.check(_ffi.lib().plhip_memcpy(hs.data_ptr(), self._loc(), send, _ffi.DEVICE, sum(sb)))
            dist.all_to_all_single(hr, hs, output_split_sizes=rb, input_split_sizes=sb, group=self.group)
            self._done()
            if sum(rb):
                _ffi.check(_ffi.lib().plhip_memcpy(recv, _ffi.DEVICE, hr.data_ptr(), self._loc(), sum(rb)))
            return 0
        except Exception:
            import traceback

            traceback.print_exc()
            return 1


class HipEngine:
    """Local passes through the C ABI with device-resident inputs and outputs."""

    def __init__(self, device):
        self.device = device

    def _frame(self, handle, cols, names) -> list[Col]:
        """Wraps a device-resident result frame as torch tensors.  Zero-copy through __cuda_array_interface__
        (the frame is freed when the last tensor over it dies); if this torch build cannot alias foreign
        device memory the columns are copied device-to-device instead."""
        import torch

        owner = _FrameOwner(handle)
        out = []
        for name, cs in zip(names, cols):
            n = cs.length
            if cs.dtype == _ffi.BOOL:  # a Boolean output column is a bitmap: unpacked into a bool tensor
                t = _unpack_bits(self._alias(cs.values, (n + 7) // 8, owner), n) if n else torch.empty(0, dtype=torch.bool, device=self.device)
            else:
                dt = _torch_dtype(cs.dtype)
                t = self._alias(cs.values, n * torch.empty(0, dtype=dt).element_size(), owner).view(dt) if n else torch.empty(0, dtype=dt, device=self.device)
            mask = None
            if cs.validity:
                nb = (n + 7) // 8
                bits = self._alias(cs.validity, nb, owner) if nb else torch.empty(0, dtype=torch.uint8, device=self.device)
                mask = _unpack_bits(bits, n)
            out.append(Col(name, t, cs.dtype, mask))
        return out

    def _alias(self, ptr: int, nbytes: int, owner):
        import torch

        if HipEngine._can_alias is not False:
            try:
                t = torch.as_tensor(_DevBuf(ptr, nbytes, owner), device=self.device)
                t._plhip_owner = owner
                HipEngine._can_alias = True
                return t
            except Exception:
                if HipEngine._can_alias:
                    raise
                HipEngine._can_alias = False
        t = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        _hip_memcpy_dtod(t.data_ptr(), ptr, nbytes)
        return t

    _can_alias = None

    def _columns(self, cols: list[Col]) -> list[Column]:
        out = []
        for c in cols:
            v = None if c.mask is None else _pack_bits(c.mask)
            out.append(Column(c.name, c.t, v, dtype=c.code, packed=True))
        return out

    def _sync_torch(self):
        # libplhip.so runs on its own stream: what torch queued on its stream (gathers, the collective) must
        # have landed before the library reads it
        import torch

        if torch.cuda.is_available():
            torch.cuda.current_stream(self.device).synchronize()

    def group_by(self, keys: list[Column], vals: list[Column], aggs, flags: int, names, hint: int = 0) -> list[Col]:
        from .frame import group_by_agg_raw

        self._sync_torch()
        handle, cols = group_by_agg_raw(keys, vals, aggs, flags | _ffi.FLAG_OUTPUT_DEVICE, hint)
        return self._frame(handle, cols, names)

    def partial(self, keys, vals, aggs, hint: int = 0) -> list[Col]:
        codes = [(ci, _ffi.AGG_NAMES[k]) for ci, k in aggs]
        return self.group_by(keys, vals, codes, _ffi.FLAG_PARTIAL, partial_names(keys, aggs), hint)

    def merge(self, keys: list[Col], parts: list[Col], ops: list[str]) -> list[Col]:
        kc, pc = self._columns(keys), self._columns(parts)
        aggs = [(i, _ffi.AGG_NAMES[op]) for i, op in enumerate(ops)]
        names = [k.name for k in keys] + [p.name for p in parts]
        return self.group_by(kc, pc, aggs, 0, names, hint=max(int(keys[0].t.numel()), 1))  # every row may be a group

    def key_hashes(self, keys: list[Col]):
        import torch

        n = keys[0].t.numel()
        out = torch.empty(n, dtype=torch.int64, device=self.device)
        if n:
            kc = self._columns(keys)
            self._sync_torch()
            arr = (_ffi.ColumnStruct * len(kc))(*[k.struct() for k in kc])
            _ffi.check(_ffi.lib().plhip_vec_hash(arr, len(kc), NULL_HASH, out.data_ptr(), _ffi.DEVICE))
        return out


    # ---- the exchange with the bucketing and the unpacking done by the library (exchange.hip) ------------
    def pack(self, keys, vals, aggs, world: int, hint: int = 0, slot_rows: int = 0):
        """Partial pass, then plhip_result_pack_by_owner: returns (records [G x rec] uint8 on the device, bucket
        after bucket by owner rank; rows per owner as a list).  With `slot_rows`: plhip_result_pack_slots, the
        buffer then holds `world` fixed-capacity slots (header + slot_rows records each)."""
        import ctypes as C

        import torch

        from .frame import group_by_agg_raw

        L = _ffi.lib()
        schema = partial_schema(keys, vals, aggs)
        rec = sum(_ffi.code_np_dtype(code).itemsize + (1 if nullable else 0) for code, nullable in schema)
        codes = [(ci, _ffi.AGG_NAMES[k]) for ci, k in aggs]
        self._sync_torch()
        handle, cols = group_by_agg_raw(keys, vals, codes, _ffi.FLAG_PARTIAL | _ffi.FLAG_OUTPUT_DEVICE, hint)
        try:
            g = int(L.plhip_result_num_rows(handle))
            if g:  # the frame's layout is a property of the schema: every rank computes the same one
                assert [(cs.dtype, bool(cs.validity)) for cs in cols] == schema, "partial frame does not match the schema"
                assert L.plhip_result_record_bytes(handle) == rec
            counts = (C.c_int64 * world)()
            if slot_rows:
                send = torch.empty(world * int(L.plhip_slot_bytes(rec, slot_rows)), dtype=torch.uint8, device=self.device)
                _ffi.check(L.plhip_result_pack_slots(handle, len(keys), NULL_HASH, world, slot_rows, send.data_ptr(), counts))
                return send, list(counts)
            send = torch.empty((max(g, 1), rec), dtype=torch.uint8, device=self.device)
            _ffi.check(L.plhip_result_pack_by_owner(handle, len(keys), NULL_HASH, world, send.data_ptr(), counts))
        finally:
            L.plhip_result_free(handle)
        return send[:g], list(counts)

    def merge_records(self, recv, keys, vals, aggs, hint: int = 0, raw: bool = False):
        """plhip_merge_records on the received records: unpacked into columns and regrouped (sum of sums /
        counts, min of mins, max of maxs) in one library call; returns the regrouped partial columns (or, with
        `raw`, the device-resident frame's handle: free it with plhip_result_free)."""
        import ctypes as C

        L = _ffi.lib()
        nk = len(keys)
        schema = partial_schema(keys, vals, aggs)
        m = int(recv.shape[0])
        self._sync_torch()
        dts = (C.c_int32 * len(schema))(*[code for code, _ in schema])
        nul = (C.c_uint8 * len(schema))(*[1 if nullable else 0 for _, nullable in schema])
        ops = _merge_ops(aggs)
        kinds = (C.c_int32 * max(len(ops), 1))(*[_ffi.AGG_NAMES[op] for op in ops])
        opts = _ffi.OptionsStruct(C.sizeof(_ffi.OptionsStruct), max(min(m, hint) if hint > 0 else m, 1))
        merged = C.c_void_p()
        _ffi.check(L.plhip_merge_records(recv.data_ptr() if m else None, m, dts, nul, len(schema), nk, kinds,
                                         _ffi.FLAG_OUTPUT_DEVICE, C.byref(opts), C.byref(merged)))
        if raw:
            return merged
        mcols = []
        for i in range(L.plhip_result_num_columns(merged)):
            cs = _ffi.ColumnStruct()
            _ffi.check(L.plhip_result_column(merged, i, C.byref(cs)))
            mcols.append(cs)
        names = [k.name for k in keys] + [f"p{i}" for i in range(len(schema) - nk)]
        return self._frame(merged, mcols, names)

    def finalize_frame(self, merged, keys, vals, aggs, names) -> list[Col]:
        """plhip_finalize_partial: the regrouped partial frame `merged` (handle; freed here) becomes the final
        columns, written by one kernel straight into torch-allocated tensors."""
        import ctypes as C

        import torch

        L = _ffi.lib()
        try:
            g = int(L.plhip_result_num_rows(merged))
            spec = [(k.name, k.dtype, k.validity is not None) for k in keys]
            finals = []
            for (ci, kind), name in zip(aggs, names):
                src = vals[ci].dtype if kind != "count" else _ffi.U32
                nullable = kind in ("mean", "min", "max") and vals[ci].validity is not None
                code = {"sum": sum_out_code(src), "count": _ffi.U32, "mean": mean_out_code(src)}.get(kind, src)
                spec.append((name, code, nullable))
                finals.append(_ffi.FinalStruct(_ffi.AGG_NAMES[kind], src))
            out, structs, bitmaps = [], [], []
            for name, code, nullable in spec:
                t = torch.empty(max(g, 1), dtype=_torch_dtype(code), device=self.device)[:g]
                bits = torch.empty((g // 64 + 1) * 8, dtype=torch.uint8, device=self.device) if nullable else None
                structs.append(_ffi.ColumnStruct(t.data_ptr() if g else None, bits.data_ptr() if nullable else None, 0, g, code, _ffi.DEVICE))
                out.append(Col(name, t, code))
                bitmaps.append(bits)
            carr = (_ffi.ColumnStruct * len(structs))(*structs)
            farr = (_ffi.FinalStruct * max(len(finals), 1))(*finals)
            _ffi.check(L.plhip_finalize_partial(merged, len(keys), farr, len(finals), carr))
        finally:
            L.plhip_result_free(merged)
        for c, bits in zip(out, bitmaps):
            if bits is not None:
                c.mask = _unpack_bits(bits, g)
        return out

    def communicator(self, group):
        """The library-side communicator of `group` (cached per group): RCCL when the group runs on it -- rank 0 draws the
        unique id, torch.distributed only carries its 128 bytes to the others -- and the host transport when the group
        is gloo (rehearsals of several ranks on one GPU): the library then calls back into `_HostTransport`, which
        stages the buffers through the host and runs dist.all_to_all_single."""
        import ctypes as C

        import torch
        import torch.distributed as dist

        # keyed by the process-group OBJECT (held in the entry, so its id cannot be recycled while the entry lives); the default
        # group is dist.group.WORLD, a new object after destroy_process_group() + init_process_group(): entries whose group is
        # no longer the one asked for are released, never reused
        pg = group if group is not None else dist.group.WORLD
        key = id(pg)
        hit = self._comms.get(key)
        if hit is not None and hit[2] is pg:
            self.transport = hit[3]
            return hit[0]
        L = _ffi.lib()
        if hit is not None:
            self._release(key)
        for k in [k for k, e in self._comms.items() if not _group_alive(e[2])]:
            self._release(k)
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        _ffi.check(L.plhip_init(self.device.index or 0))
        comm = C.c_void_p()
        if dist.get_backend(group) == "gloo":
            tr = _HostTransport(group, world)
            _ffi.check(L.plhip_comm_init_host(world, rank, tr.a2a, tr.a2av, None, C.byref(comm)))
            self.transport = "host (gloo rehearsal)"
            self._comms[key] = (comm, tr, pg, self.transport)
        else:
            import os

            # the library's own RCCL communicator; if it cannot be made on EVERY rank (no librccl the loader can find, an id that
            # could not be drawn) all ranks agree on the fallback: the group's own all_to_all_single through the host transport
            ok = os.environ.get("PLHIP_EXCHANGE_TRANSPORT", "rccl") != "torch"
            idbuf = (C.c_uint8 * _ffi.COMM_ID_BYTES)()
            if ok and rank == 0:
                ok = L.plhip_comm_unique_id(idbuf) == 0
            t = torch.tensor(list(idbuf) + [1 if ok else 0], dtype=torch.uint8, device=self.device)
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            vals = t.cpu().tolist()
            ok = bool(vals[-1])
            if ok:
                idbuf = (C.c_uint8 * _ffi.COMM_ID_BYTES)(*vals[:-1])
                ok = L.plhip_comm_init_rccl(idbuf, world, rank, C.byref(comm)) == 0
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            if int(flag.item()) == 1:
                self.transport = "rccl (library-side ncclSend / ncclRecv)"
                self._comms[key] = (comm, None, pg, self.transport)
            else:
                if ok:
                    L.plhip_comm_free(comm)
                comm = C.c_void_p()
                tr = _HostTransport(group, world, device=self.device)
                _ffi.check(L.plhip_comm_init_host(world, rank, tr.a2a, tr.a2av, None, C.byref(comm)))
                self.transport = "torch.distributed all_to_all_single (fallback: " + L.plhip_last_error().decode()[:120] + ")"
                self._comms[key] = (comm, tr, pg, self.transport)
        return comm

    transport = "host (gloo rehearsal)"

    _comms: dict = {}  # id(process group) -> (plhip_comm_t*, host transport or None, the group object, transport name)

    @classmethod
    def _release(cls, key) -> None:
        e = cls._comms.pop(key, None)
        if e is not None and e[0]:
            _ffi.lib().plhip_comm_free(e[0])

    @classmethod
    def close(cls) -> None:
        """Releases every cached communicator (plhip_comm_free): call before dist.destroy_process_group()."""
        for k in list(cls._comms):
            cls._release(k)

    def comm_info(self, group=None):
        """(world, rank, transport is RCCL) as the library's communicator of `group` reports them (plhip_comm_info: RCCL's own
        ncclCommCount / ncclCommUserRank)."""
        import ctypes as C

        w, r, t = C.c_int32(), C.c_int32(), C.c_int32()
        _ffi.check(_ffi.lib().plhip_comm_info(self.communicator(group), C.byref(w), C.byref(r), C.byref(t)))
        return w.value, r.value, bool(t.value)

    def sharded(self, keys, vals, aggs, world: int, group, hint: int, names) -> list[Col]:
        """ONE library call: plhip_group_by_agg_sharded (local pass -> pack -> exchange -> unpack + regroup -> finalize,
        the exchange being RCCL sends and receives on the library's stream).  This wrapper only builds the argument
        arrays, wraps the device-resident result and copies the call's statistics for bench.py."""
        handle, cols = self.sharded_raw(keys, vals, aggs, group, hint)
        return self._frame(handle, cols, [k.name for k in keys] + list(names))

    def sharded_raw(self, keys, vals, aggs, group, hint: int):
        """plhip_group_by_agg_sharded -> (handle, [ColumnStruct ...]) of the rank's device-resident share; the caller frees the
        handle with plhip_result_free.  The argument arrays of a query are built once and reused while the same column
        objects are passed (a host layer that re-runs a query)."""
        import ctypes as C

        L = _ffi.lib()
        comm = self.communicator(group)
        self._sync_torch()
        key = (tuple(id(c) for c in keys), tuple(id(c) for c in vals), tuple(aggs), int(hint), self.use_slots, self.dense_keys)
        if self._call_cache is None or self._call_cache[0] != key:
            codes = [(ci, _ffi.AGG_NAMES[k]) for ci, k in aggs]
            karr = (_ffi.ColumnStruct * max(1, len(keys)))(*[k.struct() for k in keys])
            varr = (_ffi.ColumnStruct * max(1, len(vals)))(*[v.struct() for v in vals])
            aarr = (_ffi.AggStruct * max(1, len(codes)))(*[_ffi.AggStruct(c, k) for c, k in codes])
            opts = _ffi.OptionsStruct(C.sizeof(_ffi.OptionsStruct), int(hint))
            flags = _ffi.FLAG_OUTPUT_DEVICE | (0 if self.use_slots else _ffi.FLAG_EXACT_EXCHANGE)
            if self.dense_keys and len(keys) == 1 and keys[0].n_categories is not None and hint:
                flags |= _ffi.FLAG_KEYS_DENSE
            handle, st = C.c_void_p(), _ffi.ShardedStats()
            args = (karr, len(keys), varr, len(vals), aarr, len(codes), flags, C.byref(opts), C.byref(handle))
            self._call_cache = (key, args, handle, st, C.byref(st), (keys, vals, opts))  # (the columns stay alive with the arrays)
        _, args, handle, st, st_ref, _ = self._call_cache
        rc = L.plhip_group_by_agg_sharded(comm, *args)
        if rc:
            _ffi.check(rc)
        L.plhip_sharded_last_stats(st_ref)
        self.last_partial_kernel_ms = (st.partial_kernel_ms, st.partial_dominant_ms, st.partial_dominant_kernel.decode())
        self.last_partial_breakdown = tuple(st.partial_breakdown_ms)
        self.last_partial_strategy = st.partial_strategy.decode()
        self.last_partial_groups = int(st.partial_groups)
        self.last_phase_ms = {"partial_pass": st.partial_ms, "pack": st.pack_ms, "exchange": st.exchange_ms, "merge": st.merge_ms,
                              "finalize": st.finalize_ms}
        self.last_shuffle_bytes = int(st.shuffle_bytes)
        self.last_host_syncs = int(st.host_syncs)
        self.last_protocol = {1: "slots: one equal-split exchange (ncclSend/ncclRecv per peer)",
                              2: "exact: counts exchange + variable-size exchange",
                              3: "dense: one equal-split exchange of the owners' code ranges (raw partial arrays, ncclSend/ncclRecv per peer)"}.get(st.protocol, "none")
        out = C.c_void_p(handle.value)
        cols = []
        for i in range(L.plhip_result_num_columns(out)):
            cs = _ffi.ColumnStruct()
            _ffi.check(L.plhip_result_column(out, i, C.byref(cs)))
            cols.append(cs)
        return out, cols

    _call_cache = None
    dense_keys = False  # bench.py: declare a single Categorical key's codes dense in the local pass
    last_host_syncs = 0
    last_partial_kernel_ms = (0.0, 0.0, "")
    last_partial_breakdown = (0.0, 0.0, 0.0)
    last_partial_strategy = ""
    last_partial_groups = 0
    last_phase_ms: dict = {}
    last_shuffle_bytes = 0
    last_protocol = ""

    use_slots = True  # tests switch the slotted exchange off to cover the exact protocol

    def _merge_call(self, recv, schema, nk, aggs, hint, m, slots=None):
        """plhip_merge_records / plhip_merge_slots -> handle of the regrouped partial frame."""
        import ctypes as C

        L = _ffi.lib()
        self._sync_torch()
        dts = (C.c_int32 * len(schema))(*[code for code, _ in schema])
        nul = (C.c_uint8 * len(schema))(*[1 if nullable else 0 for _, nullable in schema])
        ops = _merge_ops(aggs)
        kinds = (C.c_int32 * max(len(ops), 1))(*[_ffi.AGG_NAMES[op] for op in ops])
        opts = _ffi.OptionsStruct(C.sizeof(_ffi.OptionsStruct), max(min(m, hint) if hint > 0 else m, 1))
        merged = C.c_void_p()
        if slots is None:
            _ffi.check(L.plhip_merge_records(recv.data_ptr() if m else None, m, dts, nul, len(schema), nk, kinds,
                                             _ffi.FLAG_OUTPUT_DEVICE, C.byref(opts), C.byref(merged)))
        else:
            world, slot_rows, rc = slots
            _ffi.check(L.plhip_merge_slots(recv.data_ptr(), world, slot_rows, rc, dts, nul, len(schema), nk, kinds,
                                           _ffi.FLAG_OUTPUT_DEVICE, C.byref(opts), C.byref(merged)))
        return merged


def _group_alive(pg) -> bool:
    """Is `pg` still a usable process group (the default one: still dist.group.WORLD; a sub-group: still registered)?"""
    import torch.distributed as dist

    if not dist.is_initialized():
        return False
    if pg is dist.group.WORLD:
        return True
    try:
        dist.get_world_size(pg)
        return True
    except Exception:
        return False


def partial_schema(keys, vals, aggs) -> list[tuple[int, bool]]:
    """(dtype code, has a validity bitmap) of every column of a PLHIP_FLAG_PARTIAL frame (api.hip build_aggs):
    determined by the query's schema alone, hence identical on every rank whatever its shard holds."""
    signed = (_ffi.I8, _ffi.I16, _ffi.I32, _ffi.I64)
    out = [(k.dtype, k.validity is not None) for k in keys]
    for ci, kind in aggs:
        if kind == "count":
            out.append((_ffi.U64, False))
            continue
        src = vals[ci].dtype
        w = _ffi.code_np_dtype(src).itemsize
        is_f = src in (_ffi.F32, _ffi.F64)
        int_sum = _ffi.I64 if (src in signed or w < 4) else _ffi.U64
        if kind == "sum":
            out.append((_ffi.F64 if is_f else int_sum, False))
        elif kind == "mean":
            out.append((_ffi.F64 if (is_f or w > 4) else (_ffi.I64 if src in signed else _ffi.U64), False))
            out.append((_ffi.U64, False))
        else:
            out.append((src, vals[ci].validity is not None))
    return out


def _hip_memcpy_dtod(dst: int, src: int, nbytes: int) -> None:
    _ffi.check(_ffi.lib().plhip_memcpy(dst, _ffi.DEVICE, src, _ffi.DEVICE, nbytes))


def _unpack_bits(bits, n: int):
    import torch

    sh = torch.arange(8, device=bits.device, dtype=torch.uint8)
    return ((bits.unsqueeze(1) >> sh) & 1).reshape(-1)[:n].to(torch.bool)


def _pack_bits(mask):
    import torch

    n = mask.numel()
    pad = (-n) % 8
    m = mask.to(torch.uint8)
    if pad:
        m = torch.cat([m, torch.zeros(pad, dtype=torch.uint8, device=m.device)])
    w = (1 << torch.arange(8, device=m.device, dtype=torch.int32)).to(torch.uint8)
    return (m.reshape(-1, 8) * w).sum(dim=1, dtype=torch.int32).to(torch.uint8).contiguous()


def partial_names(keys, aggs) -> list[str]:
    names = [k.name for k in keys]
    for i, (ci, kind) in enumerate(aggs):
        if kind == "mean":
            names += [f"a{i}_sum", f"a{i}_cnt"]
        else:
            names.append(f"a{i}_{kind}")
    return names


# ---------------------------------------------------------------------------------------------------
# the exchange
# ---------------------------------------------------------------------------------------------------
def _as_bytes(c: Col):
    import torch

    n = c.t.numel()
    if n == 0:
        return torch.empty((0, c.t.element_size()), dtype=torch.uint8, device=c.t.device)
    return c.t.contiguous().view(torch.uint8).reshape(n, c.t.element_size())


def exchange_by_owner(cols: list[Col], owner, world: int, group=None) -> list[Col]:
    """Sends every row to rank owner[row]: the rows are sorted by owner, packed into one byte matrix
    (all columns and masks side by side) and moved with ONE all_to_all_single after the counts."""
    import torch
    import torch.distributed as dist

    dev = owner.device
    n = owner.numel()
    order = torch.argsort(owner, stable=True)
    # The record layout must be the same on every rank: an engine attaches a mask to a partial column exactly
    # when the SOURCE column has a validity bitmap (a property of the schema, not of the shard's contents).
    send_counts = torch.bincount(owner, minlength=world)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    both = torch.stack([send_counts, recv_counts]).tolist()  # the one host sync of the exchange
    sc, rc = both[0], both[1]
    pieces, layout = [], []
    for c in cols:
        pieces.append(_as_bytes(c))
        layout.append(c.t.element_size())
        if c.mask is not None:
            pieces.append(c.mask.to(torch.uint8).reshape(n, 1))
            layout.append(-1)
    rec = sum(abs(w) for w in layout)
    # a rank may send or receive nothing: the buffers still get storage (a collective must not see a NULL pointer)
    send = torch.cat(pieces, dim=1).index_select(0, order).contiguous() if n else torch.empty((1, rec), dtype=torch.uint8, device=dev)[:0]
    recv = torch.empty((max(sum(rc), 1), rec), dtype=torch.uint8, device=dev)[:sum(rc)]
    dist.all_to_all_single(recv, send, output_split_sizes=rc, input_split_sizes=sc, group=group)
    out, off, it = [], 0, iter(layout)
    m = recv.shape[0]
    for c in cols:
        w = next(it)
        t = recv[:, off:off + w].contiguous().view(c.t.dtype).reshape(m) if m else torch.empty(0, dtype=c.t.dtype, device=dev)
        off += w
        mask = None
        if c.mask is not None:
            next(it)
            mask = recv[:, off].contiguous().to(torch.bool)
            off += 1
        out.append(Col(c.name, t, c.code, mask))
    return out


def _merge_ops(aggs) -> list[str]:
    ops = []
    for _, kind in aggs:
        ops += {"sum": ["sum"], "count": ["sum"], "min": ["min"], "max": ["max"], "mean": ["sum", "sum"]}[kind]
    return ops


def _to_f64(c: Col):
    import torch

    if c.code == _ffi.U64:  # torch has no unsigned 64-bit arithmetic
        return (c.t >> 1).to(torch.float64) * 2.0 + (c.t & 1).to(torch.float64)
    return c.t.to(torch.float64)


def finalize(keys: list[Col], merged: list[Col], aggs, val_codes: list[int], names: list[str],
             val_nullable: list[bool] | None = None) -> list[Col]:
    """AggregationExpr::finalize (aggregation.rs:367-483) on the regrouped partials."""
    import torch

    out = list(keys)
    it = iter(merged)
    for (ci, kind), name in zip(aggs, names):
        src = val_codes[ci] if kind != "count" else _ffi.U32
        if kind == "sum":
            c = next(it)
            code = sum_out_code(src)
            out.append(Col(name, c.t.to(_torch_dtype(code)), code))  # 64-bit partial wraps into the column's type
        elif kind == "count":
            c = next(it)
            out.append(Col(name, c.t.to(torch.int32), _ffi.U32))
        elif kind == "mean":
            s, cnt = next(it), next(it)
            code = mean_out_code(src)
            n = cnt.t.to(torch.float64)  # a row count: far below 2^63, the int64 view of the u64 column is exact
            if src == _ffi.F32:  # the reference folds f32 in f32 before the division (aggregations/mod.rs:701-742)
                num = s.t.to(torch.float32).to(torch.float64)
            else:
                num = _to_f64(s)
            # a mean is null only for an all-null group: without a validity bitmap on the source there is none
            # (and no host sync to find out): every group then has at least one counted row
            nullable = True if val_nullable is None else val_nullable[ci]
            if nullable:
                valid = n > 0
                v = torch.where(valid, num / torch.where(valid, n, torch.ones_like(n)), torch.zeros_like(n))
            else:
                valid, v = None, num / n
            out.append(Col(name, v.to(_torch_dtype(code)), code, valid))
        else:
            c = next(it)
            out.append(Col(name, c.t, c.code, c.mask))
    return out


def group_by_agg_sharded(keys: list[Column], vals: list[Column], aggs, *, device=None, engine=None, group=None,
                         names: list[str] | None = None, n_groups_hint: int = 0, force_exchange: bool = False,
                         python_exchange: bool = False):
    """`keys` / `vals` are this rank's row shard; `aggs` = [(value index, 'sum'|'mean'|'min'|'max'|'count')].
    Returns this rank's share of the result frame (list[Col]: keys then one column per aggregation); the
    shares of all ranks are disjoint in the keys and together are the result."""
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    engine = engine or HipEngine(device)
    names = names or [f"{vals[ci].name if kind != 'count' else 'count'}_{kind}" for ci, kind in aggs]
    val_codes = [v.dtype for v in vals]
    exchange = world > 1 or (force_exchange and dist.is_initialized())  # force_exchange: rehearse the collective on one rank
    if exchange and isinstance(engine, HipEngine) and not python_exchange:
        return engine.sharded(keys, vals, aggs, world, group, n_groups_hint, names)
    part = engine.partial(keys, vals, aggs, n_groups_hint) if isinstance(engine, HipEngine) else engine.partial(keys, vals, aggs)
    nk = len(keys)
    pk, pv = part[:nk], part[nk:]
    if exchange:
        h = engine.key_hashes(pk)
        owner = ((h >> 33) & 0x7FFFFFFF) % world
        moved = exchange_by_owner(pk + pv, owner, world, group)
        pk, pv = moved[:nk], moved[nk:]
        merged = engine.merge(pk, pv, _merge_ops(aggs))
        pk, pv = merged[:nk], merged[nk:]
    return finalize(pk, pv, aggs, val_codes, names, [v.validity is not None for v in vals])
