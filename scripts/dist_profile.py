"""Where the time of one sharded step goes on ONE rank (RCCL world of 1; development aid)."""
import os, sys, time
sys.path.insert(0, ".")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
import torch, torch.distributed as dist
import polars_amd as pa
from polars_amd import _ffi, h2oai, distributed as pd_
import bench
dist.init_process_group("nccl", device_id=torch.device("cuda:0"), rank=0, world_size=1)
dev = torch.device("cuda:0")
keys_n, vals_n, aggs, wide = bench.WORKLOADS["q3"]
n = 10_000_000
t = h2oai.generate(n, seed=108, device=dev, wide=wide, columns=keys_n + vals_n)
cols = h2oai.to_columns(t, wide)
keys, vals = [cols[k] for k in keys_n], [cols[v] for v in vals_n]
eng = pd_.HipEngine(dev)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for rep in range(4):
    t0 = T(); part = eng.partial(keys, vals, aggs, 100000)
    t1 = T(); h = eng.key_hashes(part[:1]); owner = ((h >> 33) & 0x7FFFFFFF) % 1
    t2 = T(); moved = pd_.exchange_by_owner(part, owner, 1, None)
    t3 = T(); merged = eng.merge(moved[:1], moved[1:], pd_._merge_ops(aggs))
    t4 = T(); out = pd_.finalize(merged[:1], merged[1:], aggs, [v.dtype for v in vals], ["a", "b"])
    t5 = T()
    print(f"partial {1e3*(t1-t0):.3f}  hash+owner {1e3*(t2-t1):.3f}  exchange {1e3*(t3-t2):.3f}  merge {1e3*(t4-t3):.3f} ({pa.last_strategy()})  finalize {1e3*(t5-t4):.3f} ms")
print("library-side exchange:")
for rep in range(6):
    t0 = T(); send, sc = eng.pack(keys, vals, aggs, 1, 100000)
    t1 = T()
    send_counts = torch.tensor(sc, dtype=torch.int64, device=dev); recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts); rc = recv_counts.tolist()
    t2 = T()
    recv = torch.empty((max(sum(rc), 1), send.shape[1]), dtype=torch.uint8, device=dev)[:sum(rc)]
    dist.all_to_all_single(recv, send, output_split_sizes=rc, input_split_sizes=sc)
    t3 = T(); merged = eng.merge_records(recv, keys, vals, aggs, 100000)
    t4 = T(); out = pd_.finalize(merged[:1], merged[1:], aggs, [v.dtype for v in vals], ["a", "b"], [False, False])
    t5 = T()
    print(f"pack {1e3*(t1-t0):.3f}  counts {1e3*(t2-t1):.3f}  data a2a {1e3*(t3-t2):.3f}  merge_records {1e3*(t4-t3):.3f} ({pa.last_strategy()})  finalize {1e3*(t5-t4):.3f} ms")
# inside pack / merge_records
import ctypes as C
from polars_amd.frame import group_by_agg_raw
L = _ffi.lib()
codes = [(ci, _ffi.AGG_NAMES[k]) for ci, k in aggs]
for rep in range(4):
    t0 = T(); handle, cs = group_by_agg_raw(keys, vals, codes, _ffi.FLAG_PARTIAL | _ffi.FLAG_OUTPUT_DEVICE, 100000)
    t1 = T(); g = L.plhip_result_num_rows(handle); send = torch.empty((g, 28), dtype=torch.uint8, device=dev); counts = (C.c_int64 * 1)()
    t2 = T(); L.plhip_result_pack_by_owner(handle, 1, 1, 1, send.data_ptr(), counts)
    t3 = T(); L.plhip_result_free(handle)
    print(f"  partial {1e3*(t1-t0):.3f} ({pa.last_strategy()})  alloc {1e3*(t2-t1):.3f}  pack_by_owner {1e3*(t3-t2):.3f}")
dist.destroy_process_group()
