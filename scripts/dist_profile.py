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
dist.destroy_process_group()
