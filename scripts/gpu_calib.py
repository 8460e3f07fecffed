import torch, time
dev = torch.device("cuda:0")
n = 10_000_000
a = torch.randint(1, 100, (n,), device=dev, dtype=torch.int32)
b = torch.randint(1, 5, (n,), device=dev, dtype=torch.int32)
c = torch.empty_like(a)
d = torch.rand(n, device=dev, dtype=torch.float64)
def timeit(f, name, bytes_):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(20):
        s.record(); f(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    ts.sort(); t = ts[len(ts)//2]
    print(f"{name}: {t*1e3:.1f} us  {bytes_/t/1e6:.0f} GB/s")
timeit(lambda: torch.add(a, b, out=c), "add i32 (120MB)", 120e6)
timeit(lambda: a.sum(), "sum i32 (40MB)", 40e6)
timeit(lambda: d.sum(), "sum f64 (80MB)", 80e6)
big = torch.empty(256_000_000, device=dev, dtype=torch.int32)
big2 = torch.empty_like(big)
timeit(lambda: big2.copy_(big), "copy 1GB+1GB", 2.048e9)
timeit(lambda: big.sum(), "sum 1GB", 1.024e9)
