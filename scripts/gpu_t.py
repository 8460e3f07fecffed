import torch, time
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
k = torch.randint(0, 100_000_000, (100_000_000,), device=dev, dtype=torch.int64, generator=g)
v = torch.rand(100_000_000, device=dev, dtype=torch.float64, generator=g)
for name, f in (("max i64", lambda: k.max()), ("sum i64", lambda: k.sum()), ("sum f64", lambda: v.sum()), ("copy i64", lambda: k.clone())):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name}: {ms*1e3:.0f} us, {0.8/ms:.2f} TB/s read", flush=True)
