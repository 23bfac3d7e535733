import torch, time
n = 26 * 1024**3 // 4
a = torch.empty(n, dtype=torch.int32, device="cuda"); a.fill_(1)
b = torch.empty_like(a)
for name, fn in (("copy", lambda: b.copy_(a)), ("read(sum int64-free)", lambda: torch.max(a)), ("fill", lambda: b.fill_(3))):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    gb = n * 4 / 1e9 * (2 if name == "copy" else 1)
    print(name, round(ms, 2), "ms", round(gb / ms, 1), "GB/ms = TB/s")
