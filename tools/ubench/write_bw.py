"""Store / load rates of plain torch kernels over 3.4 GB (the size of the headline's packed cost volume): fill, zero, copy.
Usage: python tools/ubench/write_bw.py"""
import torch, time
x = torch.empty(3_400_000_000 // 4, dtype=torch.int32, device="cuda")
y = torch.empty_like(x)
for name, fn in (("fill", lambda: x.fill_(7)), ("zero", lambda: x.zero_()), ("copy", lambda: y.copy_(x))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    b = x.numel() * 4 * (2 if name == "copy" else 1)
    print(f"{name}: {ms:.3f} ms  {b / ms / 1e9:.2f} TB/s")
