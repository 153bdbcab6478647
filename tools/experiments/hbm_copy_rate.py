"""Round-4 experiment (GPU): what a plain device copy / read-modify-write achieves on this box, as the yardstick for the TCN
kernels (tcn_dw_k: 100 MB in 19.7 us = 5.1 TB/s; tcn_pw_k: 150 MB in 31.9 us = 4.7 TB/s)."""
import time, torch
for mb in (50, 100, 400, 2000):
    n = mb * 1024 * 1024 // 4
    x = torch.randn(n, device="cuda"); y = torch.empty_like(x)
    for name, fn, traffic in (("copy", lambda: y.copy_(x), 2), ("add_", lambda: y.add_(x), 3), ("mul_ (rw)", lambda: x.mul_(1.0001), 2)):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        k = 50
        for _ in range(k): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / k
        print(f"{mb:5d} MB tensors  {name:10s} {dt * 1e6:8.1f} us  {traffic * mb * 1.048576e6 / dt / 1e12:6.2f} TB/s")
