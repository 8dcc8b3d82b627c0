#!/usr/bin/env python3
"""HBM ceilings with plain torch ops on one MI355X: pure write (fill_), copy (1 read : 1 write), pure read (sum)."""
import time
import torch

n = 1 << 29                      # 2 GiB of float32
a = torch.empty(n, dtype=torch.float32, device="cuda")
b = torch.empty(n, dtype=torch.float32, device="cuda")


def bench(f, bytes_moved, reps=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    return bytes_moved / dt / 1e12


print("fill_  (write only)   %.2f TB/s" % bench(lambda: a.fill_(1.0), 4 * n))
print("copy_  (1 R : 1 W)    %.2f TB/s" % bench(lambda: b.copy_(a), 8 * n))
print("sum    (read only)    %.2f TB/s" % bench(lambda: a.sum(), 4 * n))
c = torch.empty(n, dtype=torch.float32, device="cuda")
d = torch.empty(n, dtype=torch.float32, device="cuda")
print("sin,cos-like 1 R : 2 W (two copies from one source)  %.2f TB/s" % bench(lambda: (b.copy_(a), c.copy_(a)), 16 * n))
