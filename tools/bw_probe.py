#!/usr/bin/env python3
"""Achievable HBM bandwidth on this box: pure write (fill), pure read (sum), copy."""
import time
import torch

def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

gb = 16
x = torch.empty(gb * (1 << 30) // 4, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
t = timed(lambda: x.fill_(1.0)); print("fill  %5.2f TB/s" % (x.numel() * 4 / t / 1e12))
t = timed(lambda: x.zero_()); print("zero  %5.2f TB/s" % (x.numel() * 4 / t / 1e12))
t = timed(lambda: y.copy_(x)); print("copy  %5.2f TB/s (read+write)" % (2 * x.numel() * 4 / t / 1e12))
t = timed(lambda: x.sum()); print("sum   %5.2f TB/s (read)" % (x.numel() * 4 / t / 1e12))
