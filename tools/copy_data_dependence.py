"""Does a plain device copy depend on the data?  (The blur passes run 20 % faster on zeros.)"""
import time
import torch

def timed(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

for mb in (537, 4096):
    n = mb * (1 << 20) // 2
    noise = torch.randint(-32768, 32768, (n,), device="cuda", dtype=torch.int16)
    zeros = torch.zeros_like(noise)
    out = torch.empty_like(noise)
    for name, src in (("noise", noise), ("zeros", zeros), ("noise", noise), ("zeros", zeros)):
        t = timed(lambda: out.copy_(src))
        print("%5d MB copy of %s: %.3f ms  %.2f TB/s (read+write)" % (mb, name, t * 1e3, 2 * n * 2 / t / 1e12))
