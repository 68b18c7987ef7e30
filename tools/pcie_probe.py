"""What the host link of the box can do: page-locked H2D, D2H, both at once; the library's
staged transfers of a pageable block (MhUpload / MhDownload) for comparison; host memcpy rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

n = 512 << 20
pin_a = torch.empty(n, dtype=torch.uint8).pin_memory()
pin_b = torch.empty(n, dtype=torch.uint8).pin_memory()
dev_a = torch.empty(n, dtype=torch.uint8, device="cuda")
dev_b = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def h2d():
    with torch.cuda.stream(s1):
        dev_a.copy_(pin_a, non_blocking=True)


def d2h():
    with torch.cuda.stream(s2):
        pin_b.copy_(dev_b, non_blocking=True)


def both():
    h2d(); d2h()


for name, fn, moved in (("H2D pinned", h2d, n), ("D2H pinned", d2h, n), ("H2D + D2H at once", both, 2 * n)):
    dt = timeit(fn)
    print("%-22s %6.1f ms  %5.1f GB/s" % (name, dt * 1e3, moved / dt / 1e9), flush=True)

page = np.ones(n, dtype=np.uint8)
page2 = np.empty(n, dtype=np.uint8)
t0 = time.perf_counter(); page2[:] = page; dt = time.perf_counter() - t0
print("host memcpy, 1 thread (first touch of dst) %.1f ms  %.1f GB/s" % (dt * 1e3, n / dt / 1e9))
t0 = time.perf_counter(); page2[:] = page; dt = time.perf_counter() - t0
print("host memcpy, 1 thread (warm)               %.1f ms  %.1f GB/s" % (dt * 1e3, n / dt / 1e9))

import imagemagick_amd as im
from imagemagick_amd import _lib
lib = im.load()
import ctypes
for threads in ("4", "8", "16"):
    im.set_option("MAGICKHIP_TRANSFER_THREADS", threads)
    def up():
        _lib.check(lib.MhUpload(0, dev_a.data_ptr(), page.ctypes.data, n, None))
    def down():
        _lib.check(lib.MhDownload(0, page2.ctypes.data, dev_a.data_ptr(), n, None))
    for name, fn in (("MhUpload pageable", up), ("MhDownload pageable", down)):
        dt = timeit(fn, 3)
        print("%-22s threads %-2s %6.1f ms  %5.1f GB/s" % (name, threads, dt * 1e3, n / dt / 1e9), flush=True)
