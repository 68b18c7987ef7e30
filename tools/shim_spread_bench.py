#!/usr/bin/env python3
"""One big host-resident image through MagickCore's own MorphologyImage (Dilate Disk:15), BlurImage
and EqualizeImage on the HIP-backed MagickCore build (shim/magickcore.py), the row bands of the frame
going round every (logical) device (shim/accelerate_hip.c, MAGICK_HIP_SPREAD_BYTES) or one device per
call (the reference's arbitration), as the environment says.  A subprocess of bench.py (the device list
is made when the libraries start); prints one JSON line: wall ms per call, source not resident, result
read on the host.      python tools/shim_spread_bench.py <edge>"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "shim"))
import numpy as np          # noqa: E402
import magickcore as mc     # noqa: E402


def main():
    edge = int(sys.argv[1])
    rng = np.random.default_rng(9)
    px = rng.integers(0, 65536, (edge, edge, 4), dtype=np.uint16)
    out = {"edge": edge, "spread_bytes": os.environ.get("MAGICK_HIP_SPREAD_BYTES"),
           "logical_devices": os.environ.get("MAGICKHIP_LOGICAL_DEVICES")}
    for name, call in (("dilate_disk15", lambda image: image.morphology("Dilate", 1, "Disk:15")),
                       ("blur_0x10", lambda image: image.blur(0.0, 10.0))):
        call(mc.Image(px)).sync()                         # warm: code objects, pools, page-locked caches
        best = None
        for _ in range(2):
            source = mc.Image(px)                         # a source without a device copy
            t0 = time.perf_counter()
            call(source).sync()                           # ... and the result read on the host
            ms = (time.perf_counter() - t0) * 1e3
            best = ms if best is None else min(best, ms)
        out[name + "_ms"] = round(best, 2)
    mc.Image(px).equalize().sync()
    best = None
    for _ in range(2):
        source = mc.Image(px)
        t0 = time.perf_counter()
        source.equalize().sync()
        ms = (time.perf_counter() - t0) * 1e3
        best = ms if best is None else min(best, ms)
    out["equalize_ms"] = round(best, 2)
    out["devices"] = [{"calls": c, "streams": s} for c, s in mc.device_statistics()]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
