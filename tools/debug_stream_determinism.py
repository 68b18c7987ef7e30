#!/usr/bin/env python3
"""x3 Lanczos of a Q16 RGBA frame, twice: are the outputs and the marked blocks the same?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import imagemagick_amd as im
im.load(); im.set_precision(im.PRECISION_FAST)
gen = torch.Generator(device="cuda").manual_seed(1)
m = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
f = int(sys.argv[2]) if len(sys.argv) > 2 else 3
src = torch.randint(0, 65536, (m, m, 4), generator=gen, device="cuda", dtype=torch.int32).to(torch.uint16)
img = im.Image(src)
outs = []
for rep in range(3):
    out = im.resize_image(img, f * m, f * m, "Lanczos")
    torch.cuda.synchronize()
    outs.append(out.tensor().clone() if hasattr(out, "tensor") else torch.from_numpy(out.numpy()))
for rep in range(1, 3):
    same = bool((outs[rep].view(torch.int16) == outs[0].view(torch.int16)).all())
    print("run %d identical to run 0: %s" % (rep, same))
im.set_precision(im.PRECISION_EXACT)
exact = im.resize_image(img, f * m, f * m, "Lanczos")
e = torch.from_numpy(exact.numpy()).to(torch.int32)
d = (torch.from_numpy(outs[0].cpu().numpy() if hasattr(outs[0], "cpu") else outs[0]).to(torch.int32) - e).abs()
print("FAST against EXACT: max", int(d.max()), "differing", int((d != 0).sum()))
