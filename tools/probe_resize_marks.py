import os, sys
sys.path.insert(0, os.getcwd())
import torch
import imagemagick_amd as im
im.load()
im.set_precision(im.PRECISION_FAST)
im.set_option("MAGICKHIP_RESIZE_STREAM_REPORT", "1")
m = 2048
gen = torch.Generator(device="cuda").manual_seed(1)
base = torch.randint(-32768, 32768, (m, m, 4), generator=gen, device="cuda", dtype=torch.int16).view(torch.uint16)
for label, lo in (("alpha 0..65535", 0), ("alpha 1..65535", 1), ("alpha 100..65535", 100)):
    px = base.clone()
    a = px.view(torch.int16)[:, :, 3].to(torch.int32) & 0xffff
    a = torch.clamp(a, min=lo)
    px.view(torch.int16)[:, :, 3] = (a - 65536 * (a >= 32768)).to(torch.int16)
    for f in (2, 3, 4, 5):
        print(label, "x%d" % f, flush=True)
        try:
            im.resize_image(im.Image(px), f * m, f * m, "Lanczos")
        except Exception as e:
            print("  ", e)
        torch.cuda.synchronize()
