#!/bin/bash
python - <<'P'
import numpy as np, torch, sys
sys.path.insert(0,'.')
import imagemagick_amd as im, bench
im.load(); im.set_precision(im.PRECISION_EXACT)
n=8192
rng=np.random.default_rng(77)
px=rng.integers(0,65536,(n,n,4),dtype=np.uint16); px[:,:,3]=rng.integers(0,4,(n,n))
dev=im.Image(torch.from_numpy(px.view(np.int16)).cuda().view(torch.uint16))
h={}
for i in range(3):
    prof=bench.kernel_profile(im,lambda: h.update(o=im.blur_image(dev,0.0,10.0)),3)
    print("tiny alpha:",{k:round(v["avg_ms"],3) for k,v in prof.items()})
P
