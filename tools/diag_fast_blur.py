import os
import torch, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imagemagick_amd as im
def run(n):
    g = torch.Generator(device="cuda").manual_seed(11)
    a = torch.randint(-32768, 32768, (n, n, 4), generator=g, device="cuda", dtype=torch.int16)
    a[: n // 8, :, 3] = torch.randint(0, 4, (n // 8, n), generator=g, device="cuda", dtype=torch.int16)
    a[n // 8: n // 4, :, 3] = 0
    img = im.Image(a.view(torch.uint16))
    im.set_precision(im.PRECISION_EXACT)
    exact = im.blur_image(img, 0.0, 10.0).pixels.view(torch.int16).to(torch.int32) & 0xffff
    im.set_precision(im.PRECISION_FAST)
    fast = im.blur_image(img, 0.0, 10.0).pixels.view(torch.int16).to(torch.int32) & 0xffff
    d=(fast-exact).abs()
    bad=(d>1).nonzero()
    print("n",n,"max",int(d.max()),"count",bad.shape[0])
    if bad.shape[0]:
        print("rows",int(bad[:,0].min()),int(bad[:,0].max()),"cols",int(bad[:,1].min()),int(bad[:,1].max()),"ch",torch.bincount(bad[:,2]).tolist())
        for k in range(min(8,bad.shape[0])):
            y,x,c=[int(t) for t in bad[k*max(1,bad.shape[0]//8)]]
            print(y,x,c,"fast",fast[y,x].tolist(),"exact",exact[y,x].tolist(),"src",(a[y,x].to(torch.int32)&0xffff).tolist())
run(1024); run(8192)
