import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imagemagick_amd as im
im.set_precision(im.PRECISION_FAST)
n=8192
a=torch.randint(-32768,32768,(n,n,4),device='cuda',dtype=torch.int16).view(torch.uint16)
img=im.Image(a)
out=[]
for sigma in (2.0,4.0,6.5,10.0,14.0):
    for _ in range(3): im.blur_image(img,0.0,sigma)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): im.blur_image(img,0.0,sigma)
    torch.cuda.synchronize(); out.append("%g: %.3f ms"%(sigma,(time.perf_counter()-t)/10*1e3))
print(os.path.basename(os.environ.get("MAGICKHIP_LIBRARY","default")),"  ".join(out))
