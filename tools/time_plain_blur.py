import torch, time, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imagemagick_amd as im
im.set_precision(im.PRECISION_FAST)
def run(ch, alpha, n=8192):
    a=torch.randint(-32768,32768,(n,n,ch),device='cuda',dtype=torch.int16).view(torch.uint16)
    img=im.Image(a,has_alpha=alpha)
    for _ in range(3): im.blur_image(img,0.0,10.0)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): im.blur_image(img,0.0,10.0)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
    print("ch",ch,"alpha",alpha,"NO_MFMA",im.get_option("MAGICKHIP_NO_MFMA"),"%.3f ms %.1f Gpix/s"%(dt*1e3,n*n/dt/1e9),flush=True)
for env in (None,"1"):
    im.set_option("MAGICKHIP_NO_MFMA", env)
    run(3,False); run(4,False); run(4,True)
