import sys,os,time,torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imagemagick_amd as im, bench
im.set_precision(im.PRECISION_FAST)
n=8192
a=torch.randint(-32768,32768,(n,n,3),device='cuda',dtype=torch.int16).view(torch.uint16)
img=im.Image(a); hold={}
def f(): hold['o']=im.gaussian_blur_image(img,0.0,10.0)
f();f();torch.cuda.synchronize();t=time.perf_counter()
for _ in range(3): f()
torch.cuda.synchronize();dt=(time.perf_counter()-t)/3
print("RGB gaussian 0x10: %.2f ms"%(dt*1e3),{k:round(v['avg_ms'],3) for k,v in bench.kernel_profile(im,f,2).items()})
