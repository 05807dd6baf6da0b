import sys, time, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','.'))
import gswm_amd
from gswm_amd import unet as U
dev='cuda'; dt=torch.float16
m = U.synthetic_init_(U.UNet2DCondition(), 0).to(dev, dt).eval()
B=int(sys.argv[1]) if len(sys.argv)>1 else 128
x=torch.randn(B,4,64,64,device=dev,dtype=dt); t=torch.full((),500,device=dev); c=torch.randn(B,77,1024,device=dev,dtype=dt)
with torch.no_grad():
    for _ in range(3): y=m(x,t,c)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(4): y=m(x,t,c)
    torch.cuda.synchronize(); d=(time.perf_counter()-t0)/4
print(f"B={B}: {d*1e3:.1f} ms {B*0.804/d:.0f} TFLOP/s", flush=True)
