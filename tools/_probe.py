import sys, time, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','.'))
import gswm_amd
from gswm_amd import unet as U
dev='cuda'; dt=torch.float16
m = U.synthetic_init_(U.UNet2DCondition(), 0).to(dev, dt).eval()
B=128
x=torch.randn(B,4,64,64,device=dev,dtype=dt); t=torch.full((),500,device=dev); c=torch.randn(B,77,1024,device=dev,dtype=dt)
with torch.no_grad():
    for _ in range(4): y=m(x,t,c)
torch.cuda.synchronize()
