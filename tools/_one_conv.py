import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','.'))
import gswm_amd
from gswm_amd import pf as P
B,C,O,H=128,960,320,64
x=torch.randn(B,C,H,H,device='cuda',dtype=torch.float16); w=(torch.randn(O,C,3,3,device='cuda',dtype=torch.float16)*0.01); b=torch.randn(O,device='cuda',dtype=torch.float16)
xp=P.PF.from_nchw(x); wp=P.pack_conv_weight(w)
for _ in range(4): y=P.conv_pf(xp,wp,b)
torch.cuda.synchronize()
