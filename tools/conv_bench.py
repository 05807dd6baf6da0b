"""Per-shape throughput of gsw_conv_pf on the UNet's 3x3 shapes (B = 128 rows) against torch/MIOpen."""
import sys, torch, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
import gswm_amd
from gswm_amd import pf as P
dev='cuda'
def tm(f, it=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/it
def run(B,C,O,H,k=3,stride=1,dt=torch.float16, ref_time=False):
    g=torch.Generator().manual_seed(C+O+H)
    x=torch.randn(B,C,H,H,generator=g).to(dt).to(dev); w=(torch.randn(O,C,k,k,generator=g)*(1.0/(C*k*k))**0.5).to(dt).to(dev); b=torch.randn(O,generator=g).to(dt).to(dev)
    Ho=H//stride
    ref=F.conv2d(x,w,b,padding=k//2,stride=stride).float()
    xp=P.PF.from_nchw(x); wp=P.pack_conv_weight(w)
    y=P.conv_pf(xp,wp,b,ksize=k,stride=stride)
    err=(y.to_nchw().float()-ref).abs().max().item()/ref.abs().max().item()
    t=tm(lambda: P.conv_pf(xp,wp,b,ksize=k,stride=stride))
    fl=2*B*Ho*Ho*C*O*k*k
    msg=f'{os.environ.get("GSW_CONV_WM","")} C={C:5d} O={O:5d} H={H:3d}: relerr {err:.1e} | gsw {t*1e3:7.0f} us {fl/t/1e9:5.0f} TF'
    if ref_time:
        tr=tm(lambda: F.conv2d(x,w,b,padding=k//2,stride=stride)); msg+=f' | miopen {tr*1e3:7.0f} us {fl/tr/1e9:5.0f} TF'
    print(msg, flush=True)
if __name__ == "__main__":
    B=128
    for (C,O,H) in ((320,320,64),(960,320,64),(640,640,32),(1920,640,32),(1280,1280,16),(2560,1280,16),(1280,1280,8)):
        run(B,C,O,H,ref_time="--ref" in sys.argv)
