import sys, torch, time
sys.path.insert(0,'.')
import torch.nn.functional as F
import gswm_amd
from gswm_amd import pf as P, codec
dev='cuda'; dt=torch.float16
def tm(f, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/it*1e3
for (T,K,N) in ((524288,320,320),(524288,320,960),(131072,640,640),(131072,640,1920),(9856,1024,320)):
    x=torch.randn(T,K,device=dev,dtype=dt); w=torch.randn(N,K,device=dev,dtype=dt)*0.05; b=torch.randn(N,device=dev,dtype=dt); r=torch.randn(T,N,device=dev,dtype=dt)
    t0=tm(lambda: F.linear(x,w,b)); t1=tm(lambda: P.linear(x,w,b)) if K<=640 or True else 0
    t2=tm(lambda: F.linear(x,w,b)+r); t3=tm(lambda: P.linear(x,w,b,resid=r))
    print(f'T={T} K={K} N={N}: torch {t0:.0f} us ({2*T*K*N/t0/1e6:.0f} TF) own {t1:.0f} us ({2*T*K*N/t1/1e6:.0f} TF) | +resid torch {t2:.0f} own {t3:.0f}', flush=True)
for (T,K,N) in ((524288,320,2560),(131072,640,5120)):
    x=torch.randn(T,K,device=dev,dtype=dt); w=torch.randn(N,K,device=dev,dtype=dt)*0.05; b=torch.randn(N,device=dev,dtype=dt)
    wp,bp=P.pack_geglu_weight(w,b)
    t0=tm(lambda: codec.geglu(F.linear(x,w,b))); t1=tm(lambda: P.linear(x,wp,bp,geglu=True))
    print(f'GEGLU T={T} K={K} N={N}: torch+geglu kernel {t0:.0f} us, own fused {t1:.0f} us', flush=True)
