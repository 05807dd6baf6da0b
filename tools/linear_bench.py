"""gsw_linear (+bias +resid epilogue) vs torch linear + add on the UNet's residual-carrying linear shapes (B=128 rows)."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gswm_amd
from gswm_amd import pf
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e-3
for M, K, N in ((524288, 320, 320), (524288, 1280, 320), (131072, 640, 640), (131072, 2560, 640), (32768, 1280, 1280), (32768, 5120, 1280),
                (524288, 320, 2560), (131072, 640, 5120), (32768, 1280, 10240)):
    x = torch.randn(M, K, device="cuda", dtype=torch.float16); w = torch.randn(N, K, device="cuda", dtype=torch.float16) * 0.02
    b = torch.randn(N, device="cuda", dtype=torch.float16); r = torch.randn(M, N, device="cuda", dtype=torch.float16)
    fl = 2.0 * M * K * N
    t_lin = t(lambda: F.linear(x, w, b))
    t_lin_add = t(lambda: F.linear(x, w, b) + r)
    t_own = t(lambda: pf.linear(x, w, b))
    t_own_r = t(lambda: pf.linear(x, w, b, resid=r))
    print(f"M={M} K={K} N={N}: torch lin {t_lin*1e6:7.0f} us ({fl/t_lin/1e12:4.0f} TF) lin+add {t_lin_add*1e6:7.0f} us | own {t_own*1e6:7.0f} us ({fl/t_own/1e12:4.0f} TF) own+resid {t_own_r*1e6:7.0f} us", flush=True)
