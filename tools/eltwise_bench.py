"""add_layernorm / geglu kernels: achieved HBM GB/s at the UNet's token shapes, B=128."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gswm_amd
from gswm_amd import codec
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e-3
B = 128
for S, C in ((4096, 320), (1024, 640), (256, 1280), (64, 1280)):
    x = torch.randn(B, S, C, device="cuda", dtype=torch.float16); r = torch.randn_like(x)
    g = torch.ones(C, device="cuda", dtype=torch.float16); b = torch.zeros(C, device="cuda", dtype=torch.float16)
    d1 = t(lambda: codec.add_layernorm(x, r, g, b, 1e-5)); d0 = t(lambda: codec.add_layernorm(x, None, g, b, 1e-5))
    y = torch.randn(B, S, 8 * C, device="cuda", dtype=torch.float16)
    d2 = t(lambda: codec.geglu(y))
    n = x.numel() * 2
    print(f"S={S} C={C}: add+LN {d1*1e6:7.1f} us ({4*n/d1/1e9:5.0f} GB/s)  LN only {d0*1e6:7.1f} us ({2*n/d0/1e9:5.0f} GB/s)  geglu {d2*1e6:7.1f} us ({12*n/d2/1e9:5.0f} GB/s)", flush=True)
