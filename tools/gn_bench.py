"""GroupNorm PF kernels: achieved HBM GB/s (stats: 1 read; apply: 1 read + 1 write) at the UNet's shapes, B=128."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gswm_amd
from gswm_amd import pf
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e-3
B = 128
for H, C in ((64, 320), (64, 960), (32, 640), (32, 1920), (16, 1280), (16, 2560), (8, 1280)):
    x = pf.PF.zeros(B, H, H, C, torch.float16, "cuda"); x.interior.normal_()
    g = torch.ones(C, device="cuda", dtype=torch.float16); b = torch.zeros(C, device="cuda", dtype=torch.float16)
    dt = t(lambda: pf.groupnorm_pf(x, g, b, 32, 1e-5))
    byts = x.M * C * 2
    print(f"{H}x{H} C={C}: stats+apply {dt*1e6:7.1f} us  -> {3*byts/dt/1e9:7.0f} GB/s over the 3 passes", flush=True)
