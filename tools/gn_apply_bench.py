"""GroupNorm behind a convolution whose epilogue left the column records (the eps model's path: gsw_gn_colstats_finish + gsw_gn_pf_apply): time and effective HBM rate
(1 read + 1 write of the tensor) per shape at B rows.  usage: python tools/gn_apply_bench.py [B]"""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gswm_amd
from gswm_amd import pf
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e-3
for H, C in ((64, 320), (32, 640), (16, 1280), (8, 1280), (32, 320), (16, 640)):
    g = torch.Generator().manual_seed(0)
    x = pf.PF.from_nchw(torch.randn(B, 64, H, H, generator=g).half().cuda())
    w = pf.pack_conv_weight((torch.randn(C, 64, 3, 3, generator=g) * 0.04).half().cuda()); b = torch.zeros(C).half().cuda()
    y = pf.conv_pf(x, w, b)
    assert y.stats is not None
    ga = torch.ones(C, device="cuda", dtype=torch.float16); be = torch.zeros(C, device="cuda", dtype=torch.float16)
    for tokens in (False, True):
        dt = t(lambda: pf.groupnorm_pf(y, ga, be, 32, 1e-5, act=not tokens, tokens=tokens))
        rd = y.rows.numel() * 2; wr = (B * H * H * C * 2) if tokens else rd
        print(f"{H:2d}x{H:<2d} C={C:4d} B={B} -> {'tokens' if tokens else 'PF    '}: finish + apply {dt*1e6:7.1f} us  {(rd + wr)/dt/1e12:5.2f} TB/s (read {rd/1e6:.0f} MB + write {wr/1e6:.0f} MB)", flush=True)
