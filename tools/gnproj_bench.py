"""GroupNorm + proj_in of the 320-channel transformers: ONE launch (gsw_gn_proj_tokens) against gsw_gn_pf_apply (tokens) + the engine's 320 x 320 GEMM.
usage: python tools/gnproj_bench.py [B=128] [H=64]"""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import gswm_amd
from gswm_amd import pf, xattn
nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
B = nums[0] if nums else 128
H = nums[1] if len(nums) > 1 else 64
dt = torch.float16
g = torch.Generator().manual_seed(0)
src = pf.PF.from_nchw(torch.randn(B, 64, H, H, generator=g).to(dt).cuda())
cw = pf.pack_conv_weight((torch.randn(320, 64, 3, 3, generator=g) * 0.06).to(dt).cuda()); cb = torch.zeros(320, dtype=dt, device="cuda")
x = pf.conv_pf(src, cw, cb)
norm = torch.nn.GroupNorm(32, 320, eps=1e-6).to(dt).cuda(); lin = torch.nn.Linear(320, 320).to(dt).cuda()

def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

with torch.no_grad():
    assert xattn.gn_proj_usable(x, norm, lin)
    one = lambda: xattn.gn_proj(x, norm, lin, eps_next=1e-5)
    def two():
        y = pf.gemm(pf.groupnorm_pf(x, norm.weight, norm.bias, 32, norm.eps, act=False, tokens=True), lin.weight, lin.bias, rowstats=True)
        return y, pf.ln_stat(y, 1e-5)
    y1, (y2, _) = one(), two()
    print("max |one - two| =", (y1.float() - y2.float()).abs().max().item(), " max |y| =", y2.float().abs().max().item())
    for i in range(3):
        u1, u2 = t(one), t(two)
        M = B * H * H
        print(f"B={B} {H}x{H}: one launch {u1:7.1f} us = {2.0 * (B * (H + 2) ** 2 + M) * 320 / u1 / 1e6:5.2f} TB/s effective;  apply + GEMM (+ record finish) {u2:7.1f} us  ({u2 / u1:4.2f} x)")
