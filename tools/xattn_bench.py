"""The cross-attention sublayer at the 320-channel level: ONE launch (gsw_xattn_fused) against the three launches it replaces (LayerNorm-folded query
projection, 77-key attention kernel, output projection + residual).  usage: python tools/xattn_bench.py [B=128] [S=4096] [sd15] [shared]"""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import gswm_amd
from gswm_amd import unet as U, xattn, pf
sd15 = "sd15" in sys.argv
nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
B = nums[0] if nums else 128
S = nums[1] if len(nums) > 1 else (9216 if sd15 else 4096)
heads, hd, cd = (8, 40, 768) if sd15 else (5, 64, 1024)
torch.manual_seed(0)
blk = U.BasicTransformerBlock(320, cd, heads, hd if not sd15 else None)
for p_ in blk.parameters():
    if p_.dim() == 2:
        torch.nn.init.normal_(p_, std=p_.shape[1] ** -0.5)
blk = blk.cuda().half().eval()
x = torch.randn(B, S, 320, device="cuda").half()
ctx = torch.randn(1, 77, cd, device="cuda").half().expand(B, -1, -1) if "shared" in sys.argv else torch.randn(B, 77, cd, device="cuda").half()
xf = x.float(); mean = xf.mean(-1); rstd = torch.rsqrt(xf.var(-1, unbiased=False) + 1e-5)
st = torch.stack([rstd, -rstd * mean], -1).reshape(-1, 2).contiguous(); del xf
a = blk.attn2

def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

with torch.no_grad():
    one = lambda: a.fused_sublayer(x, st, blk.norm2, ctx, eps_next=1e-5)
    three = lambda: a.forward_ln(x, st, blk.norm2, ctx, resid=x)
    y1, y3 = one(), three()
    print("max |one - three| =", (y1.float() - y3.float()).abs().max().item(), " max |y| =", y3.float().abs().max().item())
    us1, us3 = t(one), t(three)
M = B * S
alg = 4.0 * M * 320
print(f"B={B} S={S} heads={heads}: one launch {us1:8.1f} us = {alg / us1 / 1e6:5.2f} TB/s effective, {2.0 * M * 320 * heads * 176 / us1 / 1e6:6.1f} TFLOP/s executed;  three launches {us3:8.1f} us  ({us3 / us1:4.2f} x)")

# gsw_xattn_fused_pre: the self-attention's output projection + bias + residual + norm2 statistics as the launch's prologue, against that projection's own launch
# (with row records) followed by the plain one-launch kernel
o1 = torch.randn(B, S, 320, device="cuda").half()
lin = blk.attn1.to_out[0]
with torch.no_grad():
    def two():
        x1 = U._lin(o1, lin, x, rowstats=True)
        return a.fused_sublayer(x1, pf.ln_stat(x1, blk.norm2.eps), blk.norm2, ctx, eps_next=1e-5)
    proj = lambda: U._lin(o1, lin, x, rowstats=True)
    pre = lambda: a.fused_sublayer(x, None, blk.norm2, ctx, eps_next=1e-5, pre=(o1, lin))
    y2, yp = two(), pre()
    print("max |pre - two launches| =", (yp.float() - y2.float()).abs().max().item())
    usp, us2, usj = t(pre), t(two), t(proj)
print(f"with the self-attention's output projection: prologue form {usp:8.1f} us;  projection launch {usj:7.1f} us + sublayer = {us2:8.1f} us  ({us2 / usp:4.2f}x)")
