"""one-launch cross-attention, a few launches on fixed operands (for rocprofv3 --pmc / --kernel-trace passes).  usage: python tools/_xattn_once.py [B=128] [n=5] [pre]  (pre: the prologue form, gsw_xattn_fused_pre)"""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import gswm_amd
from gswm_amd import unet as U, xattn
nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
B = nums[0] if nums else 128
n = nums[1] if len(nums) > 1 else 5
pre = 'pre' in sys.argv
S, heads = 4096, 5
torch.manual_seed(0)
blk = U.BasicTransformerBlock(320, 1024, heads, 64)
for p_ in blk.parameters():
    if p_.dim() == 2:
        torch.nn.init.normal_(p_, std=p_.shape[1] ** -0.5)
blk = blk.cuda().half().eval()
x = torch.randn(B, S, 320, device="cuda").half()
ctx = torch.randn(B, 77, 1024, device="cuda").half()
xf = x.float(); mean = xf.mean(-1); rstd = torch.rsqrt(xf.var(-1, unbiased=False) + 1e-5)
st = torch.stack([rstd, -rstd * mean], -1).reshape(-1, 2).contiguous(); del xf
o1 = torch.randn(B, S, 320, device="cuda").half()
with torch.no_grad():
    for _ in range(n):
        y = blk.attn2.fused_sublayer(x, None, blk.norm2, ctx, eps_next=1e-5, pre=(o1, blk.attn1.to_out[0])) if pre else blk.attn2.fused_sublayer(x, st, blk.norm2, ctx, eps_next=1e-5)
torch.cuda.synchronize()
print("ok", float(y.float().abs().max()))
