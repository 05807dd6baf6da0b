"""one-launch cross-attention, a few launches on fixed operands (for rocprofv3 --pmc / --kernel-trace passes).  usage: python tools/_xattn_once.py [B=128] [n=5]"""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import gswm_amd
from gswm_amd import unet as U, xattn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
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
with torch.no_grad():
    for _ in range(n):
        y = blk.attn2.fused_sublayer(x, st, blk.norm2, ctx, eps_next=1e-5)
torch.cuda.synchronize()
print("ok", float(y.float().abs().max()))
