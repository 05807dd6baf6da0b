"""What the host-side derivation of the one-launch cross-attention's operands costs (xattn.context_operands: fp32 products of the fp16 weights with the context, packing
into the kernel's fragment order): once per (context tensor, layer), i.e. once per sampling loop over a prompt set.  usage: python tools/xattn_fold_cost.py [contexts=128]"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import gswm_amd
from gswm_amd import unet as U, xattn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
torch.manual_seed(0)
blk = U.BasicTransformerBlock(320, 1024, 5, 64).cuda().half().eval()
for rep in range(3):
    ctx = torch.randn(n, 77, 1024, device="cuda").half()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    blob, v, idx = xattn.context_operands(blk.attn2, blk.norm2, ctx, torch.float16)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    xattn.context_operands(blk.attn2, blk.norm2, ctx, torch.float16)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{n} contexts, one layer: build {1e3 * (t1 - t0):7.2f} ms (blob {blob.numel() * 2 / 1e6:.1f} MB), cached lookup {1e6 * (t2 - t1):6.1f} us")
