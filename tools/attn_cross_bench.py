"""77-key cross-attention (keys padded to 128) on gsw_attention at the three UNet levels.  usage: [GSW_ATTN_QB=1|2|3] python tools/attn_cross_bench.py [B=128]"""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import gswm_amd
from gswm_amd import pf
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for S, H in ((4096, 5), (1024, 10), (256, 20)):
    q = torch.randn(B, S, H * 64, device="cuda").half()
    k = torch.randn(B, 128, H * 64, device="cuda").half()
    vt = torch.randn(B, H * 64, 128, device="cuda").half()
    us = t(lambda: pf.attention(q, k, vt, H, valid_keys=77))
    print(f"QB={os.environ.get('GSW_ATTN_QB', 'auto')} B={B} Sq={S} H={H}: {us:7.1f} us  {4.0 * B * S * H * 64 * 2 / us / 1e6:5.2f} TB/s (q in + o out)")
