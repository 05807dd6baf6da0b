"""gsw_attention_hd64 vs torch SDPA on the UNet's self-attention shapes.  usage: python tools/attn_bench.py [B]"""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gswm_amd
from gswm_amd import pf
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e-3
print("GSW_ATTN_QB =", os.environ.get("GSW_ATTN_QB", "2 (default)"))
for S, H in ((4096, 5), (1024, 10), (256, 20)):
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B, S, H * 64, generator=g).cuda().half(); k = torch.randn(B, S, H * 64, generator=g).cuda().half(); v = torch.randn(B, S, H * 64, generator=g).cuda().half()
    vt = v.transpose(1, 2).contiguous()
    qq, kk, vv = (a.view(B, S, H, 64).transpose(1, 2) for a in (q, k, v))
    ref = F.scaled_dot_product_attention(qq, kk, vv).transpose(1, 2).reshape(B, S, H * 64)
    got = pf.attention_hd64(q, k, vt, H)
    err = (got.float() - ref.float()).abs().max().item()
    fl = 4.0 * B * H * S * S * 64
    t1 = t(lambda: pf.attention_hd64(q, k, vt, H)); t0 = t(lambda: F.scaled_dot_product_attention(qq, kk, vv))
    print(f"S={S} H={H} B={B}: own {t1*1e3:.3f} ms {fl/t1/1e12:.0f} TFLOP/s | sdpa {t0*1e3:.3f} ms {fl/t0/1e12:.0f} TFLOP/s | max err {err:.2e}", flush=True)

# cross-attention shape: 77 context tokens padded to 128 keys, masked
for S, H in ((4096, 5), (1024, 10), (256, 20)):
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B, S, H * 64, generator=g).cuda().half(); k = torch.randn(B, 77, H * 64, generator=g).cuda().half(); v = torch.randn(B, 77, H * 64, generator=g).cuda().half()
    kp = F.pad(k, (0, 0, 0, 51)); vtp = F.pad(v, (0, 0, 0, 51)).transpose(1, 2).contiguous()
    qq, kk, vv = (a.view(B, a.shape[1], H, 64).transpose(1, 2) for a in (q, k, v))
    ref = F.scaled_dot_product_attention(qq, kk, vv).transpose(1, 2).reshape(B, S, H * 64)
    got = pf.attention_hd64(q, kp, vtp, H, valid_keys=77)
    err = (got.float() - ref.float()).abs().max().item()
    t1 = t(lambda: pf.attention_hd64(q, kp, vtp, H, valid_keys=77)); t0 = t(lambda: F.scaled_dot_product_attention(qq, kk, vv))
    print(f"cross S={S} H={H} B={B}: own {t1*1e3:.3f} ms | sdpa {t0*1e3:.3f} ms | max err {err:.2e}", flush=True)
