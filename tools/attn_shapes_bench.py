"""Self-attention of the eps model at every (head_dim, keys) pair the two BASELINE UNet shapes produce, with the board's clock and power next to the time:
SD 1.5 at 768 x 768 (configs[4]: 8 heads of width 40 / 80 / 160 over 9216 / 2304 / 576 / 144 tokens) and SD 2.1 at 512 x 512 (head_dim 64).
Random against all-zero operands tells whether padded MFMA work (head_dim 40 -> 64 output rows) costs time or only issue slots: zero operands are cheap in power.
usage: python tools/attn_shapes_bench.py [B] [sd15|sd21|all] [zeros]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gswm_amd  # noqa: F401,E402
from gswm_amd import pf  # noqa: E402
from bench_board import BoardSampler  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
what = sys.argv[2] if len(sys.argv) > 2 else "all"
zeros = "zeros" in sys.argv
SHAPES = {"sd15": ((40, 9216, 8), (80, 2304, 8), (160, 576, 8), (160, 144, 8)), "sd21": ((64, 4096, 5), (64, 1024, 10), (64, 256, 20), (64, 64, 20))}


def run(name, fn, flops, seconds=1.5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0, n = time.time(), 0
    with BoardSampler(0, 0.02) as sm:
        e0.record()
        while time.time() - t0 < seconds:
            for _ in range(10):
                fn()
            n += 10
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
    mhz, w = sm.mhz[len(sm.mhz) // 4:], sm.w[len(sm.w) // 4:]
    ms = e0.elapsed_time(e1) / n
    print(f"{name:64s} {ms:8.3f} ms {flops / ms / 1e9:7.0f} algorithmic TFLOP/s | clock {sum(mhz) / max(len(mhz), 1):5.0f} MHz | power {sum(w) / max(len(w), 1):5.0f} W", flush=True)
    return ms


torch.cuda.init()
print(f"B = {B} rows (images x CFG); lib: {os.environ.get('GSWM_LIB', '(in-tree)')}", flush=True)
for fam in ("sd15", "sd21"):
    if what not in (fam, "all"):
        continue
    for d, S, H in SHAPES[fam]:
        g = torch.Generator().manual_seed(0)
        q, k, v = (torch.randn(B, S, H * d, generator=g).cuda().half() for _ in range(3))
        vt = v.transpose(1, 2).contiguous()
        fl = 4.0 * B * H * S * S * d
        # accuracy against fp32 on the first two rows (the parity tests hold the bound; this is a sanity line next to the speed)
        ref = torch.nn.functional.scaled_dot_product_attention(*(a[:2].float().view(2, S, H, d).transpose(1, 2) for a in (q, k, v))).transpose(1, 2).reshape(2, S, H * d)
        err = (pf.attention(q[:2].contiguous(), k[:2].contiguous(), vt[:2].contiguous(), H).float() - ref).abs().max().item()
        run(f"{fam} head_dim {d:3d} S={S:5d} H={H:2d} random (max err {err:.1e})", lambda: pf.attention(q, k, vt, H), fl)
        if zeros:
            z, zt = torch.zeros_like(q), torch.zeros_like(vt)
            run(f"{fam} head_dim {d:3d} S={S:5d} H={H:2d} all-zero operands", lambda: pf.attention(z, z, zt, H), fl)
        del q, k, v, vt
