"""Split-K policy sweep: per shape, the time of one launch (graph-captured back-to-back launches, so the host launch path is out of the picture)
unsplit and at forced split counts.  Decides the automatic rule of gsw_mm_launch (csrc/gswm_mm.hip)."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import gswm_amd  # noqa: E402
from gswm_amd import pf  # noqa: E402

dt = torch.float16
REP = 20


def timed(fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    ws = torch.empty(pf.SPLITK_BYTES, dtype=torch.uint8, device="cuda")
    with pf.splitk_workspace(ws), torch.cuda.graph(g):
        for _ in range(REP):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * REP)


def sweep(name, fn, ks=(1, 0, 2, 4, 8, 16, 32)):
    out = []
    for k in ks:
        pf.SPLITK_MAX = k
        out.append((k, timed(fn)))
    pf.SPLITK_MAX = 0
    print(f"{name:44s} " + "  ".join(f"{'auto' if k == 0 else 'off' if k == 1 else k}:{t:7.1f}" for k, t in out), flush=True)


for B in (1, 2, 8, 16):
    for (C, N, H) in ((1280, 1280, 8), (2560, 1280, 8), (1280, 1280, 16), (2560, 1280, 16), (640, 640, 32), (1920, 640, 32), (320, 320, 64)):
        x = pf.PF.from_nchw(torch.randn(B, C, H, H, device="cuda", dtype=dt))
        w = pf.pack_conv_weight((torch.randn(N, C, 3, 3, device="cuda") * (9 * C) ** -0.5).to(dt))
        b = torch.randn(N, device="cuda", dtype=dt)
        sweep(f"conv3x3 B={B} {H}x{H} C={C} N={N}", lambda: pf.conv_pf(x, w, b))
    for (S, K, N) in ((64, 1280, 1280), (64, 5120, 1280), (256, 1280, 1280), (256, 5120, 1280), (1024, 640, 640), (1024, 2560, 640), (4096, 320, 320), (4096, 1280, 320)):
        xx = torch.randn(B * S, K, device="cuda", dtype=dt)
        ww = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
        sweep(f"dense B={B} M={B * S} K={K} N={N}", lambda: pf.gemm(xx, ww, None))
    xx = torch.randn(B, 1280, device="cuda", dtype=dt)
    ww = (torch.randn(1280, 1280, device="cuda") * 1280 ** -0.5).to(dt)
    sweep(f"time_emb_proj M={B} K=1280 N=1280", lambda: pf.gemm(xx, ww, None))
