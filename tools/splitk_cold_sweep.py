"""Split-K policy at one / two images with HBM-COLD weights (as in a forward: 1.7 GB of weights go by between two uses of a layer): per shape the time of
partial + reduce launches at forced split counts against the automatic rule.  Launches are graph-captured back to back and cycle through weight copies."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import gswm_amd  # noqa: E402
from gswm_amd import pf  # noqa: E402

dt = torch.float16
REP = 24


def timed(fn):
    fn(0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    ws = torch.empty(pf.SPLITK_BYTES, dtype=torch.uint8, device="cuda")
    with pf.splitk_workspace(ws), torch.cuda.graph(g):
        for i in range(REP):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (4 * REP)


def sweep(name, fn, ks):
    out = []
    for k in ks:
        pf.SPLITK_MAX = k
        out.append((k, timed(fn)))
    pf.SPLITK_MAX = 0
    print(f"{name:40s} " + "  ".join(f"{'auto' if k == 0 else 'off' if k == 1 else k}:{t:6.1f}" for k, t in out), flush=True)


pf.GN_FUSED_MAX_WGS = 0          # keep the column-record epilogue out of the picture
for B in (1, 2):
    for (C, N, H) in ((1280, 1280, 8), (2560, 1280, 8), (1280, 1280, 16), (2560, 1280, 16), (1920, 1280, 16), (640, 640, 32), (1920, 640, 32), (1280, 640, 32), (320, 320, 64), (960, 320, 64), (640, 320, 64)):
        nW = max(2, min(24, (700 << 20) // (N * 9 * C * 2)))
        x = pf.PF.from_nchw(torch.randn(B, C, H, H, device="cuda", dtype=dt))
        ws = [pf.pack_conv_weight((torch.randn(N, C, 3, 3, device="cuda") * (9 * C) ** -0.5).to(dt)) for _ in range(nW)]
        b = torch.randn(N, device="cuda", dtype=dt)
        sweep(f"conv3x3 B={B} {H}x{H} C={C} N={N}", lambda i: pf.conv_pf(x, ws[i % nW], b), (0, 1, 4, 8, 16, 24, 32, 48, 64))
    for (S, K, N) in ((256, 5120, 1280), (1024, 2560, 640), (4096, 1280, 320), (256, 1280, 1280), (1024, 640, 640)):
        nW = max(2, min(24, (700 << 20) // (N * K * 2)))
        xx = torch.randn(B * S, K, device="cuda", dtype=dt)
        ws = [(torch.randn(N, K, device="cuda") * K ** -0.5).to(dt) for _ in range(nW)]
        sweep(f"dense B={B} M={B * S} K={K} N={N}", lambda i: pf.gemm(xx, ws[i % nW], None), (0, 1, 2, 4, 8, 16, 32))
