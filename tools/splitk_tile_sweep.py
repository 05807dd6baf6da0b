"""Calibration of the split-K plan at 4-64 images (csrc/gswm_mm.hip: mm_plan): time of one launch (+ its reduce) by tile rows (gsw_mm_config) and forced split count,
HBM-cold weights (the launches cycle through weight copies), graph-captured back to back.  Dense proxies of the deep convolutions (same engine loop; M = interior
pixels, K = 9 C_in) and the long-K linears.  Prints measured microseconds next to the plan's prediction (gsw_mm_predict_us is internal: the model is restated here)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import gswm_amd  # noqa: E402,F401
from gswm_amd import pf, _native  # noqa: E402

lib = _native.lib()
dt = torch.float16
REP = 12


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
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * REP)


pf.SMALL_GEMM_MAX_ROWS = 0
shapes = []
for rows in (4, 8, 16, 32, 64):
    for (px, N, Ks) in ((64, 1280, (11520, 23040)), (256, 1280, (11520, 17280, 23040)), (1024, 640, (5760, 11520, 17280))):
        for K in Ks:
            if rows * px * N <= 8192 * 1280:
                shapes.append((rows * px, K, N))
    shapes.append((rows * 256, 5120, 1280))
    shapes.append((rows * 1024, 2560, 640))
seen = set()
for (M, K, N) in shapes:
    if (M, K, N) in seen:
        continue
    seen.add((M, K, N))
    nW = max(2, min(16, (600 << 20) // (N * K * 2)))
    x = torch.randn(M, K, device="cuda", dtype=dt)
    ws_ = [(torch.randn(N, K, device="cuda") * K ** -0.5).to(dt) for _ in range(nW)]
    out = []
    for bm in (128, 256):
        assert lib.gsw_mm_config(bm, -1) == 0
        nt = -(-M // bm) * -(-N // 160)
        for k in (1, 2, 3, 4, 6, 8, 12, 16):
            if k > 1 and (k * nt > 256 or 2 * k > K // 64):
                continue
            pf.SPLITK_MAX = k
            out.append((bm, k, timed(lambda i: pf.gemm(x, ws_[i % nW], None))))
    assert lib.gsw_mm_config(0, -1) == 0
    pf.SPLITK_MAX = 0
    pf.LAUNCH_LOG = log = []
    t_auto = timed(lambda i: pf.gemm(x, ws_[i % nW], None))
    pf.LAUNCH_LOG = None
    best = min(out, key=lambda r: r[2])
    print(f"M={M:6d} K={K:6d} N={N:5d} | auto {t_auto:7.1f} us (splits {log[0].splits}) | best bm={best[0]} k={best[1]:2d} {best[2]:7.1f} | "
          + "  ".join(f"{bm}/{k}:{t:6.1f}" for bm, k, t in out), flush=True)
    del x, ws_
