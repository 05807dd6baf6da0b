"""Small-batch regime of the eps model: wall time per UNet forward, eager (ctypes launches from Python) against a HIP-graph replay of
the same forward, at 1 / 2 / 8 / 16 rows -- how much of a small-batch forward is launch overhead.  `shapes` adds the per-shape table of the
matmul-engine launches at the given row counts (which launches are tile-count-bound)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import gswm_amd  # noqa: E402
from gswm_amd import unet as U  # noqa: E402

dev, dt = "cuda", torch.float16
m = U.synthetic_init_(U.UNet2DCondition(), 0).to(dev, dt).eval()
rows_list = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 2, 8, 16]
want_shapes = "shapes" in sys.argv


def wall(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for B in rows_list:
    x = torch.randn(B, 4, 64, 64, device=dev, dtype=dt)
    t = torch.full((), 500, device=dev)
    c = torch.randn(B, 77, 1024, device=dev, dtype=dt)
    with torch.no_grad():
        for _ in range(3):
            y_eager = m(x, t, c)
        d_eager = wall(lambda: m(x, t, c), 10)
        # capture: warm-up on a side stream (weight caches, context K/V, workspaces), then one forward into a graph
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                m(x, t, c)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y_graph = m(x, t, c)
        g.replay()
        torch.cuda.synchronize()
        same = bool(torch.equal(y_graph, y_eager))
        d_graph = wall(g.replay, 20)
    print(f"rows={B:3d}: eager {d_eager * 1e3:7.2f} ms  graph {d_graph * 1e3:7.2f} ms  ({B * 0.804 / d_graph:6.0f} TFLOP/s)  graph == eager: {same}", flush=True)
    if want_shapes:
        from gswm_amd import pf
        tm = pf.ConvTimer(by_shape=True)
        pf.CONV_TIMER = tm
        with torch.no_grad():
            for _ in range(3):
                m(x, t, c)
        torch.cuda.synchronize()
        pf.CONV_TIMER = None
        tot = sum(v["ms"] for v in tm.summary().values())
        print(f"  matmul-engine / convolution launches: {tot / 3:.2f} ms per forward (event time, includes launch gaps)")
        for k, v in sorted(tm.summary().items(), key=lambda kv: -kv[1]["ms"])[:40]:
            if k[0] == "gsw_attn_fwd_kernel":          # (kernel, B, Sq, Sk, heads, head_dim)
                name, b, sq, sk, hh, hd = k
                print(f"  {name:28s} Sq={sq:5d} Sk={sk:5d} H={hh:2d} d={hd:3d}  calls/fwd={v['calls'] // 3:3d} avg={v['avg_us']:8.1f} us {v['tflops']:7.1f} TF {v['ms'] / tot * 100:5.1f} %")
            elif len(k) == 7:
                name, b, h, w, kk, n, st = k
                print(f"  {name:28s} {h:3d}x{w:<3d} K={kk:6d} N={n:5d} s{st} calls/fwd={v['calls'] // 3:3d} avg={v['avg_us']:8.1f} us {v['tflops']:7.1f} TF {v['ms'] / tot * 100:5.1f} %")
            else:
                name, m_, kk, n, mode = k
                print(f"  {name + ' ' + mode:28s} M={m_:7d} K={kk:6d} N={n:5d}    calls/fwd={v['calls'] // 3:3d} avg={v['avg_us']:8.1f} us {v['tflops']:7.1f} TF {v['ms'] / tot * 100:5.1f} %")
