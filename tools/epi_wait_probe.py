"""Does the epilogue's wait for its own parameter loads (bias: a VMEM load behind the prefetched stages of the next tile) cost the short-K
shapes time?  The same launches with and without a bias vector.  python tools/epi_wait_probe.py [rows=128]"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gswm_amd  # noqa: E402,F401
from gswm_amd import pf  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dt = torch.float16
    for S, C in ((4096, 320), (1024, 640), (256, 1280)):
        M = R * S
        g = torch.Generator().manual_seed(C)
        x = torch.randn(M, C, generator=g).to(dt).cuda()
        for name, N, mode in (("geglu", 8 * C, "geglu"), ("plain", C, "plain"), ("qk", 2 * C, "plain")):
            w = (torch.randn(N, C, generator=g) * C ** -0.5).to(dt).cuda()
            b = torch.randn(N, generator=g).to(dt).cuda()
            if mode == "geglu":
                wp, bp = pf.pack_geglu_weight(w, b)
                t1 = timeit(lambda: pf.gemm(x, wp, bp, mode="geglu"))
                t0 = timeit(lambda: pf.gemm(x, wp, None, mode="geglu"))
            else:
                t1 = timeit(lambda: pf.gemm(x, w, b))
                t0 = timeit(lambda: pf.gemm(x, w, None))
            print(f"M={M:7d} K={C:5d} N={N:6d} {name:6s} with bias {t1:8.1f} us   without {t0:8.1f} us   ({100 * (t1 - t0) / t1:+.1f} %)", flush=True)


if __name__ == "__main__":
    main()
