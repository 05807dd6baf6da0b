"""Microbenchmark of the matmul engine (gsw_gemm) on the eps model's transformer-linear shapes, next to the library GEMM torch calls
(hipBLASLt) for the same function.  python tools/gemm_bench.py [rows=128] [iters=20] > gpurun_out/gemm_bench.txt"""
import os
import sys
import json

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gswm_amd  # noqa: E402
from gswm_amd import pf  # noqa: E402


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dt = torch.float16
    rows = []
    # (name, tokens per image, C): the three transformer levels + mid block of the SD 2.1 UNet at 64 x 64 latents
    for lvl, S, C in (("L0", 4096, 320), ("L1", 1024, 640), ("L2", 256, 1280), ("mid", 64, 1280)):
        M = R * S
        shapes = [("proj", M, C, C, "plain", False), ("proj+res", M, C, C, "plain", True), ("qk", M, C, 2 * C, "plain", False),
                  ("vT", M, C, C, "trans", False), ("ff1+geglu", M, C, 8 * C, "geglu", False), ("ff2+res", M, 4 * C, C, "plain", True)]
        for name, M_, K, N, mode, res in shapes:
            g = torch.Generator().manual_seed(K + N)
            x = torch.randn(M_, K, generator=g).to(dt).cuda()
            w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dt).cuda()
            b = torch.randn(N, generator=g).to(dt).cuda()
            r = torch.randn(M_, N, generator=g).to(dt).cuda() if res else None
            flops = 2.0 * M_ * K * N
            if mode == "geglu":
                wp, bp = pf.pack_geglu_weight(w, b)
                own = lambda: pf.gemm(x, wp, bp, mode="geglu")
                from gswm_amd import codec
                lib = lambda: codec.geglu(F.linear(x, w, b))
                byts = 2.0 * (M_ * K + M_ * N // 2 + N * K)
            elif mode == "trans":
                x3 = x.view(R, S, K)
                own = lambda: pf.gemm(x3, w, None, mode="trans", tokens=S)
                lib = lambda: torch.bmm(w.unsqueeze(0).expand(R, -1, -1), x3.transpose(1, 2))
                byts = 2.0 * (M_ * K + M_ * N + N * K)
            else:
                own = lambda: pf.gemm(x, w, b, resid=r)
                lib = (lambda: F.linear(x, w, b) + r) if res else (lambda: F.linear(x, w, b))
                byts = 2.0 * (M_ * K + M_ * N * (2 if res else 1) + N * K)
            res_row = {"level": lvl, "op": name, "M": M_, "K": K, "N": N}
            t = timeit(own, iters)
            res_row["wm4_us"] = t * 1e6
            res_row["wm4_tflops"] = flops / t / 1e12
            res_row["wm4_tbs"] = byts / t / 1e12
            t = timeit(lib, iters) if "nolib" not in sys.argv else float("nan")
            res_row["lib_us"] = t * 1e6
            res_row["lib_tflops"] = flops / t / 1e12
            rows.append(res_row)
            print(f"{lvl:4s} {name:10s} M={M_:7d} K={K:5d} N={N:5d}  own256x160 {res_row['wm4_tflops']:7.1f} TF ({res_row['wm4_tbs']:.2f} TB/s, {res_row['wm4_us']:8.1f} us)"
                  f"  | torch/hipBLASLt {res_row['lib_tflops']:7.1f} TF ({res_row['lib_us']:8.1f} us)", flush=True)
            del x, w, b, r
            torch.cuda.empty_cache()
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/gemm_bench_r{R}.json", "w") as f:
        json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
