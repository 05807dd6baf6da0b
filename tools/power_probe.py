"""Clock and power while a kernel family runs for seconds: is a kernel's time set by a unit's throughput or by the board's power limit?
A sampler thread reads the GPU's sysfs sensors (hwmon power1_average / power1_input, freq1_input = shader clock; falls back to `rocm-smi --json`) every 20 ms while
the main thread replays one workload back to back; per workload: kernel time, mean / min shader clock, mean / max power.
usage: python tools/power_probe.py [attn|conv|dense|geglu|mm|all]      (mm: the convolution and dense lines on random operands only)
      (GSWM_LIB selects a side build, e.g. an ablation build of the matmul engine: tools/mm_ablate.sh)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gswm_amd  # noqa: F401,E402
from gswm_amd import pf  # noqa: E402


from bench_board import BoardSampler  # noqa: E402


def run(name, fn, seconds=3.0, flops=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0, n = time.time(), 0
    with BoardSampler(0, 0.02) as sm:
        e0.record()
        while time.time() - t0 < seconds:
            for _ in range(20):
                fn()
            n += 20
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
    mhz, w = sm.mhz[len(sm.mhz) // 4:], sm.w[len(sm.w) // 4:]          # the last three quarters: the governor has settled
    ms = e0.elapsed_time(e1) / n
    tf = f" {flops / ms / 1e9:7.0f} TFLOP/s" if flops else ""
    print(f"{name:58s} {ms:8.3f} ms{tf} | clock mean {sum(mhz) / max(len(mhz), 1):6.0f} min {min(mhz, default=0):6.0f} MHz | power mean {sum(w) / max(len(w), 1):6.0f} max {max(w, default=0):6.0f} W", flush=True)


torch.cuda.init()
print("sensors:", BoardSampler(0).sens, "| lib:", os.environ.get("GSWM_LIB", "(in-tree)"), flush=True)
what = sys.argv[1] if len(sys.argv) > 1 else "all"
dt = torch.bfloat16 if "bf16" in sys.argv else torch.float16          # (bf16: same kernels, 8-bit mantissas -- how much of the power is the multipliers' width?)
time.sleep(1.0)
with BoardSampler(0, 0.02) as sm0:
    time.sleep(0.3)
print("idle:", sm0.summary(), flush=True)
if what in ("attn", "all"):
    B, S, H = 128, 4096, 5
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(B, S, H * 64, generator=g).cuda().to(dt) for _ in range(3))
    vt = v.transpose(1, 2).contiguous()
    run("attention S=4096 H=5 B=128 (random operands)", lambda: pf.attention_hd64(q, k, vt, H), flops=4.0 * B * H * S * S * 64)
    z = torch.zeros_like(q); zt = torch.zeros_like(vt)
    run("attention S=4096 H=5 B=128 (all-zero operands)", lambda: pf.attention_hd64(z, z, zt, H), flops=4.0 * B * H * S * S * 64)
if what in ("conv", "all", "mm"):
    for (C, N, Hh, B) in ((320, 320, 64, 128), (1280, 1280, 16, 128)):
        x = pf.PF.from_nchw(torch.randn(B, C, Hh, Hh, device="cuda", dtype=dt))
        w = pf.pack_conv_weight((torch.randn(N, C, 3, 3, device="cuda") * (9 * C) ** -0.5).to(dt))
        b = torch.randn(N, device="cuda", dtype=dt)
        run(f"conv3x3 {Hh}x{Hh} C={C} N={N} B={B} (random operands)", lambda: pf.conv_pf(x, w, b), flops=2.0 * B * Hh * Hh * 9 * C * N)
        if what == "mm":
            continue
        xz = pf.PF.from_nchw(torch.zeros(B, C, Hh, Hh, device="cuda", dtype=dt)); wz = torch.zeros_like(w)
        run(f"conv3x3 {Hh}x{Hh} C={C} N={N} B={B} (all-zero operands)", lambda: pf.conv_pf(xz, wz, b), flops=2.0 * B * Hh * Hh * 9 * C * N)
if what in ("geglu", "all"):
    # the transformer's GEGLU projection at 64 x 64 / 32 x 32 (five K stages, then the gelu epilogue with the matrix pipe idle): is THIS one at the power limit?
    for (M, K, I) in ((524288, 320, 1280), (131072, 640, 2560), (32768, 1280, 5120)):
        x = torch.randn(M, K, device="cuda", dtype=dt)
        w = (torch.randn(2 * I, K, device="cuda") * K ** -0.5).to(dt); b = torch.randn(2 * I, device="cuda", dtype=dt)
        wp, bp = pf.pack_geglu_weight(w, b)
        out = torch.empty(M, I, device="cuda", dtype=dt)
        run(f"geglu M={M} K={K} N={2 * I} (random operands)", lambda: pf.gemm(x, wp, bp, mode="geglu", out=out), flops=2.0 * M * K * 2 * I)
        del x, out
if what in ("dense", "all", "mm"):
    for (M, K, N) in ((524288, 320, 320), (32768, 5120, 1280)):
        x = torch.randn(M, K, device="cuda", dtype=dt); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
        run(f"dense M={M} K={K} N={N} (random operands)", lambda: pf.gemm(x, w, None), flops=2.0 * M * K * N)
        if what == "mm":
            continue
        xz = torch.zeros_like(x); wz = torch.zeros_like(w)
        run(f"dense M={M} K={K} N={N} (all-zero operands)", lambda: pf.gemm(xz, wz, None), flops=2.0 * M * K * N)
