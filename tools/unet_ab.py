"""A/B of UNet forward variants inside ONE process (box-to-box variance is +-5 %): python tools/unet_ab.py [B]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gswm_amd
from gswm_amd import unet as U
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
m = U.synthetic_init_(U.UNet2DCondition(), 0).to("cuda", torch.float16).eval()
x = torch.randn(B, 4, 64, 64, device="cuda", dtype=torch.float16); t = torch.full((), 500, device="cuda"); c = torch.randn(B, 77, 1024, device="cuda", dtype=torch.float16)
def run(n=4):
    with torch.no_grad():
        for _ in range(2): m(x, t, c)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): m(x, t, c)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
variants = {"default": dict(GEGLU_GEMM_MAX_K=0), "geglu fused K<=320": dict(GEGLU_GEMM_MAX_K=320), "geglu fused K<=640": dict(GEGLU_GEMM_MAX_K=640),
            "geglu fused all": dict(GEGLU_GEMM_MAX_K=1280)}
if len(sys.argv) > 2:
    variants = eval(sys.argv[2])
for rep in range(3):
    for name, flags in variants.items():
        for k, v in flags.items(): setattr(U, k, v)
        d = run()
        print(f"rep {rep} {name:22s}: {d*1e3:7.1f} ms  {B*0.804/d:6.0f} TFLOP/s", flush=True)
