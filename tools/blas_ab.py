"""UNet forward with hipBLASLt vs rocBLAS as torch's GEMM backend (in-process A/B)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gswm_amd
from gswm_amd import unet as U
B = 128
m = U.synthetic_init_(U.UNet2DCondition(), 0).to("cuda", torch.float16).eval()
x = torch.randn(B, 4, 64, 64, device="cuda", dtype=torch.float16); t = torch.full((), 500, device="cuda"); c = torch.randn(B, 77, 1024, device="cuda", dtype=torch.float16)
def run(n=4):
    with torch.no_grad():
        for _ in range(2): m(x, t, c)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): m(x, t, c)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
print("default preferred:", torch.backends.cuda.preferred_blas_library())
for rep in range(2):
    for lib in ("cublaslt", "cublas"):
        torch.backends.cuda.preferred_blas_library(lib)
        d = run()
        print(f"rep {rep} {lib:9s}: {d*1e3:7.1f} ms {B*0.804/d:6.0f} TFLOP/s", flush=True)
