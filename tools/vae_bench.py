"""VAE decode + encode time per image at 512x512 (PF kernels vs torch/MIOpen).  usage: python tools/vae_bench.py [B]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gswm_amd
from gswm_amd import vae as V
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
v = V.synthetic_init_(V.AutoencoderKL(), 1).cuda().half().eval()
z = torch.randn(B, 4, 64, 64, device="cuda", dtype=torch.float16)
x = torch.rand(B, 3, 512, 512, device="cuda", dtype=torch.float16) * 2 - 1
for use in (True, False):
    V.USE_PF = use
    with torch.no_grad():
        for _ in range(2): v.decode(z); v.encode_mean(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): v.decode(z)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(3): v.encode_mean(x)
        torch.cuda.synchronize(); t2 = time.perf_counter()
    d, e = (t1 - t0) / 3 / B, (t2 - t1) / 3 / B
    print(f"USE_PF={use}: decode {d*1e3:.2f} ms/image ({1.24/d:.0f} TFLOP/s)  encode {e*1e3:.2f} ms/image ({0.57/e:.0f} TFLOP/s)  mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
