"""VAE decode + encode of B images at 512 x 512 on the hand-written path, a few iterations (for rocprofv3 --kernel-trace --stats).  usage: python tools/_vae_once.py [B=16]"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import gswm_amd
from gswm_amd import vae as V
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
v = V.synthetic_init_(V.AutoencoderKL(), 1).cuda().half().eval()
z = torch.randn(B, 4, 64, 64, device="cuda", dtype=torch.float16)
x = torch.rand(B, 3, 512, 512, device="cuda", dtype=torch.float16) * 2 - 1
with torch.no_grad():
    for _ in range(2): v.decode(z); v.encode_mean(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): v.decode(z)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(3): v.encode_mean(x)
    torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"B={B}: decode {(t1 - t0) / 3 / B * 1e3:.2f} ms/image  encode {(t2 - t1) / 3 / B * 1e3:.2f} ms/image")
