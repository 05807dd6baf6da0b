"""Per-shape table of the VAE's engine launches (decode + encode of B images at 512 x 512).  usage: python tools/vae_per_shape.py [B=16]"""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import gswm_amd
from gswm_amd import vae as V, pf
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
v = V.synthetic_init_(V.AutoencoderKL(), 1).cuda().half().eval()
z = torch.randn(B, 4, 64, 64, device="cuda", dtype=torch.float16)
x = torch.rand(B, 3, 512, 512, device="cuda", dtype=torch.float16) * 2 - 1
with torch.no_grad():
    for _ in range(2): v.decode(z); v.encode_mean(x)
    for name, fn in (("decode", lambda: v.decode(z)), ("encode", lambda: v.encode_mean(x))):
        tm = pf.ConvTimer(by_shape=True); pf.CONV_TIMER = tm
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        pf.CONV_TIMER = None
        sm = tm.summary(); tot = sum(d["ms"] for d in sm.values())
        print(f"{name}: {e0.elapsed_time(e1):.1f} ms for {B} images with events; engine launches {tot:.1f} ms")
        for k, d in sorted(sm.items(), key=lambda kv: -kv[1]["ms"])[:12]:
            print(f"   {str(k):70s} calls={d['calls']:3d} avg={d['avg_us']:8.1f} us {d['tflops']:7.1f} TFLOP/s  {d['bytes_per_launch'] / d['avg_us'] / 1e6:5.2f} TB/s alg  {d['ms'] / tot * 100:5.1f} %")
