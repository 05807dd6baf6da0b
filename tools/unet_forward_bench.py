import sys, time, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','.'))
import gswm_amd
from gswm_amd import unet as U
dev='cuda'; dt=torch.float16
m = U.synthetic_init_(U.UNet2DCondition(), 0).to(dev, dt).eval()
B=int(sys.argv[1]) if len(sys.argv)>1 else 128
x=torch.randn(B,4,64,64,device=dev,dtype=dt); t=torch.full((),500,device=dev); c=torch.randn(B,77,1024,device=dev,dtype=dt)
with torch.no_grad():
    for _ in range(3): y=m(x,t,c)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(4): y=m(x,t,c)
    torch.cuda.synchronize(); d=(time.perf_counter()-t0)/4
print(f"B={B}: {d*1e3:.1f} ms {B*0.804/d:.0f} TFLOP/s", flush=True)

if len(sys.argv) > 2 and sys.argv[2] == "convs":      # per-shape table of the convolution launches (HIP events around each)
    from gswm_amd import pf
    tm = pf.ConvTimer(by_shape=True)
    pf.CONV_TIMER = tm
    with torch.no_grad():
        for _ in range(3): m(x,t,c)
    torch.cuda.synchronize()
    pf.CONV_TIMER = None
    tot = sum(v["ms"] for v in tm.summary().values())
    print(f"matmul-engine / convolution launches: {tot/3:.1f} ms per forward")
    for k, v in sorted(tm.summary().items(), key=lambda kv: -kv[1]["ms"]):
        if len(k) == 7:          # convolution: (kernel, B, H, W, K, N, stride)
            name, b, h, w, kk, n, st = k
            print(f"{name:28s} {h:3d}x{w:<3d} K={kk:6d} N={n:5d} s{st} calls/fwd={v['calls']//3:3d} avg={v['avg_us']:8.1f} us  {v['tflops']:7.1f} TFLOP/s  {v['ms']/tot*100:5.1f} %")
        else:                    # linear: (kernel, M, K, N, mode)
            name, m_, kk, n, mode = k
            print(f"{name + ' ' + mode:28s} M={m_:7d} K={kk:6d} N={n:5d}    calls/fwd={v['calls']//3:3d} avg={v['avg_us']:8.1f} us  {v['tflops']:7.1f} TFLOP/s  {v['ms']/tot*100:5.1f} %")
