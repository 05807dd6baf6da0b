"""Forward time of the eps model and, with `convs`, the per-shape table of its matmul-engine / convolution / attention launches (HIP events around each).
usage: python tools/unet_forward_bench.py [B] [convs] [sd15] [hw=96]
The table's last column is the EFFECTIVE HBM rate of a dense launch: algorithmic bytes (x, w, y, + the residual it reads) / time -- what the HBM-bound
level-0 shapes are judged by (the achievable copy rate here is ~6.3 TB/s)."""
import sys, time, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import gswm_amd
from gswm_amd import unet as U
dev = 'cuda'; dt = torch.float16
sd15 = "sd15" in sys.argv
hw = next((int(a[3:]) for a in sys.argv if a.startswith("hw=")), 96 if sd15 else 64)
cfg = dict(cross_attention_dim=768, num_heads=(8, 8, 8, 8), head_dim=None) if sd15 else {}
m = U.synthetic_init_(U.UNet2DCondition(**cfg), 0).to(dev, dt).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
x = torch.randn(B, 4, hw, hw, device=dev, dtype=dt); t = torch.full((), 500, device=dev); c = torch.randn(B, 77, cfg.get("cross_attention_dim", 1024), device=dev, dtype=dt)
fl_row = U.count_flops_per_image(m, hw, hw) / 1e12
with torch.no_grad():
    for _ in range(3): y = m(x, t, c)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4): y = m(x, t, c)
    torch.cuda.synchronize(); d = (time.perf_counter() - t0) / 4
print(f"{'SD 1.5' if sd15 else 'SD 2.1'} shape, {hw} x {hw} latent, B={B}: {d*1e3:.1f} ms {B*fl_row/d:.0f} TFLOP/s ({fl_row:.4f} TFLOP per row)", flush=True)

if "convs" in sys.argv:      # per-shape table (HIP events around each launch)
    from gswm_amd import pf
    tm = pf.ConvTimer(by_shape=True)
    pf.CONV_TIMER = tm
    with torch.no_grad():
        for _ in range(3): m(x, t, c)
    torch.cuda.synchronize()
    pf.CONV_TIMER = None
    tot = sum(v["ms"] for v in tm.summary().values())
    print(f"matmul-engine / convolution / attention launches: {tot/3:.1f} ms per forward")
    for k, v in sorted(tm.summary().items(), key=lambda kv: -kv[1]["ms"]):
        if k[0] == "gsw_attn_fwd_kernel":          # (kernel, B, Sq, Sk, heads, head_dim)
            name, b, sq, sk, hh, hd = k
            print(f"{name:28s} Sq={sq:5d} Sk={sk:5d} H={hh:2d} d={hd:3d}  calls/fwd={v['calls']//3:3d} avg={v['avg_us']:8.1f} us  {v['tflops']:7.1f} TFLOP/s  {v['ms']/tot*100:5.1f} %")
        elif len(k) == 7:          # convolution: (kernel, B, H, W, K, N, stride)
            name, b, h, w, kk, n, st = k
            print(f"{name:28s} {h:3d}x{w:<3d} K={kk:6d} N={n:5d} s{st} calls/fwd={v['calls']//3:3d} avg={v['avg_us']:8.1f} us  {v['tflops']:7.1f} TFLOP/s  {v['ms']/tot*100:5.1f} %")
        else:                    # linear: (kernel, M, K, N, mode)
            name, m_, kk, n, mode = k
            ncols = n // 2 if mode.startswith("geglu") else n
            nbytes = 2.0 * (m_ * kk + n * kk + m_ * ncols * (2 if ("+res" in mode or mode == "tok2pf") else 1))
            print(f"{name + ' ' + mode:28s} M={m_:7d} K={kk:6d} N={n:5d}    calls/fwd={v['calls']//3:3d} avg={v['avg_us']:8.1f} us  {v['tflops']:7.1f} TFLOP/s  {v['ms']/tot*100:5.1f} %  {nbytes / v['avg_us'] / 1e6:5.2f} TB/s effective")
