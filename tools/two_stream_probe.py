"""Probe: the CFG pair (uncond rows | text rows) as TWO half-batch forwards on two HIP streams against ONE 2B-row forward.
Memory-bound kernels (GroupNorm / LayerNorm) of one stream can run beside the other stream's matmul-engine kernels."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gswm_amd
from gswm_amd import unet as U, pf

_ws = {}
def ws(device, B, groups):       # per-stream GroupNorm workspace
    k = (str(device), B * 64 * groups * 2, torch.cuda.current_stream().cuda_stream)
    if k not in _ws:
        _ws[k] = torch.empty(k[1], dtype=torch.float32, device=device)
    return _ws[k]
pf._gn_workspace = ws

dev, dt = "cuda", torch.float16
m = U.synthetic_init_(U.UNet2DCondition(), 0).to(dev, dt).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.randn(B, 4, 64, 64, device=dev, dtype=dt); t = torch.full((), 500, device=dev)
cu = torch.randn(B, 77, 1024, device=dev, dtype=dt); ct = torch.randn(B, 77, 1024, device=dev, dtype=dt)
x2, c2 = torch.cat([x, x]), torch.cat([cu, ct])
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def one():
    return m(x2, t, c2)

def two():
    main = torch.cuda.current_stream()
    s1.wait_stream(main); s2.wait_stream(main)
    with torch.cuda.stream(s1):
        a = m(x, t, cu)
    with torch.cuda.stream(s2):
        b = m(x, t, ct)
    main.wait_stream(s1); main.wait_stream(s2)
    return a, b

with torch.no_grad():
    for f, name in ((one, "one 2B-row forward"), (two, "two B-row forwards on two streams"), (one, "one 2B-row forward (again)")):
        for _ in range(3): r = f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): r = f()
        torch.cuda.synchronize(); d = (time.perf_counter() - t0) / 5
        print(f"{name}: {d*1e3:.1f} ms", flush=True)
    y = one(); a, b = two(); torch.cuda.synchronize()
    print("max |diff| uncond", (y[:B].float() - a.float()).abs().max().item(), "text", (y[B:].float() - b.float()).abs().max().item())
