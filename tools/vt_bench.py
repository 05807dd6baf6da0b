"""V^T projection as bmm(W, x^T) vs plain linear (+ what SDPA needs) on the UNet's self-attention shapes."""
import os, sys, torch, torch.nn.functional as F
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e-3
B = 128
for S, C in ((4096, 320), (1024, 640), (256, 1280)):
    x = torch.randn(B, S, C, device="cuda", dtype=torch.float16); w = torch.randn(C, C, device="cuda", dtype=torch.float16) * 0.02
    fl = 2.0 * B * S * C * C
    t_lin = t(lambda: F.linear(x, w))
    t_bmm = t(lambda: torch.bmm(w.unsqueeze(0).expand(B, -1, -1), x.transpose(1, 2)))
    t_mm = t(lambda: torch.matmul(w, x.transpose(1, 2)))
    t_lt = t(lambda: F.linear(x, w).transpose(1, 2).contiguous())
    print(f"S={S} C={C}: linear {t_lin*1e6:6.0f} us ({fl/t_lin/1e12:4.0f} TF) | bmm(W,x^T) {t_bmm*1e6:6.0f} us ({fl/t_bmm/1e12:4.0f} TF) | matmul(W,x^T) {t_mm*1e6:6.0f} us | linear+transpose copy {t_lt*1e6:6.0f} us", flush=True)
