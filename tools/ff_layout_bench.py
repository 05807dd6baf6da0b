"""FF projection / output GEMMs: normal layout vs transposed-output (channel-major) layout through hipBLASLt, B=128."""
import os, sys, torch, torch.nn.functional as F
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e-3
B = 128
for S, C in ((4096, 320), (1024, 640), (256, 1280)):
    I = 4 * C
    x = torch.randn(B, S, C, device="cuda", dtype=torch.float16)
    w1 = torch.randn(2 * I, C, device="cuda", dtype=torch.float16) * 0.02; b1 = torch.randn(2 * I, device="cuda", dtype=torch.float16)
    w2 = torch.randn(C, I, device="cuda", dtype=torch.float16) * 0.02; b2 = torch.randn(C, device="cuda", dtype=torch.float16)
    h = torch.randn(B, S, I, device="cuda", dtype=torch.float16); hT = h.transpose(1, 2).contiguous()
    f1 = 2.0 * B * S * C * 2 * I; f2 = 2.0 * B * S * I * C
    a = t(lambda: F.linear(x, w1, b1))
    bT = t(lambda: torch.bmm(w1.unsqueeze(0).expand(B, -1, -1), x.transpose(1, 2)))
    c = t(lambda: F.linear(h, w2, b2))
    dT = t(lambda: torch.bmm(hT.transpose(1, 2), w2.t().unsqueeze(0).expand(B, -1, -1)))
    print(f"S={S} C={C}: proj normal {a*1e6:6.0f} us ({f1/a/1e12:4.0f} TF) | proj transposed-out {bT*1e6:6.0f} us ({f1/bT/1e12:4.0f} TF) || out normal {c*1e6:6.0f} us ({f2/c/1e12:4.0f} TF) | out from transposed A {dT*1e6:6.0f} us ({f2/dT/1e12:4.0f} TF)", flush=True)
