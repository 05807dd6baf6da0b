"""Attention kernel against the fp32 softmax(QK^T / 8) V of the same rounded inputs: max and rms error (A/B of kernel variants through GSWM_LIB)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gswm_amd  # noqa: F401
from gswm_amd import pf
for dt in (torch.float16, torch.bfloat16):
    for S, H, B, sc in ((1024, 10, 8, 1.0), (4096, 5, 2, 1.0), (4096, 5, 2, 3.0), (256, 20, 16, 2.0)):
        g = torch.Generator().manual_seed(S)
        q, k, v = (torch.randn(B, S, H * 64, generator=g).to(dt).cuda() for _ in range(3))
        q = q * sc
        got = pf.attention_hd64(q, k, v.transpose(1, 2).contiguous(), H).float()
        qf, kf, vf = (a.float().view(B, S, H, 64).transpose(1, 2) for a in (q, k, v))
        ref = (torch.softmax(qf @ kf.transpose(-1, -2) / 8.0, dim=-1) @ vf).transpose(1, 2).reshape(B, S, H * 64)
        d = got - ref
        print(f"{str(dt)[6:]:9s} S={S:5d} H={H:2d} B={B:2d} q x{sc}: max err {d.abs().max().item():.3e}  rms err {d.pow(2).mean().sqrt().item():.3e}  (ref rms {ref.pow(2).mean().sqrt().item():.3e})")
