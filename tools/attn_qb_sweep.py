"""128- vs 256-query workgroups of the self-attention kernel over small batches (GSW_ATTN_QB=1 / 3 force a form).  python tools/attn_qb_sweep.py"""
import os, sys, subprocess
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import gswm_amd  # noqa: F401
    from gswm_amd import pf
    out = []
    for S, H in ((4096, 5), (1024, 10)):
        for B in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 64):
            q, k, v = (torch.randn(B, S, H * 64, device="cuda").half() for _ in range(3))
            vt = v.transpose(1, 2).contiguous()
            for _ in range(3): pf.attention_hd64(q, k, vt, H)
            torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20): pf.attention_hd64(q, k, vt, H)
            e.record(); torch.cuda.synchronize()
            out.append(f"{S} {H} {B} {s.elapsed_time(e) / 20 * 1e3:.1f}")
    print("\n".join(out))
else:
    res = {}
    for qb in ("1", "3"):
        r = subprocess.run([sys.executable, __file__, "run"], env=dict(os.environ, GSW_ATTN_QB=qb), capture_output=True, text=True)
        for line in r.stdout.splitlines():
            p = line.split()
            if len(p) == 4 and p[0].isdigit(): res[(int(p[0]), int(p[1]), int(p[2]), qb)] = float(p[3])
    print("S H B | 128-query WGs (us) | 256-query WGs (us) | 256-query grid")
    for (S, H, B, qb) in sorted(k for k in res if k[3] == "1"):
        print(f"{S:5d} {H:3d} {B:3d} | {res[(S, H, B, '1')]:9.1f} | {res.get((S, H, B, '3'), float('nan')):9.1f} | {S // 256 * B * H}")
