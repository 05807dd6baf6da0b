"""Self-attention launch time by batch rows and level under the current GSW_ATTN_QB setting (unset: the dispatch rule; 1: 128-query workgroups; 3: 256-query
workgroups wherever the shape allows) -- run once per setting; calibrates the query-tile rule of gsw_attention (csrc/gswm_attn.hip).  Graph-captured, 8 launches per replay."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import gswm_amd  # noqa: E402,F401
from gswm_amd import pf  # noqa: E402

tag = os.environ.get("GSW_ATTN_QB", "rule")
for rows in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64):
    for S, H in ((4096, 5), (1024, 10), (256, 20)):
        q, k, v = (torch.randn(rows, S, H * 64, device="cuda").half() for _ in range(3))
        vt = v.transpose(1, 2).contiguous()
        out = torch.empty_like(q)
        pf.attention_hd64(q, k, vt, H, out=out)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(8):
                pf.attention_hd64(q, k, vt, H, out=out)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        print(f"{tag} rows={rows:3d} S={S:5d} H={H:2d} wg256={rows * H * S // 256:6d}: {e0.elapsed_time(e1) * 1e3 / 40:8.1f} us", flush=True)
