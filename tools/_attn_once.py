import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import gswm_amd
from gswm_amd import pf
B, S, H = 128, 4096, 5
g = torch.Generator().manual_seed(0)
q = torch.randn(B, S, H * 64, generator=g).cuda().half(); k = torch.randn(B, S, H * 64, generator=g).cuda().half()
vt = torch.randn(B, H * 64, S, generator=g).cuda().half()
for _ in range(3): pf.attention_hd64(q, k, vt, H)
torch.cuda.synchronize()
