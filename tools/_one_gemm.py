import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import gswm_amd
from gswm_amd import pf as P
M, K, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (32768, 5120, 1280)
x = torch.randn(M, K, device='cuda', dtype=torch.float16); w = (torch.randn(N, K, device='cuda', dtype=torch.float16) * K ** -0.5)
for _ in range(4): y = P.gemm(x, w, None)
torch.cuda.synchronize()
