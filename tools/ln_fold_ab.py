"""A/B: UNet forward with the LayerNorms folded into their consuming GEMMs (pf.FOLD_LN) against the LayerNorm kernel + plain GEMM.  GSW_FOLD_LN=0/1"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import gswm_amd
from gswm_amd import unet as U, pf
pf.FOLD_LN = os.environ.get("GSW_FOLD_LN", "1") == "1"
m = U.synthetic_init_(U.UNet2DCondition(), 0).cuda().half().eval()
for B in (128, 64):
    x = torch.randn(B, 4, 64, 64, device="cuda", dtype=torch.float16); t = torch.full((), 500, device="cuda"); c = torch.randn(B, 77, 1024, device="cuda", dtype=torch.float16)
    with torch.no_grad():
        for _ in range(3): m(x, t, c)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): m(x, t, c)
        torch.cuda.synchronize(); d = (time.perf_counter() - t0) / 5
    print("fold_ln", pf.FOLD_LN, "B", B, round(d * 1e3, 2), "ms", round(B * 0.804 / d), "TF/s", flush=True)
