"""Kernel-level profile target: a few graph replays of the UNet forward at a small row count (rocprofv3 --kernel-trace --stats -- python3 tools/small_rows_profile.py 16)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import gswm_amd
from gswm_amd import unet as U
from gswm_amd.graph import GraphedEpsModel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
m = U.synthetic_init_(U.UNet2DCondition(), 0).cuda().half().eval()
gm = GraphedEpsModel(m, mode="always")
x = torch.randn(B, 4, 64, 64, device="cuda", dtype=torch.float16); t = torch.full((), 500, device="cuda"); c = torch.randn(B, 77, 1024, device="cuda", dtype=torch.float16)
with torch.no_grad():
    for _ in range(12):
        gm(x, t, c)
        torch.cuda.synchronize()          # tools/graph_timeline.py splits replays at idle gaps: make one between every two replays
        time.sleep(0.0005)
torch.cuda.synchronize()
