import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import gswm_amd
from gswm_amd import imaging
g = torch.Generator().manual_seed(0)
img = torch.randint(0, 256, (64, 512, 512, 3), dtype=torch.uint8, generator=g).cuda()
big = torch.randint(0, 256, (16, 1024, 1024, 3), dtype=torch.uint8, generator=g).cuda()
for _ in range(5):
    imaging.jpeg_roundtrip(img, 10, out="f16")
    imaging.resize_lanczos(big, (512, 512), out="f16")
    imaging.to_tensor(img, out="f16")
    imaging.pointwise(img, "contrast", 1.7)
torch.cuda.synchronize()
