"""Which Python lines of a small-batch forward launch torch glue kernels (copies, cats, pads)?  Runs ONE eager forward at `rows` rows under torch.profiler with stacks and
prints, per (operator, call site in this package), the number of calls -- every one of them is a node of the captured graph, ~4.7 us on the dependent chain (section 4.7).
usage: python tools/copy_sites.py [rows]"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gswm_amd  # noqa: F401,E402
from gswm_amd import unet as U  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
m = U.synthetic_init_(U.UNet2DCondition(), 0).cuda().half().eval()
x = torch.randn(B, 4, 64, 64, device="cuda", dtype=torch.float16)
t = torch.full((), 500, device="cuda")
c = torch.randn(B, 77, 1024, device="cuda", dtype=torch.float16)
with torch.no_grad():
    for _ in range(2):
        m(x, t, c)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        m(x, t, c)
        torch.cuda.synchronize()
sites = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::cat", "aten::contiguous", "aten::clone", "aten::pad", "aten::constant_pad_nd", "aten::zeros", "aten::zero_", "aten::fill_", "aten::to",
                   "aten::_to_copy", "aten::stack", "aten::index", "aten::add", "aten::mul", "aten::silu", "aten::cos", "aten::sin", "aten::exp", "aten::arange"):
        site = next((s for s in ev.stack if "a-watermark-for-diffusion-models_amd" in s or "gswm_amd" in s), None)
        if site is None:
            continue
        sites[(ev.name, site.split("a-watermark-for-diffusion-models_amd/")[-1])] += 1
print(f"rows = {B}: torch operators of one forward by call site (operator, site, calls)")
for (name, site), n in sorted(sites.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d}  {name:24s} {site}")
kern = collections.Counter(ev.name for ev in prof.events() if ev.device_type is not None and str(ev.device_type).endswith("CUDA"))
print("device kernels of the forward:", sum(kern.values()))
for k, n in kern.most_common(40):
    print(f"{n:4d}  {k[:120]}")
