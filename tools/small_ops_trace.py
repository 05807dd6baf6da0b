"""Which lines of the package issue the small torch kernels (copies, dtype conversions, concatenations) of one eager UNet forward?
python tools/small_ops_trace.py [rows=1]"""
import os, sys, collections, traceback
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gswm_amd  # noqa: E402,F401
from gswm_amd import unet as U  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
m = U.synthetic_init_(U.UNet2DCondition(), 0).cuda().half().eval()
x = torch.randn(B, 4, 64, 64, device="cuda", dtype=torch.float16); t = torch.full((), 500, device="cuda"); c = torch.randn(B, 77, 1024, device="cuda", dtype=torch.float16)
counts = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ("copy", "_to_copy", "cat", "clone", "fill", "zero", "add", "mul", "cos", "sin", "exp", "arange", "pad", "index", "where", "sub", "div")):
            fr = [f for f in traceback.extract_stack() if "diffusion-models_amd" in f.filename or "gswm_amd" in f.filename]
            where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-3:][::-1])
            info = ""
            for a in args[:2]:
                if isinstance(a, torch.Tensor):
                    info += f" {str(a.dtype).replace('torch.', '')}{tuple(a.shape)}"
            counts[(name, where, info[:60])] += 1
        return func(*args, **(kwargs or {}))


with torch.no_grad():
    m(x, t, c)
    torch.cuda.synchronize()
    with Spy():
        m(x, t, c)
torch.cuda.synchronize()
for (name, where, info), n in sorted(counts.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{n:4d}  {name:28s} {info:50s} {where}")
