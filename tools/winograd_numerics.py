"""Review item 5 (round 5): cost Winograd F(2x2, 3x3) on ONE deep convolution before anything else in the convolution bucket.  This script measures the NUMERICS half on
the CPU (no GPU needed): a 3 x 3, stride-1 convolution 1280 -> 1280 on 16 x 16 images with N(0, 1) activations behind a GroupNorm-like scale and N(0, 1 / sqrt(9 C)) weights,
(a) the way the engine computes it -- fp16 operands, exact products, fp32 accumulation, ONE rounding of the output -- and (b) as F(2x2, 3x3): input tiles transformed in fp32
and rounded to fp16 (V = B^T d B), weights transformed in fp32 and rounded to fp16 (U = G g G^T), 16 element-wise GEMMs with fp32 accumulation, output transform in fp32,
one rounding.  Both against an fp64 convolution of the SAME fp16-rounded operands.  The byte / FLOP half of the costing is arithmetic and sits in the output text.
usage: python tools/winograd_numerics.py [images=2]"""
import sys, torch
import torch.nn.functional as F
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
C = N = 1280; H = W = 16
x = torch.randn(B, C, H, W).half()
w = (torch.randn(N, C, 3, 3) * (9 * C) ** -0.5).half()
ref = F.conv2d(x.double(), w.double(), padding=1)
direct = F.conv2d(x.float(), w.float(), padding=1).half()          # fp16 products are exact in fp32; fp32 accumulation; one rounding
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
U = torch.einsum("ij,ncjk,lk->ncil", G, w.double(), G).float().half()                      # [N, C, 4, 4], rounded once
xp = F.pad(x.float(), (1, 1, 1, 1))
tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                                  # [B, C, 8, 8, 4, 4]
V = torch.einsum("ij,bcyxjk,lk->bcyxil", Bt.float(), tiles, Bt.float()).half()             # transformed in fp32, rounded once
M = torch.einsum("ncil,bcyxil->bnyxil", U.float(), V.float())                              # 16 GEMMs, fp32 accumulation
Y = torch.einsum("ij,bnyxjk,lk->bnyxil", At.float(), M, At.float())                        # [B, N, 8, 8, 2, 2]
wino = Y.permute(0, 1, 2, 4, 3, 5).reshape(B, N, H, W).half()
scale = ref.abs().max().item()
for name, y in (("direct (engine arithmetic)", direct), ("Winograd F(2x2, 3x3), fp16 U and V", wino)):
    e = (y.double() - ref).abs()
    print(f"{name:38s} max |err| = {e.max().item():.3e} ({e.max().item() / scale:.2e} of max |y|)   rms err = {e.pow(2).mean().sqrt().item():.3e}")
ed, ew = (direct.double() - ref).abs(), (wino.double() - ref).abs()
print(f"ratio Winograd / direct: max {ew.max().item() / ed.max().item():.2f} x, rms {ew.pow(2).mean().sqrt().item() / ed.pow(2).mean().sqrt().item():.2f} x   (adoption bar of the review: <= 2 x)")
