"""Microbenchmark of the image-side kernels (csrc/gswm_image.hip): achieved algorithmic HBM GB/s (u8 in + out bytes / time).
usage: python tools/image_bench.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gswm_amd  # noqa: E402
from gswm_amd import imaging  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (B, 512, 512, 3), dtype=torch.uint8, generator=g).cuda()
    big = torch.randint(0, 256, (max(1, B // 4), 1024, 1024, 3), dtype=torch.uint8, generator=g).cuda()
    npx = B * 512 * 512
    rows = []
    t = timeit(lambda: imaging.jpeg_roundtrip(img, 10)); rows.append(("jpeg_roundtrip q10 512^2 u8->u8", 6 * npx, t))
    t = timeit(lambda: imaging.jpeg_roundtrip(img, 10, out="f16")); rows.append(("jpeg_roundtrip q10 512^2 u8->f16", 9 * npx, t))
    t = timeit(lambda: imaging.resize_lanczos(img, (256, 256))); rows.append(("resize 512^2->256^2 u8", 3 * npx + 3 * npx // 4, t))
    t = timeit(lambda: imaging.resize_lanczos(big, (512, 512), out="f16"));
    rows.append(("resize 1024^2->512^2 f16 (extract.py default --width 1024 on 512 model)", big.shape[0] * (3 * 1024 * 1024 + 6 * 512 * 512), t))
    t = timeit(lambda: imaging.to_tensor(img, out="f16")); rows.append(("to_tensor 512^2 u8->f16", 9 * npx, t))
    x = torch.rand(B, 3, 512, 512, generator=g).cuda().half()
    t = timeit(lambda: imaging.tensor_to_image(x)); rows.append(("tensor_to_image 512^2 f16->u8", 9 * npx, t))
    t = timeit(lambda: imaging.pointwise(img, "contrast", 1.7)); rows.append(("contrast 512^2 (2 passes)", 9 * npx, t))
    t = timeit(lambda: imaging.pointwise(img, "noise", 0.1)); rows.append(("noise 512^2", 6 * npx, t))
    for name, byts, t in rows:
        print(f"{name:75s} B={B:4d}  {t * 1e6:9.1f} us  {byts / t / 1e9:8.1f} GB/s  {B / t:10.0f} images/s")


if __name__ == "__main__":
    main()
