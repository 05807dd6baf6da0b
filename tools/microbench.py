#!/usr/bin/env python3
"""Per-kernel timing table (HIP events over many back-to-back launches) for the codec kernels on one GPU.
Usage: python tools/microbench.py [--iters 50]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gswm_amd
from gswm_amd import codec

ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=50); ap.add_argument("--only", default="")
a = ap.parse_args()
key = bytes.fromhex("5822ff9cce6772f714192f43863f6bad1bf54b78326973897e6b66c3186b77a7"); nonce = bytes.fromhex("05072fd1c2265f6f2e2a4080a2bfbdd8")
k = codec.pad_message("lthero", 32)

def timeit(fn, iters=a.iters):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us

def row(name, us, nbytes):
    print(f"{name:58s} {us:10.1f} us {nbytes/us/1e3:9.0f} GB/s", flush=True)

N = 16384
for B in (64, 1024, 4096, 16384, 65536):
    if a.only and a.only not in "ddim": break
    x = torch.randn(B, N, device="cuda", dtype=torch.float16); e = torch.randn_like(x); o = torch.empty_like(x)
    row(f"ddim_step f16 B={B}", timeit(lambda: codec.ddim_step(x, e, 1.01, -0.02, out=o)), 3 * 2 * B * N)
    row(f"torch copy_ f16 B={B}", timeit(lambda: o.copy_(x)), 2 * 2 * B * N)
for B in (4096, 16384, 65536):
    for dt in (torch.float16, torch.float32):
        if a.only and a.only not in "extract": break
        z = codec.embed_batch(key, nonce, k, B, (4, 64, 64), seed=1, dtype=dt, fast=True)
        row(f"extract {dt} B={B} M=256", timeit(lambda: codec.extract_batch(z, key, nonce, 256)), z.element_size() * B * N)
        row(f"torch sum {dt} B={B} (read-only stream)", timeit(lambda: z.sum()), z.element_size() * B * N)
for B in (4096, 16384, 65536):
    for dt in (torch.float32, torch.float16):
        if a.only and a.only not in "embed": break
        o = torch.empty(B, 4, 64, 64, device="cuda", dtype=dt)
        row(f"embed fast {dt} B={B}", timeit(lambda: codec.embed_batch(key, nonce, k, B, (4, 64, 64), seed=1, fast=True, out=o)), o.element_size() * B * N)
        row(f"torch fill_ {dt} B={B} (write-only stream)", timeit(lambda: o.fill_(1.0)), o.element_size() * B * N)
B = 4096
o = torch.empty(B, 4, 64, 64, device="cuda", dtype=torch.float32)
row(f"embed exact f32 B={B}", timeit(lambda: codec.embed_batch(key, nonce, k, B, (4, 64, 64), seed=1, fast=False, out=o), 10), 4 * B * N)
