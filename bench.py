#!/usr/bin/env python3
"""bench.py -- watermarked images/sec (embed + extract, 512x512 SD2.1 lattice 4x64x64) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 50 --warmup 5                       # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                           # N ranks, one per GPU, RCCL

Tiers
  codec (default): one step = gsw_embed (fp32 Z_s_T for B images, in-kernel Philox u) + gsw_extract (fp16 latents of B
         images -> 256-bit messages).  Inputs resident in HBM.  Roofline: HBM (algorithmic bytes, SURVEY.md 8d:
         embed 4N B written, extract 2N B read per image).
  e2e:   embed -> 50-step DDIM sampling (CFG) -> 50-step DDIM inversion -> extract with the SD2.1-shaped UNet
         (synthetic weights).  Roofline: MFMA.   

Prints ONE JSON line on rank 0.  The CPU baseline leg (rank 0, N == 1 only) times the oracle's reference-shaped scalar
port on a bounded sample; it is a reported baseline, never the thing measured.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

README_KEY = "5822ff9cce6772f714192f43863f6bad1bf54b78326973897e6b66c3186b77a7"
README_NONCE = "05072fd1c2265f6f2e2a4080a2bfbdd8"
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16/fp16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 2 for e2e / both, 50 for --tier codec)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warmup steps (default: 1 for e2e / both, 5 for --tier codec)")
    ap.add_argument("--tier", choices=["both", "codec", "e2e"], default="both")
    ap.add_argument("--batch", type=int, default=64, help="e2e: images per GPU per step (north-star batch 64)")
    ap.add_argument("--codec-batch", type=int, default=16384, help="codec tier: images per GPU per step (HBM-bound regime)")
    ap.add_argument("--codec-steps", type=int, default=50, help="codec tier steps when it rides along with e2e")
    ap.add_argument("--codec-streams", type=int, choices=[1, 2], default=1,
                    help="codec tier: 1 (default) = embed, then extract on one stream; 2 = the embed (a pure HBM write stream) and the extract (a pure read stream) of a "
                         "step on two HIP streams sharing the HBM interface -- measured in round 5 and NOT faster (4.93e7 vs 5.00e7 images/s, "
                         "profiles/r05e_codec_one_vs_two_streams.txt), kept as a switch.  The per-kernel roofline durations always come from single-stream launches")
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--message-length", type=int, default=256)
    ap.add_argument("--exact", action="store_true", help="Cephes fp64 inverse CDF instead of the fp32 fast path")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=12, help="images of the 1-core CPU-baseline sample (0.7-2.2 s each); the all-cores leg adds a 12 s window + pool start-up")
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--image-stages", choices=["none", "vae", "vae+jpeg"], default="vae",
                    help="e2e: also run (and time) VAE decode -> uint8 image -> [JPEG QF] -> ToTensor/normalise -> VAE encode per image "
                         "(BASELINE config 4's data path; synthetic VAE weights, so the inversion still consumes the latent)")
    ap.add_argument("--jpeg-qf", type=int, default=10)
    ap.add_argument("--vae-chunk", type=int, default=8, help="images per VAE call")
    ap.add_argument("--workload", choices=["roundtrip", "txt2img"], default="roundtrip",
                    help="e2e tier: roundtrip = embed -> sampling -> image stages -> inversion -> vote (the headline); txt2img = BASELINE configs[1] as named: "
                         "embed -> 50-step CFG sampling -> VAE decode -> uint8 image, no extraction (use with --batch 8)")
    ap.add_argument("--unet", choices=["sd21", "sd15"], default="sd21", help="sd15 + --height 768 --width 768 = BASELINE config 5's shape")
    ap.add_argument("--preflight", action="store_true", help="every rank: device check, RCCL init, one broadcast + all_gather_into_tensor + all_reduce under a hard "
                                                             "time limit (GSW_PREFLIGHT_TIMEOUT_S, default 120); rank 0 prints ONE JSON line; a failing stage is named and exits 3")
    ap.add_argument("--launcher-selftest", action="store_true", help="only rendezvous, all-gather the ranks and print them (GSW_BENCH_BACKEND=gloo on a CPU host)")
    a = ap.parse_args()
    if a.steps is None:
        a.steps = 50 if a.tier == "codec" else 2
    if a.warmup is None:
        a.warmup = 5 if a.tier == "codec" else 1
    return a


def _cpu_worker(job):
    """One worker of a CPU-baseline leg: images of the reference-shaped scalar port (embed + recover), each worker its own NumPy seed; either a fixed
    number of images or as many as fit before `deadline` (time.time(); at least one)."""
    n_images, message_length, seed, deadline = job
    import types
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gs_oracle as O
    opt = types.SimpleNamespace(key_hex=README_KEY, nonce_hex=README_NONCE)
    a = types.SimpleNamespace(key=bytes.fromhex(README_KEY), nonce=bytes.fromhex(README_NONCE), l=1, message_length=message_length)
    np.random.seed(seed)
    ok, done = True, 0
    while done < n_images if deadline is None else (done == 0 or time.time() < deadline):
        z = O.gs_watermark_init_noise_scalar(opt, "lthero")
        bits = O.recover_exactracted_message_scalar(z.astype(np.float16), a)
        ok &= O.calculate_bit_accuracy((b"lthero" + b"\0" * 26).hex(), bits)[1] == 1.0
        done += 1
    return done, bool(ok), time.time()


def _usable_cpus() -> int:
    """CPUs this process may actually use: the scheduler affinity, cut down to the cgroup's CPU quota when there is one (a container on a 256-thread
    host is often entitled to a fraction of it: 256 busy processes then only thrash)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                       # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // per)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(n_images: int, message_length: int, all_cores: bool = True, window_s: float = 12.0):
    """Reference-shaped scalar port (oracle) on this host: embed + recover per image on ONE core and -- SURVEY.md 8d(ii) -- on ALL usable host cores
    (multiprocessing, one process per core, os.cpu_count() and the usable count stated), plus the vectorised NumPy restatement as a best-CPU line.
    Bounded: the all-cores leg is a fixed WINDOW of `window_s` seconds (every worker counts the images it finishes in it), so the default bench stays
    within minutes on any host.  Called BEFORE anything touches the GPU: the pool's children are fresh interpreters ("spawn") of a process without HIP state."""
    import types
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gs_oracle as O
    opt = types.SimpleNamespace(key_hex=README_KEY, nonce_hex=README_NONCE)
    a = types.SimpleNamespace(key=bytes.fromhex(README_KEY), nonce=bytes.fromhex(README_NONCE), l=1, message_length=message_length)
    t0 = time.perf_counter()
    _, ok, _ = _cpu_worker((n_images, message_length, 0, None))
    dt = time.perf_counter() - t0
    # best-CPU line: the vectorised numpy restatement
    t1 = time.perf_counter()
    nv = 64
    for _ in range(nv):
        z = O.gs_watermark_init_noise(opt, "lthero")
        O.recover_exactracted_message(z.astype(np.float16), a)
    dv = time.perf_counter() - t1
    one = {"value": n_images / dt, "unit": "images/s", "cores": 1,
           "sample": f"{n_images} images, 4x64x64, embed+recover, scalar scipy.stats.norm.ppf/cdf per element, {dt:.1f} s"}
    out = {"value": one["value"], "unit": "images/s", "cores": 1, "kind": "port",
           "sample": f"{n_images} images, 4x64x64, embed+recover, scalar scipy.stats.norm.ppf/cdf per element (reference-shaped port of "
                     f"gs_insert.py:49-66 + extract.py:72-101), {dt:.1f} s; lossless={bool(ok)}",
           "one_core": one, "vectorised_numpy_images_per_s": nv / dv, "host_cpus": os.cpu_count()}
    if all_cores:
        import multiprocessing as mp
        ncpu = _usable_cpus()
        t2 = time.perf_counter()
        with mp.get_context("spawn").Pool(ncpu) as pool:
            pool.map(_cpu_worker, [(1, message_length, 1000 + i, None) for i in range(ncpu)], chunksize=1)       # start-up + imports + first call, untimed
            t3 = time.perf_counter()
            w0 = time.time()
            res = pool.map(_cpu_worker, [(0, message_length, 2000 + i, w0 + window_s) for i in range(ncpu)], chunksize=1)
        n_all = sum(r[0] for r in res)
        ok_all = all(r[1] for r in res)
        da = max(r[2] for r in res) - w0                                  # the window runs until the last worker has finished the image it was on
        allc = {"value": n_all / da, "unit": "images/s", "cores": ncpu,
                "sample": f"{ncpu} processes (multiprocessing, one per usable core; os.cpu_count() = {os.cpu_count()}), {n_all} images in a {da:.1f} s window "
                          f"(+ {t3 - t2:.1f} s pool start-up, untimed); lossless={ok_all}"}
        # the headline CPU figure is the stronger baseline: every usable host core busy
        out.update({"value": allc["value"], "cores": ncpu, "all_cores": allc,
                    "sample": f"{ncpu} cores, {n_all} images in {da:.1f} s, 4x64x64, embed+recover, scalar scipy.stats.norm.ppf/cdf per element (reference-shaped port of "
                              f"gs_insert.py:49-66 + extract.py:72-101); 1 core: {one['value']:.2f} images/s; lossless={bool(ok and ok_all)}"})
    return out


def self_launch(args) -> int:
    """`python bench.py --gpus N` from a bare shell (no torch.distributed.run around it): start N rank processes -- one per GPU, RCCL over
    127.0.0.1 -- poll them, stop the siblings as soon as one fails, and return the worst exit code.  The parent never touches HIP (it does
    not even import torch: the launcher is loaded by file path), and no process replaces itself: the ranks are plain children, rank 0
    prints the one JSON line."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gswm_launch", os.path.join(ROOT, "a-watermark-for-diffusion-models_amd", "launch.py"))
    launch = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(launch)
    limit = float(os.environ.get("GSW_BENCH_TIMEOUT_S", "3600"))
    return launch.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus, timeout=limit)


def init_dist(args):
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    backend = os.environ.get("GSW_BENCH_BACKEND", "nccl")      # "gloo": launcher self-test on a CPU-only host
    if backend == "nccl":
        assert torch.cuda.is_available(), "bench.py needs a GPU"
        torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, world, local_rank


def ranks_seen(world, local_rank):
    """[local_rank of every rank], all-gathered: evidence in the JSON line that N distinct ranks took part"""
    import torch
    import torch.distributed as dist
    if world == 1:
        return [local_rank]
    dev = torch.device("cuda", local_rank) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([local_rank], dtype=torch.int64, device=dev)
    out = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(out, t)
    return [int(v) for v in out.tolist()]


def run_codec(args, rank, world, local_rank, steps, warmup, cpu_codec=None):
    """Codec tier: returns the result dict on rank 0 (None elsewhere)."""
    import torch
    import torch.distributed as dist
    import gswm_amd
    from gswm_amd import codec, dist as gdist

    lib = gswm_amd._native.lib()  # raises if the HIP library is missing: no fallback
    h, w = args.height // 8, args.width // 8
    shape = (4, h, w)
    n = 4 * h * w
    M = args.message_length
    B = args.codec_batch
    fast = not args.exact

    # rank 0 owns the secrets; RCCL broadcast to the other ranks
    params = gdist.broadcast_params(
        {"key": bytes.fromhex(README_KEY), "nonce": bytes.fromhex(README_NONCE), "message": codec.pad_message("lthero", M // 8),
         "seed": 2024, "height": args.height, "width": args.width} if rank == 0 else None, src=0)
    key, nonce, k = params["key"], params["nonce"], params["message"]

    dev = torch.device("cuda", local_rank)
    z32 = torch.empty((B, *shape), dtype=torch.float32, device=dev)          # embed output (resident)
    # extract input: fp16 latents of B watermarked images (what the inversion hands back, extract.py:48,70)
    z16 = codec.embed_batch(key, nonce, k, B, shape, seed=params["seed"], image_index0=rank * B, dtype=torch.float16, fast=fast, device=dev)
    bits_all = torch.empty((world, B, M // 8), dtype=torch.uint8, device=dev) if world > 1 else None

    two = args.codec_streams == 2
    side = torch.cuda.Stream(device=dev) if two else None

    def step(i, ev=None):
        idx0 = (i * world + rank) * B                                         # global image index: independent of the GPU count
        if two and ev is None:
            # the two kernels of a step touch disjoint buffers (embed writes z32, extract reads z16): the embed goes to a side stream, the extract stays on the
            # current one, and the step ends when both have (the side stream joins the current stream again)
            cur = torch.cuda.current_stream(dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                codec.embed_batch(key, nonce, k, B, shape, seed=params["seed"], image_index0=idx0, fast=fast, out=z32)
            bits, flags = codec.extract_batch(z16, key, nonce, M)
            cur.wait_stream(side)
            return bits, flags
        if ev: ev[0].record()
        codec.embed_batch(key, nonce, k, B, shape, seed=params["seed"], image_index0=idx0, fast=fast, out=z32)
        if ev: ev[1].record()
        bits, flags = codec.extract_batch(z16, key, nonce, M)
        if ev: ev[2].record()
        return bits, flags

    for i in range(warmup):
        bits, flags = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
    handles = []
    t0 = time.perf_counter()
    for i in range(steps):
        bits, flags = step(warmup + i, None if two else events[i])
        if world > 1:  # gather of the recovered bitstrings (1 KiB-class, latency-bound), overlapped with the next step
            handles.append(dist.all_gather_into_tensor(bits_all.view(-1), bits.view(-1), async_op=True))
    for hd in handles:
        hd.wait()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # correctness gate (outside the timed region): lossless => every bit of every image
    want = torch.frombuffer(bytearray(k), dtype=torch.uint8).to(dev)
    ok_bits = bool((bits == want[None]).all()) and int(flags.abs().sum()) == 0
    b2, f2 = codec.extract_batch(z32.half(), key, nonce, M)                   # the images embedded in the last timed step
    ok_bits &= bool((b2 == want[None]).all()) and int(f2.abs().sum()) == 0
    matches = codec.bit_matches(bits, M, k).sum()
    if world > 1:
        dist.all_reduce(matches)
        ok_t = torch.tensor([1 if ok_bits else 0], device=dev)
        dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
        ok_bits = bool(ok_t.item())
        if rank == 0:
            ok_bits &= bool((bits_all == want[None, None]).all())
    bit_acc = float(matches.item()) / (world * B * M)

    if two:
        # per-kernel durations for the roofline: the same launches one after the other on ONE stream, HIP events around each, outside the timed region
        for i in range(steps):
            step(warmup + steps + i, events[i])
        torch.cuda.synchronize()
    t_embed = sum(e[0].elapsed_time(e[1]) for e in events) / steps * 1e-3   # s per launch
    t_extract = sum(e[1].elapsed_time(e[2]) for e in events) / steps * 1e-3
    bytes_embed = 4.0 * n * B               # fp32 Z_s_T written (in-kernel RNG: nothing read)
    bytes_extract = 2.0 * n * B + B * (M // 8)
    dom = ("gsw_embed_kernel", bytes_embed, t_embed) if t_embed >= t_extract else ("gsw_extract_wave_kernel", bytes_extract, t_extract)

    # HBM traffic per launch from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE), when it was collected on
    # this very configuration; otherwise null
    traffic = traffic_src = None
    try:
        pmc_file = os.path.join("profiles", "r06_codec_pmc.json")          # this round's passes (the codec kernels themselves are unchanged since round 4)
        pmc = json.load(open(os.path.join(ROOT, pmc_file)))
        c = pmc["config"]
        if c["batch_per_gpu"] == B and c["lattice"] == list(shape) and c["message_bits"] == M and fast:
            traffic = pmc["kernels"]["gsw_embed_kernel" if dom[0] == "gsw_embed_kernel" else "gsw_extract_wave_kernel"]["traffic_bytes"]
            traffic_src = f"{pmc_file} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this configuration, kernel revision {pmc.get('kernel_revision')})"
    except Exception:
        pass

    out = None
    if rank == 0:
        total_images = world * B * steps
        out = {
            "metric": "watermarked images/sec (embed+extract, 512x512 SD2.1) + lossless bit-accuracy",
            "value": total_images / dt, "unit": "images/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if fast else "f64", "data": "synthetic",
            "config": {"workload": f"codec tier: gsw_embed (fp32 out, Philox u, {'fp32 fast' if fast else 'fp64 Cephes'} ndtri) + gsw_extract "
                                   f"(fp16 in, {M}-bit vote) on {shape[0]}x{shape[1]}x{shape[2]} lattices, inputs resident in HBM",
                       "batch_per_gpu": B, "global_batch": world * B, "lattice": list(shape), "message_bits": M, "hip_streams_per_step": args.codec_streams,
                       "parallelism": f"dp{world} (images sharded, no data-path collective; async all-gather of recovered bits)"},
            "bit_accuracy": bit_acc, "lossless": ok_bits,
            "roofline": {"bound": "hbm", "kernel": dom[0], "achieved": dom[1] / dom[2] / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": dom[1] / dom[2] / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": dom[1], "avg_launch_us": dom[2] * 1e6,
                         "measured_in": ("single-stream launches after the timed region (the timed steps overlap embed and extract on two streams)" if two
                                         else "the timed region"),
                         "step_GBps": (bytes_embed + bytes_extract) * steps / dt / 1e9,
                         "kernels": {"gsw_embed_kernel": {"bytes": bytes_embed, "avg_us": t_embed * 1e6, "GBps": bytes_embed / t_embed / 1e9},
                                     "gsw_extract_wave_kernel": {"bytes": bytes_extract, "avg_us": t_extract * 1e6, "GBps": bytes_extract / t_extract / 1e9}}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_codec if cpu_codec is not None else cpu_baseline(args.cpu_images, M, all_cores=False)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
    if not ok_bits:
        print("bench.py: codec tier lost bits on lossless input", file=sys.stderr)
        sys.exit(3)
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    if args.preflight:
        if "WORLD_SIZE" not in os.environ:                 # one GPU from a bare shell: a world of one, still through the same stages
            import importlib.util
            spec = importlib.util.spec_from_file_location("gswm_launch", os.path.join(ROOT, "a-watermark-for-diffusion-models_amd", "launch.py"))
            launch = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(launch)        # (a free port, not a fixed one: a busy port would read as a rendezvous failure)
            os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT") or str(launch._free_port()))
        import gswm_amd
        from gswm_amd import dist as gdist
        sys.exit(gdist.preflight_main(os.environ.get("GSW_BENCH_BACKEND", "nccl"), float(os.environ.get("GSW_PREFLIGHT_TIMEOUT_S", "120"))))
    cpu_codec = None
    if args.gpus == 1 and not args.no_cpu_baseline and not args.launcher_selftest and args.tier in ("both", "codec"):
        cpu_codec = cpu_baseline(args.cpu_images, args.message_length)        # before any HIP call: the all-cores leg starts worker processes
    rank, world, local_rank = init_dist(args)
    import torch.distributed as dist
    seen = ranks_seen(world, local_rank)
    if args.launcher_selftest:
        if rank == 0:
            print(json.dumps({"launcher_selftest": True, "n_gpus": world, "ranks_seen": seen}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    codec_res = e2e_res = None
    if args.tier in ("both", "codec"):
        k = args.steps if args.tier == "codec" else args.codec_steps
        w = args.warmup if args.tier == "codec" else 5
        codec_res = run_codec(args, rank, world, local_rank, k, w, cpu_codec)
    if args.tier in ("both", "e2e"):
        from bench_e2e import run_e2e
        e2e_res = run_e2e(args, rank, world, local_rank)
    if rank == 0:
        if args.tier == "codec":
            out = codec_res
        else:
            # primary line = the metric as BASELINE.json names it (embed + DDIM-inversion extract, batch 64, SD2.1 512x512);
            # the kernel-level tier of the hand-written HIP codec rides along under "tiers"
            out = e2e_res
            if codec_res is not None:
                out["tiers"] = {"codec": {kk: codec_res[kk] for kk in ("value", "unit", "ms_per_step", "steps", "config", "bit_accuracy",
                                                                          "lossless", "roofline", "cpu_baseline", "speedup_vs_cpu_baseline")
                                          if kk in codec_res}}
        out["ranks_seen"] = seen
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
