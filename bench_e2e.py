"""bench.py --tier e2e: watermarked images/sec end to end on the SD2.1-base-shaped UNet (synthetic weights):
one step = embed(B) -> 50-step DDIM sampling with CFG 7.5 (2B-row UNet batch) -> [VAE decode -> uint8 image -> ToTensor -> VAE encode]
-> 50-step DDIM inversion (prompt "") -> fused last step + vote.  The image stages are computed and timed, but with synthetic VAE
weights (not an autoencoder) the inversion consumes the latent: see DESIGN.md for why the lossless gate is defined there."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
README_KEY = "5822ff9cce6772f714192f43863f6bad1bf54b78326973897e6b66c3186b77a7"
README_NONCE = "05072fd1c2265f6f2e2a4080a2bfbdd8"
MFMA_PEAK_TFLOPS = 2500.0
PMC_FILE = "r06_e2e_dominant_kernel_pmc.json"     # HBM traffic of the dominant kernel from a committed rocprofv3 --pmc pass of this round


class TimedModel:
    """Wraps the eps model: HIP events around every forward (on torch's current stream, where the kernels are launched)."""

    def __init__(self, model, flops_per_row):
        import torch
        self.model, self.flops_per_row, self.torch = model, flops_per_row, torch
        self.events, self.flops, self.enabled = [], 0.0, False
        self.prefix_flops_per_row = 0.0          # FLOPs of the context-free prefix of a forward (unet.count_cfg_shared_prefix_flops), set by run_e2e

    @property
    def supports_cfg_dup(self):
        return bool(getattr(self.model, "supports_cfg_dup", False))

    def __call__(self, x, t, ctx, **kw):
        if not self.enabled:
            return self.model(x, t, ctx, **kw)
        s, e = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        s.record()
        y = self.model(x, t, ctx, **kw)
        e.record()
        # rows of the guidance batch: the shared prefix of a cfg_dup forward runs once for two rows -- its FLOPs are counted once (prefix_flops_per_row)
        self.events.append((s, e, ctx.shape[0] if kw.get("cfg_dup") else x.shape[0], x.shape[0] if kw.get("cfg_dup") else 0))
        return y

    def summary(self):
        tot_ms = sum(s.elapsed_time(e) for s, e, _, _ in self.events)
        rows = sum(r for _, _, r, _ in self.events)
        shared = sum(d for _, _, _, d in self.events)
        flops = rows * self.flops_per_row - shared * self.prefix_flops_per_row      # EXECUTED FLOPs
        return {"calls": len(self.events), "avg_ms": tot_ms / max(1, len(self.events)), "total_ms": tot_ms,
                "tflops": flops / (tot_ms * 1e-3) / 1e12 if tot_ms else 0.0, "flops_per_call_avg": flops / max(1, len(self.events)),
                "rows": rows, "rows_whose_context_free_prefix_was_shared": shared, "prefix_flops_per_row": self.prefix_flops_per_row}


def cpu_baseline_e2e(unet_cfg, ddim_steps, M, h=64, w=64, with_vae=False, forwards_per_step=3):
    """Bounded CPU sample of the same workload: the oracle's reference-shaped codec port on 1 image + ONE fp32 UNet forward
    of one image on the host cores (torch CPU), extrapolated to the 3*S forwards an image needs (2S with CFG + S inversion)."""
    import types
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gs_oracle as O
    import gswm_amd
    from gswm_amd import unet as U
    opt = types.SimpleNamespace(key_hex=README_KEY, nonce_hex=README_NONCE)
    a = types.SimpleNamespace(key=bytes.fromhex(README_KEY), nonce=bytes.fromhex(README_NONCE), l=1, message_length=M)
    np.random.seed(0)
    t0 = time.perf_counter()
    z = O.gs_watermark_init_noise_scalar(opt, "lthero")
    O.recover_exactracted_message_scalar(z.astype(np.float16), a)
    t_codec = time.perf_counter() - t0
    m = U.synthetic_init_(U.UNet2DCondition(**unet_cfg), 0).float().eval()
    x = torch.randn(1, 4, h, w); t = torch.tensor([500]); c = torch.randn(1, 77, unet_cfg.get("cross_attention_dim", 1024))
    with torch.no_grad():
        m(x, t, c)
        t1 = time.perf_counter()
        n_fw = 3
        for _ in range(n_fw):
            m(x, t, c)
        t_fw = (time.perf_counter() - t1) / n_fw
    per_image = t_codec + forwards_per_step * ddim_steps * t_fw
    vae_note = ""
    if with_vae:
        from gswm_amd import vae as V
        v = V.synthetic_init_(V.AutoencoderKL(), 1).float().eval()
        with torch.no_grad():
            t2 = time.perf_counter()
            img = V.latents_to_img(x, v)
            V.img_to_latents(img, v)
            t_vae = time.perf_counter() - t2
        per_image += t_vae
        vae_note = f" + 1 full fp32 VAE decode+encode ({t_vae:.2f} s, timed whole)"
    return {"value": 1.0 / per_image, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port", "extrapolated": True,
            "measured_s": {"codec_1_image": t_codec, "unet_forward_fp32": t_fw, "unet_forwards_timed": n_fw, "forwards_per_image": forwards_per_step * ddim_steps},
            "sample": f"EXTRAPOLATED from 1 image: reference-shaped scalar codec port ({t_codec:.2f} s, 1 core) + {n_fw} fp32 forwards of THIS BUILD'S OWN torch "
                      f"UNet module (the reference's diffusers model is not available) on the host ({t_fw:.2f} s each, {torch.get_num_threads()} threads), "
                      f"extrapolated to {forwards_per_step * ddim_steps} forwards/image" + vae_note,
            "host_cpus": os.cpu_count()}


def _bucket_scalars(fam, dt_instr):
    """The per-bucket rates as SCALAR keys of `roofline` (records that keep only scalars still carry the bucket furthest below the roof)."""
    short = {"gsw_mm_kernel": "dense", "gsw_mm_kernel(conv3x3)": "conv3x3", "gsw_mm_kernel(conv1x1)": "conv1x1", "gsw_mm_kernel(conv3x3 s2)": "conv3x3s2",
             "gsw_mm_kernel(up2x)": "up2x"}
    out = {}
    for k, v in fam.items():
        n = short.get(k, k)
        out[n + "_tflops"] = v["tflops"]
        out[n + "_frac"] = v["tflops"] / MFMA_PEAK_TFLOPS
        out[n + "_step_time_fraction"] = v["ms"] * 1e-3 / dt_instr
    if fam:
        worst = min((k for k in fam if fam[k]["ms"] * 1e-3 / dt_instr >= 0.05), key=lambda k: fam[k]["tflops"], default=None)
        if worst is not None:
            out["furthest_below_roofline"] = short.get(worst, worst)
    return out


def run_e2e(args, rank, world, local_rank):
    import torch
    import torch.distributed as dist
    import gswm_amd
    from gswm_amd import codec, unet as U, dist as gdist
    from gswm_amd.pipeline import GaussianShadingPipeline

    gswm_amd._native.lib()
    dev = torch.device("cuda", local_rank)
    B = args.batch
    M = args.message_length
    S = args.ddim_steps
    dtype = torch.float16
    params = gdist.broadcast_params(
        {"key": bytes.fromhex(README_KEY), "nonce": bytes.fromhex(README_NONCE), "message": codec.pad_message("lthero", M // 8),
         "seed": 2024, "height": args.height, "width": args.width} if rank == 0 else None, src=0)
    unet_cfg = {} if args.unet == "sd21" else dict(cross_attention_dim=768, num_heads=(8, 8, 8, 8), head_dim=None)
    ctx_dim = unet_cfg.get("cross_attention_dim", 1024)
    model = U.synthetic_init_(U.UNet2DCondition(**unet_cfg), seed=0).to(dev, dtype).eval()
    if os.environ.get("GSW_CHANNELS_LAST", "0") == "1":   # NCHW measured faster here (377 vs 343 TFLOP/s at B=64)
        model = model.to(memory_format=torch.channels_last)
    h, w = args.height // 8, args.width // 8
    flops_row = U.count_flops_per_image(model, h, w)
    from gswm_amd.graph import graphed
    eps = graphed(model, clone_output=False)                                    # forwards of <= 32 rows replay a captured HIP graph (GSW_GRAPH=never: eager A/B)
    tm = TimedModel(eps, flops_row)
    tm.prefix_flops_per_row = float(U.count_cfg_shared_prefix_flops(model, h, w)) if U.CFG_SHARED_PREFIX else 0.0
    g = torch.Generator(device="cpu").manual_seed(1)
    ctx_uncond = (torch.randn(1, 77, ctx_dim, generator=g) * 1.0).to(dev, dtype)        # stands for CLIP("")
    ctx_text = (torch.randn(B, 77, ctx_dim, generator=g) * 1.0).to(dev, dtype)          # stands for CLIP(prompt)
    pipe = GaussianShadingPipeline(tm, params["key"], params["nonce"], params["message"], height=args.height, width=args.width,
                                   num_inference_steps=S, dtype=dtype, device=dev, ctx_uncond=ctx_uncond)
    want = torch.frombuffer(bytearray(params["message"]), dtype=torch.uint8).to(dev)

    vae = None
    if args.image_stages != "none":
        from gswm_amd import vae as V, imaging
        vae = V.synthetic_init_(V.AutoencoderKL(), 1).to(dev, dtype).eval()

    def image_stages(x0):
        """G1 tail + X1 per image: VAE decode -> uint8 RGB (the watermarked image) -> [JPEG QF] -> ToTensor/fp16/2x-1 -> VAE encode.
        Synthetic VAE weights are not an autoencoder, so the re-encoded latents are computed (and timed) but the inversion below
        consumes x0: the lossless gate lives at the latent level."""
        outs = []
        for chunk in x0.split(args.vae_chunk):
            img = V.latents_to_img(chunk, vae)
            u8 = imaging.tensor_to_image(img)
            xn = imaging.jpeg_roundtrip(u8, args.jpeg_qf, out="f16") if args.image_stages == "vae+jpeg" else imaging.to_tensor(u8, out="f16")
            outs.append(V.normalised_img_to_latents(xn, vae))
        return torch.cat(outs)

    txt2img = getattr(args, "workload", "roundtrip") == "txt2img"
    if txt2img and vae is None:
        raise SystemExit("bench.py: --workload txt2img ends in the VAE decode: it needs --image-stages vae")

    def step(i):
        idx0 = (i * world + rank) * B
        if txt2img:
            # BASELINE configs[1]: the generation pipeline's call (modified_stable_diffusion_gs.pyc __call__; README.md:107-129) -- Z_s_T from the embed kernel,
            # CFG sampling, decode_latents, numpy_to_pil's uint8 conversion; nothing is extracted
            z_T = pipe.embed(B, seed=params["seed"], image_index0=idx0)
            x0 = pipe.generate(z_T, ctx_text, 7.5)
            with torch.no_grad():
                for chunk in x0.split(args.vae_chunk):
                    imaging.tensor_to_image(V.latents_to_img(chunk, vae))
            return z_T, x0, None, None
        if vae is None:
            return pipe.roundtrip(B, ctx_text, seed=params["seed"], image_index0=idx0, guidance_scale=7.5)
        z_T = pipe.embed(B, seed=params["seed"], image_index0=idx0)
        x0 = pipe.generate(z_T, ctx_text, 7.5)
        with torch.no_grad():
            image_stages(x0)
        bits, flags = pipe.invert_and_extract(x0)
        return z_T, x0, bits, flags

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    U.FALLBACKS.clear()
    if vae is not None:
        V.FALLBACKS.clear()
    matched = torch.zeros((), dtype=torch.int64, device=dev)
    flagged = torch.zeros((), dtype=torch.int64, device=dev)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]      # step boundaries (recorded, never waited on inside the region)
    from bench_board import BoardSampler
    board = BoardSampler(local_rank)               # shader clock + board power of this rank's device while the region runs (a 20 Hz sysfs reader thread)
    board.__enter__()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        z_T, x0, bits, flags = step(args.warmup + i)
        if bits is not None:
            matched += codec.bit_matches(bits, M, params["message"]).sum()
            flagged += (flags != 0).sum()
        marks[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    board.__exit__()
    if txt2img:
        # nothing was extracted inside the region: the lossless gate of this workload is the embedded Z_s_T of the last step read back by the extract kernel
        bits, flags = codec.extract_batch(z_T, params["key"], params["nonce"], M)
        matched += codec.bit_matches(bits, M, params["message"]).sum() * args.steps
        flagged += (flags != 0).sum()
    fallbacks = dict(U.FALLBACKS)
    if vae is not None:
        fallbacks.update({"vae: " + k: v for k, v in V.FALLBACKS.items()})
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    # ONE more step, instrumented (HIP events around every UNet forward and every convolution / matmul launch, on the stream they are
    # launched on): it feeds the roofline objects and stays OUT of the timed region above -- the events cost launch slots
    from gswm_amd import pf as _pf
    # A graph replay executes no Python, so the per-launch events would only see the VAE: whenever the timed region replayed graphs the instrumented
    # step runs the forwards EAGERLY (same kernels, same arguments -- tests/test_gpu_graph.py -- but with the host launch path in between) and the
    # line says so; fallbacks baked into a captured graph are reported from the capture (graph.GraphedEpsModel.capture_fallbacks).
    conv_timer = _pf.ConvTimer()
    tm.enabled = True
    _pf.CONV_TIMER = conv_timer
    replayed = eps.stats["replays"] > 0
    mode_before, eps.mode = eps.mode, ("never" if replayed else eps.mode)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    step(args.warmup + args.steps)
    e1.record()
    torch.cuda.synchronize()
    eps.mode = mode_before
    _pf.CONV_TIMER = None
    tm.enabled = False
    measured_in = ("one instrumented EAGER step after the timed region (the timed region replays HIP graphs of the same launches; event times here include the "
                   "host launch gaps, so per-launch rates are lower bounds)" if replayed else "one instrumented step after the timed region")
    for k_, v_ in getattr(eps, "capture_fallbacks", {}).items():
        fallbacks["captured graph: " + k_] = fallbacks.get("captured graph: " + k_, 0) + v_
    dt_instr = e0.elapsed_time(e1) * 1e-3
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        dist.all_reduce(matched)
        dist.all_reduce(flagged)
    bit_acc = float(matched.item()) / (world * B * args.steps * M)
    sm = tm.summary()
    if rank == 0:
        total_images = world * B * args.steps
        out = {
            "metric": "watermarked images/sec (embed+extract, 512x512 SD2.1) + lossless bit-accuracy",
            "value": total_images / dt, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": (f"txt2img (BASELINE configs[1]): gsw_embed -> {S}-step DDIM sampling (CFG 7.5, 2B-row UNet) -> VAE decode -> uint8 image, no "
                                    f"extraction (bit accuracy = the embedded Z_s_T read back after the timed region); " if txt2img else
                                    f"e2e: gsw_embed -> {S}-step DDIM sampling (CFG 7.5, 2B-row UNet) -> {S}-step DDIM inversion -> "
                                    f"fused last step + {M}-bit vote; ") + f"{'SD2.1-base' if args.unet == 'sd21' else 'SD1.5'}-shaped UNet ({sum(p.numel() for p in model.parameters()) / 1e6:.1f} M params, synthetic weights; convolutions on the hand-written MFMA implicit GEMM), "
                                   + ("no VAE" if vae is None else "image stage per image: VAE decode -> uint8 (timed)" if txt2img else f"image stages per image: VAE decode -> uint8 -> {'JPEG QF ' + str(args.jpeg_qf) + ' -> ' if args.image_stages == 'vae+jpeg' else ''}ToTensor -> VAE encode (timed; synthetic VAE, inversion consumes the latent)"),
                       "batch_per_gpu": B, "global_batch": world * B, "lattice": [4, h, w], "message_bits": M, "ddim_steps": S,
                       "parallelism": f"dp{world} (images sharded, UNet replicated, no data-path collective)"},
            "bit_accuracy": bit_acc, "lossless": bit_acc == 1.0 and int(flagged.item()) == 0, "flagged_images": int(flagged.item()),
            # latency of one step = one batch through embed -> sampling -> image stages -> inversion -> vote (device time between step marks)
            "latency": {"unit": "s per batch of %d" % B, "p50": step_ms[len(step_ms) // 2] * 1e-3, "min": step_ms[0] * 1e-3, "max": step_ms[-1] * 1e-3,
                        "steps": len(step_ms), "p50_per_image_s": step_ms[len(step_ms) // 2] * 1e-3 / B},
            "eps_model_launch_path": {"mode": eps.mode, "graph_max_rows": eps.max_rows, **eps.stats},
        }
        cs = conv_timer.summary()
        out["fallbacks_off_the_hand_written_path"] = fallbacks
        # what the board did during the timed region: under the 1400 W limit the MFMA-heavy kernels of this path run at 1.7-2.0 GHz on random operands
        # (profiles/r04h_power_cap_probe.txt), so the fractions below -- against the 2.4 GHz peak -- come with the clock they were measured at
        out["board"] = board.summary()
        if cs:
            # The dominant kernel is the matmul engine, gsw_mm_kernel<T, EPI, SPLIT, MT>: ONE kernel template that runs every convolution and every
            # dense linear of UNet and VAE.  `roofline` is the whole family (every instantiation): achieved = the FLOPs the kernel EXECUTES for the
            # operators the model defines (2 * real output rows * N * K; padded border rows not counted; the sub-pixel upsampler with the FLOPs it
            # executes) / the HIP-event time of its launches; `buckets` splits it by what the launches compute (the dense-linear bucket is the one
            # furthest from the roof).  rocprofv3's per-kernel average for the same command is committed under profiles/.
            fam = {k: v for k, v in cs.items() if k.startswith("gsw_mm_kernel")}
            rest = {k: v for k, v in cs.items() if not k.startswith("gsw_mm_kernel")}
            f_flops, f_ms, f_calls = sum(v["flops"] for v in fam.values()), sum(v["ms"] for v in fam.values()), sum(v["calls"] for v in fam.values())
            f_bytes = sum(v["bytes"] for v in fam.values())
            f_tflops = f_flops / (f_ms * 1e-3) / 1e12
            label = {"gsw_mm_kernel": "dense linears (EPI 0 rows / EPI 2 GEGLU / EPI 3 transposed)"}
            traffic = traffic_src = None
            try:
                pmc_file = os.path.join("profiles", PMC_FILE)
                pmc = json.load(open(os.path.join(ROOT, pmc_file)))
                if pmc["kernel"] == "gsw_mm_kernel" and pmc["config"] == {"batch": B, "unet": args.unet, "height": args.height, "width": args.width}:
                    traffic, traffic_src = pmc["traffic_bytes_per_launch"], f"{pmc_file} ({pmc.get('how')})"
            except Exception:
                pass
            out["roofline"] = {"bound": "mfma", "kernel": "gsw_mm_kernel (matmul engine: every instantiation, convolutions + dense linears)",
                               "achieved": f_tflops, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": f_tflops / MFMA_PEAK_TFLOPS,
                               "traffic": traffic, "traffic_source": traffic_src,
                               # every operand read once + the output written once, averaged over the family's launches: `traffic` / this = the waste ratio
                               "algorithmic_bytes_per_launch": f_bytes / f_calls, "traffic_over_algorithmic": (traffic / (f_bytes / f_calls)) if traffic and f_bytes else None,
                               "algorithmic_flops_per_launch": f_flops / f_calls, "avg_launch_us": f_ms * 1e3 / f_calls, "calls": f_calls,
                               "step_time_fraction": f_ms * 1e-3 / dt_instr, "measured_in": measured_in,
                               **_bucket_scalars(fam, dt_instr),
                               "buckets": {label.get(k, k): {kk: v[kk] for kk in ("calls", "avg_us", "tflops")} | {"frac": v["tflops"] / MFMA_PEAK_TFLOPS, "step_time_fraction": v["ms"] * 1e-3 / dt_instr}
                                           for k, v in fam.items()},
                               "other_kernels": {k: {kk: v[kk] for kk in ("calls", "avg_us", "tflops")} | {"step_time_fraction": v["ms"] * 1e-3 / dt_instr}
                                                 for k, v in rest.items()}}
        out["roofline_unet"] = {"bound": "mfma", "kernel": "UNet2DCondition forward, aggregate (every convolution, linear layer and attention on "
                                "the hand-written MFMA kernels; per-kernel shares in profiles/r06_e2e_b64_kernel_stats.csv)",
                                "achieved": sm["tflops"], "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": sm["tflops"] / MFMA_PEAK_TFLOPS,
                                "traffic": None, "algorithmic_flops_per_launch": sm["flops_per_call_avg"], "avg_launch_us": sm["avg_ms"] * 1e3,
                                "calls": sm["calls"], "unet_time_fraction": sm["total_ms"] * 1e-3 / dt_instr, "flops_per_image_forward": flops_row,
                                "measured_in": measured_in}
        if "roofline" not in out:
            out["roofline"] = out["roofline_unet"]
        if out["board"] and out["board"].get("sclk_fraction_of_nominal"):
            for key in ("roofline", "roofline_unet"):
                out[key] = dict(out[key], frac_of_peak_at_sustained_clock=out[key]["frac"] / out["board"]["sclk_fraction_of_nominal"])
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_e2e(unet_cfg, S, M, h, w, with_vae=vae is not None, forwards_per_step=2 if txt2img else 3)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
        return out
    return None
