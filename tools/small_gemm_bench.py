"""Dense linears at small M: the matmul engine (with its automatic split-K) against the small-M kernel (gsw_gemm_small) per tile configuration, on the
shapes of a one- / two-image forward.  Launches are graph-captured back to back and cycle through enough weight copies that the weights come from HBM
(as in a forward, where 1.7 GB of weights go by between two uses), not from L2 / Infinity Cache.  Also prints the error of both against fp32."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import gswm_amd  # noqa: E402
from gswm_amd import pf, _native as N  # noqa: E402
from gswm_amd.codec import _dt, _stream_ptr  # noqa: E402

dt = torch.float16
REP = 32
MODES = {"plain": 0, "geglu": 1, "trans": 2}


def timed(fn, ws=None):
    fn(0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    wsb = torch.empty(pf.SPLITK_BYTES, dtype=torch.uint8, device="cuda")
    with pf.splitk_workspace(wsb), torch.cuda.graph(g):
        for i in range(REP):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * REP)


def small(x, w, b, y, mode, cfg, S=0):
    M, K = x.shape
    Nn = w.shape[0]
    ncols = Nn // 2 if mode == "geglu" else Nn
    N.check(N.lib().gsw_gemm_small(x.data_ptr(), K, w.data_ptr(), K, b.data_ptr() if b is not None else None, None, ncols, y.data_ptr(), ncols, M, K, Nn, MODES[mode], S, 0,
                                   None, 0, 0.0, None, None, None, 0, None, cfg, _dt(x.dtype), _stream_ptr()))


rows_list = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 2]
for B in rows_list:
    print(f"--- {B} image(s)")
    for (S, K, Nn, mode) in ((4096, 320, 320, "plain"), (4096, 320, 640, "plain"), (4096, 320, 320, "trans"), (4096, 1280, 320, "plain"), (4096, 320, 2560, "geglu"),
                             (1024, 640, 640, "plain"), (1024, 640, 1280, "plain"), (1024, 640, 640, "trans"), (1024, 2560, 640, "plain"), (1024, 640, 5120, "geglu"),
                             (256, 1280, 1280, "plain"), (256, 1280, 2560, "plain"), (256, 1280, 1280, "trans"), (256, 5120, 1280, "plain"), (256, 1280, 10240, "geglu"),
                             (64, 1280, 1280, "plain"), (64, 1280, 2560, "plain"), (64, 1280, 1280, "trans"), (64, 5120, 1280, "plain"), (64, 1280, 10240, "geglu")):
        M = B * S
        nW = max(2, min(64, (600 << 20) // (Nn * K * 2)))
        x = torch.randn(M, K, device="cuda", dtype=dt)
        ws = [(torch.randn(Nn, K, device="cuda") * K ** -0.5).to(dt) for _ in range(nW)]
        b = torch.randn(Nn, device="cuda", dtype=dt)
        if mode == "geglu":
            packed = [pf.pack_geglu_weight(w, b) for w in ws]
            ws_e, b_e = [p[0] for p in packed], packed[0][1]
        else:
            ws_e, b_e = ws, b
        xs = x.view(B, S, K)
        t_eng = timed(lambda i: pf.gemm(xs, ws_e[i % nW], b_e, mode=mode, tokens=S if mode == "trans" else 0))
        y_eng = pf.gemm(xs, ws_e[0], b_e, mode=mode, tokens=S if mode == "trans" else 0)
        ref = x.float() @ ws[0].float().t() + b.float()
        if mode == "geglu":
            ref = ref[:, : Nn // 2].half().float() * torch.nn.functional.gelu(ref[:, Nn // 2:].half().float()).half().float()
        if mode == "trans":
            ref = ref.view(B, S, Nn).transpose(1, 2)
        ref = ref.reshape(y_eng.shape)
        scale = ref.abs().max().item()
        line = f"{mode:5s} M={M:5d} K={K:5d} N={Nn:5d}  engine {t_eng:6.1f} us (err {(y_eng.float() - ref).abs().max().item() / scale:.1e})  small:"
        auto = N.lib().gsw_gemm_small_config(M, K, Nn, MODES[mode])
        for cfg in range(4):
            y = torch.empty_like(y_eng)
            try:
                small(x, ws_e[0], b_e, y, mode, cfg, S)
            except Exception as e:  # noqa: BLE001
                line += f"  c{cfg}: {type(e).__name__}"
                continue
            err = (y.float() - ref).abs().max().item() / scale
            t = timed(lambda i: small(x, ws_e[i % nW], b_e, y, mode, cfg, S))
            line += f"  c{cfg}{'*' if cfg == auto else ' '}{t:6.1f} ({err:.0e})"
        print(line, flush=True)
