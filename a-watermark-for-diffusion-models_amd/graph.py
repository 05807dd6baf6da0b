"""HIP-graph replay of the eps model for the small-batch regime of the DDIM loops (rows X2 / G1 of SURVEY.md section 8a).

The reference's own use is one image per call (extract.py:112-117: `exactract_latents` -> 50 UNet evaluations for ONE latent) and BASELINE
configs[1] is batch 8.  There a UNet forward is ~520 kernel launches of a few microseconds each: issued one by one from Python through ctypes the
host cannot keep the GPU fed, so the loop runs at the speed of the launch path, not of the kernels.  The loops of ddim.py are free of host
synchronisation and every operand of a forward except (x, t, context) is a weight, so ONE forward is captured into a HIP graph per
(rows, lattice, dtype, context shape) and replayed for every step of every loop:

    x_s.copy_(x); t_s.copy_(t); graph.replay(); eps = out_s            # 3 tiny launches + one graph launch per step

What the graph bakes in, and how it stays valid:
  * the static input buffers x_s / t_s / ctx_s and the output -- owned by the entry;
  * the padded context copy and the cross-attention K / V^T of every layer (computed ONCE per context, outside the graph, cached on the
    context tensor by unet.Attention) -- when the caller passes a different context, ctx_s is overwritten and those are recomputed IN PLACE
    (unet._padded_ctx / Attention refresh into their existing buffers), so the addresses the graph reads never change;
  * weights and their packed copies -- replacing or editing parameters (load_state_dict, .to(), a LoRA merge) invalidates every graph: the
    parameters' version counters and storage pointers are checked whenever the context changes AND every CHECK_EVERY replays of an entry (a
    harness that keeps one context for its whole life would otherwise replay stale weights for ever); `reset()` forces it;
  * the launch sequence itself -- the module-level switches of unet.py / pf.py and the engine's tiling knobs (gsw_mm_config) are part of an
    entry's key, so an A/B toggle after a capture captures anew instead of silently replaying the old sequence.
Capture failing for any reason falls back to eager IN THIS PROCESS (never a re-exec) and is reported once.  Entries are kept in an LRU of
MAX_ENTRIES (each pins a private activation pool and a 40 MiB split-K scratch).

A graph replay runs exactly the launches of the eager forward with the same arguments: outputs are bit-identical (tests/test_gpu_graph.py).
"""
from __future__ import annotations

import os
import warnings
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch


def _env_mode() -> str:
    return os.environ.get("GSW_GRAPH", "auto").lower()


# "auto": graphs for forwards of at most AUTO_MAX_ROWS rows (above that a forward is tens of milliseconds of GPU work and the launch path is
# hidden behind it; the graph would only pin a second copy of the activation memory)
AUTO_MAX_ROWS = 32
CHECK_EVERY = 32        # replays of an entry between two checks of the parameters' versions (~0.3 ms per check: ~700 parameters)
# Staleness window: an IN-PLACE edit of a parameter tensor (p.data.add_(...), a hand-rolled LoRA merge) under an unchanged context is noticed at the next of
# those periodic checks, i.e. up to CHECK_EVERY - 1 replays later.  Everything that goes through the usual entry points is seen AT ONCE: load_state_dict on the
# wrapped module (a post-hook bumps the epoch below), unet / vae.load_diffusers_state_dict, extract.Models -- and anyone editing weights by hand calls
# graph.weights_changed() (or GraphedEpsModel.reset()).  The epoch is one integer compare per replay.
_WEIGHTS_EPOCH = [0]


def weights_changed() -> None:
    """Tell every GraphedEpsModel of the process that parameters were edited: the next call re-checks the versions before it replays anything."""
    _WEIGHTS_EPOCH[0] += 1

MAX_ENTRIES = 8         # captured (shape, dtype, context shape, switches) entries kept; the least recently used one is dropped beyond that


def _switches():
    """Everything outside the arguments that decides WHICH launches a forward makes: captured graphs are keyed by it."""
    from . import pf, unet as U, _native as N
    import ctypes as C
    tr, sm = C.c_int(0), C.c_int(0)
    N.lib().gsw_mm_get_config(C.byref(tr), C.byref(sm))
    return (U.FUSED_KERNELS, U.USE_PF, U.UPSAMPLE_SUBPIXEL, U.CACHE_CONTEXT_KV, U.FUSED_QK, U.FUSED_QKV, U.OWN_ATTENTION, U.OWN_GEMM, U.TEMB_TABLE,
            U.CONV_OUT_DIRECT_MAX_PIXELS, U.CFG_SHARED_PREFIX, pf.FUSE_GN_STATS, pf.GN_FUSED_MAX_WGS, pf.GN_FUSED_MAX_PIXELS, pf.FOLD_LN, pf.FOLD_LN_MIN_ROWS, pf.SPLITK_MAX,
            pf.SPLITK_BYTES, pf.SMALL_GEMM_MAX_ROWS, pf.ATTN_KEY_SPLIT, tr.value, sm.value)


class _Entry:
    __slots__ = ("graph", "x_s", "t_s", "ctx_s", "out", "ctx_id", "replays", "ws")


def _ctx_identity(ctx: torch.Tensor):
    """What makes two `ctx` arguments "the same context": the tensor that owns the storage (held by reference, so its address cannot be
    recycled for other contents), its version counter, and the view geometry -- `ctx_empty.expand(B, -1, -1)` is a new view object per loop
    but the same context."""
    base = ctx._base if ctx._base is not None else ctx
    return (base, ctx._version, ctx.storage_offset(), tuple(ctx.shape), tuple(ctx.stride()))


def _same_ctx(a, b) -> bool:
    return a[0] is b[0] and a[1:] == b[1:]


class GraphedEpsModel:
    """Callable `eps_model(x, t, ctx)` for ddim.py that replays a captured forward of `model` when that pays, else calls it eagerly.

    mode: "auto" (default; env GSW_GRAPH overrides) -> graphs up to AUTO_MAX_ROWS rows; "always"; "never".
    clone_output (default True): the result is a copy of the graph's static output buffer (4 x H x W values per row: one small launch), so two
    results of the same shape can be held side by side; False hands out the static buffer itself, valid until the next call with the same
    shape -- what the in-repo loops need, which consume eps in the scheduler-step kernel right away."""

    def __init__(self, model, mode: Optional[str] = None, max_rows: int = AUTO_MAX_ROWS, clone_output: bool = True):
        self.model = model
        m = (mode or _env_mode()).lower()
        self.mode = {"1": "always", "0": "never", "on": "always", "off": "never"}.get(m, m)
        if self.mode not in ("auto", "always", "never"):
            raise ValueError(f"GSW_GRAPH / mode must be auto | always | never (got {m!r})")
        self.max_rows = max_rows
        self.clone_output = clone_output
        self.capture_fallbacks: Dict[str, int] = {}        # unet.FALLBACKS counted while capturing: a library kernel baked into a graph is invisible at replay
        self._entries: "OrderedDict[Tuple, _Entry]" = OrderedDict()
        self._weights_key = None
        self._epoch = _WEIGHTS_EPOCH[0]
        if hasattr(model, "register_load_state_dict_post_hook"):
            model.register_load_state_dict_post_hook(lambda _m, _keys: weights_changed())
        self._failed: Dict[Tuple, str] = {}
        self.stats = {"captures": 0, "replays": 0, "eager": 0, "context_refreshes": 0}

    # anything else (parameters(), eval(), ...) is the wrapped module's
    def __getattr__(self, name):
        return getattr(self.__dict__["model"], name)

    def _wants_graph(self, x: torch.Tensor) -> bool:
        if self.mode == "never" or not x.is_cuda:
            return False
        if torch.cuda.is_current_stream_capturing():      # already inside somebody else's capture: just be part of it
            return False
        return self.mode == "always" or x.shape[0] <= self.max_rows

    def _params_key(self):
        # has any parameter been replaced or edited since the graphs were captured?
        return tuple((p.data_ptr(), p._version) for p in self.model.parameters())

    def reset(self):
        self._entries.clear()
        self._failed.clear()

    def _capture(self, key, x, t, ctx, kw) -> Optional[_Entry]:
        e = _Entry()
        e.x_s = torch.empty_like(x, memory_format=torch.contiguous_format).copy_(x)
        e.t_s = torch.empty_like(t).copy_(t)
        e.ctx_s = torch.empty(ctx.shape, dtype=ctx.dtype, device=ctx.device).copy_(ctx)
        e.ctx_id, e.replays = _ctx_identity(ctx), 0
        from . import pf
        # the split-K scratch of the matmul engine is baked into the graph too: one per graph, so replays never share scratch with eager launches
        e.ws = torch.empty(pf.SPLITK_BYTES, dtype=torch.uint8, device=x.device)
        from . import unet as U
        fb_before = dict(U.FALLBACKS)
        try:
            # warm-up on a side stream: builds the packed weights, the time-embedding table, the padded context + cross-attention K / V^T (cached on
            # ctx_s, OUTSIDE the graph's pool), kernel attributes, so that the captured forward contains the per-step launches only.  It already runs
            # with the entry's split-K scratch (no per-stream scratch is left behind)
            side = torch.cuda.Stream(device=x.device)
            side.wait_stream(torch.cuda.current_stream(x.device))
            with torch.cuda.stream(side), torch.no_grad(), pf.splitk_workspace(e.ws):
                self.model(e.x_s, e.t_s, e.ctx_s, **kw)
                self.model(e.x_s, e.t_s, e.ctx_s, **kw)
            torch.cuda.current_stream(x.device).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.no_grad(), pf.splitk_workspace(e.ws), torch.cuda.graph(g):
                e.out = self.model(e.x_s, e.t_s, e.ctx_s, **kw)
            e.graph = g
            for k_, v_ in U.FALLBACKS.items():             # launches off the hand-written path that the capture baked in (counted per warm-up + capture)
                if v_ != fb_before.get(k_, 0):
                    self.capture_fallbacks[k_] = self.capture_fallbacks.get(k_, 0) + 1
        except Exception as exc:      # noqa: BLE001 -- any capture failure means "run eagerly", in this process
            self._failed[key] = f"{type(exc).__name__}: {exc}"
            warnings.warn(f"gswm graph: capture of the eps model for {key} failed ({self._failed[key]}); running it eagerly", RuntimeWarning, stacklevel=3)
            return None
        self.stats["captures"] += 1
        return e

    @torch.no_grad()
    def __call__(self, x: torch.Tensor, t: torch.Tensor, ctx: torch.Tensor, cfg_dup: bool = False) -> torch.Tensor:
        """cfg_dup: passed on to the model (unet.UNet2DCondition.forward: ctx holds 2B contexts for the B latents of x); the row count that decides
        between graph and eager is the context's."""
        kw = {"cfg_dup": True} if cfg_dup else {}
        if not self._wants_graph(ctx if cfg_dup else x):
            self.stats["eager"] += 1
            return self.model(x, t, ctx, **kw)
        if not torch.is_tensor(t):
            t = torch.as_tensor(t, device=x.device)
        key = (tuple(x.shape), x.dtype, str(x.device), tuple(t.shape), t.dtype, tuple(ctx.shape), ctx.dtype, bool(cfg_dup), _switches())
        if key in self._failed:
            self.stats["eager"] += 1
            return self.model(x, t, ctx, **kw)
        e = self._entries.get(key)
        cid = _ctx_identity(ctx)
        if e is None or not _same_ctx(cid, e.ctx_id) or e.replays % CHECK_EVERY == 0 or self._epoch != _WEIGHTS_EPOCH[0]:
            # (checked when a graph is captured, whenever the context changes, every CHECK_EVERY replays, and at once after weights_changed():
            # ~700 parameters, ~0.3 ms)
            self._epoch = _WEIGHTS_EPOCH[0]
            wk = self._params_key()
            if wk != self._weights_key:         # parameters replaced / edited: every captured address or packed copy may be stale
                self._entries.clear()
                self._weights_key = wk
                e = None
        if e is None:
            e = self._capture(key, x, t, ctx, kw)
            if e is None:
                self.stats["eager"] += 1
                return self.model(x, t, ctx, **kw)
            self._entries[key] = e
            while len(self._entries) > MAX_ENTRIES:
                self._entries.popitem(last=False)
        elif not _same_ctx(cid, e.ctx_id):
            # a different context: overwrite the static copy and bring the padded copy + every layer's K / V^T up to date IN PLACE (eagerly, on
            # this stream, ordered in front of the replay).  Holding a reference to the caller's tensor keeps the identity test sound.
            e.ctx_s.copy_(ctx)
            e.ctx_id = cid
            self.model.prepare_context(e.ctx_s)
            self.stats["context_refreshes"] += 1
        e.x_s.copy_(x)
        e.t_s.copy_(t)
        e.graph.replay()
        e.replays += 1
        self.stats["replays"] += 1
        self._entries.move_to_end(key)
        return e.out.clone() if self.clone_output else e.out


def graphed(model, mode: Optional[str] = None, clone_output: bool = True):
    """Wrap an eps model once (idempotent)."""
    return model if isinstance(model, GraphedEpsModel) else GraphedEpsModel(model, mode, clone_output=clone_output)
