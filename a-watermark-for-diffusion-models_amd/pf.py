"""Padded-flat NHWC ("PF") activations and the MFMA implicit-GEMM convolution built on them (csrc/gswm_conv.hip).

A PF tensor stores [B, H, W, C] with a one-pixel zero border as a matrix rows x C, row(b, y, x) = b*(H+2)*(W+2) + y*(W+2) + x,
with G = W+3 zero guard rows at both ends of the allocation.  A 3x3 tap is then a constant row offset, so a convolution is a
single GEMM whose operand loader needs no boundary logic (see the kernel header)."""
from __future__ import annotations

from typing import Optional

import torch

import ctypes as _C

from . import _native as N
from .codec import _dt, _stream_ptr


class PF:
    __slots__ = ("buf", "B", "H", "W", "C", "stats", "border_valid")

    def __init__(self, buf: torch.Tensor, B: int, H: int, W: int, C: int):
        self.buf, self.B, self.H, self.W, self.C = buf, B, H, W, C
        # column records of the launch that produced the payload (ColStats) -- GroupNorm statistics without a pass over the tensor; anything
        # that writes the payload by other means must leave it None
        self.stats: Optional["ColStats"] = None
        # False: the one-pixel border was left unwritten (conv_pf(gn_only=True) whose launch wrote column records); only a record-fed GroupNorm may read the tensor
        self.border_valid = True

    @property
    def G(self) -> int:
        return self.W + 3

    @property
    def M(self) -> int:
        return self.B * (self.H + 2) * (self.W + 2)

    @property
    def rows(self) -> torch.Tensor:
        """[M, C] view of the payload rows (borders included, guards excluded)."""
        return self.buf[self.G:self.G + self.M]

    @property
    def grid(self) -> torch.Tensor:
        """[B, H+2, W+2, C] view."""
        return self.rows.view(self.B, self.H + 2, self.W + 2, self.C)

    @property
    def interior(self) -> torch.Tensor:
        """[B, H, W, C] strided view of the real pixels."""
        return self.grid[:, 1:-1, 1:-1, :]

    @staticmethod
    def empty(B, H, W, C, dtype, device) -> "PF":
        G = W + 3
        M = B * (H + 2) * (W + 2)
        # the guard rows are only ever read by the taps of BORDER output rows, which every kernel overwrites with zeros,
        # so they need to exist but not to hold anything in particular
        return PF(torch.empty((M + 2 * G, C), dtype=dtype, device=device), B, H, W, C)

    @staticmethod
    def zeros(B, H, W, C, dtype, device) -> "PF":
        G = W + 3
        return PF(torch.zeros((B * (H + 2) * (W + 2) + 2 * G, C), dtype=dtype, device=device), B, H, W, C)

    @staticmethod
    def from_nchw(x: torch.Tensor) -> "PF":
        B, C, H, W = x.shape
        p = PF.zeros(B, H, W, C, x.dtype, x.device)
        p.interior.copy_(x.permute(0, 2, 3, 1))
        return p

    def to_nchw(self) -> torch.Tensor:
        return self.interior.permute(0, 3, 1, 2).contiguous()

    def tokens(self) -> torch.Tensor:
        """[B, H*W, C] dense copy of the real pixels (transformer input)."""
        return self.interior.reshape(self.B, self.H * self.W, self.C)


# ---- split-K workspace of the matmul engine (GswMmExtras.workspace_*): one scratch buffer per (device, stream), handed to every launch in its extras.
# Launches on one stream share it (a launch and its reduce kernel are stream-ordered).
SPLITK_BYTES = 40 << 20        # 256 slabs of 160 KiB (256-row tiles; 80 KiB for 128-row tiles): every launch that splits fits
GN_ONLY_SKIPS_BORDER = True    # conv_pf(gn_only=True): no border zeroing behind a convolution whose output only a record-fed GroupNorm reads (A/B switch)
ATTN_KEY_SPLIT = True          # self-attention with few query tiles (one image) splits its keys over several workgroups (gsw_attention_ws)
LAUNCH_LOG = None              # a list: every engine launch appends its GswMmExtras (tests: which launches split, and how)
SPLITK_MAX = 0                 # 0 automatic, 1 never split, k > 1: force k-way splits wherever K allows (parity tests)
_WS = {}
_WS_OVERRIDE: Optional[torch.Tensor] = None


def _workspace(device) -> torch.Tensor:
    """the split-K scratch of the calling thread's (device, current stream) -- or the one a splitk_workspace block pins"""
    ov = _WS_OVERRIDE
    if ov is not None:
        return ov
    dev = torch.device(device)
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _WS.get(key)
    if ws is None:
        ws = _WS[key] = torch.empty(SPLITK_BYTES, dtype=torch.uint8, device=dev)
    return ws


def _extras(device, colstats: Optional[torch.Tensor] = None, rowstats: Optional[torch.Tensor] = None) -> "N.GswMmExtras":
    """GswMmExtras of one launch (include/gswm.h): the records it should write, its split-K scratch and policy"""
    ws = _workspace(device)
    ex = N.GswMmExtras()
    if colstats is not None:
        ex.colstats_dev, ex.colstats_capacity = colstats.data_ptr(), colstats.numel()
    if rowstats is not None:
        ex.rowstats_dev, ex.rowstats_capacity = rowstats.data_ptr(), rowstats.numel()
    ex.workspace_dev, ex.workspace_bytes, ex.max_splits = ws.data_ptr(), ws.numel(), int(SPLITK_MAX)
    if LAUNCH_LOG is not None:
        LAUNCH_LOG.append(ex)          # the launch fills in what it did (ex.splits, record geometry): tests read it afterwards
    return ex


class splitk_workspace:
    """with splitk_workspace(buf): every engine launch of the block uses `buf` (a uint8 device tensor) as its split-K scratch -- graph.py pins one
    per captured graph, so replays never share scratch with eager launches or other graphs."""

    def __init__(self, buf: Optional[torch.Tensor]):
        self.buf = buf

    def __enter__(self):
        global _WS_OVERRIDE
        self.prev, _WS_OVERRIDE = _WS_OVERRIDE, self.buf
        return self

    def __exit__(self, *exc):
        global _WS_OVERRIDE
        _WS_OVERRIDE = self.prev
        return False



# ---- GroupNorm statistics from the producing launch (GswMmExtras.colstats_* / gsw_groupnorm_pf_cs): the engine's convolution / token-scatter
# epilogue also writes per-block, per-column (sum, sum of squares) records of what it stores; the GroupNorm that consumes the tensor folds
# them instead of reading the tensor once more (that pass was 2.6 % of the end-to-end run).
FUSE_GN_STATS = True


class ColStats:
    __slots__ = ("buf", "rows", "npar", "blocks")

    def __init__(self, buf: torch.Tensor, rows: int, npar: int, blocks: int):
        self.buf, self.rows, self.npar, self.blocks = buf, rows, npar, blocks      # blocks: per parity buffer (the buffer's stride)


def _colstats_arm(M: int, Nn: int, device, npar: int = 1, geom=None):
    """The column-record buffer for a launch with M output pixels (per parity launch) and Nn columns -> (buffer, block capacity) or None.
    geom = (B, H, W) of the output: small batches take the one-launch GroupNorm, which reads the tensor itself (no records needed)."""
    if not FUSE_GN_STATS or (geom is not None and _gn_fused_ok(geom[0], geom[1], geom[2], Nn, 32)):
        return None
    cap_blocks = (M + 127) // 128 * 4                 # 32-row blocks of 128-row tiles (64-row blocks of 256-row tiles need fewer)
    buf = torch.empty(npar * cap_blocks * Nn, dtype=torch.float32, device=device)      # [npar][blocks][2 planes][Nn / 2]
    return buf, cap_blocks


def _colstats_collect(armed, ex, npar: int = 1) -> Optional[ColStats]:
    """What the launch reported in its extras (None when it wrote no records: split-K, whole-tensor enumeration, a kernel off the engine)."""
    if armed is None or ex.colstats_rows_per_block <= 0:
        return None
    return ColStats(armed[0], int(ex.colstats_rows_per_block), npar, armed[1])


def _stats_usable(x: "PF") -> bool:
    st = x.stats
    if st is None:
        return False
    pix = x.H * x.W if st.npar == 1 else (x.H // 2) * (x.W // 2)
    return pix % st.rows == 0 and (st.npar == 1 or (x.H % 2 == 0 and x.W % 2 == 0))


def cached(owner, name: str, params, build):
    """Derived-weight cache on a module attribute, keyed by the source parameters' storage, version counter, device and dtype, so
    that `.to()`, `load_state_dict` or any in-place edit of the weights rebuilds the packed copy."""
    key = tuple((q.data_ptr(), q._version, str(q.device), q.dtype) for q in params)
    c = getattr(owner, name, None)
    if c is None or c[0] != key:
        c = (key, build())
        setattr(owner, name, c)
    return c[1]


class ConvTimer:
    """HIP events around every convolution / matmul-engine launch, on the stream the kernel is launched on (torch's current stream),
    bucketed by the kernel libgswm picks for the shape.  bench.py installs one as `pf.CONV_TIMER` for ONE instrumented step after its
    timed region (the events themselves cost launch slots, so they stay out of the headline)."""

    def __init__(self, by_shape: bool = False):
        self.ev = []
        self.by_shape = by_shape          # bucket by (kernel, B, H, W, K, N) instead of by kernel only (tools/unet_forward_bench.py)

    def start(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def stop(self, e0, kernel: str, flops: float, launches: int = 1, nbytes: float = 0.0):
        """nbytes: the launch's ALGORITHMIC HBM bytes (every operand read once, the output written once, 2 bytes per element)"""
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.ev.append((kernel, flops, e0, e1, launches, nbytes))

    def summary(self):
        out = {}
        for k, f, a, b, nl, nb in self.ev:
            d = out.setdefault(k, {"calls": 0, "flops": 0.0, "ms": 0.0, "bytes": 0.0})
            d["calls"] += nl                     # kernel launches (the sub-pixel upsampling call is four launches)
            d["flops"] += f
            d["bytes"] += nb
            d["ms"] += a.elapsed_time(b)
        for d in out.values():
            d["avg_us"] = d["ms"] * 1e3 / d["calls"]
            d["flops_per_launch"] = d["flops"] / d["calls"]
            d["bytes_per_launch"] = d["bytes"] / d["calls"]
            d["tflops"] = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] else 0.0
        return out


CONV_TIMER: Optional[ConvTimer] = None


def _conv_kernel_name(W: int, n_out: int, ksize: int, stride: int) -> str:
    """Which kernel launch_conv_gemm (csrc/gswm_conv.hip) selects -- for the timer's buckets only."""
    if n_out % 8 == 0 and n_out >= 128 and (stride == 1 or ksize == 3):
        return ("gsw_mm_kernel(conv3x3)" if stride == 1 else "gsw_mm_kernel(conv3x3 s2)") if ksize == 3 else "gsw_mm_kernel(conv1x1)"
    return "gsw_conv_gemm_kernel"


def pack_conv_weight(w: torch.Tensor) -> torch.Tensor:
    """[N, C, kh, kw] -> [N, kh*kw*C] (tap-major, channel-minor), the K order of the PF GEMM."""
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


def conv_pf(x: PF, w_packed: torch.Tensor, bias: Optional[torch.Tensor], *, ksize: int = 3, stride: int = 1,
            rowbias: Optional[torch.Tensor] = None, resid: Optional[PF] = None, cin: Optional[int] = None, cin_offset: int = 0,
            pad_after_only: bool = False, gn_only: bool = False) -> PF:
    """y = conv(x) (+ bias + rowbias[b] + resid) as one MFMA implicit GEMM; border rows of y are zero.
    gn_only: the caller promises that nothing but a GroupNorm reads y (a resnet's conv1 -> norm2).  When the launch wrote that GroupNorm's column records the
    border zeroing is then skipped (`y.border_valid = False`; `groupnorm_pf*` zero it first if they ever have to fall back to the statistics pass).
    pad_after_only (stride 2): the asymmetric F.pad(x, (0, 1, 0, 1)) + padding-0 convolution of the SD VAE downsampler -- in the PF
    layout that is the same tap table read one row and one column further on, i.e. a shifted base pointer."""
    C = x.C if cin is None else cin
    Nn = w_packed.shape[0]
    Ho, Wo = x.H // stride, x.W // stride
    _same(w_packed, x.buf, "conv weight", Nn * ksize * ksize * C)
    _same(bias, x.buf, "conv bias", Nn)
    ldrb = _rowbias_ld(rowbias, x.buf, x.B, Nn)
    if resid is not None:
        _same(resid.buf, x.buf, "resid")
        if (resid.B, resid.H, resid.W, resid.C) != (x.B, Ho, Wo, Nn):
            raise ValueError("resid geometry does not match the convolution output")
    y = PF.empty(x.B, Ho, Wo, Nn, x.buf.dtype, x.buf.device)
    xp = x.rows.data_ptr() + cin_offset * x.buf.element_size()
    if pad_after_only:
        assert stride == 2 and ksize == 3
        xp += (x.W + 2 + 1) * x.C * x.buf.element_size()
    tm = CONV_TIMER
    with torch.cuda.device(x.buf.device):
        e0 = tm.start() if tm is not None else None
        armed = _colstats_arm(x.B * Ho * Wo, Nn, x.buf.device, geom=(x.B, Ho, Wo)) if Nn >= 128 else None
        ex = _extras(x.buf.device, colstats=None if armed is None else armed[0])
        if gn_only and GN_ONLY_SKIPS_BORDER:
            ex.flags = N.GSW_MM_GN_ONLY
        N.check(N.lib().gsw_conv_pf_ex(xp, w_packed.data_ptr(), bias.data_ptr() if bias is not None else None,
                                       rowbias.data_ptr() if rowbias is not None else None, ldrb,
                                       resid.rows.data_ptr() if resid is not None else None, y.rows.data_ptr(),
                                       x.B, Ho, Wo, C, Nn, ksize, stride, x.C, _dt(x.buf.dtype), _C.byref(ex), _stream_ptr()))
        y.stats = _colstats_collect(armed, ex)
        y.border_valid = not (ex.flags & N.GSW_MM_GN_ONLY and ex.colstats_rows_per_block > 0)
        if tm is not None:
            name = _conv_kernel_name(Wo, Nn, ksize, stride)
            tm.stop(e0, (name, x.B, Ho, Wo, ksize * ksize * C, Nn, stride) if tm.by_shape else name, 2.0 * x.B * Ho * Wo * Nn * ksize * ksize * C,
                    nbytes=2.0 * (x.B * x.H * x.W * C + Nn * ksize * ksize * C + x.B * Ho * Wo * Nn * (2 if resid is not None else 1)))
    return y


def _rowbias_ld(rb: Optional[torch.Tensor], like: torch.Tensor, B: int, Nn: int) -> int:
    """Per-image row bias [B, N]: contiguous, or a column slice of a wider row-major matrix (the time-embedding projections of every resnet
    come out of ONE GEMM).  Returns the row stride in elements (0 when there is no row bias)."""
    if rb is None:
        return 0
    if not rb.is_cuda or rb.device != like.device or rb.dtype != like.dtype:
        raise ValueError(f"rowbias must be a {like.dtype} tensor on {like.device}")
    if tuple(rb.shape) != (B, Nn) or rb.stride(1) != 1 or rb.data_ptr() % 16 or (B > 1 and (rb.stride(0) % 8 or rb.stride(0) < Nn)):
        raise ValueError(f"rowbias must be [{B}, {Nn}] with contiguous, 16-byte aligned rows a multiple of 8 elements apart")
    return rb.stride(0) if B > 1 else Nn


def _rows2d(t: torch.Tensor, like: torch.Tensor, name: str):
    """A 2-D operand that may be a column slice of a wider matrix: unit stride along the row, row stride a multiple of 8, 16-byte aligned."""
    if t.dim() != 2 or not t.is_cuda or t.device != like.device or t.dtype != like.dtype:
        raise ValueError(f"{name} must be a 2-D {like.dtype} tensor on {like.device}")
    if t.stride(1) != 1 or t.stride(0) % 8 or t.stride(0) < t.shape[1] or t.data_ptr() % 16:
        raise ValueError(f"{name}: rows must be contiguous, 16-byte aligned and a multiple of 8 elements apart")
    return t.stride(0)


def gemm_strided(x: torch.Tensor, w: torch.Tensor, out: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M, N] = x[M, K] @ w[N, K]^T (+ bias) on the matmul engine, every operand addressed through its own row stride (column slices of
    wider matrices are fine).  K % 64 == 0, N % 8 == 0."""
    if x.dtype not in (torch.float16, torch.bfloat16):
        raise ValueError(f"gemm_strided: fp16 / bf16 only (got {x.dtype})")
    ldx, ldw, ldy = _rows2d(x, x, "x"), _rows2d(w, x, "w"), _rows2d(out, x, "out")
    M, K = x.shape
    Nn = w.shape[0]
    if w.shape[1] != K or tuple(out.shape) != (M, Nn):
        raise ValueError("gemm_strided: shapes do not chain")
    _same(bias, x, "bias", Nn)
    tm = CONV_TIMER
    with torch.cuda.device(x.device):
        e0 = tm.start() if tm is not None else None
        ex = _extras(x.device)
        N.check(N.lib().gsw_gemm_ex(x.data_ptr(), ldx, w.data_ptr(), ldw, bias.data_ptr() if bias is not None else None, None, ldy,
                                    out.data_ptr(), ldy, M, K, Nn, 0, 0, 0, _dt(x.dtype), _C.byref(ex), _stream_ptr()))
        if tm is not None:
            tm.stop(e0, ("gsw_mm_kernel", M, K, Nn, "plain") if tm.by_shape else "gsw_mm_kernel", 2.0 * M * K * Nn, nbytes=2.0 * (M * K + Nn * K + M * Nn))
    return out


def softmax_rows_(x: torch.Tensor, scale: float) -> torch.Tensor:
    """In place: every row of the 2-D tensor x becomes softmax(scale * row) (gsw_softmax_rows)."""
    ld = _rows2d(x, x, "x")
    with torch.cuda.device(x.device):
        N.check(N.lib().gsw_softmax_rows(x.data_ptr(), x.shape[0], x.shape[1], ld, float(scale), _dt(x.dtype), _stream_ptr()))
    return x


def attention_single_head(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor) -> torch.Tensor:
    """softmax(q k^T / sqrt(d)) v for ONE head of any width d % 64 == 0 (the VAE mid block: d = 512), as two matmul-engine products per image
    with the row softmax kernel in between.  q, k: [B, S, d] (column slices of a fused projection are fine), vt: [B, d, S] (V transposed,
    what gemm(mode="trans") produces) -> [B, S, d].  S % 8 == 0; when S is not a multiple of 64 (the second product's K dimension) the key
    axis is zero-padded: zero probabilities times zero values."""
    B, S, d = q.shape
    if k.shape != (B, S, d) or vt.shape != (B, d, S) or d % 64 or S % 8:
        raise ValueError("attention_single_head: q, k [B, S, d], vt [B, d, S], d % 64 == 0, S % 8 == 0")
    Sp = (S + 63) // 64 * 64
    out = torch.empty((B, S, d), dtype=q.dtype, device=q.device)
    scores = torch.empty((S, Sp), dtype=q.dtype, device=q.device)         # one image at a time: S x S fp16 (32 MiB at 512x512) is reused
    if Sp != S:
        scores[:, S:].zero_()
        vt = torch.cat([vt, vt.new_zeros(B, d, Sp - S)], dim=2)
    for b in range(B):
        gemm_strided(q[b], k[b], scores[:, :S])
        softmax_rows_(scores[:, :S], d ** -0.5)
        gemm_strided(scores, vt[b], out[b])
    return out


_GN_WS = {}


def _gn_workspace(device, B, groups, C=0):
    k = (str(device), max(B * 64 * groups * 2, B * C * 2))        # slab records of the statistics pass, or per-column sums (gsw_groupnorm_pf_cs)
    if k not in _GN_WS:
        _GN_WS[k] = torch.empty(k[1], dtype=torch.float32, device=device)
    return _GN_WS[k]


# One-launch GroupNorm for small batches (gsw_groupnorm_pf_fused: a workgroup per (image, group) keeps the group in registers): used while the grid stays
# a few hundred workgroups -- beyond that the two coalesced passes (or the producer's column records) win
GN_FUSED_MAX_WGS = int(__import__("os").environ.get("GSW_GN_FUSED_MAX_WGS", "512"))
GN_FUSED_MAX_PIXELS = int(__import__("os").environ.get("GSW_GN_FUSED_MAX_PIXELS", "1024"))      # measured on one image: 32 x 32 and below 5.5-10 us against ~13.5 for the
                                                                                                 # two launches; 64 x 64 22 us (4-byte accesses 640 bytes apart)


def _gn_fused_ok(B: int, H: int, W: int, C: int, groups: int) -> bool:
    if B * groups > GN_FUSED_MAX_WGS or H * W > GN_FUSED_MAX_PIXELS or C % groups or (C // groups) % 2 or C % 8:
        return False
    npair = C // groups // 2
    if npair * W > 1024 or B * (H + 2) * (W + 2) * C >= 1 << 31:
        return False
    rl = min(1024 // (npair * W), H)
    while H % rl:
        rl -= 1
    return (H + rl - 1) // rl <= 32


def _groupnorm_fused(x: PF, x2: Optional[PF], gamma, beta, groups, eps, act, tokens):
    dev = x.buf.device
    C = x.C + (x2.C if x2 is not None else 0)
    if tokens:
        res = torch.empty((x.B, x.H * x.W, C), dtype=x.buf.dtype, device=dev)
        optr = res.data_ptr()
    else:
        res = PF.empty(x.B, x.H, x.W, C, x.buf.dtype, dev)
        optr = res.rows.data_ptr()
    with torch.cuda.device(dev):
        N.check(N.lib().gsw_groupnorm_pf_fused(x.rows.data_ptr(), x2.rows.data_ptr() if x2 is not None else None, x.C if x2 is not None else 0,
                                               gamma.data_ptr(), beta.data_ptr(), optr, x.B, x.H, x.W, C, groups, eps, 1 if act else 0,
                                               1 if tokens else 0, _dt(x.buf.dtype), _stream_ptr()))
    return res


def _ensure_border(x: PF) -> None:
    """A tensor whose border was left unwritten is about to be read by a kernel that looks at the border (the statistics pass): write the zeros now."""
    if not x.border_valid:
        g = x.grid
        g[:, 0].zero_(); g[:, -1].zero_(); g[:, :, 0].zero_(); g[:, :, -1].zero_()
        x.border_valid = True


def groupnorm_pf(x: PF, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float, *, act: bool = True, tokens: bool = False):
    """act(GroupNorm(x)) on a PF tensor -> PF (zero border) or dense tokens [B, H*W, C] (tokens=True)."""
    dev = x.buf.device
    _same(gamma, x.buf, "gamma", x.C)
    _same(beta, x.buf, "beta", x.C)
    if _gn_fused_ok(x.B, x.H, x.W, x.C, groups):
        _ensure_border(x)
        return _groupnorm_fused(x, None, gamma, beta, groups, eps, act, tokens)
    ws = _gn_workspace(dev, x.B, groups, x.C)
    if tokens:
        out = torch.empty((x.B, x.H * x.W, x.C), dtype=x.buf.dtype, device=dev)
        optr, res = out.data_ptr(), out
    else:
        y = PF.empty(x.B, x.H, x.W, x.C, x.buf.dtype, dev)
        optr, res = y.rows.data_ptr(), y
    if FUSE_GN_STATS and _stats_usable(x) and x.C <= 4096 and (x.C // groups) % 2 == 0:
        st = x.stats
        with torch.cuda.device(dev):
            N.check(N.lib().gsw_groupnorm_pf_cs(x.rows.data_ptr(), None, 0, st.buf.data_ptr(), st.rows, st.npar, st.blocks, None, 0, 0, 0,
                                                gamma.data_ptr(), beta.data_ptr(), optr, ws.data_ptr(), x.B, x.H, x.W, x.C, groups, eps,
                                                1 if act else 0, 1 if tokens else 0, _dt(x.buf.dtype), _stream_ptr()))
        return res
    _ensure_border(x)
    with torch.cuda.device(dev):
        N.check(N.lib().gsw_groupnorm_pf(x.rows.data_ptr(), gamma.data_ptr(), beta.data_ptr(), optr, ws.data_ptr(), x.B, x.H, x.W, x.C, groups,
                                         eps, 1 if act else 0, 1 if tokens else 0, _dt(x.buf.dtype), _stream_ptr()))
    return res


def pack_geglu_weight(w: torch.Tensor, b: Optional[torch.Tensor]):
    """GEGLU projection [2I, K] (rows: I value then I gate) -> rows interleaved per 16-row block as [8 value | 8 gate] of the same 8
    outputs: in the MFMA accumulator layout value and gate of one output then sit 32 lanes apart in the same register (the engine's
    GEGLU epilogue pairs them with v_permlane32_swap).  I % 80 == 0 (a 160-row tile yields 80 outputs)."""
    I = w.shape[0] // 2
    v = w[:I].reshape(I // 8, 8, -1)
    g = w[I:].reshape(I // 8, 8, -1)
    wp = torch.cat([v, g], dim=1).reshape(2 * I, -1).contiguous()
    bp = None
    if b is not None:
        bp = torch.cat([b[:I].reshape(I // 8, 8), b[I:].reshape(I // 8, 8)], dim=1).reshape(2 * I).contiguous()
    return wp, bp


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], *, resid: Optional[torch.Tensor] = None, geglu: bool = False) -> torch.Tensor:
    """x[..., K] @ w[N, K]^T + bias (+ resid) on the matmul engine; geglu=True expects pack_geglu_weight operands.  (The round-1 name of `gemm`.)"""
    return gemm(x, w, bias, resid=resid, mode="geglu" if geglu else "plain")


GEMM_MODES = {"plain": 0, "geglu": 1, "trans": 2, "tok2pf": 3}
SMALL_GEMM_MAX_ROWS = int(__import__("os").environ.get("GSW_SMALL_GEMM_MAX_ROWS", "128"))      # dense linears of at most that many rows run on gsw_gemm_small (0: never)
SMALL_GEMM_MAX_K = 2560


def _same(t: Optional[torch.Tensor], like: torch.Tensor, name: str, numel: Optional[int] = None):
    """A companion operand whose raw pointer crosses the C ABI: same device and dtype as `like`, contiguous, 16-byte aligned."""
    if t is None:
        return
    if not t.is_cuda or t.device != like.device:
        raise RuntimeError(f"{name} must live on {like.device} (got {t.device}); there is no CPU fallback")
    if t.dtype != like.dtype:
        raise ValueError(f"{name} must be {like.dtype} like the activations (got {t.dtype})")
    if not t.is_contiguous() or t.data_ptr() % 16:
        raise ValueError(f"{name} must be contiguous and 16-byte aligned")
    if numel is not None and t.numel() != numel:
        raise ValueError(f"{name} has {t.numel()} elements, expected {numel}")


def gemm(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, resid: Optional[torch.Tensor] = None, mode: str = "plain",
         tokens: int = 0, width: int = 0, out: Optional[torch.Tensor] = None, stats_for: Optional["PF"] = None,
         rowstats: bool = False) -> torch.Tensor:
    """x[..., K] @ w[N, K]^T + bias on the matmul engine (csrc/gswm_mm.hip).
    mode "plain":  -> [..., N] (+ resid[..., N]);  "geglu": pf.pack_geglu_weight operands -> [..., N/2] = value * gelu(gate);
    "trans": x is [B, S, K] -> [B, N, S] (tokens = S);  "tok2pf": rows are tokens of `tokens`-pixel images of width `width`, `out` is the
    `.rows` view of a PF tensor [B, H, W, N]: its interior rows receive x w^T + bias (+ resid, which may be `out` itself)."""
    if x.dtype not in (torch.float16, torch.bfloat16):
        raise ValueError(f"gemm: fp16 / bf16 only (got {x.dtype})")
    _same(x, x, "x")
    K = x.shape[-1]
    M = x.numel() // K
    Nn = w.shape[0]
    _same(w, x, "w", Nn * K)
    _same(bias, x, "bias", Nn)
    m = GEMM_MODES[mode]
    if mode == "plain":
        _same(resid, x, "resid", M * Nn)
        y = torch.empty((*x.shape[:-1], Nn), dtype=x.dtype, device=x.device) if out is None else out
        _same(y, x, "out", M * Nn)
    elif mode == "geglu":
        if resid is not None:
            raise ValueError("gemm: geglu takes no residual")
        y = torch.empty((*x.shape[:-1], Nn // 2), dtype=x.dtype, device=x.device) if out is None else out
        _same(y, x, "out", M * Nn // 2)
    elif mode == "trans":
        if resid is not None or tokens <= 0 or M % tokens:
            raise ValueError("gemm: trans needs tokens = rows per image and no residual")
        y = torch.empty((M // tokens, Nn, tokens), dtype=x.dtype, device=x.device) if out is None else out
        _same(y, x, "out", M * Nn)
    else:
        if out is None or tokens <= 0 or width <= 0 or tokens % width or M % tokens:
            raise ValueError("gemm: tok2pf needs out (PF rows), tokens = H*W and width = W")
        y = out
        rows = (M // tokens) * (tokens // width + 2) * (width + 2)
        _same(y, x, "out", rows * Nn)
        _same(resid, x, "resid", rows * Nn)
    tm = CONV_TIMER
    if (SMALL_GEMM_MAX_ROWS and M <= SMALL_GEMM_MAX_ROWS and mode in ("plain", "trans", "tok2pf") and K <= SMALL_GEMM_MAX_K and M % 16 == 0 and Nn % 16 == 0
            and K % 32 == 0 and K >= 128 and (mode != "trans" or tokens % 4 == 0)):
        # one or two images' 8 x 8 level (64 / 128 token rows): the small-M kernel (K split four ways inside the workgroup, one launch) beats the engine's
        # split-K pair there (7.5 vs 11.2 us, profiles/r04c_small_m_gemm_vs_engine.txt); above that the engine wins
        rs_buf = torch.empty(M * ((Nn + 31) // 32) * 2, dtype=torch.float32, device=x.device) if (rowstats and mode == "plain" and FOLD_LN and M >= FOLD_LN_MIN_ROWS and M % 8 == 0) else None
        slots = _C.c_int(0)
        with torch.cuda.device(x.device):
            e0 = tm.start() if tm is not None else None
            N.check(N.lib().gsw_gemm_small(x.data_ptr(), K, w.data_ptr(), K, bias.data_ptr() if bias is not None else None,
                                           resid.data_ptr() if resid is not None else None, Nn, y.data_ptr(), Nn, M, K, Nn, m, tokens, width, None, 0, 0.0, None, None,
                                           rs_buf.data_ptr() if rs_buf is not None else None, rs_buf.numel() if rs_buf is not None else 0, _C.byref(slots), -1,
                                           _dt(x.dtype), _stream_ptr()))
            if tm is not None:
                tm.stop(e0, ("gsw_mm_small_kernel", M, K, Nn, mode) if tm.by_shape else "gsw_mm_small_kernel", 2.0 * M * K * Nn, nbytes=2.0 * (M * K + Nn * K + M * Nn * (2 if resid is not None else 1)))
        if stats_for is not None:
            stats_for.stats = None
        if rs_buf is not None and slots.value > 0:
            y._gsw_rowstats = (rs_buf, int(slots.value))
        return y
    with torch.cuda.device(x.device):
        e0 = tm.start() if tm is not None else None
        # stats_for (tok2pf): the PF tensor whose payload this launch writes -- it gets the launch's column records (or None)
        armed = _colstats_arm(M, Nn, x.device, geom=(M // tokens, tokens // width, width)) if (stats_for is not None and mode == "tok2pf") else None
        rs_buf = None
        if rowstats and mode == "plain" and FOLD_LN and M >= FOLD_LN_MIN_ROWS and M % 8 == 0:      # (below that ln_stat never folds: no records needed)
            # row records for the LayerNorm that consumes this output (GswMmExtras.rowstats_dev): [M][2 * ceil(N / 160)][2] floats
            rs_buf = torch.empty(M * 2 * ((Nn + 159) // 160) * 2, dtype=torch.float32, device=x.device)
        ex = _extras(x.device, colstats=None if armed is None else armed[0], rowstats=rs_buf)
        ncols = Nn // 2 if mode == "geglu" else Nn
        N.check(N.lib().gsw_gemm_ex(x.data_ptr(), K, w.data_ptr(), K, bias.data_ptr() if bias is not None else None,
                                    resid.data_ptr() if resid is not None else None, Nn, y.data_ptr(), ncols, M, K, Nn, m, tokens, width,
                                    _dt(x.dtype), _C.byref(ex), _stream_ptr()))
        if stats_for is not None:
            stats_for.stats = _colstats_collect(armed, ex)
        if rs_buf is not None and ex.rowstats_slots > 0:
            y._gsw_rowstats = (rs_buf, int(ex.rowstats_slots))          # rides on the output tensor; in-place edits of y must drop it
        if tm is not None:
            tm.stop(e0, ("gsw_mm_kernel", M, K, Nn, mode + ("+res" if resid is not None and mode == "plain" else "")) if tm.by_shape else "gsw_mm_kernel", 2.0 * M * K * Nn,
                    nbytes=2.0 * (M * K + Nn * K + M * ncols * (2 if resid is not None else 1)))
    return y


# ---- LayerNorm folded into the consuming GEMM (gsw_gemm_ln): LN(x) W^T + b = rstd (x W'^T) + nrm u + v, W' = W diag(gamma), u = W' 1, v = W beta + b
FOLD_LN = True
FOLD_LN_MIN_ROWS = int(__import__("os").environ.get("GSW_FOLD_LN_MIN_ROWS", "1024"))       # below that the consumers would rather take the split-K form, which the folded epilogue does not have


def ln_stat(x: torch.Tensor, eps: float) -> Optional[torch.Tensor]:
    """(rstd, -rstd * mean) per row of x [.., C] from the row records its producer left on it, or None when there are none."""
    done = getattr(x, "_gsw_lnstat", None)      # the producer already left the finished statistics (xattn.fused: the whole row is in one wave there)
    if done is not None and FOLD_LN and done[1] == float(eps):
        return done[0]
    rs = getattr(x, "_gsw_rowstats", None)
    if rs is None or not FOLD_LN:
        return None
    C = x.shape[-1]
    M = x.numel() // C
    if M % 8 or M < FOLD_LN_MIN_ROWS:
        return None
    out = torch.empty((M, 2), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        N.check(N.lib().gsw_ln_rowstats_finish(rs[0].data_ptr(), rs[1], M, C, float(eps), out.data_ptr(), _stream_ptr()))
    return out


def fold_ln_weights(w: torch.Tensor, b: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor, geglu: bool = False):
    """(W' = W diag(gamma) in the activations' dtype, u = row sums of the ROUNDED W' (fp32), v = W beta + b (fp32)); geglu: packed like pack_geglu_weight."""
    wf = w.detach().float()
    wp = (wf * gamma.detach().float()[None, :]).to(w.dtype)
    u = wp.float().sum(dim=1)
    v = wf @ beta.detach().float()
    if b is not None:
        v = v + b.detach().float()
    if geglu:
        wp, u = pack_geglu_weight(wp, u)
        _, v = pack_geglu_weight(wp, v)
    return wp.contiguous(), u.contiguous(), v.contiguous()


def gemm_ln(x: torch.Tensor, stat: torch.Tensor, wp: torch.Tensor, u: torch.Tensor, v: torch.Tensor, *, mode: str = "plain", tokens: int = 0) -> torch.Tensor:
    """LayerNorm(x) W^T + b on the matmul engine without the normalised tensor (gsw_gemm_ln); operands from fold_ln_weights / ln_stat."""
    K = x.shape[-1]
    M = x.numel() // K
    Nn = wp.shape[0]
    _same(x, x, "x")
    _same(wp, x, "w", Nn * K)
    for t_, nm in ((stat, "stat"), (u, "u"), (v, "v")):
        if t_.dtype != torch.float32 or t_.device != x.device or not t_.is_contiguous() or t_.data_ptr() % 16:
            raise ValueError(f"gemm_ln: {nm} must be a contiguous, 16-byte aligned fp32 tensor on {x.device}")
    if stat.numel() != 2 * M or u.numel() != Nn or v.numel() != Nn:
        raise ValueError("gemm_ln: stat [M, 2], u [N], v [N]")
    if mode == "plain":
        y = torch.empty((*x.shape[:-1], Nn), dtype=x.dtype, device=x.device)
    elif mode == "geglu":
        y = torch.empty((*x.shape[:-1], Nn // 2), dtype=x.dtype, device=x.device)
    elif mode == "trans":
        if tokens <= 0 or M % tokens:
            raise ValueError("gemm_ln: trans needs tokens = rows per image")
        y = torch.empty((M // tokens, Nn, tokens), dtype=x.dtype, device=x.device)
    else:
        raise ValueError(mode)
    tm = CONV_TIMER
    with torch.cuda.device(x.device):
        e0 = tm.start() if tm is not None else None
        ex = _extras(x.device)
        N.check(N.lib().gsw_gemm_ln_ex(x.data_ptr(), stat.data_ptr(), wp.data_ptr(), u.data_ptr(), v.data_ptr(), y.data_ptr(), M, K, Nn,
                                       GEMM_MODES[mode], tokens, _dt(x.dtype), _C.byref(ex), _stream_ptr()))
        if tm is not None:
            tm.stop(e0, ("gsw_mm_kernel", M, K, Nn, mode + "+ln") if tm.by_shape else "gsw_mm_kernel", 2.0 * M * K * Nn, nbytes=2.0 * (M * K + Nn * K) + y.numel() * 2.0 + M * 8.0)
    return y


def gemm_qkv(x: torch.Tensor, w: torch.Tensor, n_rows: int, bias: Optional[torch.Tensor] = None):
    """Self-attention's three projections from ONE pass over the tokens (gsw_gemm_qkv): x [B, S, K], w [N, K] = [to_q | to_k | to_v] rows ->
    (rows [B, S, n_rows] = q | k as the attention kernel reads them, vt [B, N - n_rows, S] = V transposed).  n_rows % 160 == 0, S % 8 == 0."""
    if x.dtype not in (torch.float16, torch.bfloat16) or x.dim() != 3:
        raise ValueError("gemm_qkv: fp16 / bf16 tokens [B, S, K]")
    B, S, K = x.shape
    Nn = w.shape[0]
    _same(x, x, "x")
    _same(w, x, "w", Nn * K)
    _same(bias, x, "bias", Nn)
    if n_rows <= 0 or n_rows >= Nn or n_rows % 160 or S % 8:
        raise ValueError("gemm_qkv: n_rows must be a multiple of 160 inside (0, N), S a multiple of 8")
    rows = torch.empty((B, S, n_rows), dtype=x.dtype, device=x.device)
    vt = torch.empty((B, Nn - n_rows, S), dtype=x.dtype, device=x.device)
    tm = CONV_TIMER
    with torch.cuda.device(x.device):
        e0 = tm.start() if tm is not None else None
        N.check(N.lib().gsw_gemm_qkv(x.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, rows.data_ptr(), vt.data_ptr(),
                                     B * S, K, n_rows, Nn, S, _dt(x.dtype), _stream_ptr()))
        if tm is not None:
            tm.stop(e0, ("gsw_mm_kernel", B * S, K, Nn, "qkv") if tm.by_shape else "gsw_mm_kernel", 2.0 * B * S * K * Nn, nbytes=2.0 * (B * S * K + Nn * K + B * S * Nn))
    return rows, vt


def groupnorm_pf2(x: PF, x2: Optional[PF], gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float, *, act: bool = True) -> PF:
    """act(GroupNorm(cat([x, x2], channels))) -> one PF tensor, without materialising the concatenation."""
    if x2 is None:
        return groupnorm_pf(x, gamma, beta, groups, eps, act=act)
    dev = x.buf.device
    C = x.C + x2.C
    _same(x2.buf, x.buf, "x2")
    _same(gamma, x.buf, "gamma", C)
    _same(beta, x.buf, "beta", C)
    if (x2.B, x2.H, x2.W) != (x.B, x.H, x.W):
        raise ValueError("groupnorm_pf2: the two sources differ in geometry")
    if x.C % 8 == 0 and _gn_fused_ok(x.B, x.H, x.W, C, groups):
        _ensure_border(x); _ensure_border(x2)
        return _groupnorm_fused(x, x2, gamma, beta, groups, eps, act, False)
    ws = _gn_workspace(dev, x.B, groups, C)
    y = PF.empty(x.B, x.H, x.W, C, x.buf.dtype, dev)
    if FUSE_GN_STATS and _stats_usable(x) and _stats_usable(x2) and C <= 4096 and (C // groups) % 2 == 0 and x.C % 2 == 0:
        s1, s2 = x.stats, x2.stats
        with torch.cuda.device(dev):
            N.check(N.lib().gsw_groupnorm_pf_cs(x.rows.data_ptr(), x2.rows.data_ptr(), x.C, s1.buf.data_ptr(), s1.rows, s1.npar, s1.blocks,
                                                s2.buf.data_ptr(), s2.rows, s2.npar, s2.blocks, gamma.data_ptr(), beta.data_ptr(), y.rows.data_ptr(),
                                                ws.data_ptr(), x.B, x.H, x.W, C, groups, eps, 1 if act else 0, 0, _dt(x.buf.dtype), _stream_ptr()))
        return y
    _ensure_border(x); _ensure_border(x2)
    with torch.cuda.device(dev):
        N.check(N.lib().gsw_groupnorm_pf2(x.rows.data_ptr(), x2.rows.data_ptr(), x.C, gamma.data_ptr(), beta.data_ptr(), y.rows.data_ptr(),
                                          ws.data_ptr(), x.B, x.H, x.W, C, groups, eps, 1 if act else 0, 0, _dt(x.buf.dtype), _stream_ptr()))
    return y


def conv3x3_res_fusable(x: PF, n_out: int) -> bool:
    """The fused conv2 + shortcut launch runs on the matmul engine: N % 8 == 0 from 128 channels up, C % 64 == 0."""
    return n_out % 8 == 0 and n_out >= 128 and x.C % 64 == 0


def conv3x3_res_pf(x: PF, w_cat: torch.Tensor, bias: Optional[torch.Tensor], *, rowbias: Optional[torch.Tensor] = None,
                   resid: Optional[PF] = None, x1: Optional[PF] = None, x2: Optional[PF] = None) -> PF:
    """y = conv3x3(x) + conv1x1(cat([x1, x2])) + bias (+ rowbias + resid) in one GEMM; w_cat = [N, 9*C | C1 | C2]."""
    Nn = w_cat.shape[0]
    _same(w_cat, x.buf, "w_cat", Nn * (9 * x.C + (x1.C if x1 is not None else 0) + (x2.C if x2 is not None else 0)))
    _same(bias, x.buf, "bias", Nn)
    ldrb = _rowbias_ld(rowbias, x.buf, x.B, Nn)
    for t_, nm in ((resid, "resid"), (x1, "x1"), (x2, "x2")):
        if t_ is not None:
            _same(t_.buf, x.buf, nm)
    y = PF.empty(x.B, x.H, x.W, Nn, x.buf.dtype, x.buf.device)
    tm = CONV_TIMER
    with torch.cuda.device(x.buf.device):
        e0 = tm.start() if tm is not None else None
        armed = _colstats_arm(x.B * x.H * x.W, Nn, x.buf.device, geom=(x.B, x.H, x.W))
        ex = _extras(x.buf.device, colstats=None if armed is None else armed[0])
        N.check(N.lib().gsw_conv3x3_res_pf_ex(x.rows.data_ptr(), w_cat.data_ptr(), bias.data_ptr() if bias is not None else None,
                                              rowbias.data_ptr() if rowbias is not None else None, ldrb,
                                              resid.rows.data_ptr() if resid is not None else None, y.rows.data_ptr(),
                                              x.B, x.H, x.W, x.C, Nn,
                                              x1.rows.data_ptr() if x1 is not None else None, x1.C if x1 is not None else 0,
                                              x2.rows.data_ptr() if x2 is not None else None, x2.C if x2 is not None else 0,
                                              _dt(x.buf.dtype), _C.byref(ex), _stream_ptr()))
        y.stats = _colstats_collect(armed, ex)
        if tm is not None:
            k = 9 * x.C + (x1.C if x1 is not None else 0) + (x2.C if x2 is not None else 0)
            name = _conv_kernel_name(x.W, Nn, 3, 1)
            pix = x.B * x.H * x.W
            tm.stop(e0, (name, x.B, x.H, x.W, k, Nn, 1) if tm.by_shape else name, 2.0 * pix * Nn * k,
                    nbytes=2.0 * (pix * (k - 8 * x.C) + Nn * k + pix * Nn * (2 if resid is not None else 1)))
    return y


ATTN_HEAD_DIMS = (40, 64, 80, 160)


def attention_ok(x: torch.Tensor, heads: int, head_dim: int, n_q: int, n_k: int) -> bool:
    return x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and head_dim in ATTN_HEAD_DIMS and n_q >= 1 and n_k % 8 == 0


attention_hd64_ok = attention_ok


def dup_pf(x: "PF") -> "PF":
    """[x | x] along the batch: the B images of a PF tensor as a PF tensor of 2B images (borders and payload copied, fresh guard rows; the column
    records of x describe each half)."""
    y = PF.empty(2 * x.B, x.H, x.W, x.C, x.buf.dtype, x.buf.device)
    r = y.rows
    r[: x.M].copy_(x.rows)
    r[x.M:].copy_(x.rows)
    return y


def attention(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, heads: int, scale: Optional[float] = None,
              valid_keys: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """softmax(q k^T * scale) v on the hand-written flash-attention kernel (csrc/gswm_attn.hip), head_dim 40 / 64 / 80 / 160, any
    number of queries, keys a multiple of 8 (sequences off the 128-query / 64-key tiles run the kernel's ragged variant).
    q [B, Sq, heads*d], k [B, Sk, heads*d], vt [B, heads*d, Sk] (V transposed) -> [B, Sq, heads*d]; keys >= valid_keys are
    padding (zero weight).  q and k may be column slices of a wider row-major tensor (a fused QK projection): only their last
    dimension has to be contiguous and the batch stride has to equal rows * row stride."""
    B, Sq, inner = q.shape
    Sk = k.shape[1]
    d = inner // heads
    assert inner == heads * d and vt.shape == (B, inner, Sk) and k.shape[2] == inner

    def rows(t):          # (tensor, row stride) of a [B, S, inner] operand the kernel can address directly
        if t.stride(2) == 1 and t.stride(0) == t.shape[1] * t.stride(1) and t.stride(1) % 8 == 0 and t.data_ptr() % 16 == 0:
            return t, t.stride(1)
        t = t.contiguous()
        return t, inner

    (q, ldq), (k, ldk) = rows(q), rows(k)
    vt = vt.contiguous()
    for t_, nm in ((k, "k"), (vt, "vt")):
        if t_.dtype != q.dtype or t_.device != q.device:
            raise ValueError(f"attention: {nm} is {t_.dtype} on {t_.device}, q is {q.dtype} on {q.device}")
    if out is None:
        out = torch.empty((B, Sq, inner), dtype=q.dtype, device=q.device)
    elif tuple(out.shape) != (B, Sq, inner) or out.dtype != q.dtype or out.device != q.device or not out.is_contiguous():
        raise ValueError("attention: out must be a contiguous [B, Sq, heads * d] tensor like q")
    tm = CONV_TIMER
    with torch.cuda.device(q.device):
        e0 = tm.start() if tm is not None else None
        # the split-K scratch of this stream doubles as the key-split scratch (few query tiles against many key tiles: one image's self-attention); both are
        # free between two launches of a stream
        # (only for launches the key-split form can take -- whole 128-query / 64-key tiles, >= 32 key tiles, at most 256 query tiles, no padded keys: a
        # 77-key cross-attention or a ragged launch must not pin a scratch buffer on its stream)
        ks_ok = (ATTN_KEY_SPLIT and d in (40, 64) and Sq % 128 == 0 and Sk % 64 == 0 and Sk // 64 >= 32 and (valid_keys is None or int(valid_keys) == Sk)
                 and (Sq // 128) * B * heads <= 256)
        ws = _workspace(q.device) if ks_ok else None
        N.check(N.lib().gsw_attention_ws(q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), B, heads, d, Sq, Sk,
                                         Sk if valid_keys is None else int(valid_keys), ldq, ldk, inner,
                                         float(scale if scale is not None else d ** -0.5), _dt(q.dtype), 0 if ws is None else ws.data_ptr(),
                                         0 if ws is None else ws.numel(), _stream_ptr()))
        if tm is not None:
            tm.stop(e0, ("gsw_attn_fwd_kernel", B, Sq, Sk, heads, d) if tm.by_shape else "gsw_attn_fwd_kernel", 4.0 * B * heads * Sq * (Sk if valid_keys is None else int(valid_keys)) * d,
                    nbytes=2.0 * heads * d * (2 * out.shape[0] * Sq + 2 * B * Sk))
    return out


attention_hd64 = attention


def pack_upsample_weight(w: torch.Tensor) -> torch.Tensor:
    """3x3 weights [N, C, 3, 3] of a convolution that follows a nearest-neighbour 2x upsampling -> [4, N, 4*C]: for output parity
    (dy, dx) the 2x2 kernel over the low-resolution input, tap (a, b) = sum of the 3x3 taps that read source pixel (i+a+dy-1, j+b+dx-1)."""
    rows = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}          # (parity, tap) -> 3x3 indices merged into it
    wf = w.detach().float()
    out = []
    for dy in (0, 1):
        for dx in (0, 1):
            taps = []
            for a in (0, 1):
                for b in (0, 1):
                    taps.append(sum(wf[:, :, kh, kw] for kh in rows[(dy, a)] for kw in rows[(dx, b)]))       # [N, C]
            out.append(torch.stack(taps, dim=1).reshape(w.shape[0], -1))                                     # [N, 4*C]
    return torch.stack(out).to(w.dtype).contiguous()


def conv_up2x_fusable(x: PF, n_out: int) -> bool:
    return n_out % 8 == 0 and n_out >= 128 and x.C % 64 == 0


def conv_up2x_pf(x: PF, w4: torch.Tensor, bias: Optional[torch.Tensor]) -> PF:
    """conv3x3(nearest_upsample_2x(x)) without materialising the upsampled tensor (gsw_conv_up2x_pf)."""
    Nn = w4.shape[1]
    _same(w4, x.buf, "w4", 4 * Nn * 4 * x.C)
    _same(bias, x.buf, "bias", Nn)
    y = PF.empty(x.B, 2 * x.H, 2 * x.W, Nn, x.buf.dtype, x.buf.device)          # gsw_conv_up2x_pf zeroes the border rows itself
    tm = CONV_TIMER
    with torch.cuda.device(x.buf.device):
        e0 = tm.start() if tm is not None else None
        armed = _colstats_arm(x.B * x.H * x.W, Nn, x.buf.device, npar=4, geom=(x.B, 2 * x.H, 2 * x.W))
        ex = _extras(x.buf.device, colstats=None if armed is None else armed[0])
        N.check(N.lib().gsw_conv_up2x_pf_ex(x.rows.data_ptr(), w4.data_ptr(), bias.data_ptr() if bias is not None else None, y.rows.data_ptr(),
                                            x.B, x.H, x.W, x.C, Nn, _dt(x.buf.dtype), _C.byref(ex), _stream_ptr()))
        y.stats = _colstats_collect(armed, ex, npar=4)
        if tm is not None:      # EXECUTED FLOPs (16 C MACs per output: four 2x2 convolutions); the 3x3-on-upsampled form it replaces is 2.25x that
            name = "gsw_mm_kernel(up2x)"
            tm.stop(e0, (name, x.B, 2 * x.H, 2 * x.W, 4 * x.C, Nn, 1) if tm.by_shape else name,
                    2.0 * x.B * 4 * x.H * x.W * Nn * 4 * x.C, launches=4, nbytes=2.0 * (4 * x.B * x.H * x.W * x.C + 16 * Nn * x.C + 4 * x.B * x.H * x.W * Nn))
    return y
