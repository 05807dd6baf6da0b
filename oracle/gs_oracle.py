"""CPU ORACLE -- test infrastructure, NOT product code.

A NumPy restatement of the reference's Gaussian-Shading codec (lthero-big/A-watermark-for-Diffusion-Models
@ 2024_08_07).  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this module; the product package (`a-watermark-for-diffusion-models_amd/`) never does and fails loudly
when its HIP library is missing.

Pinning status
  * codec (ChaCha20 stream, bit expansion, ndtri sampling, cdf quantise, majority vote, bit accuracy):
    PINNED against golden vectors produced by importing the reference (tests/golden/make_golden.py,
    run under /opt/conda/bin/python3.9 with the real `cryptography`/OpenSSL, scipy, numpy) and against
    the RFC 8439 known-answer vectors.
  * DDIM step / loops (`backward_ddim`, `ddim_coefficients`, `ddim_schedule` below): PINNED against vectors
    produced by EXECUTING the reference's own shipped bytecode
    (__pycache__/inverse_stable_diffusion_gs.cpython-38.pyc: backward_ddim, forward_ddim,
    InversableStableDiffusionPipeline.backward_diffusion) -- tests/golden/make_golden_ddim.py ->
    tests/golden/ddim_bytecode.json, checked by tests/test_ddim_bytecode_golden.py.  What the bytecode reads
    from diffusers objects (the timestep list of DDIMScheduler.set_timesteps, the alphas_cumprod table) is an
    INPUT of those vectors, restated from the published SD scheduler config: diffusers==0.26.0
    (requirements.txt:1) itself is neither vendored in /root/reference nor installed here.
  * DPM-Solver++ inversion (`dpms_invert_reference`): parity unpinned (it exists only inside diffusers).

Every function cites the reference file:line it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import os
import struct
from datetime import datetime

import numpy as np
from scipy.special import ndtr, ndtri  # the same Cephes kernels scipy.stats.norm.cdf/ppf dispatch to

# extract.py:83-84 semantics, found by bisection over doubles against the reference's scipy
# (tests/golden/extract_recover.json: _thresholds)
Y1_THRESHOLD = -6.957291061679417e-17   # int(norm.cdf(z)*2) >= 1  <=>  z >= Y1_THRESHOLD
Y2_THRESHOLD = 8.292361075813597        # int(norm.cdf(z)*2) == 2  <=>  z >= Y2_THRESHOLD (reference then raises)


# ----------------------------------------------------------------------------------------------
# E2: ChaCha20 as `cryptography`/OpenSSL runs it (gs_insert.py:45-47, extract.py:77-78,87)
# ----------------------------------------------------------------------------------------------
_SIGMA = np.array(struct.unpack("<4I", b"expand 32-byte k"), dtype=np.uint32)


def _rotl(x, n):
    return (x << np.uint32(n)) | (x >> np.uint32(32 - n))


def _quarter(s, a, b, c, d):
    s[a] += s[b]; s[d] = _rotl(s[d] ^ s[a], 16)
    s[c] += s[d]; s[b] = _rotl(s[b] ^ s[c], 12)
    s[a] += s[b]; s[d] = _rotl(s[d] ^ s[a], 8)
    s[c] += s[d]; s[b] = _rotl(s[b] ^ s[c], 7)


def chacha20_keystream(key: bytes, nonce16: bytes, nbytes: int) -> bytes:
    """Keystream of `algorithms.ChaCha20(key, nonce)` with the 16-byte OpenSSL nonce layout.

    nonce16[0:4] = little-endian 32-bit initial block counter, nonce16[4:16] = RFC 8439 nonce.
    Block q uses counter c = ctr0 + q; state word 12 = c mod 2^32 and the carry goes into word 13
    (OpenSSL's 64-bit counter behaviour; SURVEY.md section 0 step 4, pinned by the `carry` fixtures).
    Vectorised over blocks.
    """
    if len(key) != 32:
        raise ValueError("ChaCha20 key must be 32 bytes")      # cryptography raises ValueError too
    if len(nonce16) != 16:
        raise ValueError("ChaCha20 nonce must be 16 bytes")
    nblk = (nbytes + 63) // 64
    if nblk == 0:
        return b""
    kw = np.frombuffer(key, dtype="<u4").astype(np.uint32)
    nw = np.frombuffer(nonce16, dtype="<u4").astype(np.uint32)
    ctr = (np.uint64(nw[0]) | (np.uint64(nw[1]) << np.uint64(32))) + np.arange(nblk, dtype=np.uint64)
    init = np.empty((16, nblk), dtype=np.uint32)
    init[0:4] = _SIGMA[:, None]
    init[4:12] = kw[:, None]
    init[12] = (ctr & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    init[13] = (ctr >> np.uint64(32)).astype(np.uint32)
    init[14] = nw[2]
    init[15] = nw[3]
    s = [init[i].copy() for i in range(16)]
    with np.errstate(over="ignore"):
        for _ in range(10):
            _quarter(s, 0, 4, 8, 12); _quarter(s, 1, 5, 9, 13); _quarter(s, 2, 6, 10, 14); _quarter(s, 3, 7, 11, 15)
            _quarter(s, 0, 5, 10, 15); _quarter(s, 1, 6, 11, 12); _quarter(s, 2, 7, 8, 13); _quarter(s, 3, 4, 9, 14)
        out = np.stack([s[i] + init[i] for i in range(16)], axis=1)  # [nblk, 16]
    return out.astype("<u4").tobytes()[:nbytes]


def chacha20_xor(key: bytes, nonce16: bytes, data: bytes) -> bytes:
    ks = np.frombuffer(chacha20_keystream(key, nonce16, len(data)), dtype=np.uint8)
    return (np.frombuffer(data, dtype=np.uint8) ^ ks).tobytes()


# ----------------------------------------------------------------------------------------------
# E1: message / key / nonce preparation
# ----------------------------------------------------------------------------------------------
def pad_message(message: str, msg_bytes: int = 32) -> bytes:
    """gs_insert.py:9-20 (32 B) and nodes.py:68-76 (msg_bytes = message_length // 8)."""
    if message:
        b = str(message).encode()
        return b + b"\x00" * (msg_bytes - len(b)) if len(b) < msg_bytes else b[:msg_bytes]
    return os.urandom(msg_bytes)


def resolve_key_nonce(key_hex: str, nonce_hex: str):
    """gs_insert.py:27-42 / nodes.py:90-99 / extract.py:200-204."""
    if key_hex and nonce_hex:
        return bytes.fromhex(key_hex), bytes.fromhex(nonce_hex)
    if key_hex and not nonce_hex:
        return bytes.fromhex(key_hex), bytes.fromhex(key_hex[16:48])
    return os.urandom(32), os.urandom(16)


def choose_watermark_length(n_elems: int) -> int:
    """nodes.py:26-49."""
    for bits in (1024, 512, 256, 128, 64):
        if n_elems >= bits * 32:
            return bits
    return 32


def plaintext_bits(k: bytes, n_elems: int) -> np.ndarray:
    """s_d bits, MSB-first: k repeated floor(N / 8|k|) times, zero tail (gs_insert.py:23; nodes.py:79-87)."""
    msg_bits = 8 * len(k)
    repeats = n_elems // msg_bits
    kb = np.unpackbits(np.frombuffer(k, dtype=np.uint8))          # MSB-first, gs_insert.py:49
    out = np.zeros(n_elems, dtype=np.uint8)
    out[: repeats * msg_bits] = np.tile(kb, repeats)
    return out


def keystream_bits(key: bytes, nonce16: bytes, n_elems: int) -> np.ndarray:
    ks = np.frombuffer(chacha20_keystream(key, nonce16, (n_elems + 7) // 8), dtype=np.uint8)
    return np.unpackbits(ks)[:n_elems]


def cipher_bits(k: bytes, key: bytes, nonce16: bytes, n_elems: int) -> np.ndarray:
    """m_bits of gs_insert.py:45-49 truncated to N (nodes.py:122-123)."""
    return plaintext_bits(k, n_elems) ^ keystream_bits(key, nonce16, n_elems)


# ----------------------------------------------------------------------------------------------
# E3-E6: embed
# ----------------------------------------------------------------------------------------------
def embed_latent(k: bytes, key: bytes, nonce16: bytes, u: np.ndarray, shape) -> np.ndarray:
    """Vectorised gs_insert.py:53-66: z = norm.ppf((u + y) / 2**l), l = 1, C-order lattice; float64."""
    n = int(np.prod(shape))
    y = cipher_bits(k, key, nonce16, n).astype(np.float64)
    u = np.asarray(u, dtype=np.float64).reshape(-1)
    with np.errstate(divide="ignore"):
        z = ndtri((u + y) / 2.0)
    return z.reshape(shape)


def gs_watermark_init_noise(opt, message: str = "", *, log_path: str | None = None) -> np.ndarray:
    """Reference-shaped (vectorised) gs_insert.gs_watermark_init_noise: uses the GLOBAL numpy RNG
    exactly like gs_insert.py:62 (np.random.uniform(0,1,N) yields the same stream as N scalar calls)."""
    k = pad_message(message, 32)
    key, nonce = resolve_key_nonce(opt.key_hex, opt.nonce_hex)
    u = np.random.uniform(0, 1, 4 * 64 * 64)
    z = embed_latent(k, key, nonce, u, (4, 64, 64))
    if log_path:
        write_info_data(log_path, key, nonce, k)
    return z


def gs_watermark_init_noise_scalar(opt, message: str = "") -> np.ndarray:
    """Reference-SHAPED scalar port used for the CPU-baseline timing: one scipy.stats.norm.ppf call per
    element, string bit handling, 3-D index store -- the cost profile of gs_insert.py:49-66."""
    from scipy.stats import norm
    k = pad_message(message, 32)
    key, nonce = resolve_key_nonce(opt.key_hex, opt.nonce_hex)
    m = chacha20_xor(key, nonce, k * 64)
    m_bits = "".join(format(byte, "08b") for byte in m)
    z = np.zeros((4, 64, 64))
    index = 0
    for i in range(0, len(m_bits), 1):
        y = int(m_bits[i:i + 1], 2)
        u = np.random.uniform(0, 1)
        z[index // 4096, (index // 64) % 64, index % 64] = norm.ppf((u + y) / 2)
        index += 1
    return z


def comfy_gs_watermark_init_noise(key_hex, nonce_hex, message, use_seed, random_seed, width, height,
                                  message_length=-1) -> np.ndarray:
    """nodes.py:51-138 (generalised lattice); returns float32 (4, H/8, W/8) like the reference's tensor."""
    h, w = height // 8, width // 8
    n = 4 * h * w
    bits = message_length if message_length != -1 else choose_watermark_length(n)
    k = pad_message(message, bits // 8)
    key, nonce = resolve_key_nonce(key_hex, nonce_hex)
    rng = np.random.RandomState(seed=random_seed) if int(use_seed) == 1 else np.random
    u = rng.uniform(0, 1, n)
    return embed_latent(k, key, nonce, u, (4, h, w)).astype(np.float32)


def write_info_data(path, key: bytes, nonce: bytes, k: bytes, extra=None):
    """gs_insert.py:68-74 (extra = the ComfyUI lines, nodes.py:130-135)."""
    with open(path, "a") as f:
        f.write(f"Time: {datetime.now().strftime('%Y-%m-%d %H:%M:%S')}\n")
        f.write(f"key: {key.hex()}\n")
        f.write(f"nonce: {nonce.hex()}\n")
        f.write(f"message: {k.hex()}\n")
        for line in extra or ():
            f.write(line + "\n")
        f.write("----------------------\n")


# ----------------------------------------------------------------------------------------------
# X3-X6: extract tail
# ----------------------------------------------------------------------------------------------
def quantise(z) -> np.ndarray:
    """extract.py:82-84 with l = 1: y = int(norm.cdf(float64(z)) * 2), elements in C order."""
    zz = np.asarray(z).astype(np.float64).reshape(-1)
    c = ndtr(zz) * 2.0
    if np.isnan(c).any():
        raise ValueError("cannot convert float NaN to integer")          # int(nan) in the reference
    return c.astype(np.int64)


def recover_bits(z, key: bytes, nonce16: bytes, message_length: int) -> str:
    """extract.py:72-101.  Raises ValueError on a saturated cdf (y == 2, extract.py:86) and IndexError
    when N is not a multiple of message_length (extract.py:98)."""
    y = quantise(z)
    if (y >= 2).any():
        raise ValueError("invalid literal for int() with base 2")        # extract.py:86
    n = y.size
    if n % 8:
        y = np.concatenate([y, np.zeros(0, np.int64)])
    # pack MSB-first (a trailing partial byte is parsed by the reference as a SHORT binary literal)
    full = n // 8
    cb = np.packbits(y[: full * 8].astype(np.uint8))
    tail = y[full * 8:]
    data = cb.tobytes()
    if tail.size:
        data += bytes([int("".join(str(int(b)) for b in tail), 2)])
    pt = chacha20_xor(key, nonce16, data)
    all_bits = np.unpackbits(np.frombuffer(pt, dtype=np.uint8))          # '{:08b}' per byte, extract.py:88
    m = int(message_length)
    nseg_full, rem = divmod(all_bits.size, m)
    if rem:
        raise IndexError("string index out of range")                    # extract.py:98 on the short segment
    seg = all_bits.reshape(nseg_full, m)
    count1 = seg.sum(axis=0)
    return "".join("1" if c > nseg_full / 2 else "0" for c in count1)    # strict majority, ties -> '0'


def recover_exactracted_message(reversed_latents, args) -> str:
    """Signature twin of extract.py:72."""
    return recover_bits(reversed_latents, args.key, args.nonce, int(args.message_length))


def recover_exactracted_message_scalar(reversed_latents, args) -> str:
    """Reference-SHAPED scalar port (one scipy.stats.norm.cdf per element, string bits, Python vote) used
    for the CPU-baseline timing -- the cost profile of extract.py:72-101."""
    from scipy.stats import norm
    bits = []
    for v in np.nditer(np.asarray(reversed_latents)):
        bits.append(int(norm.cdf(v) * 2 ** args.l))
    mb = bytes(int("".join(str(b) for b in bits[i:i + 8]), 2) for i in range(0, len(bits), 8))
    sd = chacha20_xor(args.key, args.nonce, mb)
    all_bits = "".join("{:08b}".format(b) for b in sd)
    m = int(args.message_length)
    segs = [all_bits[i:i + m] for i in range(0, len(all_bits), m)]
    out = ""
    for i in range(m):
        c1 = sum(s[i] == "1" for s in segs)
        out += "1" if c1 > len(segs) / 2 else "0"
    return out


def calculate_bit_accuracy(original_message_hex: str, extracted_message_bin: str):
    """extract.py:103-110."""
    ob = bin(int(original_message_hex, 16))[2:].zfill(len(original_message_hex) * 4)
    n = min(len(ob), len(extracted_message_bin))
    ob, eb = ob[:n], extracted_message_bin[:n]
    match = sum(1 for x, y in zip(ob, eb) if x == y)
    return ob, match / n


def bits_to_bytes(bits: str) -> bytes:
    return np.packbits(np.array([c == "1" for c in bits], dtype=np.uint8)).tobytes()


# ----------------------------------------------------------------------------------------------
# X2 / G1: DDIM (eta = 0) step -- pinned against the reference's bytecode (see module docstring)
# ----------------------------------------------------------------------------------------------
def sd_alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012) -> np.ndarray:
    """Stable-Diffusion 'scaled_linear' schedule (scheduler_config.json of stabilityai/stable-diffusion-2-1-base)."""
    betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=np.float64) ** 2
    return np.cumprod(1.0 - betas)


def backward_ddim(x_t, alpha_t, alpha_tm1, eps):
    """inverse_stable_diffusion_gs.pyc src :18-26 (closed-form eta=0 DDIM move from alpha_t to alpha_tm1)."""
    return (alpha_tm1 ** 0.5 * ((alpha_t ** -0.5 - alpha_tm1 ** -0.5) * x_t
                                + ((1 / alpha_tm1 - 1) ** 0.5 - (1 / alpha_t - 1) ** 0.5) * eps) + x_t)


def ddim_coefficients(alpha_from: float, alpha_to: float, prediction_type: str = "epsilon"):
    """Per-step scalars (a, b) with x_to = a * x_from + b * model_out (float64 on the host).

    epsilon:       x0 = (x - sqrt(1-af) e)/sqrt(af);              x' = sqrt(at) x0 + sqrt(1-at) e
    v_prediction:  x0 = sqrt(af) x - sqrt(1-af) v, e = sqrt(af) v + sqrt(1-af) x   (SD 2.1-768)
    """
    af, at = float(alpha_from), float(alpha_to)
    if prediction_type == "epsilon":
        a = (at / af) ** 0.5
        b = (1 - at) ** 0.5 - (at * (1 - af) / af) ** 0.5
    elif prediction_type == "v_prediction":
        a = (at * af) ** 0.5 + ((1 - at) * (1 - af)) ** 0.5
        b = ((1 - at) * af) ** 0.5 - (at * (1 - af)) ** 0.5
    else:
        raise ValueError(prediction_type)
    return a, b


def ddim_timesteps(num_inference_steps: int, num_train_timesteps: int = 1000, steps_offset: int = 1) -> np.ndarray:
    """'leading' spacing, descending (SD scheduler_config: steps_offset=1)."""
    ratio = num_train_timesteps // num_inference_steps
    return (np.arange(num_inference_steps) * ratio).round()[::-1].astype(np.int64) + steps_offset


def ddim_schedule(num_inference_steps: int, inverse: bool, num_train_timesteps: int = 1000,
                  prediction_type: str = "epsilon", final_alpha_cumprod: float | None = None):
    """List of (t_model, a, b) for the sampling (inverse=False) or inversion (inverse=True) loop, following
    backward_diffusion of inverse_stable_diffusion_gs.pyc src :101-185: prev_t = t - T/S, alpha_prev =
    alphas_cumprod[prev_t] if prev_t >= 0 else final_alpha_cumprod (= alphas_cumprod[0], set_alpha_to_one
    False); for reverse_process the timestep order is reversed and the two alphas are swapped."""
    ac = sd_alphas_cumprod(num_train_timesteps)
    final = ac[0] if final_alpha_cumprod is None else final_alpha_cumprod
    ts = ddim_timesteps(num_inference_steps, num_train_timesteps)
    if inverse:
        ts = ts[::-1]
    out = []
    for t in ts:
        prev = int(t) - num_train_timesteps // num_inference_steps
        a_t = ac[int(t)]
        a_prev = ac[prev] if prev >= 0 else final
        af, at = (a_prev, a_t) if inverse else (a_t, a_prev)
        a, b = ddim_coefficients(af, at, prediction_type)
        out.append((int(t), a, b))
    return out


def dpms_invert_reference(eps_fn, x0, num_inference_steps, num_train_timesteps=1000):
    """DPM-Solver++(2M) inversion written step by step the way diffusers' DPMSolverMultistepInverseScheduler does it
    (convert_model_output -> x0 prediction list -> first-order / second-order multistep update; linspace spacing, midpoint,
    lower_order_final below 15 steps).  float64; eps_fn(x, t) -> eps.  PARITY UNPINNED (no diffusers here)."""
    ac = sd_alphas_cumprod(num_train_timesteps)
    S = num_inference_steps
    ts = np.linspace(0, num_train_timesteps - 1, S + 1).round()[:-1].astype(np.int64)
    all_sig = ((1 - ac) / ac) ** 0.5
    sigmas = np.concatenate([np.interp(ts, np.arange(len(all_sig)), all_sig), [all_sig[-1]]])

    def a_s(sig):
        a = 1.0 / np.sqrt(sig * sig + 1.0)
        return a, sig * a

    x = np.array(x0, dtype=np.float64)
    outs = []
    lower_order_nums = 0
    for i, t in enumerate(ts):
        eps = eps_fn(x, int(t))
        a0, s0 = a_s(sigmas[i])
        outs.append((x - s0 * eps) / a0)                       # data prediction
        at, st = a_s(sigmas[i + 1])
        lam_t, lam_s0 = np.log(at) - np.log(st), np.log(a0) - np.log(s0)
        h = lam_t - lam_s0
        final_lower = (i == S - 1) and S < 15
        if lower_order_nums < 1 or final_lower:
            x = (st / s0) * x - at * (np.exp(-h) - 1.0) * outs[-1]
        else:
            a1, s1 = a_s(sigmas[i - 1])
            lam_s1 = np.log(a1) - np.log(s1)
            r0 = (lam_s0 - lam_s1) / h
            D0, D1 = outs[-1], (1.0 / r0) * (outs[-1] - outs[-2])
            x = (st / s0) * x - at * (np.exp(-h) - 1.0) * D0 - 0.5 * at * (np.exp(-h) - 1.0) * D1
        if lower_order_nums < 2:
            lower_order_nums += 1
    return x


# ----------------------------------------------------------------------------------------------
# In-kernel RNG of the build (NOT reference behaviour): Philox4x32-7 (Random123's philox4x32_R(7); the 10-round
# default is kept for its published known answers), restated so tests can check the HIP
# kernel's `u` stream.  Group g = e >> 2 of image `img` draws Philox(counter = (g, 0, img_lo, img_hi),
# key = 64-bit seed); element e takes word e & 3; u = (w + 0.5) * 2^-32, exactly representable in fp64.
# ----------------------------------------------------------------------------------------------
_PH_M0, _PH_M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_PH_W0, _PH_W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


PHILOX_ROUNDS = 7        # rounds of the build's throughput stream (csrc/gswm_kernels.hip: GSW_PHILOX_ROUNDS)


def philox4x32(c0, c1, c2, c3, k0, k1, rounds=10):
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint32) for x in (c0, c1, c2, c3))
    k0 = np.uint32(k0); k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(rounds):
            p0 = _PH_M0 * c0.astype(np.uint64)
            p1 = _PH_M1 * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n1 = (p1 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            n3 = (p0 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32(k0 + _PH_W0); k1 = np.uint32(k1 + _PH_W1)
    return c0, c1, c2, c3


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    return philox4x32(c0, c1, c2, c3, k0, k1, 10)


def philox_uniform(seed: int, image_index0: int, batch: int, n_elems: int) -> np.ndarray:
    """u[b, e] of the build's in-kernel RNG (float64 in (0,1))."""
    ngroups = (n_elems + 3) // 4
    g = np.arange(ngroups, dtype=np.uint32)
    out = np.empty((batch, n_elems), dtype=np.float64)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    for b in range(batch):
        img = image_index0 + b
        w = philox4x32(g, np.zeros(ngroups, np.uint32), np.full(ngroups, img & 0xFFFFFFFF, np.uint32),
                       np.full(ngroups, (img >> 32) & 0xFFFFFFFF, np.uint32), k0, k1, PHILOX_ROUNDS)
        words = np.stack(w, axis=1).reshape(-1)[:n_elems]
        out[b] = (words.astype(np.float64) + 0.5) * 2.0 ** -32
    return out
