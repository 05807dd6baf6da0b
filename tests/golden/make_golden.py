#!/opt/conda/bin/python3.9
"""Generate the golden fixtures in tests/golden/ by IMPORTING the reference.

Run in the build container only (the reference never travels to the GPU box):

    cd /root/repo && /opt/conda/bin/python3.9 tests/golden/make_golden.py

Interpreter: /opt/conda/bin/python3.9 -- the only one here that has the reference's
third-party deps for the codec (cryptography 3.4.8 / OpenSSL ChaCha20, scipy 1.7.1, numpy 1.26).
It has no torch / diffusers / torchvision, so:
  * `extract.py` is imported with inert stand-ins for those three modules; only
    `recover_exactracted_message` and `calculate_bit_accuracy` are called and they touch nothing
    but cryptography / scipy / numpy.
  * `ComfyUI_GSWaterMark/nodes.py` (the only reference for non-64x64 lattices) is imported with
    inert `comfy.*` / `latent_preview` stand-ins and a *container-only* `torch` stand-in whose
    `zeros(shape, dtype=float32).cpu()` is a numpy float32 array: the arithmetic (ChaCha20,
    MT19937 uniform, scipy norm.ppf) is the real thing, only the float32 store is numpy's
    (same IEEE round-to-nearest as torch's).

Outputs are data only (inputs + expected outputs); no reference source text is stored.
"""
import hashlib
import io
import json
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True  # /root/reference is read-only
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

import numpy as np  # noqa: E402

README_KEY = "5822ff9cce6772f714192f43863f6bad1bf54b78326973897e6b66c3186b77a7"  # README.md:61
README_NONCE = "05072fd1c2265f6f2e2a4080a2bfbdd8"  # README.md:67


class _Any(types.ModuleType):
    """Inert stand-in: any attribute is an empty class (lets module-level imports succeed)."""

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return type(k, (), {})


def _install_stubs():
    for n in ["torch", "diffusers", "diffusers.utils", "torchvision", "torchvision.transforms",
              "matplotlib", "matplotlib.pyplot", "tqdm", "PIL"]:
        if n not in sys.modules:
            try:
                __import__(n)
            except Exception:
                sys.modules[n] = _Any(n)
    if isinstance(sys.modules.get("tqdm"), _Any):
        sys.modules["tqdm"].tqdm = lambda x, *a, **k: x


def _install_comfy_stubs():
    class _F32(np.ndarray):
        def cpu(self):
            return self

    t = types.ModuleType("torch")
    t.float32 = np.float32

    def zeros(shape, dtype=np.float32, device=None):
        return np.zeros(shape, dtype=dtype).view(_F32)

    t.zeros = zeros
    sys.modules["torch"] = t
    comfy = _Any("comfy")
    sys.modules["comfy"] = comfy
    for sub in ["model_management", "sample", "sampler_helpers", "diffusers_load", "samplers", "sd", "utils"]:
        m = _Any("comfy." + sub)
        sys.modules["comfy." + sub] = m
        setattr(comfy, sub, m)
    ks = type("KSampler", (), {"SAMPLERS": ["euler"], "SCHEDULERS": ["normal"]})
    sys.modules["comfy.samplers"].KSampler = ks
    sys.modules["latent_preview"] = _Any("latent_preview")


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def keystream(key: bytes, nonce: bytes, n: int) -> bytes:
    """Real OpenSSL ChaCha20 via the reference's own dependency (gs_insert.py:45-47 call pattern)."""
    from cryptography.hazmat.primitives.ciphers import Cipher, algorithms
    from cryptography.hazmat.backends import default_backend
    enc = Cipher(algorithms.ChaCha20(key, nonce), mode=None, backend=default_backend()).encryptor()
    return enc.update(b"\x00" * n) + enc.finalize()


def main():
    os.makedirs(OUT, exist_ok=True)
    scratch = tempfile.mkdtemp(prefix="gsw_golden_")
    os.chdir(scratch)  # gs_insert appends ./info_data.txt
    sys.path.insert(0, REF)
    _install_stubs()
    import gs_insert  # noqa
    import extract  # noqa

    meta = {"generator": "tests/golden/make_golden.py", "python": sys.version.split()[0],
            "numpy": np.__version__}
    import scipy, cryptography  # noqa
    meta["scipy"] = scipy.__version__
    meta["cryptography"] = cryptography.__version__

    # ---------------------------------------------------------------- (1) ChaCha20 keystreams
    cases = {
        "readme": (README_KEY, README_NONCE),
        "key_only": (README_KEY, README_KEY[16:48]),                      # gs_insert.py:33-39
        "carry": (README_KEY, "feffffff" + "ffffffff" + "0102030405060708"),  # ctr=0xfffffffe, word13=0xffffffff
        "carry2": ("00" * 31 + "01", "ffffffff" + "00000000" + "a1a2a3a4a5a6a7a8"),
        "zero": ("00" * 32, "00" * 16),
    }
    ks = {}
    for name, (k, n) in cases.items():
        s = keystream(bytes.fromhex(k), bytes.fromhex(n), 4608)
        ks[name] = {"key_hex": k, "nonce_hex": n, "n": 4608, "sha256": sha(s), "sha256_2048": sha(s[:2048]),
                    "stream_hex": s.hex()}
    with open(os.path.join(OUT, "chacha20_keystreams.json"), "w") as f:
        json.dump({"meta": meta, "cases": ks}, f, indent=1)

    # ---------------------------------------------------------------- (2) embed, gs_insert.py
    emb_meta = {}
    arrays = {}
    embed_cases = [
        ("s0_lthero", 0, "lthero", README_KEY, README_NONCE),
        ("s1_lthero", 1, "lthero", README_KEY, README_NONCE),
        ("s42_32byte", 42, "0123456789abcdefghijklmnopqrstuv", README_KEY, README_NONCE),
        ("s42_long_utf8", 42, "水印测试-watermark-éè-" + "x" * 40, README_KEY, README_NONCE),
        ("s7_keyonly", 7, "lthero", README_KEY, ""),
        ("s3_single_char", 3, "A", "ab" * 32, "cd" * 16),
    ]
    for name, seed, msg, kh, nh in embed_cases:
        opt = types.SimpleNamespace(key_hex=kh, nonce_hex=nh)
        np.random.seed(seed)
        Z = gs_insert.gs_watermark_init_noise(opt, msg)
        assert Z.shape == (4, 64, 64) and Z.dtype == np.float64
        emb_meta[name] = {"seed": seed, "message": msg, "key_hex": kh, "nonce_hex": nh,
                          "sha256_f64": sha(Z.tobytes()), "sha256_f32": sha(Z.astype(np.float32).tobytes()),
                          "head_f64": [float(x) for x in Z.flat[:8]], "tail_f64": [float(x) for x in Z.flat[-8:]],
                          "mean": float(Z.mean()), "std": float(Z.std())}
        if name in ("s0_lthero", "s42_long_utf8", "s7_keyonly"):
            arrays["Z32_" + name] = Z.astype(np.float32)
        arrays["Z64head_" + name] = Z.reshape(-1)[:512].copy()
    # info_data.txt format pin (gs_insert.py:68-74): keep the text of the LAST call minus the time stamp
    txt = open("info_data.txt").read().strip().split("----------------------\n")
    last = [ln for ln in txt[-1].strip().splitlines()]
    emb_meta["_info_data_last_record"] = [ln if not ln.startswith("Time: ") else "Time: <%Y-%m-%d %H:%M:%S>" for ln in last]
    # empty message => os.urandom(32) watermark; pin only the log/recover relation
    np.random.seed(5)
    opt = types.SimpleNamespace(key_hex=README_KEY, nonce_hex=README_NONCE)
    Zr = gs_insert.gs_watermark_init_noise(opt, "")
    rec = open("info_data.txt").read().strip().split("----------------------")[-2].strip().splitlines()
    khex = [ln for ln in rec if ln.startswith("message: ")][0].split(": ")[1]
    a = types.SimpleNamespace(key=bytes.fromhex(README_KEY), nonce=bytes.fromhex(README_NONCE), l=1, message_length=256)
    bits = extract.recover_exactracted_message(Zr, a)
    assert extract.calculate_bit_accuracy(khex, bits)[1] == 1.0
    emb_meta["_random_message_roundtrip"] = {"message_hex_len": len(khex), "accuracy": 1.0}

    # ---------------------------------------------------------------- (3) extract, extract.py:72-110
    ext = {}
    np.random.seed(0)
    opt = types.SimpleNamespace(key_hex=README_KEY, nonce_hex=README_NONCE)
    Z0 = gs_insert.gs_watermark_init_noise(opt, "lthero")
    key, nonce = bytes.fromhex(README_KEY), bytes.fromhex(README_NONCE)
    msg_hex = (b"lthero" + b"\0" * 26).hex()

    def rec_bits(z, ml, k=key, n=nonce):
        a = types.SimpleNamespace(key=k, nonce=n, l=1, message_length=ml)
        return extract.recover_exactracted_message(z, a)

    ext["clean_f64_256"] = {"input": "Z32_s0_lthero(as f64 of gs_insert)", "message_length": 256,
                            "bits": rec_bits(Z0, 256)}
    ext["clean_f16_256"] = {"message_length": 256, "bits": rec_bits(Z0.astype(np.float16), 256)}
    ext["clean_f32_1x4x64x64_256"] = {"message_length": 256, "bits": rec_bits(Z0.astype(np.float32)[None], 256)}
    for ml in (32, 64, 128, 512, 1024, 2048, 16384):
        ext["clean_f32_%d" % ml] = {"message_length": ml, "bits": rec_bits(Z0.astype(np.float32), ml)}
    # noisy latents: accuracy < 1 cases.  noise is regenerated in the tests from the same seeds.
    for sigma, nseed in ((0.5, 100), (1.0, 101), (2.0, 102), (4.0, 103)):
        rng = np.random.RandomState(nseed)
        zn = (Z0.astype(np.float32) + np.float32(sigma) * rng.standard_normal(Z0.shape).astype(np.float32)).astype(np.float16)
        arrays["Znoisy16_%g" % sigma] = zn
        try:
            b = rec_bits(zn, 256)
            ext["noisy_sigma%g" % sigma] = {"message_length": 256, "noise_seed": nseed, "sigma": sigma, "bits": b,
                                            "accuracy": extract.calculate_bit_accuracy(msg_hex, b)[1]}
        except ValueError as e:   # a latent >= 8.2924 saturates norm.cdf -> y == 2 -> extract.py:86 raises
            ext["noisy_sigma%g" % sigma] = {"message_length": 256, "noise_seed": nseed, "sigma": sigma,
                                            "raises": "ValueError", "n_saturated": int((zn.astype(np.float64) >= 8.292361075813597).sum())}
        # the same latent clipped to +-8 always decodes
        zc = np.clip(zn, np.float16(-8), np.float16(8))
        b = rec_bits(zc, 256)
        ext["noisy_clip8_sigma%g" % sigma] = {"message_length": 256, "bits": b,
                                              "accuracy": extract.calculate_bit_accuracy(msg_hex, b)[1]}
    # wrong key => ~50 %
    b = rec_bits(Z0.astype(np.float16), 256, k=bytes.fromhex("11" * 32))
    ext["wrong_key"] = {"message_length": 256, "key_hex": "11" * 32, "bits": b,
                        "accuracy": extract.calculate_bit_accuracy(msg_hex, b)[1]}
    # synthetic ties: flip the sign of element c*256+t for exactly 32 (tie -> '0'), 31 and 33 of the 64 copies
    zt = Z0.astype(np.float32).reshape(-1).copy()
    tie_spec = {"8": 32, "9": 31, "10": 33, "0": 32, "255": 32, "13": 64}
    for t, nflip in tie_spec.items():
        for c in range(nflip):
            zt[c * 256 + int(t)] *= -1.0
    ext["ties_256"] = {"message_length": 256, "flip_spec": tie_spec, "bits": rec_bits(zt.reshape(4, 64, 64), 256)}
    # all-zero / signed-zero latents
    ext["all_pos_zero"] = {"message_length": 256, "bits": rec_bits(np.zeros((4, 64, 64), np.float32), 256)}
    ext["all_neg_zero"] = {"message_length": 256, "bits": rec_bits(-np.zeros((4, 64, 64), np.float32), 256)}
    # error semantics: saturated cdf -> ValueError (extract.py:86), N % message_length != 0 -> IndexError (:98)
    zs = Z0.astype(np.float32).copy()
    zs[1, 2, 3] = 9.0
    for nm, fn in (("saturated", lambda: rec_bits(zs, 256)), ("ragged_ml", lambda: rec_bits(Z0, 1000)),
                   ("nan", lambda: rec_bits(np.full((4, 64, 64), np.nan, np.float32), 256))):
        try:
            fn()
            ext["error_" + nm] = {"raises": None}
        except Exception as e:  # noqa
            ext["error_" + nm] = {"raises": type(e).__name__}

    # ---------------------------------------------------------------- (4) X3 edge scalars
    from scipy.stats import norm
    edge_in = [0.0, -0.0, -5.5e-17, -6.9e-17, -6.957291061679417e-17, -6.96e-17, -7.0e-17, -1e-16, 1e-300, -1e-300,
               8.2, 8.29, 8.292361075813595, 8.292361075813597, 8.3, 37.0, -37.0, -40.0, float("inf"), float("-inf"),
               6.0e-8, -6.0e-8, 65504.0, -65504.0]
    edge = [{"z": repr(float(z)), "y": int(norm.cdf(np.float64(z)) * 2)} for z in edge_in]
    # exact decision thresholds by bisection on doubles
    def bisect(lo, hi, want):
        lo, hi = np.float64(lo), np.float64(hi)
        while True:
            mid = np.float64(lo + (hi - lo) / 2)
            if mid == lo or mid == hi:
                return float(hi)
            if int(norm.cdf(mid) * 2) >= want:
                hi = mid
            else:
                lo = mid
    thr1 = bisect(-1e-15, 0.0, 1)
    thr2 = bisect(8.0, 9.0, 2)
    ext["_thresholds"] = {"y_ge_1_iff_z_ge": repr(thr1), "y_ge_2_iff_z_ge": repr(thr2)}
    ext["_edge_scalars"] = edge

    # ---------------------------------------------------------------- (6) calculate_bit_accuracy
    acc = []
    for hx, bn in ((msg_hex, ext["clean_f64_256"]["bits"]), ("ff00", "1111000011110000"), ("0f", "00001111" + "1010"),
                   ("abcdef", "1010"), ("00ff", "1" * 16), (msg_hex, ext["noisy_clip8_sigma2"]["bits"])):
        ob, a_ = extract.calculate_bit_accuracy(hx, bn)
        acc.append({"hex": hx, "bin": bn, "original_bin": ob, "accuracy": a_})
    ext["_bit_accuracy"] = acc

    with open(os.path.join(OUT, "embed_gs_insert.json"), "w") as f:
        json.dump({"meta": meta, "cases": emb_meta}, f, indent=1, ensure_ascii=True)
    with open(os.path.join(OUT, "extract_recover.json"), "w") as f:
        json.dump({"meta": meta, "msg_hex": msg_hex, "key_hex": README_KEY, "nonce_hex": README_NONCE, "cases": ext}, f, indent=1)

    # ---------------------------------------------------------------- (5) ComfyUI generalised lattice
    _install_comfy_stubs()
    sys.path.insert(0, os.path.join(REF, "ComfyUI_GSWaterMark"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("gsw_ref_nodes", os.path.join(REF, "ComfyUI_GSWaterMark", "nodes.py"))
    nodes = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(nodes)
    comfy = {}
    comfy_cases = [
        ("c768_ml256", 768, 768, 256, 11, "lthero"),
        ("c768_auto", 768, 768, -1, 12, "lthero-comfy"),
        ("c512_auto", 512, 512, -1, 13, "lthero"),
        ("c512x768_ml512", 512, 768, 512, 14, "rect"),
        ("c200x136_auto", 200, 136, -1, 15, "tiny"),      # 4*25*17 = 1700 elements: ragged tail, blocks straddle
        ("c768_ml1000", 768, 768, 1000, 16, "nondiv"),    # N % ml != 0: zero tail padding path (nodes.py:85-87)
    ]
    for name, W, H, ml, seed, msg in comfy_cases:
        Z = nodes.gs_watermark_init_noise(README_KEY, README_NONCE, "cpu", msg, 1, seed, W, H, ml)
        Z = np.asarray(Z)
        assert Z.dtype == np.float32 and Z.shape == (4, H // 8, W // 8), (Z.dtype, Z.shape)
        n = Z.size
        eff_ml = ml if ml != -1 else nodes.choose_watermark_length(n)
        comfy[name] = {"width": W, "height": H, "message_length": ml, "effective_message_length": eff_ml,
                       "seed": seed, "message": msg, "shape": list(Z.shape), "sha256_f32": sha(Z.tobytes()),
                       "head": [float(x) for x in Z.flat[:8]], "tail": [float(x) for x in Z.flat[-8:]]}
        if n % eff_ml == 0:
            a = types.SimpleNamespace(key=key, nonce=nonce, l=1, message_length=eff_ml)
            comfy[name]["recovered_bits"] = extract.recover_exactracted_message(Z, a)
        if name in ("c768_auto", "c200x136_auto", "c768_ml1000"):
            arrays["Zc32_" + name] = Z.copy()
    comfy["_choose_watermark_length"] = {str(n): nodes.choose_watermark_length(n)
                                        for n in (0, 1, 2047, 2048, 4095, 4096, 8191, 8192, 16383, 16384, 32767, 32768, 36864, 65536, 1 << 20)}
    # unseeded mode uses the global numpy stream (nodes.py:114-115)
    np.random.seed(21)
    Zg = np.asarray(nodes.gs_watermark_init_noise(README_KEY, README_NONCE, "cpu", "lthero", 0, 999, 512, 512, 256))
    comfy["c512_global_rng"] = {"np_seed": 21, "sha256_f32": sha(Zg.tobytes()), "head": [float(x) for x in Zg.flat[:8]]}
    rec = open("info_data.txt").read().strip().split("----------------------")[-2].strip().splitlines()
    comfy["_info_data_last_record"] = [ln if not ln.startswith("Time: ") else "Time: <%Y-%m-%d %H:%M:%S>" for ln in rec]
    with open(os.path.join(OUT, "embed_comfy_nodes.json"), "w") as f:
        json.dump({"meta": meta, "key_hex": README_KEY, "nonce_hex": README_NONCE, "cases": comfy}, f, indent=1)

    np.savez_compressed(os.path.join(OUT, "arrays.npz"), **arrays)
    print("wrote", sorted(os.listdir(OUT)))
    for fn in sorted(os.listdir(OUT)):
        print("%9d %s" % (os.path.getsize(os.path.join(OUT, fn)), fn))


if __name__ == "__main__":
    main()
