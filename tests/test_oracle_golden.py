"""CPU: the oracle (oracle/gs_oracle.py) against every golden vector generated from the reference
(tests/golden/make_golden.py) and against the public RFC 8439 known answers."""
import hashlib
import types

import numpy as np
import pytest

import gs_oracle as O
from conftest import README_KEY, README_NONCE


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


# ------------------------------------------------------------------ E2 ChaCha20
def test_chacha20_matches_openssl_fixtures(golden):
    for name, c in golden["chacha"]["cases"].items():
        s = O.chacha20_keystream(bytes.fromhex(c["key_hex"]), bytes.fromhex(c["nonce_hex"]), c["n"])
        assert s.hex() == c["stream_hex"], name
        assert sha(s[:2048]) == c["sha256_2048"]


def test_chacha20_rfc8439_block_function():
    # RFC 8439 section 2.3.2: key 00..1f, nonce 000000090000004a00000000, counter 1
    key = bytes(range(32))
    nonce16 = (1).to_bytes(4, "little") + bytes.fromhex("000000090000004a00000000")
    ks = O.chacha20_keystream(key, nonce16, 64)
    assert ks.hex() == ("10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4e"
                        "d2826446079faa0914c2d705d98b02a2b5129cd1de164eb9cbd083e8a2503c4e")


def test_chacha20_rfc8439_encryption():
    # RFC 8439 section 2.4.2
    key = bytes(range(32))
    nonce16 = (1).to_bytes(4, "little") + bytes.fromhex("000000000000004a00000000")
    pt = (b"Ladies and Gentlemen of the class of '99: If I could offer you only one tip for the future, "
          b"sunscreen would be it.")
    ct = O.chacha20_xor(key, nonce16, pt)
    assert ct.hex() == ("6e2e359a2568f98041ba0728dd0d6981e97e7aec1d4360c20a27afccfd9fae0b"
                        "f91b65c5524733ab8f593dabcd62b3571639d624e65152ab8f530c359f0861d8"
                        "07ca0dbf500d6a6156a38e088a22b65e52bc514d16ccf806818ce91ab7793736"
                        "5af90bbf74a35be6b40b8eedf2785e42874d")


def test_chacha20_bad_sizes():
    with pytest.raises(ValueError):
        O.chacha20_keystream(b"x" * 31, b"y" * 16, 64)
    with pytest.raises(ValueError):
        O.chacha20_keystream(b"x" * 32, b"y" * 12, 64)
    assert O.chacha20_keystream(b"x" * 32, b"y" * 16, 0) == b""


# ------------------------------------------------------------------ E1-E6 embed
def test_embed_matches_gs_insert_fixtures(golden):
    arrays = golden["arrays"]
    for name, c in golden["embed"]["cases"].items():
        if name.startswith("_"):
            continue
        np.random.seed(c["seed"])
        z = O.gs_watermark_init_noise(types.SimpleNamespace(key_hex=c["key_hex"], nonce_hex=c["nonce_hex"]), c["message"])
        assert z.shape == (4, 64, 64) and z.dtype == np.float64
        assert sha(z.tobytes()) == c["sha256_f64"], name                      # bit-exact fp64
        assert sha(z.astype(np.float32).tobytes()) == c["sha256_f32"], name
        np.testing.assert_array_equal(z.reshape(-1)[:512], arrays["Z64head_" + name])
        if "Z32_" + name in arrays.files:
            np.testing.assert_array_equal(z.astype(np.float32), arrays["Z32_" + name])


def test_embed_scalar_port_equals_vectorised():
    opt = types.SimpleNamespace(key_hex=README_KEY, nonce_hex=README_NONCE)
    # 16384 scalar scipy.stats.norm.ppf calls take ~1 s: this is the reference-shaped port used as the CPU baseline
    np.random.seed(0)
    a = O.gs_watermark_init_noise_scalar(opt, "lthero")
    np.random.seed(0)
    b = O.gs_watermark_init_noise(opt, "lthero")
    np.testing.assert_array_equal(a, b)


def test_embed_comfy_lattices(golden):
    g = golden["comfy"]
    for name, c in g["cases"].items():
        if name.startswith("_") or "width" not in c:
            continue
        z = O.comfy_gs_watermark_init_noise(g["key_hex"], g["nonce_hex"], c["message"], 1, c["seed"], c["width"], c["height"],
                                            c["message_length"])
        assert list(z.shape) == c["shape"] and z.dtype == np.float32
        assert sha(z.tobytes()) == c["sha256_f32"], name
        if "Zc32_" + name in golden["arrays"].files:
            np.testing.assert_array_equal(z, golden["arrays"]["Zc32_" + name])
        if "recovered_bits" in c:
            assert O.recover_bits(z, bytes.fromhex(g["key_hex"]), bytes.fromhex(g["nonce_hex"]), c["effective_message_length"]) == c["recovered_bits"]
    np.random.seed(21)
    z = O.comfy_gs_watermark_init_noise(g["key_hex"], g["nonce_hex"], "lthero", 0, 999, 512, 512, 256)
    assert sha(z.tobytes()) == g["cases"]["c512_global_rng"]["sha256_f32"]
    for n, bits in g["cases"]["_choose_watermark_length"].items():
        assert O.choose_watermark_length(int(n)) == bits


def test_info_data_format(golden, tmp_path):
    p = tmp_path / "info_data.txt"
    O.write_info_data(str(p), bytes.fromhex("ab" * 32), bytes.fromhex("cd" * 16), b"A" + b"\0" * 31)
    got = p.read_text().splitlines()
    want = golden["embed"]["cases"]["_info_data_last_record"]
    assert got[0].startswith("Time: ") and len(got[0]) == len("Time: 2024-01-01 00:00:00")
    assert got[1:] == want[1:]


# ------------------------------------------------------------------ X3-X6 extract
def test_recover_matches_extract_fixtures(golden, keys):
    key, nonce = keys
    x, arrays = golden["extract"]["cases"], golden["arrays"]
    z0 = arrays["Z32_s0_lthero"]
    assert O.recover_bits(z0.astype(np.float64), key, nonce, 256) == x["clean_f64_256"]["bits"]
    assert O.recover_bits(z0.astype(np.float16), key, nonce, 256) == x["clean_f16_256"]["bits"]
    assert O.recover_bits(z0[None], key, nonce, 256) == x["clean_f32_1x4x64x64_256"]["bits"]
    for ml in (32, 64, 128, 512, 1024, 2048, 16384):
        assert O.recover_bits(z0, key, nonce, ml) == x["clean_f32_%d" % ml]["bits"]
    for s in ("0.5", "1", "2", "4"):
        zn, c = arrays["Znoisy16_" + s], x["noisy_sigma" + s]
        if c.get("raises"):
            with pytest.raises(ValueError):
                O.recover_bits(zn, key, nonce, 256)
            assert int((zn.astype(np.float64) >= O.Y2_THRESHOLD).sum()) == c["n_saturated"]
        else:
            assert O.recover_bits(zn, key, nonce, 256) == c["bits"]
        zc = np.clip(zn, np.float16(-8), np.float16(8))
        b = O.recover_bits(zc, key, nonce, 256)
        assert b == x["noisy_clip8_sigma" + s]["bits"]
        assert O.calculate_bit_accuracy(golden["extract"]["msg_hex"], b)[1] == x["noisy_clip8_sigma" + s]["accuracy"]
    assert O.recover_bits(z0.astype(np.float16), bytes.fromhex(x["wrong_key"]["key_hex"]), nonce, 256) == x["wrong_key"]["bits"]


def test_recover_ties_and_zeros(golden, keys):
    key, nonce = keys
    x = golden["extract"]["cases"]
    zt = golden["arrays"]["Z32_s0_lthero"].reshape(-1).copy()
    for t, nflip in x["ties_256"]["flip_spec"].items():
        for c in range(nflip):
            zt[c * 256 + int(t)] *= -1.0
    assert O.recover_bits(zt, key, nonce, 256) == x["ties_256"]["bits"]
    assert O.recover_bits(np.zeros((4, 64, 64), np.float32), key, nonce, 256) == x["all_pos_zero"]["bits"]
    assert O.recover_bits(-np.zeros((4, 64, 64), np.float32), key, nonce, 256) == x["all_neg_zero"]["bits"]


def test_recover_error_semantics(golden, keys):
    key, nonce = keys
    x = golden["extract"]["cases"]
    z0 = golden["arrays"]["Z32_s0_lthero"].copy()
    assert x["error_saturated"]["raises"] == "ValueError" and x["error_ragged_ml"]["raises"] == "IndexError"
    assert x["error_nan"]["raises"] == "ValueError"
    zs = z0.copy(); zs[1, 2, 3] = 9.0
    with pytest.raises(ValueError):
        O.recover_bits(zs, key, nonce, 256)
    with pytest.raises(IndexError):
        O.recover_bits(z0, key, nonce, 1000)
    with pytest.raises(ValueError):
        O.recover_bits(np.full((4, 64, 64), np.nan, np.float32), key, nonce, 256)


def test_quantise_edge_scalars(golden):
    x = golden["extract"]["cases"]
    assert float(x["_thresholds"]["y_ge_1_iff_z_ge"]) == O.Y1_THRESHOLD
    assert float(x["_thresholds"]["y_ge_2_iff_z_ge"]) == O.Y2_THRESHOLD
    for d in x["_edge_scalars"]:
        z = float(d["z"])
        assert int(O.quantise(np.array([z]))[0]) == d["y"], d
        thr = 2 if z >= O.Y2_THRESHOLD else (1 if z >= O.Y1_THRESHOLD else 0)
        assert thr == d["y"], d
    # the threshold form is exactly the cdf form on a dense sample around both thresholds
    lo = np.nextafter(O.Y1_THRESHOLD, -1.0) + np.arange(-50, 50) * 1e-19
    z = np.concatenate([lo, np.linspace(-1e-15, 1e-15, 4001), np.nextafter(O.Y2_THRESHOLD, 0.0) + np.arange(-50, 50) * 1.8e-15,
                        np.random.RandomState(0).standard_normal(100000) * 3])
    y = O.quantise(z)
    np.testing.assert_array_equal(y, (z >= O.Y1_THRESHOLD).astype(np.int64) + (z >= O.Y2_THRESHOLD))


def test_recover_scalar_port_equals_vectorised(golden, keys):
    key, nonce = keys
    z = golden["arrays"]["Znoisy16_2"]
    zc = np.clip(z, np.float16(-8), np.float16(8))
    a = types.SimpleNamespace(key=key, nonce=nonce, l=1, message_length=256)
    assert O.recover_exactracted_message_scalar(zc, a) == O.recover_exactracted_message(zc, a)


def test_bit_accuracy(golden):
    for c in golden["extract"]["cases"]["_bit_accuracy"]:
        ob, acc = O.calculate_bit_accuracy(c["hex"], c["bin"])
        assert ob == c["original_bin"] and acc == c["accuracy"]


# ------------------------------------------------------------------ DDIM self-consistency (the bytecode pin lives in test_ddim_bytecode_golden.py)
def test_ddim_closed_form_equals_coefficients():
    ac = O.sd_alphas_cumprod()
    rng = np.random.RandomState(0)
    x, e = rng.standard_normal(1000), rng.standard_normal(1000)
    for t, tp in ((981, 961), (21, 1), (500, 480)):
        a, b = O.ddim_coefficients(ac[t], ac[tp])
        np.testing.assert_allclose(O.backward_ddim(x, ac[t], ac[tp], e), a * x + b * e, rtol=0, atol=1e-12)
        a2, b2 = O.ddim_coefficients(ac[tp], ac[t])          # inversion = the same move with alphas swapped
        x2 = a2 * (a * x + b * e) + b2 * e
        np.testing.assert_allclose(x2, x, atol=1e-12)        # exact inverse when eps is held fixed


def test_ddim_schedule_shapes():
    s = O.ddim_schedule(50, inverse=False)
    si = O.ddim_schedule(50, inverse=True)
    assert [t for t, _, _ in s] == list(range(981, 0, -20))
    assert [t for t, _, _ in si] == list(range(1, 1000, 20))
    assert len(O.ddim_schedule(30, inverse=True)) == 30


def test_philox_reference_vectors():
    # Random123 known answers for Philox4x32-10
    r = O.philox4x32_10(np.uint32(0), np.uint32(0), np.uint32(0), np.uint32(0), 0, 0)
    assert [int(v) for v in r] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    r = O.philox4x32_10(np.uint32(0xffffffff), np.uint32(0xffffffff), np.uint32(0xffffffff), np.uint32(0xffffffff), 0xffffffff, 0xffffffff)
    assert [int(v) for v in r] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    r = O.philox4x32_10(np.uint32(0x243f6a88), np.uint32(0x85a308d3), np.uint32(0x13198a2e), np.uint32(0x03707344), 0xa4093822, 0x299f31d0)
    assert [int(v) for v in r] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    # ... and for philox4x32_R(7), the variant the build's throughput stream runs (Random123 kat_vectors)
    kat7 = [((0, 0, 0, 0), (0, 0), [0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48]),
            ((0xffffffff,) * 4, (0xffffffff,) * 2, [0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662]),
            ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), [0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a])]
    for c, k, want in kat7:
        assert [int(v) for v in O.philox4x32(*(np.uint32(x) for x in c), k[0], k[1], rounds=7)] == want
    assert O.PHILOX_ROUNDS == 7
    u = O.philox_uniform(7, 3, 2, 100)
    assert u.shape == (2, 100) and (u >= 0).all() and (u < 1).all()
