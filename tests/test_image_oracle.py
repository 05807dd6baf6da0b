"""CPU (no GPU): pins oracle/image_oracle.py -- the restatement of Pillow's Lanczos resampler / Blend / Convert and of libjpeg's lossy
stages -- bit for bit against the PIL that is importable here (the third-party code the reference calls at extract.py:31-37 and
distortions:131-233), and checks the pure-host C-ABI functions (gsw_lanczos_plan, gsw_jpeg_quant_tables) against the oracle."""
import io
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import image_oracle as IO  # noqa: E402

PIL = pytest.importorskip("PIL")
from PIL import Image, ImageEnhance  # noqa: E402


def synth(h, w, seed, noise=20.0):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    base = np.stack([127 + 100 * np.sin(x / 17.0 + c) * np.cos(y / 23.0 - c) for c in range(3)], -1)
    return np.clip(base + rng.normal(0, noise, (h, w, 3)), 0, 255).astype(np.uint8)


def pil_jpeg(img, q):
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, format="JPEG", quality=q)
    buf.seek(0)
    return np.asarray(Image.open(buf).convert("RGB"))


@pytest.mark.parametrize("hw,size", [((64, 64), (32, 32)), ((64, 48), (100, 37)), ((128, 128), (64, 64)), ((100, 130), (130, 100)),
                                     ((37, 41), (41, 37)), ((96, 96), (96, 50)), ((60, 40), (128, 128)), ((50, 50), (50, 50)),
                                     ((9, 7), (3, 2)), ((5, 5), (40, 1))])
def test_resize_lanczos_matches_pil(hw, size):
    img = synth(*hw, seed=hw[0] * 1000 + size[0])
    ref = np.asarray(Image.fromarray(img).resize(size, Image.Resampling.LANCZOS))
    assert np.array_equal(IO.resize_lanczos(img, size), ref)


def test_resize_512_to_256_and_768_matches_pil():
    img = synth(512, 512, seed=5)
    for size in ((256, 256), (768, 768), (384, 512)):
        assert np.array_equal(IO.resize_lanczos(img, size), np.asarray(Image.fromarray(img).resize(size, Image.Resampling.LANCZOS)))


def test_normalise_matches_torch_chain():
    torch = pytest.importorskip("torch")
    v = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, axis=2)
    x = (torch.from_numpy(v).permute(2, 0, 1).float() / 255.0).to(torch.float16)       # ToTensor, .to(float16)
    ref = (2.0 * x - 1.0).numpy()                                                        # img_to_latents
    assert np.array_equal(IO.normalise_like_reference(v), ref)


@pytest.mark.parametrize("hw", [(64, 64), (48, 80), (33, 47), (17, 16), (100, 130), (18, 16), (24, 16), (2, 3), (1, 1), (7, 9), (5, 4)])
@pytest.mark.parametrize("q", [10, 75, 95])
def test_jpeg_roundtrip_matches_pil(hw, q):
    img = synth(*hw, seed=hw[0] * 131 + hw[1] + q)
    assert np.array_equal(IO.jpeg_roundtrip(img, q), pil_jpeg(img, q))


@pytest.mark.parametrize("q", [1, 10, 25, 50, 90, 100])
def test_jpeg_roundtrip_noise_image_512(q):
    img = np.random.default_rng(q).integers(0, 256, (128, 144, 3), dtype=np.uint8) if q != 10 else synth(512, 512, seed=3)
    assert np.array_equal(IO.jpeg_roundtrip(img, q), pil_jpeg(img, q))


@pytest.mark.parametrize("q", [1, 10, 49, 50, 75, 100])
def test_quant_tables_match_pil_and_cabi(q):
    buf = io.BytesIO()
    Image.fromarray(synth(16, 16, 0)).save(buf, format="JPEG", quality=q)
    buf.seek(0)
    tabs = Image.open(buf).quantization
    ql, qc = IO.jpeg_quant_tables(q)
    # Pillow hands the tables back in the file's (zig-zag) order
    zz = sorted(range(64), key=lambda i: ((i // 8 + i % 8), (i % 8 if (i // 8 + i % 8) % 2 else i // 8)))
    for t, mine in ((tabs[0], ql), (tabs[1], qc)):
        t = list(t)
        assert sorted(t) == sorted(mine.ravel().tolist())
        assert t == mine.ravel()[zz].tolist() or t == mine.ravel().tolist()
    import gswm_amd
    from gswm_amd import imaging
    l2, c2 = imaging.jpeg_quant_tables(q)
    assert np.array_equal(l2, ql) and np.array_equal(c2, qc)


@pytest.mark.parametrize("sizes", [(512, 256), (64, 100), (100, 64), (37, 41), (768, 512), (5, 40), (40, 1)])
def test_cabi_lanczos_plan_equals_oracle(sizes):
    import gswm_amd
    from gswm_amd import imaging
    b, k, ks = imaging.lanczos_plan_host(*sizes)
    ob, ok, oks = IO.lanczos_coeffs(*sizes)
    assert ks == oks and np.array_equal(b, ob) and np.array_equal(k, ok)


@pytest.mark.parametrize("f", [0.0, 0.3, 1.0, 1.7, 5.0, 16.0])
def test_enhance_matches_pil(f):
    img = synth(40, 56, seed=int(f * 10))
    pi = Image.fromarray(img)
    assert np.array_equal(IO.enhance_brightness(img, f), np.asarray(ImageEnhance.Brightness(pi).enhance(f)))
    assert np.array_equal(IO.enhance_contrast(img, f), np.asarray(ImageEnhance.Contrast(pi).enhance(f)))
    assert np.array_equal(IO.rgb_to_l(img), np.asarray(pi.convert("L")))


def test_relative_strength_table():
    import gswm_amd
    from gswm_amd import imaging
    assert imaging.relative_strength_to_absolute(0.9, "compression") == pytest.approx(10.0)     # QF 10 of BASELINE config 4
    assert imaging.relative_strength_to_absolute(0.0, "compression") == 100
    assert imaging.relative_strength_to_absolute(0.5, "scaling") == 0.5
    assert imaging.relative_strength_to_absolute(1.0, "brightness") == 16


@pytest.mark.parametrize("hw", [(64, 64), (33, 47), (5, 9), (1, 7), (20, 3)])
@pytest.mark.parametrize("radius", [0, 1, 2, 3, 5, 10, 20, 0.5, 1.7])
def test_gaussian_blur_matches_pil(hw, radius):
    from PIL import ImageFilter
    img = np.random.default_rng(int(radius * 10) + hw[0]).integers(0, 256, (*hw, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).filter(ImageFilter.GaussianBlur(radius)))
    assert np.array_equal(IO.gaussian_blur(img, radius), ref)


@pytest.mark.parametrize("radius", [1, 2, 3, 5, 7, 10, 13, 20, 0.5, 1.7, 30])
def test_cabi_gaussian_blur_params_equal_oracle(radius):
    import gswm_amd
    from gswm_amd import imaging
    assert imaging.gaussian_blur_params(radius) == IO.box_blur_params(IO.gaussian_box_radius(radius))
