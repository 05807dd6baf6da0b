"""GPU parity (through the C ABI): the image-side kernels of csrc/gswm_image.hip against oracle/image_oracle.py and, where PIL is
importable, against PIL itself -- bit-exact (byte / integer work)."""
import io
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import image_oracle as IO  # noqa: E402


def synth(h, w, seed, noise=20.0):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    base = np.stack([127 + 100 * np.sin(x / 17.0 + c) * np.cos(y / 23.0 - c) for c in range(3)], -1)
    return np.clip(base + rng.normal(0, noise, (h, w, 3)), 0, 255).astype(np.uint8)


def batch(h, w, n, seed):
    return np.stack([synth(h, w, seed + i) for i in range(n)])


@pytest.fixture(scope="module")
def im():
    import gswm_amd
    from gswm_amd import imaging
    return imaging


@pytest.mark.parametrize("hw,size", [((64, 64), (32, 32)), ((64, 48), (100, 37)), ((100, 130), (130, 100)), ((37, 41), (41, 37)), ((96, 96), (96, 50)),
                                     ((96, 96), (50, 96)), ((60, 40), (128, 128)), ((50, 50), (50, 50)), ((9, 7), (3, 2)), ((5, 5), (40, 1))])
def test_resize_u8_equals_oracle(im, hw, size):
    imgs = batch(*hw, 3, seed=hw[0] + size[0])
    got = im.resize_lanczos(torch.from_numpy(imgs).cuda(), size).cpu().numpy()
    for i in range(3):
        assert np.array_equal(got[i], IO.resize_lanczos(imgs[i], size))


def test_resize_512_full_size_against_pil(im):
    PILImage = pytest.importorskip("PIL.Image")
    imgs = batch(512, 512, 2, seed=7)
    dev = torch.from_numpy(imgs).cuda()
    for size in ((256, 256), (768, 768), (384, 512), (512, 512)):
        got = im.resize_lanczos(dev, size).cpu().numpy()
        for i in range(2):
            assert np.array_equal(got[i], np.asarray(PILImage.fromarray(imgs[i]).resize(size, PILImage.Resampling.LANCZOS)))


@pytest.mark.parametrize("size", [(32, 48), (64, 64), (20, 64), (64, 20)])
def test_resize_fused_tensor_outputs(im, size):
    """out='f16' is the reference's VAE-encoder input (ToTensor -> fp16 -> 2x-1), out='f32' is ToTensor; both bit-exact."""
    imgs = batch(64, 64, 2, seed=11)
    dev = torch.from_numpy(imgs).cuda()
    f16 = im.resize_lanczos(dev, size, out="f16")
    f32 = im.resize_lanczos(dev, size, out="f32")
    assert f16.shape == (2, 3, size[1], size[0]) and f16.dtype == torch.float16
    for i in range(2):
        r = IO.resize_lanczos(imgs[i], size)
        assert np.array_equal(f16[i].cpu().numpy(), IO.normalise_like_reference(r))
        ref32 = (torch.from_numpy(r).permute(2, 0, 1).float() / 255.0).numpy()
        assert np.array_equal(f32[i].cpu().numpy(), ref32)


def test_to_tensor_all_byte_values(im):
    v = np.arange(256, dtype=np.uint8).reshape(1, 16, 16, 1).repeat(3, axis=3)
    got = im.to_tensor(torch.from_numpy(v).cuda(), out="f16").cpu()
    x = (torch.from_numpy(v[0]).permute(2, 0, 1).float() / 255.0).to(torch.float16)
    assert torch.equal(got[0], 2.0 * x - 1.0)


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32, torch.bfloat16])
def test_tensor_to_image(im, dtype):
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, 3, 24, 40, generator=g).to(dtype)
    got = im.tensor_to_image(x.cuda()).cpu().numpy()
    ref = (x.permute(0, 2, 3, 1).float().numpy() * 255).round().astype("uint8")           # numpy_to_pil
    assert np.array_equal(got, ref)
    lat = (torch.randn(2, 3, 24, 40, generator=g) * 1.5).to(dtype)
    got = im.tensor_to_image(lat.cuda(), denormalise=True).cpu().numpy()
    ref = ((lat / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).float().numpy() * 255).round().astype("uint8")
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("hw", [(64, 64), (48, 80), (33, 47), (17, 16), (100, 130), (18, 16), (24, 16), (2, 3), (1, 1), (7, 9), (5, 4)])
@pytest.mark.parametrize("q", [10, 75, 95])
def test_jpeg_equals_oracle(im, hw, q):
    imgs = batch(*hw, 2, seed=hw[0] * 131 + hw[1] + q)
    got = im.jpeg_roundtrip(torch.from_numpy(imgs).cuda(), q).cpu().numpy()
    for i in range(2):
        assert np.array_equal(got[i], IO.jpeg_roundtrip(imgs[i], q))


@pytest.mark.parametrize("q", [1, 10, 50, 100])
def test_jpeg_512_against_pil(im, q):
    PILImage = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(q)
    imgs = np.stack([synth(512, 512, seed=q), rng.integers(0, 256, (512, 512, 3), dtype=np.uint8)])
    got = im.jpeg_roundtrip(torch.from_numpy(imgs).cuda(), q).cpu().numpy()
    for i in range(2):
        buf = io.BytesIO()
        PILImage.fromarray(imgs[i]).save(buf, format="JPEG", quality=q)
        buf.seek(0)
        assert np.array_equal(got[i], np.asarray(PILImage.open(buf).convert("RGB")))


def test_jpeg_fused_f16_output(im):
    imgs = batch(48, 64, 2, seed=3)
    got = im.jpeg_roundtrip(torch.from_numpy(imgs).cuda(), 10, out="f16").cpu().numpy()
    for i in range(2):
        assert np.array_equal(got[i], IO.normalise_like_reference(IO.jpeg_roundtrip(imgs[i], 10)))


@pytest.mark.parametrize("f", [0.0, 0.3, 1.0, 1.7, 5.0, 16.0])
def test_brightness_contrast(im, f):
    imgs = batch(40, 56, 3, seed=int(f * 10))
    dev = torch.from_numpy(imgs).cuda()
    b = im.pointwise(dev, "brightness", f).cpu().numpy()
    c = im.pointwise(dev, "contrast", f).cpu().numpy()
    for i in range(3):
        assert np.array_equal(b[i], IO.enhance_brightness(imgs[i], f))
        assert np.array_equal(c[i], IO.enhance_contrast(imgs[i], f))


def test_invert_gray_flips(im):
    imgs = batch(33, 47, 2, seed=9)
    dev = torch.from_numpy(imgs).cuda()
    assert np.array_equal(im.pointwise(dev, "invert").cpu().numpy(), 255 - imgs)
    assert np.array_equal(im.pointwise(dev, "horizontal_flip").cpu().numpy(), imgs[:, :, ::-1])
    assert np.array_equal(im.pointwise(dev, "vertical_flip").cpu().numpy(), imgs[:, ::-1])
    g = im.pointwise(dev, "togray").cpu().numpy()
    for i in range(2):
        assert np.array_equal(g[i], IO.rgb_to_l(imgs[i])[..., None].repeat(3, axis=2))


def test_noise_statistics_and_determinism(im):
    imgs = np.full((4, 128, 128, 3), 128, dtype=np.uint8)
    dev = torch.from_numpy(imgs).cuda()
    a = im.pointwise(dev, "noise", 0.1, seed=5).cpu().numpy().astype(np.float64)
    b = im.pointwise(dev, "noise", 0.1, seed=5).cpu().numpy().astype(np.float64)
    assert np.array_equal(a, b)
    d = (a - 128.0) / 255.0
    assert abs(d.mean()) < 2e-3 and abs(d.std() - 0.1) < 3e-3
    # batch-split independence: image 2 of the batch == image 0 of a launch with image_index0 = 2
    from gswm_amd import imaging
    solo = imaging.pointwise(dev[2:3], "noise", 0.1, seed=5, image_index0=2).cpu().numpy()
    assert np.array_equal(solo[0], a[2].astype(np.uint8))
    # channels are independent draws
    assert abs(np.corrcoef(d[..., 0].ravel(), d[..., 1].ravel())[0, 1]) < 0.02


def test_apply_distortion_dispatch(im):
    imgs = batch(64, 64, 2, seed=21)
    dev = torch.from_numpy(imgs).cuda()
    got = im.apply_distortion(dev, "compression", 0.9).cpu().numpy()          # relative 0.9 -> QF 10
    assert np.array_equal(got[0], IO.jpeg_roundtrip(imgs[0], 10))
    got = im.apply_distortion(dev, "scaling", 0.5).cpu().numpy()
    assert np.array_equal(got[1], IO.resize_lanczos(imgs[1], (32, 32)))
    with pytest.raises(ValueError):
        im.apply_distortion(dev, "rotation", 0.5)


def test_bad_arguments(im):
    with pytest.raises(ValueError):
        im.resize_lanczos(torch.zeros(1, 8, 8, 3), (4, 4))                   # host tensor
    with pytest.raises(ValueError):
        im.jpeg_roundtrip(torch.zeros(1, 8, 8, 4, dtype=torch.uint8).cuda(), 10)


@pytest.mark.parametrize("hw", [(64, 64), (33, 47), (5, 9), (1, 7), (20, 3), (512, 512)])
@pytest.mark.parametrize("radius", [0, 1, 3, 10, 20, 1.7])
def test_gaussian_blur_equals_oracle_and_pil(im, hw, radius):
    imgs = np.random.default_rng(int(radius * 10) + hw[0]).integers(0, 256, (2, *hw, 3), dtype=np.uint8)
    got = im.gaussian_blur(torch.from_numpy(imgs).cuda(), radius).cpu().numpy()
    for i in range(2):
        assert np.array_equal(got[i], IO.gaussian_blur(imgs[i], radius))
    PILImage = pytest.importorskip("PIL.Image")
    from PIL import ImageFilter
    assert np.array_equal(got[0], np.asarray(PILImage.fromarray(imgs[0]).filter(ImageFilter.GaussianBlur(radius))))


def test_apply_distortion_blurring(im):
    imgs = batch(48, 64, 2, seed=31)
    got = im.apply_distortion(torch.from_numpy(imgs).cuda(), "blurring", 0.25).cpu().numpy()       # relative 0.25 -> radius int(5.0)
    assert np.array_equal(got[1], IO.gaussian_blur(imgs[1], 5))


@pytest.mark.parametrize("hw", [(20, 257), (40, 300), (16, 272), (33, 513), (17, 255), (48, 768)])
def test_jpeg_tile_boundaries_against_pil(im, hw):
    """Widths around the 256-pixel tile of gsw_jpeg_blocks_kernel: fast (16-byte) and clamped (byte) staging paths in one image."""
    PILImage = pytest.importorskip("PIL.Image")
    imgs = np.random.default_rng(hw[1]).integers(0, 256, (2, *hw, 3), dtype=np.uint8)
    got = im.jpeg_roundtrip(torch.from_numpy(imgs).cuda(), 35).cpu().numpy()
    for i in range(2):
        buf = io.BytesIO()
        PILImage.fromarray(imgs[i]).save(buf, format="JPEG", quality=35)
        buf.seek(0)
        assert np.array_equal(got[i], np.asarray(PILImage.open(buf).convert("RGB")))


def test_random_sizes_resize_blur_jpeg_against_pil(im):
    """A seeded sweep of odd geometries through the three bit-exact image kernels."""
    PILImage = pytest.importorskip("PIL.Image")
    from PIL import ImageFilter
    rng = np.random.default_rng(2024)
    for _ in range(12):
        h, w = int(rng.integers(1, 90)), int(rng.integers(1, 400))
        img = rng.integers(0, 256, (1, h, w, 3), dtype=np.uint8)
        dev = torch.from_numpy(img).cuda()
        pil = PILImage.fromarray(img[0])
        size = (int(rng.integers(1, 300)), int(rng.integers(1, 120)))
        assert np.array_equal(im.resize_lanczos(dev, size).cpu().numpy()[0], np.asarray(pil.resize(size, PILImage.Resampling.LANCZOS))), (h, w, size)
        r = int(rng.integers(0, 21))
        assert np.array_equal(im.gaussian_blur(dev, r).cpu().numpy()[0], np.asarray(pil.filter(ImageFilter.GaussianBlur(r)))), (h, w, r)
        q = int(rng.integers(1, 101))
        buf = io.BytesIO()
        pil.save(buf, format="JPEG", quality=q)
        buf.seek(0)
        assert np.array_equal(im.jpeg_roundtrip(dev, q).cpu().numpy()[0], np.asarray(PILImage.open(buf).convert("RGB"))), (h, w, q)
