"""CPU (no GPU): the C-ABI library loads and exports every symbol include/gswm.h declares; host-side logic of the package
(message / key handling, error mapping, sharding maths); the product path refuses to run without a GPU instead of
falling back to a CPU path."""
import ctypes
import os
import re
import types

import numpy as np
import pytest
import torch

import gs_oracle as O
from conftest import README_KEY, README_NONCE, ROOT

import gswm_amd
from gswm_amd import _native as N, codec, extract as gext, dist as gdist


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "gswm.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gsw_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = N.lib()
    syms = header_symbols()
    assert len(syms) >= 11
    for s in syms:
        assert hasattr(lib, s), f"libgswm.so does not export {s}"
    assert sorted(N.exported_symbols()) == syms            # the ctypes prototypes cover the whole header
    assert lib.gsw_version() == 500
    assert lib.gsw_strerror(0) == b"ok" and b"IndexError" in lib.gsw_strerror(N.GSW_ERR_RAGGED)


def test_constants_match_header():
    txt = open(os.path.join(ROOT, "include", "gswm.h")).read()
    for name in ("GSW_F32", "GSW_F16", "GSW_BF16", "GSW_F64", "GSW_OK", "GSW_ERR_BAD_ARG", "GSW_ERR_UNSUPPORTED", "GSW_ERR_RAGGED", "GSW_ERR_HIP", "GSW_WARN_NO_RECORDS"):
        m = re.search(rf"\b{name}\s*=\s*(\d+)", txt)
        assert m and int(m.group(1)) == getattr(N, name), name
    for name in ("GSW_EMBED_FAST_F32", "GSW_FLAG_SATURATED", "GSW_FLAG_NAN", "GSW_MSG_INLINE_MAX"):
        m = re.search(rf"#define\s+{name}\s+(\d+)", txt)
        assert m and int(m.group(1)) == getattr(N, name), name


def test_argument_validation_needs_no_gpu():
    """Status codes that are decided before any HIP call (no compute is attempted on this box)."""
    lib = N.lib()
    key, nonce = bytes(32), bytes(16)
    assert lib.gsw_keystream(key, nonce, None, 0, None) == N.GSW_OK
    assert lib.gsw_keystream(None, nonce, None, 0, None) == N.GSW_ERR_BAD_ARG
    assert lib.gsw_embed(key, nonce, b"x", 1, None, 0, 0, None, 0, 1, 16384, 0, None) == N.GSW_ERR_BAD_ARG       # null output
    assert lib.gsw_embed(key, nonce, b"x", 1, None, 0, 0, ctypes.c_void_p(16), 0, 1, 16383, 0, None) == N.GSW_ERR_BAD_ARG  # n % 4
    assert lib.gsw_embed(key, nonce, b"x", 1, None, 0, 0, ctypes.c_void_p(16), 9, 1, 16384, 0, None) == N.GSW_ERR_BAD_ARG  # dtype
    assert lib.gsw_embed(key, nonce, b"x", 1, None, 0, 0, ctypes.c_void_p(16), 0, 0, 16384, 0, None) == N.GSW_OK           # empty batch
    p = ctypes.c_void_p(16)
    assert lib.gsw_extract(p, 1, key, nonce, 1000, p, None, p, 1, 16384, None) == N.GSW_ERR_RAGGED
    assert lib.gsw_extract(p, 1, key, nonce, 32, p, None, p, 1, 1700, None) == N.GSW_ERR_RAGGED                 # 1704 % 32
    assert lib.gsw_extract(p, 1, key, nonce, 256, p, None, p, 0, 16384, None) == N.GSW_OK
    assert lib.gsw_extract(None, 1, key, nonce, 256, p, None, p, 1, 16384, None) == N.GSW_ERR_BAD_ARG
    assert lib.gsw_ddim_step(p, p, p, 1.0, 0.0, 0, 0, None) == N.GSW_OK
    assert lib.gsw_ddim_step(p, p, p, 1.0, 0.0, 3, 8, None) == N.GSW_ERR_BAD_ARG                                # f64 unsupported
    assert lib.gsw_ddim_step_cfg(p, p, None, p, 1.0, 0.0, 7.5, 1, 8, None) == N.GSW_ERR_BAD_ARG
    with pytest.raises(IndexError):
        N.check(N.GSW_ERR_RAGGED)
    with pytest.raises(ValueError):
        N.check(N.GSW_ERR_BAD_ARG)
    with pytest.raises(N.GswError):
        N.check(N.GSW_ERR_UNSUPPORTED)


def test_no_cpu_fallback():
    """CPU tensors are rejected: the hot path exists only as HIP kernels."""
    key, nonce = bytes.fromhex(README_KEY), bytes.fromhex(README_NONCE)
    z = torch.zeros(1, 4, 64, 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        codec.extract_batch(z, key, nonce, 256)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        codec.ddim_step(z, z, 1.0, 0.0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        codec.embed_batch(key, nonce, b"k" * 32, 1, (4, 64, 64), out=z)
    # nothing under the product package imports the oracle
    pkg = os.path.join(ROOT, "a-watermark-for-diffusion-models_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(import|from)\s+\S*oracle", src, flags=re.M), f
                assert "gs_oracle" not in src and "importlib" not in src.replace("import importlib\n        return importlib.import_module", ""), f
            elif f.endswith((".hip", ".h", ".inc", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"#\s*include[^\n]*oracle", src), f


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(N, "_lib", None)
    monkeypatch.setattr(N, "LIB_PATH", "/nonexistent/libgswm.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        N.lib()


def test_host_message_and_key_handling():
    # gs_insert.py:9-20 / nodes.py:68-76
    assert codec.pad_message("lthero", 32) == b"lthero" + b"\0" * 26 == O.pad_message("lthero", 32)
    long = "水印测试-watermark-éè-" + "x" * 40
    assert codec.pad_message(long, 32) == long.encode()[:32] == O.pad_message(long, 32)
    assert codec.pad_message("abc", 128)[:3] == b"abc" and len(codec.pad_message("abc", 128)) == 128
    r = codec.pad_message("", 32)
    assert len(r) == 32 and r != codec.pad_message("", 32)                    # os.urandom
    assert codec.pad_message(12345, 4) == b"1234"                               # nodes.py:69 str(message)
    # gs_insert.py:27-42
    k, n = codec.resolve_key_nonce(README_KEY, README_NONCE)
    assert (k, n) == (bytes.fromhex(README_KEY), bytes.fromhex(README_NONCE))
    k, n = codec.resolve_key_nonce(README_KEY, "")
    assert n == bytes.fromhex(README_KEY)[8:24] == O.resolve_key_nonce(README_KEY, "")[1]
    k, n = codec.resolve_key_nonce("", "")
    assert len(k) == 32 and len(n) == 16
    with pytest.raises(ValueError):
        codec.resolve_key_nonce("zz", "00")
    for tot in (0, 1, 2047, 2048, 4096, 8192, 16384, 32768, 36864, 1 << 20):
        assert codec.choose_watermark_length(tot) == O.choose_watermark_length(tot)
    with pytest.raises(ValueError):
        codec.keystream(b"x" * 31, b"y" * 16, 64)
    with pytest.raises(ValueError):
        codec.extract_batch(torch.zeros(1, 16), b"x" * 32, b"y" * 12, 8)


def test_bit_accuracy_twin(golden):
    for c in golden["extract"]["cases"]["_bit_accuracy"]:
        ob, acc = gext.calculate_bit_accuracy(c["hex"], c["bin"])
        assert ob == c["original_bin"] and acc == c["accuracy"]
    assert codec.bits_to_str(np.array([0x6C, 0x01], np.uint8)) == "0110110000000001"


def test_shard_range_and_param_packing():
    for total in (0, 1, 7, 8, 64, 255, 256):
        for world in (1, 2, 3, 8):
            spans = [gdist.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    p = {"key": bytes(range(32)), "nonce": bytes(range(16)), "message": b"lthero" + b"\0" * 26, "seed": 2 ** 63 + 5, "height": 768, "width": 512}
    q = gdist.unpack_params(gdist.pack_params(p))
    assert q == p
    assert gdist.broadcast_params(p) == p                                       # no process group: identity
    with pytest.raises(ValueError):
        gdist.pack_params(dict(p, message=b"x" * 5000))


@pytest.mark.parametrize("seed", [0, 1, 42, 2**32 - 1])
def test_mt19937_seed_equals_numpy_legacy_seeding(seed):
    """gsw_mt19937_seed (pure host) == the key np.random.RandomState(seed) starts from (nodes.py:52-53)."""
    import numpy as np
    import gswm_amd
    from gswm_amd import codec
    st = np.random.RandomState(seed).get_state()
    assert st[2] == 624 and np.array_equal(codec.mt19937_seed(seed), st[1])
    with pytest.raises(ValueError):
        codec.mt19937_seed(2**32)


def test_argument_validation_of_eps_model_and_image_entry_points():
    """Status codes of the UNet / VAE / image entry points that are decided before any HIP call."""
    lib = N.lib()
    p = ctypes.c_void_p(64)
    BAD, UNS = N.GSW_ERR_BAD_ARG, N.GSW_ERR_UNSUPPORTED
    # attention: null operand, head width, tile divisibility, strides
    assert lib.gsw_attention(None, p, p, p, 1, 5, 64, 256, 256, 256, 320, 320, 320, 0.125, 1, None) == BAD
    assert lib.gsw_attention(p, p, p, p, 1, 5, 48, 256, 256, 256, 240, 240, 240, 0.125, 1, None) == UNS          # head_dim 48
    assert lib.gsw_attention(p, p, p, p, 1, 5, 64, 256, 100, 77, 320, 320, 320, 0.125, 1, None) == UNS           # Sk % 8
    assert lib.gsw_attention(p, p, p, p, 1, 5, 64, 256, 128, 129, 320, 320, 320, 0.125, 1, None) == BAD          # valid keys > keys
    assert lib.gsw_attention(p, p, p, p, 1, 5, 64, 256, 128, 77, 316, 320, 320, 0.125, 1, None) == UNS           # row stride < H * d
    assert lib.gsw_attention(p, p, p, p, 1, 5, 64, 256, 128, 77, 320, 320, 320, 0.125, 0, None) == BAD           # fp32
    # convolutions on padded-flat activations
    assert lib.gsw_conv_pf(p, p, None, None, 0, None, p, 1, 8, 8, 60, 64, 3, 1, 60, 1, None) == UNS                 # C % 64
    assert lib.gsw_conv_pf(p, p, None, None, 0, None, p, 1, 8, 8, 64, 64, 5, 1, 64, 1, None) == BAD                 # 5x5
    assert lib.gsw_conv_pf(p, p, None, p, 60, None, p, 1, 8, 8, 64, 64, 3, 1, 64, 1, None) == BAD                  # rowbias stride < N
    assert lib.gsw_conv3x3_res_pf(p, p, None, None, 0, None, p, 1, 8, 8, 64, 64, p, 64, None, 0, 1, None) == UNS    # N % 160
    assert lib.gsw_conv_up2x_pf(p, p, None, p, 1, 8, 8, 64, 64, 1, None) == UNS                                  # N % 160
    assert lib.gsw_conv_up2x_pf(None, p, None, p, 1, 8, 8, 64, 160, 1, None) == BAD
    # the matmul engine and the row softmax
    assert lib.gsw_gemm_strided(p, 60, p, 64, None, None, 160, p, 160, 4, 64, 160, 0, 0, 0, 1, None) == UNS      # ldx < K
    assert lib.gsw_gemm_strided(p, 64, p, 64, None, None, 160, p, 160, 4, 64, 164, 0, 0, 0, 1, None) == UNS      # N % 8
    assert lib.gsw_gemm_strided(p, 64, p, 64, None, None, 160, p, 80, 4, 64, 168, 1, 0, 0, 1, None) == UNS       # GEGLU needs N % 160
    assert lib.gsw_gemm_strided(p, 64, p, 64, None, None, 160, p, 100, 4, 64, 160, 0, 0, 0, 1, None) == UNS      # ldy % 8
    assert lib.gsw_gemm_strided(p, 64, p, 64, None, None, 160, p, 96, 4, 64, 160, 0, 0, 0, 1, None) == BAD       # ldy < N
    assert lib.gsw_gemm(None, p, None, None, p, 4, 64, 160, 0, 0, 0, 1, None) == BAD
    assert lib.gsw_gemm(p, p, ctypes.c_void_p(72), None, p, 4, 64, 160, 0, 0, 0, 1, None) == BAD                 # bias: 16-byte aligned (LDS-DMA pieces)
    assert lib.gsw_gemm_qkv(p, p, None, p, p, 256, 320, 600, 960, 128, 1, None) == UNS                           # a column tile must be of one kind
    assert lib.gsw_gemm_qkv(p, p, None, p, None, 256, 320, 640, 960, 128, 1, None) == BAD                        # no transposed output
    assert lib.gsw_gemm_qkv(p, p, None, p, p, 256, 320, 640, 960, 100, 1, None) == UNS                           # S % 8
    # (ABI 0.5.0: the thread-local one-shot requests and the thread-local workspace are gone -- their validation lives on the GswMmExtras checks below)
    for gone in ("gsw_mm_set_workspace", "gsw_mm_next_colstats", "gsw_mm_last_colstats", "gsw_mm_next_rowstats", "gsw_mm_last_rowstats", "gsw_linear", "gsw_attention_hd64"):
        assert not hasattr(lib, gone), gone
    assert lib.gsw_groupnorm_pf_cs(p, None, 0, p, 64, 1, 4, None, 0, 0, 0, p, p, p, p, 1, 8, 8, 330, 33, 1e-5, 1, 0, 1, None) == UNS  # C % 8
    assert lib.gsw_groupnorm_pf_cs(p, None, 0, p, 64, 1, 4, None, 0, 0, 0, p, p, p, p, 1, 8, 8, 96, 32, 1e-5, 1, 0, 1, None) == UNS   # odd group width (column pairs)
    assert lib.gsw_groupnorm_pf_cs(p, None, 0, p, 48, 1, 4, None, 0, 0, 0, p, p, p, p, 1, 8, 8, 320, 32, 1e-5, 1, 0, 1, None) == UNS  # 64 pixels per image, 48-row blocks
    assert lib.gsw_groupnorm_pf_cs(p, None, 0, p, 64, 3, 4, None, 0, 0, 0, p, p, p, p, 1, 8, 8, 320, 32, 1e-5, 1, 0, 1, None) == BAD  # parity count
    # LayerNorm folded into the consuming GEMM
    assert lib.gsw_gemm_ln(p, None, p, p, p, p, 256, 320, 640, 0, 0, 1, None) == BAD                              # no row statistics
    assert lib.gsw_gemm_ln(p, p, p, p, p, p, 252, 320, 640, 0, 0, 1, None) == UNS                                 # M % 8
    assert lib.gsw_gemm_ln(p, p, p, p, p, p, 256, 320, 648, 1, 0, 1, None) == UNS                                 # GEGLU needs N % 160
    assert lib.gsw_gemm_ln(p, p, p, p, p, p, 256, 320, 640, 3, 0, 1, None) == BAD                                 # no token scatter
    assert lib.gsw_gemm_ln(p, p, p, ctypes.c_void_p(68), p, p, 256, 320, 640, 0, 0, 1, None) == BAD               # u alignment
    assert lib.gsw_ln_rowstats_finish(None, 4, 256, 320, 1e-5, p, None) == BAD and lib.gsw_ln_rowstats_finish(p, 0, 256, 320, 1e-5, p, None) == BAD
    # GswMmExtras: validated before anything touches the device, nothing armed, nothing left behind
    ex = N.GswMmExtras()
    ex.colstats_dev, ex.colstats_capacity = 68, 16
    assert lib.gsw_gemm_ex(p, 64, p, 64, None, None, 160, p, 160, 4, 64, 160, 0, 0, 0, 1, ctypes.byref(ex), None) == BAD         # colstats alignment
    ex = N.GswMmExtras()
    ex.workspace_dev, ex.workspace_bytes, ex.max_splits = p.value, 1024, 65
    assert lib.gsw_gemm_ex(p, 64, p, 64, None, None, 160, p, 160, 4, 64, 160, 0, 0, 0, 1, ctypes.byref(ex), None) == BAD         # max_splits
    ex = N.GswMmExtras()
    assert lib.gsw_gemm_ex(p, 64, p, 64, None, None, 160, p, 96, 4, 64, 160, 0, 0, 0, 1, ctypes.byref(ex), None) == BAD          # ldy < N, as gsw_gemm_strided
    assert lib.gsw_gemm_ln_ex(p, None, p, p, p, p, 256, 320, 640, 0, 0, 1, ctypes.byref(ex), None) == BAD
    assert lib.gsw_conv_pf_ex(None, p, None, None, 0, None, p, 1, 8, 8, 64, 128, 3, 1, 64, 1, ctypes.byref(ex), None) == BAD
    assert lib.gsw_conv3x3_res_pf_ex(p, p, None, None, 0, None, p, 1, 8, 8, 64, 64, None, 0, None, 0, 1, ctypes.byref(ex), None) == UNS   # N < 128
    assert lib.gsw_conv_up2x_pf_ex(p, p, None, p, 1, 8, 8, 60, 128, 1, ctypes.byref(ex), None) == UNS                          # C % 64
    assert lib.gsw_softmax_rows(p, 4, 100, 104, 1.0, 1, None) == UNS                                             # cols % 8
    assert lib.gsw_softmax_rows(p, 4, 128, 64, 1.0, 1, None) == BAD                                              # ld < cols
    assert lib.gsw_softmax_rows(p, 0, 128, 128, 1.0, 1, None) == N.GSW_OK
    assert lib.gsw_add_layernorm(p, None, p, p, None, p, 4, 1544, 1e-5, 1, None) == UNS                          # C > 1536
    assert lib.gsw_add_layernorm(p, p, p, p, None, p, 4, 320, 1e-5, 1, None) == BAD                              # delta without x_new
    # image stages
    assert lib.gsw_resize_lanczos(None, 1, 8, 8, p, 4, 4, 0, None, None, None, 0, None, None, 0, None) == BAD
    assert lib.gsw_resize_lanczos(p, 1, 8, 8, p, 4, 4, 7, None, None, None, 0, None, None, 0, None) == BAD       # output mode
    assert lib.gsw_resize_lanczos(p, 1, 8, 8, p, 4, 4, 0, None, None, None, 0, None, None, 0, None) == BAD       # resampling without a plan
    assert lib.gsw_lanczos_plan(0, 4, None, None, 0) == -BAD
    assert lib.gsw_lanczos_plan(512, 256, None, None, 0) == 13                                                   # ceil(3 * 2) * 2 + 1
    assert lib.gsw_jpeg_roundtrip(p, 1, 8, 8, 10, p, 0, None, None) == BAD                                       # no workspace
    assert lib.gsw_jpeg_roundtrip(p, 1, 70000, 8, 10, p, 0, p, None) == UNS                                      # JPEG's size limit
    assert lib.gsw_jpeg_workspace_bytes(2, 16, 16) == 2 * (4 + 2) * 64
    assert lib.gsw_gaussian_blur(p, 1, 8, 8, -1.0, p, p, None) == BAD
    assert lib.gsw_gaussian_blur(p, 1, 8, 8, 2.0, p, None, None) == BAD
    assert lib.gsw_image_pointwise(p, 1, 8, 8, 9, 1.0, 0, 0, p, 0, None, None) == BAD                            # unknown op
    assert lib.gsw_image_pointwise(p, 1, 8, 8, N.GSW_PW_CONTRAST, 1.0, 0, 0, p, 0, None, None) == BAD            # contrast needs a workspace
    assert lib.gsw_tensor_to_image(p, 3, 1, 8, 8, 0, p, None) == BAD                                             # fp64
    assert lib.gsw_mt19937_uniform(None, 0, p, 4, None, None) == BAD
    key = (ctypes.c_uint32 * 624)()
    assert lib.gsw_mt19937_uniform(key, 625, p, 4, None, None) == BAD
    assert lib.gsw_mt19937_uniform(key, 624, p, 0, None, None) == N.GSW_OK


def test_extras_struct_layout_matches_the_header(tmp_path):
    """include/gswm.h is the boundary: the ctypes mirror of GswMmExtras must have the C compiler's size and field offsets (a silent mismatch would hand the
    engine a workspace pointer where it expects a capacity).  gcc compiles a probe against the header itself; the header must also be plain C."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    names = [f[0] for f in N.GswMmExtras._fields_]
    src = tmp_path / "probe.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "gswm.h"\nint main(void) {\n  printf("%zu\\n", sizeof(GswMmExtras));\n'
                   + "".join(f'  printf("{n} %zu\\n", offsetof(GswMmExtras, {n}));\n' for n in names) + "  return 0;\n}\n")
    exe = tmp_path / "probe"
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    assert int(out[0]) == ctypes.sizeof(N.GswMmExtras)
    for line in out[1:]:
        if line:
            n, off = line.split()
            assert getattr(N.GswMmExtras, n).offset == int(off), n
    # every field of the C struct is mirrored (count the declarators between the braces of the typedef)
    hdr = open(os.path.join(ROOT, "include", "gswm.h")).read()
    body = re.search(r"typedef struct GswMmExtras \{(.*?)\} GswMmExtras;", hdr, re.S).group(1)
    decl = [d.strip().split()[-1].lstrip("*") for stmt in body.split(";") if stmt.strip() for d in stmt.split(",")]
    assert decl == names


def test_ctypes_prototypes_match_the_header_declarations():
    """Every declaration of include/gswm.h against its ctypes prototype: number of parameters, and per parameter pointer / int / int64 / uint32 / uint64 / float / double
    (an int where the header says int64_t passes on x86-64 until a size crosses 2^31)."""
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "gswm.h")).read(), flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    decls = re.findall(r"([A-Za-z_][\w\s\*]*?)\b(gsw_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", txt)
    assert len(decls) == len(header_symbols())

    def kind_of_c(t):
        t = t.strip()
        if "*" in t or "[" in t:            # (an array parameter is a pointer)
            return "ptr"
        for key, kind in (("uint64_t", "u64"), ("size_t", "u64"), ("int64_t", "i64"), ("uint32_t", "u32"), ("unsigned", "u32"), ("double", "f64"), ("float", "f32"), ("int", "i32")):
            if re.search(rf"\b{key}\b", t):
                return kind
        raise AssertionError(f"unmapped C type {t!r}")

    def kind_of_ctypes(t):
        if t is None:
            return "void"
        if t in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(t, "contents") or issubclass(t, ctypes._Pointer):
            return "ptr"
        return {ctypes.c_int: "i32", ctypes.c_int64: "i64", ctypes.c_uint64: "u64", ctypes.c_uint32: "u32", ctypes.c_float: "f32", ctypes.c_double: "f64",
                ctypes.c_size_t: "u64", ctypes.c_long: "i64", ctypes.c_ulong: "u64", ctypes.c_longlong: "i64", ctypes.c_ulonglong: "u64", ctypes.c_uint: "u32"}[t]

    for ret, name, params in decls:
        res, args = N._PROTOTYPES[name]
        plist = [p for p in (q.strip() for q in params.split(",")) if p and p != "void"]
        # strip the parameter name (the last identifier) unless the declaration is type-only
        ctypes_kinds = [kind_of_ctypes(a) for a in args]
        c_kinds = [kind_of_c(p if "[" in p else (re.sub(r"\b[A-Za-z_]\w*$", "", p) if re.search(r"[\*\s][A-Za-z_]\w*$", p) else p)) for p in plist]
        assert c_kinds == ctypes_kinds, (name, c_kinds, ctypes_kinds)
        assert (kind_of_c(ret) if ret.strip() != "void" else "void") == kind_of_ctypes(res), (name, ret)
