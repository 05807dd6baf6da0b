"""Host-side check of the built code objects (no GPU): every kernel of libgswm.so is a gfx950 kernel and NONE of them uses scratch memory -- a spilling instantiation
(hipcc at a launch-bounds register cap) costs far more than it looks and has slipped in before (ABI hygiene).  Reads the clang offload bundle in the
library's .hip_fatbin section and the AMDGPU metadata note (msgpack) of the code object."""
import os
import struct

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "a-watermark-for-diffusion-models_amd", "libgswm.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(blob: bytes):
    """(triple, bytes) of every entry of every (uncompressed) clang offload bundle in the file"""
    out, pos = [], 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return out
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            out.append((triple, blob[pos + off:pos + off + size]))
            p += 24 + tl
        pos += len(MAGIC)


def _kernel_metadata(elf: bytes):
    """the amdhsa.kernels list of a code object: walk the ELF64 section headers to the SHT_NOTE sections, find the AMDGPU metadata note (type 32)"""
    import msgpack
    assert elf[:4] == b"\x7fELF" and elf[4] == 2
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for i in range(shnum):
        sh = shoff + i * shentsize
        sh_type, = struct.unpack_from("<I", elf, sh + 4)
        off, size = struct.unpack_from("<QQ", elf, sh + 0x18)
        if sh_type != 7:          # SHT_NOTE
            continue
        p = off
        while p + 12 <= off + size:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            name = elf[p + 12:p + 12 + namesz].rstrip(b"\0")
            d0 = p + 12 + (namesz + 3) // 4 * 4
            if name == b"AMDGPU" and ntype == 32:
                return msgpack.unpackb(elf[d0:d0 + descsz], raw=False)["amdhsa.kernels"]
            p = d0 + (descsz + 3) // 4 * 4
    return None


@pytest.mark.skipif(not os.path.isfile(LIB), reason="libgswm.so is not built (python __graft_entry__.py)")
def test_every_kernel_is_gfx950_and_none_uses_scratch():
    blob = open(LIB, "rb").read()
    objs = [(t, b) for t, b in _code_objects(blob) if "amdgcn" in t and len(b) > 0]
    assert objs, "no device code object found in libgswm.so"
    assert all("gfx950" in t for t, _ in objs), [t for t, _ in objs]
    kernels = []
    for _, elf in objs:
        md = _kernel_metadata(elf)
        assert md is not None
        kernels += md
    assert len(kernels) >= 150          # the template instantiations of the engine, attention, codec, image and small-batch kernels
    # (scalar registers parked in lanes of a vector register -- .sgpr_spill_count -- cost no memory traffic and are not counted)
    bad = [(k[".name"], k.get(".private_segment_fixed_size"), k.get(".vgpr_spill_count")) for k in kernels
           if k.get(".private_segment_fixed_size", 0) or k.get(".vgpr_spill_count", 0)]
    assert not bad, bad
    assert all(k.get(".wavefront_size") == 64 for k in kernels)


@pytest.mark.skipif(not os.path.isfile(LIB), reason="libgswm.so is not built (python __graft_entry__.py)")
def test_production_library_has_no_measurement_switches():
    """The shipped libgswm.so must be compiled with none of the measurement switches of csrc/gswm_ablate.inc (MM_TRACE, MM_ABL_*): those builds carry
    cycle stamps or compile parts of the main loop out and compute wrong results by design.  gsw_build_flags() is a plain host symbol: no GPU needed."""
    import ctypes
    lib = ctypes.CDLL(LIB)
    lib.gsw_build_flags.restype = ctypes.c_int
    assert lib.gsw_build_flags() == 0


def test_no_measurement_switch_outside_the_ablation_header():
    """#ifdef-guarded experiment code stays out of the kernels: the switches are hook macros defined in csrc/gswm_ablate.inc and nowhere else."""
    import re
    csrc = os.path.join(ROOT, "a-watermark-for-diffusion-models_amd", "csrc")
    pat = re.compile(r"^\s*#\s*(if|ifdef|ifndef|elif)\b.*\b(MM_ABL_|ATTN_ABL_|MM_TRACE|MM_ODD_BARRIER|ATTN_MINWAVES)")
    bad = []
    for name in sorted(os.listdir(csrc)):
        if not name.endswith((".hip", ".h")):
            continue
        for i, line in enumerate(open(os.path.join(csrc, name), errors="replace"), 1):
            if pat.search(line):
                bad.append(f"{name}:{i}: {line.strip()}")
    assert not bad, bad
