"""The product's gfx950 code object, disassembled (no GPU): properties of the COMPILED kernels that the design relies on.

  * NO kernel of the product carries a packed-FP32 VALU instruction (docs/HW_NOTE_packed_fp32.md).  With v_pk_mul_f32 /
    v_pk_add_f32 in it, a wino4_bridge_kernel workgroup sharing a CU with a workgroup of the f16x3 GEMM stored wrong V' words now and then
    (tools/coresident_probe.py HZ8: 8 of 8 frames differ with them, 0 of 8 without, everything else equal).  What else the bridge had that
    the other packed kernels lacked was never found, so since round 6 the whole library is compiled with -target-feature
    -packed-fp32-ops (csrc/Makefile NOPK); the bridge keeps its function attribute as well.  The test fails the moment either is dropped
    or stops working, for any kernel.
  * the check can see such instructions at all: the reproducer build (conv_wino4.hip compiled with them) has them."""
import os
import re
import shutil
import struct
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(os.path.dirname(HERE), "sivo_amd", "libsivo_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
PK = re.compile(r"\bv_pk_(mul|add|fma)_f32\b")


def _code_objects(lib, tmp):
    """Every gfx950 ELF of the library's .hip_fatbin section (one clang offload bundle per translation unit)."""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(tmp, "copy.so")], check=True)
    d = open(fat, "rb").read()
    out, pos = [], 0
    while True:
        i = d.find(MAGIC, pos)
        if i < 0:
            break
        n, = struct.unpack_from("<Q", d, i + len(MAGIC))
        o = i + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", d, o)
            triple = d[o + 24:o + 24 + tl].decode()
            o += 24 + tl
            if "gfx950" in triple and size:
                path = os.path.join(tmp, f"co{len(out)}.elf")
                open(path, "wb").write(d[i + off:i + off + size])
                out.append(path)
        pos = i + len(MAGIC)
    return out


def _kernels(tmp_path, lib):
    """{mangled kernel name: [instruction lines]} over a whole library."""
    kernels = {}
    for co in _code_objects(lib, str(tmp_path)):
        txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co], check=True, capture_output=True, text=True).stdout
        name = None
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
            if m:
                name = m.group(1)
                kernels[name] = []
            elif name and line.strip():
                kernels[name].append(line)
    return kernels


needs_tools = pytest.mark.skipif(not (os.path.exists(LIB) and shutil.which(os.path.join(LLVM, "llvm-objdump"))), reason="needs the built library and the ROCm LLVM tools")
PKLIB = os.path.join(os.path.dirname(LIB), "libsivo_hip_diag_pkbridge.so")


@needs_tools
def test_no_kernel_of_the_product_has_packed_fp32_instructions(tmp_path):
    kernels = _kernels(tmp_path, LIB)
    bodies = {k: v for k, v in kernels.items() if len(v) > 8}
    assert len(bodies) >= 100, len(bodies)                                    # (the whole library was disassembled, not one translation unit)
    for must in ("wino4_bridge_kernel", "wino4_input_kernel", "wino4_output_kernel", "wino4_gemm_h3_kernel", "conv3_h3_kernel", "conv_cls_h3_kernel",
                 "conv7_h3_kernel", "fast_cells_kernel", "pyramid_kernel", "orient_describe_kernel", "entropy_gate_kernel", "maxpool2"):
        assert any(must in k for k in bodies), f"{must}: not found in the code object"
    bridge = {k: v for k, v in bodies.items() if "wino4_bridge_kernel" in k}
    assert len(bridge) == 2 and all(len(v) > 500 for v in bridge.values()), {k: len(v) for k, v in bridge.items()}      # <PACK = false>, <PACK = true>: real bodies
    hits = {k: [l.strip() for l in v if PK.search(l)] for k, v in bodies.items()}
    hits = {k: v for k, v in hits.items() if v}
    assert not hits, (f"{len(hits)} kernels carry packed-FP32 instructions, e.g. " + "; ".join(f"{k}: {len(v)} ({v[0]})" for k, v in list(hits.items())[:4])
                      + " - see DESIGN 3.3 before allowing them back")


@needs_tools
@pytest.mark.skipif(not os.path.exists(PKLIB), reason="needs the reproducer build (make -C sivo_amd/csrc diag_pkbridge)")
def test_the_detector_sees_packed_fp32_instructions_in_the_reproducer_build(tmp_path):
    kernels = _kernels(tmp_path, PKLIB)
    packed = [k for k, v in kernels.items() if "wino4_bridge_kernel" in k and any(PK.search(l) for l in v)]
    assert len(packed) == 2, packed                                           # both forms of the bridge, as they were until round 5
    others = [k for k, v in kernels.items() if "wino4" not in k and any(PK.search(l) for l in v)]
    assert not others, others[:4]                                             # ... and only the one translation unit
