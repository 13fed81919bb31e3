"""The product's gfx950 code object, disassembled (no GPU): properties of the COMPILED kernels that the design relies on.

  * wino4_bridge_kernel carries no packed-FP32 VALU instruction.  With v_pk_mul_f32 / v_pk_add_f32 in it, a bridge
    workgroup sharing a CU with a workgroup of the f16x3 GEMM stored wrong V' words now and then (tools/coresident_probe.py HZ8 / HZ9:
    8 of 8 frames differ with them, 0 of 8 without, everything else equal).  The kernel is compiled with
    __attribute__((target("no-packed-fp32-ops"))); this test fails the moment the attribute is dropped or stops working.
  * the check can see such instructions at all: the transform kernels beside it still have them."""
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


def _kernels(tmp_path):
    """{mangled kernel name: [instruction lines]} over the whole product library."""
    kernels = {}
    for co in _code_objects(LIB, str(tmp_path)):
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


@pytest.mark.skipif(not (os.path.exists(LIB) and shutil.which(os.path.join(LLVM, "llvm-objdump"))), reason="needs the built library and the ROCm LLVM tools")
def test_bridge_kernel_has_no_packed_fp32_instructions(tmp_path):
    kernels = _kernels(tmp_path)
    bridge = {k: v for k, v in kernels.items() if "wino4_bridge_kernel" in k}
    assert len(bridge) == 2, sorted(bridge)                                   # <PACK = false>, <PACK = true>
    for name, ins in bridge.items():
        assert len(ins) > 500, (name, len(ins))                               # (a real body, not a stub)
        hits = [l.strip() for l in ins if PK.search(l)]
        assert not hits, f"{name}: {len(hits)} packed-FP32 instructions, e.g. {hits[:3]} — see DESIGN 3.3 before allowing them back"
    # the detector sees them where they are allowed: the plain input / output transforms of the same file
    others = [k for k, v in kernels.items() if ("wino4_input_kernel" in k or "wino4_output_kernel" in k) and any(PK.search(l) for l in v)]
    assert others, "no packed-FP32 instruction found in any Winograd transform kernel: the check does not see them"
