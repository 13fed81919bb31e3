"""CPU (hipcc cross-compiles without a GPU): audit of the machine code of the f16x3 GEMM (sivo_amd/csrc/conv_wino4_h3.hip).

The kernel keeps two register sets of V' loads and the LDS-DMA of two stages in flight and counts vmcnt by hand; its loads are
issued in inline assembly, which hipcc schedules as opaque statements: the compiler does not know that a destination register
is written LATER than the statement.  What has to hold for that to be correct is checked here on the compiled code of every
instantiation: no spills, no scratch, every `s_waitcnt vmcnt` in the kernel is one of the hand-written ones, and no compiler
`v_mov` ever reads a register an asm `buffer_load_dword` writes (such a copy, placed between load and wait, would copy stale
bytes)."""
import os
import re
import subprocess
import tempfile

from conftest import ROOT

SRC = os.path.join(ROOT, "sivo_amd", "csrc", "conv_wino4_h3.hip")
HIPCC = "/opt/rocm/bin/hipcc"


def test_h3_gemm_machine_code_keeps_the_hand_counted_pipeline():
    if not os.path.exists(HIPCC):
        import pytest
        pytest.skip("no hipcc")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "h3.s")
        subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only",
                        "-I", os.path.dirname(SRC), SRC, "-o", out], check=True, capture_output=True)
        text = open(out).read()
    kernels = re.findall(r"^(_ZN4sivo20wino4_gemm_h3_kernel\w+):[^\n]*\n(.*?)s_endpgm", text, flags=re.S | re.M)
    assert len(kernels) == 3, [k for k, _ in kernels]
    for name, body in kernels:
        loads = set(re.findall(r"buffer_load_dword (v\d+),", body))
        assert len(loads) in (16, 32), (name, len(loads))          # one or two octets per lane, two register sets
        for m in re.finditer(r"v_mov_b32_e32 v\d+, (v\d+)\b", body):
            assert m.group(1) not in loads, (name, m.group(0))
        # every vmcnt wait sits inside an inline-asm statement (ours); the compiler added none of its own
        in_asm = False
        for line in body.splitlines():
            if "#ASMSTART" in line:
                in_asm = True
            elif "#ASMEND" in line:
                in_asm = False
            elif "s_waitcnt" in line and "vmcnt" in line:
                assert in_asm, (name, line.strip())
        assert body.count("v_mfma_f32_32x32x16_f16") >= 48
    for field in ("vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size"):
        vals = re.findall(r"\.%s:\s+(\d+)" % field, text)
        assert vals and all(v == "0" for v in vals), (field, vals)
    assert all(int(v) <= 256 for v in re.findall(r"\.vgpr_count:\s+(\d+)", text))


def test_direct_f16x3_machine_code_keeps_its_loads_in_flight():
    """conv3_h3.hip (direct 3x3, f16x3): its patch loads are asm statements whose destination registers are written one or two
    stages later than the statement.  On the compiled code of the four production instantiations (plain / through an Upsample,
    phased / interleaved stage loop): between an asm load and the next hand-written `s_waitcnt vmcnt(0)` no instruction reads
    the load's destination (a compiler copy or spill there would move stale bytes), no vector register is spilled, nothing
    lives in scratch, and a stage still issues its 108 MFMAs."""
    if not os.path.exists(HIPCC):
        import pytest
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "sivo_amd", "csrc", "conv3_h3.hip")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "d3.s")
        subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only",
                        "-I", os.path.dirname(src), src, "-o", out], check=True, capture_output=True)
        text = open(out).read()
    kernels = re.findall(r"^(_ZN4sivo15conv3_h3_kernel\w+):[^\n]*\n(.*?)s_endpgm", text, flags=re.S | re.M)
    assert len(kernels) == 4, [k for k, _ in kernels]
    for name, body in kernels:
        pending, in_asm, nloads = set(), False, 0
        for line in body.splitlines():
            l = line.strip()
            if "#ASMSTART" in l:
                in_asm = True
                continue
            if "#ASMEND" in l:
                in_asm = False
                continue
            if not l or l[0] in ";.":
                continue
            m = re.match(r"buffer_load_(?:dword|ubyte) (v\d+),", l)
            if m and in_asm:
                pending.add(m.group(1))
                nloads += 1
                continue
            if in_asm and "s_waitcnt" in l and "vmcnt(0)" in l:
                pending.clear()
                continue
            regs = set("v" + r for r in re.findall(r"\bv(\d+)\b", l))
            for a, b in re.findall(r"v\[(\d+):(\d+)\]", l):
                regs |= set("v%d" % i for i in range(int(a), int(b) + 1))
            assert not (regs & pending), (name, l)
        assert nloads >= 48, (name, nloads)                     # prologue + loop, 24 (48 through an Upsample) per stage
        assert body.count("v_mfma_f32_32x32x16_f16") >= 108, name
    names = re.findall(r"\.name:\s+(_ZN4sivo15conv3_h3_kernel\w+)", text)
    for field in ("vgpr_spill_count", "private_segment_fixed_size"):
        vals = re.findall(r"\.%s:\s+(\d+)" % field, text)
        assert vals and all(v == "0" for v in vals), (field, vals)
    assert len(names) == 4 and all(int(v) <= 256 for v in re.findall(r"\.vgpr_count:\s+(\d+)", text))
