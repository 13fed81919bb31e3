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
        for a, b in re.findall(r"buffer_load_dwordx2 v\[(\d+):(\d+)\],", body):          # FORM 2 (256 x 256 items): 8-byte loads, two tiles per lane
            loads |= {"v%d" % i for i in range(int(a), int(b) + 1)}
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
    """conv3_h3.hip (direct 3x3, f16x3): its register loads are asm statements whose destination registers are written one or two
    stages later than the statement.  On the compiled code of the nine production instantiations (fp32 input plain / through an
    Upsample in the phased and the interleaved stage loop, fp32 input with packed output, packed input by LDS-DMA and through
    an Upsample, each with fp32 and packed output): between an asm load and the next hand-written `s_waitcnt vmcnt(0)` no
    instruction reads the load's destination (a compiler copy or spill there would move stale bytes), no vector register is
    spilled, nothing lives in scratch, a stage still issues its 108 MFMAs, no scalar operand of an asm memory instruction
    comes out of a v_readlane / v_readfirstlane fewer than five wait states earlier (nothing pads that hazard inside an asm
    statement) — and the steady-state loop of the packed-input forms, whose point is the instruction budget, stays below
    700 instructions per stage outside the output stage, with at most a dozen SGPR spill reloads in the blocks every stage runs."""
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
    assert len(kernels) == 9, [k for k, _ in kernels]

    def regs_of(operand_text):
        regs = set("v" + r for r in re.findall(r"\bv(\d+)\b", operand_text))
        for a, b in re.findall(r"v\[(\d+):(\d+)\]", operand_text):
            regs |= set("v%d" % i for i in range(int(a), int(b) + 1))
        return regs

    for name, body in kernels:
        in_form = int(re.search(r"kernelILi(\d)E", name).group(1))          # 0 fp32, 1 fp32 through an Upsample, 2 packed (DMA), 3 packed through an Upsample
        pending, in_asm, nloads = set(), False, 0
        sgpr_written_by_valu = {}                                            # sgpr -> wait states since a v_readlane / v_readfirstlane wrote it
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
            nop = re.match(r"s_nop (\d+)", l)
            for k in list(sgpr_written_by_valu):
                sgpr_written_by_valu[k] += (int(nop.group(1)) + 1) if nop else 1
            m = re.match(r"v_read(?:first)?lane_b32 (s\d+),", l)
            if m:
                sgpr_written_by_valu[m.group(1)] = 0
            if in_asm and re.match(r"(buffer|global)_", l):
                sregs = set("s" + r for r in re.findall(r"\bs(\d+)\b", l))
                for a, b in re.findall(r"s\[(\d+):(\d+)\]", l):
                    sregs |= set("s%d" % i for i in range(int(a), int(b) + 1))
                for r in sregs:
                    assert sgpr_written_by_valu.get(r, 99) > 5, (name, r, l)
            m = re.match(r"(?:buffer|global)_load_(?:dwordx4|dword|ubyte) (v\d+|v\[\d+:\d+\]),", l)
            if m and in_asm:
                pending |= regs_of(m.group(1))
                nloads += 1
                continue
            if in_asm and "s_waitcnt" in l and "vmcnt(0)" in l:
                pending.clear()
                continue
            assert not (regs_of(l) & pending), (name, l)
        # prologue + loop: 24 per stage (48 through an Upsample) for the fp32 forms, none for the DMA form, 3 for the packed Upsample form
        assert nloads >= {0: 48, 1: 96, 2: 0, 3: 6}[in_form], (name, nloads)
        assert body.count("v_mfma_f32_32x32x16_f16") >= 108, name
        if in_form >= 2:
            # the steady-state loop = the backward branch around >= 100 MFMAs; the two copies of the output stage (ReLU / plain)
            # inside it run once per item, not per stage: count what is outside them (an output stage is the only place with stores)
            lines = body.splitlines()
            pos = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
            loops = []
            for i, l in enumerate(lines):
                m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
                if m and m.group(1) in pos and pos[m.group(1)] < i:
                    seg = lines[pos[m.group(1)]:i + 1]
                    if sum("v_mfma" in x for x in seg) >= 100:
                        loops.append(seg)
            assert loops, name
            seg = min(loops, key=len)
            blocks, cur = [], []
            for l in seg:
                if re.match(r"^\.LBB\d+_\d+:", l):
                    blocks.append(cur); cur = []
                else:
                    x = l.strip()
                    if x and x[0] not in ";.":
                        cur.append(x)
            blocks.append(cur)
            stage = [x for b in blocks if not any(re.match(r"buffer_store", y) for y in b) for x in b]
            hot = [x for b in blocks if any("v_mfma" in y for y in b) for x in b]             # the blocks every stage runs through
            n_ins, n_rl, n_hot, n_hot_rl = len(stage), sum(x.startswith("v_readlane") for x in stage), len(hot), sum(x.startswith("v_readlane") for x in hot)
            print(f"{name}: steady loop outside the output stage: {n_ins} instructions ({n_rl} v_readlane, item changes included); "
                  f"the blocks that hold its {sum('v_mfma' in x for x in hot)} MFMAs: {n_hot} instructions, {n_hot_rl} v_readlane")
            # (round 3's fp32-input form: about 1250 instructions and 73 v_readlane per stage)
            assert n_ins <= 700 and n_hot <= 580 and n_hot_rl <= 12, (name, n_ins, n_hot, n_hot_rl)
    names = re.findall(r"\.name:\s+(_ZN4sivo15conv3_h3_kernel\w+)", text)
    for field in ("vgpr_spill_count", "private_segment_fixed_size"):
        vals = re.findall(r"\.%s:\s+(\d+)" % field, text)
        assert vals and all(v == "0" for v in vals), (field, vals)
    assert len(names) == 9 and all(int(v) <= 256 for v in re.findall(r"\.vgpr_count:\s+(\d+)", text))
