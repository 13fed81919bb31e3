"""The co-residency fault of DESIGN 3.3, its two mitigations guarded deterministically, and the frame's kernels beside each other.

Round 3 found frames that were not reproducible when a wino4_bridge_kernel workgroup shared a CU with a workgroup of the f16x3 GEMM.
Round 5 narrowed it down (tools/coresident_probe.py HZ5 / HZ7 / HZ8 / HZ9): LDS, M and the bridge's window reads are clean; the wrong
V' words come out of the bridge's arithmetic when — and only when — the kernel is compiled with packed-FP32 VALU instructions
(v_pk_mul_f32 / v_pk_add_f32).  The product (a) compiles the bridge without them (tests/test_codeobj.py checks the code object on the
CPU) and (b) still keeps the constellation from arising: every kernel that issues LDS-DMA in inline assembly leaves no LDS on its CU
for a foreign workgroup.
  * test_bridge_as_shipped_is_reproducible_beside_the_exact_lds_gemm — (b) switched off in the diagnostic build (the GEMM asks for
    its exact LDS, so bridge workgroups DO share CUs with it): eight full-size frames on three lanes must equal the one-lane
    handle bit for bit.  With the packed form of the bridge (libsivo_hip_diag_pkbridge.so, the reproducer) 6 - 8 of 8 differ; that run
    is reported, not asserted.
  * test_two_kernel_reproducer_bridge_beside_gemm — the same pair without the network (tools/bridge_pair_repro.py): one bridged layer
    on two streams, every GEMM + bridge run twice and compared; 0 differing V' words as shipped, ~10^4 with the packed form (reported).
  * test_lds_dma_kernels_leave_no_lds_beside_them — the launchers note the dynamic LDS they ask for per CU; one frame of each
    reference net later every note must be the CU's whole 160 KB.  Fails the moment somebody removes the claim from a launcher.
  * test_frame_kernels_beside_each_other — the one pairing that is NOT excluded by construction: the f16x3 classifier runs two
    80 KB workgroups per CU, and at the head / tail of its launch a CU may hold one of them plus foreign workgroups (ORB, stereo
    matching, the entropy gate on its high-priority stream).  The whole frame is run with the ORB / gate work started at a sweep of
    delays so that it lands on every phase of the network, the classifier included: class / confidence / entropy maps, keys,
    descriptors, stereo matches and the gate's outputs must equal the solo results bit for bit, every time."""
import ctypes as C
import threading
import time

import numpy as np
import pytest
import torch

from sivo_amd import _lib, netspec, selection, weights as wts
from sivo_amd.frame import StereoFramePipeline
from sivo_amd.segnet import BayesianSegNet

pytestmark = pytest.mark.gpu
H, W = 352, 1024
WHOLE_LDS = 160 * 1024


def _net(kind, T):
    text = (netspec.standard_prototxt if kind == "standard" else netspec.basic_prototxt)(T, H, W)
    layers = netspec.parse_layers(text)
    return BayesianSegNet(prototxt=text, weights=wts.pack(layers, wts.synth_weights(layers, 42)), T=T)


def _maps():
    return (torch.empty((H, W), dtype=torch.uint8, device="cuda"), torch.empty((H, W), dtype=torch.float64, device="cuda"),
            torch.empty((H, W), dtype=torch.float64, device="cuda"))


def _probe(name):
    """One variant of tools/coresident_probe.py in its own process (the diagnostic switches are read once per process)."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import coresident_probe as cp
    env = dict(os.environ)
    env.update(dict((n, e) for n, e in cp.VARIANTS)[name])
    env["PROBE_SEEDS8"] = "1"
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "coresident_probe.py"), "--one", name], env=env, capture_output=True, text=True, timeout=600)
    m = re.search(r"frames that differ: (\d+) of (\d+)", out.stdout)
    assert out.returncode == 0 and m, out.stdout[-2000:] + out.stderr[-2000:]
    return int(m.group(1)), int(m.group(2)), out.stdout


def test_bridge_as_shipped_is_reproducible_beside_the_exact_lds_gemm():
    bad, n, _ = _probe("HZ8 exact LDS, the bridge as shipped (no packed-FP32 instructions)")
    assert (bad, n) == (0, 8)
    bad_pk, n_pk, _ = _probe("HZ8 exact LDS, the bridge with packed-FP32 instructions (the reproducer)")
    print(f"[coresident] GEMM with its exact LDS (bridge workgroups share CUs with it), 3 lanes against 1 lane, full size: the bridge as shipped {bad} of {n} "
          f"frames differ; the bridge compiled with packed-FP32 instructions {bad_pk} of {n_pk} frames differ")


def test_two_kernel_reproducer_bridge_beside_gemm():
    """tools/bridge_pair_repro.py: no network — two streams each run one bridged F(4x4) layer (f16x3 GEMM, then the bridge) over and over
    on random data, the GEMM with its exact LDS so that one lane's bridge workgroups share CUs with the other lane's GEMM; every GEMM +
    bridge is run twice and compared word for word.  The bridge as shipped: no V' word may differ in 1200 layer runs; the packed-FP32 form
    of the bridge (the reproducer build) beside a GEMM that claims the whole LDS: none either.  The packed form beside the exact-LDS GEMM —
    both mitigations off — is reported: ~10^4 words differ per second of run time."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import bridge_pair_repro as bp
    found = {}
    for name in ("bridge as shipped, GEMM with its exact LDS, 2 lanes", "packed bridge, GEMM with its exact LDS, 2 lanes", "packed bridge, GEMM claiming 160 KB, 2 lanes"):
        env = dict(os.environ)
        env.update(next(v for v in bp.VARIANTS if v[0] == name)[2])
        env["SIVO_W4_VERIFY"] = "1"
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "bridge_pair_repro.py"), "--one", name], env=env, capture_output=True, text=True, timeout=300)
        m = re.search(r"(\d+) layer runs compared .*M words that differ (\d+), V' words that differ (\d+)", out.stdout)
        assert out.returncode == 0 and m, out.stdout[-2000:] + out.stderr[-2000:]
        found[name] = tuple(int(g) for g in m.groups())
    shipped, packed = found["bridge as shipped, GEMM with its exact LDS, 2 lanes"], found["packed bridge, GEMM with its exact LDS, 2 lanes"]
    assert shipped[0] >= 1200 and shipped[1:] == (0, 0), shipped
    assert packed[1] == 0, packed                      # the GEMM itself is deterministic in either build
    claimed = found["packed bridge, GEMM claiming 160 KB, 2 lanes"]
    assert claimed[0] >= 1200 and claimed[1:] == (0, 0), claimed      # the other mitigation alone: even the packed bridge is safe when no GEMM workgroup fits beside it
    print(f"[coresident] two-kernel reproducer, {shipped[0]} layer runs each: V' words that differ between two runs of the same GEMM + bridge: "
          f"bridge as shipped {shipped[2]}, bridge with packed-FP32 instructions {packed[2]}, the packed bridge beside a GEMM that claims the CU's whole LDS {claimed[2]}")
    if packed[2] == 0:
        print("[coresident] NOTE: the packed form did not fail on this box in this run — the zeros above then say less than they should")


def test_lds_dma_kernels_leave_no_lds_beside_them():
    from bench import make_inputs
    d_bgr = torch.from_numpy(make_inputs(H, W)[0]).cuda()
    out = (C.c_uint32 * 4)()
    _lib.check(_lib.dbg().sivo_debug_lds_claims(out, 1))
    m = _maps()
    std = _net("standard", 4)
    std.segment_into(d_bgr, 1, m)
    torch.cuda.synchronize()
    _lib.check(_lib.dbg().sivo_debug_lds_claims(out, 1))
    gemm, conv3, cls, conv7 = list(out)
    assert (gemm, conv3, cls) == (WHOLE_LDS, WHOLE_LDS, WHOLE_LDS) and conv7 == 0, list(out)       # (no 7x7 layer in SegNet-Standard)
    del std
    basic = _net("basic", 2)
    basic.segment_into(d_bgr, 1, m)
    torch.cuda.synchronize()
    _lib.check(_lib.dbg().sivo_debug_lds_claims(out, 1))
    assert out[3] == WHOLE_LDS and all(v in (0, WHOLE_LDS) for v in out), list(out)


def test_frame_kernels_beside_each_other():
    from bench import make_inputs
    bgr, left, right = make_inputs(H, W)
    d_bgr, d_left, d_right = (torch.from_numpy(a).cuda() for a in (bgr, left, right))
    sn = _net("standard", 12)
    solo = _maps()
    sn.segment_into(d_bgr, 2000, solo)
    torch.cuda.synchronize()
    fp0 = StereoFramePipeline()
    ref = fp0.finish(fp0.start_orb(d_left, d_right), solo[0].cpu().numpy())       # ORB + matching with nothing beside them
    k, d = ref["keys"], ref["depth"]
    xyz = np.stack([(k["x"] - 498.692) * d / 718.856, (k["y"] - 173.215) * d / 718.856, d], 1).astype(np.float64)
    gate_args = (np.eye(6) * 1e-4, 718.856, 718.856, 386.1448 / 718.856, fp0.ex_l.GetScaleSigmaSquares(), 4.0)
    gate_ref = selection.entropy_gate_map_dev(k, d, xyz, solo[2], *gate_args)
    assert not sn.take_overflow()

    # the network alone: how long a frame is, so that the sweep below covers it end to end
    t0 = time.perf_counter()
    for _ in range(3):
        sn.segment_into(d_bgr, 2000, _maps())
    torch.cuda.synchronize()
    frame_s = (time.perf_counter() - t0) / 3
    delays = [f * frame_s for f in (0.0, 0.3, 0.6, 0.8, 0.88, 0.94, 1.0)]
    checked = 0
    for delay in delays:
        for _ in range(3):
            fp = StereoFramePipeline(start_delay_s=delay)
            m = _maps()
            stop = threading.Event()
            gate_bad = []

            def gate_loop():
                # the entropy gate (its own high-priority stream) over and over while the frame runs, on the solo frame's map
                while not stop.is_set():
                    g = selection.entropy_gate_map_dev(k, d, xyz, solo[2], *gate_args)
                    if not all(np.array_equal(a, b) for a, b in zip(g, gate_ref)):
                        gate_bad.append(1)
            th = threading.Thread(target=gate_loop)
            th.start()
            sn.segment_into(d_bgr, 2000, m)
            pending = fp.start_orb(d_left, d_right)
            torch.cuda.synchronize()
            got = fp.finish(pending, m[0].cpu().numpy())
            stop.set(); th.join()
            assert all(torch.equal(a, b) for a, b in zip(m, solo)), f"maps differ from the solo frame (ORB started {1e3 * delay:.1f} ms into the frame)"
            for key in ("keys", "desc", "right", "depth"):
                assert got[key].tobytes() == ref[key].tobytes(), (key, delay)
            assert not gate_bad
            checked += 1
    assert not sn.take_overflow()
    print(f"[coresident] {checked} frames with ORB / stereo matching / the entropy gate started 0 .. {1e3 * delays[-1]:.1f} ms into a {1e3 * frame_s:.2f} ms frame: "
          "maps, keys, descriptors, matches and gate outputs equal the solo results bit for bit")
