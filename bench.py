#!/usr/bin/env python3
"""bench.py — frames/sec of SIVO's per-frame perception path on MI355X.

A "step" is one stereo frame through the hot path (BASELINE.json configs[2]):
  SegNet-Standard forward for T=12 Monte-Carlo dropout samples at 352x1024 (fp32 MFMA)
  -> per-pixel softmax sum -> [RCCL all-reduce when N > 1] -> mean / class / confidence / entropy maps
  ORB 2000 x 8 levels on the left and right images, semantic key filter (class <= TERRAIN),
  stereo matching (Hamming + SAD)                                   [rank 0, overlapped on its own streams]
All inputs are resident in HBM before the timed region starts.

Multi-GPU (one process per GPU, torch.distributed over RCCL): the T samples of a frame are
sharded over the ranks (global sample index keys the dropout masks, so the result does not
depend on N), each rank recomputes the sample-invariant prefix, one all-reduce(SUM) of the
15x352x1024 fp32 probability sums per frame.  T is fixed -> "scaling": "strong".

Prints ONE JSON line (rank 0).  `roofline`: the dominant kernel (the MFMA kernel with the
largest share of GPU time): executed matrix-core FLOP/s from HIP events on its launch stream
during the timed region against the dense peak of the instruction type it issues (frac <= 1),
with the algorithmic (direct-convolution) rate beside it and the whole-frame matrix-core
utilisation.  At N = 1 the line also carries `configs` (BASELINE configs[1], [3] on one GPU and
[4], each with its own roofline and the tests that cover it) and `host_boundary_fps` (the
reference's segmentImage boundary, PCIe included).  `cpu_baseline`: this repo's CPU oracle
(a restatement of the reference path — not Caffe/cuDNN/OpenCV, which are unavailable)
timed on the host cores for a bounded sample and extrapolated to the same frame.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
HBM_PEAK_GBS = 8000.0


def synthetic_frame(seed, rows=352, cols=1024):
    rng = np.random.default_rng(seed)
    img = np.full((rows, cols), 90.0)
    for _ in range(40):
        x0, x1 = sorted(rng.integers(0, cols, 2)); y0, y1 = sorted(rng.integers(0, rows, 2))
        img[y0:y1 + 1, x0:x1 + 1] = rng.integers(0, 256)
    img += rng.normal(0, 4, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def make_inputs(H, W):
    """Synthetic KITTI-shaped stereo pair: left gray + 8 px disparity right, colour left for the net."""
    wide = synthetic_frame(1234, H, W + 64)
    left = np.ascontiguousarray(wide[:, :W])
    rng = np.random.default_rng(4321)
    right = np.clip(np.rint(wide[:, 8:8 + W].astype(np.float64) + rng.normal(0, 1.0, (H, W))), 0, 255).astype(np.uint8)
    bgr = np.stack([left, np.roll(left, 1, 1), np.roll(left, 1, 0)], axis=2)
    return np.ascontiguousarray(bgr), left, right


def host_cpus():
    """(CPUs this process may use, OpenMP threads for the CPU baseline).  os.cpu_count() is the machine's (256 hardware threads on the GPU
    box); the container's cgroup grants a CPU-time quota (cpu.max: 16 CPUs there) — with 256 runnable threads on 16 CPUs' worth of time
    the OpenMP loops ran at 57 GFLOP/s where 32 threads reach 1671 (tools/cpu_threads_probe.py, profiles/r06_cpu_threads.log)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n, min(len(os.sched_getaffinity(0)), 2 * n if n < len(os.sched_getaffinity(0)) else n)


def cpu_baseline(kind, T, H, W, text, w, bgr, left, right, budget_note):
    """Two CPU figures on the host cores, about 30 s of CPU work at the full geometry.  value_dedup: the oracle — prefix once + up to FOUR
    of the T MC samples of the suffix (their mean, x T) + ORB pair + MC reduction of 2 samples (x T / 2).  value: the reference-equivalent
    figure — T full forwards with im2col + SGEMM convolutions (below)."""
    from oracle import oracle as O, prototxt as oproto
    cores, threads = host_cpus()
    O.lib().omp_set_num_threads(threads)          # (libgomp, through liboracle.so)
    net = oproto.parse(text)
    net["shape"][0] = 1
    first_drop = next(i for i, L in enumerate(net["layers"]) if L["type"] == "Dropout")
    prefix = dict(net, layers=net["layers"][:first_drop])
    blob = O.preprocess(bgr, 1, H, W)
    t0 = time.perf_counter()
    pb = O.run_net(prefix, w, blob, 7, keep=[L["top"][j] for L in prefix["layers"] for j in range(len(L["top"]))])
    t_prefix = time.perf_counter() - t0
    # suffix on K of the T samples, re-using the prefix blobs
    K = max(1, min(T, 4))
    t0 = time.perf_counter()
    last = None
    for smp in range(K):
        blobs = {k: v for k, v in pb.items() if k != "__last__"}
        site = 0
        for L in net["layers"][first_drop:]:
            t = L["type"]; bot = [blobs[b] for b in L["bottom"]]
            if t == "Convolution": out = O.conv2d(bot[0], *w[L["name"]], L["pad"])
            elif t == "BN": out = O.bn_inference(bot[0], *w[L["name"]])
            elif t == "ReLU": out = O.relu(bot[0])
            elif t == "Pooling":
                out, m = O.maxpool(bot[0]); blobs[L["top"][1]] = m
            elif t == "Upsample": out = O.unpool(bot[0], bot[1], bot[0].shape[2] * 2, bot[0].shape[3] * 2)
            elif t == "Dropout":
                out = O.dropout(bot[0], site, smp, 7); site += 1
            elif t == "LRN": out = O.lrn(bot[0], L["local_size"], L["alpha"], L["beta"])
            elif t == "Softmax": out = O.softmax(bot[0])
            blobs[L["top"][0]] = out; last = out
    t_suffix = (time.perf_counter() - t0) / K
    t0 = time.perf_counter()
    prob2 = np.concatenate([last, last])
    O.mc_finalize(O.mc_mean(prob2))
    t_mc = (time.perf_counter() - t0) * T / 2
    t0 = time.perf_counter()
    ex_l, ex_r = O.OrbExtractor(), O.OrbExtractor()
    res = {}
    th = [threading.Thread(target=lambda k=k, e=e, im=im: res.__setitem__(k, e(im))) for k, e, im in (("l", ex_l, left), ("r", ex_r, right))]
    [t.start() for t in th]; [t.join() for t in th]        # Frame.cc:126-129: two extractor threads
    kl, dl = res["l"]; kr, dr = res["r"]
    O.stereo_matches(kl, dl, kr, dr, ex_l.scale, ex_l.inv_scale, [ex_l.level(l) for l in range(8)],
                     [ex_r.level(l) for l in range(8)], 386.1448, 386.1448 / 718.856)
    t_orb = time.perf_counter() - t0
    frame_dedup = t_prefix + T * t_suffix + t_mc + t_orb
    # The reference-equivalent figure: what the reference's CPU path does per frame — T FULL forwards of the net, one per batch slot, no
    # de-duplication of the sample-invariant prefix (bayesian_segnet.cpp:174-177 copies the image into all T slots, :310 runs them), every
    # convolution as im2col + SGEMM per image the way Caffe's CPU ConvolutionLayer computes it (oracle/caffe_cpu.c: a blocked, FMA,
    # OpenMP SGEMM standing in for the BLAS Caffe links), the other layers by the oracle's OpenMP loops.  Whole forwards are measured until
    # ~20 s are spent (all T when the host is fast enough), the rest extrapolated from their mean.
    one = dict(net)
    O.caffe_conv2d(np.zeros((1, 64, H, W), np.float32), np.zeros((64, 64, 3, 3), np.float32), None, 1)     # (Caffe allocates its col buffer at set-up)
    ts = []
    t_begin = time.perf_counter()
    while len(ts) < T and (not ts or time.perf_counter() - t_begin + ts[-1] < 20.0):
        t0 = time.perf_counter()
        O.run_net(one, w, blob, 7, sample0=len(ts), keep=[], conv=O.caffe_conv2d)
        ts.append(time.perf_counter() - t0)
    t_fwd = float(np.mean(ts))
    frame_ref = T * t_fwd + t_mc + t_orb
    return {"value": 1.0 / frame_ref, "value_dedup": 1.0 / frame_dedup, "unit": "frames/s", "cores": cores, "threads": threads, "machine_hardware_threads": os.cpu_count(), "kind": "port",
            "samples_measured": len(ts), "samples_extrapolated": T - len(ts), "samples_measured_dedup": K, "samples_extrapolated_dedup": T - K,
            "sample": (f"value = reference-equivalent: {T} full forwards (no prefix de-duplication, as the reference runs its T batch slots), convolutions as im2col + blocked FMA SGEMM "
                       f"per image (what Caffe-CPU does; C, {threads} OpenMP threads on the {cores} CPUs the container's cgroup grants this process): {len(ts)} of {T} forwards measured, {t_fwd:.2f}s each (x{T}) + MC reduction {t_mc:.2f}s + ORB stereo pair "
                       f"and matching {t_orb:.2f}s on {kind} {H}x{W}.  value_dedup = the CPU oracle itself (plain C loop nests, the same {threads} threads, fp32 chain order; a restatement "
                       f"of the reference path, not Caffe/OpenCV) WITH the prefix computed once: prefix {t_prefix:.2f}s + {K} of {T} MC samples of the suffix measured, {t_suffix:.2f}s each (x{T}) "
                       "+ the same MC reduction and ORB times")}


# Matrix-core work of a kernel family relative to the ALGORITHMIC (direct-convolution) FLOPs it is credited with, and the
# dense peak of the instruction type it issues (/opt/skills/guides/MI355X_MICROARCH.md):
#   Winograd F(4x4,3x3) multiplies 36 instead of 144 per 4x4 outputs (1/4), F(2x2,3x3) 16 instead of 36 (1/2.25);
#   the bf16x6 GEMM issues six bf16 MFMA products per fp32 product (6/4 of the algorithmic count, on the bf16 pipe), the
#   f16x3 GEMM three fp16 products (3/4, on the fp16 pipe, same dense peak).
BF16_MFMA_PEAK_TFLOPS = 2500.0
KERNEL_CLASS = [("conv7_h3", 3.0, BF16_MFMA_PEAK_TFLOPS, "fp16 MFMA (direct 7x7; fp32 operands as fp16 hi + lo planes, 3 products per fp32 product, fp32 accumulate)"),
                ("conv_cls_h3", 3.0, BF16_MFMA_PEAK_TFLOPS, "fp16 MFMA (direct 3x3 classifier + Softmax + MC statistics; fp16 hi + lo planes, 3 products, fp32 accumulate)"),
                ("conv7_x6", 6.0, BF16_MFMA_PEAK_TFLOPS, "bf16 MFMA (direct 7x7; fp32 operands split into 3 bf16 planes, 6 products, fp32 accumulate)"),
                ("wino4_gemm_h3", 3.0 / 4.0, BF16_MFMA_PEAK_TFLOPS, "fp16 MFMA (fp32 operands as fp16 hi + lo planes, 3 products, fp32 accumulate)"),
                ("conv3_h3", 3.0, BF16_MFMA_PEAK_TFLOPS, "fp16 MFMA (direct 3x3; fp32 operands as fp16 hi + lo planes, 3 products per fp32 product, fp32 accumulate)"),
                ("wino4_gemm_x6", 6.0 / 4.0, BF16_MFMA_PEAK_TFLOPS, "bf16 MFMA (fp32 operands split into 3 bf16 planes, 6 products, fp32 accumulate)"),
                ("wino4_gemm", 1.0 / 4.0, FP32_MFMA_PEAK_TFLOPS, "fp32 MFMA"),
                ("conv_wino4f", 1.0 / 4.0, FP32_MFMA_PEAK_TFLOPS, "fp32 MFMA"),
                ("conv_wino", 1.0 / 2.25, FP32_MFMA_PEAK_TFLOPS, "fp32 MFMA"),
                ("conv_mfma", 1.0, FP32_MFMA_PEAK_TFLOPS, "fp32 MFMA")]


def kernel_class(name):
    for prefix, ratio, peak, what in KERNEL_CLASS:
        if name.startswith(prefix):
            return ratio, peak, what
    return None


def aggregate(rows):
    agg = {}
    for p in rows:
        if not p["launches"]:
            continue
        k = agg.setdefault(p["kernel"], {"ms": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0})
        k["ms"] += p["ms_total"]; k["launches"] += p["kernel_launches"]
        k["flops"] += p["flops_per_sample"] * p["samples"] * p["launches"]
        k["bytes"] += p["bytes_per_sample"] * p["samples"] * p["launches"]
    return agg


def traffic_from_profiles(dom_name, tag="main"):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes of the same configuration
    (profiles/*pmc_traffic_<tag>.json, tools/profile_round.sh; bench cannot collect PMC itself): the dispatch-weighted mean
    over the kernel's template instantiations — the same set of launches `achieved` averages over."""
    files = sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if "pmc_traffic" in f and f.endswith(".json")), reverse=True)
    files = [f for f in files if f"pmc_traffic_{tag}" in f] + [f for f in files if f"pmc_traffic_{tag}" not in f and tag == "main" and f.endswith("pmc_traffic.json")]
    for cand in files:
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", cand)))
            hits = [v for k, v in tj.items() if k.split("<")[0] == dom_name.split("<")[0] and isinstance(v, dict) and v.get("bytes") and v.get("dispatches")]
            if hits:
                n = sum(v["dispatches"] for v in hits)
                return int(sum(v["bytes"] * v["dispatches"] for v in hits) / n), "profiles/" + cand
        except Exception:
            pass
    return None, None


def mfma_roofline(prof_timed, prof_detail, n_timed_frames, n_detail, ms_per_frame, note, tag="main"):
    """roofline object of a SegNet run.  achieved / frac = ALGORITHMIC (direct-convolution) FLOP/s of the dominant
    kernel against the dense peak of the instruction type it issues; executed_tflops / executed_frac = the matrix-core
    products it really issues (Winograd multiplies 1/4 of the direct products, the split arithmetic 3 or 6 per fp32 product)."""
    by_kernel = aggregate(prof_detail)
    prof_timed, prof_lanes = prof_timed if isinstance(prof_timed, tuple) else (prof_timed, [])
    timed = aggregate(prof_timed) if prof_timed else {}
    lanes = aggregate(prof_lanes) if prof_lanes else {}
    mfma = {k: v for k, v in timed.items() if v["flops"] > 0 and v["ms"] > 0 and kernel_class(k)}
    frames = n_timed_frames
    if not mfma:
        mfma = {k: v for k, v in by_kernel.items() if v["flops"] > 0 and v["ms"] > 0 and kernel_class(k)}
        frames = n_detail
    dom_name, dom = max(mfma.items(), key=lambda kv: kv[1]["ms"])
    ratio, peak, what = kernel_class(dom_name)
    alg = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    executed = alg * ratio
    # whole frame: time the matrix cores would need at peak for the executed work of every MFMA kernel / frame time
    t_peak_ms = 0.0
    for k, v in by_kernel.items():
        c = kernel_class(k)
        if c and v["flops"] > 0:
            t_peak_ms += v["flops"] / n_detail * c[0] / (c[1] * 1e12) * 1e3
    traffic, tsrc = traffic_from_profiles(dom_name, tag)
    # The dominant kernel is ONE of the (up to) four kernels of an F(4x4) layer and is credited with the layer's whole convolution:
    # chain = the same FLOPs over the GEMM AND its transform kernels (input / bridge / output), conv_stack = every
    # convolution FLOP of the frame over every convolution kernel — both against the same fp16 dense peak, from the untimed
    # single-lane frames in which every kernel carries events (the transforms are not bracketed in the timed frames)
    chain_ms = sum(v["ms"] for k, v in by_kernel.items() if k.startswith("wino4_"))
    chain_fl = sum(v["flops"] for k, v in by_kernel.items() if k.startswith("wino4_gemm"))
    stack_ms = sum(v["ms"] for k, v in by_kernel.items() if kernel_class(k) or k.startswith("wino4_"))
    stack_fl = sum(v["flops"] for k, v in by_kernel.items() if kernel_class(k))
    chain = chain_fl / (chain_ms * 1e-3) / 1e12 if chain_ms > 0 else None
    stack = stack_fl / (stack_ms * 1e-3) / 1e12 if stack_ms > 0 else None
    # the same kernel in the frames that kept their lanes (the timed configuration): its launches carry half the samples each and
    # share the chip with the other lane's kernels; time = sum over the lanes
    two = {}
    tl = lanes.get(dom_name)
    if tl and tl["ms"] > 0 and tl["flops"] > 0:
        alg2 = tl["flops"] / (tl["ms"] * 1e-3) / 1e12
        n2 = tl["flops"] / (dom["flops"] / max(frames, 1))           # frames' worth of this kernel's work behind the two-lane rows
        two = {"frac_two_lane": round(alg2 / peak, 4), "executed_frac_two_lane": round(alg2 * ratio / peak, 4), "achieved_two_lane": round(alg2, 2),
               "avg_launch_ms_two_lane": tl["ms"] / max(tl["launches"], 1), "launches_per_frame_two_lane": round(tl["launches"] / max(n2, 1e-9), 2),
               "ms_per_frame_two_lane": round(tl["ms"] / max(n2, 1e-9), 3), "ms_per_frame_one_lane": round(dom["ms"] / max(frames, 1), 3),
               "two_lane_note": "the dominant kernel in the timed frames that keep their two sample groups on two streams (every "
                                "frame but the profiled one-lane ones runs like this): HIP events per lane, times summed over the lanes"}
    return {"bound": "mfma", "kernel": "sivo::" + dom_name, "instruction": what, **two,
            "chain_tflops": round(chain, 2) if chain else None, "chain_frac": round(chain / BF16_MFMA_PEAK_TFLOPS, 4) if chain else None,
            "chain_ms_per_frame": round(chain_ms / n_detail, 3),
            "conv_stack_tflops": round(stack, 2) if stack else None, "conv_stack_frac": round(stack / BF16_MFMA_PEAK_TFLOPS, 4) if stack else None,
            "conv_stack_ms_per_frame": round(stack_ms / n_detail, 3),
            "measured_on": {"timed_frames (inside the driver-timed region)": ["achieved", "frac", "executed_tflops", "executed_frac", "avg_launch_ms", "launches_per_frame", "flops_per_launch"]
                            if timed and frames == n_timed_frames and mfma is not None and prof_timed else [],
                            "untimed_single_lane_frames (after the timed region, every kernel bracketed by events)":
                                ["chain_*", "conv_stack_*", "mfma_kernels", "kernels_ms_per_frame", "segnet_kernel_ms_per_frame", "whole_frame_mfma_util (numerator)"],
                            "committed rocprofv3 --pmc passes of the same command": ["traffic"]},
            "achieved": round(alg, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(alg / peak, 4),
            "executed_tflops": round(executed, 2), "executed_frac": round(executed / peak, 4), "executed_per_algorithmic": round(ratio, 4),
            "traffic": traffic, "traffic_source": f"{tsrc} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, mean bytes per launch)" if traffic else None,
            "launches_per_frame": dom["launches"] / max(frames, 1), "flops_per_launch": dom["flops"] / max(dom["launches"], 1),
            "avg_launch_ms": dom["ms"] / max(dom["launches"], 1),
            "whole_frame_mfma_util": round(t_peak_ms / ms_per_frame, 4),
            # every matrix-core kernel of the frame (the dominant one above is the largest of these by time)
            "mfma_kernels": [{"kernel": "sivo::" + k, "ms_per_frame": round(v["ms"] / n_detail, 3), "launches_per_frame": round(v["launches"] / n_detail, 2),
                              "achieved": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2), "peak": kernel_class(k)[1],
                              "frac": round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / kernel_class(k)[1], 4),
                              "executed_frac": round(v["flops"] / (v["ms"] * 1e-3) / 1e12 * kernel_class(k)[0] / kernel_class(k)[1], 4),
                              "flops_per_launch": v["flops"] / max(v["launches"], 1), "traffic": traffic_from_profiles(k, tag)[0]}
                             for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1]["ms"]) if kernel_class(k) and v["flops"] > 0 and v["ms"] > 0],
            "kernels_ms_per_frame": {k: round(v["ms"] / n_detail, 3) for k, v in sorted(by_kernel.items())},
            "segnet_kernel_ms_per_frame": round(sum(v["ms"] for v in by_kernel.values()) / n_detail, 3),
            "note": note}


# Algorithmic HBM bytes of the integer / memory-bound kernels: pyramid level areas for (2000, 1.2, 8) at 1024 x 352
_LV = [(1024, 352), (853, 293), (711, 244), (593, 204), (494, 170), (412, 141), (343, 118), (286, 98)]


def membound_block(prof, fp, d_left, H, W, T, classes):
    """HBM-roofline figures of the memory-bound kernels, measured in THIS process right after the timed region (nothing else on
    the GPU): achieved = algorithmic bytes / mean HIP-event time, against 8 TB/s.  SegNet's streaming kernels come from the
    per-kernel events of the profiled frames (prof); the ORB kernel groups from the extractor's own events (sivo_orb_profile);
    Hamming and the MC reduction from torch events around single launches."""
    import torch
    from sivo_amd import matcher
    from sivo_amd.segnet import mc_segment
    rows = []

    def row(kernel, us, nbytes, note=""):
        gbs = nbytes / us / 1e3 if us > 0 else 0.0
        rows.append({"kernel": kernel, "avg_us": round(us, 2), "algorithmic_bytes": int(nbytes), "achieved_GBps": round(gbs, 1),
                     "frac_of_8TBps": round(gbs / HBM_PEAK_GBS, 4), **({"note": note} if note else {})})
    for k, v in sorted(aggregate(prof).items()):
        if kernel_class(k) is None and v["bytes"] > 0 and v["ms"] > 0 and v["launches"]:
            row("sivo::" + k, 1e3 * v["ms"] / v["launches"], v["bytes"] / v["launches"], "SegNet, per launch (all samples of a layer)")
    if fp is not None:
        area = [w * h for w, h in _LV]; pyr = sum(area); padded = sum((w + 38) * (h + 38) for w, h in _LV)
        fp.ex_l.profile(True)
        for _ in range(20):
            kl, _ = fp.ex_l(d_left)
        ms, calls, keys = fp.ex_l.profile_read()
        fp.ex_l.profile(False)
        n = max(keys, 1.0)
        small = "working set 0.3-2 MB: launch / latency bound by construction (0.04-0.3 us at 8 TB/s)"
        row("sivo::pyramid_kernel (all 8 levels, one launch)", 1e3 * ms["pyramid"], 2 * area[0] + sum(area[:-1]) + sum(area[1:]), small)
        row("sivo::blur_border_kernel (all levels)", 1e3 * ms["blur"], 2 * pyr + (padded - pyr), small)
        row("sivo::fast_cells_kernel (FAST-9/16 + scan + ordered emission, all levels)", 1e3 * ms["fast"], pyr + 2 * 4 * 20000, small)
        row("sivo::orient_describe_kernel (IC_Angle + rBRIEF)", 1e3 * ms["angle"], n * (749 + 512 + 36), small)
    rng = np.random.default_rng(0)
    A = torch.from_numpy(rng.integers(0, 256, (2000, 32), dtype=np.uint8)).cuda()
    B = torch.from_numpy(rng.integers(0, 256, (2000, 32), dtype=np.uint8)).cuda()
    lg = torch.randn(T, classes, H, W, device="cuda")

    def timed(fn, reps=20):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / reps
    row("sivo::hamming_matrix_kernel 2000 x 2000 (DescriptorDistance, dense)", timed(lambda: matcher.descriptor_distance_matrix(A, B)), 32 * 4000 + 4 * 2000 * 2000)
    row("sivo::hamming_argmin2_kernel 2000 queries x 2000 (brute force)", timed(lambda: matcher.bruteforce(A, B)), 32 * 4000 + 3 * 4 * 2000,
        "every B row per query comes from L2")
    row(f"sivo::mc_reduce_finalize_kernel T = {T} (softmax, f64 mean, argmax / max / entropy)", timed(lambda: mc_segment(lg)),
        T * classes * H * W * 4 + H * W * 17)
    return rows


def time_segnet(sn, frame, steps, warmup, barrier, profile_every=8, events=True, flush=lambda: None):
    """Warm up, time `steps` calls of frame(seed) between barriers, return (elapsed s, MFMA-kernel rows of the timed region — a pair:
    the frames profiled in one lane, the frames profiled with their lanes kept —, all-kernel rows of min(steps, 10) further untimed
    single-lane frames).  Inside the timed region frame i carries events when i % profile_every is 0 (issued in ONE lane: a launch has
    the GPU to itself) or profile_every / 2 (lanes kept: the kernel's time while it shares the chip with the other sample group)."""
    for i in range(warmup):
        frame(1000 + i)
    flush()
    barrier()
    one_lane, two_lane = [], []
    half = profile_every // 2
    t0 = time.perf_counter()
    for i in range(steps):
        ph = i % profile_every
        # (switching the events off does not wait for them; they are read one frame later, when the profiled frame has long completed and the
        # next one is already enqueued: reading them right away drained the two-frame pipeline five times in twenty frames, 2.5 % of the line)
        if events and ph == 0:
            sn.profile(True, mfma_only=True, reset=True)
        elif events and ph == 1:
            sn.profile(False)
        elif events and ph == 2:
            one_lane += sn.profile_read()
        elif events and ph == half:
            sn.profile(True, mfma_only=True, reset=True, keep_lanes=True)
        elif events and ph == half + 1:
            sn.profile(False)
        elif events and ph == half + 2:
            two_lane += sn.profile_read()
        frame(2000 + i)
    flush()                      # (frames still in flight are completed inside the timed region)
    barrier()
    elapsed = time.perf_counter() - t0
    if events:
        sn.profile(False)
        last = (steps - 1) % profile_every           # a profiled frame whose rows were not read inside the loop
        if last in (0, 1):
            one_lane += sn.profile_read()
        elif last in (half, half + 1):
            two_lane += sn.profile_read()
    prof_timed = (one_lane, two_lane)
    n_detail = min(steps, 10)
    sn.profile(True, reset=True)
    for i in range(n_detail):
        frame(3000 + i)
    flush()
    barrier()
    prof = sn.profile_read()
    sn.profile(False)
    return elapsed, prof_timed, prof, n_detail


def ba_scene(seed=99, n_kf=20, n_pts=3000):
    """20 keyframes along a forward trajectory, 3000 points in a frustum box, KITTI-00 intrinsics,
    an edge for every (keyframe, point) that projects into the image (80 % stereo)."""
    from sivo_amd.optimizer import EDGE_DTYPE
    rng = np.random.default_rng(seed)
    fx = fy = 718.856; cx, cy, bf = 498.692, 173.215, 386.1448
    poses = np.zeros((n_kf, 12))
    for k in range(n_kf):
        yaw = np.deg2rad(rng.uniform(-2, 2))
        Rwc = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]])
        twc = np.array([rng.normal(0, 0.05), rng.normal(0, 0.02), 1.0 * k])
        poses[k, :9] = Rwc.T.ravel(); poses[k, 9:] = -Rwc.T @ twc
    pts = np.stack([rng.uniform(-20, 20, n_pts), rng.uniform(-5, 5, n_pts), rng.uniform(2, 62, n_pts)], 1)
    parts = []
    for k in range(n_kf):
        pc = pts @ poses[k, :9].reshape(3, 3).T + poses[k, 9:]
        z = pc[:, 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            u = fx * pc[:, 0] / z + cx; v = fy * pc[:, 1] / z + cy
        idx = np.nonzero((z > 0.5) & (z < 80) & (u >= 0) & (u < 1024) & (v >= 0) & (v < 352))[0]
        e = np.zeros(len(idx), EDGE_DTYPE)
        sig = 1.2 ** rng.integers(0, 8, len(idx))
        e["pose"], e["point"], e["stereo"] = k, idx, rng.random(len(idx)) < 0.8
        e["obs"][:, 0] = u[idx] + rng.normal(0, sig); e["obs"][:, 1] = v[idx] + rng.normal(0, sig)
        e["obs"][:, 2] = u[idx] - bf / z[idx] + rng.normal(0, sig)
        e["inv_sigma2"] = 1.0 / (sig * sig)
        parts.append(e)
    return poses, pts, np.concatenate(parts), (fx, fy, cx, cy, bf)


def tracking_config(last, fp, intr, device, reps=20):
    """The tracking thread's per-frame solver work on the timed frame's own keys (Tracking::TrackWithMotionModel, reference
    src/orbslam/Tracking.cc:603-617): ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th = 7, bMono = false) and
    Optimizer::PoseOptimization(&mCurrentFrame).  The last frame's map points are its stereo keys unprojected (UnprojectStereo),
    the current frame holds the same keys seen from a pose 5 cm / 0.2 deg away; the motion model predicts that pose with a 1 cm
    error.  Mean ms per call over `reps` calls, each search on a NEWLY built frame view (a frame is new every image)."""
    from sivo_amd import matcher, optimizer
    from sivo_amd.optimizer import EDGE_DTYPE
    fx, fy, cx, cy, bf = intr
    keys, desc, right, depth = last["keys"], last["desc"], last["right"], last["depth"]
    ex = fp.ex_l
    scale, sigma2, inv_sigma2 = ex.GetScaleFactors(), ex.GetScaleSigmaSquares(), ex.GetInverseScaleSigmaSquares()
    has = depth > 0
    Xw = np.stack([(keys["x"] - cx) * depth / fx, (keys["y"] - cy) * depth / fy, depth], 1).astype(np.float64)      # last frame at the origin
    # the current camera: yaw 0.2 deg, 5 cm forward; its keys = the projections of the points (+ the keys without depth as they are)
    yaw = np.deg2rad(0.2)
    Rcw = np.array([[np.cos(yaw), 0, -np.sin(yaw)], [0, 1, 0], [np.sin(yaw), 0, np.cos(yaw)]]); tcw = np.array([0.0, 0.0, -0.05])
    Xc = Xw @ Rcw.T + tcw
    cur = keys.copy(); cur_right = right.copy()
    with np.errstate(divide="ignore", invalid="ignore"):
        cur["x"][has] = (fx * Xc[has, 0] / Xc[has, 2] + cx).astype(np.float32); cur["y"][has] = (fy * Xc[has, 1] / Xc[has, 2] + cy).astype(np.float32)
        cur_right[has] = (cur["x"][has] - bf / Xc[has, 2]).astype(np.float32)
    inside = (cur["x"] >= 0) & (cur["x"] < 1024) & (cur["y"] >= 0) & (cur["y"] < 352)
    cur, cur_desc, cur_right = cur[inside], desc[inside], cur_right[inside]
    # motion-model prediction: 1 cm off
    t_pred = tcw + np.array([0.01, 0.0, 0.0])
    Xp = (Xw @ Rcw.T + t_pred).astype(np.float32)
    valid = (has & (Xp[:, 2] > 0)).astype(np.uint8)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv_z = np.where(valid, 1.0 / Xp[:, 2], 0).astype(np.float32)
        u = (fx * Xp[:, 0] * inv_z + cx).astype(np.float32); v = (fy * Xp[:, 1] * inv_z + cy).astype(np.float32)
    bounds = (0.0, 1024.0, 0.0, 352.0)
    occ = np.full(len(cur), -1, np.int32); obs = np.ones(len(keys), np.int32)
    args = (valid, u, v, inv_z, keys["octave"], keys["angle"], desc, obs, 7.0, False, False, bf, True, occ)

    def search():
        F = matcher.MatchFrame(cur, cur_right, cur_desc, bounds, scale, sigma2, inv_sigma2, device=device)
        return matcher.search_by_projection_frame(F, *args)
    nm, match, _ = search()
    t0 = time.perf_counter()
    for _ in range(reps):
        search()
    t_search = (time.perf_counter() - t0) / reps
    # PoseOptimization on those matches (Optimizer.cc:273-491: stereo edge where mvuRight >= 0, else mono)
    k = np.nonzero(match >= 0)[0]
    edges = np.zeros(len(k), EDGE_DTYPE)
    edges["point"] = match[k]; edges["stereo"] = cur_right[k] >= 0
    edges["obs"][:, 0] = cur["x"][k]; edges["obs"][:, 1] = cur["y"][k]; edges["obs"][:, 2] = np.where(cur_right[k] >= 0, cur_right[k], 0)
    edges["inv_sigma2"] = inv_sigma2[cur["octave"][k]]
    pose0 = np.concatenate([Rcw.ravel(), t_pred])
    g = optimizer.pose_optimize(pose0, Xw, edges, intr)
    t0 = time.perf_counter()
    for _ in range(reps):
        g = optimizer.pose_optimize(pose0, Xw, edges, intr)
    t_pose = (time.perf_counter() - t0) / reps
    return {"name": "tracking thread, per frame (Tracking::TrackWithMotionModel, Tracking.cc:603-617) on the timed frame's semantic keys: "
                    "ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, 7, false) + Optimizer::PoseOptimization(&mCurrentFrame)",
            "metric": "ms per call (mean of %d, through the Python bindings; host gather of the projections not included)" % reps,
            "search_by_projection_ms": round(1e3 * t_search, 4), "pose_optimization_ms": round(1e3 * t_pose, 4),
            "value": round(1e3 * (t_search + t_pose), 4),
            "last_frame_map_points": int(valid.sum()), "current_frame_keys": int(len(cur)), "matches": int(nm),
            "pose_edges": int(len(k)), "pose_inliers": int(g["inliers"]), "lm_iterations": int(g["iterations"]), "lm_trials": int(g["trials"]),
            "translation_error_m": round(float(np.abs(g["pose"][9:] - tcw).max()), 6),
            "note": "each search builds a new frame view (sivo_mframe_create: grid + ONE upload into a pooled slab, no allocation) and runs one launch + one "
                    "read-back; PoseOptimization is one launch of one persistent workgroup reading its edges from pinned host memory, no allocation, no copy calls",
            "parity": "tests/test_pin_matcher.py (144 cases against the reference's ORBmatcher.cc compiled untouched, device leg in -m gpu), tests/test_gpu_search.py; "
                      "tests/test_pin_optimizer.py (22 cases against the reference's Optimizer.cc compiled untouched), tests/test_gpu_ba_solve.py (1e-7 vs the oracle)"}


def launch_command(argv, n, port=None):
    """The command `python bench.py --gpus N ...` re-executes itself as when it was started plainly (no torch.distributed.run around it):
    one rank per GPU of this node, rendezvous on 127.0.0.1 (the container's hostname may not resolve)."""
    if port is None:
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
            os.path.abspath(__file__)] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--T", type=int, default=12, help="Monte-Carlo samples per frame (12 = BASELINE configs[2], 48 = configs[3])")
    ap.add_argument("--net", default="standard", choices=["standard", "basic"])
    ap.add_argument("--height", type=int, default=352)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--no-orb", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial", action="store_true", help="N = 1: one frame in flight (the host tail of a frame is not overlapped with the next frame's device work)")
    ap.add_argument("--configs", default="all", help="N = 1 only: which further BASELINE configs to measure after the main one and "
                    "report under \"configs\": all | none | comma list of basic,t48,ba,host,track,shards (rocprofv3 runs use one at a time)")
    ap.add_argument("--per-layer", action="store_true", help="print the per-layer event timings to stderr")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # started plainly (`python bench.py --gpus N`): become the launcher of N ranks; rank 0 prints the one JSON line to this stdout
        cmd = launch_command(sys.argv[1:], args.gpus)
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # (dmabuf IPC: what RCCL needs on this driver)
        env.setdefault("OMP_NUM_THREADS", "8")
        os.execvpe(cmd[0], cmd, env)

    import torch
    import torch.distributed as dist
    from sivo_amd import netspec, orb, parallel, weights as wts
    from sivo_amd._lib import require_gpu
    from sivo_amd.segnet import BayesianSegNet

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} (or plainly: bench.py launches its own ranks)")
    if not torch.cuda.is_available():
        raise SystemExit(f"bench.py rank {rank} of {world}: no HIP device visible (there is no CPU path)")
    require_gpu()
    # SIVO_BENCH_SHARE_GPU=1 + SIVO_BENCH_BACKEND=gloo: rehearse the N>1 path on a single-GPU box (every rank on
    # cuda:0, reduction through gloo).  The driver's multi-GPU runs use neither: one rank per GPU over RCCL.
    if os.environ.get("SIVO_BENCH_SHARE_GPU") == "1":
        local = local % torch.cuda.device_count()
    backend = os.environ.get("SIVO_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    T, H, W = args.T, args.height, args.width
    sample0, n_local = parallel.shard_samples(T, world, rank)
    t_alloc = max(2, parallel.max_shard(T, world))

    def build_net(kind, t):
        text = (netspec.standard_prototxt if kind == "standard" else netspec.basic_prototxt)(t, H, W)
        layers = netspec.parse_layers(text)        # (the oracle is imported by the cpu_baseline leg only)
        w = wts.synth_weights(layers, 42)
        return text, w, BayesianSegNet(prototxt=text, weights=wts.pack(layers, w), T=t, device=local)

    text, w, sn = build_net(args.net, t_alloc)
    _g = sn.guard_report()       # the load-time accuracy guard of the matrix-core layers: what THESE weights cost
    guard_summary = {"layers_guarded": len(_g["layers"]), "largest_layer_rel_err": max([r["rel_err"] for r in _g["layers"]], default=None),
                     "predicted_logit_rel_err": _g["predicted"], "budget": _g["budget"], "plans": _g["builds"], "layers_rerouted": sum(1 for r in _g["layers"] if r["level"]),
                     "ms_at_construction": round(_g["ms"], 1),
                     "note": "per layer: max |production kernel - direct fp32 kernel| / max |direct fp32| on the same input (two built-in frames x 2 MC samples); predicted = 0.5 sqrt(sum err^2) "
                             "of the logit scale against 1e-3 / 30; tests/test_gpu_segnet.py::test_accuracy_guard_measures_every_layer_and_reroutes_an_inaccurate_plan"}
    bgr, left, right = make_inputs(H, W)
    d_bgr = torch.from_numpy(bgr).cuda()
    d_left = torch.from_numpy(left).cuda()
    d_right = torch.from_numpy(right).cuda()
    prob_sum = torch.zeros((sn.classes, H, W), dtype=torch.float32, device="cuda")

    def new_maps():
        return (torch.empty((H, W), dtype=torch.uint8, device="cuda"), torch.empty((H, W), dtype=torch.float64, device="cuda"),
                torch.empty((H, W), dtype=torch.float64, device="cuda"))
    maps = new_maps()
    do_orb = (rank == 0) and not args.no_orb
    stats = {"kps": 0, "matches": 0, "recomputed": 0, "gate_s": 0.0, "gate_n": 0, "selected": 0, "last": None}
    from sivo_amd import selection
    # the reference's own configuration of this sequence (config/kitti/KITTI00-02.yaml:8-11,25,38): Camera.fx / fy / cx / cy / bf and
    # ThEntropyReduction = 4 bits; a pose covariance of the size PoseOptimization leaves (1e-4 rad^2 / m^2)
    KFX = KFY = 718.856; KCX, KCY, KBF = 498.692, 173.215, 386.1448; KBL = KBF / KFX
    STATE_COV = np.eye(6) * 1e-4
    GATE_TH = 4.0
    # Frame.cc:125-174 on the device (sivo_amd/frame.py): the network is enqueued FIRST, the two extractors and the matching of
    # every left key run beside it, the semantic filter + median cull wait for the class map
    from sivo_amd.frame import StereoFramePipeline
    orb_delay = float(os.environ.get("SIVO_BENCH_ORB_DELAY_MS", "0")) * 1e-3       # experiment: start ORB this long after the network
    # SIVO_BENCH_ORB_MODE=0..3: sivo_orb_set_launch_mode of the frame's two extractors (A/B of the one-launch forms beside the network)
    orb_mode = os.environ.get("SIVO_BENCH_ORB_MODE")
    fp = StereoFramePipeline(device=local, start_delay_s=orb_delay, orb_launch_mode=None if orb_mode is None else int(orb_mode)) if do_orb else None
    tail_probe = [] if os.environ.get("SIVO_BENCH_TAIL_PROBE") else None         # experiment: host time of the cull behind the class map

    rank_events = []         # N > 1: (start, band done, gather done, forward done, all-reduce done) of every frame on this rank's stream
    # N > 1: the sample-invariant prefix in row bands over the ranks + one all-gather of the slots instead of N recomputations
    # (SIVO_BENCH_BANDS=0 keeps the recomputation)
    banded = world > 1 and os.environ.get("SIVO_BENCH_BANDS", "1") != "0"
    # The band of frame k + 1 (and its all-gather) is issued on a side stream while frame k's samples run: a band's kernels leave most
    # of the chip idle (one work item per CU at a fifth of the rows), the per-sample kernels of one or two samples do too, and the frames
    # are independent (the next image is needed one frame early — the same extra frame of latency the N = 1 loop takes for its host
    # tail).  SIVO_BENCH_BAND_OVERLAP=0: band -> all-gather -> samples in sequence on one stream.
    band_overlap = banded and os.environ.get("SIVO_BENCH_BAND_OVERLAP", "1") != "0"
    if banded:
        band_plan = sn.prefix_bands(world)
        nbuf = 2 if band_overlap else 1
        my_slot = [torch.zeros(band_plan["slot_bytes"], dtype=torch.uint8, device="cuda") for _ in range(nbuf)]
        all_slots = [torch.zeros((world, band_plan["slot_bytes"]), dtype=torch.uint8, device="cuda") for _ in range(nbuf)]
        slot_views = [[a[r] for r in range(world)] for a in all_slots]
        side = torch.cuda.Stream() if band_overlap else None
        band_ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(nbuf)]      # band start / band done / gathered
        consumed = [torch.cuda.Event() for _ in range(nbuf)]                                             # the samples that read this buffer are enqueued
        band_state = {"cur": 0, "primed": False}

        def gather_slots(b):
            # RCCL: straight into the (world, slot) tensor; gloo (the one-GPU rehearsal) has no all_gather_into_tensor for device tensors
            if backend == "nccl":
                dist.all_gather_into_tensor(all_slots[b].view(-1), my_slot[b])
            else:
                dist.all_gather(slot_views[b], my_slot[b])

        def issue_band(b):
            """This rank's band of the prefix + the all-gather of every rank's slot into buffer b, on the side stream."""
            with torch.cuda.stream(side):
                side.wait_event(consumed[b])             # (the frame that last read buffer b)
                band_ev[b] = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                band_ev[b][0].record()
                sn.prefix_band_into(d_bgr, rank, world, my_slot[b])
                band_ev[b][1].record()
                gather_slots(b)
                band_ev[b][2].record()

    # The rank that runs ORB (rank 0) keeps TWO frames in flight.  A frame's device work (network [+ all-reduce + finalize], ORB,
    # matching) is enqueued; its host tail — semantic filter, median cull, entropy gate, ≈0.35 ms during which the GPU would otherwise
    # idle — runs after the NEXT frame's device work has been enqueued.  Same per-frame results (the frames are independent:
    # Frame::Frame needs the image only), one frame more latency; --serial restores the strict sequence and the N = 1 line reports
    # both rates.  (The other ranks of an N > 1 run never wait inside the loop anyway.)
    pipelined = do_orb and not args.serial
    slots = 2 if pipelined else 1
    maps_s = [maps] + [new_maps() for _ in range(slots - 1)]
    cls_pin = [torch.empty((H, W), dtype=torch.uint8).pin_memory() for _ in range(slots)] if do_orb else None
    done_ev = [torch.cuda.Event() for _ in range(slots)]
    inflight = []            # [(slot, seed, pending)] issued and not completed

    def network(seed, out):
        """This rank's share of the frame's network work, enqueued on the current stream; ORB (rank 0) is started beside it."""
        if world == 1:
            # one device holds all T samples: segmentImage on device-resident data (f64 mean, no probability sum in memory)
            sn.segment_into(d_bgr, seed, out)           # asynchronous: ~65 launches enqueued in ~0.5 ms
            return fp.start_orb(d_left, d_right) if do_orb else None
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record()
        if band_overlap:
            if not band_state["primed"]:
                consumed[0].record(); consumed[1].record()
                issue_band(0)
                band_state["primed"] = True
            b = band_state["cur"]
            ev[1], ev[2] = band_ev[b][1], band_ev[b][2]          # (this frame's band ran during the previous frame; ev[0] -> ev[3] is what the main stream spent)
            ev_band0 = band_ev[b][0]
            torch.cuda.current_stream().wait_event(band_ev[b][2])
            issue_band(1 - b)                                    # the NEXT frame's band, beside this frame's samples
            if n_local:
                sn.forward_banded_into(all_slots[b], world, seed, prob_sum, n_samples=n_local, sample0=sample0)
            else:
                prob_sum.zero_()                       # more ranks than samples: contribute nothing
            consumed[b].record()
            band_state["cur"] = 1 - b
            ev.append(ev_band0)
        elif banded:
            sn.prefix_band_into(d_bgr, rank, world, my_slot[0])
            ev[1].record()
            gather_slots(0)
            ev[2].record()
            if n_local:
                sn.forward_banded_into(all_slots[0], world, seed, prob_sum, n_samples=n_local, sample0=sample0)
            else:
                prob_sum.zero_()                       # more ranks than samples: contribute nothing
        else:
            ev[1].record(); ev[2].record()
            if n_local:
                sn.forward_into(d_bgr, seed, prob_sum, n_samples=n_local, sample0=sample0)
            else:
                prob_sum.zero_()
        ev[3].record()
        pending = fp.start_orb(d_left, d_right) if do_orb else None
        parallel.all_reduce_prob_sum(prob_sum)
        ev[4].record()
        sn.finalize(prob_sum, t_total=T, out=out)
        rank_events.append(ev)
        return pending

    def issue(seed):
        slot = (inflight[-1][0] + 1) % slots if inflight else 0
        pending = network(seed, maps_s[slot])
        cls_pin[slot].copy_(maps_s[slot][0], non_blocking=True)      # 360 KB D2H behind the frame's last kernel
        done_ev[slot].record()
        inflight.append((slot, seed, pending))

    def complete():
        slot, seed, pending = inflight.pop(0)
        done_ev[slot].synchronize()                       # this frame's maps are complete (a later frame may still be running)
        if world == 1 and sn.take_overflow():
            # (one pinned word) an activation left the fp16 range in a frame issued so far: drain, and every frame still in flight
            # once more (the first of them runs without f16x3, the scales back off).  [N > 1: a rank cannot redo a collective on its
            # own; sivo_segnet_create_multi is the multi-device form that recomputes, DESIGN 4]
            torch.cuda.synchronize()
            sn.take_overflow()                            # what the frames still in flight raised meanwhile belongs to the same event
            for sl, sd, _ in [(slot, seed, None)] + inflight:
                sn.segment_into(d_bgr, sd, maps_s[sl])
                cls_pin[sl].copy_(maps_s[sl][0], non_blocking=True)
                done_ev[sl].record()
                stats["recomputed"] += 1
            done_ev[slot].synchronize()
        cls_host = cls_pin[slot].numpy()
        t0 = time.perf_counter()
        r = fp.finish(pending, cls_host)
        # entropy feature selection over the frame's semantic keys (Tracking.cc:934-1023): the keys are on the host, the
        # f64 entropy map stays where the network wrote it (sivo_entropy_gate_map_dev)
        t1 = time.perf_counter()
        d = r["depth"]; k = r["keys"]
        xyz = np.stack([(k["x"] - KCX) * d / KFX, (k["y"] - KCY) * d / KFY, d], 1).astype(np.float64)
        _, _, acc = selection.entropy_gate_map_dev(k, d, xyz, maps_s[slot][2], STATE_COV, KFX, KFY, KBL, fp.ex_l.GetScaleSigmaSquares(), GATE_TH)
        stats["gate_s"] += time.perf_counter() - t1; stats["gate_n"] += 1
        stats["selected"] = int(acc.sum())
        if tail_probe is not None:
            tail_probe.append(time.perf_counter() - t0)
        stats["kps"], stats["matches"] = r["semantic_keys"], r["stereo_matches"]
        stats["last"] = r
        if keep is not None:            # (the self-check below: a frame's maps, keys, matches and selection by seed)
            keep[seed] = (maps_s[slot][0].clone(), maps_s[slot][1].clone(), maps_s[slot][2].clone(), r["keys"].copy(), r["right"].copy(), r["depth"].copy(), acc.copy())

    keep = None

    def flush():
        while inflight:
            complete()

    def frame(seed):
        if do_orb:
            issue(seed)
            if len(inflight) >= slots:
                complete()
        else:
            network(seed, maps)                          # (no host work depends on the result: nothing to wait for inside the loop)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The frame loop is host code in Python: a generational collection of CPython's GC in the middle of a frame stalls the
    # thread hand-over to the ORB extractor threads by ~3 ms (measured: tools/frame_timeline.py — whether a full collection
    # lands in every frame depends on the exact count of live objects).  Everything set up so far is long-lived: move it
    # out of the collector's reach, as a real-time host loop would.
    import gc
    gc.collect()
    gc.freeze()

    # Every 8th timed frame carries HIP events around its MFMA kernels (that is what `roofline` is measured on; those frames
    # issue the forward in one lane so that a launch has the GPU to itself); SIVO_BENCH_NO_EVENTS=1: none at all.
    PROFILE_EVERY = 8
    events = os.environ.get("SIVO_BENCH_NO_EVENTS") != "1"
    elapsed, prof_timed, prof, n_detail = time_segnet(sn, frame, args.steps, args.warmup, barrier, PROFILE_EVERY, events, flush)
    pipeline_check = None
    if pipelined and world == 1:
        # self-check of the two-frames-in-flight loop: the frame with seed 777 between two others, against the same frame alone
        keep = {}
        for sd in (776, 777, 778):
            frame(sd)
        flush()
        two = keep[777]
        keep = {}
        slots = 1
        frame(777); flush()
        one = keep[777]
        slots = 2
        keep = None
        pipeline_check = bool(all(torch.equal(x, y) if torch.is_tensor(x) else np.array_equal(x, y) for x, y in zip(two, one)))
    serial_fps = None
    if pipelined and world == 1:              # the same loop with one frame in flight, for the record
        slots = 1
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            frame(4000 + i)
        flush()
        barrier()
        serial_fps = args.steps / (time.perf_counter() - t0)
    multi = None
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        # fp16 range guard of the f16x3 layers: a rank cannot redo a collective on its own, so this loop does not
        # recompute an overflowed frame the way the N = 1 loop does (sivo_segnet_create_multi, the in-handle multi-device form, does).
        # It must not report a rate made of wrong frames either: any rank's flag fails the run loudly.
        ov = torch.tensor([1.0 if sn.take_overflow() else 0.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(ov, op=dist.ReduceOp.MAX)
        if float(ov.item()) > 0:
            raise SystemExit("an activation left the fp16 range of an f16x3 layer on some rank during the timed frames: their maps are wrong and "
                             "this N > 1 loop does not recompute them — no line is reported (SIVO_GEMM=x6 runs the fp32-range kernels)")
        # where a frame's time goes on every rank: its own shard's forward (prefix + n_local samples), then the all-reduce, which
        # ends when the slowest rank has arrived — so "all-reduce" on a light rank is mostly waiting, on the heaviest rank the wire time
        torch.cuda.synchronize()
        evs = rank_events[args.warmup:args.warmup + args.steps]
        # overlapped bands: a frame's band / gather events come from the side stream (issued one frame earlier): band = e[5] -> e[1],
        # gather = e[1] -> e[2], and "forward" = what the main stream spent between the frame's start and the end of its samples
        spans = ((5, 1), (1, 2), (0, 3), (3, 4)) if band_overlap else ((0, 1), (1, 2), (2, 3), (3, 4))
        mine = torch.tensor([float(np.mean([e[a].elapsed_time(e[b]) for e in evs])) for a, b in spans], dtype=torch.float64, device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        multi = {"prefix": ("row bands over the ranks + one all-gather of the slots (DESIGN 4)" + (", issued one frame ahead on a side stream beside the previous frame's samples" if band_overlap else ""))
                           if banded else "recomputed on every rank",
                 "prefix_band_ms_per_rank": [round(float(t[0]), 3) for t in allr] if banded else None,
                 "band_allgather_incl_wait_ms_per_rank": [round(float(t[1]), 3) for t in allr] if banded else None,
                 "band_allgather_bytes_per_rank": int(band_plan["slot_bytes"]) if banded else None,
                 "forward_ms_per_rank": [round(float(t[2]), 3) for t in allr], "allreduce_incl_wait_ms_per_rank": [round(float(t[3]), 3) for t in allr],
                 "allreduce_wire_ms": round(min(float(t[3]) for t in allr), 3),
                 "allreduce_bytes": int(prob_sum.numel() * 4),
                 "note": "HIP events on each rank's stream, mean over the timed frames: prefix band = the rank's rows of the sample-invariant prefix; band all-gather "
                         "incl. wait = until every rank's slot has arrived; forward = unpacking + the rank's samples (with the prefix recomputed: prefix + samples); "
                         "all-reduce incl. wait = from the end of the rank's forward to the end of the collective (waiting for the slowest rank included); "
                         "wire = the smallest of those (the rank that arrives last waits for nobody)"}

    if rank == 0 and tail_probe:
        print(f"host tail (semantic filter + median cull) mean {1e3 * float(np.mean(tail_probe)):.3f} ms over {len(tail_probe)} frames", file=sys.stderr)
    if rank == 0:
        fps = args.steps / elapsed
        ms_frame = 1e3 * elapsed / args.steps
        if args.per_layer:
            for p in prof:
                ms = p["ms_total"] / max(p["launches"], 1)
                if not p["launches"]:
                    continue
                fl = p["flops_per_sample"] * p["samples"]
                print(f'{p["layer"]:14s} {p["kernel"]:42s} N={p["samples"]:2d} {ms:8.4f} ms  {fl / ms / 1e9 if ms else 0:7.1f} TFLOP/s  '
                      f'{p["bytes_per_sample"] * p["samples"] / ms / 1e6 if ms else 0:8.1f} GB/s(alg)', file=sys.stderr)
        n_timed = len(range(0, args.steps, PROFILE_EVERY))
        note = (f"dominant kernel: HIP events on its launch stream in every {PROFILE_EVERY}th of the {args.steps} timed frames (those frames run the forward in one "
                f"lane, one launch per layer; the others split the samples over SIVO_LANES sample groups on separate streams, default two); kernels_ms_per_frame: {n_detail} further untimed single-lane "
                "frames with every kernel bracketed.  achieved / frac = direct-convolution FLOPs (SURVEY 8d) / kernel time against the dense fp16 / bf16 peak; "
                "executed_* = the matrix-core products issued (the Winograd-domain GEMM of F(4x4,3x3) multiplies 1/4 of the direct convolution's products, the direct "
                "f16x3 kernel all of them; each fp32 product is three fp16 MFMA products — so executed = 3/4 of algorithmic for the GEMM and 3x for the direct "
                "kernel).  mfma_kernels lists every matrix-core kernel of the frame with the same figures: the two f16x3 kernels take about the same time, and "
                "which of them is 'dominant' can change from run to run")
        roofline = mfma_roofline(prof_timed, prof, n_timed, n_detail, ms_frame, note)
        out = {"metric": "frames/sec, SIVO per-frame path (ORB+SegNet T=%d+entropy) %dx%d" % (T, H, W),
               "value": round(fps, 4), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_frame, 3), "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f32 (f16x3)", "data": "synthetic",
               "arithmetic": ("activations, weights, transforms, accumulators and outputs fp32; the batched GEMM of the Winograd F(4x4,3x3) layers and the direct "
                              "3x3 kernel of the layers with <= 128 channels (<= 256 in the sample-invariant prefix) multiply fp32 operands as fp16 hi + lo pairs "
                              "(power-of-two layer scales, 2^-22 relative) / 3 fp16 MFMA products with fp32 accumulation (error at the level of the fp32 FMA chain "
                              "they replace: tests/test_gpu_h3_gemm.py, tests/test_gpu_conv3_h3.py, tests/test_gpu_segnet_fullsize.py; SIVO_GEMM=x6 / f32 select the "
                              "bf16x6 / fp32 MFMA kernels, SIVO_D3=0 the fp32 fused Winograd kernels); MC mean / confidence / entropy in f64"),
               "config": {"workload": f"full per-frame path: ORB 2000x8 stereo + SegNet-{args.net} T={T} MC-dropout + entropy maps + semantic key filter + stereo match, {H}x{W}, synthetic stereo pair, seeded random weights",
                          "T": T, "samples_per_rank": [parallel.shard_samples(T, world, r)[1] for r in range(world)],
                          "frames_in_flight": 2 if pipelined else 1,
                          "pipeline": ("a frame's host tail (semantic filter, median cull, entropy gate) runs while the NEXT frame's device work is already enqueued; "
                                       "every frame's results are those of the serial loop (tests/test_gpu_frame_e2e.py), latency + 1 frame; serial_fps = the same loop with one frame in flight")
                                      if pipelined else None,
                          "serial_fps": round(serial_fps, 4) if serial_fps else None,
                          "pipelined_frame_equals_serial_frame": pipeline_check,      # maps (u8 / f64 / f64), keys, mvuRight, mvDepth, selection: bit for bit
                          "orb": bool(do_orb), "semantic_keys": stats["kps"], "stereo_matches": stats["matches"],
                          "entropy_gate": ({"in_timed_frame": True, "keys_selected": stats["selected"], "threshold_bits": GATE_TH,
                                            "ms_per_call": round(1e3 * stats["gate_s"] / max(stats["gate_n"], 1), 4),
                                            "parity": "tests/test_gpu_match_ba.py::test_entropy_gate_matches_oracle, tests/test_pin_helpers.py (the reference's sivo_helpers chain)"}
                                           if do_orb else None),
                          "algorithmic_gflop_per_frame": round((sn.flops_shared + T * sn.flops_per_sample) / 1e9, 2),
                          "reference_equivalent_gflop_per_frame": round(T * (sn.flops_shared + sn.flops_per_sample) / 1e9, 2),
                          "gemm": dict(zip(("mode", "fp16_overflow_frames"), sn.gemm_status()[:2]), frames_recomputed=stats["recomputed"]),
                          "accuracy_guard": guard_summary,
                          "parity": "tests/test_gpu_frame_e2e.py (this frame end to end against the oracle pipeline), tests/test_gpu_segnet_fullsize.py (this network "
                                    "configuration, every logit, oracle-checked), tests/test_gpu_orb.py, tests/test_gpu_match_ba.py"},
               "roofline": roofline}

        if world == 1:
            out["membound"] = membound_block(prof, fp, d_left, H, W, T, sn.classes)
        if multi:
            out["multi_gpu"] = multi

    # ------------------------------------------------------------------------------------------ further configs (N = 1)
    want = set() if (world > 1 or args.configs == "none") else set(("basic,t48,ba,host,track,shards" if args.configs == "all" else args.configs).split(","))
    extra = []
    if rank == 0 and want:
        if "host" in want:
            # the reference boundary (segmentImage: host image in, host maps out): H2D 1.08 MB + D2H 6.1 MB inside the loop
            for i in range(2):
                sn.segment_image(bgr, seed=i)
            t0 = time.perf_counter()
            nh = max(5, args.steps // 2)
            for i in range(nh):
                sn.segment_image(bgr, seed=100 + i)
            out["host_boundary_fps"] = round(nh / (time.perf_counter() - t0), 3)
            out["host_boundary_note"] = ("sivo_segnet_segment through the C ABI: pageable host BGR frame in, classes / confidence / entropy maps out "
                                         "(PCIe both ways, no ORB); never the reported value")
        del sn
        torch.cuda.empty_cache()

        def segnet_config(name, kind, t, steps, parity, tag):
            _, _, net = build_net(kind, t)
            m = new_maps()
            el, pt_, pd_, nd = time_segnet(net, lambda seed: net.segment_into(d_bgr, seed, m), steps, 2, barrier, 8, events)
            ms = 1e3 * el / steps
            r = mfma_roofline(pt_, pd_, len(range(0, steps, 8)), nd, ms, "as the main roofline; SegNet only (no ORB)", tag)
            alg = (net.flops_shared + t * net.flops_per_sample) / 1e9
            del net
            torch.cuda.empty_cache()
            return {"name": name, "metric": "frames/sec", "value": round(steps / el, 3), "ms_per_step": round(ms, 3), "steps": steps,
                    "algorithmic_gflop_per_frame": round(alg, 2), "roofline": r, "parity": parity}
        if "basic" in want:
            extra.append(segnet_config("BASELINE configs[1]: Bayesian SegNet Basic, T=6, 352x1024, 1 MI355X (SegNet + MC maps)", "basic", 6, 20,
                                       "tests/test_gpu_segnet_fullsize.py [basic-6-*] (every logit within 1e-3 of the oracle, default lanes)", "basic"))
        if "t48" in want:
            extra.append(segnet_config("BASELINE configs[3] on ONE GPU: SegNet Standard T=48 (the 8-GPU form shards 6 samples per rank)", "standard", 48, 6,
                                       "tests/test_gpu_segnet_fullsize.py::test_t48_and_its_shards_at_full_size (T = 48 in one handle and the 6-sample shards, oracle-checked), tests/test_distributed_cpu.py", "t48"))
        if "shards" in want:
            # what ONE rank computes per frame when the T samples are sharded over N ranks (its band of the prefix + its samples + finalize), measured
            # on this GPU: the ceiling of the strong-scaling curve the driver's N = 2, 4, 8 runs can reach (all-gather / all-reduce wire time on top)
            def timed_ms(fn):
                for i in range(3):
                    fn(i)
                barrier()
                t0 = time.perf_counter()
                for i in range(10):
                    fn(10 + i)
                barrier()
                return round(1e2 * (time.perf_counter() - t0), 3)

            def shard_rows(net, t_frame, with_orb):
                ps = torch.zeros((net.classes, H, W), dtype=torch.float32, device="cuda")
                m = new_maps()
                rows = []
                for nr in (1, 2, 4, 8):
                    shares = [parallel.shard_samples(t_frame, nr, r)[1] for r in range(nr)]
                    nl = max(shares)
                    heavy = max(r for r in range(nr) if shares[r] == nl)      # (the last rank: the largest band of the prefix as well)
                    def one(seed, nl=nl):
                        net.forward_into(d_bgr, seed, ps, n_samples=nl, sample0=0)
                        net.finalize(ps, t_total=t_frame, out=m)
                    row = {"ranks": nr, "samples_per_rank": shares, "samples_on_the_heaviest_rank": nl, "ms_per_frame_prefix_recomputed": timed_ms(one)}
                    if nr > 1:
                        # the same rank with the prefix in row bands: ITS band + unpacking the gathered slots + its samples + finalize; the
                        # other ranks' slots are computed once, outside the timing (they arrive by all-gather)
                        plan = net.prefix_bands(nr)
                        slots = torch.zeros((nr, plan["slot_bytes"]), dtype=torch.uint8, device="cuda")
                        for r in range(nr):
                            net.prefix_band_into(d_bgr, r, nr, slots[r])
                        def one_b(seed, nl=nl, nr=nr, slots=slots, br=heavy):
                            net.prefix_band_into(d_bgr, br, nr, slots[br])
                            net.forward_banded_into(slots, nr, seed, ps, n_samples=nl, sample0=0)
                            net.finalize(ps, t_total=t_frame, out=m)
                        row["ms_per_frame_band_in_sequence"] = timed_ms(one_b)
                        # ... and as the N > 1 loop runs it: the NEXT frame's band on a side stream beside this frame's samples
                        side_s = torch.cuda.Stream()
                        slots_next = slots.clone()
                        def one_o(seed, nl=nl, nr=nr, slots=slots, slots_next=slots_next, side_s=side_s, br=heavy, orb_too=False):
                            side_s.wait_stream(torch.cuda.current_stream())
                            with torch.cuda.stream(side_s):
                                net.prefix_band_into(d_bgr, br, nr, slots_next[br])
                            if nl:
                                net.forward_banded_into(slots, nr, seed, ps, n_samples=nl, sample0=0)
                            else:
                                ps.zero_()
                            pend = fp.start_orb(d_left, d_right) if orb_too else None
                            net.finalize(ps, t_total=t_frame, out=m)
                            torch.cuda.current_stream().wait_stream(side_s)
                            if orb_too:           # rank 0's host side of the frame: class map to the host, semantic filter + cull + matching results
                                fp.finish(pend, m[0].cpu().numpy())
                        row["ms_per_frame"] = timed_ms(one_o)
                        row["band_rows_of_the_image"] = plan["input_rows"][heavy][1] - plan["input_rows"][heavy][0]
                        row["allgather_bytes_per_rank"] = plan["slot_bytes"]
                        if with_orb and fp is not None:
                            # rank 0 (serial, one frame in flight): band 0 + ITS samples + finalize + both ORB extractors + matching + the host tail
                            import functools
                            row["rank0_samples"] = shares[0]
                            row["rank0_ms_per_frame_with_orb"] = timed_ms(functools.partial(one_o, nl=shares[0], br=0, orb_too=True))
                            if parallel.orb_rank_is_free(t_frame, nr):       # ... and what it would take with the even split's one-or-more samples
                                even = t_frame // nr
                                row["rank0_ms_per_frame_with_orb_if_it_kept_%d_sample%s" % (even, "" if even == 1 else "s")] = timed_ms(functools.partial(one_o, nl=even, br=0, orb_too=True))
                    else:
                        row["ms_per_frame"] = row["ms_per_frame_prefix_recomputed"]
                    rows.append(row)
                for r_ in rows:
                    r_["speedup_ceiling"] = round(rows[0]["ms_per_frame"] / r_["ms_per_frame"], 2)
                    r_["speedup_ceiling_prefix_recomputed"] = round(rows[0]["ms_per_frame"] / r_["ms_per_frame_prefix_recomputed"], 2)
                return rows
            legend = ("the heaviest rank's share: its band of the prefix + unpacking + its samples + finalize; the band of the NEXT frame on a side stream beside the samples, "
                      "as the N > 1 loop issues it; no all-gather / all-reduce wire time; *_band_in_sequence = band, then samples, on one stream; *_prefix_recomputed = every rank "
                      "computing the whole prefix, as before round 5; rank0_* = rank 0's frame with ORB + matching + host tail, one frame in flight")
            shard_parity = "tests/test_gpu_prefix_bands.py (banded prefix == whole-image forward, bit for bit, world 2 / 4 / 8), tests/test_gpu_segnet_fullsize.py::test_t48_and_its_shards_at_full_size, tests/test_distributed_cpu.py"
            _, _, net = build_net(args.net, T)
            rows = shard_rows(net, T, True)
            del net
            torch.cuda.empty_cache()
            extra.append({"name": f"sample shards of the T = {T} frame on one GPU ({legend})",
                          "metric": "ms per frame of the heaviest rank", "value": rows[-1]["ms_per_frame"], "shards": rows, "parity": shard_parity})
            if T != 48 and args.net == "standard":
                # BASELINE configs[3]: T = 48 over 8 GPUs, 6 samples per rank — the balanced configuration, and its own ceiling
                _, _, net = build_net("standard", 48)
                rows48 = shard_rows(net, 48, False)
                del net
                torch.cuda.empty_cache()
                extra.append({"name": f"BASELINE configs[3] per rank: sample shards of the T = 48 frame on one GPU, 6 samples per rank at 8 ranks ({legend})",
                              "metric": "ms per frame of the heaviest rank", "value": rows48[-1]["ms_per_frame"], "shards": rows48, "parity": shard_parity})
        if "track" in want and stats["last"] is not None:
            extra.append(tracking_config(stats["last"], fp, (KFX, KFY, KCX, KCY, KBF), local))
        if "ba" in want:
            from sivo_amd import optimizer
            poses, pts, edges, intr = ba_scene()
            rng = np.random.default_rng(3)
            fixed = np.zeros(len(poses), np.uint8); fixed[:2] = 1
            P0 = poses.copy(); P0[2:, 9:] += rng.normal(0, 0.02, (len(poses) - 2, 3))
            X0 = pts + rng.normal(0, 0.05, pts.shape)
            # (the configurations above leave garbage behind — handles of 20 GB, event lists.  Collected BEFORE the warm-up call, not between it and
            # the timed calls: the collection takes long enough for the idle GPU to drop its clocks, and the first call behind it then took
            # 17 - 20 ms — the "outlier" of rounds 5 / 6, always call 0 of the 30 (profiles/r06_*_bench_line.json); `--configs ba` alone never
            # showed it.  calls_ms lists every call.)
            import gc
            gc.collect()
            torch.cuda.synchronize()
            for _ in range(2):
                optimizer.local_ba(P0, fixed, X0, edges, intr, cov_pose=19)
            ts = []
            for _ in range(30):
                t0 = time.perf_counter(); g = optimizer.local_ba(P0, fixed, X0, edges, intr, cov_pose=19); ts.append(time.perf_counter() - t0)
            t_call = float(np.mean(ts))                                         # the mean, like every other figure of the line
            nE, it = len(edges), max(g["iterations"], 1)
            alg_bytes = nE * (48 + 96 + 24 + (3 + 18 + 9 + 1 + 18 + 18) * 8) * it      # ~0.9 KB per edge per LM iteration
            extra.append({"name": "BASELINE configs[4]: local BA, 20 keyframes x 3000 map points (whole Optimizer::LocalBundleAdjustment solve on the GPU: "
                                  "per-edge residuals / Jacobians, Schur complement, LM, marginal covariance)",
                          "metric": "ms per LocalBundleAdjustment call (mean of 30)", "value": round(t_call * 1e3, 3), "min_ms": round(min(ts) * 1e3, 3),
                          "median_ms": round(float(np.median(ts)) * 1e3, 3), "std_ms": round(float(np.std(ts)) * 1e3, 3), "calls": len(ts),
                          "calls_ms": [round(t * 1e3, 2) for t in ts], "edges": int(nE), "lm_iterations": g["iterations"],
                          "ms_per_lm_iteration": round(t_call * 1e3 / it, 3),
                          "roofline": {"bound": "hbm", "achieved": round(alg_bytes / t_call / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(alg_bytes / t_call / 1e9 / HBM_PEAK_GBS, 5),
                                       "note": "algorithmic bytes (edge + pose + point in, err / Jp / Jx / W / Y per edge per LM iteration) / whole-call wall time incl. host CSR build, "
                                               "H2D / D2H and one host decision per trial: this size (36 k edges, 32 MB per iteration) is launch- and latency-bound, not bandwidth-bound"},
                          "parity": "tests/test_gpu_ba_solve.py (poses 1e-9, covariance 1e-7 vs the oracle), tests/test_gpu_match_ba.py (per-edge outputs bit-exact)"})
    if rank == 0:
        if extra:
            out["configs"] = extra
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.net, T, H, W, text, w, bgr, left, right, "")
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
