#!/usr/bin/env python3
"""GPU box: what the heaviest of 8 ranks runs per frame with the prefix in row bands (its band + unpacking + its 2 samples +
finalize) against the same rank recomputing the prefix — wall time per frame, and (under rocprofv3 --kernel-trace --stats) the
kernels behind it.    python tools/band_probe.py [world] [iters]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from sivo_amd import netspec, parallel, weights as wts
    from sivo_amd.segnet import BayesianSegNet
    from bench import make_inputs
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    mode = sys.argv[3] if len(sys.argv) > 3 else "both"
    H, W, T = 352, 1024, 12
    nl = parallel.max_shard(T, world)
    text = netspec.standard_prototxt(max(2, nl), H, W)
    layers = netspec.parse_layers(text)
    if os.environ.get("PROBE_DIAG"):         # the diagnostic build (reads SIVO_BAND_GRAPH=0: the band as eager launches)
        from sivo_amd import _lib
        with _lib.use("diag"):
            sn = BayesianSegNet(prototxt=text, weights=wts.pack(layers, wts.synth_weights(layers, 42)), T=max(2, nl))
    else:
        sn = BayesianSegNet(prototxt=text, weights=wts.pack(layers, wts.synth_weights(layers, 42)), T=max(2, nl))
    d = torch.from_numpy(make_inputs(H, W)[0]).cuda()
    ps = torch.zeros((sn.classes, H, W), dtype=torch.float32, device="cuda")
    maps = (torch.empty((H, W), dtype=torch.uint8, device="cuda"), torch.empty((H, W), dtype=torch.float64, device="cuda"),
            torch.empty((H, W), dtype=torch.float64, device="cuda"))
    plan = sn.prefix_bands(world)
    slots = torch.zeros((world, plan["slot_bytes"]), dtype=torch.uint8, device="cuda")
    for r in range(world):
        sn.prefix_band_into(d, r, world, slots[r])

    def recomputed(seed):
        sn.forward_into(d, seed, ps, n_samples=nl, sample0=0)
        sn.finalize(ps, t_total=T, out=maps)

    def banded(seed):
        sn.prefix_band_into(d, world - 1, world, slots[world - 1])
        sn.forward_banded_into(slots, world, seed, ps, n_samples=nl, sample0=0)
        sn.finalize(ps, t_total=T, out=maps)

    def band_only(seed):
        sn.prefix_band_into(d, world - 1, world, slots[world - 1])

    for name, fn in (("recomputed", recomputed), ("banded", banded), ("band only", band_only)):
        if mode != "both" and mode != name.split()[0]:
            continue
        for i in range(5):
            fn(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(iters):
            fn(10 + i)
        torch.cuda.synchronize()
        print(f"[band probe world {world}] {name}: {1e3 * (time.perf_counter() - t0) / iters:.3f} ms per frame ({nl} samples on the rank)", flush=True)


if __name__ == "__main__":
    main()
