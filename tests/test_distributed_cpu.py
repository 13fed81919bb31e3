"""CPU, world_size 2 over gloo: the N > 1 path of the per-frame pipeline — sample sharding keyed by the
global sample index, one all-reduce(SUM) of the probability sums, finalize on every rank — gives the
single-process result.  The compute leg is the oracle here (no GPU); the sharding / reduction code is
the product's (sivo_amd.parallel), the same functions bench.py uses over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as O, prototxt as oproto
from sivo_amd import netspec, parallel, weights as wts


def test_shard_samples_partition():
    for T in (1, 2, 6, 12, 48, 7):
        for world in (1, 2, 3, 4, 8, 16):
            parts = [parallel.shard_samples(T, world, r) for r in range(world)]
            covered = [s for s0, n in parts for s in range(s0, s0 + n)]
            assert covered == list(range(T))
            assert max(n for _, n in parts) == parallel.max_shard(T, world)
            assert parallel.max_shard(T, world) == -(-T // world)            # giving rank 0 nothing never lengthens the heaviest shard
            rest = [n for _, n in parts[1:]] if world > 1 and parallel.orb_rank_is_free(T, world) else [n for _, n in parts]
            assert max(rest) - min(rest) <= 1
            assert parts[0][1] == min(n for _, n in parts)                   # rank 0 (ORB + host side of the frame) never has the larger share
    assert parallel.shard_samples(48, 8, 3) == (18, 6)             # BASELINE configs[3]: 6 samples per GPU
    assert [parallel.shard_samples(12, 8, r)[1] for r in range(8)] == [0, 1, 1, 2, 2, 2, 2, 2]        # the T = 12 frame on 8 GPUs: rank 0 only runs ORB
    assert [parallel.shard_samples(12, 4, r)[1] for r in range(4)] == [3, 3, 3, 3]


def test_bench_launches_its_own_ranks_when_started_plainly():
    """`python bench.py --gpus 2` without torch.distributed.run around it (how the driver starts it) must become a 2-rank job.  There is no
    GPU here and bench.py has no CPU path: the ranks must come up, see their RANK and a WORLD_SIZE of 2 and stop at the device check."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    cmd = bench.launch_command(["--gpus", "2", "--steps", "3"], 2, port=29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=2" in cmd and cmd[-4:] == ["--gpus", "2", "--steps", "3"]
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    if torch.cuda.is_available():
        pytest.skip("the device check does not stop the ranks on a GPU box (tests/test_gpu_frame_e2e.py runs the real thing)")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    import re
    # (the launcher ends the job when the first rank fails: the other one may not get to its own message)
    assert re.search(r"bench\.py rank [01] of 2: no HIP device visible", out.stderr), out.stderr[-3000:]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, T, H, W, seed, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    text = netspec.tiny_prototxt(T, H, W)
    net = oproto.parse(text)
    w = wts.synth_weights(net["layers"], 42)
    img = np.random.default_rng(0).integers(0, 256, (H, W, 3), dtype=np.uint8)
    s0, n = parallel.shard_samples(T, world, rank)
    shard = dict(net, shape=[n, 3, H, W])
    prob_sum = torch.zeros((15, H, W), dtype=torch.float32)
    if n:
        prob = O.run_net(shard, w, O.preprocess(img, n, H, W), seed, sample0=s0)["__last__"]
        prob_sum += torch.from_numpy(prob.astype(np.float64).sum(0).astype(np.float32))
    parallel.all_reduce_prob_sum(prob_sum)
    cls, conf, ent = O.mc_finalize(prob_sum.numpy().astype(np.float64) / T)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), cls=cls, conf=conf, ent=ent, ps=prob_sum.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("T", [4, 5])
def test_two_rank_sharded_frame_equals_single_process(tmp_path, T):
    H, W, seed = 16, 32, 77
    mp.spawn(_worker, args=(2, _free_port(), T, H, W, seed, str(tmp_path)), nprocs=2, join=True)
    text = netspec.tiny_prototxt(T, H, W)
    net = oproto.parse(text)
    w = wts.synth_weights(net["layers"], 42)
    img = np.random.default_rng(0).integers(0, 256, (H, W, 3), dtype=np.uint8)
    res = O.segment(net, w, img, seed)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    assert np.array_equal(r0["ps"], r1["ps"])                      # all-reduce: every rank holds the same sum
    np.testing.assert_allclose(r0["ps"] / T, res["mean"], atol=1e-6)
    np.testing.assert_allclose(r0["conf"], res["confidence"], atol=1e-6)
    np.testing.assert_allclose(r0["ent"], res["entropy"], atol=1e-5)
    assert (r0["cls"] == res["classes"]).mean() > 0.999


# ---------------------------------------------------------------------------------------------------------------------------
# The sample-invariant prefix in row bands over the ranks: rank r computes ITS rows of the prefix output from a band of
# the image, one all-gather reassembles the prefix.  The compute leg is the oracle (plain C convolutions); the partition is
# sivo_amd.parallel.band_rows / band_input_rows, which restate the product's plan (the GPU test checks that they agree).
def _prefix_net(T, H, W):
    net = oproto.parse(netspec.standard_prototxt(T, H, W))
    cut = next(i for i, L in enumerate(net["layers"]) if L["type"] == "Dropout" and L["sample_weights_test"])
    layers = net["layers"][:cut]
    return dict(net, layers=layers), layers


def _codes(mask, w_in):
    m = mask.astype(np.int64)
    return ((m // w_in) % 2) * 2 + (m % w_in) % 2


def test_band_partition_properties():
    _, layers = _prefix_net(2, 352, 1024)
    chain = [L for L in layers if L["type"] in ("Convolution", "Pooling")]
    for world in (1, 2, 3, 4, 8, 11):
        rows = parallel.band_rows(44, world)
        assert rows[0] == 0 and rows[-1] == 44 and len(rows) == world + 1
        sizes = [b - a for a, b in zip(rows, rows[1:])]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes)
        for r in range(world):
            lo, hi = parallel.band_input_rows(chain, 352, rows[r], rows[r + 1])
            assert lo % 8 == 0 and hi % 8 == 0
            assert lo == max(0, (8 * rows[r] - 18) // 8 * 8) and hi == min(352, -((-8 * rows[r + 1] - 18) // 8) * 8)      # halo 18, aligned to 8
    assert parallel.band_rows(44, 8) == [0, 5, 10, 15, 20, 26, 32, 38, 44]


def _band_worker(rank, world, port, H, W, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net, layers = _prefix_net(2, H, W)
    w = wts.synth_weights(oproto.parse(netspec.standard_prototxt(2, H, W))["layers"], 42)
    img = np.random.default_rng(0).integers(0, 256, (H, W, 3), dtype=np.uint8)
    chain = [L for L in layers if L["type"] in ("Convolution", "Pooling")]
    rows = parallel.band_rows(H // 8, world)
    lo, hi = parallel.band_input_rows(chain, H, rows[rank], rows[rank + 1])
    band = O.run_net(dict(net, shape=[1, 3, hi - lo, W]), w, O.preprocess(img[lo:hi], 1, hi - lo, W), 0,
                     keep=["pool3", "pool1_mask", "pool2_mask", "pool3_mask"])
    # the valid rows of every exchanged blob, as the product packs them: values as they are, masks as window codes
    rmax = max(b - a for a, b in zip(rows, rows[1:]))
    parts = {}
    for name, level, w_in in (("pool3", 3, None), ("pool3_mask", 3, W // 4), ("pool2_mask", 2, W // 2), ("pool1_mask", 1, W)):
        sh = 3 - level
        x = band[name][0]
        x = _codes(x, w_in).astype(np.float32) if w_in else x
        first = (rows[rank] << sh) - (lo >> level)
        n = (rows[rank + 1] - rows[rank]) << sh
        slot = np.zeros((x.shape[0], rmax << sh, x.shape[2]), np.float32)
        slot[:, :n] = x[:, first:first + n]
        parts[name] = torch.from_numpy(slot)
    gathered = {}
    for name, t in parts.items():
        outs = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        gathered[name] = outs
    if rank == 0:
        full = {}
        for name, level in (("pool3", 3), ("pool3_mask", 3), ("pool2_mask", 2), ("pool1_mask", 1)):
            sh = 3 - level
            full[name] = np.concatenate([gathered[name][r].numpy()[:, :(rows[r + 1] - rows[r]) << sh] for r in range(world)], axis=1)
        np.savez(os.path.join(out_dir, "bands.npz"), **full)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_banded_prefix_equals_the_whole_image(tmp_path):
    """world_size 2 over gloo: each rank runs the prefix (conv1_1 .. pool3 of SegNet-Standard) on its band of the image, the valid
    rows are all-gathered, and the reassembled pooled values and pooling switches equal the whole-image prefix BIT FOR BIT."""
    H, W = 48, 64
    mp.spawn(_band_worker, args=(2, _free_port(), H, W, str(tmp_path)), nprocs=2, join=True)
    net, _ = _prefix_net(2, H, W)
    w = wts.synth_weights(oproto.parse(netspec.standard_prototxt(2, H, W))["layers"], 42)
    img = np.random.default_rng(0).integers(0, 256, (H, W, 3), dtype=np.uint8)
    ref = O.run_net(dict(net, shape=[1, 3, H, W]), w, O.preprocess(img, 1, H, W), 0, keep=["pool3", "pool1_mask", "pool2_mask", "pool3_mask"])
    got = np.load(tmp_path / "bands.npz")
    assert np.array_equal(got["pool3"], ref["pool3"][0])
    for name, w_in in (("pool3_mask", W // 4), ("pool2_mask", W // 2), ("pool1_mask", W)):
        assert np.array_equal(got[name], _codes(ref[name][0], w_in).astype(np.float32)), name
