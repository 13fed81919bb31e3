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
            assert max(n for _, n in parts) - min(n for _, n in parts) <= 1
    assert parallel.shard_samples(48, 8, 3) == (18, 6)             # BASELINE configs[3]: 6 samples per GPU


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
