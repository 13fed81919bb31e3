"""Monte-Carlo sample sharding over ranks (one process per GPU, torch.distributed; the "nccl"
backend is RCCL on ROCm).  The only axis of the per-frame path that shards is T: each MC sample is
an independent forward of the same image (reference bayesian_segnet.cpp:174-177) and the samples are
coupled only by the mean over the batch axis (:294), a linear reduction — one all-reduce(SUM) of the
classes x H x W fp32 probability sums per frame.  Dropout masks are keyed by the GLOBAL sample index,
so the result does not depend on the number of ranks."""


def orb_rank_is_free(T, world):
    """Rank 0 also runs the two ORB extractors, the stereo matching and the host side of the frame.  It takes NO samples when the other
    world - 1 ranks can take all T without the heaviest of them getting more than it would have anyway (T = 12 on 8 GPUs: 0,1,1,2,2,2,2,2
    instead of 1,1,1,1,2,2,2,2 — the frame ends with the ranks that hold 2 samples either way, and rank 0's ORB work no longer competes with a
    sample for its GPU).  Never when that would lengthen the heaviest shard (T = 48 on 8: 6 each stays; T = 12 on 4: 3 each stays)."""
    return world >= 4 and -(-T // (world - 1)) == -(-T // world)


def shard_samples(T, world, rank):
    """Contiguous shard of the T samples: returns (sample0, n_local).  When the samples do not divide evenly the LAST ranks take one
    extra sample, so rank 0 (ORB + host side of the frame) gets the lighter share or — orb_rank_is_free — none.  n_local may be 0."""
    if world > 1 and orb_rank_is_free(T, world):
        if rank == 0:
            return 0, 0
        world, rank = world - 1, rank - 1
    base, extra = divmod(T, world)
    first_heavy = world - extra
    n_local = base + (1 if rank >= first_heavy else 0)
    sample0 = rank * base + max(0, rank - first_heavy)
    return sample0, n_local


def max_shard(T, world):
    return max(shard_samples(T, world, r)[1] for r in range(world))


def all_reduce_prob_sum(prob_sum):
    """In-place SUM over ranks of the probability-sum tensor (no-op for a single process)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(prob_sum, op=dist.ReduceOp.SUM)
    return prob_sum


# ---- the sample-invariant prefix in row bands over the ranks (the product's plan is sivo_segnet_prefix_bands in
# sivo_amd/csrc/segnet.cpp — these two functions restate it for the host-side tests and for documentation)
def band_rows(h_out, world):
    """Rows of the prefix output (the pooling in front of the first test-time Dropout) per rank: [world + 1] boundaries.  The LAST
    h_out % world ranks take one row more — rank 0, which also runs ORB and the host side of the frame, never the larger share."""
    base, extra = divmod(h_out, world)
    return [r * base + max(0, r - (world - extra)) for r in range(world + 1)]


def band_input_rows(prefix_layers, H, y0, y1):
    """Image rows [lo, hi) a rank needs for the rows [y0, y1) of the prefix output.  prefix_layers: the parsed layers of the prefix in
    order (dicts with "type" and, for Convolution / Pooling, "kernel_size").  A 2x2 pooling doubles the range, a k x k convolution
    widens it by k // 2 on both sides (clipped to the layer's height); the result is aligned to 2^poolings rows so that every pooling
    window of the band is a pooling window of the frame."""
    pools = sum(1 for L in prefix_layers if L["type"] == "Pooling")
    heights = []
    h = H
    for L in prefix_layers:
        heights.append(h)
        if L["type"] == "Pooling":
            h //= 2
    lo, hi = y0, y1
    for L, h_in in zip(reversed(prefix_layers), reversed(heights)):
        if L["type"] == "Pooling":
            lo, hi = 2 * lo, 2 * hi
        elif L["type"] == "Convolution":
            lo, hi = lo - L["kernel_size"] // 2, hi + L["kernel_size"] // 2
        lo, hi = max(lo, 0), min(hi, h_in)
    a = 1 << pools
    return lo // a * a, min(H, (hi + a - 1) // a * a)
