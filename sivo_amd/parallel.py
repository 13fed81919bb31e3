"""Monte-Carlo sample sharding over ranks (one process per GPU, torch.distributed; the "nccl"
backend is RCCL on ROCm).  The only axis of the per-frame path that shards is T: each MC sample is
an independent forward of the same image (reference bayesian_segnet.cpp:174-177) and the samples are
coupled only by the mean over the batch axis (:294), a linear reduction — one all-reduce(SUM) of the
classes x H x W fp32 probability sums per frame.  Dropout masks are keyed by the GLOBAL sample index,
so the result does not depend on the number of ranks."""


def shard_samples(T, world, rank):
    """Contiguous shard of the T samples: returns (sample0, n_local).  When T is not a multiple of the
    world size the LAST T % world ranks take one extra sample: rank 0 also runs the ORB extractors, the
    stereo matching and the host side of the frame, so it gets the lighter share (T = 12 on 8 GPUs:
    1,1,1,1,2,2,2,2).  n_local may be 0 when world > T."""
    base, extra = divmod(T, world)
    first_heavy = world - extra
    n_local = base + (1 if rank >= first_heavy else 0)
    sample0 = rank * base + max(0, rank - first_heavy)
    return sample0, n_local


def max_shard(T, world):
    base, extra = divmod(T, world)
    return base + (1 if extra else 0)


def all_reduce_prob_sum(prob_sum):
    """In-place SUM over ranks of the probability-sum tensor (no-op for a single process)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(prob_sum, op=dist.ReduceOp.SUM)
    return prob_sum
