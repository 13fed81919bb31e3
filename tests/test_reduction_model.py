"""Model of the halving butterfly of sivo_amd/csrc/ba_solve.hip (block_sum_h / PoseReducer::sum28): a numpy re-enactment of the
lane exchanges, so that the slot a lane ends up with (halving_slot) and the totals are pinned on the CPU.  The device kernels are
checked against the oracle under -m gpu (tests/test_gpu_ba_solve.py); this file pins the INDEX ARITHMETIC the kernels share."""
import numpy as np
import pytest


def halving_slot(lane, P):
    """halving_slot<P> of ba_solve.hip: bit b of the lane index selects the half kept at step b, most significant slot bit first."""
    top = {64: 5, 32: 4, 16: 3}[P]
    s = 0
    b = 0
    while (1 << b) < P:
        s |= ((lane >> b) & 1) << (top - b)
        b += 1
    return s


def wave_reduce(values, K, P):
    """values: (64 lanes, K).  Returns (per-lane final value, per-lane slot) after the first step on the K values (slots >= K are
    zero and never materialised), the remaining halving steps, and the plain butterfly over the lanes that share a slot."""
    v = np.zeros((64, P))
    v[:, :K] = values
    lanes = np.arange(64)
    n = P
    step = 0
    while n > 1:
        M = 1 << step
        half = n // 2
        bit = (lanes & M) != 0
        keep = np.where(bit[:, None], v[:, half:n], v[:, :half])
        send = np.where(bit[:, None], v[:, :half], v[:, half:n])
        recv = send[lanes ^ M]                       # lane_xor<M>: the partner's `send`
        v = np.zeros((64, P))
        v[:, :half] = keep + recv
        n = half
        step += 1
    t = v[:, 0].copy()
    while (1 << step) < 64:                          # P < 64: the remaining butterfly adds the lanes that hold the same slot
        t = t + t[lanes ^ (1 << step)]
        step += 1
    return t, np.array([halving_slot(int(l), P) for l in lanes])


@pytest.mark.parametrize("K,P", [(28, 32), (27, 32), (42, 64), (64, 64), (16, 16), (9, 16)])
def test_every_slot_total_lands_in_the_lanes_halving_slot_names(K, P):
    rng = np.random.default_rng(K * 100 + P)
    x = rng.integers(-1000, 1000, (64, K)).astype(np.float64)       # integers: sums are exact whatever the order
    t, slot = wave_reduce(x, K, P)
    want = x.sum(axis=0)
    for lane in range(64):
        s = slot[lane]
        assert t[lane] == (want[s] if s < K else 0.0), (lane, s)
    # the kernels publish lanes 0 .. P - 1: together they hold every slot exactly once
    assert sorted(slot[:P].tolist()) == list(range(P))
    # ... and sum28's explicit formula (PoseReducer, P = 32) is the same map
    if P == 32:
        for lane in range(64):
            assert slot[lane] == ((lane & 1) << 4) | ((lane & 2) << 2) | (lane & 4) | ((lane & 8) >> 2) | ((lane & 16) >> 4)
