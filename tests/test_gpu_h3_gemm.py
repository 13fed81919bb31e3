"""The f16x3 GEMM of the three-kernel F(4x4,3x3) path (sivo_amd/csrc/conv_wino4_h3.hip) alone, through the C ABI
(sivo_debug_h3_gemm), against an fp64 evaluation of the same 36 products  M_xi = U_xi^T V_xi.

fp32 operands are multiplied as fp16 hi + lo pairs (three v_mfma_f32_32x32x16_f16 products, fp32 accumulate).  Error
model: each operand is represented to 2^-22 relative and the lo x lo product (2^-22) is dropped, i.e. at most 3 * 2^-22 per
product, plus the fp32 roundings of the accumulator; asserted: |err| <= 2^-20 * sum |u| |v| per element (measured worst
case over all shapes: 1.0 * 2^-21; an fp32 FMA chain of C = 64 terms is allowed 64 * 2^-24 = 2^-18).  Operands are random and asymmetric (a
transposed operand or a swapped output index cannot pass); the shapes exercise the five workgroup tiles (all couts in one
item: 128 x 512 and 64 x 512 tiles x couts; 256 x 256 and 128 x 256; 256 x 128 for 128 couts), ragged tile counts (P not a
multiple of 32 / 128 / 256), several items per workgroup, item boundaries inside the software pipeline, and the two largest
GEMMs of SegNet-Standard at T = 12 (512 -> 512 at 44 x 128: one lane and one of three lanes)."""
import numpy as np
import pytest

from sivo_amd.segnet import h3_gemm

pytestmark = pytest.mark.gpu

# (C, Kp, P): tile chosen by the launcher (h3_tile) — Kp % 512 == 0: 128 x 512 when 36 * ceil(P / 128) >= 768, else 64 x 512;
# Kp % 256 == 0: 256 x 256 when 36 * ceil(P / 256) * (Kp / 256) >= 768, else 128 x 256; Kp = 128: 256 x 128
SHAPES = [(32, 256, 200), (64, 256, 1152), (128, 512, 900), (64, 512, 2700), (96, 256, 5700), (64, 128, 300), (128, 128, 2100),
          (512, 512, 1152), (256, 512, 1408), (512, 512, 4224)]


@pytest.mark.parametrize("C,Kp,P", SHAPES, ids=[f"C{c}-K{k}-P{p}" for c, k, p in SHAPES])
def test_h3_gemm_against_fp64(C, Kp, P):
    rng = np.random.default_rng(C * 7 + Kp + P)
    Pp = (P + 127) // 128 * 128
    # Winograd-domain magnitudes: V spread over three decades (relu-like, mostly positive), U small and signed
    V = (rng.standard_normal((36, C, Pp)) * np.exp(rng.uniform(-3, 3, (36, C, 1)))).astype(np.float32)
    V[:, :, P:] = 0
    U = (rng.standard_normal((36, C, Kp)) * 0.02 * np.exp(rng.uniform(-2, 2, (36, 1, Kp)))).astype(np.float32)
    M, _ = h3_gemm(V, U, P)
    worst = 0.0
    for xi in (range(36) if C * Kp * P < 3e8 else (0, 7, 14, 21, 28, 35)):          # (the fp64 reference of the largest shapes: six positions)
        ref = U[xi].astype(np.float64).T @ V[xi].astype(np.float64)                      # (Kp, Pp)
        bound = np.abs(U[xi]).astype(np.float64).T @ np.abs(V[xi]).astype(np.float64)
        err = np.abs(M[xi][:, :P] - ref[:, :P])
        assert np.isfinite(M[xi][:, :P]).all(), xi
        rel = float((err / np.maximum(bound[:, :P], 1e-30)).max())
        worst = max(worst, rel)
        assert rel < 2.0 ** -20, (xi, rel)
        assert np.abs(ref).max() > 1e-3
    print(f"[h3 gemm C={C} Kp={Kp} P={P}] worst |err| / sum|u||v| = {worst:.2e} (bound 2^-20 = {2.0 ** -20:.2e})")


def test_h3_gemm_small_values_keep_their_precision():
    """Values far below the layer's maximum lose the lo plane gradually (fp16 subnormals): the ABSOLUTE error stays at
    2^-25 of the scaled maximum, which is what the scale choice promises."""
    rng = np.random.default_rng(5)
    C, Kp, P = 64, 256, 256
    V = rng.standard_normal((36, C, P)).astype(np.float32)
    V[:, :, 128:] *= 1e-5                       # the second half of the tiles: 2^-17 of the maximum
    U = (rng.standard_normal((36, C, Kp)) * 0.05).astype(np.float32)
    M, _ = h3_gemm(V, U, P)
    vmax = float(np.abs(V).max())
    for xi in (0, 17, 35):
        ref = U[xi].astype(np.float64).T @ V[xi].astype(np.float64)
        err = np.abs(M[xi] - ref)
        assert err[:, 128:].max() < 2.0 ** -20 * vmax * np.abs(U[xi]).sum(axis=0).max() * 2.0 ** -8


@pytest.mark.parametrize("C,Kp,P", [(64, 256, 2816), (128, 128, 2100), (512, 512, 4224), (256, 512, 1100)], ids=["256x256", "256x128", "conv4_2", "128x256"])
def test_the_three_stage_loop_forms_are_bit_identical(C, Kp, P, monkeypatch):
    """FORM 3 (the product's: the fragment reads as a rotating pipeline, the stage barrier one term early), FORM 2 (8-byte V' loads, two tiles
    per lane), FORM 1 (4-byte loads, memory side inside the multiply phase) and FORM 0 (the phased loop of rounds 3 - 5) issue the same
    MFMAs in the same order per accumulator: M word for word, also on ragged last items and on the 128-tile / 128-cout kernels.  FORM 0 - 2
    live in the diagnostic build."""
    from sivo_amd import _lib
    rng = np.random.default_rng(C + Kp + P)
    Pp = (P + 127) // 128 * 128
    V = (rng.standard_normal((36, C, Pp)) * np.exp(rng.uniform(-2, 2, (36, C, 1)))).astype(np.float32)
    V[:, :, P:] = 0
    U = (rng.standard_normal((36, C, Kp)) * 0.03).astype(np.float32)
    ref, _ = h3_gemm(V, U, P)                           # the product library: FORM 3
    for form in ("3", "2", "1", "0"):
        monkeypatch.setenv("SIVO_H3_FORM", form)
        with _lib.use("diag"):
            M, _ = h3_gemm(V, U, P)
        assert np.array_equal(M.view(np.uint32)[:, :, :P], ref.view(np.uint32)[:, :, :P]), f"FORM {form}"
