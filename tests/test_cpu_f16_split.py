"""The arithmetic of the shipping GEMM kernels (round 4; csrc/tile16h.h), restated in torch on the CPU: both operands scaled by
powers of two, split into TWO fp16 pieces (hi = fp16(X), lo = fp16(X - hi), the unscaled residual, fp16 subnormals included),
three products hi*lo + lo*hi + hi*hi accumulated in fp32.  Pinned here (VERDICT r3 next #1a): with the scales the kernels choose
the result is as accurate as a plain fp32 GEMM on activation-, weight- and adjoint-like operands from 1e-9 to 1e3; without a
scale small operands are lost -- which is why every operand class has one.  The hardware numbers (v_mfma_f32_32x32x16_f16 does
not flush fp16 subnormals; 1.9e-7 against 2.3e-7 for the exact fp32 MFMA) are in profiles/r04_ab_f16x2_chain_micro.json."""
import math
import struct

import pytest
import torch

H2_WT, H2_AT = 3, 8        # csrc/tile16h.h: scaled maximum of the weights in [8, 16), of dynamically scaled operands in [256, 512)


def _f16(x):
    return x.to(torch.float16).to(torch.float32)


def pow2_scale(max_abs: float, target: int) -> float:
    """tile16h.h pow2_scale_bits: S = 2^k with S * max in [2^target, 2^(target+1)), k clamped to +-60, from the fp32 bits."""
    bits = struct.unpack("<I", struct.pack("<f", max_abs))[0]
    e = (bits >> 23) & 0xFF
    k = max(-60, min(60, target - (e - 127)))
    return 2.0 ** k


def split2(x):
    hi = _f16(x)
    lo = _f16(x - hi)
    return hi, lo


def gemm_f16x2(A, W, sa, sw):
    ah, al = split2(A * sa)
    wh, wl = split2(W * sw)
    acc = ah @ wl                  # the order the kernels issue them in: smallest terms first
    acc = acc + al @ wh
    acc = acc + ah @ wh
    return acc / (sa * sw)


def _rel(x, ref):
    return ((x.double() - ref).norm() / ref.norm()).item()


def test_scale_puts_the_maximum_where_the_kernels_expect_it():
    for m in (1e-9, 3.3e-7, 0.11, 1.0, 1.5, 255.9, 1e3, 4.2e6):
        s = pow2_scale(m, H2_AT)
        assert 256.0 <= s * m < 512.0 and math.log2(s) == int(math.log2(s))
        s = pow2_scale(m, H2_WT)
        assert 8.0 <= s * m < 16.0
    assert pow2_scale(0.0, H2_AT) == 2.0 ** 60 and pow2_scale(float("inf"), H2_AT) == 2.0 ** -60      # clamped: finite


def test_two_piece_split_represents_to_2_pow_minus_22_relative_or_2_pow_minus_25_absolute():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1 << 16, generator=g) * torch.logspace(-8, 2.5, 1 << 16)
    hi, lo = split2(x)
    err = (x.double() - (hi.double() + lo.double())).abs()
    assert torch.equal(_f16(hi), hi) and torch.equal(_f16(lo), lo)          # both pieces are fp16 values
    # hi is x to 11 bits (error <= 2^-12 ulp-wise: half an fp16 ulp), lo the residual to 11 bits again: <= 2^-22 |x| in the worst
    # case and 2^-24.6 |x| rms -- or 2^-25 absolute where lo is an fp16 subnormal (spacing 2^-24)
    assert (err <= torch.maximum(x.abs().double() * 2.0 ** -22, torch.tensor(2.0 ** -25, dtype=torch.float64))).all()
    big = x.abs() >= 1.0
    rms = ((err[big] / x.abs()[big].double()) ** 2).mean().sqrt().item()
    assert rms < 2.0 ** -24, rms
    # fp16's range is the constraint the scales exist for
    assert torch.isinf(_f16(torch.tensor([7e4]))).all()


@pytest.mark.parametrize("name,amag,spread,wmag", [
    ("softplus-like activations", 1.0, 0.0, 0.088), ("small activations", 1e-3, 0.0, 0.088), ("adjoints 1e-6", 1e-6, 0.0, 0.088),
    ("adjoints 1e-9", 1e-9, 0.0, 0.3), ("heavy-tailed adjoints", 1e-5, 3.0, 0.088), ("values 1e3", 1e3, 0.0, 0.02),
    ("tiny weights", 1.0, 0.0, 1e-4)])
def test_three_product_gemm_is_fp32_accurate_with_the_kernels_scales(name, amag, spread, wmag):
    g = torch.Generator().manual_seed(1)
    M, K, N = 2048, 256, 256
    z = torch.randn(M, K, generator=g) * 0.3
    A = torch.nn.functional.softplus(z, beta=100) if "activ" in name else torch.randn(M, K, generator=g)
    A = A * amag * torch.exp(torch.randn(M, 1, generator=g) * spread)
    W = torch.randn(K, N, generator=g) * wmag
    ref = A.double() @ W.double()
    sa, sw = pow2_scale(A.abs().max().item(), H2_AT), pow2_scale(W.abs().max().item(), H2_WT)
    if spread > 0:      # heavy-tailed operands are scaled per 64-row tile (the chains' LDS images, the weight-gradient kernel's X)
        out = torch.cat([gemm_f16x2(A[t:t + 64], W, pow2_scale(A[t:t + 64].abs().max().item(), H2_AT), sw) for t in range(0, M, 64)])
    else:
        out = gemm_f16x2(A, W, sa, sw)
    e_split = _rel(out, ref)
    # (the plain fp32 GEMM in the same 64-row pieces where the split ran in pieces: torch's blocking, hence its rounding, depends on M)
    e_fp32 = _rel(torch.cat([A[t:t + 64] @ W for t in range(0, M, 64)]) if spread > 0 else A @ W, ref)
    print(f"{name}: two-piece fp16 / three products {e_split:.3e}   plain fp32 GEMM {e_fp32:.3e}   (S_a = 2^{int(math.log2(sa))}, S_w = 2^{int(math.log2(sw))})")
    # three separately rounded fp32 partial GEMMs here; the matrix core keeps one accumulator (measured BELOW the fp32 MFMA).
    # Homogeneous operands: within 3 % of the plain fp32 GEMM.  Heavy-tailed rows under a per-tile scale behave like block floating
    # point: the rows that dominate the tile are fp32-accurate (the two-piece representation is 2^-24.6 rms against fp32's
    # 2^-25.8: +10 % here), a row far below its tile's maximum keeps an ABSOLUTE error of 2^-25 of that maximum per element --
    # its own contribution to anything summed over rows (a weight gradient) is as small as it is
    assert e_split <= (1.15 if spread > 0 else 1.03) * e_fp32
    assert e_split < 3.5e-7
    if spread > 0:
        err_rows = (out.double() - ref).norm(dim=1)
        tile_top = ref.norm(dim=1).view(-1, 64).max(dim=1).values.repeat_interleave(64)
        assert (err_rows <= 4e-7 * tile_top).all()


def test_the_static_scale_of_O1_operands_is_enough():
    """Softplus / ReLU activations and embeddings are carried at the constant H2_XS = 16 (no maximum is tracked for them)."""
    g = torch.Generator().manual_seed(2)
    A = torch.nn.functional.softplus(torch.randn(2048, 256, generator=g) * 0.3, beta=100)
    W = torch.randn(256, 256, generator=g) * 0.088
    ref = A.double() @ W.double()
    e = _rel(gemm_f16x2(A, W, 16.0, pow2_scale(W.abs().max().item(), H2_WT)), ref)
    assert e <= 1.03 * _rel(A @ W, ref)


def test_unscaled_small_operands_are_lost_and_a_launch_wide_scale_fails_on_heavy_tails():
    g = torch.Generator().manual_seed(3)
    W = torch.randn(256, 256, generator=g) * 0.088
    sw = pow2_scale(W.abs().max().item(), H2_WT)
    A = torch.randn(1024, 256, generator=g) * 1e-6
    ref = A.double() @ W.double()
    assert _rel(gemm_f16x2(A, W, 1.0, sw), ref) > 1e-3                # fp16 cannot hold 1e-6 to more than a few bits
    # one outlier row sets a launch-wide scale: the typical rows land in fp16's subnormals.  Per-tile scales (the weight-gradient
    # kernel's scheme for adjoints and tangents: the heavy operand by its tile's maximum, the other operand divided by the same
    # power of two) keep every tile accurate.
    A = torch.randn(1024, 256, generator=g) * 1e-7
    A[5] *= 1e7
    ref = A.double() @ W.double()
    typical = torch.arange(1024) != 5
    e_global = _rel(gemm_f16x2(A, W, pow2_scale(A.abs().max().item(), H2_AT), sw)[typical], ref[typical])
    out = torch.empty(1024, 256)
    for t in range(0, 1024, 64):
        st = pow2_scale(A[t:t + 64].abs().max().item(), H2_AT)
        out[t:t + 64] = gemm_f16x2(A[t:t + 64], W, st, sw)
    e_tile = _rel(out[typical][64:], ref[typical][64:])               # the tiles without the outlier
    print(f"typical rows: launch-wide scale {e_global:.2e}, per-tile scales {e_tile:.2e}")
    assert e_global > 1e-4 and e_tile < 3.5e-7
