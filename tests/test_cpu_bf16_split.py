"""The arithmetic the shipping GEMM kernels use (csrc/tile16.h), restated in torch on the CPU: split both
operands into three bf16 pieces with exact fp32 residuals, accumulate the six leading products in fp32.  The result must
be an fp32-accurate GEMM: no worse than the plain fp32 GEMM against fp64, and the split must be exact."""
import torch


def _bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _split3(x):
    h1 = _bf16(x)
    r1 = x - h1
    h2 = _bf16(r1)
    r2 = r1 - h2
    h3 = _bf16(r2)
    return h1, h2, h3, r2 - h3


def test_three_piece_split_is_exact_to_2_pow_minus_24():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1 << 16, generator=g) * torch.logspace(-6, 6, 1 << 16)
    h1, h2, h3, rest = _split3(x)
    # residual subtractions are exact in fp32, so the three pieces plus the final remainder reproduce x bit for bit
    assert torch.equal(((h1.double() + h2.double()) + h3.double()) + rest.double(), x.double())
    assert (rest.abs() <= x.abs() * 2.0 ** -24).all()
    for h in (h1, h2, h3):
        assert torch.equal(_bf16(h), h)            # every piece is a bf16 value


def test_six_product_gemm_is_fp32_accurate():
    g = torch.Generator().manual_seed(1)
    A = torch.randn(2048, 256, generator=g)
    W = torch.randn(256, 256, generator=g) / 16
    ref = A.double() @ W.double()
    a = _split3(A)[:3]
    w = _split3(W)[:3]
    acc = torch.zeros(2048, 256)
    for i, j in ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)):      # the order mfma6() issues them in
        acc = acc + a[i] @ w[j]
    err_split = ((acc.double() - ref).norm() / ref.norm()).item()
    err_fp32 = (((A @ W).double() - ref).norm() / ref.norm()).item()
    print(f"six-product split {err_split:.2e}  plain fp32 GEMM {err_fp32:.2e}")
    assert err_split < 3e-7
    assert err_split < 1.5 * err_fp32
    # two pieces / three products are NOT enough: this is why the kernels use three pieces
    acc2 = a[1] @ w[0] + a[0] @ w[1] + a[0] @ w[0]
    assert ((acc2.double() - ref).norm() / ref.norm()).item() > 1e-6
