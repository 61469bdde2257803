"""How accurate is an fp32 GEMM emulated with bf16 pieces and fp32 accumulation (the scheme of bf16x3_micro.hip)?
CPU-only, numpy/torch.  [4096 x 256] @ [256 x 256], errors relative to fp64."""
import torch

torch.manual_seed(0)


def bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def split(x, n):
    parts, r = [], x.clone()
    for _ in range(n):
        h = bf16(r)
        parts.append(h)
        r = r - h
    return parts


M, K, N = 4096, 256, 256
A = torch.randn(M, K)
W = torch.randn(K, N) / 16
ref = A.double() @ W.double()


def err(x):
    return ((x.double() - ref).abs().max() / ref.abs().max()).item(), ((x.double() - ref).norm() / ref.norm()).item()


print("fp32 GEMM                       max %.2e  rel %.2e" % err(A @ W))
for na, terms in [(1, [(0, 0)]), (2, [(0, 0), (0, 1), (1, 0)]), (3, [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]),
                  (3, [(i, j) for i in range(3) for j in range(3)])]:
    a, b = split(A, na), split(W, na)
    acc = torch.zeros(M, N)
    for i, j in sorted(terms, key=lambda t: -(t[0] + t[1])):     # small terms first, fp32 accumulation
        acc = acc + a[i] @ b[j]
    print("bf16 x%d pieces, %d MFMA products  max %.2e  rel %.2e" % ((na, len(terms)) + err(acc)))
