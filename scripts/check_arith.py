"""Development check: the three arithmetics (two-piece fp16, three-piece bf16, native fp32 MFMA) stage by stage through the C ABI
on the same weights and points -- outputs, every saved tile class and the flat gradient, each against the fp32-MFMA twin.
    python scripts/check_arith.py [nrays] [n_per_ray]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynhor_amd import _lib  # noqa: E402
from tests.util import flat_from_oracle, randomized_models  # noqa: E402


def main():
    nrays = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    n_per_ray = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    dev = torch.device("cuda:0")
    L = _lib.lib()
    sdf, col, var = randomized_models(seed=5, device=dev, jitter=0.05)
    flat = flat_from_oracle(sdf, var, col)
    packed = torch.empty(L.dh_packed_floats(), device=dev)
    _lib.check(L.dh_pack_weights(_lib.ptr(flat), _lib.ptr(packed), _lib.stream()))
    g = torch.Generator(device="cpu").manual_seed(1)
    npts = nrays * n_per_ray
    pts = ((torch.rand(npts, 3, generator=g) * 2 - 1) * 1.1).to(dev)
    dirs = torch.nn.functional.normalize(torch.randn(nrays, 3, generator=g), dim=-1).to(dev)
    # adjoints with the spread a render step produces: most points carry almost nothing
    w = torch.exp(torch.randn(npts, 1, generator=g) * 3.0).to(dev) * 1e-5
    d_sdf = (torch.randn(npts, generator=g).to(dev) * w[:, 0]).contiguous()
    d_normals0 = (torch.randn(npts, 3, generator=g).to(dev) * w).contiguous()
    d_colors = (torch.randn(npts, 3, generator=g).to(dev) * w).contiguous()
    infer, fwd, total = _lib.workspace_floats(npts)
    res = {}
    for name, ar in (("fp32", _lib.ARITH_FP32_MFMA), ("bf16x3", _lib.ARITH_SPLIT_BF16), ("f16x2", _lib.ARITH_SPLIT_F16)):
        ws = torch.zeros(total, device=dev)
        o_sdf = torch.full((npts,), float("nan"), device=dev)
        o_n = torch.full((npts, 3), float("nan"), device=dev)
        o_c = torch.full((npts, 3), float("nan"), device=dev)
        sd = torch.full((npts,), float("nan"), device=dev)
        _lib.check(L.dh_sdf_nograd_ex(ar, _lib.ptr(packed), _lib.ptr(pts), npts, _lib.ptr(sd), _lib.stream()))
        _lib.check(L.dh_mlp_forward_ex(ar, _lib.ptr(packed), _lib.ptr(pts), _lib.ptr(dirs), n_per_ray, npts, _lib.ptr(ws),
                                       _lib.ptr(o_sdf), _lib.ptr(o_n), _lib.ptr(o_c), _lib.stream()))
        grad = torch.zeros(flat.numel(), device=dev)
        d_normals = d_normals0.clone()
        _lib.check(L.dh_mlp_backward_ex(ar, _lib.ptr(packed), _lib.ptr(flat), _lib.ptr(pts), npts, _lib.ptr(ws), _lib.ptr(o_c),
                                        _lib.ptr(d_sdf), _lib.ptr(d_normals), _lib.ptr(d_colors), _lib.ptr(grad), _lib.stream()))
        torch.cuda.synchronize()
        res[name] = dict(nograd=sd, sdf=o_sdf, normals=o_n, color=o_c, d_normals=d_normals, grad=grad, ws=ws[64 * 64:fwd].clone(),
                         wsb=ws[fwd:total].clone())
    ref = res["fp32"]
    bad = False
    for name in ("bf16x3", "f16x2"):
        r = res[name]
        line = [name]
        for k in ("nograd", "sdf", "normals", "color", "d_normals"):
            line.append(f"{k} {float((r[k] - ref[k]).abs().max()):.2e}")
            bad |= not bool(torch.isfinite(r[k]).all())
        for k in ("grad", "ws", "wsb"):
            a, b = r[k].double(), ref[k].double()
            fin = torch.isfinite(a) & torch.isfinite(b)
            line.append(f"{k} rel {float((a - b)[fin].norm() / b[fin].norm()):.2e}")
        bad |= not bool(torch.isfinite(r["grad"]).all())
        print("  ".join(line))
    print("NON-FINITE VALUES" if bad else "finite")


if __name__ == "__main__":
    main()
