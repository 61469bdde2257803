import pytest
import torch

from tests.util import flat_from_oracle, randomized_models

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nrays,n_per_ray", [(3, 64), (32, 128), (65, 100)])
@pytest.mark.parametrize("jitter", [0.0, 0.05])
def test_mlp_forward_matches_oracle(hiplib, nrays, n_per_ray, jitter):
    from dynhor_amd import _lib
    dev = torch.device("cuda:0")
    sdf, col, var = randomized_models(seed=5, device=dev, jitter=jitter)
    flat = flat_from_oracle(sdf, var, col)
    packed = torch.empty(hiplib.dh_packed_floats(), device=dev)
    _lib.check(hiplib.dh_pack_weights(_lib.ptr(flat), _lib.ptr(packed), _lib.stream()))
    g = torch.Generator(device="cpu").manual_seed(nrays * 1000 + n_per_ray)
    npts = nrays * n_per_ray
    pts = ((torch.rand(npts, 3, generator=g) * 2 - 1) * 1.1).to(dev)
    dirs = torch.nn.functional.normalize(torch.randn(nrays, 3, generator=g), dim=-1).to(dev)
    _, _, total = _lib.workspace_floats(npts)
    ws = torch.empty(total, device=dev)
    o_sdf = torch.full((npts,), float("nan"), device=dev)
    o_n = torch.full((npts, 3), float("nan"), device=dev)
    o_c = torch.full((npts, 3), float("nan"), device=dev)
    _lib.check(hiplib.dh_mlp_forward(_lib.ptr(packed), _lib.ptr(pts), _lib.ptr(dirs), n_per_ray, npts, _lib.ptr(ws),
                                     _lib.ptr(o_sdf), _lib.ptr(o_n), _lib.ptr(o_c), _lib.stream()))
    torch.cuda.synchronize()

    def oracle(dtype):
        s, c = sdf.to(dtype), col.to(dtype)
        p = pts.to(dtype).clone()
        out = s(p)
        grad = s.gradient(p).squeeze(1)
        d = dirs.to(dtype).repeat_interleave(n_per_ray, dim=0)
        cc = c(p, grad, d, out[:, 1:])
        return out[:, 0].detach(), grad.detach(), cc.detach()

    r32 = oracle(torch.float32)
    r64 = oracle(torch.float64)
    sdf.float(); col.float()
    for name, got, a32, a64, tol in (("sdf", o_sdf, r32[0], r64[0], 2e-5), ("normal", o_n, r32[1], r64[1], 2e-4),
                                     ("color", o_c, r32[2], r64[2], 2e-5)):
        e_hip = (got.double() - a64).abs().max().item()
        e_t32 = (a32.double() - a64).abs().max().item()
        print(f"{name}: |hip-f64|={e_hip:.3e} |torch32-f64|={e_t32:.3e}")
        assert torch.isfinite(got).all(), name
        assert e_hip < tol or e_hip < 10 * e_t32, name
