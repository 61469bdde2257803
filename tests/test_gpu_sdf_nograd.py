import pytest
import torch

from tests.util import flat_from_oracle, randomized_models

pytestmark = pytest.mark.gpu


def _run_nograd(hiplib, flat, pts):
    from dynhor_amd import _lib
    packed = torch.empty(hiplib.dh_packed_floats(), device=pts.device)
    _lib.check(hiplib.dh_pack_weights(_lib.ptr(flat), _lib.ptr(packed), _lib.stream()))
    out = torch.full((pts.shape[0],), float("nan"), device=pts.device)
    _lib.check(hiplib.dh_sdf_nograd(_lib.ptr(packed), _lib.ptr(pts), pts.shape[0], _lib.ptr(out), _lib.stream()))
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("npts", [1, 127, 128, 129, 1000, 32768 + 5, 100003])
@pytest.mark.parametrize("jitter", [0.0, 0.05])
def test_sdf_nograd_matches_oracle(hiplib, npts, jitter):
    dev = torch.device("cuda:0")
    sdf, col, var = randomized_models(seed=11, device=dev, jitter=jitter)
    flat = flat_from_oracle(sdf, var, col)
    g = torch.Generator(device="cpu").manual_seed(npts)
    pts = ((torch.rand(npts, 3, generator=g) * 2 - 1) * 1.2).to(dev)
    out = _run_nograd(hiplib, flat, pts)
    with torch.no_grad():
        ref32 = sdf.sdf(pts).reshape(-1)
        ref64 = sdf.double().sdf(pts.double()).reshape(-1)
    e_hip = (out.double() - ref64).abs().max().item()
    e_t32 = (ref32.double() - ref64).abs().max().item()
    print(f"npts={npts} jitter={jitter}: |hip-f64|={e_hip:.3e} |torch32-f64|={e_t32:.3e}")
    assert torch.isfinite(out).all()
    # tolerance: fp32 parity, stated: 2e-5 absolute on an O(1) sdf (and no worse than 10x torch-fp32's own error)
    assert e_hip < 2e-5 or e_hip < 10 * e_t32


def test_empty_input_is_ok(hiplib):
    from dynhor_amd import _lib
    dev = torch.device("cuda:0")
    packed = torch.zeros(hiplib.dh_packed_floats(), device=dev)
    pts = torch.zeros(1, 3, device=dev)
    out = torch.zeros(1, device=dev)
    assert hiplib.dh_sdf_nograd(_lib.ptr(packed), _lib.ptr(pts), 0, _lib.ptr(out), _lib.stream()) == 0
    assert hiplib.dh_sdf_nograd(_lib.ptr(packed), _lib.ptr(pts), -1, _lib.ptr(out), _lib.stream()) == -1
