"""GPU: the tile-PAIR forms of the two-piece fp16 chains (csrc/pair16h.h, csrc/chain_pair.hip; round 6) against their tile-resident
twins (csrc/kernels_mlp_h.hip) -- BIT for bit.

A pair kernel holds a layer's weight slice in registers across two 64-point tiles and deals one tile's epilogue under the other tile's
MFMAs; the arithmetic (scales, piece planes, MFMA order inside a k-chunk, epilogue expressions) is the tile form's, so every output,
every saved tile and every scale word must be IDENTICAL.  The tile form is the one every oracle parity test of rounds 4-5 ran on
(tests/test_gpu_mlp_forward.py ... at sizes below the pair threshold still do); at the bench's size the launcher picks the pair form
and tests/test_gpu_fullsize_and_runner.py / test_gpu_bench_config.py compare THAT with the oracle.

Sizes: an even number of tiles, an odd number with a ragged last tile (the pair's second tile replays the first with its outputs off),
more tiles than workgroups (several pairs per workgroup: the weight registers wrap from the last layer to the next pair's first), and
one size above the automatic threshold (the default form there must be the pair form: checked through the kernel's own results being
equal to the forced pair form's and through include/dynhor_hip.h's rule)."""
import pytest
import torch

from tests.util import flat_from_oracle, randomized_models

pytestmark = pytest.mark.gpu
F16, TILE, PAIR = 2, 0x100, 0x200


def _setup(hiplib, npts, n_per_ray, seed):
    from dynhor_amd import _lib
    dev = torch.device("cuda:0")
    sdf, col, var = randomized_models(seed=seed, device=dev, jitter=0.05)
    flat = flat_from_oracle(sdf, var, col)
    packed = torch.empty(hiplib.dh_packed_floats(), device=dev)
    _lib.check(hiplib.dh_pack_weights(_lib.ptr(flat), _lib.ptr(packed), _lib.stream()))
    g = torch.Generator(device="cpu").manual_seed(seed * 7919 + npts)
    pts = ((torch.rand(npts, 3, generator=g) * 2 - 1) * 0.9).to(dev)
    nrays = (npts + n_per_ray - 1) // n_per_ray
    dirs = torch.nn.functional.normalize(torch.randn(nrays, 3, generator=g), dim=-1).to(dev)
    return packed, pts, dirs


def _forward(hiplib, packed, pts, dirs, n_per_ray, form, save, grad_form=0):
    """sdf forward -> input gradient (in grad_form) -> colour forward (in form) on a zeroed workspace: (colour, workspace, normals)."""
    from dynhor_amd import _lib
    p = _lib.ptr
    npts = pts.shape[0]
    _, fwd, _ = _lib.workspace_floats(npts)
    ws = torch.zeros(fwd, device=pts.device)
    sdf = torch.empty(npts, device=pts.device)
    normals = torch.empty(npts, 3, device=pts.device)
    color = torch.full((npts, 3), float("nan"), device=pts.device)
    st = _lib.stream()
    _lib.check(hiplib.dh_sdf_forward_ex(F16, p(packed), p(pts), npts, p(ws), p(sdf), st))
    _lib.check(hiplib.dh_sdf_gradient_ex(F16 | grad_form, p(packed), p(pts), npts, p(ws), p(normals), save, st))
    _lib.check(hiplib.dh_color_forward_ex(F16 | form, p(packed), p(pts), p(dirs), n_per_ray, p(normals), npts, p(ws), p(color), save, st))
    torch.cuda.synchronize()
    return color, ws, normals


@pytest.mark.parametrize("npts,n_per_ray", [(64 * 6, 64), (64 * 7 + 5, 4), (64 * 1301 + 17, 128), (64, 1)])
@pytest.mark.parametrize("save", [1, 0])
def test_colour_forward_pair_form_is_bit_identical_to_the_tile_form(hiplib, npts, n_per_ray, save):
    packed, pts, dirs = _setup(hiplib, npts, n_per_ray, seed=3)
    c_tile, ws_tile, _ = _forward(hiplib, packed, pts, dirs, n_per_ray, TILE, save, TILE)
    c_pair, ws_pair, _ = _forward(hiplib, packed, pts, dirs, n_per_ray, PAIR, save, TILE)
    assert torch.isfinite(c_tile).all() and c_tile.min().item() >= 0 and c_tile.max().item() <= 1 and c_tile.std().item() > 1e-3
    assert torch.equal(c_tile.view(torch.int32), c_pair.view(torch.int32)), \
        f"colours differ: max |d| {(c_tile - c_pair).abs().max().item():.3e}"
    d = (ws_tile.view(torch.int32) != ws_pair.view(torch.int32))
    assert not d.any(), f"{int(d.sum())} workspace words differ (first at float offset {int(d.nonzero()[0])})"
    # twice the pair form: the same bits again (no dependence on which workgroup or phase handled a tile)
    c_again, ws_again, _ = _forward(hiplib, packed, pts, dirs, n_per_ray, PAIR, save, TILE)
    assert torch.equal(c_again, c_pair) and torch.equal(ws_again, ws_pair)


@pytest.mark.parametrize("npts", [64 * 6, 64 * 7 + 5, 64 * 1301 + 17, 64])
@pytest.mark.parametrize("save", [1, 0, 2])
def test_input_gradient_pair_form_is_bit_identical_to_the_tile_form(hiplib, npts, save):
    """normals, the saved tiles a_0..a_7, the class maxima, the activation range word (and gesave with save = 2: pose refinement)."""
    packed, pts, dirs = _setup(hiplib, npts, 4, seed=6)
    _, ws_tile, n_tile = _forward(hiplib, packed, pts, dirs, 4, TILE, 1 if save else 0, TILE) if save != 2 else _grad_only(hiplib, packed, pts, TILE, 2)
    _, ws_pair, n_pair = _forward(hiplib, packed, pts, dirs, 4, TILE, 1 if save else 0, PAIR) if save != 2 else _grad_only(hiplib, packed, pts, PAIR, 2)
    assert torch.isfinite(n_tile).all() and n_tile.abs().max().item() > 1e-3
    assert torch.equal(n_tile.view(torch.int32), n_pair.view(torch.int32)), f"normals differ: max |d| {(n_tile - n_pair).abs().max().item():.3e}"
    d = (ws_tile.view(torch.int32) != ws_pair.view(torch.int32))
    assert not d.any(), f"{int(d.sum())} workspace words differ (first at float offset {int(d.nonzero()[0])})"


def _grad_only(hiplib, packed, pts, form, save):
    from dynhor_amd import _lib
    p = _lib.ptr
    npts = pts.shape[0]
    ws = torch.zeros(_lib.workspace_floats(npts)[2], device=pts.device)          # (gesave lies behind the backward's buffers)
    sdf = torch.empty(npts, device=pts.device)
    normals = torch.empty(npts, 3, device=pts.device)
    _lib.check(hiplib.dh_sdf_forward_ex(F16, p(packed), p(pts), npts, p(ws), p(sdf), _lib.stream()))
    _lib.check(hiplib.dh_sdf_gradient_ex(F16 | form, p(packed), p(pts), npts, p(ws), p(normals), save, _lib.stream()))
    torch.cuda.synchronize()
    return None, ws, normals


@pytest.mark.parametrize("npts", [64 * 6, 64 * 7 + 5, 64 * 1301 + 17, 64])
def test_colour_backward_pair_form_is_bit_identical_to_the_tile_form(hiplib, npts):
    """czbar_0..3, featbar, the normals' adjoint, the tile partial sums (bias gradients, colour lin4), tmax / absmax words."""
    from dynhor_amd import _lib
    p = _lib.ptr
    packed, pts, dirs = _setup(hiplib, npts, 4, seed=8)
    g = torch.Generator(device="cpu").manual_seed(npts)
    d_colors = (torch.randn(npts, 3, generator=g) * 10 ** (torch.rand(npts, 1, generator=g) * 4 - 4)).cuda()      # heavy-tailed adjoints
    out = {}
    for form in (TILE, PAIR):
        _, fwd, total = _lib.workspace_floats(npts)
        ws = torch.zeros(total, device="cuda:0")
        sdf = torch.empty(npts, device="cuda:0"); normals = torch.empty(npts, 3, device="cuda:0"); color = torch.empty(npts, 3, device="cuda:0")
        st = _lib.stream()
        _lib.check(hiplib.dh_sdf_forward_ex(F16, p(packed), p(pts), npts, p(ws), p(sdf), st))
        _lib.check(hiplib.dh_sdf_gradient_ex(F16 | TILE, p(packed), p(pts), npts, p(ws), p(normals), 1, st))
        _lib.check(hiplib.dh_color_forward_ex(F16 | TILE, p(packed), p(pts), p(dirs), 4, p(normals), npts, p(ws), p(color), 1, st))
        d_normals = torch.full((npts, 3), 0.25, device="cuda:0")           # the stage ACCUMULATES into it
        _lib.check(hiplib.dh_color_backward_ex(F16 | form, p(packed), p(color), p(d_colors), npts, p(ws), p(d_normals), st))
        torch.cuda.synchronize()
        out[form] = (ws, d_normals)
    assert torch.isfinite(out[TILE][1]).all() and (out[TILE][1] - 0.25).abs().max().item() > 0
    assert torch.equal(out[TILE][1].view(torch.int32), out[PAIR][1].view(torch.int32)), \
        f"d_normals differ: max |d| {(out[TILE][1] - out[PAIR][1]).abs().max().item():.3e}"
    d = out[TILE][0].view(torch.int32) != out[PAIR][0].view(torch.int32)
    assert not d.any(), f"{int(d.sum())} workspace words differ (first at float offset {int(d.nonzero()[0])})"


def test_the_default_form_follows_the_size_rule_and_the_flags_are_checked(hiplib):
    from dynhor_amd import _lib
    p = _lib.ptr
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    big = 64 * 2 * cus + 64
    packed, pts, dirs = _setup(hiplib, big, 128, seed=4)
    c_auto, ws_auto, n_auto = _forward(hiplib, packed, pts, dirs, 128, 0, 1, 0)
    c_pair, ws_pair, n_pair = _forward(hiplib, packed, pts, dirs, 128, PAIR, 1, TILE)
    assert torch.equal(n_auto, n_pair)
    assert torch.equal(c_auto, c_pair) and torch.equal(ws_auto, ws_pair)
    # flags with another arithmetic, two flags at once, a flag on a stage without a pair form: refused
    npts = 128
    ws = torch.zeros(_lib.workspace_floats(npts)[1], device="cuda:0")
    out = torch.empty(npts, 3, device="cuda:0")
    nrm = torch.zeros(npts, 3, device="cuda:0")
    args = (p(packed), p(pts), p(dirs), 128, p(nrm), npts, p(ws), p(out), 0, _lib.stream())
    assert hiplib.dh_color_forward_ex(0 | PAIR, *args) == -1
    assert hiplib.dh_color_forward_ex(F16 | PAIR | TILE, *args) == -1
    assert hiplib.dh_color_forward_ex(F16 | 0x400, *args) == -1
    assert hiplib.dh_sdf_forward_ex(F16 | PAIR, p(packed), p(pts), npts, p(ws), p(out), _lib.stream()) == -1
