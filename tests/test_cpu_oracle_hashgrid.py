"""CPU: properties of the hash-grid oracle (oracle/hashgrid_oracle.py; parity unpinned, see its header)."""
import math

import torch

from oracle import hashgrid_oracle as H
from oracle import neus_oracle as O


def test_level_geometry():
    e = H.HashGridEncoding()
    assert e.resolutions[0] == 16 and e.resolutions[-1] == 2049 and len(e.resolutions) == 16
    assert all(b > a for a, b in zip(e.resolutions, e.resolutions[1:]))
    assert e.dense == [r ** 3 <= (1 << 19) for r in e.resolutions]
    assert all(s == (1 << 19) for s, d in zip(e.sizes, e.dense) if not d)
    assert e.n_entries == sum(e.sizes) and e.table.shape == (e.n_entries, 2)
    assert abs(e.per_level_scale - math.exp(math.log(2048 / 16) / 15)) < 1e-12


def test_encoding_interpolates_table_rows_at_grid_nodes_and_is_continuous():
    torch.manual_seed(0)
    e = H.HashGridEncoding().double()
    with torch.no_grad():
        e.table.copy_(torch.randn_like(e.table))
    l = 2                                                    # a dense level: node (i,j,k) <-> x01 = (node - 0.5) / scale
    node = torch.tensor([[5, 9, 3]])
    x = (node.double() - 0.5) / e.scales[l]
    out = e(x)[:, 2 * l:2 * l + 2]
    row = e.table[e.level_index(l, node[:, 0], node[:, 1], node[:, 2])]
    assert torch.allclose(out, row, atol=1e-9)
    a = torch.rand(64, 3, dtype=torch.float64) * 0.8 + 0.1
    d = torch.randn(64, 3, dtype=torch.float64) * 1e-7
    assert (e(a + d) - e(a)).abs().max().item() < 1e-2       # Lipschitz: finest level has 2048 cells across, |table| ~ 3


def test_sphere_init_and_fd_normals():
    sdf, col = H.build_models(seed=3)
    x = torch.nn.functional.normalize(torch.randn(200, 3), dim=-1) * torch.rand(200, 1)
    s = sdf.sdf(x).reshape(-1)
    r = x.norm(dim=-1)
    assert torch.corrcoef(torch.stack([s, r]))[0, 1].item() > 0.85, "sphere init (one hidden layer: coarse): sdf grows with |x|"
    assert sdf.sdf(torch.zeros(1, 3)).item() < 0 < sdf.sdf(torch.tensor([[0.95, 0.0, 0.0]])).item()
    g = sdf.gradient(x).squeeze(1)
    assert ((g * x).sum(-1)[r > 0.1] > 0).all(), "FD normals point outward at init"
    out = sdf(x)
    assert out.shape == (200, 14) and torch.equal(out[:, 0], out[:, 1]), "feature = all 13 outputs (incl. the sdf channel)"


def test_sh4_basis_is_orthonormal_on_the_sphere():
    torch.manual_seed(1)
    d = torch.nn.functional.normalize(torch.randn(200000, 3, dtype=torch.float64), dim=-1)
    Y = H.sh4(d)
    G = (Y.T @ Y) / d.shape[0] * 4 * math.pi
    assert (G - torch.eye(16, dtype=torch.float64)).abs().max().item() < 0.03


def test_plugs_into_the_neus_renderer_and_trains():
    sdf, col = H.build_models(seed=2)
    var = O.SingleVarianceNetwork(0.3)
    r = O.NeuSRenderer(None, sdf, var, col, 16, 16, 0, 4, 1.0)
    o = torch.tensor([[0.0, 0.0, -2.2]]).repeat(24, 1)
    d = torch.nn.functional.normalize(torch.randn(24, 3) * 0.1 + torch.tensor([0, 0, 1.0]), dim=-1)
    near, far = O.near_far_from_sphere(o, d)
    out = r.render(o, d, near, far, cos_anneal_ratio=0.5)
    assert out["color_fine"].shape == (24, 3) and torch.isfinite(out["color_fine"]).all()
    (out["color_fine"].sum() + out["gradient_error"]).backward()
    assert sdf.lin0.weight_v.grad.abs().sum().item() > 0 and col.lin2.weight_v.grad.abs().sum().item() > 0


def test_out_of_box_coordinates_wrap_as_uint32():
    """Samples just outside the box (the renderer's last mid-point overshoots the unit sphere) index the table with
    uint32 wrap-around, the convention the HIP kernels implement: checked against independent numpy uint32 arithmetic."""
    import numpy as np
    e = H.HashGridEncoding()
    g = torch.Generator().manual_seed(3)
    c = torch.randint(-40, 60, (500, 3), generator=g)
    for l in (0, 4, 5, 15):                                  # two dense, two hashed levels
        got = e.level_index(l, c[:, 0], c[:, 1], c[:, 2]).numpy() - e.offsets[l]
        u = c.numpy().astype(np.int64).astype(np.uint32)     # two's-complement wrap
        res = np.uint32(e.resolutions[l])
        with np.errstate(over="ignore"):
            if e.dense[l]:
                ref = (u[:, 0] + u[:, 1] * res + u[:, 2] * res * res) % np.uint32(e.sizes[l])
            else:
                ref = ((u[:, 0] * np.uint32(1)) ^ (u[:, 1] * np.uint32(2654435761)) ^ (u[:, 2] * np.uint32(805459861))) \
                      % np.uint32(e.T)
        assert (got == ref.astype(np.int64)).all(), l
        assert got.min() >= 0 and got.max() < e.sizes[l]
    # the encoding stays finite and continuous across the box face
    e = e.double()
    with torch.no_grad():
        e.table.copy_(torch.randn_like(e.table))
    x = torch.tensor([[1.0 - 1e-9, 0.3, 0.7], [1.0 + 1e-9, 0.3, 0.7], [-1e-9, 0.5, 0.5], [1e-9, 0.5, 0.5]], dtype=torch.float64)
    out = e(x)
    assert torch.isfinite(out).all()
    assert (out[0] - out[1]).abs().max() < 1e-4 and (out[2] - out[3]).abs().max() < 1e-4


def test_batched_gather_form_equals_the_level_loop_in_values_and_gradients():
    """Round 6: forward() fetches every corner row with one index_select (the loop form's backward cost 15 GPU-minutes per PSNR seed);
    forward_loop is rounds 1-5's form.  Same arithmetic in the same order: equal to the last bit in fp64 and fp32, and so are the table
    gradients up to the order in which repeated rows are summed."""
    torch.manual_seed(5)
    for dt in (torch.float64, torch.float32):
        e = H.HashGridEncoding().to(dt)
        with torch.no_grad():
            e.table.copy_(torch.randn_like(e.table))
        x = torch.rand(300, 3, dtype=dt) * 1.02 - 0.01                    # a few points outside the box
        a, b = e(x), e.forward_loop(x)
        assert torch.equal(a, b)
        w = torch.randn_like(a)
        ga, = torch.autograd.grad((a * w).sum(), e.table)
        gb, = torch.autograd.grad((b * w).sum(), e.table)
        tol = 1e-12 if dt == torch.float64 else 1e-5
        assert (ga - gb).abs().max().item() <= tol * gb.abs().max().item()
    sdf, _ = H.build_models(seed=4)
    x = torch.randn(50, 3) * 0.4
    eps = sdf.fd_eps
    offs = torch.eye(3) * eps
    ref = torch.cat([(sdf.sdf(x + offs[i]) - sdf.sdf(x - offs[i])) * (0.5 / eps) for i in range(3)], dim=-1)
    assert torch.equal(sdf.gradient(x).squeeze(1), ref)
