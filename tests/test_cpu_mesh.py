"""Host logic: marching tetrahedra + PLY writer (dynhor_amd/mesh.py) on analytic fields, CPU tensors."""
import os

import torch

from dynhor_amd.mesh import marching_tetrahedra, write_ply


def _grid(n):
    ax = torch.linspace(-1, 1, n)
    return torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), -1)


def test_sphere_surface_is_closed_and_on_the_level_set():
    g = _grid(40)
    u = 0.5 - g.norm(dim=-1)                         # the renderer hands over -sdf
    v, f = marching_tetrahedra(u, 0.0, [-1, -1, -1], [1, 1, 1])
    assert v.shape[0] > 3000 and f.shape[0] > 6000 and f.min().item() >= 0 and f.max().item() < v.shape[0]
    assert (v.norm(dim=1) - 0.5).abs().max().item() < 5e-3          # linear interpolation on a 0.05 grid
    # closed 2-manifold: every undirected edge belongs to exactly two triangles, Euler characteristic 2
    e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]).sort(dim=1).values
    uniq, cnt = torch.unique(e, dim=0, return_counts=True)
    assert (cnt == 2).all()
    assert v.shape[0] - uniq.shape[0] + f.shape[0] == 2
    # consistently oriented: signed volume = +-(4/3) pi r^3
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    vol = (a * torch.cross(b, c, dim=1)).sum(dim=1).sum().item() / 6.0
    assert abs(abs(vol) - 4.0 / 3.0 * torch.pi * 0.125) < 0.01


def test_threshold_and_bounds_are_respected(tmp_path):
    g = _grid(24)
    u = 0.5 - g.norm(dim=-1)
    v, f = marching_tetrahedra(u, 0.2, [-2, -2, -2], [2, 2, 2])       # level set u = 0.2 <=> radius 0.3 in grid units, x2 scale
    assert (v.norm(dim=1) - 0.6).abs().max().item() < 0.03
    empty_v, empty_f = marching_tetrahedra(u, 10.0, [-1, -1, -1], [1, 1, 1])
    assert empty_v.shape[0] == 0 and empty_f.shape[0] == 0
    path = os.path.join(tmp_path, "m.ply")
    write_ply(path, v, f)
    head = open(path, "rb").read(200).decode("latin1")
    assert head.startswith("ply") and f"element vertex {v.shape[0]}" in head and f"element face {f.shape[0]}" in head


def test_hausdorff_distance_to_the_analytic_scene_surface_at_resolution_64():
    """VERDICT r3 next #4c: upstream extract_geometry triangulates with marching cubes, this repo with marching tetrahedra -- two
    triangulations of the same level set.  The number behind "the mesh is the surface": on the bench's analytic scene (sphere +
    rounded box, dynhor_amd/scene.py) at resolution 64 over [-1, 1]^3 (cell 0.0317) the two-sided Hausdorff distance between the
    mesh and the analytic zero set is bounded by a small fraction of a cell."""
    from dynhor_amd.scene import scene_sdf
    N = 64
    g = _grid(N)
    u = -scene_sdf(g.reshape(-1, 3)).reshape(N, N, N)
    v, f = marching_tetrahedra(u, 0.0, [-1, -1, -1], [1, 1, 1])
    cell = 2.0 / (N - 1)
    # mesh -> surface: every vertex and every triangle centroid (a unit-gradient SDF: |sdf| IS the distance)
    d_v = scene_sdf(v).abs().max().item()
    cen = (v[f[:, 0]] + v[f[:, 1]] + v[f[:, 2]]) / 3.0
    d_c = scene_sdf(cen).abs().max().item()
    # surface -> mesh: points projected onto the analytic surface by Newton steps along the gradient, then the nearest mesh vertex
    gen = torch.Generator().manual_seed(0)
    p = torch.nn.functional.normalize(torch.randn(4000, 3, generator=gen), dim=1) * 0.45
    for _ in range(12):
        p = p.detach().requires_grad_(True)
        s = scene_sdf(p)
        (gr,) = torch.autograd.grad(s.sum(), p)
        p = p - s.detach()[:, None] * gr / (gr.norm(dim=1, keepdim=True) ** 2 + 1e-12)
    p = p.detach()
    assert scene_sdf(p).abs().max().item() < 1e-5
    d_s = torch.cdist(p, v).min(dim=1).values.max().item()
    print(f"resolution 64: vertices {v.shape[0]}, triangles {f.shape[0]}; mesh -> surface {max(d_v, d_c):.2e} (vertices {d_v:.2e}, "
          f"centroids {d_c:.2e}); surface -> nearest vertex {d_s:.2e}; cell {cell:.4f}")
    assert d_v < 0.25 * cell            # linear interpolation along cell edges; the maximum sits at the crease where sphere and box meet
    assert d_c < 0.5 * cell             # flat triangles against a curved surface and across that crease
    assert d_s < 1.0 * cell             # no hole: some vertex within one cell of every surface point
    e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]).sort(dim=1).values
    _, cnt = torch.unique(e, dim=0, return_counts=True)
    assert (cnt == 2).all(), "closed 2-manifold"
