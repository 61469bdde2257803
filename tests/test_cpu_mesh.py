"""Host logic: marching cubes (upstream's algorithm, generated case table), marching tetrahedra and the PLY writer (dynhor_amd/mesh.py)
on analytic fields, CPU tensors."""
import os

import pytest
import torch

from dynhor_amd.mesh import marching_cubes, marching_cubes_table, marching_tetrahedra, write_ply

METHODS = {"cubes": marching_cubes, "tetrahedra": marching_tetrahedra}


def _grid(n):
    ax = torch.linspace(-1, 1, n)
    return torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), -1)


def test_generated_case_table_has_the_classic_shape():
    tri, n = marching_cubes_table()
    assert tri.shape == (256, 5, 3) and int(n[0]) == 0 and int(n[255]) == 0
    # 820 triangles over the 256 cases, at most 5 per case, complementary cases mirror each other's counts only where no face is
    # ambiguous (the face rule separates INSIDE corners: e.g. two inside corners on a face diagonal give 2 triangles, two outside 4)
    assert int(n.sum()) == 820 and int(n.max()) == 5
    assert torch.bincount(n).tolist() == [2, 16, 50, 80, 76, 32]
    for c in (1, 2, 4, 8, 16, 32, 64, 128):            # one corner inside: one triangle on the three edges that leave it
        assert int(n[c]) == 1 and int(n[255 - c]) == 1
    assert sorted(tri[1, 0].tolist()) == [0, 3, 8] and sorted(tri[64, 0].tolist()) == [5, 6, 10]
    # every edge id a case uses is a cut edge of that case
    from dynhor_amd.mesh import _MC_EDGES
    for case in range(256):
        cut = {i for i, (a, b) in enumerate(_MC_EDGES) if ((case >> a) & 1) != ((case >> b) & 1)}
        used = set(tri[case][: int(n[case])].reshape(-1).tolist())
        assert used == cut, case


@pytest.mark.parametrize("method", ["cubes", "tetrahedra"])
def test_sphere_surface_is_closed_and_on_the_level_set(method):
    g = _grid(40)
    u = 0.5 - g.norm(dim=-1)                         # the renderer hands over -sdf
    v, f = METHODS[method](u, 0.0, [-1, -1, -1], [1, 1, 1])
    lo = (1500, 3000) if method == "cubes" else (3000, 6000)
    assert v.shape[0] > lo[0] and f.shape[0] > lo[1] and f.min().item() >= 0 and f.max().item() < v.shape[0]
    assert (v.norm(dim=1) - 0.5).abs().max().item() < 5e-3          # linear interpolation on a 0.05 grid
    # closed 2-manifold: every undirected edge belongs to exactly two triangles, Euler characteristic 2
    e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]).sort(dim=1).values
    uniq, cnt = torch.unique(e, dim=0, return_counts=True)
    assert (cnt == 2).all()
    assert v.shape[0] - uniq.shape[0] + f.shape[0] == 2
    # consistently oriented: signed volume = +-(4/3) pi r^3
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    vol = (a * torch.cross(b, c, dim=1)).sum(dim=1).sum().item() / 6.0
    assert abs(vol - 4.0 / 3.0 * torch.pi * 0.125) < 0.01          # positive: normals point outward


def test_marching_cubes_stays_closed_on_fields_full_of_ambiguous_faces():
    """Two touching blobs, a thin torus and smooth noise: faces with diagonal inside corners occur by the hundred; the face rule of
    the generated table keeps every undirected edge in exactly two triangles, with opposite directions (consistent orientation)."""
    g = _grid(33)
    torch.manual_seed(3)
    noise = torch.nn.functional.interpolate(torch.randn(1, 1, 9, 9, 9), size=(33, 33, 33), mode="trilinear", align_corners=True)[0, 0]
    fields = {
        "touching blobs": torch.maximum(0.36 - (g - torch.tensor([0.33, 0.0, 0.0])).norm(dim=-1), 0.36 - (g + torch.tensor([0.33, 0.02, 0.01])).norm(dim=-1)),
        "thin torus": 0.07 - ((g[..., :2].norm(dim=-1) - 0.55) ** 2 + g[..., 2] ** 2).sqrt(),
        "smooth noise": noise,
        "white noise": torch.randn(33, 33, 33),
    }
    for name, u in fields.items():
        ins = u > 0
        amb = 0
        for ax in range(3):                          # count ambiguous faces (diagonal corners agree, neighbours differ)
            a = ins.movedim(ax, 0)
            c00, c01, c10, c11 = a[:, :-1, :-1], a[:, :-1, 1:], a[:, 1:, :-1], a[:, 1:, 1:]
            amb += int(((c00 == c11) & (c01 == c10) & (c00 != c01)).sum())
        v, f = marching_cubes(u, 0.0, [-1, -1, -1], [1, 1, 1])
        # away from the grid boundary every directed edge has its reverse exactly once
        d = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        und, cnt = torch.unique(d.sort(dim=1).values, dim=0, return_counts=True)
        interior = ((v[und] .abs().amax(dim=-1)) < 1.0 - 1e-6).all(dim=1)      # an edge with a vertex on the box boundary may be open
        assert (cnt[interior] == 2).all(), name
        dirs, dcnt = torch.unique(d, dim=0, return_counts=True)
        assert (dcnt == 1).all(), name                # no directed edge twice: orientation is consistent
        print(f"{name}: {amb} ambiguous faces, {v.shape[0]} vertices, {f.shape[0]} triangles")
        if "noise" in name:
            assert amb > 50


@pytest.mark.parametrize("method", ["cubes", "tetrahedra"])
def test_threshold_and_bounds_are_respected(tmp_path, method):
    g = _grid(24)
    u = 0.5 - g.norm(dim=-1)
    v, f = METHODS[method](u, 0.2, [-2, -2, -2], [2, 2, 2])       # level set u = 0.2 <=> radius 0.3 in grid units, x2 scale
    assert (v.norm(dim=1) - 0.6).abs().max().item() < 0.03
    empty_v, empty_f = METHODS[method](u, 10.0, [-1, -1, -1], [1, 1, 1])
    assert empty_v.shape[0] == 0 and empty_f.shape[0] == 0
    path = os.path.join(tmp_path, "m.ply")
    write_ply(path, v, f)
    head = open(path, "rb").read(200).decode("latin1")
    assert head.startswith("ply") and f"element vertex {v.shape[0]}" in head and f"element face {f.shape[0]}" in head


@pytest.mark.parametrize("method", ["cubes", "tetrahedra"])
def test_hausdorff_distance_to_the_analytic_scene_surface_at_resolution_64(method):
    """VERDICT r3 next #4c: upstream extract_geometry triangulates with marching cubes, this repo with marching tetrahedra -- two
    triangulations of the same level set.  The number behind "the mesh is the surface": on the bench's analytic scene (sphere +
    rounded box, dynhor_amd/scene.py) at resolution 64 over [-1, 1]^3 (cell 0.0317) the two-sided Hausdorff distance between the
    mesh and the analytic zero set is bounded by a small fraction of a cell."""
    from dynhor_amd.scene import scene_sdf
    N = 64
    g = _grid(N)
    u = -scene_sdf(g.reshape(-1, 3)).reshape(N, N, N)
    v, f = METHODS[method](u, 0.0, [-1, -1, -1], [1, 1, 1])
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
    print(f"{method}, resolution 64: vertices {v.shape[0]}, triangles {f.shape[0]}; mesh -> surface {max(d_v, d_c):.2e} (vertices {d_v:.2e}, "
          f"centroids {d_c:.2e}); surface -> nearest vertex {d_s:.2e}; cell {cell:.4f}")
    assert d_v < 0.25 * cell            # linear interpolation along cell edges; the maximum sits at the crease where sphere and box meet
    assert d_c < 0.5 * cell             # flat triangles against a curved surface and across that crease
    assert d_s < 1.0 * cell             # no hole: some vertex within one cell of every surface point
    e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]).sort(dim=1).values
    _, cnt = torch.unique(e, dim=0, return_counts=True)
    assert (cnt == 2).all(), "closed 2-manifold"
