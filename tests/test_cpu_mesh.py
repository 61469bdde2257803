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
