"""Iso-surface extraction for ``validate_mesh`` (upstream NeuSRenderer.extract_geometry uses
mcubes.marching_cubes, not installable here).  Marching tetrahedra on the SDF grid: every cube is split into 6
tetrahedra around its main diagonal; each tetrahedron contributes 0, 1 or 2 triangles with vertices linearly
interpolated on the edges where the field crosses the threshold.  Plain torch ops (host-side plumbing around the HIP
SDF queries), watertight by construction because neighbouring cubes split their shared faces identically."""
from __future__ import annotations

import numpy as np
import torch

# cube corners (x,y,z bits) and the 6 tetrahedra sharing the diagonal 0-7
_CORNERS = torch.tensor([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [0, 0, 1], [1, 0, 1], [0, 1, 1], [1, 1, 1]])
_TETS = torch.tensor([[0, 1, 3, 7], [0, 3, 2, 7], [0, 2, 6, 7], [0, 6, 4, 7], [0, 4, 5, 7], [0, 5, 1, 7]])


def marching_tetrahedra(u: torch.Tensor, threshold: float, bound_min, bound_max):
    """u [N,N,N] scalar field (upstream convention: u = -sdf, inside > threshold).  Returns (vertices [V,3] float32 in
    world units, triangles [T,3] int64)."""
    dev = u.device
    N = u.shape[0]
    bmin = torch.as_tensor(bound_min, dtype=torch.float32, device=dev)
    bmax = torch.as_tensor(bound_max, dtype=torch.float32, device=dev)
    # active cells first (8 shifted views of the grid: no per-cell index tensors for the empty 99 % of the volume)
    f = u - threshold
    sl = (slice(0, N - 1), slice(1, N))
    vmax = torch.full((N - 1, N - 1, N - 1), -float("inf"), device=dev)
    vmin = torch.full((N - 1, N - 1, N - 1), float("inf"), device=dev)
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                c = f[sl[dx], sl[dy], sl[dz]]
                vmax = torch.maximum(vmax, c)
                vmin = torch.minimum(vmin, c)
    base = ((vmax > 0) & (vmin <= 0)).nonzero().reshape(-1, 1, 3)              # [C,1,3] active cells only
    corners = base + _CORNERS.to(dev).reshape(1, 8, 3)                          # [C,8,3]
    vals = f[corners[..., 0], corners[..., 1], corners[..., 2]]               # [C,8]
    tets = _TETS.to(dev)
    tv = vals[:, tets]                                                         # [C,6,4]
    tp = corners[:, tets].float()                                              # [C,6,4,3]
    tv = tv.reshape(-1, 4); tp = tp.reshape(-1, 4, 3)
    inside = tv > 0
    n_in = inside.sum(dim=1)
    tris = []

    def interp(pa, pb, va, vb):
        t = (va / (va - vb)).clamp(0, 1).unsqueeze(-1)
        return pa + t * (pb - pa)

    # one corner inside (or one outside): a single triangle on the three edges leaving that corner
    for count, flip in ((1, False), (3, True)):
        sel = n_in == count
        if sel.any():
            v, p, ins = tv[sel], tp[sel], inside[sel]
            lone = (ins if count == 1 else ~ins).float().argmax(dim=1)          # index of the lone corner
            others = torch.stack([(lone + k) % 4 for k in (1, 2, 3)], dim=1)
            ar = torch.arange(v.shape[0], device=dev)
            pa, va = p[ar, lone], v[ar, lone]
            pts = [interp(pa, p[ar, others[:, k]], va, v[ar, others[:, k]]) for k in range(3)]
            tri = torch.stack(pts, dim=1)
            # orientation: make the normal point from inside (u > thr) to outside
            n = torch.linalg.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
            to_lone = pa - tri.mean(dim=1)
            wrong = ((n * to_lone).sum(-1) > 0) if count == 1 else ((n * to_lone).sum(-1) < 0)
            tri[wrong] = tri[wrong][:, [0, 2, 1]]
            tris.append(tri)
    # two inside / two outside: a quad (two triangles) across the four mixed edges
    sel = n_in == 2
    if sel.any():
        v, p, ins = tv[sel], tp[sel], inside[sel]
        order = torch.argsort(ins.int(), dim=1, descending=True, stable=True)   # [in0, in1, out0, out1]
        ar = torch.arange(v.shape[0], device=dev)
        g = lambda k: (p[ar, order[:, k]], v[ar, order[:, k]])
        (pi0, vi0), (pi1, vi1), (po0, vo0), (po1, vo1) = g(0), g(1), g(2), g(3)
        q00, q01 = interp(pi0, po0, vi0, vo0), interp(pi0, po1, vi0, vo1)
        q10, q11 = interp(pi1, po0, vi1, vo0), interp(pi1, po1, vi1, vo1)
        for tri in (torch.stack([q00, q01, q11], 1), torch.stack([q00, q11, q10], 1)):
            n = torch.linalg.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
            to_in = 0.5 * (pi0 + pi1) - tri.mean(dim=1)
            wrong = (n * to_in).sum(-1) > 0
            tri[wrong] = tri[wrong][:, [0, 2, 1]]
            tris.append(tri)
    if not tris:
        return torch.zeros(0, 3, device=dev), torch.zeros(0, 3, dtype=torch.int64, device=dev)
    tri = torch.cat(tris, dim=0)                                               # [T,3,3] in grid units
    area2 = torch.linalg.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]).norm(dim=-1)
    tri = tri[area2 > 1e-12]
    # weld vertices shared between tetrahedra (quantised to 1e-5 grid units)
    flat = tri.reshape(-1, 3)
    key = torch.round(flat * 1e5).to(torch.int64)
    uniq, inv = torch.unique(key, dim=0, return_inverse=True)
    verts = torch.zeros(uniq.shape[0], 3, device=dev).index_copy_(0, inv, flat)
    faces = inv.reshape(-1, 3)
    verts = verts / (N - 1) * (bmax - bmin) + bmin
    return verts, faces


def write_ply(path: str, verts: torch.Tensor, faces: torch.Tensor):
    v = verts.detach().cpu().numpy().astype(np.float32)
    f = faces.detach().cpu().numpy().astype(np.int32)
    with open(path, "wb") as fh:
        fh.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\n"
                  "property float z\nelement face %d\nproperty list uchar int vertex_indices\nend_header\n"
                  % (len(v), len(f))).encode())
        fh.write(v.tobytes())
        rec = np.empty(len(f), dtype=[("n", "u1"), ("i", "<i4", (3,))])
        rec["n"] = 3
        rec["i"] = f
        fh.write(rec.tobytes())
