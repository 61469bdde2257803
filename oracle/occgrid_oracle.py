"""Occupancy-grid ray marching with packed variable-length rays -- the sampler of the instant-nsr-pl variant the reference
names as its direction (/root/reference/README.md:11,13: "Replace NeuS with instant-nsr-pl"; the code is on an unmounted `dev`
branch).  TEST INFRASTRUCTURE: a plain-PyTorch restatement, PARITY UNPINNED: instant-nsr-pl (bennyguo/
instant-nsr-pl, systems/neus.py + models/neus.py) and the nerfacc 0.3 OccupancyGrid / ray_marching it calls are absent from
/root/reference and unversioned there, so this file restates their published behaviour from memory and is the normative
definition for the HIP path (dynhor_amd/csrc/march.hip):

  * grid: res^3 cells over the cube [-radius, radius]^3; occ <- max(occ * decay, alpha(cell)); binary = occ > min(mean(occ), thre)
    (nerfacc OccupancyGrid._update / every_n_step);
  * alpha(cell) from the SDF at a jittered point of the cell (instant-nsr-pl NeuSModel.update_step.occ_eval_fn):
      inv_s = clip(exp(10 variance), 1e-6, 1e6); prev / next = sdf +/- step/2; alpha = clip((sig(prev inv_s) - sig(next inv_s)
      + 1e-5) / (sig(prev inv_s) + 1e-5), 0, 1);
  * marching (nerfacc.ray_marching, stratified, cone_angle = 0): samples [t_k, t_k + step], t_k = near + (k + u) * step with ONE
    uniform offset u per ray, kept iff t_k + step <= far and the cell holding the mid-point is occupied; packed front to back;
  * compositing: NeuS alpha from (sdf, normal, dir, dist = step) exactly as oracle/neus_oracle.py:render_core, weights
    w = alpha * prod_{earlier}(1 - alpha + 1e-7) within the ray's segment.
Build decisions (stated in DESIGN.md section 7): near / far are the unit-sphere bounds the rest of the repo uses (mid -/+ 1), at
most `max_samples` (128) samples per ray are kept (the first ones, front to back; nerfacc has no per-ray cap but limits the
total by adapting the ray count), and every grid update re-evaluates EVERY cell -- nerfacc does that only during its warm-up
(training step < 256) and afterwards refreshes a uniform quarter of the cells plus the occupied ones; that schedule is restated too
(OccupancyGrid.update(mask) / quarter_refresh_mask; product: hash_renderer.grid_refresh = "nerfacc") and was measured against the
default with 8 paired seeds (DESIGN_NEXT_ROWS.md section 7).  Both start from the initial SDF (nerfacc's step-0 update).
"""
from __future__ import annotations

import math

import torch


class OccupancyGrid:
    def __init__(self, res=128, radius=1.0, decay=0.95, thre=0.001, device="cpu"):
        self.res, self.radius, self.decay, self.thre = int(res), float(radius), float(decay), float(thre)
        self.occ = torch.zeros(self.res ** 3, device=device)
        self.binary = torch.ones(self.res ** 3, dtype=torch.bool, device=device)

    def cell_points(self, jitter: torch.Tensor) -> torch.Tensor:
        """[res^3,3] points, cell (ix,iy,iz) <-> linear index (ix*res + iy)*res + iz, jitter in [0,1)^3 inside the cell."""
        r = self.res
        ax = torch.arange(r, device=jitter.device, dtype=torch.float32)
        ix, iy, iz = torch.meshgrid(ax, ax, ax, indexing="ij")
        idx = torch.stack([ix, iy, iz], -1).reshape(-1, 3)
        return ((idx + jitter) / r * 2.0 - 1.0) * self.radius

    def update(self, alpha: torch.Tensor, mask: torch.Tensor | None = None):
        """mask None: every cell (nerfacc's warm-up rule, this repo's default at every update); mask [res^3] bool: only those cells
        are decayed and re-evaluated (nerfacc's rule after the warm-up: quarter_refresh_mask below)."""
        new = torch.maximum(self.occ * self.decay, alpha.reshape(-1))
        self.occ = new if mask is None else torch.where(mask.reshape(-1), new, self.occ)
        self.binary = self.occ > torch.clamp(self.occ.mean(), max=self.thre)

    def quarter_refresh_mask(self, u: torch.Tensor) -> torch.Tensor:
        """nerfacc 0.3 OccupancyGrid._update for step >= warmup_steps: n = cells // 4 cells drawn uniformly (with replacement) plus
        the occupied cells (all of them when there are at most n, else n drawn from them with replacement).  Restated as per-cell
        inclusion probabilities -- 1 - exp(-n / cells) for the uniform draw, min(1, 1 - exp(-n / occupied)) for the occupied draw --
        decided by ONE uniform u per cell: below p_uni the cell is a uniform draw, the rest of the range rescaled decides the occupied
        draw."""
        cells = self.occ.numel()
        n = cells // 4
        k = float(self.binary.sum())
        p_uni = 1.0 - math.exp(-n / cells)
        p_occ = 1.0 if k <= n else 1.0 - math.exp(-n / k)
        hit = u < p_uni
        return hit | (self.binary & ((u - p_uni) / (1.0 - p_uni) < p_occ))

    def query(self, x: torch.Tensor) -> torch.Tensor:
        """occupied? [N] for points [N,3]; outside the cube -> False.  fp32 op order = the HIP kernel's."""
        r = self.res
        g = (x * (0.5 / self.radius) + 0.5) * float(r)
        i = torch.floor(g).long()
        inside = ((i >= 0) & (i < r)).all(-1)
        i = i.clamp(0, r - 1)
        lin = (i[:, 0] * r + i[:, 1]) * r + i[:, 2]
        return self.binary[lin] & inside


def occ_alpha(sdf: torch.Tensor, inv_s: torch.Tensor, step: float) -> torch.Tensor:
    prev = torch.sigmoid((sdf + 0.5 * step) * inv_s)
    nxt = torch.sigmoid((sdf - 0.5 * step) * inv_s)
    return ((prev - nxt + 1e-5) / (prev + 1e-5)).clip(0.0, 1.0)


def march(rays_o, rays_d, near, far, u, grid: OccupancyGrid, step: float, max_samples=128):
    """Packed samples of B rays.  Returns dict(t_start [N], ray_idx [N], off [B], cnt [B], truncated [B] bool).  All fp32."""
    B = rays_o.shape[0]
    near, far, u = near.reshape(B), far.reshape(B), u.reshape(B)
    kmax = int(torch.ceil((far - near).max() / step).item()) + 1
    k = torch.arange(kmax, device=rays_o.device, dtype=torch.float32)
    t0 = near[:, None] + (k[None, :] + u[:, None]) * step                       # [B,K]
    tm = t0 + 0.5 * step
    x = rays_o[:, None, :] + rays_d[:, None, :] * tm[..., None]
    occ = grid.query(x.reshape(-1, 3)).reshape(B, kmax) & ((t0 + step) <= far[:, None])
    rank = torch.cumsum(occ.long(), dim=1) - 1
    keep = occ & (rank < max_samples)
    cnt = keep.sum(1)
    off = torch.cumsum(cnt, 0) - cnt
    ridx = torch.arange(B, device=rays_o.device)[:, None].expand(B, kmax)
    return {"t_start": t0[keep], "ray_idx": ridx[keep], "off": off, "cnt": cnt, "truncated": occ.sum(1) > max_samples}


def render_packed(pts, sdf, normals, colors, rays_d, ray_idx, off, cnt, step, inv_s, cos_anneal_ratio, background_rgb=None):
    """NeuS compositing over packed segments (dist = step for every sample).  pts [N,3] (sample mid-points), sdf [N],
    normals [N,3], colors [N,3].  Returns dict(color_fine [B,3], weight_sum [B,1], weights [N], normal_map [B,3],
    gradient_error (scalar, relax-masked like render_core), alpha [N])."""
    B = off.shape[0]
    N = sdf.shape[0]
    dirs = rays_d[ray_idx]
    true_cos = (dirs * normals).sum(-1)
    iter_cos = -(torch.relu(-true_cos * 0.5 + 0.5) * (1.0 - cos_anneal_ratio) + torch.relu(-true_cos) * cos_anneal_ratio)
    en = sdf + iter_cos * step * 0.5
    ep = sdf - iter_cos * step * 0.5
    prev, nxt = torch.sigmoid(ep * inv_s), torch.sigmoid(en * inv_s)
    alpha = ((prev - nxt + 1e-5) / (prev + 1e-5)).clip(0.0, 1.0)
    # transmittance inside each segment: pad to [B, maxcnt]
    m = int(cnt.max().item()) if N > 0 else 0
    pos = torch.arange(N, device=sdf.device) - off[ray_idx]
    dense = torch.zeros(B, max(m, 1), dtype=sdf.dtype, device=sdf.device)
    dense[ray_idx, pos] = alpha
    T = torch.cumprod(torch.cat([torch.ones(B, 1, dtype=sdf.dtype, device=sdf.device), 1.0 - dense + 1e-7], -1), -1)[:, :-1]
    w = alpha * T[ray_idx, pos]
    wsum = torch.zeros(B, dtype=sdf.dtype, device=sdf.device).index_add_(0, ray_idx, w)
    color = torch.zeros(B, 3, dtype=sdf.dtype, device=sdf.device).index_add_(0, ray_idx, w[:, None] * colors)
    nmap = torch.zeros(B, 3, dtype=sdf.dtype, device=sdf.device).index_add_(0, ray_idx, w[:, None] * normals)
    if background_rgb is not None:
        color = color + background_rgb * (1.0 - wsum[:, None])
    gn = torch.linalg.norm(normals, dim=-1)
    relax = (torch.linalg.norm(pts, dim=-1) < 1.2).to(sdf.dtype)
    gerr = (relax * (gn - 1.0) ** 2).sum() / (relax.sum() + 1e-5)
    return {"color_fine": color, "weight_sum": wsum[:, None], "weights": w, "normal_map": nmap, "gradient_error": gerr,
            "alpha": alpha}
