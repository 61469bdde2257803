"""Plain-PyTorch restatement of the NeuS training hot path (TEST INFRASTRUCTURE -- see oracle/__init__.py).

PARITY UNPINNED at the reference: the stage is absent from /root/reference (README.md:7-11,55-58).
Every function cites the SURVEY.md Appendix-A paragraph (recollection of public Totoro97/NeuS) it
restates, and the reference file:line that constrains its inputs where one exists.

Device-agnostic, dtype-agnostic (fp32 for the CPU baseline, fp64 for tight checks).
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# A.1 Embedder
# ----------------------------------------------------------------------------------------------
def embed(x: torch.Tensor, multires: int) -> torch.Tensor:
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^{L-1} x), cos(2^{L-1} x)]  (SURVEY App. A.1)."""
    outs = [x]
    for k in range(multires):
        f = float(2 ** k)
        outs.append(torch.sin(x * f))
        outs.append(torch.cos(x * f))
    return torch.cat(outs, dim=-1)


class _WNLinear(nn.Module):
    """Linear with legacy weight-norm parameter names (bias, weight_g, weight_v); SURVEY §5 checkpoint row."""

    def __init__(self, weight: torch.Tensor, bias: torch.Tensor):
        super().__init__()
        self.bias = nn.Parameter(bias.clone())
        self.weight_g = nn.Parameter(weight.norm(dim=1, keepdim=True).clone())
        self.weight_v = nn.Parameter(weight.clone())

    def weight(self) -> torch.Tensor:
        return self.weight_g * self.weight_v / self.weight_v.norm(dim=1, keepdim=True)

    def forward(self, x):
        return F.linear(x, self.weight(), self.bias)


# ----------------------------------------------------------------------------------------------
# A.2 SDFNetwork
# ----------------------------------------------------------------------------------------------
class SDFNetwork(nn.Module):
    def __init__(self, d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=(4,), multires=6, bias=0.5,
                 scale=1.0, geometric_init=True, weight_norm=True, inside_outside=False):
        super().__init__()
        assert weight_norm and geometric_init and not inside_outside
        self.multires = multires
        d0 = d_in + 2 * d_in * multires if multires > 0 else d_in
        dims = [d0] + [d_hidden] * n_layers + [d_out]
        self.dims = dims
        self.num_layers = len(dims)
        self.skip_in = tuple(skip_in)
        self.scale = scale
        for l in range(self.num_layers - 1):
            out_dim = dims[l + 1] - dims[0] if (l + 1) in self.skip_in else dims[l + 1]
            w = torch.empty(out_dim, dims[l])
            b = torch.empty(out_dim)
            if l == self.num_layers - 2:
                nn.init.normal_(w, mean=math.sqrt(math.pi) / math.sqrt(dims[l]), std=0.0001)
                nn.init.constant_(b, -bias)
            elif multires > 0 and l == 0:
                nn.init.constant_(b, 0.0)
                nn.init.constant_(w[:, 3:], 0.0)
                nn.init.normal_(w[:, :3], 0.0, math.sqrt(2) / math.sqrt(out_dim))
            elif multires > 0 and l in self.skip_in:
                nn.init.constant_(b, 0.0)
                nn.init.normal_(w, 0.0, math.sqrt(2) / math.sqrt(out_dim))
                nn.init.constant_(w[:, -(dims[0] - 3):], 0.0)
            else:
                nn.init.constant_(b, 0.0)
                nn.init.normal_(w, 0.0, math.sqrt(2) / math.sqrt(out_dim))
            setattr(self, "lin" + str(l), _WNLinear(w, b))
        self.activation = nn.Softplus(beta=100)

    def forward(self, inputs):
        inputs = inputs * self.scale
        e = embed(inputs, self.multires) if self.multires > 0 else inputs
        x = e
        for l in range(self.num_layers - 1):
            lin = getattr(self, "lin" + str(l))
            if l in self.skip_in:
                x = torch.cat([x, e], 1) / math.sqrt(2)
            x = lin(x)
            if l < self.num_layers - 2:
                x = self.activation(x)
        return torch.cat([x[:, :1] / self.scale, x[:, 1:]], dim=-1)

    def sdf(self, x):
        return self.forward(x)[:, :1]

    def gradient(self, x):
        x = x.requires_grad_(True)
        with torch.enable_grad():
            y = self.sdf(x)
            d_output = torch.ones_like(y, requires_grad=False)
            g = torch.autograd.grad(outputs=y, inputs=x, grad_outputs=d_output, create_graph=True,
                                    retain_graph=True, only_inputs=True)[0]
        return g.unsqueeze(1)


# ----------------------------------------------------------------------------------------------
# A.3 RenderingNetwork (mode idr) + SingleVarianceNetwork
# ----------------------------------------------------------------------------------------------
class RenderingNetwork(nn.Module):
    def __init__(self, d_feature=256, mode="idr", d_in=9, d_out=3, d_hidden=256, n_layers=4, weight_norm=True,
                 multires_view=4, squeeze_out=True):
        super().__init__()
        assert mode == "idr" and weight_norm
        self.multires_view = multires_view
        self.squeeze_out = squeeze_out
        d0 = d_in + d_feature + (2 * 3 * multires_view if multires_view > 0 else 0)
        dims = [d0] + [d_hidden] * n_layers + [d_out]
        self.num_layers = len(dims)
        for l in range(self.num_layers - 1):
            lin = nn.Linear(dims[l], dims[l + 1])  # default torch init, as upstream
            setattr(self, "lin" + str(l), _WNLinear(lin.weight.data, lin.bias.data))

    def forward(self, points, normals, view_dirs, feature_vectors):
        if self.multires_view > 0:
            view_dirs = embed(view_dirs, self.multires_view)
        x = torch.cat([points, view_dirs, normals, feature_vectors], dim=-1)
        for l in range(self.num_layers - 1):
            x = getattr(self, "lin" + str(l))(x)
            if l < self.num_layers - 2:
                x = F.relu(x)
        if self.squeeze_out:
            x = torch.sigmoid(x)
        return x


class SingleVarianceNetwork(nn.Module):
    def __init__(self, init_val=0.3):
        super().__init__()
        self.register_parameter("variance", nn.Parameter(torch.tensor(float(init_val))))

    def forward(self, x):
        return torch.ones([len(x), 1], device=x.device, dtype=self.variance.dtype) * torch.exp(self.variance * 10.0)


# ----------------------------------------------------------------------------------------------
# A.6 sample_pdf / up_sample / cat_z_vals ; A.7 render_core ; A.5 render
# ----------------------------------------------------------------------------------------------
def sample_pdf(bins, weights, n_samples, det=False):
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if det:
        u = torch.linspace(0.0 + 0.5 / n_samples, 1.0 - 0.5 / n_samples, steps=n_samples,
                           device=bins.device, dtype=bins.dtype)
        u = u.expand(list(cdf.shape[:-1]) + [n_samples])
    else:
        u = torch.rand(list(cdf.shape[:-1]) + [n_samples], device=bins.device, dtype=bins.dtype)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.max(torch.zeros_like(inds - 1), inds - 1)
    above = torch.min((cdf.shape[-1] - 1) * torch.ones_like(inds), inds)
    inds_g = torch.stack([below, above], -1)
    matched_shape = [inds_g.shape[0], inds_g.shape[1], cdf.shape[-1]]
    cdf_g = torch.gather(cdf.unsqueeze(1).expand(matched_shape), 2, inds_g)
    bins_g = torch.gather(bins.unsqueeze(1).expand(matched_shape), 2, inds_g)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_g[..., 0]) / denom
    return bins_g[..., 0] + t * (bins_g[..., 1] - bins_g[..., 0])


class NeuSRenderer:
    def __init__(self, nerf, sdf_network, deviation_network, color_network, n_samples, n_importance, n_outside,
                 up_sample_steps, perturb):
        assert n_outside == 0, "n_outside>0 (NeRF++ background) is out of scope with masks (SURVEY App. A.5)"
        self.nerf = nerf
        self.sdf_network = sdf_network
        self.deviation_network = deviation_network
        self.color_network = color_network
        self.n_samples = n_samples
        self.n_importance = n_importance
        self.n_outside = n_outside
        self.up_sample_steps = up_sample_steps
        self.perturb = perturb

    def up_sample(self, rays_o, rays_d, z_vals, sdf, n_importance, inv_s):
        batch_size, n_samples = z_vals.shape
        pts = rays_o[:, None, :] + rays_d[:, None, :] * z_vals[..., :, None]
        radius = torch.linalg.norm(pts, ord=2, dim=-1, keepdim=False)
        inside_sphere = (radius[:, :-1] < 1.0) | (radius[:, 1:] < 1.0)
        sdf = sdf.reshape(batch_size, n_samples)
        prev_sdf, next_sdf = sdf[:, :-1], sdf[:, 1:]
        prev_z_vals, next_z_vals = z_vals[:, :-1], z_vals[:, 1:]
        mid_sdf = (prev_sdf + next_sdf) * 0.5
        cos_val = (next_sdf - prev_sdf) / (next_z_vals - prev_z_vals + 1e-5)
        prev_cos_val = torch.cat([torch.zeros([batch_size, 1], device=z_vals.device, dtype=z_vals.dtype),
                                  cos_val[:, :-1]], dim=-1)
        cos_val = torch.stack([prev_cos_val, cos_val], dim=-1)
        cos_val, _ = torch.min(cos_val, dim=-1, keepdim=False)
        cos_val = cos_val.clip(-1e3, 0.0) * inside_sphere
        dist = next_z_vals - prev_z_vals
        prev_esti_sdf = mid_sdf - cos_val * dist * 0.5
        next_esti_sdf = mid_sdf + cos_val * dist * 0.5
        prev_cdf = torch.sigmoid(prev_esti_sdf * inv_s)
        next_cdf = torch.sigmoid(next_esti_sdf * inv_s)
        alpha = (prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)
        weights = alpha * torch.cumprod(
            torch.cat([torch.ones([batch_size, 1], device=z_vals.device, dtype=z_vals.dtype),
                       1.0 - alpha + 1e-7], -1), -1)[:, :-1]
        return sample_pdf(z_vals, weights, n_importance, det=True).detach()

    def cat_z_vals(self, rays_o, rays_d, z_vals, new_z_vals, sdf, last=False):
        batch_size, n_samples = z_vals.shape
        _, n_importance = new_z_vals.shape
        pts = rays_o[:, None, :] + rays_d[:, None, :] * new_z_vals[..., :, None]
        z_vals = torch.cat([z_vals, new_z_vals], dim=-1)
        z_vals, index = torch.sort(z_vals, dim=-1, stable=True)
        if not last:
            new_sdf = self.sdf_network.sdf(pts.reshape(-1, 3)).reshape(batch_size, n_importance)
            sdf = torch.cat([sdf, new_sdf], dim=-1)
            sdf = torch.gather(sdf, 1, index)
        return z_vals, sdf

    def render_core(self, rays_o, rays_d, z_vals, sample_dist, background_rgb=None, cos_anneal_ratio=0.0):
        batch_size, n_samples = z_vals.shape
        dists = z_vals[..., 1:] - z_vals[..., :-1]
        dists = torch.cat([dists, torch.full_like(dists[..., :1], sample_dist)], -1)
        mid_z_vals = z_vals + dists * 0.5
        pts = rays_o[:, None, :] + rays_d[:, None, :] * mid_z_vals[..., :, None]
        dirs = rays_d[:, None, :].expand(pts.shape)
        pts = pts.reshape(-1, 3)
        dirs = dirs.reshape(-1, 3)

        sdf_nn_output = self.sdf_network(pts)
        sdf = sdf_nn_output[:, :1]
        feature_vector = sdf_nn_output[:, 1:]
        gradients = self.sdf_network.gradient(pts).squeeze(1)
        sampled_color = self.color_network(pts, gradients, dirs, feature_vector).reshape(batch_size, n_samples, 3)

        inv_s = self.deviation_network(torch.zeros([1, 3], device=pts.device))[:, :1].clip(1e-6, 1e6)
        inv_s = inv_s.expand(batch_size * n_samples, 1)
        true_cos = (dirs * gradients).sum(-1, keepdim=True)
        iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - cos_anneal_ratio) +
                     F.relu(-true_cos) * cos_anneal_ratio)
        estimated_next_sdf = sdf + iter_cos * dists.reshape(-1, 1) * 0.5
        estimated_prev_sdf = sdf - iter_cos * dists.reshape(-1, 1) * 0.5
        prev_cdf = torch.sigmoid(estimated_prev_sdf * inv_s)
        next_cdf = torch.sigmoid(estimated_next_sdf * inv_s)
        p = prev_cdf - next_cdf
        c = prev_cdf
        alpha = ((p + 1e-5) / (c + 1e-5)).reshape(batch_size, n_samples).clip(0.0, 1.0)

        pts_norm = torch.linalg.norm(pts, ord=2, dim=-1, keepdim=True).reshape(batch_size, n_samples)
        inside_sphere = (pts_norm < 1.0).to(z_vals.dtype).detach()
        relax_inside_sphere = (pts_norm < 1.2).to(z_vals.dtype).detach()

        weights = alpha * torch.cumprod(
            torch.cat([torch.ones([batch_size, 1], device=z_vals.device, dtype=z_vals.dtype),
                       1.0 - alpha + 1e-7], -1), -1)[:, :-1]
        weights_sum = weights.sum(dim=-1, keepdim=True)
        color = (sampled_color * weights[:, :, None]).sum(dim=1)
        if background_rgb is not None:
            color = color + background_rgb * (1.0 - weights_sum)

        gradient_error = (torch.linalg.norm(gradients.reshape(batch_size, n_samples, 3), ord=2, dim=-1) - 1.0) ** 2
        gradient_error = (relax_inside_sphere * gradient_error).sum() / (relax_inside_sphere.sum() + 1e-5)
        return {
            "color": color, "sdf": sdf, "dists": dists,
            "gradients": gradients.reshape(batch_size, n_samples, 3),
            "s_val": 1.0 / inv_s, "mid_z_vals": mid_z_vals, "weights": weights,
            "cdf": c.reshape(batch_size, n_samples), "gradient_error": gradient_error,
            "inside_sphere": inside_sphere, "sampled_color": sampled_color,
        }

    def sample_z(self, rays_o, rays_d, near, far, perturb_overwrite=-1, t_rand=None):
        """Coarse + hierarchical sampling part of render() (App. A.5); returns final z_vals [B, n_s+n_i]."""
        batch_size = len(rays_o)
        z_vals = torch.linspace(0.0, 1.0, self.n_samples, device=rays_o.device, dtype=rays_o.dtype)
        z_vals = near + (far - near) * z_vals[None, :]
        perturb = self.perturb
        if perturb_overwrite >= 0:
            perturb = perturb_overwrite
        if perturb > 0:
            if t_rand is None:
                t_rand = torch.rand([batch_size, 1], device=rays_o.device, dtype=rays_o.dtype)
            z_vals = z_vals + (t_rand - 0.5) * 2.0 / self.n_samples
        if self.n_importance > 0:
            with torch.no_grad():
                pts = rays_o[:, None, :] + rays_d[:, None, :] * z_vals[..., :, None]
                sdf = self.sdf_network.sdf(pts.reshape(-1, 3)).reshape(batch_size, self.n_samples)
                for i in range(self.up_sample_steps):
                    new_z_vals = self.up_sample(rays_o, rays_d, z_vals, sdf,
                                                self.n_importance // self.up_sample_steps, 64 * 2 ** i)
                    z_vals, sdf = self.cat_z_vals(rays_o, rays_d, z_vals, new_z_vals, sdf,
                                                  last=(i + 1 == self.up_sample_steps))
        return z_vals

    def render(self, rays_o, rays_d, near, far, perturb_overwrite=-1, background_rgb=None, cos_anneal_ratio=0.0,
               t_rand=None, z_vals=None):
        batch_size = len(rays_o)
        sample_dist = 2.0 / self.n_samples
        if z_vals is None:
            z_vals = self.sample_z(rays_o, rays_d, near, far, perturb_overwrite, t_rand)
        n_samples = z_vals.shape[1]
        ret_fine = self.render_core(rays_o, rays_d, z_vals, sample_dist, background_rgb=background_rgb,
                                    cos_anneal_ratio=cos_anneal_ratio)
        weights = ret_fine["weights"]
        weights_sum = weights.sum(dim=-1, keepdim=True)
        s_val = ret_fine["s_val"].reshape(batch_size, n_samples).mean(dim=-1, keepdim=True)
        return {
            "color_fine": ret_fine["color"], "s_val": s_val, "cdf_fine": ret_fine["cdf"],
            "weight_sum": weights_sum, "weight_max": torch.max(weights, dim=-1, keepdim=True)[0],
            "gradients": ret_fine["gradients"], "weights": weights,
            "gradient_error": ret_fine["gradient_error"], "inside_sphere": ret_fine["inside_sphere"],
            "z_vals": z_vals, "sdf": ret_fine["sdf"], "sampled_color": ret_fine["sampled_color"],
        }


# ----------------------------------------------------------------------------------------------
# Ray generation (upstream Dataset.gen_random_rays_at + Dynhor conventions)
# ----------------------------------------------------------------------------------------------
def rays_from_pixels(px, py, K, R, T):
    """Rays in the OBJECT frame for pixels (px,py) of a frame with x_cam = R x_obj + T.

    Conventions: K per ObjTracker/run.py:119-123; pose per run.py:166, vis.py:52 (saved R is object->camera,
    T is [1,3]).  o = -R^T T, d = R^T normalize(K^-1 [u,v,1]).
    """
    p = torch.stack([px, py, torch.ones_like(px)], dim=-1).to(K.dtype)
    Kinv = torch.inverse(K)
    p = p @ Kinv.T
    v = p / torch.linalg.norm(p, ord=2, dim=-1, keepdim=True)
    d = v @ R            # R^T v  (row-vector form)
    o = (-(T.reshape(1, 3) @ R)).expand(d.shape)
    return o, d


def near_far_from_sphere(rays_o, rays_d):
    a = torch.sum(rays_d ** 2, dim=-1, keepdim=True)
    b = 2.0 * torch.sum(rays_o * rays_d, dim=-1, keepdim=True)
    mid = 0.5 * (-b) / a
    return mid - 1.0, mid + 1.0


def gather_rays(frames: Dict[str, torch.Tensor], img_idx: int, px, py):
    """[B,14] = o(3) d(3) rgb(3) obj_mask(1) keep_mask(1) mono_normal(3).

    frames: rgb u8 [F,H,W,3]; label i8 [F,H,W] with 1 object / 0 background / -1 hand
    (ObjTracker/run.py:66, utils/maskutils.py:24-28); normal u8 [F,H,W,3] (camera frame, n = u8/255*2-1);
    R [F,3,3], T [F,3], K [3,3].  obj_mask=(label>0), keep_mask=(label>=0) as pose_initializtion.py:60-61.
    """
    rgb = frames["rgb"][img_idx][(py, px)].to(torch.float32) / 255.0
    lab = frames["label"][img_idx][(py, px)]
    nrm = frames["normal"][img_idx][(py, px)].to(torch.float32) / 255.0 * 2.0 - 1.0
    o, d = rays_from_pixels(px, py, frames["K"], frames["R"][img_idx], frames["T"][img_idx])
    obj = (lab > 0).to(torch.float32)[:, None]
    keep = (lab >= 0).to(torch.float32)[:, None]
    return torch.cat([o, d, rgb, obj, keep, nrm], dim=-1)


# ----------------------------------------------------------------------------------------------
# Losses (App. A.8 + SURVEY §8 a11: hand-gated colour/mask losses, MonoSDF-style normal loss)
# ----------------------------------------------------------------------------------------------
def neus_losses(render_out, true_rgb, obj_mask, keep_mask, igr_weight=0.1, mask_weight=0.1,
                normal_weight=0.0, mono_normal=None, R=None):
    """Loss stack. Colour: L1 over (obj&keep) pixels normalised by their count (App. A.8 with mask := obj*keep);
    eikonal; BCE(weight_sum, obj) averaged over keep pixels (hand pixels contribute nothing, the gating
    precedent of utils/losses.py:69-71); optional monocular-normal L1 + (1-cos) over obj&keep with the
    rendered normal rotated into the camera frame by R (object->camera, run.py:166)."""
    color_fine = render_out["color_fine"]
    m = obj_mask * keep_mask
    mask_sum = m.sum() + 1e-5
    color_error = (color_fine - true_rgb) * m
    color_loss = color_error.abs().sum() / mask_sum
    psnr = 20.0 * torch.log10(1.0 / (((color_fine - true_rgb) ** 2 * m).sum() / (mask_sum * 3.0)).sqrt())
    eik = render_out["gradient_error"]
    ws = render_out["weight_sum"].clip(1e-3, 1.0 - 1e-3)
    bce = -(obj_mask * torch.log(ws) + (1.0 - obj_mask) * torch.log(1.0 - ws))
    mask_loss = (bce * keep_mask).sum() / (keep_mask.sum() + 1e-5)
    loss = color_loss + igr_weight * eik + mask_weight * mask_loss
    out = {"loss": loss, "color_loss": color_loss, "eikonal_loss": eik, "mask_loss": mask_loss, "psnr": psnr}
    if normal_weight > 0.0 and mono_normal is not None:
        n_obj = (render_out["gradients"] * render_out["weights"][:, :, None]).sum(dim=1)
        n_cam = n_obj @ R.T                      # x_cam = R x_obj
        n_hat = n_cam / (torch.linalg.norm(n_cam, dim=-1, keepdim=True) + 1e-6)
        l1 = (n_hat - mono_normal).abs().sum(-1, keepdim=True)
        cs = 1.0 - (n_hat * mono_normal).sum(-1, keepdim=True)
        normal_loss = ((l1 + cs) * m).sum() / mask_sum
        out["normal_loss"] = normal_loss
        out["loss"] = loss + normal_weight * normal_loss
    return out


# ----------------------------------------------------------------------------------------------
# Dense-correspondence reprojection loss (BASELINE.json configs[4] "DKM correspondence")
# ----------------------------------------------------------------------------------------------
# The reference names only the input folder (README.md:43 "correspondence_infos # dense correspondence obtained using DKM
# for reconstruction and outlier-voting"); neither file format nor loss is released, so -- like the monocular-normal term --
# the form below is this build's specification, parity unpinned:
#   a match says pixel p of frame i (the ray) and pixel q = (u, v) of frame j see the same surface point, certainty c.
#   expected ray depth   t^ = sum_k w_k m_k        (w: compositing weights, m: mid-point depths of render_core, detached)
#   surface estimate     x  = o + t^ d              (object frame)
#   reprojection         y  = R_j x + T_j ;  pi = (fx y0/y2 + cx, fy y1/y2 + cy)          (x_cam = R x_obj + T, run.py:166)
#   residual             s  = || pi - q || / fx                                            (focal-normalised pixels)
#   robust cost          rho(s) = s^2 / (2 delta) if s <= delta else s - delta/2,  delta = delta_px / fx      (Huber)
#   L_corr = sum_r c_r v_r rho(s_r) / (sum_r c_r v_r + 1e-5),   v_r = [y2 > 1e-3]  (point in front of camera j)
# "outlier voting" is a data-side step (vote_correspondences below): it only edits the certainties c.
def correspondence_loss(weights, z_vals, sample_dist, rays_o, rays_d, corr, R_all, T_all, K, delta_px=4.0):
    """corr [B,4] = (u_j, v_j, certainty, frame_j); certainty 0 marks a ray without a match.  Returns dict(loss, residual_px
    [B] (detached), depth [B])."""
    dists = torch.cat([z_vals[:, 1:] - z_vals[:, :-1], torch.full_like(z_vals[:, :1], sample_dist)], -1)
    mid = (z_vals + 0.5 * dists).detach()
    depth = (weights * mid).sum(-1)
    x = rays_o + depth[:, None] * rays_d
    j = corr[:, 3].long()
    y = torch.einsum("bij,bj->bi", R_all[j], x) + T_all[j].reshape(-1, 3)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    valid = (y[:, 2] > 1e-3).to(weights.dtype)
    y2 = torch.where(y[:, 2] > 1e-3, y[:, 2], torch.ones_like(y[:, 2]))
    pu = fx * y[:, 0] / y2 + cx
    pv = fy * y[:, 1] / y2 + cy
    e2 = (pu - corr[:, 0]) ** 2 + (pv - corr[:, 1]) ** 2
    s = torch.sqrt(e2 + 1e-24) / fx
    delta = delta_px / fx
    rho = torch.where(s <= delta, s * s / (2.0 * delta), s - 0.5 * delta)
    c = corr[:, 2] * valid
    loss = (c * rho).sum() / (c.sum() + 1e-5)
    return {"loss": loss, "residual_px": (s * fx).detach(), "depth": depth, "valid": valid}


def vote_correspondences(residual_px, corr, pair_id, tau_px=8.0, min_pair_inlier_ratio=0.5):
    """Outlier voting on the certainties (returns a new certainty vector): a match votes `inlier` if its reprojection
    residual under the current geometry is below tau_px; a frame PAIR whose inlier ratio is below min_pair_inlier_ratio is
    voted out as a whole (wrong pose or matcher failure), and inside kept pairs the outlier matches are dropped."""
    conf = corr[:, 2].clone()
    has = conf > 0
    inl = (residual_px < tau_px) & has
    out = conf.clone()
    for pid in torch.unique(pair_id[has]).tolist():
        sel = has & (pair_id == pid)
        ratio = inl[sel].float().mean()
        if ratio < min_pair_inlier_ratio:
            out[sel] = 0.0
        else:
            out[sel & ~inl] = 0.0
    return out


# ----------------------------------------------------------------------------------------------
# Schedules (App. A.8)
# ----------------------------------------------------------------------------------------------
def lr_factor(iter_step, warm_up_end, end_iter, alpha):
    if iter_step < warm_up_end:
        return iter_step / warm_up_end
    progress = (iter_step - warm_up_end) / (end_iter - warm_up_end)
    return (math.cos(math.pi * progress) + 1.0) * 0.5 * (1 - alpha) + alpha


def cos_anneal_ratio(iter_step, anneal_end):
    if anneal_end == 0.0:
        return 1.0
    return min(1.0, iter_step / anneal_end)


def build_models(seed=1234, device="cpu", dtype=torch.float32):
    g = torch.random.fork_rng(devices=[])
    with g:
        torch.manual_seed(seed)
        sdf = SDFNetwork()
        col = RenderingNetwork()
        dev = SingleVarianceNetwork(0.3)
    return sdf.to(device=device, dtype=dtype), col.to(device=device, dtype=dtype), dev.to(device=device, dtype=dtype)


def train_step(renderer: NeuSRenderer, optimizer, rays, cos_anneal, igr_weight=0.1, mask_weight=0.1,
               normal_weight=0.0, R=None, t_rand=None, corr_weight=0.0, corr=None, R_all=None, T_all=None, K=None,
               corr_delta_px=4.0):
    """One full training iteration (render -> losses -> backward -> Adam); the cpu_baseline unit of work."""
    rays_o, rays_d, true_rgb = rays[:, :3], rays[:, 3:6], rays[:, 6:9]
    obj, keep, mono = rays[:, 9:10], rays[:, 10:11], rays[:, 11:14]
    near, far = near_far_from_sphere(rays_o, rays_d)
    out = renderer.render(rays_o, rays_d, near, far, cos_anneal_ratio=cos_anneal, t_rand=t_rand)
    losses = neus_losses(out, true_rgb, obj, keep, igr_weight, mask_weight, normal_weight, mono, R)
    if corr_weight > 0.0 and corr is not None:
        cl = correspondence_loss(out["weights"], out["z_vals"], 2.0 / renderer.n_samples, rays_o, rays_d, corr, R_all, T_all, K,
                                 corr_delta_px)
        losses["corr_loss"] = cl["loss"]
        losses["loss"] = losses["loss"] + corr_weight * cl["loss"]
    optimizer.zero_grad()
    losses["loss"].backward()
    optimizer.step()
    return losses
