"""CPU: the dense-correspondence reprojection loss of the oracle (oracle/neus_oracle.py:correspondence_loss, the
specification of BASELINE.json configs[4]'s "DKM correspondence" term -- the reference releases neither format nor loss,
README.md:43) against closed forms on an analytic sphere seen by two cameras, plus the outlier-voting rule."""
import math

import torch

from oracle import neus_oracle as O


def _look_at(pos):
    pos = torch.tensor(pos, dtype=torch.float64)
    up = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64)
    zf = -pos / pos.norm()
    xr = torch.linalg.cross(zf, up); xr = xr / xr.norm()
    yd = torch.linalg.cross(zf, xr)
    R = torch.stack([xr, yd, zf])
    return R, -(R @ pos)


def _setup():
    H = W = 128
    f = 1.2 * min(H, W)                                   # reference intrinsics, ObjTracker/run.py:119-123
    K = torch.tensor([[f, 0, W // 2], [0, f, H // 2], [0, 0, 1]], dtype=torch.float64)
    Ri, Ti = _look_at([2.2, 0.3, 0.4])
    Rj, Tj = _look_at([1.6, 1.5, -0.2])
    R_all, T_all = torch.stack([Ri, Rj]), torch.stack([Ti, Tj])
    px = torch.tensor([64.0, 70.0, 58.0, 66.0], dtype=torch.float64)
    py = torch.tensor([64.0, 60.0, 69.0, 72.0], dtype=torch.float64)
    rays_o, rays_d = O.rays_from_pixels(px, py, K, Ri, Ti)
    r = 0.4                                               # analytic sphere |x| = r: first intersection depth
    b = (rays_o * rays_d).sum(-1)
    t_star = -b - torch.sqrt(b * b - (rays_o * rays_o).sum(-1) + r * r)
    x = rays_o + t_star[:, None] * rays_d
    assert torch.allclose(x.norm(dim=-1), torch.full_like(t_star, r))
    y = x @ Rj.T + Tj
    q = torch.stack([f * y[:, 0] / y[:, 2] + K[0, 2], f * y[:, 1] / y[:, 2] + K[1, 2]], -1)
    return K, R_all, T_all, rays_o, rays_d, t_star, q, f


def _one_hot_render(t_star, n=8, sample_dist=0.05, shift=0.0):
    """weights = one-hot on sample 3 whose MID-point depth is t_star + shift (z spacing d => mid = z + d/2)."""
    B = t_star.shape[0]
    d = 0.02
    z = (t_star + shift)[:, None] - 0.5 * d + (torch.arange(n, dtype=torch.float64)[None, :] - 3) * d
    w = torch.zeros(B, n, dtype=torch.float64); w[:, 3] = 1.0
    return w, z, sample_dist


def test_true_surface_point_reprojects_onto_its_match():
    K, R_all, T_all, o, d, t_star, q, f = _setup()
    w, z, sd = _one_hot_render(t_star)
    corr = torch.cat([q, torch.ones(4, 1, dtype=torch.float64), torch.ones(4, 1, dtype=torch.float64)], -1)
    out = O.correspondence_loss(w, z, sd, o, d, corr, R_all, T_all, K, delta_px=4.0)
    assert torch.allclose(out["depth"], t_star, atol=1e-12)
    assert out["residual_px"].max().item() < 1e-8 and out["loss"].item() < 1e-12


def test_depth_error_gives_the_analytic_reprojection_residual_and_huber_branches():
    K, R_all, T_all, o, d, t_star, q, f = _setup()
    corr = torch.cat([q, torch.ones(4, 1, dtype=torch.float64), torch.ones(4, 1, dtype=torch.float64)], -1)
    for shift, delta_px in ((0.004, 4.0), (0.2, 4.0)):                    # quadratic branch, linear branch
        w, z, sd = _one_hot_render(t_star, shift=shift)
        out = O.correspondence_loss(w, z, sd, o, d, corr, R_all, T_all, K, delta_px=delta_px)
        x = o + (t_star + shift)[:, None] * d
        y = x @ R_all[1].T + T_all[1]
        pi = torch.stack([f * y[:, 0] / y[:, 2] + K[0, 2], f * y[:, 1] / y[:, 2] + K[1, 2]], -1)
        res_px = (pi - q).norm(dim=-1)
        assert torch.allclose(out["residual_px"], res_px, atol=1e-9)
        s, dl = res_px / f, delta_px / f
        rho = torch.where(s <= dl, s * s / (2 * dl), s - 0.5 * dl)
        assert (res_px <= delta_px).all() if shift < 0.01 else (res_px > delta_px).all()
        assert abs(out["loss"].item() - (rho.sum() / (4 + 1e-5)).item()) < 1e-12


def test_certainty_weights_gradient_and_points_behind_the_partner_camera():
    K, R_all, T_all, o, d, t_star, q, f = _setup()
    w, z, sd = _one_hot_render(t_star, shift=0.01)
    w = (w * 0.7 + 0.3 / w.shape[1]).requires_grad_(True)               # spread weights: every sample gets a gradient
    conf = torch.tensor([1.0, 0.5, 0.0, 0.25], dtype=torch.float64)
    corr = torch.cat([q, conf[:, None], torch.ones(4, 1, dtype=torch.float64)], -1)
    out = O.correspondence_loss(w, z, sd, o, d, corr, R_all, T_all, K)
    out["loss"].backward()
    g = w.grad.clone()
    assert g[2].abs().max().item() == 0.0, "certainty 0 = ray without a match: no gradient"
    # d loss / d w_k = (d loss / d depth) * mid_k : rows are proportional to the mid-point depths
    mid = z + 0.5 * torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], sd)], -1)
    for r in (0, 1, 3):
        ratio = g[r] / mid[r]
        assert torch.allclose(ratio, ratio[0].expand_as(ratio), rtol=1e-9)
    # finite differences of the scalar loss w.r.t. one weight
    eps = 1e-6
    for (r, k) in ((0, 2), (3, 5)):
        wp = w.detach().clone(); wp[r, k] += eps
        wm = w.detach().clone(); wm[r, k] -= eps
        fd = (O.correspondence_loss(wp, z, sd, o, d, corr, R_all, T_all, K)["loss"]
              - O.correspondence_loss(wm, z, sd, o, d, corr, R_all, T_all, K)["loss"]) / (2 * eps)
        assert abs(fd.item() - g[r, k].item()) < 1e-6 * max(1.0, abs(g[r, k].item()))
    # a surface estimate behind camera j contributes nothing (and no NaN)
    T_far = T_all.clone(); T_far[1, 2] = -5.0
    out2 = O.correspondence_loss(w.detach(), z, sd, o, d, corr, R_all, T_far, K)
    assert out2["valid"].sum().item() == 0 and out2["loss"].item() == 0.0 and torch.isfinite(out2["residual_px"]).all()


def test_outlier_voting_drops_bad_matches_and_bad_pairs():
    n = 20
    corr = torch.zeros(2 * n + 4, 4); corr[:2 * n, 2] = 1.0           # last 4 rays carry no match
    pair = torch.cat([torch.zeros(n), torch.ones(n), torch.full((4,), 7.0)]).long()
    res = torch.full((2 * n + 4,), 1.0)
    res[:6] = 30.0                                                     # pair 0: 30 % outliers -> kept, outliers dropped
    res[n:n + 14] = 30.0                                               # pair 1: 70 % outliers -> the whole pair is voted out
    new = O.vote_correspondences(res, corr, pair, tau_px=8.0, min_pair_inlier_ratio=0.5)
    assert new[:6].sum().item() == 0 and (new[6:n] == 1).all()
    assert new[n:2 * n].sum().item() == 0
    assert new[2 * n:].sum().item() == 0
