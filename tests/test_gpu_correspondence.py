"""GPU: the dense-correspondence reprojection term (BASELINE.json configs[4] "DKM correspondence"; specification =
oracle/neus_oracle.py:correspondence_loss, closed-form checked in tests/test_cpu_oracle_correspondence.py) through the C ABI:
dh_corr_loss vs the oracle (loss, residuals, adjoint w.r.t. the compositing weights), the fused training step with the term
switched on vs the oracle's autograd, and the data side (synthetic matches, on-disk folder, outlier voting)."""
import os

import pytest
import torch

from oracle import neus_oracle as O
from tests.test_gpu_render_forward import make_pair

pytestmark = pytest.mark.gpu


def _random_case(B, n, seed, F=5):
    g = torch.Generator(device="cpu").manual_seed(seed)
    o = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1) * 2.2
    d = torch.nn.functional.normalize(-o + 0.2 * torch.randn(B, 3, generator=g), dim=-1)
    z = (1.4 + torch.sort(torch.rand(B, n, generator=g), dim=-1).values * 1.6)
    w = torch.rand(B, n, generator=g) ** 4
    w = w / w.sum(-1, keepdim=True) * (0.5 + 0.5 * torch.rand(B, 1, generator=g))
    R_all = torch.linalg.qr(torch.randn(F, 3, 3, generator=g))[0]
    T_all = torch.randn(F, 3, generator=g) * 0.2 + torch.tensor([0.0, 0.0, 2.3])
    K = torch.tensor([[614.4, 0, 256.0], [0, 614.4, 256.0], [0, 0, 1]])
    corr = torch.stack([torch.rand(B, generator=g) * 511, torch.rand(B, generator=g) * 511, torch.rand(B, generator=g),
                        torch.randint(0, F, (B,), generator=g).float()], -1)
    corr[::7, 2] = 0.0                                   # rays without a match
    # make a third of the matches nearly consistent (quadratic Huber branch): partner pixel = true reprojection + noise
    x = o + (w * z).sum(-1, keepdim=True) * d
    j = corr[:, 3].long()
    y = torch.einsum("bij,bj->bi", R_all[j], x) + T_all[j]
    good = (torch.arange(B) % 3 == 0) & (y[:, 2] > 0.1)
    corr[good, 0] = (K[0, 0] * y[good, 0] / y[good, 2] + K[0, 2]) + torch.randn(int(good.sum()), generator=g) * 1.0
    corr[good, 1] = (K[1, 1] * y[good, 1] / y[good, 2] + K[1, 2]) + torch.randn(int(good.sum()), generator=g) * 1.0
    T_all[F - 1, 2] = -3.0                               # frame F-1 looks away: every point is behind it
    return [t.cuda().contiguous() for t in (o, d, z, w, corr, R_all, T_all, K)]


@pytest.mark.parametrize("B,n", [(300, 64), (2048, 128), (5, 33)])
def test_corr_loss_kernel_matches_oracle(B, n):
    from dynhor_amd import _lib
    from dynhor_amd.renderer import _p
    o, d, z, w, corr, R_all, T_all, K = _random_case(B, n, seed=B + n)
    sd, delta, cw = 2.0 / 64, 4.0, 0.3
    stats = torch.empty(4, device="cuda"); res = torch.empty(B, device="cuda"); dw = torch.empty(B, n, device="cuda")
    padj = torch.empty(B, 7, device="cuda")
    _lib.check(_lib.lib().dh_corr_loss(_p(o), _p(d), _p(z), _p(w), _p(corr), _p(R_all), _p(T_all), R_all.shape[0], _p(K), B, n,
                                      sd, delta, cw, _p(stats), _p(res), _p(dw), _p(padj), _lib.stream()))
    dw0 = torch.empty(B, n, device="cuda"); st0 = torch.empty(4, device="cuda"); rs0 = torch.empty(B, device="cuda")
    _lib.check(_lib.lib().dh_corr_loss(_p(o), _p(d), _p(z), _p(w), _p(corr), _p(R_all), _p(T_all), R_all.shape[0], _p(K), B, n,
                                      sd, delta, cw, _p(st0), _p(rs0), _p(dw0), None, _lib.stream()))
    assert torch.equal(dw0, dw) and torch.equal(st0, stats), "the optional pose adjoints change nothing else"
    w64 = w.double().requires_grad_(True)
    o64, d64 = o.double().requires_grad_(True), d.double().requires_grad_(True)
    R64, T64 = R_all.double().requires_grad_(True), T_all.double().requires_grad_(True)
    ref = O.correspondence_loss(w64, z.double(), sd, o64, d64, corr.double(), R64, T64, K.double(), delta)
    (cw * ref["loss"]).backward()
    c = corr[:, 2].double() * ref["valid"]
    assert c.sum().item() > 0 and (ref["valid"] == 0).any(), "the case must contain valid matches and points behind the partner camera"
    s = ref["residual_px"] / K[0, 0].double()
    assert ((s <= delta / K[0, 0].double()) & (c > 0)).any() and ((s > delta / K[0, 0].double()) & (c > 0)).any(), "both Huber branches"
    assert abs(stats[0].item() - ref["loss"].item()) < 2e-5 * max(1.0, ref["loss"].item())
    assert abs(stats[1].item() - c.sum().item()) < 1e-4 * c.sum().item()
    assert abs(stats[3].item() - cw * ref["loss"].item()) < 2e-5
    live = (c > 0)
    assert (res.double() - ref["residual_px"] * live)[live].abs().max().item() < 2e-2, "residuals in pixels (fp32 projection)"
    nomatch = corr[:, 2] == 0
    assert (res[nomatch] == 0).all(), "rays without a match report 0"
    behind = (~live) & (~nomatch)
    assert behind.any() and torch.isinf(res[behind]).all(), "a match that projects behind the partner camera is +inf (an outlier), not 0"
    gref = w64.grad
    rel = ((dw.double() - gref).norm() / gref.norm()).item()
    print(f"corr loss {stats[0].item():.6f} (oracle {ref['loss'].item():.6f}); d_weights rel err {rel:.2e}")
    assert rel < 2e-4
    assert (dw[corr[:, 2] == 0] == 0).all(), "rays without a match get an exactly-zero adjoint"
    # pose adjoints (refine_poses + the correspondence term): d / d x, d / d y and t^ per ray
    # vs the oracle differentiated w.r.t. the rays (at fixed weights) and w.r.t. every frame's pose
    d_x, d_y, t_hat = padj[:, :3].double(), padj[:, 3:6].double(), padj[:, 6:7].double()
    assert ((d_x - o64.grad).norm() / o64.grad.norm()).item() < 2e-4
    assert ((t_hat * d_x - d64.grad).norm() / d64.grad.norm()).item() < 2e-4
    j = corr[:, 3].long()
    x = o.double() + t_hat * d.double()
    dR = torch.zeros_like(R64).index_add_(0, j, d_y[:, :, None] * x[:, None, :])
    dT = torch.zeros_like(T64).index_add_(0, j, d_y)
    relR, relT = ((dR - R64.grad).norm() / R64.grad.norm()).item(), ((dT - T64.grad).norm() / T64.grad.norm()).item()
    print(f"partner-pose adjoints: d_R rel {relR:.2e}, d_T rel {relT:.2e}")
    assert relR < 2e-4 and relT < 2e-4
    assert (padj[~live][:, :6] == 0).all()


def test_fused_train_step_with_correspondence_term_matches_oracle():
    from dynhor_amd.dataset import Dataset
    ds = Dataset.from_synthetic(n_frames=24, H=96, W=96, seed=7, device="cuda:0", correspondences=256)
    assert ds.corr is not None and ds.corr.shape[0] > 500
    o_r, p_r = make_pair(seed=41, jitter=0.05, n_samples=32, n_importance=32)
    B, frame, car, cw = 128, 3, 0.4, 0.5
    g = torch.Generator(device="cuda:0"); g.manual_seed(5)
    rays, corr, midx = ds.gen_corr_rays_at(frame, B, 48, generator=g)
    assert (midx[:80] == -1).all() and (midx[80:] >= 0).all() and (corr[:80, 2] == 0).all() and (corr[80:, 2] > 0).all()
    near, far = ds._last_near_far
    t_rand = torch.rand(B, 1, device="cuda:0", generator=g)
    R_all, T_all, K = ds.corr_frames()
    z = o_r.sample_z(rays[:, :3], rays[:, 3:6], near, far, t_rand=t_rand)
    mods = (o_r.sdf_network, o_r.deviation_network, o_r.color_network)
    for m in mods:
        m.double(); m.zero_grad()
    r64 = rays.double()
    out = o_r.render(r64[:, :3], r64[:, 3:6], near.double(), far.double(), cos_anneal_ratio=car, z_vals=z.double())
    ref = O.neus_losses(out, r64[:, 6:9], r64[:, 9:10], r64[:, 10:11], 0.1, 0.1, 0.05, r64[:, 11:14], ds.R[frame].double())
    cl = O.correspondence_loss(out["weights"], out["z_vals"], 2.0 / 32, r64[:, :3], r64[:, 3:6], corr.double(), R_all.double(),
                               T_all.double(), K.double(), 4.0)
    total = ref["loss"] + cw * cl["loss"]
    total.backward()
    gref = torch.cat([p.grad.reshape(-1) for m in mods for p in m.parameters()])
    for m in mods:
        m.float()
    p_r.sample_z = lambda *a, **k: z
    stats = p_r.train_step_core(rays, near, far, ds.R[frame], car, 0.1, 0.1, 0.05, corr=corr, corr_weight=cw,
                                corr_frames=ds.corr_frames(), corr_delta_px=4.0)
    torch.cuda.synchronize()
    assert abs(stats[0].item() - total.item()) < 2e-5 * max(1.0, abs(total.item()))
    assert abs(p_r.last_corr_stats[0].item() - cl["loss"].item()) < 2e-5 * max(1.0, cl["loss"].item())
    rel = ((p_r.store.grad_flat.double() - gref).norm() / gref.norm()).item()
    print(f"train step with correspondence term: loss {stats[0].item():.6f} (oracle {total.item():.6f}), corr {cl['loss'].item():.5f}, grad rel err {rel:.2e}")
    assert rel < 1e-4
    # the term really contributes: the gradient differs from the one without it
    p_r.train_step_core(rays, near, far, ds.R[frame], car, 0.1, 0.1, 0.05)
    assert ((p_r.store.grad_flat.double() - gref).norm() / gref.norm()).item() > 1e-3


def test_synthetic_matches_disk_roundtrip_and_outlier_voting(tmp_path):
    from dynhor_amd.dataset import Dataset
    from dynhor_amd.scene import make_correspondences, make_sequence, write_correspondences_to_disk, write_sequence_to_disk
    frames = make_sequence(n_frames=16, H=96, W=96, seed=3, device="cpu")
    matches = make_correspondences(frames, n_per_pair=200, offsets=(1, 2), seed=9)
    assert len(matches) >= 16 and sum(m["kpts0"].shape[0] for m in matches) > 1000
    root = str(tmp_path / "seq")
    write_sequence_to_disk(frames, root)
    write_correspondences_to_disk(matches, root)
    assert len(os.listdir(os.path.join(root, "correspondence_infos"))) == len(matches)       # reference folder name, README.md:43
    ds = Dataset({"dataroot": root}, device="cuda:0")
    frames["matches"] = matches
    ds_mem = Dataset(frames=frames, device="cuda:0")
    assert torch.equal(ds.corr, ds_mem.corr) and torch.equal(ds.corr_pair, ds_mem.corr_pair)
    # true-geometry residuals: inliers reproject to within the 0.5 px matching noise, planted outliers do not
    planted = torch.cat([m["is_outlier"] for m in sorted(matches, key=lambda m: (m["i"], m["j"]))]).cuda()
    from dynhor_amd.scene import scene_sdf
    res = torch.empty(ds.corr.shape[0], device="cuda:0")
    for f, (lo, hi) in ds._corr_range.items():
        m = ds.corr[lo:hi]
        rays = ds.gen_rays_at_pixels(f, m[:, 0].long(), m[:, 1].long())
        o, d = rays[:, :3], rays[:, 3:6]
        t = (-(o * d).sum(-1) - 0.75).clone()
        for _ in range(64):
            sdf = scene_sdf(o + d * t[:, None])
            t = torch.where(sdf < 5e-4, t, t + sdf.clamp(min=1e-4)).clamp(max=6.0)
        x = o + d * t[:, None]
        j = m[:, 5].long()
        y = torch.einsum("bij,bj->bi", ds.R[j], x) + ds.T[j]
        pu = ds.K[0, 0] * y[:, 0] / y[:, 2] + ds.K[0, 2]; pv = ds.K[1, 1] * y[:, 1] / y[:, 2] + ds.K[1, 2]
        res[lo:hi] = ((pu - m[:, 2]) ** 2 + (pv - m[:, 3]) ** 2).sqrt()
    assert res[~planted].median().item() < 1.5 and (res[planted] > 8.0).float().mean().item() > 0.9
    info = ds.vote_correspondences(res, tau_px=8.0)
    kept = ds.corr[:, 4] > 0
    assert (kept & planted).float().sum().item() <= 0.1 * planted.sum().item(), "gross outliers are voted out"
    assert (kept & ~planted).float().sum().item() >= 0.97 * (~planted).sum().item(), "inliers survive"
    assert info["pairs_dropped"] == 0
    # a pair whose matches are all wrong (bad pose / matcher failure) is dropped as a whole, and voting is not sticky
    bad_pair = int(ds.corr_pair[0])
    res2 = res.clone(); res2[ds.corr_pair == bad_pair] = 50.0
    info2 = ds.vote_correspondences(res2, tau_px=8.0)
    assert info2["pairs_dropped"] == 1 and (ds.corr[ds.corr_pair == bad_pair, 4] == 0).all()
    ds.vote_correspondences(res, tau_px=8.0)
    assert (ds.corr[(ds.corr_pair == bad_pair) & ~planted, 4] > 0).float().mean().item() > 0.9


def test_runner_trains_with_the_full_loss_stack(tmp_path):
    from dynhor_amd.runner import Runner
    conf = {"seq_name": "full", "exp_name": "e",
            "data_info": {"synthetic": {"n_frames": 16, "H": 96, "W": 96, "seed": 3, "correspondences": 256}},
            "train": {"batch_size": 512, "normal_weight": 0.05, "corr_weight": 0.1, "corr_fraction": 0.25, "corr_vote_freq": 20,
                      "report_freq": 10, "save_freq": 10 ** 9, "val_freq": 0, "warm_up_end": 20, "end_iter": 1000}}
    r = Runner(conf=conf, device="cuda:0", exp_root=str(tmp_path))
    first = None
    for _ in range(45):
        s = r.train_iteration()
        first = first if first is not None else float(s[0])
    assert float(s[0]) < first and torch.isfinite(r.store.flat).all()
    cs = r.renderer.last_corr_stats
    assert cs[1].item() > 0 and torch.isfinite(cs).all()
    assert hasattr(r, "last_vote") and r.last_vote["matches"] >= 0
    rec = r.report(s)
    assert "Loss/corr_loss" in rec and "Statistics/corr_residual_px" in rec
