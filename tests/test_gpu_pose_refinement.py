"""GPU: per-frame pose refinement.  d loss / d rays_o, d loss / d rays_d and the normal loss's direct
d loss / d R from the HIP path (dh_color_backward_rays, dh_sdf_backward_rays incl. the second-order path,
dh_render_scan_bwd_rays) against the oracle's autograd in fp64 with the rays as leaves (sample depths constant in both);
then the full chain into the reference's pose parameters (6-D rotation + translation, ObjTracker/utils/geometry.py:7-25)
against the oracle differentiated end to end w.r.t. those parameters; finally the Runner wiring."""
import pytest
import torch

from oracle import neus_oracle as O
from tests.test_gpu_render_forward import make_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny_dataset():
    from dynhor_amd.dataset import Dataset
    return Dataset.from_synthetic(n_frames=3, H=96, W=96, seed=7, device="cuda:0")


def _oracle_loss(o_r, o, d, near, far, z, rays, R, car, nw):
    out = o_r.render(o, d, near, far, cos_anneal_ratio=car, z_vals=z)
    return O.neus_losses(out, rays[:, 6:9], rays[:, 9:10], rays[:, 10:11], 0.1, 0.1, nw, rays[:, 11:14], R)["loss"]


@pytest.mark.parametrize("normal_w", [0.0, 0.05])
def test_ray_gradients_match_oracle(tiny_dataset, normal_w):
    ds = tiny_dataset
    o_r, p_r = make_pair(seed=51, jitter=0.05, n_samples=32, n_importance=32)
    B, frame, car = 96, 1, 0.4
    g = torch.Generator(device="cpu").manual_seed(5)
    px = torch.randint(0, ds.W, [B], generator=g).cuda(); py = torch.randint(0, ds.H, [B], generator=g).cuda()
    t_rand = torch.rand(B, 1, generator=g).cuda()
    rays = ds.gen_rays_at_pixels(frame, px, py)
    near, far = ds._last_near_far
    z = o_r.sample_z(rays[:, :3], rays[:, 3:6], near, far, t_rand=t_rand)
    mods = (o_r.sdf_network, o_r.deviation_network, o_r.color_network)
    for m in mods:
        m.double(); m.zero_grad()
    r64 = rays.double()
    o = r64[:, :3].clone().requires_grad_(True); d = r64[:, 3:6].clone().requires_grad_(True)
    R = ds.R[frame].double().clone().requires_grad_(True)
    # the sample points themselves (the tensor render_core hands to the SDF network, the gradient pass and the colour network): their
    # total adjoint is what the HIP path leaves PER POINT in s.d_pts
    seen = {"on": False}
    def grab(mod, inp):
        if "pts" not in seen:
            seen["pts"] = inp[0]
            # (a tensor hook, gated: retain_grad would also collect the gradient of the inner autograd.grad call that forms the normals)
            inp[0].register_hook(lambda gr: seen.__setitem__("grad", gr.detach().clone()) if seen["on"] else None)
    hook = o_r.sdf_network.register_forward_pre_hook(grab)
    loss64 = _oracle_loss(o_r, o, d, near.double(), far.double(), z.double(), r64, R, car, normal_w)
    hook.remove()
    seen["on"] = True
    loss64.backward()
    pts_ref = seen["grad"]
    gref = torch.cat([p.grad.reshape(-1) for m in mods for p in m.parameters()])
    for m in mods:
        m.float()
    p_r.sample_z = lambda *a, **k: z
    p_r.train_step_core(rays, near, far, ds.R[frame], car, 0.1, 0.1, normal_w, ray_grads=True)
    torch.cuda.synchronize()
    d_o, d_d, d_R = p_r.last_ray_grads
    rel_w = ((p_r.store.grad_flat.double() - gref).norm() / gref.norm()).item()
    rel_o = ((d_o.double() - o.grad).norm() / o.grad.norm()).item()
    rel_d = ((d_d.double() - d.grad).norm() / d.grad.norm()).item()
    print(f"normal_w {normal_w}: weight grad rel {rel_w:.2e}; d_rays_o rel {rel_o:.2e}; d_rays_d rel {rel_d:.2e}")
    assert rel_w < 1e-4, "the pose-refinement kernel variants must leave the weight gradient unchanged"
    assert rel_o < 5e-4 and rel_d < 5e-4
    # per-ray, not only in norm: the largest per-ray deviation against the largest per-ray gradient
    assert (d_o.double() - o.grad).abs().max().item() < 2e-3 * o.grad.abs().max().item()
    assert (d_d.double() - d.grad).abs().max().item() < 2e-3 * d.grad.abs().max().item()
    # PER POINT: the adjoint tiles of the two-piece fp16 chains are scaled per 64-point tile, so "fp32 accuracy" is a
    # statement relative to the tile.  Measured here against fp64, not ray-summed: the absolute error of every point relative to its
    # tile's largest adjoint, the relative error of the points within 10x of it -- and the SAME numbers for the native fp32-MFMA kernels,
    # which have no tile scale: what the split adds is the difference.
    from dynhor_amd import _lib

    def per_point(dp):
        P = dp.shape[0]
        nt = (P + 63) // 64
        pad = nt * 64 - P
        mag = torch.nn.functional.pad(pts_ref.abs().amax(dim=1), (0, pad)).view(nt, 64)
        err = torch.nn.functional.pad((dp - pts_ref).abs().amax(dim=1), (0, pad)).view(nt, 64)
        tmax = mag.amax(dim=1, keepdim=True).clamp_min(1e-300)
        big = mag >= 0.1 * tmax
        small = (mag < 1e-3 * tmax) & (mag > 0)
        return (err / tmax).max().item(), float((err[big] / mag[big]).max()), int(small.sum()), float((err / tmax)[small].max()) if bool(small.any()) else 0.0

    assert p_r.last_state.d_pts.shape == pts_ref.shape
    res = {"split_f16": per_point(p_r.last_state.d_pts.double())}
    p_r.arithmetic = _lib.ARITH_FP32_MFMA
    p_r.train_step_core(rays, near, far, ds.R[frame], car, 0.1, 0.1, normal_w, ray_grads=True)
    torch.cuda.synchronize()
    res["fp32_mfma"] = per_point(p_r.last_state.d_pts.double())
    for k, (a_t, r_big, n_small, a_small) in res.items():
        print(f"per-point d_pts vs fp64 [{k}]: max |err| / tile max = {a_t:.2e}; points within 10x of their tile max: max rel {r_big:.2e}; "
              f"{n_small} points more than 1e-3 below their tile max: max |err| / tile max = {a_small:.2e}")
    # the error of a point is bounded relative to its TILE's largest adjoint; the two arithmetics agree on that bound to within a factor
    assert res["split_f16"][0] < 2e-4 and res["split_f16"][1] < 1e-3
    assert res["split_f16"][0] < 4.0 * res["fp32_mfma"][0] + 1e-6, "the two-piece split must not add to the exact-fp32 kernels' per-point error"
    if normal_w > 0:
        rel_R = ((d_R.double() - R.grad).norm() / R.grad.norm()).item()
        print(f"d loss / d R (normal loss, direct) rel {rel_R:.2e}")
        assert rel_R < 5e-4
    else:
        assert d_R is None


def test_pose_parameter_gradients_match_oracle_end_to_end(tiny_dataset):
    from dynhor_amd.pose import PoseRefiner, rot6d_to_matrix
    ds = tiny_dataset
    o_r, p_r = make_pair(seed=52, jitter=0.05, n_samples=32, n_importance=32)
    B, frame, car, nw = 128, 2, 0.3, 0.05
    g = torch.Generator(device="cpu").manual_seed(9)
    px = torch.randint(0, ds.W, [B], generator=g).cuda(); py = torch.randint(0, ds.H, [B], generator=g).cuda()
    rays = ds.gen_rays_at_pixels(frame, px, py)
    near, far = ds._last_near_far
    z = o_r.sample_z(rays[:, :3], rays[:, 3:6], near, far, t_rand=torch.rand(B, 1, generator=g).cuda())
    ref = PoseRefiner(ds.R, ds.T).cuda()
    # the refiner reproduces the stage-1 poses and the HIP ray gather
    Rn, Tn = ref.poses()
    assert (Rn - ds.R).abs().max().item() < 1e-6 and torch.equal(Tn.detach(), ds.T)
    o_t, d_t, R_t = ref.rays(frame, px, py, ds.Kinv)
    assert (o_t - rays[:, :3]).abs().max().item() < 1e-5 and (d_t - rays[:, 3:6]).abs().max().item() < 1e-5
    # oracle, fp64, differentiated w.r.t. the pose parameters themselves
    for m in (o_r.sdf_network, o_r.deviation_network, o_r.color_network):
        m.double()
    r6 = ref.rot6d.detach().double().clone().requires_grad_(True)
    tr = ref.trans.detach().double().clone().requires_grad_(True)
    R64 = rot6d_to_matrix(r6[frame:frame + 1])[0].T
    pix = torch.stack([px.double(), py.double(), torch.ones(B, dtype=torch.float64, device="cuda")], -1)
    dc = torch.nn.functional.normalize(pix @ torch.inverse(ds.K.double()).T, dim=-1)
    d64 = dc @ R64; o64 = (-(tr[frame] @ R64)).expand_as(d64)
    _oracle_loss(o_r, o64, d64, near.double(), far.double(), z.double(), rays.double(), R64, car, nw).backward()
    for m in (o_r.sdf_network, o_r.deviation_network, o_r.color_network):
        m.float()
    p_r.sample_z = lambda *a, **k: z
    p_r.train_step_core(rays, near, far, ds.R[frame], car, 0.1, 0.1, nw, ray_grads=True)
    d_o, d_d, d_R = p_r.last_ray_grads
    ref.opt.zero_grad()
    torch.autograd.backward([o_t, d_t, R_t], [d_o, d_d, d_R])
    rel_r = ((ref.rot6d.grad[frame].double() - r6.grad[frame]).norm() / r6.grad[frame].norm()).item()
    rel_t = ((ref.trans.grad[frame].double() - tr.grad[frame]).norm() / tr.grad[frame].norm()).item()
    print(f"pose gradients vs oracle end to end: rot6d rel {rel_r:.2e}, translation rel {rel_t:.2e}")
    assert rel_r < 2e-3 and rel_t < 2e-3
    others = [f for f in range(ds.n_images) if f != frame]
    assert ref.rot6d.grad[others].abs().max().item() == 0 and ref.trans.grad[others].abs().max().item() == 0


def test_all_arithmetics_give_the_same_ray_adjoints(tiny_dataset):
    """the native fp32-MFMA twin set covers the pose-refinement kernels too (dh_color_backward_rays /
    dh_sdf_backward_rays): the independent, exact-fp32 arithmetic cross-checks d loss / d rays of the shipping two-piece fp16
    kernels and of the three-piece bf16 ones on the same step."""
    from dynhor_amd import _lib
    ds = tiny_dataset
    _, p_r = make_pair(seed=53, jitter=0.05, n_samples=32, n_importance=32)
    g = torch.Generator(device="cuda:0"); g.manual_seed(2)
    rays = ds.gen_random_rays_at(0, 192, generator=g)
    near, far = ds._last_near_far
    t_rand = torch.rand(192, 1, device="cuda:0", generator=g)
    res = {}
    z = None
    try:
        for mode in (_lib.ARITH_FP32_MFMA, _lib.ARITH_SPLIT_F16, _lib.ARITH_SPLIT_BF16):
            p_r.arithmetic = mode                     # passed with every launch (the `_ex` entry points)
            if z is None:
                z = p_r.sample_z(rays[:, :3].contiguous(), rays[:, 3:6].contiguous(), near, far, t_rand=t_rand)
                p_r.sample_z = lambda *a, **k: z
            stats = p_r.train_step_core(rays, near, far, ds.R[0], 0.3, 0.1, 0.1, 0.05, ray_grads=True)
            torch.cuda.synchronize()
            d_o, d_d, d_R = p_r.last_ray_grads
            res[mode] = (stats.clone(), p_r.store.grad_flat.clone(), d_o.clone(), d_d.clone(), d_R.clone())
    finally:
        p_r.arithmetic = None
    b = res[_lib.ARITH_FP32_MFMA]
    rel = lambda x, y: ((x.double() - y.double()).norm() / y.double().norm()).item()
    for name, mode in (("split-f16", _lib.ARITH_SPLIT_F16), ("split-bf16", _lib.ARITH_SPLIT_BF16)):
        a = res[mode]
        print(f"{name} vs fp32-MFMA, pose refinement step: loss diff {(a[0][0] - b[0][0]).abs().item():.2e}; weight grad {rel(a[1], b[1]):.2e}; "
              f"d_rays_o {rel(a[2], b[2]):.2e}; d_rays_d {rel(a[3], b[3]):.2e}; d_R {rel(a[4], b[4]):.2e}")
        assert (a[0][:6] - b[0][:6]).abs().max().item() < 5e-6
        assert rel(a[1], b[1]) < 2e-5 and rel(a[2], b[2]) < 5e-5 and rel(a[3], b[3]) < 5e-5 and rel(a[4], b[4]) < 5e-5


def test_runner_refines_poses_and_checkpoints_them(tmp_path):
    from dynhor_amd.runner import Runner
    conf = {"seq_name": "pose", "exp_name": "e", "data_info": {"synthetic": {"n_frames": 4, "H": 64, "W": 64, "seed": 5}},
            "train": {"batch_size": 256, "normal_weight": 0.05, "refine_poses": True, "pose_lr": 1e-3, "report_freq": 10 ** 9,
                      "save_freq": 10 ** 9, "val_freq": 0, "warm_up_end": 10, "end_iter": 100}}
    r = Runner(conf=conf, device="cuda:0", exp_root=str(tmp_path))
    R0, T0 = r.dataset.R.clone(), r.dataset.T.clone()
    r.train(8)
    assert torch.isfinite(r.dataset.R).all() and torch.isfinite(r.store.flat).all()
    moved = (r.dataset.T - T0).abs().amax(dim=1)
    assert (moved > 0).all() and moved.max().item() < 0.05, "every visited frame's pose moved, by a small step"
    assert ((r.dataset.R @ r.dataset.R.transpose(1, 2)) - torch.eye(3, device="cuda")).abs().max().item() < 1e-5
    path = r.save_checkpoint()
    ck = torch.load(path, weights_only=False)
    assert "pose_refiner" in ck
    r2 = Runner(conf=conf, device="cuda:0", exp_root=str(tmp_path), is_continue=True)
    assert torch.allclose(r2.dataset.R, r.dataset.R, atol=1e-6) and torch.allclose(r2.dataset.T, r.dataset.T)


def test_pose_gradients_with_the_correspondence_term_match_oracle():
    """refine_poses together with corr_weight > 0.  The correspondence term depends on the rays directly
    (x = o + t^ d) and on the partner frames' poses (y = R_j x + T_j); the oracle differentiated w.r.t. the rays and EVERY frame's
    pose must agree with last_ray_grads / last_partner_pose_grads, and the chained 6-D rotation / translation gradients too."""
    from dynhor_amd.dataset import Dataset
    from dynhor_amd.pose import PoseRefiner
    ds = Dataset.from_synthetic(n_frames=12, H=96, W=96, seed=7, device="cuda:0", correspondences=256)
    o_r, p_r = make_pair(seed=53, jitter=0.05, n_samples=32, n_importance=32)
    B, frame, car, cw, nw = 128, 3, 0.4, 0.5, 0.05
    g = torch.Generator(device="cuda:0"); g.manual_seed(5)
    rays, corr, _ = ds.gen_corr_rays_at(frame, B, 64, generator=g)
    near, far = ds._last_near_far
    px, py = ds._last_pixels
    R_all, T_all, K = ds.corr_frames()
    z = o_r.sample_z(rays[:, :3], rays[:, 3:6], near, far, t_rand=torch.rand(B, 1, device="cuda:0", generator=g))
    mods = (o_r.sdf_network, o_r.deviation_network, o_r.color_network)
    for m in mods:
        m.double(); m.zero_grad()
    r64 = rays.double()
    o = r64[:, :3].clone().requires_grad_(True); d = r64[:, 3:6].clone().requires_grad_(True)
    Rf = ds.R[frame].double().clone().requires_grad_(True)
    Ra, Ta = R_all.double().clone().requires_grad_(True), T_all.double().clone().requires_grad_(True)
    out = o_r.render(o, d, near.double(), far.double(), cos_anneal_ratio=car, z_vals=z.double())
    ref = O.neus_losses(out, r64[:, 6:9], r64[:, 9:10], r64[:, 10:11], 0.1, 0.1, nw, r64[:, 11:14], Rf)
    cl = O.correspondence_loss(out["weights"], out["z_vals"], 2.0 / 32, o, d, corr.double(), Ra, Ta, K.double(), 4.0)
    (ref["loss"] + cw * cl["loss"]).backward()
    gref = torch.cat([p.grad.reshape(-1) for m in mods for p in m.parameters()])
    for m in mods:
        m.float()
    p_r.sample_z = lambda *a, **k: z
    p_r.train_step_core(rays, near, far, ds.R[frame], car, 0.1, 0.1, nw, corr=corr, corr_weight=cw, corr_frames=ds.corr_frames(),
                        corr_delta_px=4.0, ray_grads=True)
    torch.cuda.synchronize()
    d_o, d_d, d_R = p_r.last_ray_grads
    dRa, dTa = p_r.last_partner_pose_grads
    rel = lambda a, b: ((a.double() - b).norm() / b.norm()).item()
    print(f"pose + correspondence: weights {rel(p_r.store.grad_flat, gref):.2e}; d_o {rel(d_o, o.grad):.2e}; d_d {rel(d_d, d.grad):.2e}; "
          f"partner d_R {rel(dRa, Ra.grad):.2e}, d_T {rel(dTa, Ta.grad):.2e}")
    assert rel(p_r.store.grad_flat, gref) < 1e-4
    assert rel(d_o, o.grad) < 5e-4 and rel(d_d, d.grad) < 5e-4
    assert Ra.grad.norm().item() > 0 and rel(dRa, Ra.grad) < 5e-4 and rel(dTa, Ta.grad) < 5e-4
    assert rel(d_R, Rf.grad) < 5e-4
    # without the direct terms the ray adjoints are measurably wrong (this is what the advisor found missing)
    p_r.train_step_core(rays, near, far, ds.R[frame], car, 0.1, 0.1, nw, ray_grads=True)
    assert rel(p_r.last_ray_grads[0], o.grad) > 1e-2 and p_r.last_partner_pose_grads is None
    # chained into the reference's pose parameters: the partner frames now receive a gradient as well
    ref_p = PoseRefiner(ds.R, ds.T).cuda()
    o_t, d_t, R_t = ref_p.rays(frame, px, py, ds.Kinv)
    ref_p.step(o_t, d_t, R_t, d_o, d_d, d_R, partner=(dRa, dTa))
    partners = sorted(set(corr[corr[:, 2] > 0][:, 3].long().tolist()))
    assert partners and all(f != frame for f in partners)
    Rn, Tn = ref_p.poses()
    moved = [(Tn[f] - ds.T[f]).abs().max().item() > 0 for f in partners]
    assert all(moved), "every partner frame's translation takes a step"
