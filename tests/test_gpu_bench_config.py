"""GPU: the benchmark's OWN configuration (BASELINE.json configs[1] as bench.py builds it: 64 frames of 512 x 512, 2048 rays x
64+64 samples) -- VERDICT r1 weak #3: the ray-gather index arithmetic at this size and one Runner.train_iteration on this
dataset were only ever run, never checked.  gen_rays vs the oracle's gather on the first / last / a middle frame including
the four image corners; then one full training iteration of the Runner against the oracle (GPU-eager fp32) on the very rays
and perturbation the Runner drew."""
import pytest
import torch

from oracle import neus_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bench_runner(tmp_path_factory):
    from dynhor_amd.runner import Runner
    conf = {"seq_name": "bench_cfg", "exp_name": "t",
            "data_info": {"synthetic": {"n_frames": 64, "H": 512, "W": 512, "seed": 4321}},          # bench.py's dataset
            "train": {"batch_size": 2048, "normal_weight": 0.05, "report_freq": 10 ** 9, "save_freq": 10 ** 9, "val_freq": 0}}
    return Runner(conf=conf, device="cuda:0", exp_root=str(tmp_path_factory.mktemp("exps")))


def _frames(ds):
    return {"rgb": ds.rgb, "label": ds.label, "normal": ds.normal, "R": ds.R, "T": ds.T, "K": ds.K}


def test_gen_rays_on_the_bench_dataset_matches_oracle(bench_runner):
    ds = bench_runner.dataset
    assert (ds.n_images, ds.H, ds.W) == (64, 512, 512)
    g = torch.Generator(device="cpu").manual_seed(3)
    corners_x = torch.tensor([0, 511, 0, 511, 256, 255])
    corners_y = torch.tensor([0, 0, 511, 511, 256, 511])
    px = torch.cat([corners_x, torch.randint(0, 512, [4090], generator=g)]).cuda()
    py = torch.cat([corners_y, torch.randint(0, 512, [4090], generator=g)]).cuda()
    for f in (0, 31, 63):
        got = ds.gen_rays_at_pixels(f, px, py)
        ref = O.gather_rays(_frames(ds), f, px, py)
        assert got.shape == ref.shape == (4096, 14)
        assert (got[:, 9:11] == ref[:, 9:11]).all(), "mask decode (obj / keep) must be exact"
        assert (got[:, 6:9] - ref[:, 6:9]).abs().max().item() <= 6e-8          # u8 / 255 within 1 ulp
        assert (got - ref).abs().max().item() < 2e-6
        near, far = ds._last_near_far
        rn, rf = O.near_far_from_sphere(ref[:, :3], ref[:, 3:6])
        assert (near - rn).abs().max().item() < 1e-5 and (far - rf).abs().max().item() < 1e-5
    # the last frame's last pixel is the last byte of the resident frame buffers: an index overflow would read past them
    last = ds.gen_rays_at_pixels(63, torch.tensor([511]).cuda(), torch.tensor([511]).cuda())
    assert (last[0, 6:9] - ds.rgb[63, 511, 511].float() / 255.0).abs().max().item() <= 6e-8
    # full-frame generation (validate_image path) agrees with the per-pixel gather
    rays, h, w = ds.gen_rays_at(63, 4)
    assert (h, w) == (128, 128) and rays.shape == (128 * 128, 14)


def test_runner_iteration_on_the_bench_dataset_matches_oracle(bench_runner):
    r = bench_runner
    ds = r.dataset
    # the oracle twin starts from the Runner's weights
    o_sdf, o_col, o_var = O.build_models(seed=1, device="cuda:0")
    o_sdf.load_state_dict(r.sdf_network.state_dict()); o_col.load_state_dict(r.color_network.state_dict())
    o_var.load_state_dict(r.deviation_network.state_dict())
    o_r = O.NeuSRenderer(None, o_sdf, o_var, o_col, 64, 64, 0, 4, 1.0)
    r.iter_step = 3                                   # a non-zero learning rate and cos-anneal ratio
    car = r.get_cos_anneal_ratio()
    w_before = r.store.flat.clone()
    stats = r.train_iteration()
    torch.cuda.synchronize()
    rays, z = r._last_rays, r.renderer.last_state.z_vals
    frame = r.frame_perm.frame(3)
    assert rays.shape == (2048, 14)
    # the rays the Runner drew are rays of that frame: re-gather them with the oracle from the pixel coordinates they encode
    ref_o = -(ds.R[frame].T @ ds.T[frame])
    assert (rays[:, :3] - ref_o).abs().max().item() < 1e-5, "ray origins = camera centre of the permuted frame"
    near, far = O.near_far_from_sphere(rays[:, :3], rays[:, 3:6])
    out = o_r.render(rays[:, :3], rays[:, 3:6], near, far, cos_anneal_ratio=car, z_vals=z)
    ref = O.neus_losses(out, rays[:, 6:9], rays[:, 9:10], rays[:, 10:11], 0.1, 0.1, 0.05, rays[:, 11:14], ds.R[frame])
    ref["loss"].backward()
    gref = torch.cat([p.grad.reshape(-1) for m in (o_sdf, o_var, o_col) for p in m.parameters()])
    names = ["loss", "color_loss", "eikonal_loss", "mask_loss", "normal_loss", "psnr"]
    for i, k in enumerate(names):
        if k in ref:
            assert abs(stats[i].item() - float(ref[k])) < 2e-5 * max(1.0, abs(float(ref[k]))), k
    got = r.store.grad_flat
    rel = ((got - gref).norm() / gref.norm()).item()
    print(f"bench-config iteration: loss {stats[0].item():.6f} (oracle {float(ref['loss']):.6f}), flat gradient rel err {rel:.2e}")
    assert rel < 2e-4        # fp32 vs fp32 at 262,144 points (measured ~4e-6)
    # and the fused Adam moved the weights by one torch.optim.Adam step of the same gradient at the scheduled learning rate
    opt = torch.optim.Adam([torch.nn.Parameter(w_before.clone())], lr=r.learning_rate * 3 / r.warm_up_end)
    opt.param_groups[0]["params"][0].grad = got.clone()
    opt.step()
    assert (opt.param_groups[0]["params"][0].detach() - r.store.flat).abs().max().item() < 1e-6
