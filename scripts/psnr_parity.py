"""PSNR @ N iterations: HIP path (dynhor_amd.Runner) vs the oracle in GPU-eager mode (stock PyTorch-ROCm ops), same
synthetic sequence, same initial weights, same ray/perturbation RNG stream, same global batch (BASELINE.md §2 last row).
Writes profiles/psnr_parity_rXX.json.  Run on the GPU box:  python scripts/psnr_parity.py --iters 2000"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dynhor_amd.runner import Runner
from dynhor_amd import schedules
from oracle import neus_oracle as O
from oracle import hashgrid_oracle as HO


def val_psnr(render_fn, ds, frames, level):
    tot_se, tot_n = 0.0, 0.0
    for f in frames:
        rays, h, w = ds.gen_rays_at(f, level)
        near, far = ds._last_near_far
        col = render_fn(rays, near, far)
        m = rays[:, 9:10] * rays[:, 10:11]
        tot_se += float((((col - rays[:, 6:9]) ** 2) * m).sum())
        tot_n += float(m.sum()) * 3.0
    mse = tot_se / (tot_n + 1e-5)
    return 20.0 * torch.log10(torch.tensor(1.0 / (mse ** 0.5))).item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--seed", type=int, default=777, help="ray / perturbation RNG seed (shared by both paths)")
    ap.add_argument("--weight-seed", type=int, default=1234)
    ap.add_argument("--family", choices=["neus", "hash"], default="neus", help="model family (BASELINE.json configs[1] / configs[3])")
    ap.add_argument("--lr", type=float, default=5e-4)
    ap.add_argument("--noise-floor", action="store_true",
                    help="replace the HIP path by a SECOND oracle whose initial weights differ by 1e-7 relative: how far apart do two fp32 runs of the oracle itself land?")
    ap.add_argument("--out", type=str, default=os.path.join(ROOT, "profiles", "psnr_parity_r01.json"))
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    conf = {"seq_name": "psnr_parity", "exp_name": "hip",
            "data_info": {"synthetic": {"n_frames": args.frames, "H": args.res, "W": args.res, "seed": 4321}},
            "train": {"batch_size": args.batch, "normal_weight": 0.05, "report_freq": 10 ** 9, "save_freq": 10 ** 9,
                      "val_freq": 0, "end_iter": 300000, "warm_up_end": 5000, "anneal_end": 50000, "seed": args.weight_seed},
            "model": {"family": args.family}}
    runner = Runner(conf=conf, device=dev, exp_root="/tmp/dynhor_psnr")
    ds = runner.dataset
    # oracle twin: identical initial weights
    def build_oracle_models():
        if args.family == "hash":
            a, b = HO.build_models(seed=1234, device=dev)
            return a, b, O.SingleVarianceNetwork(0.3).to(dev)
        return O.build_models(seed=1234, device=dev)

    o_sdf, o_col, o_var = build_oracle_models()
    o_sdf.load_state_dict(runner.sdf_network.state_dict()); o_col.load_state_dict(runner.color_network.state_dict())
    o_var.load_state_dict(runner.deviation_network.state_dict())
    o_r = O.NeuSRenderer(None, o_sdf, o_var, o_col, 64, 64, 0, 4, 1.0)
    opt = torch.optim.Adam(list(o_sdf.parameters()) + list(o_var.parameters()) + list(o_col.parameters()), lr=args.lr)
    if args.noise_floor:
        t_sdf, t_col, t_var = build_oracle_models()
        t_sdf.load_state_dict(o_sdf.state_dict()); t_col.load_state_dict(o_col.state_dict()); t_var.load_state_dict(o_var.state_dict())
        with torch.no_grad():
            gp = torch.Generator(device=dev); gp.manual_seed(5)
            for p in list(t_sdf.parameters()) + list(t_col.parameters()):
                p.mul_(1.0 + 1e-7 * torch.randn(p.shape, device=dev, generator=gp))
        t_r = O.NeuSRenderer(None, t_sdf, t_var, t_col, 64, 64, 0, 4, 1.0)
        t_opt = torch.optim.Adam(list(t_sdf.parameters()) + list(t_var.parameters()) + list(t_col.parameters()), lr=args.lr)
    perm = runner.image_perm.clone()
    gen_h = torch.Generator(device=dev); gen_h.manual_seed(args.seed)
    gen_o = torch.Generator(device=dev); gen_o.manual_seed(args.seed)

    def draw(gen):
        px = torch.randint(0, ds.W, [args.batch], device=dev, generator=gen)
        py = torch.randint(0, ds.H, [args.batch], device=dev, generator=gen)
        t = torch.rand([args.batch, 1], device=dev, generator=gen)
        return px, py, t

    curve = []
    t_h = t_o = 0.0
    for it in range(args.iters):
        frame = int(perm[it % ds.n_images])
        car = schedules.cos_anneal_ratio(it, 50000)
        lr = args.lr * schedules.lr_factor(it, 5000, 300000, 0.05)
        # ---- HIP path
        torch.cuda.synchronize(); t0 = time.perf_counter()
        px, py, tr = draw(gen_h)
        rays = ds.gen_rays_at_pixels(frame, px, py)
        near, far = ds._last_near_far
        if args.noise_floor:
            for g in t_opt.param_groups:
                g["lr"] = lr
            l_t = O.train_step(t_r, t_opt, rays, car, 0.1, 0.1, 0.05, R=ds.R[frame], t_rand=tr)
            stats_h = torch.tensor([float(l_t["loss"]), 0, 0, 0, 0, float(l_t["psnr"])])
        else:
            stats_h = runner.renderer.train_step_core(rays, near, far, ds.R[frame], car, 0.1, 0.1, 0.05, t_rand=tr)
            runner.store.adam_step(lr)
        torch.cuda.synchronize(); t_h += time.perf_counter() - t0
        # ---- oracle, GPU eager, same rays
        t0 = time.perf_counter()
        px, py, tr = draw(gen_o)
        rays_o = ds.gen_rays_at_pixels(frame, px, py)
        for g in opt.param_groups:
            g["lr"] = lr
        l_o = O.train_step(o_r, opt, rays_o, car, 0.1, 0.1, 0.05, R=ds.R[frame], t_rand=tr)
        torch.cuda.synchronize(); t_o += time.perf_counter() - t0
        if (it + 1) % 100 == 0 or it == 0:
            rec = {"iter": it + 1, "hip_loss": float(stats_h[0]), "hip_psnr_batch": float(stats_h[5]),
                   "oracle_loss": float(l_o["loss"]), "oracle_psnr_batch": float(l_o["psnr"])}
            curve.append(rec)
            print(rec, flush=True)
    # held-out style validation: 4 frames at 1/4 resolution, no perturbation
    frames = [0, ds.n_images // 4, ds.n_images // 2, (3 * ds.n_images) // 4]

    @torch.no_grad()
    def render_hip(rays, near, far):
        out = []
        for s in range(0, rays.shape[0], 4096):
            o, d = rays[s:s + 4096, :3].contiguous(), rays[s:s + 4096, 3:6].contiguous()
            z = runner.renderer.sample_z(o, d, near[s:s + 4096], far[s:s + 4096], perturb_overwrite=0)
            out.append(runner.renderer._forward_core(o, d, z, 1.0, None, want_nmap=False).color)
        return torch.cat(out)

    def render_orc(rays, near, far):
        out = []
        for s in range(0, rays.shape[0], 2048):
            o, d = rays[s:s + 2048, :3], rays[s:s + 2048, 3:6]
            r = o_r.render(o, d, near[s:s + 2048], far[s:s + 2048], perturb_overwrite=0, cos_anneal_ratio=1.0)
            out.append(r["color_fine"].detach())
        return torch.cat(out)

    def render_twin(rays, near, far):
        out = []
        for s in range(0, rays.shape[0], 2048):
            o, d = rays[s:s + 2048, :3], rays[s:s + 2048, 3:6]
            out.append(t_r.render(o, d, near[s:s + 2048], far[s:s + 2048], perturb_overwrite=0, cos_anneal_ratio=1.0)["color_fine"].detach())
        return torch.cat(out)

    p_h = val_psnr(render_twin if args.noise_floor else render_hip, ds, frames, 4)
    p_o = val_psnr(render_orc, ds, frames, 4)
    res = {"family": args.family, "lr": args.lr, "seed": args.seed, "weight_seed": args.weight_seed, "iters": args.iters, "batch": args.batch, "frames": args.frames, "res": args.res,
           "mode": "noise_floor: oracle vs 1e-7-perturbed oracle" if args.noise_floor else "hip vs oracle", "val_psnr_hip": p_h, "val_psnr_oracle_gpu_eager": p_o, "abs_diff_db": abs(p_h - p_o),
           "sec_per_iter_hip": t_h / args.iters, "sec_per_iter_oracle_gpu_eager": t_o / args.iters,
           "note": "oracle = this repo's PyTorch restatement of NeuS (parity unpinned at the reference)", "curve": curve}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "curve"}))


if __name__ == "__main__":
    main()
