"""Is the HIP hash-family training step NOISIER than the eager-fp32 oracle's?  (locate the -0.2 dB.)

The lock-step (profiles/psnr_parity_r05_hash_lockstep.json) shows no per-step BIAS in the loss; a PSNR deficit can also come from a
larger gradient ERROR (variance) of one arm.  This script trains the bench-shaped hash runner with the HIP path, and at a few iterations
evaluates ONE training step's flat gradient three ways on the same rays and the same sampled depths -- HIP kernels, the oracle in eager
fp32, the oracle in fp64 -- and reports each fp32 arm's relative error against fp64 per parameter group (the table level by level, the
small linears).  Equal errors: the two arms are equally good estimators and the deficit is not a gradient-precision effect.

    python scripts/hash_grad_noise.py [--iters 0,500,2000] [--rays 2048] [--out gpurun_out/hash_grad_noise.json]
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=str, default="0,500,2000")
    ap.add_argument("--rays", type=int, default=2048)
    ap.add_argument("--seed", type=int, default=77)
    ap.add_argument("--out", type=str, default=None)
    args = ap.parse_args()
    import torch
    import psnr_parity as PP
    from oracle import neus_oracle as O
    dev = torch.device("cuda:0")
    r = PP.make_runner("hash", 1000 + args.seed, args.rays, 64, 512, dev, "gradnoise")
    ds = r.dataset
    marks = sorted(int(x) for x in args.iters.split(","))
    res = []
    gen = torch.Generator(device=dev); gen.manual_seed(args.seed)
    it = 0
    for mark in marks:
        while it < mark:
            r.train_iteration(); it += 1
        torch.cuda.synchronize()
        arm = PP.OracleArm("hash", r, 5e-3, dev)                       # copies the runner's current weights
        mods = (arm.sdf, arm.var, arm.col)
        frame = int(it % ds.n_images)
        px = torch.randint(0, ds.W, [args.rays], device=dev, generator=gen)
        py = torch.randint(0, ds.H, [args.rays], device=dev, generator=gen)
        tr = torch.rand([args.rays, 1], device=dev, generator=gen)
        rays = ds.gen_rays_at_pixels(frame, px, py)
        near, far = ds._last_near_far
        R = ds.R[frame]
        car = 0.5
        with torch.no_grad():
            z = arm.renderer.sample_z(rays[:, :3], rays[:, 3:6], near, far, t_rand=tr)

        def oracle_grad(dtype):
            for m in mods:
                m.to(dtype); m.zero_grad()
            c = lambda t: t.to(dtype)
            out = arm.renderer.render(c(rays[:, :3]), c(rays[:, 3:6]), c(near), c(far), cos_anneal_ratio=car, z_vals=c(z))
            L = O.neus_losses(out, c(rays[:, 6:9]), c(rays[:, 9:10]), c(rays[:, 10:11]), 0.1, 0.1, 0.05, c(rays[:, 11:14]), c(R))
            L["loss"].backward()
            g = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).double() for m in mods for p in m.parameters()])
            for m in mods:
                m.float()
            return g, float(L["loss"])
        g64, l64 = oracle_grad(torch.float64)
        g32, l32 = oracle_grad(torch.float32)
        ren = r.renderer
        keep = ren.sample_z
        ren.sample_z = lambda *a, **k: z
        stats = ren.train_step_core(rays, near, far, R, car, 0.1, 0.1, 0.05, t_rand=tr)
        ren.sample_z = keep
        torch.cuda.synchronize()
        gh = r.store.grad_flat.double().clone()
        enc = arm.sdf.encoding
        ntab = enc.n_entries * 2
        groups = [(f"table_L{l:02d}", slice(enc.offsets[l] * 2, (enc.offsets[l] + enc.sizes[l]) * 2)) for l in range(enc.L)]
        groups += [("table_all", slice(0, ntab)), ("mlps", slice(ntab, g64.numel())), ("everything", slice(0, g64.numel()))]
        rec = {"iter": it, "loss_fp64": l64, "loss_eager32_minus_fp64": l32 - l64, "loss_hip_minus_fp64": float(stats[0]) - l64, "groups": {}}
        for name, sl in groups:
            den = g64[sl].norm().item()
            rec["groups"][name] = {"norm_fp64": den, "rel_err_hip": (gh[sl] - g64[sl]).norm().item() / max(den, 1e-300),
                                   "rel_err_eager32": (g32[sl] - g64[sl]).norm().item() / max(den, 1e-300),
                                   # the error components ALONG the true gradient (a systematic shrink / stretch of the step)
                                   "gain_hip": float((gh[sl] * g64[sl]).sum() / max(den * den, 1e-300)),
                                   "gain_eager32": float((g32[sl] * g64[sl]).sum() / max(den * den, 1e-300))}
        res.append(rec)
        t = rec["groups"]
        print(f"iter {it}: loss err hip {rec['loss_hip_minus_fp64']:+.2e} eager {rec['loss_eager32_minus_fp64']:+.2e} | table rel err hip "
              f"{t['table_all']['rel_err_hip']:.2e} eager {t['table_all']['rel_err_eager32']:.2e} | mlps hip {t['mlps']['rel_err_hip']:.2e} eager "
              f"{t['mlps']['rel_err_eager32']:.2e} | gain-1 hip {t['everything']['gain_hip'] - 1:+.2e} eager {t['everything']['gain_eager32'] - 1:+.2e}", flush=True)
        for l in range(enc.L):
            g = t[f"table_L{l:02d}"]
            print(f"    L{l:02d} |g| {g['norm_fp64']:.2e}  hip {g['rel_err_hip']:.2e}  eager {g['rel_err_eager32']:.2e}", flush=True)
        del arm
    if args.out:
        json.dump({"rays": args.rays, "seed": args.seed, "records": res}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
