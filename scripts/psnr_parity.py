"""PSNR at equal iterations: the HIP path vs the oracle (BASELINE.json north_star: "PSNR within 0.1 dB of reference at
equal iterations"; the oracle is this repo's restatement of NeuS -- parity UNPINNED at the reference).

Round-2 protocol (VERDICT r1 "next" item 1): two fp32 trajectories diverge chaotically, so ONE pair of runs evaluated on a
handful of low-resolution frames cannot resolve 0.1 dB.  This script therefore
  * evaluates EVERY frame of the sequence (64), with the HIP forward-only renderer for both arms (its forward parity with the
    oracle is 2e-5, tests/test_gpu_render_forward.py; --cross-check renders the oracle's weights with the oracle's own
    renderer too), at a window of checkpoints (default iterations 1800..2000 every 50) and reports the window mean per arm;
  * runs >= 8 paired seeds (ray stream + initial weights vary with the seed, both arms share them) and reports the paired
    difference mean +- s.e.;
  * offers arms that isolate WHAT differs between the two trajectories:
      hip_vs_oracle       shipping HIP kernels vs the oracle in GPU-eager PyTorch (the north-star comparison)
      hip_vs_hip_f32      shipping two-piece fp16 kernels vs the native fp32-MFMA kernel set (second, independent arithmetic)
      hip_vs_hip_bf16     shipping two-piece fp16 kernels vs the three-piece bf16 kernels of rounds 1-3
      hip_noise_floor     the HIP path vs itself with initial weights perturbed by 1e-7 relative (chaos only: no kernel differs)
      oracle_noise_floor  the oracle vs itself, same perturbation
      hip_scatter         (hash family) shipping table scatter vs the unmerged per-evaluation scatter
      hip_torch_adam      the HIP kernels' gradients stepped by torch.optim.Adam vs by the fused dh_adam_step (hybrid: isolates the optimiser)
      hip_occgrid_vs_hierarchical  (hash family) occupancy-grid marching vs the NeuS sampler: a quality report, not parity
  * --lockstep K: the HIP arm takes the oracle's weights and Adam state every K iterations; per segment it reports the loss
    difference on the first step (pure kernel error), its growth over the segment (chaos) and the SIGNED mean difference
    (a kernel bias would show as a non-zero mean).

    python scripts/psnr_parity.py --mode hip_vs_oracle --seeds 11,22,33,44,55,66,77,88 --out profiles/psnr_parity_r02.json
"""
import argparse, json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dynhor_amd import _lib, schedules
from dynhor_amd.runner import Runner

WARM_UP, END_ITER, LR_ALPHA, ANNEAL_END = 5000, 300000, 0.05, 50000
LOSS_W = (0.1, 0.1, 0.05)          # eikonal, mask, mono-normal (the bench configuration)
LOG = sys.stdout                    # bench.py --psnr points this at stderr


HASH_RENDERER_EXTRA = {}            # --march-samples / --max-samples (occupancy-grid sampler experiments)


def make_runner(family, weight_seed, batch, frames, res, dev, tag, hash_sampler="hierarchical"):
    conf = {"seq_name": "psnr_parity", "exp_name": tag,
            "data_info": {"synthetic": {"n_frames": frames, "H": res, "W": res, "seed": 4321}},
            "train": {"batch_size": batch, "normal_weight": LOSS_W[2], "report_freq": 10 ** 9, "save_freq": 10 ** 9,
                      "val_freq": 0, "end_iter": END_ITER, "warm_up_end": WARM_UP, "anneal_end": ANNEAL_END, "seed": weight_seed},
            "model": {"family": family, "hash_renderer": {"sampler": hash_sampler, **(HASH_RENDERER_EXTRA if hash_sampler == "occgrid" else {})}}}
    return Runner(conf=conf, device=dev, exp_root="/tmp/dynhor_psnr")


class HipArm:
    """The product path: HIP fused training step + fused Adam."""
    def __init__(self, runner, arithmetic=None, scatter_mode=None, torch_adam=False):
        self.r, self.arith, self.scatter = runner, arithmetic, scatter_mode
        self.seconds = 0.0
        self.eval_through_scratch = False
        # torch_adam: the HIP kernels' flat gradient stepped by stock torch.optim.Adam on the flat vector instead of the fused dh_adam_step
        # (a hybrid arm: separates the optimiser from the kernels when an HIP-vs-oracle difference is being located)
        self.torch_adam, self._opt, self._p = torch_adam, None, None

    def _modes(self):
        if self.arith is not None:
            self.r.renderer.arithmetic = self.arith        # passed with every launch
        if self.scatter is not None:               # the merge ablations exist in the float-atomic form of the scatter only
            self.r.renderer.reproducible_table_grad = False
            _lib.check(_lib.lib().dh_hash_set_scatter_mode(self.scatter))

    def step(self, rays, near, far, R, car, lr, t_rand):
        self._modes()
        stats = self.r.renderer.train_step_core(rays, near, far, R, car, *LOSS_W, t_rand=t_rand)
        if self.torch_adam:
            st = self.r.store
            if self._opt is None:
                self._p = torch.nn.Parameter(st.flat)            # shares the flat vector's storage
                self._opt = torch.optim.Adam([self._p], lr=lr)
            for g in self._opt.param_groups:
                g["lr"] = lr
            self._p.grad = st.grad_flat
            self._opt.step()
            st.bump()
        else:
            self.r.store.adam_step(lr)
        return stats[0], stats[5]

    def eval_runner(self, scratch):
        self._modes()
        if self.eval_through_scratch and scratch is not None:
            # evaluate THIS arm's weights with the scratch Runner's renderer (same sampler for both arms: separates what the
            # training sampler did to the model from what the evaluation sampler does to the picture)
            scratch.sdf_network.load_state_dict(self.r.sdf_network.state_dict())
            scratch.color_network.load_state_dict(self.r.color_network.state_dict())
            scratch.deviation_network.load_state_dict(self.r.deviation_network.state_dict())
            scratch.store.bump()
            return scratch
        return self.r

    def perturb(self, rel, seed):
        g = torch.Generator(device=self.r.device); g.manual_seed(seed)
        with torch.no_grad():
            self.r.store.flat.mul_(1.0 + rel * torch.randn(self.r.store.flat.shape, device=self.r.device, generator=g))
        self.r.store.bump()

    def flat(self):
        return self.r.store.flat


class OracleArm:
    """The checker: oracle/ networks trained by stock PyTorch ops + torch.optim.Adam on the GPU (eager)."""
    def __init__(self, family, like_runner, lr, dev):
        from oracle import neus_oracle as O
        self.O = O
        if family == "hash":
            from oracle import hashgrid_oracle as HO
            sdf, col = HO.build_models(seed=1234, device=dev)
            var = O.SingleVarianceNetwork(0.3).to(dev)
        else:
            sdf, col, var = O.build_models(seed=1234, device=dev)
        sdf.load_state_dict(like_runner.sdf_network.state_dict()); col.load_state_dict(like_runner.color_network.state_dict())
        var.load_state_dict(like_runner.deviation_network.state_dict())
        self.sdf, self.col, self.var = sdf, col, var
        self.renderer = O.NeuSRenderer(None, sdf, var, col, 64, 64, 0, 4, 1.0)
        self.opt = torch.optim.Adam(list(sdf.parameters()) + list(var.parameters()) + list(col.parameters()), lr=lr)
        self.seconds = 0.0

    def step(self, rays, near, far, R, car, lr, t_rand):
        for g in self.opt.param_groups:
            g["lr"] = lr
        l = self.O.train_step(self.renderer, self.opt, rays, car, *LOSS_W, R=R, t_rand=t_rand)
        return l["loss"].detach(), torch.as_tensor(l["psnr"]).detach()

    def eval_runner(self, scratch):
        """Evaluation goes through the HIP forward-only renderer: load this arm's weights into the scratch Runner."""
        scratch.sdf_network.load_state_dict(self.sdf.state_dict()); scratch.color_network.load_state_dict(self.col.state_dict())
        scratch.deviation_network.load_state_dict(self.var.state_dict())
        scratch.store.bump()
        return scratch

    def perturb(self, rel, seed):
        g = torch.Generator(device=next(self.sdf.parameters()).device); g.manual_seed(seed)
        with torch.no_grad():
            for p in list(self.sdf.parameters()) + list(self.col.parameters()):
                p.mul_(1.0 + rel * torch.randn(p.shape, device=p.device, generator=g))

    def flat(self):
        return torch.cat([p.detach().reshape(-1) for m in (self.sdf, self.var, self.col) for p in m.state_dict().values()])


@torch.no_grad()
def eval_psnr(runner, it, level, frames=None, chunk=8192):
    """Masked PSNR over ALL frames (upstream validate_image's formula, aggregated): HIP forward-only render, no perturbation."""
    ds = runner.dataset
    car = schedules.cos_anneal_ratio(it, ANNEAL_END)
    se, n = 0.0, 0.0
    for f in (range(ds.n_images) if frames is None else frames):
        rays, h, w = ds.gen_rays_at(f, level)
        near, far = ds._last_near_far
        for s in range(0, rays.shape[0], chunk):
            r = rays[s:s + chunk]
            o, d = r[:, :3].contiguous(), r[:, 3:6].contiguous()
            col, _ = runner.renderer.render_rays(o, d, near[s:s + chunk], far[s:s + chunk], car, None, want_nmap=False)
            m = r[:, 9:10] * r[:, 10:11]
            se += float((((col - r[:, 6:9]) ** 2) * m).sum())
            n += float(m.sum()) * 3.0
    return 10.0 * math.log10((n + 1e-5) / max(se, 1e-30))


def oracle_render_psnr(arm, ds, it, frames, level):
    """The oracle's weights rendered by the oracle's OWN renderer (cross-check of the evaluation protocol)."""
    se = n = 0.0
    for f in frames:
        rays_f, _, _ = ds.gen_rays_at(f, level)
        nr, fr = ds._last_near_far
        for s in range(0, rays_f.shape[0], 2048):
            rr = rays_f[s:s + 2048]
            c = arm.renderer.render(rr[:, :3], rr[:, 3:6], nr[s:s + 2048], fr[s:s + 2048], perturb_overwrite=0,
                                    cos_anneal_ratio=schedules.cos_anneal_ratio(it, ANNEAL_END))["color_fine"].detach()
            m = rr[:, 9:10] * rr[:, 10:11]
            se += float((((c - rr[:, 6:9]) ** 2) * m).sum()); n += float(m.sum()) * 3.0
    return 10.0 * math.log10((n + 1e-5) / se)


ARM_NAMES = {"hip_vs_oracle": ("hip", "oracle_gpu_eager"), "oracle_noise_floor": ("oracle_perturbed_1e-7", "oracle"),
             "hip_vs_hip_f32": ("hip_split_f16", "hip_fp32_mfma"), "hip_vs_hip_bf16": ("hip_split_f16", "hip_split_bf16"), "hip_noise_floor": ("hip_perturbed_1e-7", "hip"),
             "hip_scatter": ("hip_scatter_merged", "hip_scatter_per_evaluation"),
             "hip_torch_adam": ("hip_kernels_torch_adam", "hip_kernels_fused_adam"),
             "hip_occgrid_vs_hierarchical": ("hip_occgrid_sampler", "hip_hierarchical_sampler")}


def run_seed(args, seed, dev):
    wseed = 1000 + seed
    tag = f"{args.family}_{args.mode}_{seed}"
    r_a = make_runner(args.family, wseed, args.batch, args.frames, args.res, dev, tag + "_a",
                      hash_sampler="occgrid" if args.mode == "hip_occgrid_vs_hierarchical" else "hierarchical")
    ds = r_a.dataset
    scratch = None
    mode = args.mode
    if mode == "hip_vs_oracle":
        A = HipArm(r_a)
        B = OracleArm(args.family, r_a, args.lr, dev)
        scratch = make_runner(args.family, wseed, args.batch, 2, 64, dev, tag + "_scratch")
        scratch.dataset = ds
    elif mode == "oracle_noise_floor":
        A = OracleArm(args.family, r_a, args.lr, dev); A.perturb(1e-7, 5)
        B = OracleArm(args.family, r_a, args.lr, dev)
        scratch = r_a
    else:
        r_b = make_runner(args.family, wseed, args.batch, 2, 64, dev, tag + "_b")
        r_b.dataset = ds
        if mode == "hip_vs_hip_f32":
            A, B = HipArm(r_a, arithmetic=_lib.ARITH_SPLIT_F16), HipArm(r_b, arithmetic=_lib.ARITH_FP32_MFMA)
        elif mode == "hip_vs_hip_bf16":
            A, B = HipArm(r_a, arithmetic=_lib.ARITH_SPLIT_F16), HipArm(r_b, arithmetic=_lib.ARITH_SPLIT_BF16)
        elif mode == "hip_noise_floor":
            A, B = HipArm(r_a), HipArm(r_b); A.perturb(1e-7, 5)
        elif mode == "hip_scatter":
            A, B = HipArm(r_a, scatter_mode=0), HipArm(r_b, scatter_mode=2)
        elif mode == "hip_torch_adam":
            A, B = HipArm(r_a, torch_adam=True), HipArm(r_b)
        elif mode == "hip_occgrid_vs_hierarchical":      # a QUALITY report of the two samplers (hash family), not a parity claim
            A, B = HipArm(r_a), HipArm(r_b)
            if args.eval_sampler == "hierarchical":      # both arms' weights rendered by the hierarchical sampler
                scratch = make_runner(args.family, wseed, args.batch, 2, 64, dev, tag + "_scratch")
                scratch.dataset = ds
                A.eval_through_scratch = True
        else:
            raise ValueError(mode)
    fp = schedules.FramePermutation(ds.n_images, 4321 + seed)
    gen = torch.Generator(device=dev); gen.manual_seed(seed)
    evals = sorted(set(args.eval_iters))
    rec = {"seed": seed, "weight_seed": wseed, "arms": list(ARM_NAMES[mode]), "eval": [], "curve": [], "lockstep": []}
    seg = None

    def close_segment():
        fa, fb = A.flat(), B.flat()
        seg["param_rel_divergence_at_end"] = float((fa - fb).norm() / fb.norm())
        d = torch.tensor(seg.pop("_dl"))
        seg.update(dloss_first_step=float(d[0]), dloss_abs_last=float(d[-1].abs()), dloss_signed_mean=float(d.mean()),
                   dloss_abs_max=float(d.abs().max()))
        rec["lockstep"].append(seg)

    for it in range(args.iters):
        frame = fp.frame(it)
        car = schedules.cos_anneal_ratio(it, ANNEAL_END)
        lr = args.lr * schedules.lr_factor(it, WARM_UP, END_ITER, LR_ALPHA)
        px = torch.randint(0, ds.W, [args.batch], device=dev, generator=gen)
        py = torch.randint(0, ds.H, [args.batch], device=dev, generator=gen)
        tr = torch.rand([args.batch, 1], device=dev, generator=gen)
        rays = ds.gen_rays_at_pixels(frame, px, py)
        near, far = ds._last_near_far
        if args.lockstep and it % args.lockstep == 0 and mode == "hip_vs_oracle":
            if seg is not None:
                close_segment()
            # the HIP arm takes the oracle's weights and optimiser state
            r_a.sdf_network.load_state_dict(B.sdf.state_dict()); r_a.color_network.load_state_dict(B.col.state_dict())
            r_a.deviation_network.load_state_dict(B.var.state_dict())
            r_a.store.load_optimizer_state_dict(B.opt.state_dict())
            r_a.store.bump()
            seg = {"start_iter": it, "_dl": []}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        la, pa = A.step(rays, near, far, ds.R[frame], car, lr, tr)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        lb, pb = B.step(rays, near, far, ds.R[frame], car, lr, tr)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        A.seconds += t1 - t0; B.seconds += t2 - t1
        if seg is not None:
            seg["_dl"].append(float(la) - float(lb))
        if (it + 1) % 100 == 0:
            rec["curve"].append({"iter": it + 1, "loss_a": float(la), "loss_b": float(lb), "psnr_batch_a": float(pa), "psnr_batch_b": float(pb)})
        if (it + 1) in evals:
            level = args.final_level if (it + 1) == evals[-1] else args.eval_level
            e = {"iter": it + 1, "level": level, "psnr_a": eval_psnr(A.eval_runner(scratch), it + 1, level),
                 "psnr_b": eval_psnr(B.eval_runner(scratch), it + 1, level)}
            if (it + 1) == evals[-1] and level != args.eval_level:      # the window statistic needs the same level everywhere
                e["psnr_a_window_level"] = eval_psnr(A.eval_runner(scratch), it + 1, args.eval_level)
                e["psnr_b_window_level"] = eval_psnr(B.eval_runner(scratch), it + 1, args.eval_level)
            if args.cross_check and (it + 1) == evals[-1] and mode == "hip_vs_oracle":
                fr = [0, ds.n_images // 4, ds.n_images // 2, (3 * ds.n_images) // 4]
                # independent evaluation: the same four frames with the HIP arm rendered by the HIP renderer and the
                # oracle arm by the ORACLE's own renderer -- a forward bias common to both arms cannot cancel in this difference
                e["hip_weights_by_hip_renderer_4f"] = eval_psnr(A.eval_runner(scratch), it + 1, 4, frames=fr)
                e["oracle_weights_by_hip_renderer_4f"] = eval_psnr(B.eval_runner(scratch), it + 1, 4, frames=fr)
                e["oracle_weights_by_oracle_renderer_4f"] = oracle_render_psnr(B, ds, it + 1, fr, 4)
                rec["delta_4f_shared_renderer_db"] = e["hip_weights_by_hip_renderer_4f"] - e["oracle_weights_by_hip_renderer_4f"]
                rec["delta_4f_independent_renderers_db"] = e["hip_weights_by_hip_renderer_4f"] - e["oracle_weights_by_oracle_renderer_4f"]
            rec["eval"].append(e)
            print(json.dumps({"seed": seed, **e}), flush=True, file=LOG)
    if seg is not None:
        close_segment()
    _lib.check(_lib.lib().dh_hash_set_scatter_mode(0))
    wa = [e.get("psnr_a_window_level", e["psnr_a"]) for e in rec["eval"]]
    wb = [e.get("psnr_b_window_level", e["psnr_b"]) for e in rec["eval"]]
    rec["window_mean_a"], rec["window_mean_b"] = sum(wa) / len(wa), sum(wb) / len(wb)
    rec["window_delta_db"] = rec["window_mean_a"] - rec["window_mean_b"]
    rec["final_delta_db"] = rec["eval"][-1]["psnr_a"] - rec["eval"][-1]["psnr_b"]
    rec["sec_per_iter_a"], rec["sec_per_iter_b"] = A.seconds / args.iters, B.seconds / args.iters
    return rec


def summarize(recs, key):
    d = [r[key] for r in recs]
    n = len(d)
    mean = sum(d) / n
    sd = (sum((x - mean) ** 2 for x in d) / (n - 1)) ** 0.5 if n > 1 else float("nan")
    return {"n": n, "mean_db": mean, "sd_db": sd, "se_db": sd / n ** 0.5 if n > 1 else float("nan"),
            "mean_abs_db": sum(abs(x) for x in d) / n, "per_seed": d}


def run_parity(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="hip_vs_oracle", choices=sorted(ARM_NAMES))
    ap.add_argument("--family", choices=["neus", "hash"], default="neus")
    ap.add_argument("--seeds", type=str, default="11,22,33,44,55,66,77,88")
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--lr", type=float, default=None, help="default 5e-4 (neus) / 5e-3 (hash)")
    ap.add_argument("--eval-iters", type=str, default="1800,1850,1900,1950,2000")
    ap.add_argument("--eval-level", type=int, default=4, help="resolution level of the window checkpoints (4: 128^2 of every frame)")
    ap.add_argument("--final-level", type=int, default=2, help="resolution level of the last checkpoint (1 = full 512^2)")
    ap.add_argument("--lockstep", type=int, default=0)
    ap.add_argument("--cross-check", action="store_true")
    ap.add_argument("--out", type=str, default=None)
    ap.add_argument("--march-samples", type=int, default=None, help="occgrid sampler: marching steps per cube diagonal (default 512)")
    ap.add_argument("--max-samples", type=int, default=None, help="occgrid sampler: capacity in samples per ray (default 128)")
    ap.add_argument("--grid-refresh", choices=["all", "nerfacc"], default=None,
                    help="occgrid sampler: which cells a grid update re-evaluates (default all; nerfacc = all cells for 256 steps, then a quarter + the occupied ones)")
    ap.add_argument("--eval-sampler", choices=["own", "hierarchical"], default="own",
                    help="hip_occgrid_vs_hierarchical: evaluate each arm with its own sampler, or both with the hierarchical one")
    args = ap.parse_args(argv)
    if args.march_samples:
        HASH_RENDERER_EXTRA["march_samples_per_ray"] = args.march_samples
    if args.max_samples:
        HASH_RENDERER_EXTRA["max_samples"] = args.max_samples
    if args.grid_refresh:
        HASH_RENDERER_EXTRA["grid_refresh"] = args.grid_refresh
    args.eval_iters = [int(x) for x in args.eval_iters.split(",") if int(x) <= args.iters] or [args.iters]
    args.lr = args.lr if args.lr is not None else (5e-3 if args.family == "hash" else 5e-4)
    dev = torch.device("cuda:0")
    recs = []
    t0 = time.time()
    for seed in [int(s) for s in args.seeds.split(",")]:
        recs.append(run_seed(args, seed, dev))
        r = recs[-1]
        print(json.dumps({"seed": seed, "window_mean_a": r["window_mean_a"], "window_mean_b": r["window_mean_b"],
                          "window_delta_db": r["window_delta_db"], "final_delta_db": r["final_delta_db"],
                          "elapsed_s": time.time() - t0}), flush=True, file=LOG)
        if args.out:        # a call cut off by its time limit keeps the seeds that finished
            part = (args.out if os.path.isabs(args.out) else os.path.join(ROOT, args.out)) + ".partial"
            os.makedirs(os.path.dirname(part), exist_ok=True)
            json.dump({"mode": args.mode, "family": args.family, "seeds": recs}, open(part, "w"))
    res = {"family": args.family, "mode": args.mode, "arms": recs[0]["arms"], "iters": args.iters, "batch": args.batch,
           "frames": args.frames, "res": args.res, "lr": args.lr, "eval_iters": args.eval_iters, "eval_level": args.eval_level,
           "final_level": args.final_level, "lockstep": args.lockstep, "hash_renderer_extra": dict(HASH_RENDERER_EXTRA), "eval_sampler": args.eval_sampler,
           "protocol": "paired seeds (ray stream + initial weights per seed, shared by both arms); PSNR = masked MSE aggregated over "
                       "ALL frames, rendered by the HIP forward-only path for both arms; window = mean over the eval checkpoints",
           "window_delta": summarize(recs, "window_delta_db"), "final_delta": summarize(recs, "final_delta_db"),
           **({"delta_4f_shared_renderer": summarize(recs, "delta_4f_shared_renderer_db"),
               "delta_4f_independent_renderers": summarize(recs, "delta_4f_independent_renderers_db"),
               "evaluator_difference_4f": summarize([{"d": r["delta_4f_independent_renderers_db"] - r["delta_4f_shared_renderer_db"]} for r in recs], "d")}
              if all("delta_4f_independent_renderers_db" in r for r in recs) else {}),
           "window_mean_a": sum(r["window_mean_a"] for r in recs) / len(recs), "window_mean_b": sum(r["window_mean_b"] for r in recs) / len(recs),
           "sec_per_iter_a": sum(r["sec_per_iter_a"] for r in recs) / len(recs), "sec_per_iter_b": sum(r["sec_per_iter_b"] for r in recs) / len(recs),
           "note": "oracle = this repo's PyTorch restatement of NeuS (parity unpinned at the reference)", "seeds": recs}
    if args.lockstep:
        segs = [s for r in recs for s in r["lockstep"]]
        if segs:
            sm = [s["dloss_signed_mean"] for s in segs]
            m = sum(sm) / len(sm)
            sd = (sum((x - m) ** 2 for x in sm) / max(len(sm) - 1, 1)) ** 0.5
            res["lockstep_summary"] = {"segments": len(segs), "K": args.lockstep,
                                       "first_step_abs_dloss_max": max(abs(s["dloss_first_step"]) for s in segs),
                                       "first_step_abs_dloss_mean": sum(abs(s["dloss_first_step"]) for s in segs) / len(segs),
                                       "segment_signed_mean_dloss_mean": m, "segment_signed_mean_dloss_se": sd / len(sm) ** 0.5,
                                       "param_rel_divergence_at_end_mean": sum(s["param_rel_divergence_at_end"] for s in segs) / len(segs),
                                       "param_rel_divergence_at_end_max": max(s["param_rel_divergence_at_end"] for s in segs)}
    if args.out:
        out = args.out if os.path.isabs(args.out) else os.path.join(ROOT, args.out)
        os.makedirs(os.path.dirname(out), exist_ok=True)
        json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "seeds"}), file=LOG)
    return res


if __name__ == "__main__":
    run_parity()
