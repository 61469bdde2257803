"""End-to-end soak: train the HIP path for N iterations on the synthetic hand-held-object sequence, report validation PSNR
and how close the extracted mesh lies to the analytic ground-truth surface (mean |sdf_gt| over mesh vertices)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dynhor_amd.runner import Runner
from dynhor_amd.scene import scene_sdf

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=10000)
ap.add_argument("--out", type=str, default=os.path.join(ROOT, "profiles", "long_train_r01.json"))
ap.add_argument("--family", choices=["neus", "hash"], default="neus")
ap.add_argument("--lr", type=float, default=5e-4)
ap.add_argument("--warm-up-end", type=int, default=5000)
ap.add_argument("--anneal-end", type=int, default=50000)
ap.add_argument("--arithmetic", type=str, default=None, help="split_f16 (default) | split_bf16 | fp32_mfma")
ap.add_argument("--seed", type=int, default=1234, help="initial weights; the ray stream is seeded with seed + 3087")
args = ap.parse_args()
conf = {"seq_name": "soak", "exp_name": "hip", "data_info": {"synthetic": {"n_frames": 64, "H": 512, "W": 512, "seed": 4321}},
        "train": {"batch_size": 2048, "normal_weight": 0.05, "report_freq": 500, "save_freq": 10 ** 9, "val_freq": 0,
                  "end_iter": 300000, "warm_up_end": args.warm_up_end, "anneal_end": args.anneal_end,
                  "learning_rate": args.lr, "seed": args.seed, "ray_seed": args.seed + 3087},
        "model": {"family": args.family}}
if args.arithmetic:
    conf["model"]["arithmetic"] = args.arithmetic
r = Runner(conf=conf, device="cuda:0", exp_root="/tmp/dh_soak")
log = []
t0 = time.perf_counter()
while r.iter_step < args.iters:
    r.train(n_iters=500)
    torch.cuda.synchronize()
    ps = [r.validate_image(idx=i, resolution_level=4) for i in (0, 16, 32, 48)]
    rec = dict(r.scalars[-1]); rec["val_psnr"] = sum(ps) / len(ps); rec["wall_s"] = time.perf_counter() - t0
    assert all(v == v for v in rec.values()), f"NaN at iter {r.iter_step}: {rec}"
    log.append(rec)
    print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in rec.items()}, flush=True)
v, f = r.validate_mesh(resolution=256, save=False)
d = scene_sdf(v).abs()
far = d > 0.05
res = {"family": args.family, "arithmetic": args.arithmetic or "split_f16", "seed": args.seed, "mesh_vertices_farther_than_0.05": int(far.sum()),
       "mesh_far_vertices_radius_range": ([float(v[far].norm(dim=1).min()), float(v[far].norm(dim=1).max())] if bool(far.any()) else None),
       "lr": args.lr, "iters": r.iter_step, "val_psnr": log[-1]["val_psnr"], "mesh_vertices": int(v.shape[0]), "mesh_triangles": int(f.shape[0]),
       "mesh_mean_abs_gt_sdf": float(d.mean()), "mesh_p95_abs_gt_sdf": float(d.quantile(0.95)),
       "object_radius": 0.5, "log": log}
json.dump(res, open(args.out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "log"}))
