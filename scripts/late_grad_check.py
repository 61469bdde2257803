"""Do the three arithmetics still agree LATE in training?  tests/test_gpu_arithmetic_modes.py compares one step near the initial
weights; here the bench configuration is trained (shipping arithmetic) and at several checkpoints ONE step's losses and flat gradient
are formed under each arithmetic on the same rays -- once with every arithmetic's own sampled depths, once on the fp32-MFMA
arithmetic's depths (isolates the MLP / weight-gradient kernels from the inverse-CDF sampler's conditioning).
    python scripts/late_grad_check.py [--iters 10000] [--out profiles/r04_late_grad_check.json]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dynhor_amd import _lib
from dynhor_amd.runner import Runner

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=10000)
ap.add_argument("--checkpoints", type=str, default="0,500,2000,5000,10000")
ap.add_argument("--out", type=str, default=os.path.join(ROOT, "gpurun_out", "r04_late_grad_check.json"))
ap.add_argument("--train-arithmetic", type=str, default="split_f16")
args = ap.parse_args()
conf = {"seq_name": "late", "exp_name": "hip", "data_info": {"synthetic": {"n_frames": 64, "H": 512, "W": 512, "seed": 4321}},
        "train": {"batch_size": 2048, "normal_weight": 0.05, "report_freq": 10 ** 9, "save_freq": 10 ** 9, "val_freq": 0,
                  "end_iter": 300000, "warm_up_end": 5000, "anneal_end": 50000, "learning_rate": 5e-4},
        "model": {"arithmetic": args.train_arithmetic}}
r = Runner(conf=conf, device="cuda:0", exp_root="/tmp/dh_late")
modes = {"split_f16": _lib.ARITH_SPLIT_F16, "split_bf16": _lib.ARITH_SPLIT_BF16, "fp32_mfma": _lib.ARITH_FP32_MFMA}
train_mode = r.renderer.arithmetic
g = torch.Generator(device="cuda:0"); g.manual_seed(5)
out = []
for ck in [int(x) for x in args.checkpoints.split(",")]:
    if ck > r.iter_step:
        r.train(n_iters=ck - r.iter_step)
    torch.cuda.synchronize()
    rays = r.dataset.gen_random_rays_at(3, 2048, generator=g)
    near, far = r.dataset._last_near_far
    t_rand = torch.rand(2048, 1, device="cuda:0", generator=g)
    car = r.get_cos_anneal_ratio()
    rec = {"iter": r.iter_step, "inv_s": float(1.0 / max(r.scalars[-1]["Statistics/s_val"], 1e-9)) if r.scalars else None}
    orig_sample = r.renderer.sample_z
    res = {}
    for tag, shared in (("own_z", False), ("shared_z", True)):
        if shared:
            z = res[("own_z", "fp32_mfma")][2]
            r.renderer.sample_z = lambda *a, **k: z
        for name, mode in modes.items():
            r.renderer.arithmetic = mode
            stats = r.renderer.train_step_core(rays, near, far, r.dataset.R[3], car, 0.1, 0.1, 0.05, t_rand=t_rand)
            torch.cuda.synchronize()
            res[(tag, name)] = (stats.clone(), r.store.grad_flat.clone(), r.renderer.last_state.z_vals.clone())
        r.renderer.sample_z = orig_sample
        b = res[(tag, "fp32_mfma")]
        for name in ("split_f16", "split_bf16"):
            a = res[(tag, name)]
            d = a[1].double() - b[1].double()
            rec[f"{tag}/{name}/grad_rel"] = float(d.norm() / b[1].double().norm())
            rec[f"{tag}/{name}/grad_max_abs_over_max"] = float(d.abs().max() / b[1].double().abs().max())
            rec[f"{tag}/{name}/loss_abs"] = float((a[0][:6] - b[0][:6]).abs().max())
            rec[f"{tag}/{name}/nan"] = int(torch.isnan(a[1]).sum())
    rec["grad_norm"] = float(res[("shared_z", "fp32_mfma")][1].double().norm())
    r.renderer.arithmetic = train_mode
    out.append(rec)
    print(json.dumps(rec), flush=True)
os.makedirs(os.path.dirname(args.out), exist_ok=True)
json.dump({"what": __doc__, "train_arithmetic": args.train_arithmetic, "records": out}, open(args.out, "w"), indent=1)
