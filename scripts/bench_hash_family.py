"""Per-stage timing of the hash-grid family's fused training iteration (BASELINE.json configs[3] shape: 2048 rays x 128
samples) on one MI355X: python scripts/bench_hash_family.py [--steps 20] [--rays 2048]."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rays", type=int, default=2048)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from dynhor_amd.runner import Runner
    conf = {"exp_name": "hash_bench",
            "data_info": {"synthetic": {"n_frames": args.frames, "H": 512, "W": 512, "seed": 4321}},
            "train": {"batch_size": args.rays, "report_freq": 10 ** 9, "val_freq": 10 ** 9, "save_freq": 10 ** 9},
            "model": {"family": "hash"}}
    r = Runner(conf=conf, device="cuda:0", exp_root="/tmp/dynhor_hash_bench")
    for _ in range(args.warmup):
        r.train_iteration()
    torch.cuda.synchronize()
    # pass 1: un-instrumented wall clock (HIP events between stages drain the queue: ~1 ms on a 4 ms step)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r.train_iteration()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    # pass 2: per-stage durations from HIP events on the launch stream
    r.renderer.timer.enabled = True
    r.renderer.timer.reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r.train_iteration()
    torch.cuda.synchronize()
    dt_instr = (time.perf_counter() - t0) / args.steps
    kern = r.renderer.timer.summary()
    res = {"family": "hash", "rays": args.rays, "samples_per_ray": r.renderer.n_samples + r.renderer.n_importance,
           "ms_per_step": dt * 1e3, "rays_per_s": args.rays / dt, "ms_per_step_with_stage_events": dt_instr * 1e3,
           "stages_ms": {k: round(v[0] * v[1] / args.steps, 4) for k, v in kern.items()},
           "params": r.store.n}
    # roofline of the dominant stage (dh_hash_weight_grads = table-gradient scatter + the five dW reductions), HBM-bound:
    # algorithmic bytes = every add the per-evaluation scatter defines (7 evaluations x 16 levels x 8 corners x 2 features
    # x 4 B per sample) + one read of the dW operands (geometry: 7 rows per sample of 64+36 and 13+64 floats; colour: one
    # row of 64+32, 64+64, 3+64 floats).  Peak: HBM 8 TB/s (MI355X_MICROARCH.md); the guide's memory-side float-atomic
    # rate (1.3 TB/s of added bytes) is the tighter bound for the scatter half and is reported beside it.
    n = args.rays * res["samples_per_ray"]
    add_bytes = 7 * n * 16 * 8 * 2 * 4
    dw_bytes = 7 * n * (64 + 36 + 13 + 64) * 4 + n * (64 + 32 + 64 + 64 + 3 + 64) * 4
    t = res["stages_ms"]["hash_weight_grads"] * 1e-3
    res["roofline"] = {"bound": "hbm", "stage": "hash_weight_grads", "achieved": (add_bytes + dw_bytes) / t / 1e9,
                       "peak": 8000.0, "unit": "GB/s", "frac": (add_bytes + dw_bytes) / t / 8e12, "traffic": None,
                       "algorithmic_bytes": {"scatter_adds": add_bytes, "dw_operands": dw_bytes},
                       "atomic_peak_GBps": 1300.0}
    print(json.dumps(res))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
