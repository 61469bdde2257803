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
    print(json.dumps(res))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
