"""bench.py -- training rays/s of the NeuS reconstruction hot path on N MI355X (contract in the task brief).

One "step" = one full training iteration on 2048 rays x (64 coarse + 64 importance) samples of a synthetic 512x512
sequence (BASELINE.json configs[1]): HIP ray gather -> hierarchical up-sampling -> SDF/colour MLPs
-> volume rendering -> losses -> backward -> (N>1: RCCL all-reduce of the flat gradient) -> fused Adam.  Frames shard
data-parallel over ranks (weak scaling: 2048 rays per rank).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

# algorithmic MACs per sample point of each MLP stage (DESIGN.md "Algorithmic work")
MACS = {
    "sdf_forward": 524544, "sdf_gradient": 459008, "color_forward": 271360, "color_backward": 271360,
    "sdf_tangent": 458752, "sdf_backward": 514560, "weight_grads_gemm": 1254656,
    "sdf_nograd_coarse": 459008, "sdf_nograd_fine": 459008,
}
FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, Chip-level parameters
BF16_MFMA_PEAK_TFLOPS = 2500.0     # same table: bf16 dense
FLOP_PER_RAY_TRAIN = 1081270272


def cpu_baseline(n_rays, n_samples, n_importance):
    """The oracle (PyTorch restatement, oracle/) timed on the host cores: 1 warm-up + 2 timed training iterations on a
    bounded sample of the same workload (n_rays rays x 128 samples)."""
    from oracle import neus_oracle as O
    # the GPU box advertises far more logical CPUs than a PyTorch-CPU run of this size can use (256 threads ran
    # >100x slower than 8); use the cores this process may run on, capped at 8
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count()
    torch.set_num_threads(max(1, min(8, ncpu)))
    sdf, col, var = O.build_models(seed=1234, device="cpu")
    r = O.NeuSRenderer(None, sdf, var, col, n_samples, n_importance, 0, 4, 1.0)
    opt = torch.optim.Adam(list(sdf.parameters()) + list(var.parameters()) + list(col.parameters()), lr=5e-4)
    g = torch.Generator().manual_seed(0)
    o = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=g), dim=-1) * 2.3
    d = torch.nn.functional.normalize((torch.rand(n_rays, 3, generator=g) - 0.5) * 0.8 - o, dim=-1)
    rays = torch.cat([o, d, torch.rand(n_rays, 3, generator=g), (torch.rand(n_rays, 1, generator=g) > 0.5).float(),
                      torch.ones(n_rays, 1), torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=g), dim=-1)], -1)
    times = []
    for it in range(3):
        t0 = time.perf_counter()
        O.train_step(r, opt, rays, 0.5)
        times.append(time.perf_counter() - t0)
    dt = sum(times[1:]) / 2
    return {"value": n_rays / dt, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_rays} rays x {n_samples}+{n_importance} samples, full training iteration (render, losses, "
                      f"backward, Adam), 1 warm-up + 2 timed, fp32, {dt:.2f} s/iter"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--rays-per-rank", type=int, default=2048,
                    help="2048 = throughput mode (weak scaling, the headline); 2048/N = fixed global batch (PSNR-parity mode)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=512)
    ap.add_argument("--backend", type=str, default="nccl", help="torch.distributed backend (nccl == RCCL; gloo for tests)")
    ap.add_argument("--share-gpu", action="store_true", help="TEST ONLY: every rank uses cuda:0 (with --backend gloo)")
    ap.add_argument("--check-sync", action="store_true", help="verify all ranks hold identical parameters at the end")
    ap.add_argument("--force-dist", action="store_true", help="TEST ONLY: initialise the process group even for one rank")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_gpu:
        local_rank = 0
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(args.backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)

    from dynhor_amd.runner import Runner
    conf = {"seq_name": "bench_synth", "exp_name": f"r{rank}",
            "data_info": {"synthetic": {"n_frames": args.frames, "H": 512, "W": 512, "seed": 4321}},
            "train": {"batch_size": args.rays_per_rank, "normal_weight": 0.05, "report_freq": 10 ** 9, "save_freq": 10 ** 9,
                      "val_freq": 0}}
    runner = Runner(conf=conf, device=device, exp_root=os.path.join("/tmp", "dynhor_bench_exps"))
    B = runner.batch_size
    n_samples = runner.renderer.n_samples + runner.renderer.n_importance

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        runner.train_iteration()
    runner.renderer.timer.enabled = True
    runner.renderer.timer.reset()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stats = runner.train_iteration()
    sync()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    runner.renderer.timer.enabled = False
    kern = runner.renderer.timer.summary()
    if args.check_sync and world > 1:
        ref = runner.store.flat.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(ref, runner.store.flat), f"rank {rank}: parameters diverged from rank 0"
        frames = torch.tensor([float(runner.image_perm[(i * world + rank) % runner.dataset.n_images]) for i in range(4)],
                              device=device)
        allf = [torch.empty_like(frames) for _ in range(world)]
        dist.all_gather(allf, frames)
        flat = torch.stack(allf).reshape(-1).tolist()
        assert len(set(flat)) == len(flat), "ranks must draw disjoint frames"
        if rank == 0:
            print("check-sync ok: identical parameters on all ranks, disjoint frames", flush=True)

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        P = B * n_samples
        pts = {"sdf_nograd_coarse": B * runner.renderer.n_samples,
               "sdf_nograd_fine": B * (runner.renderer.n_importance // max(runner.renderer.up_sample_steps, 1))}
        per_kernel = {}
        for k, (mean_ms, cnt) in kern.items():
            if k not in MACS:
                per_kernel[k] = {"ms": round(mean_ms, 4), "launches_per_step": cnt / args.steps}
                continue
            npts = pts.get(k, P)
            tf = 2.0 * MACS[k] * npts / (mean_ms * 1e-3) / 1e12
            per_kernel[k] = {"ms": round(mean_ms, 4), "launches_per_step": cnt / args.steps, "tflops": round(tf, 2)}
        dom = max((k for k in per_kernel if per_kernel[k]["launches_per_step"] <= 1.01 and "tflops" in per_kernel[k]),
                  key=lambda k: per_kernel[k]["ms"])
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(dom, {}).get("hbm_bytes_per_launch")
        all_f32 = os.environ.get("DH_ALL_F32") is not None        # every GEMM on native fp32 MFMA (the A/B twins)
        dw_f32 = all_f32 or os.environ.get("DH_DW_F32") is not None or os.environ.get("DH_DW_REGS") is not None
        if all_f32:
            hip_names = {"weight_grads_gemm": "dw_lds_kernel", "sdf_forward": "sdf_fwd_train_kernel",
                         "sdf_gradient": "sdf_grad_kernel", "color_forward": "color_fwd_kernel",
                         "color_backward": "color_bwd_kernel", "sdf_tangent": "sdf_tangent_kernel",
                         "sdf_backward": "sdf_bwd_kernel", "sdf_nograd_coarse": "sdf_nograd_kernel"}
        else:
            hip_names = {"weight_grads_gemm": "dw_lds_kernel" if dw_f32 else "dw_bf16x3_kernel", "sdf_forward": "sdf_fwd_train_s_kernel",
                         "sdf_gradient": "sdf_grad16_kernel", "color_forward": "color_fwd_s_kernel",
                         "color_backward": "color_bwd16_kernel", "sdf_tangent": "sdf_tangent_s_kernel",
                         "sdf_backward": "sdf_bwd_s_kernel", "sdf_nograd_coarse": "sdf_nograd_s_kernel"}
        # peak of the dominant kernel's own instruction mix: every GEMM runs each fp32 product as 6 bf16 products (3-way
        # split of both operands, fp32 accumulate: 2^-24 relative) on v_mfma_f32_32x32x16_bf16, so the ceiling in ALGORITHMIC
        # (fp32-product) FLOP/s is the dense bf16 peak / 6; DH_ALL_F32 runs v_mfma_f32_32x32x2_f32 (157.3 TFLOP/s)
        split = not (all_f32 or (dom == "weight_grads_gemm" and dw_f32))
        peak = BF16_MFMA_PEAK_TFLOPS / 6.0 if split else FP32_MFMA_PEAK_TFLOPS
        roof = {"bound": "mfma", "kernel": hip_names.get(dom, dom), "stage": dom, "achieved": per_kernel[dom]["tflops"],
                "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(per_kernel[dom]["tflops"] / peak, 4),
                "peak_basis": ("bf16 dense MFMA 2500 TFLOP/s / 6 split products per fp32 product" if split
                               else "fp32 MFMA v_mfma_f32_32x32x2_f32"),
                "frac_of_fp32_mfma_peak": round(per_kernel[dom]["tflops"] / FP32_MFMA_PEAK_TFLOPS, 4),
                "traffic": traffic, "avg_launch_ms": per_kernel[dom]["ms"],
                "whole_step_tflops": round(value / world * FLOP_PER_RAY_TRAIN / 1e12, 2),
                "whole_step_frac_of_split_bf16_peak": round(value / world * FLOP_PER_RAY_TRAIN / 1e12 / (BF16_MFMA_PEAK_TFLOPS / 6.0), 4),
                "whole_step_frac_of_fp32_mfma_peak": round(value / world * FLOP_PER_RAY_TRAIN / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)}
        out = {"metric": "training rays/sec", "value": round(value, 1), "unit": "rays/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
               "scaling": "weak" if args.rays_per_rank == 2048 else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "custom_shoes-shaped synthetic seq, 512x512, 2048 rays x (64+64) samples per rank, "
                                      "NeuS SDF(8x256, skip 4, softplus100) + colour(4x256) MLP, full training iteration",
                          "frames": args.frames, "rays_per_rank": B, "samples_per_ray": n_samples,
                          "parallelism": f"dp{world}", "loss": "rgb L1 + 0.1 eikonal + 0.1 mask BCE + 0.05 mono-normal",
                          "arithmetic": ("fp32 everywhere, every GEMM on v_mfma_f32_32x32x2_f32 (DH_ALL_F32)" if all_f32 else
                                         "fp32 in / fp32 out everywhere; GEMMs as 3-way bf16 split of both operands (6 MFMA "
                                         "products, fp32 accumulate: 2^-24 relative = fp32 accuracy)")},
               "roofline": roof, "kernels": per_kernel,
               "final_stats": {"loss": round(float(stats[0]), 5), "psnr": round(float(stats[5]), 3)}}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_rays, runner.renderer.n_samples, runner.renderer.n_importance)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
