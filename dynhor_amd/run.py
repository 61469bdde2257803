"""CLI in the reference's style (ObjTracker/run.py:90-95): python -m dynhor_amd.run --config_path X.yaml [--mode train].

    python -m dynhor_amd.run --config_path configs/synthetic.yaml                 # one GPU
    python -m dynhor_amd.run --config_path configs/synthetic.yaml --gpus 8        # frames shard 8-way data-parallel (RCCL)

With --gpus N > 1 this process starts N ranks (one per GPU) through dynhor_amd.launch before it touches the GPU and exits
with their code; under an external `torch.distributed.run` (WORLD_SIZE set) it is one of the ranks.
"""
import argparse
import os
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config_path", type=str, required=True)
    ap.add_argument("--mode", type=str, default="train", choices=["train", "validate_image", "validate_mesh"])
    ap.add_argument("--is_continue", action="store_true")
    ap.add_argument("--iters", type=int, default=None)
    ap.add_argument("--gpus", type=int, default=1, help="data-parallel ranks on this node (one process per GPU)")
    ap.add_argument("--backend", type=str, default="nccl", help="torch.distributed backend (nccl == RCCL; gloo for tests)")
    ap.add_argument("--share-gpu", action="store_true", help="TEST ONLY: every rank uses cuda:0 (with --backend gloo)")
    ap.add_argument("--exp_root", type=str, default="exps")
    args = ap.parse_args()

    from . import launch
    if args.gpus > 1 and not launch.launched_by_torchrun():
        sys.exit(launch.spawn_ranks("dynhor_amd.run", sys.argv[1:], args.gpus, module=True))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"dynhor_amd.run: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}", file=sys.stderr, flush=True)
        sys.exit(2)
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(args.backend)

    from .runner import Runner
    runner = Runner(conf_path=args.config_path, mode=args.mode, is_continue=args.is_continue,
                    device=f"cuda:{local_rank}", exp_root=args.exp_root)
    if args.mode == "train":
        runner.train(args.iters)
        if runner.rank == 0:
            print(f"trained to iteration {runner.iter_step} on {world} rank(s)", flush=True)
    elif args.mode == "validate_image":
        print("psnr", runner.validate_image())
    else:
        print("surface crossings", runner.validate_mesh()[1])
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
