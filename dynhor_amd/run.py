"""CLI in the reference's style (ObjTracker/run.py:90-95): python -m dynhor_amd.run --config_path X.yaml [--mode train]."""
import argparse

from .runner import Runner


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config_path", type=str, required=True)
    ap.add_argument("--mode", type=str, default="train", choices=["train", "validate_image", "validate_mesh"])
    ap.add_argument("--is_continue", action="store_true")
    ap.add_argument("--iters", type=int, default=None)
    args = ap.parse_args()
    runner = Runner(conf_path=args.config_path, mode=args.mode, is_continue=args.is_continue)
    if args.mode == "train":
        runner.train(args.iters)
    elif args.mode == "validate_image":
        print("psnr", runner.validate_image())
    else:
        print("surface crossings", runner.validate_mesh()[1])


if __name__ == "__main__":
    main()
