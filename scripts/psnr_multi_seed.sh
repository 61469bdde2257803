for s in 11 22 33; do timeout 900 python scripts/psnr_parity.py --iters 2000 --seed $s --weight-seed $((1000+s)) --out gpurun_out/psnr_parity_seed$s.json 2>&1 | grep -v Warn | tail -1; done
