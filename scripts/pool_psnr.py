"""Pool paired-seed PSNR records of scripts/psnr_parity.py into one file (the record bench.py's `psnr_at_2k` cites).
    python scripts/pool_psnr.py OUT.json IN1.json IN2.json ... [--what "text"] [--arithmetic split_f16] [--family hash] [--note "text"]"""
import json
import statistics
import sys

args = sys.argv[1:]
what, arith, family, note = None, "split_f16", "neus", None
for flag in ("--what", "--arithmetic", "--family", "--note"):
    if flag in args:
        i = args.index(flag)
        val = args[i + 1]
        del args[i:i + 2]
        if flag == "--what":
            what = val
        elif flag == "--family":
            family = val
        elif flag == "--note":
            note = val
        else:
            arith = val
out, ins = args[0], args[1:]


def summ(d):
    n = len(d)
    m = sum(d) / n
    sd = statistics.stdev(d) if n > 1 else float("nan")
    return {"n": n, "mean_db": m, "sd_db": sd, "se_db": sd / n ** 0.5 if n > 1 else float("nan"), "median_db": statistics.median(d),
            "mean_abs_db": sum(abs(x) for x in d) / n, "per_seed": d}


win, fin, shared, indep, seeds, files = [], [], [], [], [], []
for f in ins:
    d = json.load(open(f))
    files.append(f.split("/")[-1])
    win += d["window_delta"]["per_seed"]
    fin += d["final_delta"]["per_seed"]
    seeds += [r["seed"] for r in d.get("seeds", []) if isinstance(r, dict) and "seed" in r]
    if "delta_4f_independent_renderers" in d:
        shared += d["delta_4f_shared_renderer"]["per_seed"]
        indep += d["delta_4f_independent_renderers"]["per_seed"]
res = {"family": family, "mode": "hip_vs_oracle", "arithmetic": arith, "files": files, **({"note": note} if note else {}),
       "what": what or f"HIP ({arith}) minus oracle (GPU-eager PyTorch), paired seeds (ray stream + initial weights per seed, shared by both "
                       "arms), PSNR = masked MSE over ALL 64 frames in a window of checkpoints at 1800..2000 iterations, 2048 rays x (64+64)",
       "seeds": seeds, "window_delta": summ(win), "final_delta": summ(fin)}
if indep:
    res["independent_evaluator_4_frames_final_checkpoint"] = {
        "what": "the same four frames: HIP arm by the HIP renderer minus oracle arm by the HIP renderer (shared) / by the ORACLE's own renderer "
                "(independent); their difference is what a forward bias common to both arms would hide",
        "shared_renderer": summ(shared), "independent_renderers": summ(indep),
        "difference_of_the_two_evaluations": summ([a - b for a, b in zip(indep, shared)])}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: (v if not isinstance(v, dict) or "per_seed" not in v else {kk: vv for kk, vv in v.items() if kk != "per_seed"}) for k, v in res.items()}, indent=1))
