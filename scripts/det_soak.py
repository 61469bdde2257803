"""dev: is a training run bitwise reproducible?  Two Runners with the same seeds, N iterations each; parameter checksums every 100.
    python scripts/det_soak.py [N] [arithmetic|-] [neus|hash] [hierarchical|occgrid]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dynhor_amd.runner import Runner
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
ar = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None
family = sys.argv[3] if len(sys.argv) > 3 else "neus"
sampler = sys.argv[4] if len(sys.argv) > 4 else "hierarchical"
def run(tag):
    conf = {"seq_name": "det", "exp_name": tag, "data_info": {"synthetic": {"n_frames": 64, "H": 512, "W": 512, "seed": 4321}},
            "train": {"batch_size": 2048, "normal_weight": 0.05, "report_freq": 10 ** 9, "save_freq": 10 ** 9, "val_freq": 0,
                      "end_iter": 300000, "warm_up_end": 5000, "anneal_end": 50000, "learning_rate": 5e-3 if family == "hash" else 5e-4},
            "model": {"family": family, "hash_renderer": {"sampler": sampler}}}
    if ar: conf["model"]["arithmetic"] = ar
    r = Runner(conf=conf, device="cuda:0", exp_root="/tmp/dh_det")
    sums = []
    while r.iter_step < N:
        r.train(n_iters=100)
        torch.cuda.synchronize()
        f = r.store.flat
        sums.append((r.iter_step, float(f.double().sum()), float(f.double().norm()), f.clone()))
    return sums
a = run("a"); b = run("b")
first = None
for (i, s1, n1, f1), (_, s2, n2, f2) in zip(a, b):
    same = torch.equal(f1, f2)
    if not same and first is None:
        first = i
        d = (f1 - f2).abs()
        print("first difference at iteration", i, "max abs", float(d.max()), "n differing", int((d > 0).sum()))
print("family", family, sampler if family == "hash" else "", "arith", ar, "iterations", N, "bitwise identical:", first is None, "first differing checkpoint:", first)
