"""How long does the host take to ENQUEUE one training iteration (no device sync)?  If this approaches the GPU time per
iteration the loop becomes host-bound (matters for the 8-process data-parallel run)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynhor_amd.runner import Runner
conf = {"seq_name": "h", "exp_name": "h", "data_info": {"synthetic": {"n_frames": 8, "H": 512, "W": 512, "seed": 1}},
        "train": {"batch_size": 2048, "normal_weight": 0.05, "report_freq": 10 ** 9, "save_freq": 10 ** 9, "val_freq": 0},
        "model": {"family": sys.argv[1] if len(sys.argv) > 1 else "neus"}}
r = Runner(conf=conf, device="cuda:0", exp_root="/tmp/dh_host")
for _ in range(5):
    r.train_iteration()
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n):
    r.train_iteration()
t_enq = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / n
print(f"host enqueue {t_enq*1e3:.2f} ms/iter, end-to-end {t_all*1e3:.2f} ms/iter")
