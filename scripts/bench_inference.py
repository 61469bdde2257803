"""Secondary measurement: full-frame forward-only rendering (validate_image path) and mesh extraction.
Forward-only work per ray: 438,755,328 FLOP."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynhor_amd.runner import Runner
conf = {"seq_name": "inf", "exp_name": "inf", "data_info": {"synthetic": {"n_frames": 4, "H": 512, "W": 512, "seed": 1}},
        "train": {"batch_size": 2048, "report_freq": 10 ** 9, "save_freq": 10 ** 9, "val_freq": 0}}
r = Runner(conf=conf, device="cuda:0", exp_root="/tmp/dh_inf")
for _ in range(20):
    r.train_iteration()
r.render_image(0, resolution_level=4)
torch.cuda.synchronize(); t0 = time.perf_counter()
img, nrm, rays = r.render_image(1, resolution_level=1, chunk=8192)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
n = 512 * 512
print(f"full 512x512 frame: {dt*1e3:.1f} ms, {n/dt:.0f} rays/s forward-only, {n*438755328/dt/1e12:.1f} TFLOP/s "
      f"({n*438755328/dt/1e12/157.3*100:.1f}% of fp32 MFMA peak)")
for res in (128, 256):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    v, f = r.validate_mesh(resolution=res, save=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"validate_mesh {res}^3: {dt*1e3:.1f} ms, {v.shape[0]} vertices, {f.shape[0]} triangles")
