"""The shipping kernels form every fp32 product from bf16 pieces on the bf16 matrix cores; DH_ALL_F32=1 selects
the native fp32-MFMA twin of every kernel.  Both must produce the same full-size training step: losses to 1e-6, flat
gradient to a few 1e-6 relative (each is 3.7e-6 from the eager-fp32 oracle, tests/test_gpu_fullsize_and_runner.py).
The switches are read once per process, so each mode runs in its own subprocess."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys, torch
sys.path.insert(0, {root!r})
from dynhor_amd.runner import Runner
conf = {{"seq_name": "t", "exp_name": "modes", "data_info": {{"synthetic": {{"n_frames": 4, "H": 128, "W": 128, "seed": 11}}}},
        "train": {{"batch_size": 2048, "normal_weight": 0.05, "report_freq": 10 ** 9, "save_freq": 10 ** 9, "val_freq": 0}}}}
r = Runner(conf=conf, device="cuda:0", exp_root={exp!r})
g = torch.Generator(device="cuda:0"); g.manual_seed(5)
rays = r.dataset.gen_random_rays_at(1, 2048, generator=g)
near, far = r.dataset._last_near_far
t_rand = torch.rand(2048, 1, device="cuda:0", generator=g)
stats = r.renderer.train_step_core(rays, near, far, r.dataset.R[1], 0.3, 0.1, 0.1, 0.05, t_rand=t_rand)
torch.cuda.synchronize()
torch.save({{"stats": stats.cpu(), "grad": r.store.grad_flat.cpu()}}, {out!r})
"""


def _run(tmp_path, name, env_extra):
    out = str(tmp_path / (name + ".pt"))
    env = dict(os.environ)
    for k in ("DH_ALL_F32", "DH_CHAIN_PIECES"):
        env.pop(k, None)
    env.update(env_extra)
    code = SCRIPT.format(root=ROOT, exp=str(tmp_path / name), out=out)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    return torch.load(out)


def test_split_bf16_and_fp32_mfma_kernels_agree(tmp_path):
    a = _run(tmp_path, "split", {})
    b = _run(tmp_path, "f32", {"DH_ALL_F32": "1"})
    c = _run(tmp_path, "pieces", {"DH_CHAIN_PIECES": "1"})       # every chain in its piece-plane form
    for name, other in (("fp32-MFMA", b), ("piece-plane", c)):
        ds = (a["stats"][:6] - other["stats"][:6]).abs().max().item()
        rel = ((a["grad"].double() - other["grad"].double()).norm() / other["grad"].double().norm()).item()
        print(f"shipping kernels vs {name}: loss/stat max abs diff {ds:.2e}; flat gradient rel diff {rel:.2e}")
        assert ds < 5e-6
        assert rel < 1e-5
