"""dev: which slab entries differ between dW variants? (round-2 debugging)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dynhor_amd import _lib
from dynhor_amd.runner import Runner
from dynhor_amd.renderer import _p
L = _lib.lib()
L.dh_dev_variant.restype = ctypes.c_int; L.dh_dev_variant.argtypes = [ctypes.c_int, ctypes.c_int]
conf = {"seq_name": "ab", "exp_name": "ab", "data_info": {"synthetic": {"n_frames": 4, "H": 128, "W": 128, "seed": 4321}},
        "train": {"batch_size": 2048, "normal_weight": 0.05, "report_freq": 10 ** 9, "save_freq": 10 ** 9, "val_freq": 0}}
r = Runner(conf=conf, device="cuda:0", exp_root="/tmp/dh_ab")
for _ in range(3):
    r.train_iteration()
torch.cuda.synchronize()
s, st = r.renderer.last_state, r.store
P = s.B * s.n
NBS = [2, 8, 8, 8, 8, 8, 8, 8, 2, 8, 8, 2, 8, 8, 8]
gstride = sum(8 * nb * 1024 for nb in NBS)
inf, fwd, tot = _lib.workspace_floats(P)
G = 256
slab_off = tot - (G + 1) * gstride
def run(v):
    L.dh_dev_variant(0, v)
    _lib.check(L.dh_weight_grads_gemm(P, _p(s.ws), _lib.stream()))
    torch.cuda.synchronize()
    return s.ws[slab_off:slab_off + G * gstride].view(G, gstride).clone()
ref = run(0)
ref2 = run(0)
print("v0 rerun identical:", torch.equal(ref, ref2))
for v in (1, 2, 4):
    a = run(v); b = run(v)
    print(f"variant {v}: rerun identical {torch.equal(a, b)}")
    off = 0
    for j, nb in enumerate(NBS):
        n = 8 * nb * 1024
        d = (a[:, off:off + n] - ref[:, off:off + n])
        bad_blocks = (d.abs().amax(dim=1) > 1e-6 * ref[:, off:off + n].abs().amax()).sum().item()
        rel = float(d.norm() / (ref[:, off:off + n].norm() + 1e-30))
        print(f"   job {j:2d} nb {nb}: rel diff {rel:.3e}, workgroups with differences {bad_blocks}/{G}")
        off += n
L.dh_dev_variant(0, 0)
