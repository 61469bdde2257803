"""dev: rerun the weight-gradient GEMM on one fixed workspace many times and count the launches whose slabs differ from the first."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dynhor_amd import _lib
if len(sys.argv) > 2:
    _lib.LIB_PATH = os.path.join(ROOT, sys.argv[2])
from dynhor_amd.runner import Runner
from dynhor_amd.renderer import _p
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
AR = int(sys.argv[3]) if len(sys.argv) > 3 else 2
conf = {"seq_name": "det", "exp_name": "dw", "data_info": {"synthetic": {"n_frames": 64, "H": 512, "W": 512, "seed": 4321}},
        "train": {"batch_size": 2048, "normal_weight": 0.05, "report_freq": 10 ** 9, "save_freq": 10 ** 9, "val_freq": 0,
                  "end_iter": 300000, "warm_up_end": 5000, "anneal_end": 50000, "learning_rate": 5e-4}, "model": {}}
r = Runner(conf=conf, device="cuda:0", exp_root="/tmp/dh_det")
r.train(n_iters=300)
torch.cuda.synchronize()
L = _lib.lib()
s = r.renderer.last_state
P = s.B * s.n
nt = P // 64
TF = 64 * 256; AX = 64 * 64
o = 4096 + (21 * nt + 3) // 4 * 4 + 8 * nt * TF + nt * AX + nt * TF + 8 * nt * TF + 4 * nt * TF + nt * AX + nt * TF + 7 * nt * TF + nt * AX + 8 * nt * TF + 8 * nt * TF + 4 * nt * TF + nt * 20 * 256 + 64 * 20 * 256
gs = 8 * 1024 * (2 + 8 * 7 + 2 + 8 + 8 + 2 + 8 * 3)
nbs = [2, 8, 8, 8, 8, 8, 8, 8, 2, 8, 8, 2, 8, 8, 8]
offs = [0]
for nb in nbs: offs.append(offs[-1] + 8 * nb * 1024)
ref = None
bad = 0
per_job = {}
for rep in range(N):
    _lib.check(L.dh_weight_grads_gemm_ex(AR, P, _p(s.ws), _lib.stream()))
    cur = s.ws[o:o + 256 * gs].view(torch.int32)
    if ref is None:
        ref = cur.clone()
        continue
    if rep % 50 == 0 or rep == N - 1:
        pass
    d = (cur != ref)
    if bool(d.any()):
        bad += 1
        dd = d.view(256, gs)
        for j in range(15):
            if bool(dd[:, offs[j]:offs[j + 1]].any()):
                per_job[j] = per_job.get(j, 0) + 1
                if bad <= 6:
                    sub = dd[:, offs[j]:offs[j + 1]]
                    wgs = sorted(set(sub.nonzero()[:, 0].tolist()))
                    for wg in wgs[:3]:
                        w = sub[wg].view(-1, 16, 64)          # [(wave, nt)][r][lane]
                        tiles = [(t // nbs[j], t % nbs[j], int(w[t].sum())) for t in range(w.shape[0]) if bool(w[t].any())]
                        cf = s.ws[o:o + 256 * gs].view(256, gs)[wg, offs[j]:offs[j + 1]]
                        rf = ref.view(torch.float32).view(256, gs)[wg, offs[j]:offs[j + 1]]
                        rel = float((cf - rf).abs().max() / rf.abs().max())
                        lanes = sorted(set(w.nonzero()[:, 2].tolist()))
                        print("rep", rep, "job", j, "wg", wg, "(out tile, n tile, words):", tiles[:16], "max diff / max", rel, "lanes", lanes, "r", sorted(set(w.nonzero()[:, 1].tolist())), flush=True)
if len(sys.argv) > 4 and sys.argv[4] == "--probe":          # a -DDW_AUX_PROBE build: where did the values go wrong?
    import ctypes
    buf = (ctypes.c_uint * 8)()
    fn = L.dh_dev_read_dw_probe
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    torch.cuda.synchronize()
    assert fn(ctypes.cast(buf, ctypes.c_void_p), 0) == 0
    print("probe counters: own tile read back differs", buf[0], "| two reads of tile 8 in one step differ", buf[1], "| repeated split differs", buf[2],
          "| raw registers changed", buf[3], "| steps checked", buf[4])
print("arith", AR, "lib", sys.argv[2] if len(sys.argv) > 2 else "default", "reruns", N - 1, "launches differing from the first:", bad, "by job:", per_job)
