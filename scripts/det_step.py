"""dev: find the stage that is not run-to-run deterministic: while training, evaluate every step TWICE on the same rays from the
same weights and compare the flat gradient and every workspace region bit for bit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dynhor_amd import _lib
from dynhor_amd.runner import Runner
N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
ar = sys.argv[2] if len(sys.argv) > 2 else None
conf = {"seq_name": "det", "exp_name": "s", "data_info": {"synthetic": {"n_frames": 64, "H": 512, "W": 512, "seed": 4321}},
        "train": {"batch_size": 2048, "normal_weight": 0.05, "report_freq": 10 ** 9, "save_freq": 10 ** 9, "val_freq": 0,
                  "end_iter": 300000, "warm_up_end": 5000, "anneal_end": 50000, "learning_rate": 5e-4}, "model": {}}
if ar: conf["model"]["arithmetic"] = ar
r = Runner(conf=conf, device="cuda:0", exp_root="/tmp/dh_det")
g = torch.Generator(device="cuda:0"); g.manual_seed(5)
npts = 2048 * 128
nt = npts // 64
TF = 64 * 256; AX = 64 * 64
regs = []; o = 0
def take(name, n):
    global o
    regs.append((name, o, o + n)); o += n
take("absmax", 4096); take("tmax", (21 * nt + 3) // 4 * 4); take("act", 8 * nt * TF); take("eaux", nt * AX); take("feat", nt * TF)
take("asave", 8 * nt * TF); take("cact", 4 * nt * TF); take("caux", nt * AX); take("featbar", nt * TF); take("tsave", 7 * nt * TF)
take("t0aux", nt * AX); take("rsave", 8 * nt * TF); take("zbar", 8 * nt * TF); take("czbar", 4 * nt * TF); take("tpart", nt * 20 * 256)
total = _lib.workspace_floats(npts)[2]
take("tred", 64 * 20 * 256)
gs = 8 * 1024 * (2 + 8 * 7 + 2 + 8 + 8 + 2 + 8 * 3)
take("slabs", 256 * gs); take("red", gs)
found = 0
for it in range(N):
    r.train_iteration()
    rays = r.dataset.gen_random_rays_at(it % 64, 2048, generator=g)
    near, far = r.dataset._last_near_far
    t_rand = torch.rand(2048, 1, device="cuda:0", generator=g)
    car = r.get_cos_anneal_ratio()
    outs = []
    for rep in range(3):
        stats = r.renderer.train_step_core(rays, near, far, r.dataset.R[it % 64], car, 0.1, 0.1, 0.05, t_rand=t_rand)
        torch.cuda.synchronize()
        s = r.renderer.last_state
        outs.append((stats.clone(), r.store.grad_flat.clone(), s.ws[:o].clone(), s.z_vals.clone(), s.sdf.clone(), s.colors.clone()))
    a, b, c3 = outs
    if not torch.equal(b[1], c3[1]) or not torch.equal(a[1], b[1]):
        print('iteration', it, 'grad equal 1==2', torch.equal(a[1], b[1]), '2==3', torch.equal(b[1], c3[1]), '1==3', torch.equal(a[1], c3[1]), flush=True)
    if not torch.equal(a[1], b[1]) or not torch.equal(a[2], b[2]):
        found += 1
        bad = []
        for name, s0, e0 in regs:
            x, y = a[2][s0:e0], b[2][s0:e0]
            if not torch.equal(x, y):
                xi, yi = x.view(torch.int32), y.view(torch.int32)
                nd = int((xi != yi).sum())
                if name in ("act", "asave", "rsave", "zbar", "tsave", "czbar", "cact"):
                    per = (xi != yi).view(-1, nt * TF).sum(dim=1).tolist()
                elif name == "slabs":
                    nbs = [2, 8, 8, 8, 8, 8, 8, 8, 2, 8, 8, 2, 8, 8, 8]
                    offs = [0]
                    for nb in nbs: offs.append(offs[-1] + 8 * nb * 1024)
                    dd = (xi != yi).view(256, gs)
                    per = [(j, int(dd[:, offs[j]:offs[j + 1]].sum()), sorted(set(dd[:, offs[j]:offs[j + 1]].nonzero()[:, 0].tolist()))[:8]) for j in range(15) if bool(dd[:, offs[j]:offs[j + 1]].any())]
                else:
                    per = None
                idx = (xi != yi).nonzero().flatten()[:4].tolist()
                bad.append((name, nd, per, idx, float((x - y).abs().max())))
        print("iteration", it, "z equal", torch.equal(a[3], b[3]), "sdf equal", torch.equal(a[4], b[4]), "colors equal", torch.equal(a[5], b[5]),
              "grad equal", torch.equal(a[1], b[1]), "regions:", bad, flush=True)
        if found >= 8:
            break
print("done", N, "iterations; non-reproducible steps:", found)
