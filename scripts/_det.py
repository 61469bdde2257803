import os, sys, torch
sys.path.insert(0, "/root/repo")
from dynhor_amd import _lib
from tests.util import flat_from_oracle, randomized_models
if len(sys.argv) > 2: _lib.LIB_PATH = os.path.join("/root/repo", sys.argv[2])
dev = torch.device("cuda:0"); L = _lib.lib()
sdf, col, var = randomized_models(seed=5, device=dev, jitter=0.05)
flat = flat_from_oracle(sdf, var, col)
packed = torch.empty(L.dh_packed_floats(), device=dev)
_lib.check(L.dh_pack_weights(_lib.ptr(flat), _lib.ptr(packed), _lib.stream()))
g = torch.Generator(device="cpu").manual_seed(1)
nrays, n_per_ray = 2048, 128
npts = nrays * n_per_ray
pts = ((torch.rand(npts, 3, generator=g) * 2 - 1) * 1.1).to(dev)
dirs = torch.nn.functional.normalize(torch.randn(nrays, 3, generator=g), dim=-1).to(dev)
w = torch.exp(torch.randn(npts, 1, generator=g) * 3.0).to(dev) * 1e-5
d_sdf = (torch.randn(npts, generator=g).to(dev) * w[:, 0]).contiguous()
d_normals0 = (torch.randn(npts, 3, generator=g).to(dev) * w).contiguous()
d_colors = (torch.randn(npts, 3, generator=g).to(dev) * w).contiguous()
infer, fwd, total = _lib.workspace_floats(npts)
ar = int(sys.argv[1]) if len(sys.argv) > 1 else 2
if len(sys.argv) > 2: _lib.LIB_PATH = os.path.join("/root/repo", sys.argv[2])
ws = torch.zeros(total, device=dev)
nt = (npts + 63) // 64
TF = 64 * 256; AX = 64 * 64
# region map (workspace.h carve order)
regs = []; o = 0
def take(name, n):
    global o
    regs.append((name, o, o + n)); o += n
take("absmax", 4096); take("tmax", (21 * nt + 3) // 4 * 4); take("act", 8 * nt * TF); take("eaux", nt * AX); take("feat", nt * TF)
take("asave", 8 * nt * TF); take("cact", 4 * nt * TF); take("caux", nt * AX); take("featbar", nt * TF); take("tsave", 7 * nt * TF)
take("t0aux", nt * AX); take("rsave", 8 * nt * TF); take("zbar", 8 * nt * TF); take("czbar", 4 * nt * TF); take("tpart", nt * 20 * 256); take("tred", 64 * 20 * 256); take("slabs", total - o)
def run():
    o_sdf = torch.empty(npts, device=dev); o_n = torch.empty(npts, 3, device=dev); o_c = torch.empty(npts, 3, device=dev)
    _lib.check(L.dh_mlp_forward_ex(ar, _lib.ptr(packed), _lib.ptr(pts), _lib.ptr(dirs), n_per_ray, npts, _lib.ptr(ws),
                                   _lib.ptr(o_sdf), _lib.ptr(o_n), _lib.ptr(o_c), _lib.stream()))
    grad = torch.zeros(flat.numel(), device=dev)
    dn = d_normals0.clone()
    _lib.check(L.dh_mlp_backward_ex(ar, _lib.ptr(packed), _lib.ptr(flat), _lib.ptr(pts), npts, _lib.ptr(ws), _lib.ptr(o_c),
                                    _lib.ptr(d_sdf), _lib.ptr(dn), _lib.ptr(d_colors), _lib.ptr(grad), _lib.stream()))
    torch.cuda.synchronize()
    return dict(sdf=o_sdf, n=o_n, c=o_c, dn=dn, grad=grad, ws=ws.clone())
a = run()
# the weight-gradient GEMM alone, repeatedly on the same inputs
sl = [r for r in regs if r[0] == "slabs"][0]
prev = None
for rep in range(4):
    _lib.check(L.dh_weight_grads_gemm_ex(ar, npts, _lib.ptr(ws), _lib.stream()))
    torch.cuda.synchronize()
    cur = ws[sl[1]:sl[2]].clone()
    if prev is not None:
        d = (cur != prev)
        gs = 8 * 1024 * (2 + 8 * 7 + 2 + 8 + 8 + 2 + 8 * 3)
        idx = d.nonzero().flatten()
        nbs = [2, 8, 8, 8, 8, 8, 8, 8, 2, 8, 8, 2, 8, 8, 8]
        offs = [0]
        for nb in nbs: offs.append(offs[-1] + 8 * nb * 1024)
        G = 256
        blocks = cur[:G * gs].view(G, gs); pb = prev[:G * gs].view(G, gs)
        per_job = []
        for j in range(15):
            a_, b_ = blocks[:, offs[j]:offs[j + 1]], pb[:, offs[j]:offs[j + 1]]
            dd = (a_ != b_)
            if dd.any():
                per_job.append((j, int(dd.sum()), float((a_ - b_).abs().max() / b_.abs().max()), sorted(set((dd.nonzero()[:, 0]).tolist()))[:6]))
        print("per job (job, words, max diff / max value, first wgs):", per_job)
        print("dW rerun", rep, "slab words differing:", int(d.sum()), "of", cur.numel(), "nan:", int(torch.isnan(cur).sum()),
              "first diffs (wg, offset in block):", [(int(i) // gs, int(i) % gs) for i in idx[:6]])
    prev = cur
for rep in range(3):
    b = run()
    bad = [k for k in ("sdf", "n", "c", "dn", "grad") if not torch.equal(a[k], b[k])]
    badr = [(name, int((a["ws"][s:e] != b["ws"][s:e]).sum())) for name, s, e in regs if not torch.equal(a["ws"][s:e], b["ws"][s:e])]
    print("rep", rep, "outputs differing:", bad, "regions differing:", badr)
