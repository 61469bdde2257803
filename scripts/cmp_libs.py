"""dev: per-region fingerprints of one forward + backward on fixed inputs with a given library; diff two runs.
   python scripts/cmp_libs.py dynhor_amd/libdynhor_hip_base.so out_a.pt ; python scripts/cmp_libs.py dynhor_amd/libdynhor_hip.so out_b.pt ; python scripts/cmp_libs.py --diff out_a.pt out_b.pt"""
import os, sys, torch
sys.path.insert(0, "/root/repo")
if sys.argv[1] == "--diff":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        x, y = a[k].double(), b[k].double()
        d = (x - y).norm().item() / max(y.norm().item(), 1e-30)
        print(f"{k:10s} rel diff {d:.3e}  max abs {float((x - y).abs().max()):.3e}  nan {int(torch.isnan(x).sum())}/{int(torch.isnan(y).sum())}")
    sys.exit(0)
from dynhor_amd import _lib
from tests.util import flat_from_oracle, randomized_models
_lib.LIB_PATH = os.path.join("/root/repo", sys.argv[1])
dev = torch.device("cuda:0"); L = _lib.lib()
sdf, col, var = randomized_models(seed=5, device=dev, jitter=0.05)
flat = flat_from_oracle(sdf, var, col)
packed = torch.empty(L.dh_packed_floats(), device=dev)
_lib.check(L.dh_pack_weights(_lib.ptr(flat), _lib.ptr(packed), _lib.stream()))
g = torch.Generator(device="cpu").manual_seed(1)
nrays, n_per_ray = 256, 128
npts = nrays * n_per_ray
pts = ((torch.rand(npts, 3, generator=g) * 2 - 1) * 1.1).to(dev)
dirs = torch.nn.functional.normalize(torch.randn(nrays, 3, generator=g), dim=-1).to(dev)
w = torch.exp(torch.randn(npts, 1, generator=g) * 3.0).to(dev) * 1e-5
d_sdf = (torch.randn(npts, generator=g).to(dev) * w[:, 0]).contiguous()
d_normals0 = (torch.randn(npts, 3, generator=g).to(dev) * w).contiguous()
d_colors = (torch.randn(npts, 3, generator=g).to(dev) * w).contiguous()
infer, fwd, total = _lib.workspace_floats(npts)
ar = 2
ws = torch.zeros(total, device=dev)
nt = (npts + 63) // 64
TF = 64 * 256; AX = 64 * 64
regs = []; o = 0
def take(name, n):
    global o
    regs.append((name, o, o + n)); o += n
take("absmax", 4096); take("tmax", (21 * nt + 3) // 4 * 4); take("act", 8 * nt * TF); take("eaux", nt * AX); take("feat", nt * TF)
take("asave", 8 * nt * TF); take("cact", 4 * nt * TF); take("caux", nt * AX); take("featbar", nt * TF); take("tsave", 7 * nt * TF)
take("t0aux", nt * AX); take("rsave", 8 * nt * TF); take("zbar", 8 * nt * TF); take("czbar", 4 * nt * TF); take("tpart", nt * 20 * 256)
o_sdf = torch.empty(npts, device=dev); o_n = torch.empty(npts, 3, device=dev); o_c = torch.empty(npts, 3, device=dev)
_lib.check(L.dh_mlp_forward_ex(ar, _lib.ptr(packed), _lib.ptr(pts), _lib.ptr(dirs), n_per_ray, npts, _lib.ptr(ws),
                               _lib.ptr(o_sdf), _lib.ptr(o_n), _lib.ptr(o_c), _lib.stream()))
dn = d_normals0.clone()
S = _lib.stream()
_lib.check(L.dh_color_backward_ex(ar, _lib.ptr(packed), _lib.ptr(o_c), _lib.ptr(d_colors), npts, _lib.ptr(ws), _lib.ptr(dn), S))
_lib.check(L.dh_sdf_tangent_ex(ar, _lib.ptr(packed), _lib.ptr(pts), _lib.ptr(dn), npts, _lib.ptr(ws), S))
torch.cuda.synchronize()
mid = {}
for name, s_, e_ in regs:
    if name in ("rsave", "tsave", "zbar"):
        for l in range((e_ - s_) // (nt * TF)):
            mid[f"mid_{name}{l}"] = ws[s_ + l * nt * TF:s_ + (l + 1) * nt * TF].clone().cpu()
_lib.check(L.dh_sdf_backward_ex(ar, _lib.ptr(packed), _lib.ptr(d_sdf), npts, _lib.ptr(ws), S))
torch.cuda.synchronize()
grad = torch.zeros(1, device=dev)
out = dict(sdf=o_sdf, n=o_n, c=o_c, dn=dn, grad=grad)
for name, s, e in regs:
    t = ws[s:e]
    if name in ("absmax", "tmax"):
        t = t.view(torch.int32).float()
    if name in ("act", "asave", "rsave", "zbar", "tsave", "czbar", "cact"):
        nl = (e - s) // (nt * TF)
        for l in range(nl):
            out[f"{name}{l}"] = t[l * nt * TF:(l + 1) * nt * TF].clone().cpu()
    else:
        out[name] = t.clone().cpu()
out.update(mid)
torch.save({k: v.cpu() for k, v in out.items()}, sys.argv[2])
print("saved", sys.argv[2])
