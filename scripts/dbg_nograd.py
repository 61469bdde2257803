"""Debug aid: layer-by-layer accumulators of a -DDH_T_DEBUG build of the register-resident no-grad chain against a plain fp64
restatement of the SDF MLP.   python scripts/dbg_nograd.py --b dynhor_amd/libdynhor_hip_ntd.so"""
import argparse, ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--b", required=True); args = ap.parse_args()
from dynhor_amd import _lib
import torch
from dynhor_amd.fields import SDFNetwork, RenderingNetwork, SingleVarianceNetwork, ParamStore
La = _lib.lib(); Lb = ctypes.CDLL(os.path.join(ROOT, args.b))
vp, i64 = ctypes.c_void_p, ctypes.c_int64
Lb.dh_sdf_nograd.restype = ctypes.c_int; Lb.dh_sdf_nograd.argtypes = [vp, vp, i64, vp, vp]
Lb.dh_dev_nograd_t_debug.restype = None; Lb.dh_dev_nograd_t_debug.argtypes = [vp, ctypes.c_int]
dev = "cuda:0"; P = lambda t: ctypes.c_void_p(t.data_ptr()); stream = _lib.stream()
n = 1000
torch.manual_seed(3)
pts = (torch.rand(n, 3, device=dev) * 2 - 1) * 0.9
torch.manual_seed(11)
sdf, col, var = SDFNetwork(), RenderingNetwork(), SingleVarianceNetwork(0.3)
st = ParamStore(sdf, var, col, dev)
with torch.no_grad():
    for name, p in sdf.named_parameters():
        p.add_(0.02 * torch.randn_like(p))
packed = st.ensure_packed()
sd = {k: v.detach().double() for k, v in sdf.state_dict().items()}
x = pts.double()
emb = torch.cat([x] + [f(x * 2.0 ** k) for k in range(6) for f in (torch.sin, torch.cos)], 1)
h, pre = emb, []
for l in range(9):
    v, g, b = sd["lin%d.weight_v" % l], sd["lin%d.weight_g" % l].reshape(-1), sd["lin%d.bias" % l]
    W = g[:, None] * v / v.norm(dim=1, keepdim=True)
    if l == 4:
        h = torch.cat([h, emb], 1) / math.sqrt(2.0)
    z = h @ W.T
    pre.append(z)
    if l < 8:
        h = torch.nn.functional.softplus(z + b, beta=100)
ref_sdf = (pre[8] + sd["lin8.bias"])[:, 0]
out = torch.zeros(n, device=dev)
for l in range(8):
    dbg = torch.full((n, 256), float("nan"), device=dev)
    Lb.dh_dev_nograd_t_debug(P(dbg), l)
    assert Lb.dh_sdf_nograd(P(packed), P(pts), n, P(out), stream) == 0
    torch.cuda.synchronize()
    z = pre[l][:, :256]
    w = z.shape[1]
    e = (dbg[:, :w].double() - z).abs()
    bad = (e > 1e-4 * (1 + z.abs()))
    cols = bad.any(0).nonzero().flatten().tolist()
    rows = bad.any(1).nonzero().flatten().tolist()
    print("layer %d: max err %.3e (|z| mean %.3f) nan %d | bad cols %d %s | bad rows %d %s" % (
        l, float(e.nan_to_num(9).max()), float(z.abs().mean()), int(dbg[:, :w].isnan().sum()), len(cols), cols[:24], len(rows), rows[:12]), flush=True)
    if cols and l <= 1:
        c0 = cols[0]
        print("   col %d: got %s ref %s" % (c0, dbg[:4, c0].tolist(), z[:4, c0].tolist()))
print("final: max |sdf - ref| %.3e" % float((out.double() - ref_sdf).abs().max()))
oa = torch.zeros(n, device=dev)
_lib.check(La.dh_sdf_nograd(P(packed), P(pts), n, P(oa), stream)); torch.cuda.synchronize()
print("shipping: max |sdf - ref| %.3e" % float((oa.double() - ref_sdf).abs().max()))
