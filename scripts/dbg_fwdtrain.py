"""Debug aid: where do the saved tiles of two builds differ?  python scripts/dbg_fwdtrain.py --a ..._s.so --b ....so"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--a", required=True); ap.add_argument("--b", required=True); args = ap.parse_args()
from dynhor_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, args.a)
import torch
from dynhor_amd.fields import SDFNetwork, RenderingNetwork, SingleVarianceNetwork, ParamStore
La = _lib.lib(); Lb = ctypes.CDLL(os.path.join(ROOT, args.b))
vp, i64 = ctypes.c_void_p, ctypes.c_int64
Lb.dh_sdf_forward.restype = ctypes.c_int; Lb.dh_sdf_forward.argtypes = [vp, vp, i64, vp, vp, vp]
dev = "cuda:0"; P = lambda t: ctypes.c_void_p(t.data_ptr()); stream = _lib.stream()
torch.manual_seed(5)
sdf, col, var = SDFNetwork(), RenderingNetwork(), SingleVarianceNetwork(0.3)
st = ParamStore(sdf, var, col, dev)
with torch.no_grad():
    for p in sdf.parameters():
        p.add_(0.02 * torch.randn_like(p))
packed = st.ensure_packed()
n = 128
nt = 2
pts = (torch.rand(n, 3, device=dev) * 2 - 1) * 0.9
fw, _, _ = _lib.workspace_floats(n)
wa = torch.zeros(fw, device=dev); wb = torch.zeros(fw, device=dev)
sa = torch.zeros(n, device=dev); sb = torch.zeros(n, device=dev)
_lib.check(La.dh_sdf_forward(P(packed), P(pts), n, P(wa), P(sa), stream)); assert Lb.dh_sdf_forward(P(packed), P(pts), n, P(wb), P(sb), stream) == 0
torch.cuda.synchronize()
TILE, AUX = 16384, 4096
def native_to_rows(t):      # [TILE] native -> [64 points, 256 features]
    x = t.view(4, 2, 2, 4, 64, 4)          # w, m, t, r4, lane, rr
    out = torch.zeros(64, 256, device=t.device)
    for w in range(4):
        for m in range(2):
            for tt in range(2):
                for r4 in range(4):
                    blk = x[w, m, tt, r4]              # [lane, rr]
                    lane = torch.arange(64, device=t.device)
                    rows = (m * 32 + 8 * r4 + 4 * (lane // 32))[:, None] + torch.arange(4, device=t.device)[None, :]
                    cols = (64 * w + 32 * tt + (lane % 32))[:, None].expand(64, 4)
                    out[rows, cols] = blk
    return out
off = 0
for name, cnt, sz in (("act", 8 * nt, TILE), ("eaux", nt, AUX), ("feat", nt, TILE)):
    for i in range(cnt):
        a, b = wa[off:off + sz], wb[off:off + sz]
        d = (a - b).abs()
        if name != "eaux":
            ra, rb = native_to_rows(a), native_to_rows(b)
            dd = (ra - rb).abs() > 1e-5
            badp = dd.any(1).nonzero().flatten().tolist(); badf = dd.any(0).nonzero().flatten().tolist()
            print("%s[%d] (layer %d tile %d): max %.3e bad %d | bad points %d %s | bad features %d %s" % (
                name, i, i // nt, i % nt, float(d.max()), int(dd.sum()), len(badp), badp[:16], len(badf), badf[:40]), flush=True)
        else:
            print("%s[%d]: max %.3e bad %d of %d; nonzero a %d b %d" % (name, i, float(d.max()), int((d > 1e-5).sum()), sz, int((a != 0).sum()), int((b != 0).sum())), flush=True)
            x = (d > 1e-5).view(2, 2, 4, 64, 4)          # m, t, r4, lane, rr
            print("   bad by m", x.sum((1, 2, 3, 4)).tolist(), "by t", x.sum((0, 2, 3, 4)).tolist(), "by r4", x.sum((0, 1, 3, 4)).tolist(), "by rr", x.sum((0, 1, 2, 3)).tolist(), "lanes", x.sum((0, 1, 2, 4)).nonzero().flatten().tolist()[:20])
        off += sz
