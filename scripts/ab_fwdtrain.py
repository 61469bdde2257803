"""Development tool: dh_sdf_forward (the training forward: sdf + every saved tile) of two builds of the library on the same
input, whole workspace compared element for element, both timed.

    python scripts/ab_fwdtrain.py --a dynhor_amd/libdynhor_hip_s.so --b dynhor_amd/libdynhor_hip.so
"""
import argparse, ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--a", required=True); ap.add_argument("--b", required=True)
    ap.add_argument("--reps", type=int, default=20); ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from dynhor_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, args.a)
    import torch
    from dynhor_amd.fields import SDFNetwork, RenderingNetwork, SingleVarianceNetwork, ParamStore
    La = _lib.lib(); Lb = ctypes.CDLL(os.path.join(ROOT, args.b))
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    Lb.dh_sdf_forward.restype = ctypes.c_int; Lb.dh_sdf_forward.argtypes = [vp, vp, i64, vp, vp, vp]
    Lb.dh_packed_floats.restype = i64
    assert Lb.dh_packed_floats() == La.dh_packed_floats()
    dev = "cuda:0"; P = lambda t: ctypes.c_void_p(t.data_ptr()); stream = _lib.stream()
    torch.manual_seed(5)
    sdf, col, var = SDFNetwork(), RenderingNetwork(), SingleVarianceNetwork(0.3)
    st = ParamStore(sdf, var, col, dev)
    with torch.no_grad():
        for p in sdf.parameters():
            p.add_(0.02 * torch.randn_like(p))
    packed = st.ensure_packed()
    res = {"a": args.a, "b": args.b, "cases": []}
    for n in (262144, 64, 65, 1000, 129, 32768 + 64):
        pts = (torch.rand(n, 3, device=dev) * 2 - 1) * 0.9
        fwd_floats, _, _ = _lib.workspace_floats(n)
        wa = torch.zeros(fwd_floats, device=dev); wb = torch.zeros(fwd_floats, device=dev)
        sa = torch.full((n,), float("nan"), device=dev); sb = torch.full((n,), float("nan"), device=dev)
        _lib.check(La.dh_sdf_forward(P(packed), P(pts), n, P(wa), P(sa), stream))
        assert Lb.dh_sdf_forward(P(packed), P(pts), n, P(wb), P(sb), stream) == 0
        torch.cuda.synchronize()
        d = (wa - wb).abs()
        nz = (wa != 0) | (wb != 0)
        c = {"npts": n, "ws_floats": fwd_floats, "sdf_max_abs_diff": float((sa - sb).abs().max()), "ws_max_abs_diff": float(d.max()),
             "ws_written_a": int((wa != 0).sum()), "ws_written_b": int((wb != 0).sum()), "ws_written_by_one_only": int(((wa != 0) ^ (wb != 0)).sum()),
             "ws_frac_diff_gt_1e-5": float((d[nz] > 1e-5).float().mean()) if nz.any() else 0.0, "finite": bool(torch.isfinite(wb).all() and torch.isfinite(sb).all())}
        if d.max() > 1e-4:
            idx = int(d.argmax()); c["worst_index"] = idx; c["worst_a_b"] = [float(wa[idx]), float(wb[idx])]
        if n >= 262144:
            ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.reps)]
            for e in ev:
                e[0].record(); La.dh_sdf_forward(P(packed), P(pts), n, P(wa), P(sa), stream)
                e[1].record(); Lb.dh_sdf_forward(P(packed), P(pts), n, P(wb), P(sb), stream)
                e[2].record()
            torch.cuda.synchronize()
            ta = sorted(e[0].elapsed_time(e[1]) for e in ev)[len(ev) // 2]; tb = sorted(e[1].elapsed_time(e[2]) for e in ev)[len(ev) // 2]
            fl = 2.0 * 524544 * n
            c.update(ms_a=ta, ms_b=tb, tflops_a=fl / ta * 1e-9, tflops_b=fl / tb * 1e-9)
        print(c, flush=True); res["cases"].append(c)
    if args.out:
        json.dump(res, open(os.path.join(ROOT, args.out), "w"), indent=1)


if __name__ == "__main__":
    main()
