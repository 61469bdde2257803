"""Development tool: the no-grad SDF chain of two builds of the library side by side in one process -- values and time.

    python scripts/ab_nograd.py --a dynhor_amd/libdynhor_hip.so --b dynhor_amd/libdynhor_hip_nt.so [--out gpurun_out/ab_nograd.json]

Both libraries pack the same weights (library A packs; the packed layout is the same source); each then runs dh_sdf_nograd on the
same points: the bench's coarse (131,072) and fine (32,768) launch sizes, a 2 M-point launch and ragged sizes.  Reported: max
|a - b|, each against the fp32-MFMA twin of library A (an independent arithmetic), median HIP-event time over interleaved launches.
"""
import argparse, ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--a", type=str, default="dynhor_amd/libdynhor_hip.so")
    ap.add_argument("--b", type=str, required=True)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--out", type=str, default=None)
    args = ap.parse_args()
    from dynhor_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, args.a)
    import torch
    from dynhor_amd.fields import SDFNetwork, RenderingNetwork, SingleVarianceNetwork, ParamStore
    La = _lib.lib()
    Lb = ctypes.CDLL(os.path.join(ROOT, args.b))
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    Lb.dh_sdf_nograd.restype = ctypes.c_int
    Lb.dh_sdf_nograd.argtypes = [vp, vp, i64, vp, vp]
    Lb.dh_packed_floats.restype = i64
    assert Lb.dh_packed_floats() == La.dh_packed_floats(), "the two builds disagree on the packed layout"
    torch.manual_seed(7)
    dev = "cuda:0"
    sdf, col, var = SDFNetwork(), RenderingNetwork(), SingleVarianceNetwork(0.3)
    st = ParamStore(sdf, var, col, dev)
    # a trained-looking network: perturb the geometric init so that every layer carries signal
    with torch.no_grad():
        for p in sdf.parameters():
            p.add_(0.02 * torch.randn_like(p))
    packed = st.ensure_packed()
    stream = _lib.stream()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    res = {"a": args.a, "b": args.b, "cases": []}
    for n in (131072, 32768, 2097152, 1000, 129, 1):
        pts = (torch.rand(n, 3, device=dev) * 2 - 1) * 0.9
        oa, ob, of = (torch.full((n,), float("nan"), device=dev) for _ in range(3))
        _lib.check(La.dh_sdf_nograd(P(packed), P(pts), n, P(oa), stream))
        assert Lb.dh_sdf_nograd(P(packed), P(pts), n, P(ob), stream) == 0
        _lib.set_arithmetic(1)
        _lib.check(La.dh_sdf_nograd(P(packed), P(pts), n, P(of), stream))
        _lib.set_arithmetic(0)
        torch.cuda.synchronize()
        c = {"npts": n, "max_abs_a_minus_b": float((oa - ob).abs().max()), "max_abs_a_minus_fp32": float((oa - of).abs().max()),
             "max_abs_b_minus_fp32": float((ob - of).abs().max()), "b_finite": bool(torch.isfinite(ob).all()),
             "sdf_abs_mean": float(of.abs().mean())}
        if n >= 32768:
            ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.reps)]
            for e in ev:
                e[0].record(); La.dh_sdf_nograd(P(packed), P(pts), n, P(oa), stream)
                e[1].record(); Lb.dh_sdf_nograd(P(packed), P(pts), n, P(ob), stream)
                e[2].record()
            torch.cuda.synchronize()
            ta = sorted(e[0].elapsed_time(e[1]) for e in ev)[len(ev) // 2]
            tb = sorted(e[1].elapsed_time(e[2]) for e in ev)[len(ev) // 2]
            fl = 2.0 * 459008 * n
            c.update(ms_a=ta, ms_b=tb, tflops_a=fl / ta * 1e-9, tflops_b=fl / tb * 1e-9)
        print(c, flush=True)
        res["cases"].append(c)
    if args.out:
        with open(os.path.join(ROOT, args.out), "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
