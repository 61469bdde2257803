"""Phase stamps of the tile-pair colour forward (csrc/chain_pair.hip under -DDH_STAMPS; build: bash scripts/stamps.sh).

    python scripts/pair_stamps.py [--lib dynhor_amd/libdynhor_hip_stamps.so] [--out gpurun_out/pair_stamps.json]
Slots (per workgroup's third pair, wave 0..3): 0 pair start | 1 prologue done | per layer l: 2+4l phase 1 done, 3+4l behind its barrier,
4+4l phase 2 (+ aux GEMM) done, 5+4l behind its barrier | 86 tail epilogue done | 87 outputs written; k-chunk stamps of layer 2:
phase 1 at 18 + 2 kc, phase 2 at 52 + 2 kc."""
import argparse, ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", type=str, default="dynhor_amd/libdynhor_hip_stamps.so")
    ap.add_argument("--out", type=str, default=None)
    ap.add_argument("--stage", choices=["color_forward", "sdf_gradient"], default="color_forward")
    args = ap.parse_args()
    from dynhor_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, args.lib)
    import numpy as np
    import torch
    sys.argv = [sys.argv[0]]
    from tests.util import flat_from_oracle, randomized_models
    L = _lib.lib()
    dev = torch.device("cuda:0")
    sdf, col, var = randomized_models(seed=3, device=dev, jitter=0.05)
    flat = flat_from_oracle(sdf, var, col)
    packed = torch.empty(L.dh_packed_floats(), device=dev)
    _lib.check(L.dh_pack_weights(_lib.ptr(flat), _lib.ptr(packed), _lib.stream()))
    npts = 2048 * 128
    g = torch.Generator(device="cpu").manual_seed(1)
    pts = ((torch.rand(npts, 3, generator=g) * 2 - 1) * 0.9).to(dev)
    dirs = torch.nn.functional.normalize(torch.randn(2048, 3, generator=g), dim=-1).to(dev)
    p = _lib.ptr
    ws = torch.zeros(_lib.workspace_floats(npts)[1], device=dev)
    o1 = torch.empty(npts, device=dev); nrm = torch.empty(npts, 3, device=dev); colr = torch.empty(npts, 3, device=dev)
    st = _lib.stream()
    F16, PAIR = 2, 0x200
    _lib.check(L.dh_sdf_forward_ex(F16, p(packed), p(pts), npts, p(ws), p(o1), st))
    _lib.check(L.dh_sdf_gradient_ex(F16, p(packed), p(pts), npts, p(ws), p(nrm), 1, st))
    def launch():
        if args.stage == "color_forward":
            return L.dh_color_forward_ex(F16 | PAIR, p(packed), p(pts), p(dirs), 128, p(nrm), npts, p(ws), p(colr), 1, st)
        return L.dh_sdf_gradient_ex(F16 | PAIR, p(packed), p(pts), npts, p(ws), p(nrm), 1, st)
    for _ in range(3):
        _lib.check(launch())
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    _lib.check(launch())
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    n = 256 * 4 * 96
    buf = (ctypes.c_ulonglong * n)()
    fn = L.dh_dev_read_stamps_p
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
    assert fn(ctypes.cast(buf, ctypes.c_void_p), n) == 0
    s = np.frombuffer(buf, dtype=np.uint64).reshape(256, 4, 96).astype(np.float64)
    s = s[(s[:, :, 0] > 0).all(axis=1)]
    d = lambda i, j: float((s[:, :, j] - s[:, :, i]).mean())
    if args.stage == "sdf_gradient":
        # slots: 0 start | 1 prologue done | layer index li = 7 - l: 2+4li phase 1 (+ skip GEMM at l = 4) done, 3+4li behind barrier, 4+4li phase 2
        # done, 5+4li behind barrier | 30 tail epilogue done | 31 normals written; chunk stamps of l = 5: phase 1 at 32 + 2 kc, phase 2 at 64 + 2 kc
        res = {"launch_ms": ms, "workgroups": int(s.shape[0]), "pair_total": d(0, 31), "prologue": d(0, 1)}
        for li in range(7):
            res[f"l{7 - li}"] = {"phase1": round(d(1 if li == 0 else 5 + 4 * (li - 1), 2 + 4 * li)), "barrier1": round(d(2 + 4 * li, 3 + 4 * li)),
                                 "phase2": round(d(3 + 4 * li, 4 + 4 * li)), "barrier2": round(d(4 + 4 * li, 5 + 4 * li))}
        res["tail_epilogue"] = d(29, 30)
        res["final_aux_gemm_and_normals"] = d(30, 31)
        for name, base, end in (("l5_phase1_chunks", 32, 10), ("l5_phase2_chunks", 64, 12)):
            res[name] = [round(d(base + 2 * kc, base + 2 * (kc + 1) if kc < 15 else end), 1) for kc in range(16)]
        print(json.dumps(res, indent=1))
        if args.out:
            json.dump(res, open(args.out, "w"), indent=1)
        return
    res = {"launch_ms": ms, "workgroups": int(s.shape[0]), "pair_total": d(0, 87), "prologue": d(0, 1)}
    for l in range(4):
        res[f"layer{l}"] = {"phase1": d(1 if l == 0 else 5 + 4 * (l - 1), 2 + 4 * l), "barrier1": d(2 + 4 * l, 3 + 4 * l),
                            "phase2": d(3 + 4 * l, 4 + 4 * l), "barrier2": d(4 + 4 * l, 5 + 4 * l)}
    res["prologue_parts"] = {"loads_issued+extras": d(0, 88), "handoff_A": d(88, 89), "handoff_B": d(89, 90), "caux_tiles": d(90, 91), "weight_loads_issued": d(91, 1)}
    res["tail_epilogue"] = d(17, 86)
    res["outputs"] = d(86, 87)
    for name, base in (("layer2_phase1_chunks", 18), ("layer2_phase2_chunks", 52)):
        ch = []
        for kc in range(16):
            nxt = base + 2 * (kc + 1) if kc < 15 else (10 if base == 18 else 12)
            ch.append(round(d(base + 2 * kc, nxt), 1))
        res[name] = ch
    print(json.dumps(res, indent=1))
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
