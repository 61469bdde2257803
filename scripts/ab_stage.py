"""Single MLP stages timed on the benchmark's own workspace, and the per-phase cycle stamps of a -DDH_STAMPS build
(development tool; the round-2 A/B variants it was written for are in git history, their numbers in DESIGN.md section 3).

    python scripts/ab_stage.py [--lib dynhor_amd/libdynhor_hip_stamps.so --stamps] [--reps 20]      # scripts/stamps.sh

Runs a few real training iterations of the bench configuration (2048 rays x 64+64 samples) so that the saved tiles hold real
data, then re-launches individual C-ABI stages on that workspace, interleaved A/B/A/B, timed with HIP events on the launch
stream.  Re-launching is idempotent: every stage reads tiles an earlier stage wrote and overwrites its own outputs.
"""
import argparse, ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", type=str, default=None)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--out", type=str, default=None)
    ap.add_argument("--blocks", type=int, default=512, help="workgroups that wrote stamps (a -DDH_GRID_DIV=2 build: 256)")
    ap.add_argument("--stamps", action="store_true", help="the library is a -DDH_STAMPS build: dump per-phase cycle stamps")
    ap.add_argument("--save0", action="store_true", help="colour forward in forward-only mode (save = 0: no saved-tile stores)")
    ap.add_argument("--stamps-t", action="store_true", help="-DDH_STAMPS build: per-layer stamps of the register-resident SDF chains (chain_t.hip)")
    ap.add_argument("--stamps-h", action="store_true", help="-DDH_STAMPS build: phase stamps of the two-piece fp16 chains (kernels_mlp_h.hip)")
    args = ap.parse_args()
    from dynhor_amd import _lib
    if args.lib:
        _lib.LIB_PATH = os.path.join(ROOT, args.lib)
    import torch
    from dynhor_amd.runner import Runner
    from dynhor_amd.renderer import _p
    L = _lib.lib()
    conf = {"seq_name": "ab", "exp_name": "ab", "data_info": {"synthetic": {"n_frames": 8, "H": 512, "W": 512, "seed": 4321}},
            "train": {"batch_size": 2048, "normal_weight": 0.05, "report_freq": 10 ** 9, "save_freq": 10 ** 9, "val_freq": 0}}
    r = Runner(conf=conf, device="cuda:0", exp_root="/tmp/dh_ab")
    ren = r.renderer
    cap = {}
    orig = ren._net_backward

    def capture(s, d_sdf, d_normals, d_colors, grad):
        cap.update(s=s, d_sdf=d_sdf, d_normals=d_normals.clone(), d_colors=d_colors, grad=grad)
        return orig(s, d_sdf, d_normals, d_colors, grad)

    for _ in range(3):
        r.train_iteration()
    ren._net_backward = capture
    r.train_iteration()
    ren._net_backward = orig
    torch.cuda.synchronize()
    s, st = cap["s"], ren.store
    P = s.B * s.n
    packed = st.ensure_packed()
    stream = _lib.stream()
    dn = cap["d_normals"]          # colour backward ACCUMULATES into d_normals: give it a scratch copy each time

    # the sampler's no-grad passes at their two sizes: 64 coarse samples per ray, 16 new samples per up-sampling step
    ng_pts = {"coarse": s.pts[:s.B * 64].contiguous(), "fine": s.pts[:s.B * 16].contiguous()}
    ng_out = torch.empty(s.B * 64, device=s.pts.device)
    stages = {
        "sdf_nograd_coarse": lambda: L.dh_sdf_nograd(_p(packed), _p(ng_pts["coarse"]), s.B * 64, _p(ng_out), stream),
        "sdf_nograd_fine": lambda: L.dh_sdf_nograd(_p(packed), _p(ng_pts["fine"]), s.B * 16, _p(ng_out), stream),
        "sdf_forward": lambda: L.dh_sdf_forward(_p(packed), _p(s.pts), P, _p(s.ws), _p(s.sdf), stream),
        "sdf_gradient": lambda: L.dh_sdf_gradient(_p(packed), _p(s.pts), P, _p(s.ws), _p(s.normals), 1, stream),
        "color_forward": lambda: L.dh_color_forward(_p(packed), _p(s.pts), _p(s.rays_d), s.n, _p(s.normals), P, _p(s.ws), _p(s.colors), 0 if args.save0 else 1, stream),
        "color_backward": lambda: L.dh_color_backward(_p(packed), _p(s.colors), _p(cap["d_colors"]), P, _p(s.ws), _p(dn.clone()), stream),
        "sdf_tangent": lambda: L.dh_sdf_tangent(_p(packed), _p(s.pts), _p(dn), P, _p(s.ws), stream),
        "sdf_backward": lambda: L.dh_sdf_backward(_p(packed), _p(cap["d_sdf"]), P, _p(s.ws), stream),
        "weight_grads_gemm": lambda: L.dh_weight_grads_gemm(P, _p(s.ws), stream),
        "weight_grads_fold": lambda: L.dh_weight_grads_fold(_p(packed), _p(st.flat), P, _p(s.ws), _p(cap["grad"]), stream),
    }

    def time_stage(fn, reps):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record(); _lib.check(fn()); b.record()
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in ev)
        return {"median_ms": t[len(t) // 2], "min_ms": t[0], "mean_ms": sum(t) / len(t)}

    res = {"lib": args.lib or "libdynhor_hip.so", "stages": {}}
    for name, fn in stages.items():
        if name in ("weight_grads_gemm", "weight_grads_fold"):
            continue
        for _ in range(2):
            _lib.check(fn())
        res["stages"][name] = time_stage(fn, args.reps)
        print(name, res["stages"][name], flush=True)
    res["stages"]["weight_grads_gemm"] = time_stage(stages["weight_grads_gemm"], args.reps)
    print("weight_grads_gemm", res["stages"]["weight_grads_gemm"], flush=True)
    res["stages"]["weight_grads_fold"] = time_stage(stages["weight_grads_fold"], args.reps)
    print("weight_grads_fold", res["stages"]["weight_grads_fold"], flush=True)
    g = cap["grad"].double()
    res["grad_checksum"] = {"l2": float(g.norm()), "sum": float(g.sum())}
    print("grad checksum", res["grad_checksum"], flush=True)
    if args.stamps:
        import numpy as np
        n = 512 * 4 * 2 * 10 * 8
        out = {}
        # (the forward chain's stamps went with its tile16.h kernel: the training forward is chain_t.hip's kernel since round 3)
        for key, reader, stage in (("sdf_tangent", "dh_dev_read_stamps_bwd", "sdf_tangent"),):
            _lib.check(stages[stage]())
            torch.cuda.synchronize()
            buf = (ctypes.c_ulonglong * n)()
            fn = getattr(L, reader)
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
            assert fn(ctypes.cast(buf, ctypes.c_void_p), n) == 0
            a = np.frombuffer(buf, dtype=np.uint64).reshape(512, 4, 2, 10, 8).astype(np.float64)[:args.blocks]
            d = {}
            nl = 8
            # per layer (1..6: steady layers), phase durations in cycles, averaged over blocks / waves / the two recorded iterations
            lay = a[:, :, :, 1:7, :]
            names = ["gemm", "epilogue_math(+loads)", "tile_store_issue", "barrier1", "lds_write", "barrier2"]
            for i, nm in enumerate(names):
                dd = lay[..., i + 1] - lay[..., i]
                d[nm] = {"mean": float(dd.mean()), "p10": float(np.percentile(dd, 10)), "p90": float(np.percentile(dd, 90))}
            tot = lay[..., 6] - lay[..., 0]
            d["layer_total"] = {"mean": float(tot.mean()), "p10": float(np.percentile(tot, 10)), "p90": float(np.percentile(tot, 90))}
            # phase alignment across the chip: spread of the layer-3 start stamp (iteration 2) over workgroups
            st3 = a[:, 0, 0, 3, 0]
            d["layer3_start_spread_cycles"] = {"p5": float(np.percentile(st3 - st3.min(), 5)), "p50": float(np.percentile(st3 - st3.min(), 50)),
                                               "p95": float(np.percentile(st3 - st3.min(), 95))}
            # are the two co-resident workgroups (b, b + 256 share a CU only by chance) in phase?  report the histogram of
            # (start of layer 3) mod (mean layer time) over all workgroups
            ph = np.mod(st3 - st3.min(), d["layer_total"]["mean"]) / d["layer_total"]["mean"]
            d["layer3_phase_hist10"] = np.histogram(ph, bins=10, range=(0, 1))[0].tolist()
            if args.blocks == 512:
                # the two workgroups of a CU are b and b + 256 (scripts/micro/hwid_probe.hip): their phase offset, as a
                # fraction of the layer time, at the start of layer 3 of both recorded iterations
                for itn in (0, 1):
                    dl = np.mod(a[256:, 0, itn, 3, 0] - a[:256, 0, itn, 3, 0], d["layer_total"]["mean"]) / d["layer_total"]["mean"]
                    d["cu_pair_phase_hist10_it%d" % itn] = np.histogram(dl, bins=10, range=(0, 1))[0].tolist()
            out[key] = d
            print(key, json.dumps(d, indent=1), flush=True)
        res["stamps"] = out
    if args.stamps_h:
        import numpy as np
        n = 512 * 4 * 2 * 10 * 8
        out = {}
        # (stage, layers recorded in execution order): slots 0 layer start | 1 GEMM done | 2 epilogue math + tile stores issued |
        # 3 tile maximum published | 4 behind barrier 1 | 5 pieces written to LDS | 6 behind barrier 2
        for stage, layers in (("color_forward", [0, 1, 2, 3]), ("sdf_gradient", [7, 6, 5, 4, 3, 2, 1])):
            _lib.check(stages[stage]())
            torch.cuda.synchronize()
            buf = (ctypes.c_ulonglong * n)()
            fn = L.dh_dev_read_stamps_h
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
            assert fn(ctypes.cast(buf, ctypes.c_void_p), n) == 0
            a = np.frombuffer(buf, dtype=np.uint64).reshape(512, 4, 2, 10, 8).astype(np.float64)[:args.blocks]
            lay = a[:, :, :, layers[1:-1], :]          # steady layers (not the first, whose input comes from a tile load)
            names = ["gemm", "epilogue_math+tile_stores", "tile_max", "barrier1", "scale+lds_piece_write", "barrier2"]
            d = {}
            for i, nm in enumerate(names):
                dd = lay[..., i + 1] - lay[..., i]
                d[nm] = {"mean": float(dd.mean()), "p10": float(np.percentile(dd, 10)), "p90": float(np.percentile(dd, 90))}
            tot = lay[..., 6] - lay[..., 0]
            d["layer_total"] = {"mean": float(tot.mean()), "p10": float(np.percentile(tot, 10)), "p90": float(np.percentile(tot, 90))}
            # the two workgroups of a CU are b and b + 256 (scripts/micro/hwid_probe.hip): their phase offset at a steady layer's start
            if args.blocks == 512:
                L0 = layers[2]
                for itn in (0, 1):
                    dl = np.mod(a[256:, 0, itn, L0, 0] - a[:256, 0, itn, L0, 0], d["layer_total"]["mean"]) / d["layer_total"]["mean"]
                    d["cu_pair_phase_hist10_it%d" % itn] = np.histogram(dl, bins=10, range=(0, 1))[0].tolist()
            # waves of one workgroup: spread of their GEMM-done stamps (who waits for whom at barrier 1)
            g = a[:, :, :, layers[2], 1]
            d["gemm_done_spread_within_workgroup"] = {"mean": float((g.max(axis=1) - g.min(axis=1)).mean())}
            if stage == "color_forward":
                # per-chunk stamps of layer 1's GEMM (stamp layers 5, 6: chunk starts; 7 / slot 0: GEMM end)
                ch = np.concatenate([a[:, :, :, 5, :], a[:, :, :, 6, :], a[:, :, :, 7, :1]], axis=-1)      # [..., 17]
                dch = ch[..., 1:] - ch[..., :-1]
                ok = (ch > 0).all(axis=-1)
                d["gemm_chunk_cycles_layer1"] = [float(dch[ok][:, i].mean()) for i in range(16)]
                d["gemm_chunk_cycles_layer1_p90"] = [float(np.percentile(dch[ok][:, i], 90)) for i in range(16)]
            out[stage] = d
            print(stage, json.dumps(d, indent=1), flush=True)
        res["stamps_h"] = out
    if args.stamps_t:
        import numpy as np
        n = 1024 * 4 * 16
        names = ["embedding", "lin0 (3 k-steps)", "lin1", "lin2", "lin3", "lin4 (14 + 3 k-steps)", "lin5", "lin6", "lin7", "lin8 row 0 + sdf store + next points"]
        out = {}
        for stage, last in (("sdf_nograd_coarse", 10), ("sdf_forward", 11)):
            _lib.check(stages[stage]())
            torch.cuda.synchronize()
            buf = (ctypes.c_ulonglong * n)()
            fn = L.dh_dev_read_stamps_t
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
            assert fn(ctypes.cast(buf, ctypes.c_void_p), n) == 0
            a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 4, 16).astype(np.float64)
            a = a[(a[:, :, 0] > 0).all(axis=1)]                       # workgroups that ran a second tile
            d = {"workgroups": int(a.shape[0])}
            nm = names + (["lin8 rows 1..256 + feature tile"] if last == 11 else [])
            for i, k in enumerate(nm):
                dd = a[:, :, i + 1] - a[:, :, i]
                d[k] = {"mean": float(dd.mean()), "p10": float(np.percentile(dd, 10)), "p90": float(np.percentile(dd, 90))}
            tot = a[:, :, last] - a[:, :, 0]
            d["tile_total"] = {"mean": float(tot.mean()), "p10": float(np.percentile(tot, 10)), "p90": float(np.percentile(tot, 90))}
            d["per_k_step_in_the_256_wide_layers"] = float(np.mean([d[k]["mean"] for k in ("lin1", "lin2", "lin3", "lin5", "lin6", "lin7")]) / 16.0)
            out[stage] = d
            print(stage, json.dumps(d, indent=1), flush=True)
        res["stamps_t"] = out
    if args.out:
        os.makedirs(os.path.dirname(os.path.join(ROOT, args.out)), exist_ok=True)
        json.dump(res, open(os.path.join(ROOT, args.out), "w"), indent=1)


if __name__ == "__main__":
    main()
