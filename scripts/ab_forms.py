"""Same-process A/B of the two kernel FORMS of a two-piece fp16 chain stage (include/dynhor_hip.h DH_CHAIN_FORM_TILE / _PAIR): the
bench-sized workspace of real training data, the stage re-launched alternately in both forms, HIP events on the launch stream, plus a
bitwise comparison of everything the stage writes (development tool, round 6).

    python scripts/ab_forms.py [--reps 30] [--stages color_forward,...] [--out gpurun_out/ab_forms.json]
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--stages", type=str, default="color_forward")
    ap.add_argument("--out", type=str, default=None)
    args = ap.parse_args()
    import torch
    from dynhor_amd import _lib
    from dynhor_amd.runner import Runner
    from dynhor_amd.renderer import _p
    L = _lib.lib()
    conf = {"seq_name": "ab", "exp_name": "abf", "data_info": {"synthetic": {"n_frames": 8, "H": 512, "W": 512, "seed": 4321}},
            "train": {"batch_size": 2048, "normal_weight": 0.05, "report_freq": 10 ** 9, "save_freq": 10 ** 9, "val_freq": 0}}
    r = Runner(conf=conf, device="cuda:0", exp_root="/tmp/dh_abf")
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
    F16, TILE, PAIR = _lib.ARITH_SPLIT_F16, 0x100, 0x200
    color = torch.empty(P, 3, device=s.pts.device)
    colors_in = s.colors.clone()
    dn = cap["d_normals"]
    scratch = torch.empty_like(dn)

    def colour_fwd(form):
        return L.dh_color_forward_ex(F16 | form, _p(packed), _p(s.pts), _p(s.rays_d), s.n, _p(s.normals), P, _p(s.ws), _p(color), 1, stream)

    def sdf_grad(form):
        return L.dh_sdf_gradient_ex(F16 | form, _p(packed), _p(s.pts), P, _p(s.ws), _p(s.normals), 1, stream)

    def colour_bwd(form):
        scratch.copy_(dn)
        return L.dh_color_backward_ex(F16 | form, _p(packed), _p(colors_in), _p(cap["d_colors"]), P, _p(s.ws), _p(scratch), stream)
    table = {"color_forward": colour_fwd, "sdf_gradient": sdf_grad, "color_backward": colour_bwd}
    res = {}
    for name in args.stages.split(","):
        fn = table[name]
        outs = {}
        for form, tag in ((TILE, "tile"), (PAIR, "pair")):
            rc = fn(form)
            torch.cuda.synchronize()
            if rc != 0:
                outs[tag] = None
                print(f"{name} {tag}: rc {rc}")
                continue
            outs[tag] = (s.ws.clone(), color.clone(), s.normals.clone(), scratch.clone())
        same = None
        if outs.get("tile") and outs.get("pair"):
            same = all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(outs["tile"], outs["pair"]))
            if not same:
                d = outs["tile"][0].view(torch.int32) != outs["pair"][0].view(torch.int32)
                print(f"{name}: {int(d.sum())} workspace words differ, first at {int(d.nonzero()[0]) if d.any() else None}; "
                      f"colour max |d| {(outs['tile'][1] - outs['pair'][1]).abs().max().item():.3e}")
        del outs
        ms = {"tile": [], "pair": []}
        for _ in range(3):
            fn(TILE); fn(PAIR)
        for _ in range(args.reps):
            for form, tag in ((TILE, "tile"), (PAIR, "pair")):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); rc = fn(form); b.record()
                torch.cuda.synchronize()
                if rc == 0:
                    ms[tag].append(a.elapsed_time(b))
        med = {k: (sorted(v)[len(v) // 2] if v else None) for k, v in ms.items()}
        mn = {k: (min(v) if v else None) for k, v in ms.items()}
        res[name] = {"median_ms": med, "min_ms": mn, "bit_identical": same, "points": P}
        print(name, json.dumps(res[name]))
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
