import sys, torch
sys.path.insert(0, "/root/repo")
from dynhor_amd import _lib
from dynhor_amd.runner import Runner
conf = {"seq_name": "t", "exp_name": "modes", "data_info": {"synthetic": {"n_frames": 4, "H": 128, "W": 128, "seed": 11}},
        "train": {"batch_size": 2048, "normal_weight": 0.05, "report_freq": 10 ** 9, "save_freq": 10 ** 9, "val_freq": 0}}
r = Runner(conf=conf, device="cuda:0", exp_root="/tmp/x")
g = torch.Generator(device="cuda:0"); g.manual_seed(5)
rays = r.dataset.gen_random_rays_at(1, 2048, generator=g)
near, far = r.dataset._last_near_far
t_rand = torch.rand(2048, 1, device="cuda:0", generator=g)
res = {}
def run(name, ar, dw=None):
    r.renderer.arithmetic = ar
    if dw is not None: r.renderer._dw_arith = dw
    elif hasattr(r.renderer, "_dw_arith"): del r.renderer._dw_arith
    stats = r.renderer.train_step_core(rays, near, far, r.dataset.R[1], 0.3, 0.1, 0.1, 0.05, t_rand=t_rand)
    torch.cuda.synchronize()
    res[name] = (stats.clone(), r.store.grad_flat.clone(), r.renderer.last_state.z_vals.clone())
run("f16_first", 2); run("fp32", 1); run("bf16", 0); run("f16", 2); run("f16_again", 2); run("f16_dwbf16", 2, 0); run("f16_dwfp32", 2, 1); run("bf16_dwf16", 0, 2)
b = res["fp32"][1].double()
sl = r.store.slices
for k in res:
    a = res[k][1].double()
    print(k, "grad rel", float((a-b).norm()/b.norm()), "z differ", float(((res[k][2]-res["fp32"][2]).abs()>1e-4).float().mean()), "bitwise == f16:", bool(torch.equal(res[k][1], res["f16"][1])))
# per-parameter breakdown for f16
a = res["f16"][1].double()
names = [n for m in (r.sdf_network, r.deviation_network, r.color_network) for n, _ in m.named_parameters()]
for (p, off, cnt), n in zip(sl, range(len(sl))):
    d = (a[off:off+cnt]-b[off:off+cnt]).norm()/max(b[off:off+cnt].norm(), 1e-30)
    if d > 2e-5: print("slice", n, off, cnt, float(d), float(b[off:off+cnt].norm()))
# class maxima
ws = r.renderer.last_state.ws if hasattr(r.renderer, "last_state") else None
