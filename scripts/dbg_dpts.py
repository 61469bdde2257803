"""dev: per-point d_pts of the HIP path against the oracle's autograd (which tensor / which order?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import neus_oracle as O
from tests.test_gpu_render_forward import make_pair
from dynhor_amd.dataset import Dataset
ds = Dataset.from_synthetic(n_frames=3, H=96, W=96, seed=7, device="cuda:0")
o_r, p_r = make_pair(seed=51, jitter=0.05, n_samples=32, n_importance=32)
B, frame, car, nw = 96, 1, 0.4, 0.05
g = torch.Generator(device="cpu").manual_seed(5)
px = torch.randint(0, ds.W, [B], generator=g).cuda(); py = torch.randint(0, ds.H, [B], generator=g).cuda()
t_rand = torch.rand(B, 1, generator=g).cuda()
rays = ds.gen_rays_at_pixels(frame, px, py)
near, far = ds._last_near_far
z = o_r.sample_z(rays[:, :3], rays[:, 3:6], near, far, t_rand=t_rand)
mods = (o_r.sdf_network, o_r.deviation_network, o_r.color_network)
for m in mods:
    m.double(); m.zero_grad()
r64 = rays.double()
o = r64[:, :3].clone().requires_grad_(True); d = r64[:, 3:6].clone().requires_grad_(True)
R = ds.R[frame].double().clone().requires_grad_(True)
seen = {"on": False}
def grab(mod, inp):
    if "pts" not in seen:
        seen["pts"] = inp[0]; inp[0].register_hook(lambda gr: seen.__setitem__("grad", gr.detach().clone()) if seen["on"] else None)
h = o_r.sdf_network.register_forward_pre_hook(grab)
out = o_r.render(o, d, near.double(), far.double(), cos_anneal_ratio=car, z_vals=z.double())
loss = O.neus_losses(out, r64[:, 6:9], r64[:, 9:10], r64[:, 10:11], 0.1, 0.1, nw, r64[:, 11:14], R)["loss"]
h.remove(); seen["on"] = True; loss.backward()
ref = seen["grad"]
for m in mods:
    m.float()
p_r.sample_z = lambda *a, **k: z
p_r.train_step_core(rays, near, far, ds.R[frame], car, 0.1, 0.1, nw, ray_grads=True)
torch.cuda.synchronize()
s = p_r.last_state
dp = s.d_pts.double()
n = s.n
print("shapes", ref.shape, dp.shape, "n", n)
print("ray sums: ref", ref.view(B, n, 3).sum(1)[0].tolist(), "hip", dp.view(B, n, 3).sum(1)[0].tolist(), "o.grad", o.grad[0].tolist())
e = (dp - ref).abs().amax(1).view(B, n)
m = ref.abs().amax(1).view(B, n)
print("per-point err / max|ref| over everything:", float(e.max() / m.max()))
r = int(e.amax(1).argmax())
print("worst ray", r)
for j in range(0, n, 4):
    print(j, ["%.3e" % v for v in ref.view(B, n, 3)[r, j].tolist()], ["%.3e" % v for v in dp.view(B, n, 3)[r, j].tolist()])
# is it a shift along the ray?
for sh in (-1, 1):
    e2 = (torch.roll(dp.view(B, n, 3), sh, 1) - ref.view(B, n, 3)).abs().amax()
    print("shift", sh, float(e2 / m.max()))
