import sys, torch
sys.path.insert(0, "/root/repo")
from tests.test_gpu_hash_family import make_hash_pair, _rays
from dynhor_amd import _lib
o_r, p_r = make_hash_pair(seed=5)
B, car = 48, 0.6
rays_o, rays_d, near, far = _rays(B, seed=2)
g = torch.Generator(device="cpu").manual_seed(9)
t_rand = torch.rand(B, 1, generator=g).cuda()
tgt = torch.rand(B, 3, generator=g).cuda()
with torch.no_grad():
    z = o_r.sample_z(rays_o, rays_d, near, far, t_rand=t_rand)
out = p_r.render(rays_o, rays_d, near, far, cos_anneal_ratio=car, z_vals=z)
loss = (out["color_fine"] - tgt).abs().mean() + 0.1 * out["gradient_error"]
loss.backward()
torch.cuda.synchronize()
st = p_r.store
n = B * z.shape[1]
E = 7
def take(o, cnt):
    return o, o + (cnt + 63) // 64 * 64
o = 0
offs = {}
for name, cnt in (("d_o", n * 4), ("h2", n * 64), ("dz2", n * 64), ("h1", n * 64), ("dz1", n * 64), ("cin", n * 32),
                  ("x01", E * n * 3), ("din", E * n * 32)):
    offs[name], o = take(o, cnt)
ws = p_r._ws
x01 = ws[offs["x01"]:offs["x01"] + E * n * 3].view(E * n, 3).contiguous()
din = ws[offs["din"]:offs["din"] + E * n * 32].view(16, E * n, 2).permute(1, 0, 2).reshape(E * n, 32).contiguous()
ntab = st.slices[0][2]
ref = torch.zeros(ntab, device="cuda")
_lib.check(_lib.lib().dh_hashgrid_encode_backward(_lib.ptr(x01), _lib.ptr(din), E * n, _lib.ptr(ref), _lib.stream()))
torch.cuda.synchronize()
got = st.grad_flat[:ntab]
diff = (got - ref).abs()
print("max diff", diff.max().item(), "ref max", ref.abs().max().item(), "rel", ((got - ref).norm() / ref.norm()).item())
L = _lib.lib()
import ctypes
offs_l = []
for l in range(16):
    s_, r_, o_, d_ = ctypes.c_float(), ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
    L.dh_hashgrid_level(l, ctypes.byref(s_), ctypes.byref(r_), ctypes.byref(o_), ctypes.byref(d_))
    offs_l.append(o_.value)
offs_l.append(ntab // 2)
for l in range(16):
    a, b = offs_l[l] * 2, offs_l[l + 1] * 2
    print(l, "level rel err", ((got[a:b] - ref[a:b]).norm() / (ref[a:b].norm() + 1e-30)).item(), "sum got", got[a:b].sum().item(), "sum ref", ref[a:b].sum().item())
import os
os.makedirs("/root/repo/gpurun_out", exist_ok=True)
lv = []
for l in range(16):
    s_, r_, o_, d_ = ctypes.c_float(), ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
    L.dh_hashgrid_level(l, ctypes.byref(s_), ctypes.byref(r_), ctypes.byref(o_), ctypes.byref(d_))
    lv.append((s_.value, r_.value, o_.value, d_.value))
torch.save({"x01": x01.cpu(), "din": din.cpu(), "levels": lv, "n": n, "got10": got[offs_l[10] * 2: offs_l[12] * 2].cpu(),
            "ref10": ref[offs_l[10] * 2: offs_l[12] * 2].cpu(), "off10": offs_l[10]}, "/root/repo/gpurun_out/hash_dbg.pt")
