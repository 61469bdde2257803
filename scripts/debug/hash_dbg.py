import sys, torch
sys.path.insert(0, "/root/repo")
from tests.test_gpu_hash_family import make_hash_pair, _rays
o_r, p_r = make_hash_pair(seed=5)
B, car = 48, 0.6
rays_o, rays_d, near, far = _rays(B, seed=2)
g = torch.Generator(device="cpu").manual_seed(9)
t_rand = torch.rand(B, 1, generator=g).cuda()
with torch.no_grad():
    z = o_r.sample_z(rays_o, rays_d, near, far, t_rand=t_rand)
    out = p_r.render(rays_o, rays_d, near, far, cos_anneal_ratio=car, z_vals=z)
    s = p_r._forward_core(rays_o, rays_d, z, car, None, want_nmap=False)
    pts = s.pts
    sn = o_r.sdf_network
    g32 = sn.gradient(pts).squeeze(1)
    sn.double()
    g64 = sn.gradient(pts.double()).squeeze(1)
    sn.float()
    err = (s.normals.double() - g64).abs().max(dim=1).values
    e32 = (g32.double() - g64).abs().max(dim=1).values
    idx = err.argsort(descending=True)[:8]
    for i in idx.tolist():
        print(i, i % s.n, pts[i].tolist(), "|x|", pts[i].norm().item(), "hip", s.normals[i].tolist(), "ref", g64[i].tolist(), "err", err[i].item(), "eager err", e32[i].item())
    print("max eager", e32.max().item(), " n bad", (err > 1e-2).sum().item(), "of", err.numel())
