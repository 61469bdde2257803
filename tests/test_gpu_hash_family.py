"""Hash-grid model family (BASELINE.json configs[3]) through the C ABI vs oracle/hashgrid_oracle.py + the NeuS oracle's
renderer / losses.  Floating point: the finite-difference normals divide fp32 sdf differences by 2e-3, so every
comparison is made against the oracle in fp64 and judged next to what the SAME oracle does in eager fp32 (stated
tolerances below)."""
import pytest
import torch

from oracle import hashgrid_oracle as HO
from oracle import neus_oracle as O

pytestmark = pytest.mark.gpu


def make_hash_pair(seed=3, table_scale=0.3, jitter=0.05, n_samples=16, n_importance=16, up_sample_steps=2):
    """(oracle renderer, product renderer) sharing randomised weights: table U(-s,s), every matrix jittered so the
    encoding columns, weight-norm and biases all matter."""
    from dynhor_amd.fields import SingleVarianceNetwork
    from dynhor_amd.hash_fields import HashNeuSRenderer, HashSDFNetwork, SHRenderingNetwork
    o_sdf, o_col = HO.build_models(seed=seed, device="cuda")
    o_var = O.SingleVarianceNetwork(0.3).cuda()
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    with torch.no_grad():
        o_sdf.encoding.table.copy_(((torch.rand(o_sdf.encoding.table.shape, generator=g) * 2 - 1) * table_scale).cuda())
        for m in (o_sdf, o_col):
            for name, p in m.named_parameters():
                if name == "encoding.table":
                    continue
                noise = torch.randn(p.shape, generator=g).cuda() * jitter
                if name.endswith("weight_g"):
                    p.mul_(1.0 + noise)
                else:
                    p.add_(noise * (0.2 if name.endswith("bias") else 0.25))
    p_sdf, p_col, p_var = HashSDFNetwork(), SHRenderingNetwork(), SingleVarianceNetwork(0.3)
    p_sdf.load_state_dict(o_sdf.state_dict())
    p_col.load_state_dict(o_col.state_dict())
    p_var.load_state_dict(o_var.state_dict())
    kw = dict(n_samples=n_samples, n_importance=n_importance, n_outside=0, up_sample_steps=up_sample_steps, perturb=1.0)
    o_r = O.NeuSRenderer(None, o_sdf, o_var, o_col, **kw)
    p_r = HashNeuSRenderer(None, p_sdf, p_var, p_col, **kw)
    return o_r, p_r


def _pts(n, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(n, 3, generator=g)
    x = x / x.norm(dim=1, keepdim=True) * torch.rand(n, 1, generator=g) ** (1 / 3) * 0.98
    return x.cuda()


def test_layout_matches_state_dict_order():
    o_r, p_r = make_hash_pair()
    mods = (o_r.sdf_network, o_r.deviation_network, o_r.color_network)
    flat = torch.cat([p.detach().reshape(-1) for m in mods for p in m.state_dict().values()])
    assert flat.numel() == p_r.store.n
    assert torch.equal(flat, p_r.store.flat)


@pytest.mark.parametrize("n", [1, 255, 4099])
def test_geo_forward_matches_oracle(n):
    o_r, p_r = make_hash_pair()
    x = _pts(n, seed=n)
    sdf_net = o_r.sdf_network
    with torch.no_grad():
        ref32 = sdf_net(x)
        g32 = sdf_net.gradient(x).squeeze(1)
        sdf_net.double()
        ref = sdf_net(x.double())
        gref = sdf_net.gradient(x.double()).squeeze(1)
        sdf_net.float()
    got_sdf = p_r.sdf(x)
    s = type("S", (), {})()
    s.B, s.n, s.pts, s.rays_d, s.infer_only = n, 1, x, x, True
    s.ws = p_r._workspace(n, infer_only=True)
    s.sdf = torch.empty(n, device="cuda"); s.normals = torch.empty(n, 3, device="cuda"); s.colors = torch.empty(n, 3, device="cuda")
    p_r._net_forward(s, p_r.store.ensure_packed())
    torch.cuda.synchronize()
    e_sdf = (got_sdf.double() - ref[:, :1]).abs().max().item()
    e_sdf2 = (s.sdf.double() - ref[:, 0]).abs().max().item()
    e_feat = (s.feat.double() - ref[:, 1:]).abs().max().item()
    e_eager = (ref32.double() - ref).abs().max().item()
    e_g = (s.normals.double() - gref).abs().max().item()
    e_g_eager = (g32.double() - gref).abs().max().item()
    print(f"sdf {e_sdf:.2e}/{e_sdf2:.2e} feat {e_feat:.2e} (eager {e_eager:.2e}); fd-grad {e_g:.2e} (eager {e_g_eager:.2e})")
    assert max(e_sdf, e_sdf2, e_feat) < max(2e-6, 3 * e_eager)
    # fp32 sdf differences / 2e-3: both implementations sit at ~1e-4 absolute
    assert e_g < max(5e-4, 3 * e_g_eager)


def test_color_forward_matches_oracle():
    o_r, p_r = make_hash_pair()
    n_rays, per = 37, 8
    n = n_rays * per
    g = torch.Generator(device="cpu").manual_seed(1)
    feat = torch.randn(n, 13, generator=g).cuda()
    nrm = torch.randn(n, 3, generator=g).cuda()
    d = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=g), dim=1).cuda()
    col = o_r.color_network
    with torch.no_grad():
        col.double()
        ref = col(None, nrm.double(), d.double()[:, None, :].expand(n_rays, per, 3).reshape(-1, 3), feat.double())
        col.float()
    out = torch.empty(n, 3, device="cuda")
    from dynhor_amd import _lib
    _lib.check(_lib.lib().dh_hash_color_forward(_lib.ptr(p_r.store.ensure_packed()), _lib.ptr(feat), _lib.ptr(nrm),
                                                _lib.ptr(d), per, n, _lib.ptr(out), None, _lib.stream()))
    torch.cuda.synchronize()
    err = (out.double() - ref).abs().max().item()
    print("colour err", err)
    assert err < 2e-6


def _rays(B, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    o = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1) * 2.5
    tgt = torch.randn(B, 3, generator=g) * 0.25
    d = torch.nn.functional.normalize(tgt - o, dim=1)
    rays_o, rays_d = o.cuda(), d.cuda()
    near, far = O.near_far_from_sphere(rays_o, rays_d)
    return rays_o, rays_d, near, far


def _oracle_grads(o_r, rays_o, rays_d, near, far, z, car, loss_fn, dtype):
    mods = (o_r.sdf_network, o_r.deviation_network, o_r.color_network)
    for m in mods:
        m.to(dtype); m.zero_grad()
    c = lambda t: t.to(dtype)
    out = o_r.render(c(rays_o), c(rays_d), c(near), c(far), cos_anneal_ratio=car, z_vals=c(z))
    loss = loss_fn(out)
    loss.backward()
    gflat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).double()
                       for m in mods for p in m.parameters()])
    for m in mods:
        m.float()
    return out, loss.item(), gflat


def test_render_forward_and_backward_match_oracle():
    o_r, p_r = make_hash_pair(seed=5)
    B, car = 48, 0.6
    rays_o, rays_d, near, far = _rays(B, seed=2)
    g = torch.Generator(device="cpu").manual_seed(9)
    t_rand = torch.rand(B, 1, generator=g).cuda()
    tgt = torch.rand(B, 3, generator=g).cuda()
    with torch.no_grad():
        z = o_r.sample_z(rays_o, rays_d, near, far, t_rand=t_rand)

    def loss_fn(out):
        t = tgt.to(out["color_fine"].dtype)
        return ((out["color_fine"] - t).abs().mean() + 0.1 * out["gradient_error"]
                + 0.1 * torch.nn.functional.binary_cross_entropy(out["weight_sum"].clip(1e-3, 1 - 1e-3),
                                                                 (t[:, :1] > 0.5).to(t.dtype)))

    ref, ref_loss, gref = _oracle_grads(o_r, rays_o, rays_d, near, far, z, car, loss_fn, torch.float64)
    eager, eager_loss, geager = _oracle_grads(o_r, rays_o, rays_d, near, far, z, car, loss_fn, torch.float32)

    out = p_r.render(rays_o, rays_d, near, far, cos_anneal_ratio=car, z_vals=z)
    loss = loss_fn(out)
    loss.backward()
    torch.cuda.synchronize()
    for k in ("color_fine", "weight_sum", "weights", "gradients", "gradient_error"):
        e = (out[k].detach().double() - ref[k].detach()).abs().max().item()
        e_eager = (eager[k].detach().double() - ref[k].detach()).abs().max().item()
        print(k, f"hip {e:.2e}  eager fp32 oracle {e_eager:.2e}")
        assert e < max(2e-5, 3 * e_eager), k
    got = p_r.store.grad_flat.double()
    st = p_r.store
    ntab = st.slices[0][2]
    for name, sl in (("table", slice(0, ntab)), ("mlps", slice(ntab, st.n))):
        den = gref[sl].norm().item()
        e = (got[sl] - gref[sl]).norm().item() / den
        e_eager = (geager[sl] - gref[sl]).norm().item() / den
        print(f"grad[{name}] rel err hip {e:.2e}  (eager fp32 oracle {e_eager:.2e})  |g| {den:.3e}")
        assert e < max(1e-3, 3 * e_eager), name
    assert abs(loss.item() - ref_loss) < 1e-4 * max(1.0, abs(ref_loss))


def test_sampler_runs_on_hash_sdf():
    """Hierarchical up-sampling driven by the hash sdf kernel: same z as the oracle's sampler within the sampler's own
    fp32 conditioning (compared like tests/test_gpu_render_forward.py does for the NeuS family)."""
    o_r, p_r = make_hash_pair(seed=8, table_scale=0.02)
    B = 64
    rays_o, rays_d, near, far = _rays(B, seed=4)
    t_rand = torch.rand(B, 1).cuda()
    with torch.no_grad():
        z_ref = o_r.sample_z(rays_o, rays_d, near, far, t_rand=t_rand)
    z = p_r.sample_z(rays_o, rays_d, near, far, t_rand=t_rand)
    torch.cuda.synchronize()
    assert z.shape == z_ref.shape
    assert (z[:, 1:] >= z[:, :-1]).all()
    frac_close = ((z - z_ref).abs() < 1e-4).float().mean().item()
    print("fraction of samples within 1e-4 of the oracle's:", frac_close)
    assert frac_close > 0.97


def test_fused_training_reduces_loss_and_tracks_oracle():
    """A few fused iterations (sample -> render -> loss -> backward -> Adam on the 24 M-entry flat vector) next to the
    oracle's train_step on the same rays / jitter: losses agree step by step within the fp32 noise of the family."""
    from dynhor_amd.dataset import Dataset
    ds = Dataset.from_synthetic(n_frames=2, H=64, W=64, seed=3, device="cuda:0")
    o_r, p_r = make_hash_pair(seed=11, table_scale=1e-4, jitter=0.0, n_samples=16, n_importance=16)
    mods = (o_r.sdf_network, o_r.deviation_network, o_r.color_network)
    opt = torch.optim.Adam([p for m in mods for p in m.parameters()], lr=1e-3)
    g = torch.Generator(device="cpu").manual_seed(0)
    losses_h, losses_o = [], []
    px = torch.randint(0, ds.W, [128], generator=g).cuda()
    py = torch.randint(0, ds.H, [128], generator=g).cuda()
    for it in range(6):
        t_rand = torch.rand(128, 1, generator=g).cuda()
        rays = ds.gen_rays_at_pixels(0, px, py)
        near, far = ds._last_near_far
        stats = p_r.train_step_core(rays, near, far, ds.R[0], 0.5, 0.1, 0.1, 0.0, t_rand=t_rand)
        p_r.store.adam_step(1e-3)
        lo = O.train_step(o_r, opt, rays, 0.5, 0.1, 0.1, 0.0, R=ds.R[0], t_rand=t_rand)
        losses_h.append(stats[0].item()); losses_o.append(lo["loss"].item())
    torch.cuda.synchronize()
    print("hip   ", ["%.5f" % v for v in losses_h])
    print("oracle", ["%.5f" % v for v in losses_o])
    for a, b in zip(losses_h, losses_o):
        assert abs(a - b) < 2e-2 * max(1.0, abs(b))
    assert abs(losses_h[0] - losses_o[0]) < 1e-4
    assert losses_h[-1] < losses_h[0]


def test_bad_arguments_are_rejected():
    from dynhor_amd import _lib
    L = _lib.lib()
    x = torch.zeros(4, 3, device="cuda")
    out = torch.zeros(4, device="cuda")
    o_r, p_r = make_hash_pair()
    st = p_r.store
    assert L.dh_hash_sdf_nograd(_lib.ptr(st.flat), _lib.ptr(st.ensure_packed()), _lib.ptr(x), -1, 1.0, _lib.ptr(out), None) != 0
    assert L.dh_hash_sdf_nograd(_lib.ptr(st.flat), _lib.ptr(st.packed), _lib.ptr(x), 4, 0.0, _lib.ptr(out), None) != 0
    assert L.dh_hash_sdf_nograd(None, _lib.ptr(st.packed), _lib.ptr(x), 4, 1.0, _lib.ptr(out), None) != 0
    assert L.dh_hash_color_forward(_lib.ptr(st.packed), _lib.ptr(x), _lib.ptr(x), _lib.ptr(x), 3, 4, _lib.ptr(out), None, None) != 0
    with pytest.raises(TypeError):
        from dynhor_amd.hash_fields import HashNeuSRenderer
        HashNeuSRenderer(None, o_r.sdf_network, o_r.deviation_network, o_r.color_network, n_samples=16, n_importance=16,
                         n_outside=0, up_sample_steps=2, perturb=1.0)


def test_runner_trains_the_hash_family(tmp_path):
    """model.family = "hash" through the same Runner: full-size batch (2048 x 128), loss goes down, checkpoint round trip
    in the upstream layout (sdf_network_fine carries encoding.table), validation image + mesh extraction work."""
    import os
    from dynhor_amd.runner import Runner
    conf = {"seq_name": "t", "exp_name": "hash",
            "data_info": {"synthetic": {"n_frames": 4, "H": 128, "W": 128, "seed": 11}},
            "train": {"batch_size": 2048, "learning_rate": 5e-3, "normal_weight": 0.05, "report_freq": 10,
                      "save_freq": 10 ** 9, "val_freq": 0, "warm_up_end": 50, "end_iter": 1000},
            "model": {"family": "hash"}}
    r = Runner(conf=conf, device="cuda:0", exp_root=str(tmp_path))
    first = None
    for _ in range(40):
        s = r.train_iteration()
        first = first if first is not None else float(s[0])
    last = float(s[0])
    assert last == last and last < first, (first, last)
    st = r.renderer.last_state
    assert st.weights.shape == (2048, 128) and torch.isfinite(st.color).all() and (st.wsum <= 1.0 + 1e-4).all()
    path = r.save_checkpoint()
    ck = torch.load(path, weights_only=False)
    assert list(ck["sdf_network_fine"].keys())[0] == "encoding.table"
    flat_before = r.store.flat.clone()
    r.store.flat.add_(1.0); r.store.bump()
    r.load_checkpoint(path)
    assert torch.equal(r.store.flat, flat_before)
    psnr = r.validate_image(idx=0, resolution_level=4)
    assert psnr == psnr and psnr > 0
    verts, faces = r.validate_mesh(resolution=48)
    assert verts.shape[0] > 100 and faces.shape[0] > 200
    assert os.path.exists(os.path.join(r.base_exp_dir, "meshes", "{:0>8d}.ply".format(r.iter_step)))


@pytest.mark.parametrize("B,ns,ni,steps", [(1, 8, 8, 2), (5, 8, 8, 2), (3, 10, 0, 1), (67, 6, 6, 1)])
def test_ragged_sizes_backward(B, ns, ni, steps):
    """Sample counts that are multiples of nothing the kernels tile by (row quads of the dW GEMM, 256-thread blocks, 16-sample
    waves of the scatter): gradients still match the fp64 oracle next to the eager fp32 oracle."""
    o_r, p_r = make_hash_pair(seed=21, n_samples=ns, n_importance=ni, up_sample_steps=steps)
    rays_o, rays_d, near, far = _rays(B, seed=B)
    g = torch.Generator(device="cpu").manual_seed(B)
    t_rand = torch.rand(B, 1, generator=g).cuda()
    tgt = torch.rand(B, 3, generator=g).cuda()
    with torch.no_grad():
        z = o_r.sample_z(rays_o, rays_d, near, far, t_rand=t_rand)
    assert z.shape == (B, ns + ni)

    def loss_fn(out):
        return ((out["color_fine"] - tgt.to(out["color_fine"].dtype)).abs().mean() + 0.1 * out["gradient_error"]
                + 0.05 * out["weight_sum"].mean())

    _, ref_loss, gref = _oracle_grads(o_r, rays_o, rays_d, near, far, z, 0.3, loss_fn, torch.float64)
    _, _, geager = _oracle_grads(o_r, rays_o, rays_d, near, far, z, 0.3, loss_fn, torch.float32)
    out = p_r.render(rays_o, rays_d, near, far, cos_anneal_ratio=0.3, z_vals=z)
    loss = loss_fn(out)
    loss.backward()
    torch.cuda.synchronize()
    got = p_r.store.grad_flat.double()
    assert torch.isfinite(got).all()
    e = (got - gref).norm().item() / gref.norm().item()
    e_eager = (geager - gref).norm().item() / gref.norm().item()
    print(f"B={B} n={ns + ni}: grad rel err hip {e:.2e} (eager fp32 oracle {e_eager:.2e})")
    assert e < max(1e-3, 3 * e_eager)
    assert abs(loss.item() - ref_loss) < 1e-4 * max(1.0, abs(ref_loss))
