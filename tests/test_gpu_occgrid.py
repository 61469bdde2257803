"""GPU: occupancy-grid ray marching with packed variable-length rays for the hash family (BASELINE.json configs[3]; SURVEY.md
section 8f n3; specification oracle/occgrid_oracle.py -- instant-nsr-pl / nerfacc behaviour restated, parity unpinned):
the marcher selects exactly the oracle's samples, the packed render scan and its adjoint match the oracle's autograd, the
fused training step on packed rays matches the oracle end to end, and the Runner trains / validates with sampler = occgrid."""
import pytest
import torch

from oracle import hashgrid_oracle as HO
from oracle import neus_oracle as O
from oracle import occgrid_oracle as G

pytestmark = pytest.mark.gpu


def _rays(B, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    o = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1) * (2.0 + 0.5 * torch.rand(B, 1, generator=g))
    d = torch.nn.functional.normalize(-o + 0.35 * torch.randn(B, 3, generator=g), dim=-1)
    o, d = o.cuda().contiguous(), d.cuda().contiguous()
    near, far = O.near_far_from_sphere(o, d)
    u = torch.rand(B, generator=g).cuda()
    return o, d, near.contiguous(), far.contiguous(), u


def _blob_grid(res, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    grid = G.OccupancyGrid(res=res, radius=1.0, device="cuda")
    ax = (torch.arange(res) + 0.5) / res * 2 - 1
    x, y, z = torch.meshgrid(ax, ax, ax, indexing="ij")
    occ = torch.zeros(res, res, res, dtype=torch.bool)
    for _ in range(6):
        c = (torch.rand(3, generator=g) - 0.5) * 1.0
        r = 0.12 + 0.2 * torch.rand(1, generator=g).item()
        occ |= ((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) < r * r
    grid.binary = occ.reshape(-1).cuda()
    return grid


def _hip_march(o, d, near, far, u, grid, step, max_samples=128):
    from types import SimpleNamespace
    from dynhor_amd import _lib
    from dynhor_amd.renderer import _p
    L = _lib.lib()
    B = o.shape[0]
    binary = grid.binary.to(torch.uint8).contiguous()
    stepf = float(torch.tensor(step, dtype=torch.float32)); half = float(torch.tensor(0.5 * step, dtype=torch.float32))
    cnt = torch.empty(B, dtype=torch.int32, device="cuda")
    nr, fr = near.view(-1).contiguous(), far.view(-1).contiguous()
    _lib.check(L.dh_march_count(_p(o), _p(d), _p(nr), _p(fr), _p(u), _p(binary), grid.res, grid.radius, stepf, half, max_samples, B,
                                _p(cnt), _lib.stream()))
    cs = torch.cumsum(cnt, 0, dtype=torch.int64); off = (cs - cnt).contiguous(); N = int(cs[-1])
    m = SimpleNamespace(N=N, off=off, cnt=cnt, step=stepf, t_start=torch.empty(N, device="cuda"), pts=torch.empty(N, 3, device="cuda"),
                        dirs=torch.empty(N, 3, device="cuda"), ray_idx=torch.empty(N, dtype=torch.int32, device="cuda"))
    _lib.check(L.dh_march_emit(_p(o), _p(d), _p(nr), _p(fr), _p(u), _p(binary), grid.res, grid.radius, stepf, half, max_samples, B,
                               _p(off), None, _p(m.t_start), _p(m.pts), _p(m.dirs), _p(m.ray_idx), _lib.stream()))
    return m


@pytest.mark.parametrize("B,res,step,cap", [(513, 64, 0.0135, 128), (2048, 128, 1.732 * 2 / 512, 128), (37, 32, 0.02, 16)])
def test_marcher_selects_exactly_the_oracle_samples(B, res, step, cap):
    o, d, near, far, u = _rays(B, seed=B)
    grid = _blob_grid(res, seed=res)
    stepf = float(torch.tensor(step, dtype=torch.float32))
    ref = G.march(o, d, near, far, u, grid, stepf, max_samples=cap)
    m = _hip_march(o, d, near, far, u, grid, step, cap)
    assert int(ref["cnt"].sum()) > 5 * B // 4 or cap == 16, "the case must actually produce samples"
    assert torch.equal(m.cnt.long(), ref["cnt"]), "per-ray sample counts"
    assert torch.equal(m.off, ref["off"]) and m.N == ref["t_start"].shape[0]
    assert torch.equal(m.t_start, ref["t_start"]), "interval starts, bit for bit (same fp32 expression order)"
    assert torch.equal(m.ray_idx.long(), ref["ray_idx"])
    tm = m.t_start + float(torch.tensor(0.5 * step, dtype=torch.float32))
    assert torch.equal(m.pts, o[m.ray_idx.long()] + d[m.ray_idx.long()] * tm[:, None])
    assert torch.equal(m.dirs, d[m.ray_idx.long()])
    assert (m.cnt <= cap).all() and (m.cnt == cap).any() == bool(ref["truncated"].any())
    # every sample's mid-point lies in an occupied cell, inside [near, far]
    assert grid.query(m.pts).all()
    assert (m.t_start >= near.view(-1)[m.ray_idx.long()] - 1e-6).all() and (m.t_start + stepf <= far.view(-1)[m.ray_idx.long()] + 1e-6).all()
    # no stratified offsets (validation): u = null means 0.5
    m2 = _hip_march(o, d, near, far, None, grid, step, cap)
    ref2 = G.march(o, d, near, far, torch.full_like(u, 0.5), grid, stepf, max_samples=cap)
    assert torch.equal(m2.t_start, ref2["t_start"])


@pytest.mark.parametrize("step,cap", [(0.01, 128), (0.0025, 1024)])
def test_packed_render_scan_and_adjoint_match_oracle(step, cap):
    """cap 1024: segments longer than one 128-sample trip of the wave (transmittance / suffix carries across trips)."""
    from dynhor_amd import _lib
    from dynhor_amd.renderer import _p
    L = _lib.lib()
    B = 300
    o, d, near, far, u = _rays(B, seed=7)
    grid = _blob_grid(64, seed=3)
    m = _hip_march(o, d, near, far, u, grid, step, cap)
    N = m.N
    if cap > 128:
        assert (m.cnt > 128).any() and (m.cnt > 256).any(), "the case must contain multi-trip rays"
        ref_m = G.march(o, d, near, far, u, grid, float(torch.tensor(step, dtype=torch.float32)), max_samples=cap)
        assert torch.equal(m.t_start, ref_m["t_start"]) and torch.equal(m.cnt.long(), ref_m["cnt"])
    g = torch.Generator(device="cpu").manual_seed(1)
    sdf = (torch.randn(N, generator=g) * (0.05 if cap == 128 else 0.2) + (0.0 if cap == 128 else 0.15)).cuda()   # long rays: keep T alive
    normals = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).cuda() * (0.7 + 0.6 * torch.rand(N, 1, generator=g).cuda())
    colors = torch.rand(N, 3, generator=g).cuda()
    inv_s = torch.tensor([35.0], device="cuda")
    car, bg = 0.3, torch.tensor([0.2, 0.5, 0.9], device="cuda")
    w = torch.empty(N, device="cuda"); cdf = torch.empty(N, device="cuda"); ins = torch.empty(N, device="cuda")
    col = torch.empty(B, 3, device="cuda"); ws = torch.empty(B, 1, device="cuda"); wm = torch.empty(B, 1, device="cuda")
    eik = torch.empty(B, 2, device="cuda"); nm = torch.empty(B, 3, device="cuda")
    _lib.check(L.dh_render_scan_fwd_packed(_p(o), _p(d), _p(m.t_start), _p(sdf), _p(normals), _p(colors), _p(inv_s), car, m.step, _p(bg), B,
                                           _p(m.off), _p(m.cnt), _p(w), _p(col), _p(ws), _p(wm), _p(cdf), _p(ins), _p(eik), _p(nm), _lib.stream()))
    s64, n64, c64 = sdf.double().requires_grad_(True), normals.double().requires_grad_(True), colors.double().requires_grad_(True)
    ref = G.render_packed(m.pts.double(), s64, n64, c64, d.double(), m.ray_idx.long(), m.off, m.cnt.long(), float(m.step), inv_s.double(), car, bg.double())
    assert (col.double() - ref["color_fine"]).abs().max().item() < 2e-5
    assert (ws.double() - ref["weight_sum"]).abs().max().item() < 2e-5
    assert (w.double() - ref["weights"]).abs().max().item() < 2e-5
    assert (nm.double() - ref["normal_map"]).abs().max().item() < 5e-5
    ge = eik[:, 0].sum() / (eik[:, 1].sum() + 1e-5)
    assert abs(ge.item() - ref["gradient_error"].item()) < 1e-5
    empty = m.cnt == 0
    assert empty.any() and (ws[empty] == 0).all() and torch.allclose(col[empty], bg.expand(int(empty.sum()), 3))
    # adjoint: a random linear functional of the outputs
    gc = torch.randn(B, 3, generator=g).cuda(); gw = torch.randn(B, generator=g).cuda(); gn = torch.randn(B, 3, generator=g).cuda()
    ec = torch.tensor([0.37], device="cuda")
    functional = ((ref["color_fine"] * gc.double()).sum() + (ref["weight_sum"][:, 0] * gw.double()).sum()
                  + (ref["normal_map"] * gn.double()).sum()
                  + ec.double()[0] * (eik[:, 1].sum().double() + 1e-5) * ref["gradient_error"])
    functional.backward()
    d_sdf = torch.empty(N, device="cuda"); d_n = torch.empty(N, 3, device="cuda"); d_c = torch.empty(N, 3, device="cuda"); d_is = torch.empty(B, device="cuda")
    null = _lib.ptr(torch.empty(1, device="cuda")).__class__(0)
    _lib.check(L.dh_render_scan_bwd_packed(_p(o), _p(d), _p(m.t_start), _p(sdf), _p(normals), _p(colors), _p(inv_s), car, m.step, _p(bg), B,
                                           _p(m.off), _p(m.cnt), _p(gc), _p(gw), null, null, _p(gn), _p(ec), _p(d_sdf), _p(d_n), _p(d_c),
                                           _p(d_is), _lib.stream()))
    for name, got, want in (("d_sdf", d_sdf, s64.grad), ("d_normals", d_n, n64.grad), ("d_colors", d_c, c64.grad)):
        rel = ((got.double() - want).norm() / want.norm()).item()
        print(name, "rel err", f"{rel:.2e}")
        assert rel < 1e-4, name


def _oracle_hash_models(like_runner, dev):
    sdf, col = HO.build_models(seed=1234, device=dev)
    var = O.SingleVarianceNetwork(0.3).to(dev)
    sdf.load_state_dict(like_runner.sdf_network.state_dict()); col.load_state_dict(like_runner.color_network.state_dict())
    var.load_state_dict(like_runner.deviation_network.state_dict())
    return sdf, col, var


def _occ_runner(root):
    from dynhor_amd.runner import Runner
    conf = {"seq_name": "occ", "exp_name": "e", "data_info": {"synthetic": {"n_frames": 6, "H": 96, "W": 96, "seed": 11}},
            "train": {"batch_size": 512, "normal_weight": 0.05, "learning_rate": 5e-3, "report_freq": 10 ** 9, "save_freq": 10 ** 9,
                      "val_freq": 0, "warm_up_end": 20, "end_iter": 2000},
            "model": {"family": "hash", "hash_renderer": {"sampler": "occgrid", "march_samples_per_ray": 256, "grid_res": 64,
                                                         "grid_update_every": 8}}}
    return Runner(conf=conf, device="cuda:0", exp_root=str(root))


def test_nerfacc_refresh_schedule_matches_the_oracle_restatement(tmp_path):
    """hash_renderer.grid_refresh = "nerfacc": all cells while the training step is below grid_warmup_steps, afterwards a uniform
    quarter + the occupied cells (oracle/occgrid_oracle.py quarter_refresh_mask).  The product's device-side mask equals the
    oracle's from the same uniforms; unselected cells keep their occupancy (no decay); the inclusion rates are nerfacc's."""
    from dynhor_amd.hash_fields import refresh_mask
    r = _occ_runner(tmp_path)
    ren = r.renderer
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(5)
    cells = ren.grid.res ** 3
    ren.update_grid(jitter=torch.rand(cells, 3, device=dev, generator=g))            # step 0: every cell
    og = G.OccupancyGrid(res=ren.grid.res, radius=1.0, device=dev)
    og.occ, og.binary = ren.grid.occ.clone(), ren.grid.binary.bool().clone()
    occ_before = ren.grid.occ.clone()
    u = torch.rand(cells, device=dev, generator=g)
    mask = refresh_mask(ren.grid.binary, u)
    assert torch.equal(mask, og.quarter_refresh_mask(u))
    k = int(ren.grid.binary.sum())
    frac_uni = (u < 1.0 - torch.exp(torch.tensor(-0.25))).float().mean().item()
    sel_occ = mask[ren.grid.binary.bool()].float().mean().item()
    print(f"occupied {k} of {cells}; selected {mask.float().mean().item():.4f} of all cells ({frac_uni:.4f} by the uniform draw), {sel_occ:.4f} of the occupied ones")
    assert abs(frac_uni - 0.2212) < 0.005
    n = cells // 4
    want = 1.0 if k <= n else 0.2212 + (1 - 0.2212) * (1 - 2.718281828 ** (-n / k))
    assert abs(sel_occ - want) < 0.01
    jit = torch.rand(cells, 3, device=dev, generator=g)
    ren.grid.update(lambda p: ren.sdf(p), ren.store.inv_s(), ren.march_step, jitter=jit, refresh="quarter", select=u)
    with torch.no_grad():
        sdf = ren.sdf(og.cell_points(jit).contiguous()).reshape(-1)
        og.update(G.occ_alpha(sdf, ren.store.inv_s(), ren.march_step), mask=mask)
    assert torch.equal(ren.grid.occ, og.occ) and torch.equal(ren.grid.binary.bool(), og.binary)
    assert torch.equal(ren.grid.occ[~mask], occ_before[~mask])                      # untouched cells: not even decayed
    # the schedule: below the warm-up every update is "all", afterwards "quarter"
    ren.grid_refresh, ren.grid_warmup_steps = "nerfacc", 4
    seen = []
    orig = ren.grid.update
    ren.grid.update = lambda *a, **kw: (seen.append(kw.get("refresh")), orig(*a, **kw))[1]
    for it in (0, 3, 4, 100):
        ren._march_iter = it
        ren.update_grid()
    assert seen == ["all", "all", "quarter", "quarter"]
    with pytest.raises(ValueError):
        orig(lambda p: ren.sdf(p), ren.store.inv_s(), ren.march_step, refresh="half")


def test_fused_training_step_on_packed_rays_matches_oracle(tmp_path):
    r = _occ_runner(tmp_path)
    ren, ds = r.renderer, r.dataset
    dev = torch.device("cuda:0")
    # the geometric initialisation zeroes the encoding columns of lin0 (the table would get an exactly-zero gradient) and the
    # table itself is ~1e-4: jitter both so every parameter group carries signal
    gj = torch.Generator(device=dev); gj.manual_seed(17)
    with torch.no_grad():
        r.sdf_network.lin0.weight_v.add_(0.05 * torch.randn(r.sdf_network.lin0.weight_v.shape, device=dev, generator=gj))
        r.sdf_network.encoding.table.add_(0.005 * torch.randn(r.sdf_network.encoding.table.shape, device=dev, generator=gj))
    r.store.bump()
    o_sdf, o_col, o_var = _oracle_hash_models(r, dev)
    g = torch.Generator(device=dev); g.manual_seed(3)
    jitter = torch.rand(ren.grid.res ** 3, 3, device=dev, generator=g)
    ren.update_grid(jitter=jitter)
    frac = ren.grid.occupied_fraction()
    # the grid the product built == the oracle's update from the same jitter (sdf through the oracle network)
    og = G.OccupancyGrid(res=ren.grid.res, radius=1.0, device=dev)
    with torch.no_grad():
        inv_s = o_var(torch.zeros(1, 3, device=dev))[:, :1].clip(1e-6, 1e6).reshape(())
        og.update(G.occ_alpha(o_sdf.sdf(og.cell_points(jitter)).reshape(-1), inv_s, ren.march_step))
    agree = (og.binary == ren.grid.binary.bool()).float().mean().item()
    print(f"occupied fraction {frac:.3f}; grid agreement with the oracle {agree:.5f}")
    assert 0.0 < frac < 0.9 and agree > 0.999          # a cell can flip only where fp32 sdf rounding straddles the threshold
    B, frame, car = 512, 2, 0.2
    rays = ds.gen_random_rays_at(frame, B, generator=g)
    near, far = ds._last_near_far
    u = torch.rand(B, 1, device=dev, generator=g)
    ren._march_iter = 1                                   # no grid update inside the step
    stats = ren.train_step_core(rays, near, far, ds.R[frame], car, 0.1, 0.1, 0.05, t_rand=u)
    torch.cuda.synchronize()
    m = ren.last_state.m
    lm = ren.last_march
    N = int(m.n_dev)
    assert N > B and lm["samples"] == N and lm["capacity"] == B * 128 and lm["per_ray_cap"] in (128, 256, 512, 1024)
    assert lm["rays_truncated"] == 0 or lm["per_ray_cap"] < 1024
    # the packed arrays live at the fixed capacity; rows >= N belong to nobody
    from types import SimpleNamespace
    m = SimpleNamespace(N=N, pts=m.pts[:N], dirs=m.dirs[:N], ray_idx=m.ray_idx[:N], off=m.off, cnt=m.cnt, step=m.step)
    # oracle on the SAME packed samples (the marcher itself is checked above), fp64 networks
    for mod in (o_sdf, o_col, o_var):
        mod.double(); mod.zero_grad()
    pts = m.pts.double()
    out = o_sdf(pts)
    sdf = out[:, 0]                                       # forward = cat[sdf, feature(13)] (oracle/hashgrid_oracle.py:118)
    nrm = o_sdf.gradient(pts).reshape(-1, 3)
    dirs = m.dirs.double()
    colr = o_col(pts, nrm, dirs, out[:, 1:])
    inv_s = o_var(torch.zeros(1, 3, device=dev, dtype=torch.float64))[:, :1].clip(1e-6, 1e6).reshape(())
    ro = G.render_packed(pts, sdf, nrm, colr, rays[:, 3:6].double(), m.ray_idx.long(), m.off, m.cnt.long(), float(m.step), inv_s, car)
    rd = {"color_fine": ro["color_fine"], "weight_sum": ro["weight_sum"], "gradient_error": ro["gradient_error"],
          "gradients": ro["normal_map"][:, None, :], "weights": torch.ones(B, 1, dtype=torch.float64, device=dev)}
    r64 = rays.double()
    ref = O.neus_losses(rd, r64[:, 6:9], r64[:, 9:10], r64[:, 10:11], 0.1, 0.1, 0.05, r64[:, 11:14], ds.R[frame].double())
    ref["loss"].backward()
    names = ["loss", "color_loss", "eikonal_loss", "mask_loss", "normal_loss", "psnr"]
    for i, k in enumerate(names):
        assert abs(stats[i].item() - float(ref[k])) < 1e-4 * max(1.0, abs(float(ref[k]))), (k, stats[i].item(), float(ref[k]))
    # gradient by parameter group (table: float atomics + fp32 finite-difference normals, as in tests/test_gpu_hash_family.py)
    st = r.store
    named = {}
    for mod, pre in ((o_sdf, "sdf."), (o_var, "var."), (o_col, "col.")):
        for n_, p in mod.named_parameters():
            named[pre + n_] = p.grad
    tab = named["sdf.encoding.table"] if "sdf.encoding.table" in named else [v for k, v in named.items() if "table" in k][0]
    (p0, off0, cnt0) = st.slices[0]
    got_tab = st.grad_flat[off0:off0 + cnt0].double().view_as(tab)
    rel_tab = ((got_tab - tab).norm() / tab.norm()).item()
    print(f"packed-ray step: loss {stats[0].item():.6f} (oracle {float(ref['loss']):.6f}); table grad rel {rel_tab:.2e}; samples/ray {m.N / B:.1f}")
    assert tab.norm().item() > 0 and rel_tab < 2e-3
    # the small MLPs' gradients (weight-norm folded), every slice after the table
    worst = 0.0
    flat_ref = torch.cat([named[k].reshape(-1) for k in named if "table" not in k and k.startswith("sdf.")]
                         + [named["var.variance"].reshape(-1)] + [named[k].reshape(-1) for k in named if k.startswith("col.")])
    got_rest = st.grad_flat[off0 + cnt0:].double()
    rel_rest = ((got_rest - flat_ref).norm() / flat_ref.norm()).item()
    print(f"MLP + variance gradients rel {rel_rest:.2e}")
    assert rel_rest < 2e-3
    for mod in (o_sdf, o_col, o_var):
        mod.float()


def test_runner_trains_and_validates_with_occupancy_grid_sampler(tmp_path):
    r = _occ_runner(tmp_path)
    first = None
    for _ in range(40):
        s = r.train_iteration()
        first = first if first is not None else float(s[0])
    assert float(s[0]) < first and torch.isfinite(r.store.flat).all()
    lm = r.renderer.last_march
    assert 0 < lm["samples_per_ray"] <= 128 and lm["samples"] <= lm["capacity"]
    occ_frac = float(r.renderer.grid.binary.float().mean())
    assert 0.0 < occ_frac < 0.6, "the grid prunes empty space"
    psnr = r.validate_image(idx=0, resolution_level=2)
    assert psnr == psnr and psnr > 5
    with pytest.raises(ValueError):
        r.renderer.train_step_core(r._last_rays, *r.dataset._last_near_far, r.dataset.R[0], 0.1, ray_grads=True)


def test_runner_reports_and_validates_in_chunks_without_touching_the_training_buffers(tmp_path):
    """Runner.report() indexed the packed state's flat cdf as [B, n] (IndexError after report_freq iterations: the
    occupancy-grid path could not be trained through the CLI), and a validation pass with a few chunk sizes evicted -- and
    re-allocated -- the training batch's sample buffers that last_state still points into."""
    r = _occ_runner(tmp_path)
    r.report_freq = 2
    r.train(n_iters=4)                                      # reports twice; must not raise
    recs = [x for x in r.scalars if "Statistics/cdf" in x]
    assert len(recs) >= 2 and all(0.0 <= x["Statistics/cdf"] <= 1.0 and 0.0 <= x["Statistics/weight_max"] <= 1.0 for x in recs)
    assert r._board is None, "train() closes the scalar writer"
    r.train_iteration()
    st = r.renderer.last_state
    ptr, w0 = st.weights.data_ptr(), st.weights.clone()
    ds = r.dataset
    rays, _, _ = ds.gen_rays_at(0, 2)
    near, far = ds._last_near_far
    for n in (700, 300, 1000, 55):                          # ragged validation chunks of several sizes
        o, d = rays[:n, :3].contiguous(), rays[:n, 3:6].contiguous()
        c, _ = r.renderer.render_rays(o, d, near[:n], far[:n], 0.5, None, want_nmap=False)
        assert c.shape == (n, 3) and torch.isfinite(c).all()
    assert torch.equal(st.weights, w0), "inference chunks must not overwrite the training step's state"
    r.train_iteration()
    assert r.renderer.last_state.weights.data_ptr() == ptr, "the training buffers are allocated once"


def test_device_side_count_matches_exact_size_launch():
    """the packed stages take the sample count from the DEVICE (n_active) on buffers laid out for a fixed
    capacity.  Same samples through (a) exact-size launches (n = N, n_active = null) and (b) capacity launches (n = cap >
    N, n_active -> N) with the rows past N poisoned with NaN: identical outputs on the first N rows, identical weight
    gradients except float-atomic order in the table."""
    import ctypes
    from dynhor_amd import _lib
    from dynhor_amd.hash_fields import HashNeuSRenderer, build_hash_models
    from dynhor_amd.renderer import _p
    dev = torch.device("cuda:0")
    L = _lib.lib()
    sdf, var, col = build_hash_models(seed=5, device=dev)
    ren = HashNeuSRenderer(None, sdf, var, col, 64, 64, 0, 4, 1.0, device=dev)
    g = torch.Generator(device=dev); g.manual_seed(1)
    with torch.no_grad():
        sdf.lin0.weight_v.add_(0.05 * torch.randn(sdf.lin0.weight_v.shape, device=dev, generator=g))
        sdf.encoding.table.add_(0.005 * torch.randn(sdf.encoding.table.shape, device=dev, generator=g))
    ren.store.bump()
    st = ren.store
    packed = st.ensure_packed()
    N, cap = 1000, 1536                                   # N is not a multiple of 64 (ragged last tile), cap is a multiple of 8
    pts_n = (torch.rand(N, 3, device=dev, generator=g) * 1.2 - 0.6)
    dirs_n = torch.nn.functional.normalize(torch.randn(N, 3, device=dev, generator=g), dim=-1)
    d_sdf_n = torch.randn(N, device=dev, generator=g); d_nrm_n = torch.randn(N, 3, device=dev, generator=g) * 0.1
    d_col_n = torch.randn(N, 3, device=dev, generator=g)

    def run(n, n_act, poison):
        ws = torch.full((ren._workspace_need(n, False),), float("nan") if poison else 0.0, device=dev)
        fill = float("nan") if poison else 0.0
        pad = lambda t: torch.cat([t, torch.full((n - N,) + t.shape[1:], fill, device=dev)]).contiguous()
        pts, dirs = pad(pts_n), pad(dirs_n)
        o_sdf = torch.full((n,), fill, device=dev); o_feat = torch.full((n, 13), fill, device=dev)
        o_nrm = torch.full((n, 3), fill, device=dev); o_col = torch.full((n, 3), fill, device=dev)
        na = None if n_act is None else torch.tensor([n_act], dtype=torch.int64, device=dev)
        _lib.check(L.dh_hash_geo_forward(_p(st.flat), _p(packed), _p(pts), n, 1.0, 1e-3, _p(ws), 1, _p(o_sdf), _p(o_feat), _p(o_nrm),
                                         _p(na), _lib.stream()))
        _lib.check(L.dh_hash_color_forward(_p(packed), _p(o_feat), _p(o_nrm), _p(dirs), 1, n, _p(o_col), _p(na), _lib.stream()))
        d_feat = torch.full((n, 13), fill, device=dev)
        d_nrm = pad(d_nrm_n)
        _lib.check(L.dh_hash_color_backward(_p(packed), _p(o_feat), _p(o_nrm), _p(dirs), _p(pad(d_col_n)), 1, n, _p(ws), _p(d_feat),
                                            _p(d_nrm), _p(na), _lib.stream()))
        _lib.check(L.dh_hash_geo_backward(_p(st.flat), _p(packed), _p(pts), _p(pad(d_sdf_n)), _p(d_feat), _p(d_nrm), n, 1.0, 1e-3,
                                          _p(ws), _p(na), _lib.stream()))
        grad = torch.zeros(st.n, device=dev)
        _lib.check(L.dh_hash_weight_grads(_p(st.flat), _p(packed), n, _p(ws), _p(grad), _p(na), _lib.stream()))
        torch.cuda.synchronize()
        return o_sdf[:N], o_feat[:N], o_nrm[:N], o_col[:N], d_feat[:N], grad

    a = run(N, None, poison=False)
    b = run(cap, N, poison=True)
    for x, y, name in zip(a[:5], b[:5], ("sdf", "feature", "normal", "colour", "d_feature")):
        assert torch.equal(x, y), name
    ga, gb = a[5], b[5]
    assert torch.isfinite(gb).all(), "no stale (NaN) row may reach a gradient"
    (_, off0, cnt0) = st.slices[0]
    # the slab partition of the weight-gradient reduction follows the row count, so the two launches sum their ~7,000 fp32
    # terms per entry in different orders: a few 1e-5 relative is fp32 accumulation noise on these O(1) random cotangents (the
    # gradients themselves are checked against the fp64 oracle in test_fused_training_step_on_packed_rays_matches_oracle)
    rel_mlp = ((ga[off0 + cnt0:].double() - gb[off0 + cnt0:].double()).norm() / ga[off0 + cnt0:].double().norm()).item()
    rel = ((ga[:cnt0].double() - gb[:cnt0].double()).norm() / ga[:cnt0].double().norm()).item()
    print(f"capacity launch vs exact-size launch: MLP gradients rel {rel_mlp:.2e}, table gradient rel {rel:.2e} (summation order only)")
    assert rel_mlp < 1e-4, rel_mlp
    assert rel < 1e-5, f"table gradient (float atomics: order only) {rel}"
    b2 = run(cap, N, poison=True)
    assert torch.equal(b[5][off0 + cnt0:], b2[5][off0 + cnt0:]), "same launch shape -> the MLP gradients are bitwise reproducible"
    # n_active = 0: nothing runs, gradients of the MLPs are zero
    c = run(cap, 0, poison=True)
    assert torch.isfinite(c[5]).all() and float(c[5][off0 + cnt0:].abs().sum()) == 0.0 and float(c[5][:cnt0].abs().sum()) == 0.0
    # capacity must be a multiple of 8 when a device count is given
    na = torch.tensor([5], dtype=torch.int64, device=dev)
    x = torch.zeros(1004 * 13, device=dev)
    assert L.dh_hash_color_backward(_p(packed), _p(x), _p(x), _p(x), _p(x), 1, 1004, _p(x), _p(x), _p(x), _p(na), _lib.stream()) != 0
    assert L.dh_hash_weight_grads(_p(st.flat), _p(packed), 1004, _p(x), _p(x), _p(na), _lib.stream()) != 0


def test_adaptive_per_ray_cap_and_no_host_sync_in_the_step(tmp_path):
    """The per-ray cap is the largest ladder rung whose total fits the fixed capacity, chosen on the device; marching with it
    equals the oracle's marcher at that cap; the training step never reads the count back."""
    r = _occ_runner(tmp_path)
    ren = r.renderer
    ds = r.dataset
    dev = r.device
    g = torch.Generator(device=dev); g.manual_seed(9)
    for _ in range(20):
        r.train_iteration()
    B = 777
    rays = ds.gen_random_rays_at(1, B, generator=g)
    near, far = ds._last_near_far
    u = torch.rand(B, 1, device=dev, generator=g)
    o, d = rays[:, :3].contiguous(), rays[:, 3:6].contiguous()
    m = ren.march(o, d, near, far, u)
    cap = int(m.cap_dev)
    assert cap in ren._cap_ladder and int(m.n_dev) <= m.cap == (B * ren.max_samples + 7) // 8 * 8
    # a larger rung (if any) would not have fitted
    idx = ren._cap_ladder.index(cap)
    if idx > 0:
        assert int(torch.minimum(m.cnt_raw, torch.tensor(ren._cap_ladder[idx - 1], device=dev)).sum()) > m.cap
    og = G.OccupancyGrid(res=ren.grid.res, radius=1.0, device=dev)
    og.binary = ren.grid.binary.bool()
    ref = G.march(o, d, near, far, u.view(-1), og, float(m.step), max_samples=cap)
    N = int(m.n_dev)
    assert torch.equal(m.cnt.long(), ref["cnt"]) and torch.equal(m.off, ref["off"]) and N == ref["t_start"].shape[0]
    assert torch.equal(m.t_start[:N], ref["t_start"]) and torch.equal(m.ray_idx[:N].long(), ref["ray_idx"])
    # the training step of either sampler never waits for the device: PyTorch's own synchronisation detector stays silent
    r.train_iteration()
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        for _ in range(3):
            r.train_iteration()
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert r.renderer.last_march["samples"] > 0          # the statistics are available afterwards (this read does synchronise)


def test_neus_training_step_never_synchronises(tmp_path):
    from dynhor_amd.runner import Runner
    conf = {"seq_name": "nosync", "exp_name": "e", "data_info": {"synthetic": {"n_frames": 4, "H": 64, "W": 64, "seed": 3}},
            "train": {"batch_size": 256, "normal_weight": 0.05, "report_freq": 10 ** 9, "save_freq": 10 ** 9, "val_freq": 0}}
    r = Runner(conf=conf, device="cuda:0", exp_root=str(tmp_path))
    for _ in range(2):
        r.train_iteration()
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        for _ in range(3):
            r.train_iteration()
    finally:
        torch.cuda.set_sync_debug_mode("default")
