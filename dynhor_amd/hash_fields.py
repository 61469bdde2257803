"""Hash-grid model family (BASELINE.json configs[3]): the instant-nsr-pl style geometry / texture
networks the reference README names as its direction (README.md:11,13 -- the code itself is on an unmounted branch, so
the shapes follow oracle/hashgrid_oracle.py's restatement; parity unpinned).

Parameter containers + the renderer that runs them through the fused HIP kernels (csrc/hash_mlp.hip):
  * HashSDFNetwork: 16x2 multiresolution hash grid + Linear(35,64) softplus(100) Linear(64,13), sphere initialised,
    normals by central finite differences.
  * SHRenderingNetwork: [feature(13), SH4(view dir)(16), normal(3)] -> 64 -> 64 -> 3.
  * HashNeuSRenderer: NeuSRenderer (same sampler, compositing, losses and fused training step) with the three network
    stages swapped for the hash-family kernels.
No eager forward: everything evaluates through the C ABI; the oracle is only used by tests.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import _lib
from .fields import ParamStore, SingleVarianceNetwork, _WNLinearParams
from .renderer import NeuSRenderer, _p

import ctypes as _ct
_NULLP = _ct.c_void_p(0)


class HashGridTable(nn.Module):
    """The [entries, 2] feature table (state_dict key ``encoding.table``), U(-1e-4, 1e-4) initialised."""

    def __init__(self):
        super().__init__()
        n = int(_lib.lib().dh_hashgrid_entries())
        self.table = nn.Parameter(torch.empty(n, 2).uniform_(-1e-4, 1e-4))


class HashSDFNetwork(nn.Module):
    def __init__(self, radius=1.0, n_hidden=64, feature_dim=13, sphere_init_radius=0.5, fd_eps=1e-3):
        super().__init__()
        if (n_hidden, feature_dim) != (64, 13):
            raise ValueError("unsupported HashSDFNetwork configuration for the gfx950 kernels (DH_ERR_UNSUPPORTED): "
                             "n_hidden=64, feature_dim=13 only")
        self.radius, self.fd_eps, self.feature_dim = float(radius), float(fd_eps), feature_dim
        self.encoding = HashGridTable()
        d_in = 3 + 32
        w0 = torch.zeros(n_hidden, d_in)
        b0 = torch.zeros(n_hidden)
        nn.init.normal_(w0[:, :3], 0.0, math.sqrt(2) / math.sqrt(n_hidden))
        w1 = torch.empty(feature_dim, n_hidden)
        b1 = torch.empty(feature_dim)
        nn.init.normal_(w1, mean=math.sqrt(math.pi) / math.sqrt(n_hidden), std=1e-4)
        nn.init.constant_(b1, -sphere_init_radius)
        self.lin0 = _WNLinearParams(w0, b0)
        self.lin1 = _WNLinearParams(w1, b1)


class SHRenderingNetwork(nn.Module):
    def __init__(self, feature_dim=13, n_hidden=64):
        super().__init__()
        if (feature_dim, n_hidden) != (13, 64):
            raise ValueError("unsupported SHRenderingNetwork configuration for the gfx950 kernels (DH_ERR_UNSUPPORTED)")
        dims = [feature_dim + 16 + 3, n_hidden, n_hidden, 3]
        for l in range(3):
            lin = nn.Linear(dims[l], dims[l + 1])
            setattr(self, "lin" + str(l), _WNLinearParams(lin.weight.data, lin.bias.data))


def hash_param_layout(net: int, layer: int):
    import ctypes
    b, g, v = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    o, i = ctypes.c_int32(), ctypes.c_int32()
    _lib.check(_lib.lib().dh_hash_param_layout(net, layer, ctypes.byref(b), ctypes.byref(g), ctypes.byref(v),
                                               ctypes.byref(o), ctypes.byref(i)))
    return b.value, g.value, v.value, o.value, i.value


class HashParamStore(ParamStore):
    """Flat vector of the hash family: table | geometry lin0, lin1 | variance | colour lin0..2."""

    def _layout(self, sdf_network, deviation_network, color_network):
        L = _lib.lib()
        slices = []
        _, _, toff, entries, feats = hash_param_layout(3, 0)
        assert tuple(sdf_network.encoding.table.shape) == (entries, feats)
        slices.append((sdf_network.encoding.table, toff, entries * feats))
        assert toff == 0, "the table leads the flat vector (the overlapped all-reduce reduces grad[:table_floats])"
        self.table_floats = entries * feats
        for net, mod, n_layers in ((0, sdf_network, 2), (2, color_network, 3)):
            for l in range(n_layers):
                b, g, v, out_dim, in_dim = hash_param_layout(net, l)
                lin = getattr(mod, "lin" + str(l))
                assert tuple(lin.weight_v.shape) == (out_dim, in_dim), (net, l, lin.weight_v.shape)
                slices += [(lin.bias, b, out_dim), (lin.weight_g, g, out_dim), (lin.weight_v, v, out_dim * in_dim)]
        _, _, voff, _, _ = hash_param_layout(1, 0)
        slices.append((deviation_network.variance, voff, 1))
        return int(L.dh_hash_num_params()), slices, voff, int(L.dh_hash_packed_floats())

    def _pack(self, arithmetic=None):          # (one arithmetic in this family)
        _lib.check(_lib.lib().dh_hash_pack_weights(_lib.ptr(self.flat), _lib.ptr(self.packed), _lib.stream()))


def refresh_mask(binary, u):
    """nerfacc 0.3 OccupancyGrid._sample_uniform_and_occupied_cells(n = cells / 4) as inclusion probabilities: a cell is among the n
    uniform draws (with replacement) with probability 1 - (1 - 1/cells)^n = 1 - exp(-1/4); an occupied cell is among the occupied
    draws with probability 1 when there are at most n of them, else 1 - exp(-n / occupied) (n draws with replacement).  u: one
    uniform per cell; the two draws are made independent by splitting u's range."""
    cells = binary.numel()
    n = cells // 4
    occ = binary.to(torch.bool)
    k = occ.sum().to(torch.float32)
    p_uni = 1.0 - math.exp(-n / cells)
    p_occ = torch.where(k <= n, torch.ones_like(k), 1.0 - torch.exp(-n / k.clamp(min=1.0)))
    # u < p_uni: uniform draw; the occupied draw uses the remaining range rescaled to [0, 1)
    hit_uni = u < p_uni
    u2 = torch.where(hit_uni, u / p_uni, (u - p_uni) / (1.0 - p_uni))
    return hit_uni | (occ & (u2 < p_occ))


class OccupancyGrid:
    """res^3 occupancy image over [-radius, radius]^3 driving the ray marcher (nerfacc OccupancyGrid semantics as restated in
    oracle/occgrid_oracle.py): occ <- max(occ * decay, alpha(cell)), binary = occ > min(mean(occ), thre); alpha(cell) is the
    NeuS opacity of one marching step at a jittered point of the cell (instant-nsr-pl's occ_eval_fn).  The SDF queries run
    through the HIP no-grad kernel; the rest is a handful of elementwise ops on res^3 floats every `update_every` iterations."""

    def __init__(self, res=128, radius=1.0, decay=0.95, thre=0.001, device="cuda"):
        self.res, self.radius, self.decay, self.thre = int(res), float(radius), float(decay), float(thre)
        self.device = torch.device(device)
        self.occ = torch.zeros(self.res ** 3, device=self.device)
        self.binary = torch.ones(self.res ** 3, dtype=torch.uint8, device=self.device)       # all occupied until the first update
        self.updates = 0
        ax = torch.arange(self.res, device=self.device, dtype=torch.float32)
        ix, iy, iz = torch.meshgrid(ax, ax, ax, indexing="ij")
        self._idx = torch.stack([ix, iy, iz], -1).reshape(-1, 3)

    def cell_points(self, jitter):
        return ((self._idx + jitter) / self.res * 2.0 - 1.0) * self.radius

    @torch.no_grad()
    def update(self, sdf_fn, inv_s, step, generator=None, jitter=None, refresh="all", select=None):
        """refresh = "all" (this repo's default): every cell is re-evaluated and decayed at every update.  refresh = "quarter": nerfacc
        0.3's rule after its warm-up (OccupancyGrid._update: n = cells / 4 uniformly drawn cells + n cells drawn from the occupied
        ones; only those are decayed and re-evaluated) as per-cell Bernoulli draws with the same inclusion probabilities -- the count of
        occupied cells stays on the device, so the training path still never waits for a device -> host read.  select [res^3]
        uniforms (tests)."""
        if jitter is None:
            jitter = torch.rand(self.res ** 3, 3, device=self.device, generator=generator)
        sdf = sdf_fn(self.cell_points(jitter).contiguous()).reshape(-1)
        prev = torch.sigmoid((sdf + 0.5 * step) * inv_s)
        nxt = torch.sigmoid((sdf - 0.5 * step) * inv_s)
        alpha = ((prev - nxt + 1e-5) / (prev + 1e-5)).clip(0.0, 1.0)
        new = torch.maximum(self.occ * self.decay, alpha)
        if refresh == "quarter":
            self.occ = torch.where(refresh_mask(self.binary, select if select is not None else
                                                torch.rand(self.res ** 3, device=self.device, generator=generator)), new, self.occ)
        elif refresh == "all":
            self.occ = new
        else:
            raise ValueError("refresh must be 'all' or 'quarter'")
        self.binary = (self.occ > torch.clamp(self.occ.mean(), max=self.thre)).to(torch.uint8).contiguous()
        self.updates += 1

    def occupied_fraction(self) -> float:
        """Share of occupied cells (a device -> host read: diagnostics only, never on the training path)."""
        return float(self.binary.float().mean())

    def state_dict(self):
        return {"occ": self.occ.clone(), "updates": self.updates}

    def load_state_dict(self, sd):
        self.occ = sd["occ"].to(self.device, torch.float32).reshape(-1).clone()
        self.updates = int(sd["updates"])
        self.binary = (self.occ > torch.clamp(self.occ.mean(), max=self.thre)).to(torch.uint8).contiguous()


class HashNeuSRenderer(NeuSRenderer):
    """NeuSRenderer over the hash-grid networks.  Same constructor, render() dict and train_step_core().
    sampler = "hierarchical": the NeuS 64 + 64 sampler of the fp32 family (four no-grad SDF passes per iteration).
    sampler = "occgrid": instant-nsr-pl's occupancy-grid marching with packed variable-length rays (csrc/march.hip): the
    iteration evaluates the networks only on the samples that fall into occupied cells; the fused training step and the
    forward-only frame renderer take this path, render() (the autograd dict API) stays on the hierarchical sampler."""

    def __init__(self, nerf, sdf_network: HashSDFNetwork, deviation_network: SingleVarianceNetwork,
                 color_network: SHRenderingNetwork, *args, sampler="hierarchical", march_samples_per_ray=512, grid_res=128,
                 grid_update_every=16, max_samples=128, max_samples_per_ray=1024, reproducible_table_grad=True,
                 grid_refresh="all", grid_warmup_steps=256, **kwargs):
        if not isinstance(sdf_network, HashSDFNetwork) or not isinstance(color_network, SHRenderingNetwork):
            raise TypeError("HashNeuSRenderer needs HashSDFNetwork + SHRenderingNetwork")
        if sampler not in ("hierarchical", "occgrid"):
            raise ValueError("sampler must be 'hierarchical' or 'occgrid'")
        if not 0 < max_samples <= 1024 or not max_samples <= max_samples_per_ray <= 1024:
            raise ValueError("the packed render scan handles at most 1024 samples per ray (max_samples <= max_samples_per_ray <= 1024)")
        super().__init__(nerf, sdf_network, deviation_network, color_network, *args, **kwargs)
        self.radius = sdf_network.radius
        self.fd_eps = sdf_network.fd_eps
        self.sampler = sampler
        # the table scatter in fixed point (include/dynhor_hip.h dh_hash_weight_grads_parts, parts bit 4; the default: it costs nothing,
        # profiles/r05_hash_reproducible_scatter.json): the family's training step is bitwise reproducible like the NeuS family's.
        # False: float atomics, the library's only order-dependent sums (kept for comparison)
        self.reproducible_table_grad = bool(reproducible_table_grad)
        # packed-ray budget: the sample buffers of a batch of B rays hold B * max_samples samples (a fixed capacity, so nothing on
        # the training path waits for a device -> host read of the count); a single ray may keep up to max_samples_per_ray of
        # them.  The per-ray cap of an iteration is the largest of (1024, 512, 256, 128, ..., max_samples) <= max_samples_per_ray
        # whose total fits the capacity -- chosen on the device; max_samples itself always fits.
        self.max_samples = int(max_samples)
        self.max_samples_per_ray = int(max_samples_per_ray)
        self._cap_ladder = sorted({c for c in (1024, 512, 256, 128, 64, 32) if self.max_samples < c <= self.max_samples_per_ray}
                                  | {self.max_samples}, reverse=True)
        self._packed_buf = {}
        # instant-nsr-pl: render_step_size = 1.732 * 2 * radius / num_samples_per_ray
        self.march_step = 1.732 * 2.0 * self.radius / float(march_samples_per_ray)
        self.grid_update_every = int(grid_update_every)
        # "all": every update re-evaluates every cell (this repo's default).  "nerfacc": nerfacc 0.3's schedule -- all cells while
        # the training step is below grid_warmup_steps, then a uniform quarter + the occupied cells (OccupancyGrid.update)
        if grid_refresh not in ("all", "nerfacc"):
            raise ValueError("grid_refresh must be 'all' or 'nerfacc'")
        self.grid_refresh, self.grid_warmup_steps = grid_refresh, int(grid_warmup_steps)
        self.grid = OccupancyGrid(grid_res, self.radius, device=self.store.device) if sampler == "occgrid" else None
        self._march_iter = 0
        self._last_march = None

    def _make_store(self, sdf_network, deviation_network, color_network, device):
        return HashParamStore(sdf_network, deviation_network, color_network, device)

    def _workspace_need(self, npts: int, infer_only: bool) -> int:
        import ctypes
        infer, total = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(_lib.lib().dh_hash_workspace_floats(npts, ctypes.byref(infer), ctypes.byref(total)))
        return infer.value if infer_only else total.value

    def _net_sdf_nograd(self, tag, pts, n, out):
        st = self.store
        self.timer(tag, _lib.lib().dh_hash_sdf_nograd, _p(st.flat), _p(st.packed), _p(pts), n, self.radius, _p(out),
                   _lib.stream())

    def _net_forward(self, s, packed):
        L, T, st = _lib.lib(), self.timer, self.store
        P = s.B * s.n
        s.feat = torch.empty(P, 13, device=s.pts.device)
        T("hash_geo_forward", L.dh_hash_geo_forward, _p(st.flat), _p(packed), _p(s.pts), P, self.radius, self.fd_eps,
          _p(s.ws), 0 if s.infer_only else 1, _p(s.sdf), _p(s.feat), _p(s.normals), _NULLP, _lib.stream())
        T("hash_color_forward", L.dh_hash_color_forward, _p(packed), _p(s.feat), _p(s.normals), _p(s.rays_d), s.n, P,
          _p(s.colors), _NULLP, _lib.stream())

    def _weight_grads(self, P, ws, grad, n_dev):
        """dh_hash_weight_grads_parts in its two parts, CONCURRENTLY: the table scatter (bound by the memory side's atomic request rate:
        3.8 % of its wave-cycles issue an instruction, a quarter of the HBM rate) on the compute stream, the five small linears'
        weight-gradient GEMMs (an HBM stream) on a side stream that joins before this returns -- the two read and write disjoint parts
        of the workspace and of grad (DESIGN_NEXT_ROWS.md section 7).  With a table_grad_hook (the data-parallel Runner's) the hook --
        an asynchronous all-reduce of the 49 MB table slice, on the collective's own stream -- is issued right behind the scatter and
        overlaps the rest as well; its handle is left in self.pending_table_reduce for the caller to wait on
        before the optimiser step.  concurrent_weight_grads = False: one stream, table first (timing comparisons)."""
        L, T, st = _lib.lib(), self.timer, self.store
        hook = getattr(self, "table_grad_hook", None)
        fix = 4 if self.reproducible_table_grad else 0
        self.pending_table_reduce = None
        side = None
        if getattr(self, "concurrent_weight_grads", True) and grad.is_cuda:
            side = getattr(self, "_dw_stream", None)
            if side is None:
                side = self._dw_stream = torch.cuda.Stream(device=grad.device)
            side.wait_stream(torch.cuda.current_stream(grad.device))
            with torch.cuda.stream(side):
                T("hash_weight_grads_mlp", L.dh_hash_weight_grads_parts, _p(st.flat), _p(st.packed), P, _p(ws), _p(grad), n_dev, 2, _lib.stream())
        T("hash_weight_grads", L.dh_hash_weight_grads_parts, _p(st.flat), _p(st.packed), P, _p(ws), _p(grad), n_dev, 1 | fix, _lib.stream())
        if hook is not None:
            self.pending_table_reduce = hook(grad[:st.table_floats])
        if side is None:
            T("hash_weight_grads_mlp", L.dh_hash_weight_grads_parts, _p(st.flat), _p(st.packed), P, _p(ws), _p(grad), n_dev, 2, _lib.stream())
        else:
            torch.cuda.current_stream(grad.device).wait_stream(side)

    def _net_backward(self, s, d_sdf, d_normals, d_colors, grad):
        L, T, st = _lib.lib(), self.timer, self.store
        P = s.B * s.n
        d_feat = torch.empty(P, 13, device=s.pts.device)
        T("hash_color_backward", L.dh_hash_color_backward, _p(st.packed), _p(s.feat), _p(s.normals), _p(s.rays_d),
          _p(d_colors), s.n, P, _p(s.ws), _p(d_feat), _p(d_normals), _NULLP, _lib.stream())
        T("hash_geo_backward", L.dh_hash_geo_backward, _p(st.flat), _p(st.packed), _p(s.pts), _p(d_sdf), _p(d_feat),
          _p(d_normals), P, self.radius, self.fd_eps, _p(s.ws), _NULLP, _lib.stream())
        self._weight_grads(P, s.ws, grad, _NULLP)


    # ------------------------------------------------------------------ occupancy-grid marching path (packed rays)
    @torch.no_grad()
    def update_grid(self, generator=None, jitter=None):
        self.store.ensure_packed()
        quarter = self.grid_refresh == "nerfacc" and self._march_iter >= self.grid_warmup_steps and self.grid.updates > 0
        self.grid.update(lambda p: self.sdf(p), self.store.inv_s(), self.march_step, generator=generator, jitter=jitter,
                         refresh="quarter" if quarter else "all")

    def _buffers(self, B: int):
        """Sample buffers of a batch of B rays at the fixed capacity B * max_samples (rounded to 8 rows).  The TRAINING batch size
        (the first one seen by a training step; last_state and Runner.report() keep pointing into its buffers) owns an entry that
        is never evicted; every other size (validation / vote chunks, ragged last chunks) shares ONE inference buffer set that is
        allocated at the largest size seen and sliced (per-B entries with eviction re-allocated the training buffers)."""
        if getattr(self, "_train_B", None) is None and getattr(self, "_in_train_step", False):
            self._train_B = B
        role = "train" if B == getattr(self, "_train_B", None) else "infer"
        buf = self._packed_buf.get(role)
        if buf is not None and (buf.B < B if role == "infer" else buf.B != B):
            buf = None
        if buf is not None and buf.B != B:
            buf = self._slice_buffers(buf, B)
        if buf is None:
            from types import SimpleNamespace
            dev = self.store.device
            cap = (B * self.max_samples + 7) // 8 * 8
            z = lambda *shape, dt=torch.float32: torch.zeros(*shape, dtype=dt, device=dev)
            buf = SimpleNamespace(B=B, cap=cap, cnt_raw=z(B, dt=torch.int32), t_start=z(cap), pts=z(cap, 3), dirs=z(cap, 3),
                                  ray_idx=z(cap, dt=torch.int32), sdf=z(cap), normals=z(cap, 3), colors=z(cap, 3), feat=z(cap, 13),
                                  weights=z(cap), cdf=z(cap), inside=z(cap), d_sdf=z(cap), d_normals=z(cap, 3), d_colors=z(cap, 3),
                                  d_feat=z(cap, 13), caps=torch.tensor(self._cap_ladder, dtype=torch.int32, device=dev))
            self._packed_buf[role] = buf
        return buf

    def _slice_buffers(self, big, B: int):
        """A view of the inference buffer set for a smaller chunk of B rays (same storage, capacity B * max_samples)."""
        from types import SimpleNamespace
        cap = (B * self.max_samples + 7) // 8 * 8
        out = {}
        for k, v in vars(big).items():
            if k in ("B", "cap", "caps"):
                continue
            out[k] = v[:B] if k == "cnt_raw" else v[:cap]
        return SimpleNamespace(B=B, cap=cap, caps=big.caps, **out)

    @torch.no_grad()
    def march(self, rays_o, rays_d, near, far, u):
        """Packed samples of the rays: SimpleNamespace(cap, n_dev [1] i64 (device), off [B] i64, cnt [B] i32, cap_dev (the per-ray
        cap chosen, device), t_start / pts / dirs / ray_idx at the capacity).  NO device -> host read: the count stays on the
        device and every later stage takes it as `n_active`."""
        from types import SimpleNamespace
        L = _lib.lib()
        B = rays_o.shape[0]
        buf = self._buffers(B)
        near = near.contiguous().view(-1); far = far.contiguous().view(-1)
        u = None if u is None else u.contiguous().view(-1)
        step = float(torch.tensor(self.march_step, dtype=torch.float32))
        half = float(torch.tensor(0.5 * self.march_step, dtype=torch.float32))
        g = self.grid
        top = self._cap_ladder[0]
        _lib.check(L.dh_march_count(_p(rays_o), _p(rays_d), _p(near), _p(far), _p(u), _p(g.binary), g.res, g.radius, step, half,
                                    top, B, _p(buf.cnt_raw), _lib.stream()))
        # the largest per-ray cap of the ladder whose total fits the capacity (its last rung, max_samples, always does)
        totals = torch.minimum(buf.cnt_raw[None, :], buf.caps[:, None]).sum(dim=1, dtype=torch.int64)       # [len(ladder)]
        fits = totals <= buf.cap
        first = torch.argmax(fits.to(torch.int8)).view(1)                                                    # first rung that fits
        cap_dev = buf.caps.gather(0, first).view(())     # (indexing with a 0-dim tensor would read it back to the host)
        cnt = torch.minimum(buf.cnt_raw, cap_dev).contiguous()
        csum = torch.cumsum(cnt, 0, dtype=torch.int64)
        off = (csum - cnt).contiguous()
        n_dev = csum[-1:].contiguous()
        _lib.check(L.dh_march_emit(_p(rays_o), _p(rays_d), _p(near), _p(far), _p(u), _p(g.binary), g.res, g.radius, step, half,
                                   top, B, _p(off), _p(cnt), _p(buf.t_start), _p(buf.pts), _p(buf.dirs), _p(buf.ray_idx),
                                   _lib.stream()))
        return SimpleNamespace(cap=buf.cap, n_dev=n_dev, off=off, cnt=cnt, cnt_raw=buf.cnt_raw, cap_dev=cap_dev, step=step,
                               t_start=buf.t_start, pts=buf.pts, dirs=buf.dirs, ray_idx=buf.ray_idx, buf=buf)

    @torch.no_grad()
    def _forward_packed(self, rays_o, rays_d, m, cos_anneal_ratio, background_rgb, want_nmap, infer_only=False):
        from types import SimpleNamespace
        L, T, st = _lib.lib(), self.timer, self.store
        packed = st.ensure_packed()
        dev = rays_o.device
        B, P, buf = rays_o.shape[0], m.cap, m.buf
        s = SimpleNamespace(B=B, m=m, car=float(cos_anneal_ratio), bg=background_rgb, rays_o=rays_o, rays_d=rays_d,
                            infer_only=infer_only)
        s.ws = self._workspace(P, infer_only)
        self._ws_token += 1
        s.ws_token = self._ws_token
        s.sdf, s.normals, s.colors, s.feat = buf.sdf, buf.normals, buf.colors, buf.feat
        T("hash_geo_forward", L.dh_hash_geo_forward, _p(st.flat), _p(packed), _p(m.pts), P, self.radius, self.fd_eps,
          _p(s.ws), 0 if infer_only else 1, _p(s.sdf), _p(s.feat), _p(s.normals), _p(m.n_dev), _lib.stream())
        T("hash_color_forward", L.dh_hash_color_forward, _p(packed), _p(s.feat), _p(s.normals), _p(m.dirs), 1, P,
          _p(s.colors), _p(m.n_dev), _lib.stream())
        s.inv_s = st.inv_s()
        s.weights, s.cdf, s.inside = buf.weights, buf.cdf, buf.inside
        s.color = torch.empty(B, 3, device=dev); s.wsum = torch.empty(B, 1, device=dev); s.wmax = torch.empty(B, 1, device=dev)
        s.eik = torch.empty(B, 2, device=dev)
        s.nmap = torch.empty(B, 3, device=dev) if want_nmap else None
        _lib.check(L.dh_render_scan_fwd_packed(_p(rays_o), _p(rays_d), _p(m.t_start), _p(s.sdf), _p(s.normals), _p(s.colors),
                                               _p(s.inv_s), s.car, m.step, _p(background_rgb), B, _p(m.off), _p(m.cnt),
                                               _p(s.weights), _p(s.color), _p(s.wsum), _p(s.wmax), _p(s.cdf), _p(s.inside),
                                               _p(s.eik), _p(s.nmap), _lib.stream()))
        return s

    @torch.no_grad()
    def _backward_packed(self, s, d_color, d_wsum, d_nmap, eik_coef):
        L, T, st = _lib.lib(), self.timer, self.store
        if s.ws_token != self._ws_token:
            raise RuntimeError("workspace was overwritten by a later render before backward()")
        m, B, P, buf = s.m, s.B, s.m.cap, s.m.buf
        dev = s.color.device
        d_sdf, d_normals, d_colors, d_feat = buf.d_sdf.zero_(), buf.d_normals.zero_(), buf.d_colors.zero_(), buf.d_feat
        d_inv_s = torch.empty(B, device=dev)
        _lib.check(L.dh_render_scan_bwd_packed(_p(s.rays_o), _p(s.rays_d), _p(m.t_start), _p(s.sdf), _p(s.normals), _p(s.colors),
                                               _p(s.inv_s), s.car, m.step, _p(s.bg), B, _p(m.off), _p(m.cnt), _p(d_color),
                                               _p(d_wsum), _NULLP, _NULLP, _p(d_nmap), _p(eik_coef), _p(d_sdf), _p(d_normals),
                                               _p(d_colors), _p(d_inv_s), _lib.stream()))
        grad = st.grad_bucket()
        T("hash_color_backward", L.dh_hash_color_backward, _p(st.packed), _p(s.feat), _p(s.normals), _p(m.dirs), _p(d_colors),
          1, P, _p(s.ws), _p(d_feat), _p(d_normals), _p(m.n_dev), _lib.stream())
        T("hash_geo_backward", L.dh_hash_geo_backward, _p(st.flat), _p(st.packed), _p(m.pts), _p(d_sdf), _p(d_feat),
          _p(d_normals), P, self.radius, self.fd_eps, _p(s.ws), _p(m.n_dev), _lib.stream())
        self._weight_grads(P, s.ws, grad, _p(m.n_dev))
        raw = torch.exp(st.flat[st.var_off] * 10.0)
        passthrough = ((raw >= 1e-6) & (raw <= 1e6)).float()
        grad[st.var_off] = d_inv_s.sum() * 10.0 * raw * passthrough
        st.grad_flat = grad
        return grad

    @torch.no_grad()
    def train_step_core(self, rays, near, far, R, cos_anneal_ratio, igr_weight=0.1, mask_weight=0.1, normal_weight=0.0,
                        background_rgb=None, t_rand=None, **kw):
        if self.sampler != "occgrid":
            return super().train_step_core(rays, near, far, R, cos_anneal_ratio, igr_weight, mask_weight, normal_weight,
                                           background_rgb=background_rgb, t_rand=t_rand, **kw)
        if kw.get("corr") is not None or kw.get("ray_grads"):
            raise ValueError("the correspondence term / pose refinement run on the hierarchical sampler")
        L = _lib.lib()
        dev = rays.device
        B = rays.shape[0]
        if self._march_iter % self.grid_update_every == 0:
            self.update_grid()
        self._march_iter += 1
        rays_o = rays[:, 0:3].contiguous(); rays_d = rays[:, 3:6].contiguous()
        if t_rand is None and self.perturb > 0:
            t_rand = torch.rand(B, 1, device=dev)
        self._in_train_step = True           # _buffers: this batch size owns the never-evicted training entry
        try:
            m = self.march(rays_o, rays_d, near, far, t_rand if self.perturb > 0 else None)
        finally:
            self._in_train_step = False
        bg = None if background_rgb is None else background_rgb.reshape(-1).contiguous().float()
        s = self._forward_packed(rays_o, rays_d, m, cos_anneal_ratio, bg, want_nmap=normal_weight > 0.0)
        stats = torch.empty(8, device=dev)
        d_color = torch.empty(B, 3, device=dev); d_wsum = torch.empty(B, device=dev)
        d_nmap = torch.empty(B, 3, device=dev) if normal_weight > 0.0 else None
        eik_coef = torch.empty(1, device=dev)
        Rc = R.contiguous().float() if R is not None else None
        _lib.check(L.dh_neus_loss(_p(s.color), _p(s.wsum), _p(s.nmap), _p(s.eik), _p(rays), _p(Rc), B, float(igr_weight),
                                  float(mask_weight), float(normal_weight), _p(stats), _p(d_color), _p(d_wsum), _p(d_nmap),
                                  _p(eik_coef), _lib.stream()))
        self._backward_packed(s, d_color, d_wsum, d_nmap, eik_coef)
        self.last_state = s
        self._last_march = m
        return stats

    @property
    def last_march(self):
        """Statistics of the last marched batch (device -> host reads: call it when reporting, not per iteration)."""
        m = getattr(self, "_last_march", None)
        if m is None:
            return None
        n, B, cap = int(m.n_dev), int(m.cnt.shape[0]), int(m.cap_dev)
        return {"samples": n, "samples_per_ray": n / max(B, 1), "capacity": int(m.cap), "per_ray_cap": cap,
                "rays_at_cap": int((m.cnt_raw >= cap).sum()), "rays_truncated": int((m.cnt_raw > m.cnt).sum())}

    @torch.no_grad()
    def render_rays(self, rays_o, rays_d, near, far, cos_anneal_ratio, background_rgb=None, want_nmap=True):
        """Forward-only colour (+ normal map) of a chunk of rays with this renderer's sampler (validation frames)."""
        if self.sampler != "occgrid":
            z = self.sample_z(rays_o, rays_d, near, far, perturb_overwrite=0)
            st = self._forward_core(rays_o, rays_d, z, cos_anneal_ratio, background_rgb, want_nmap=want_nmap, infer_only=True)
            return st.color, st.nmap
        if self.grid.updates == 0:          # a validate-only / mesh run after load_checkpoint: never march through the all-ones grid
            self.update_grid()
        m = self.march(rays_o, rays_d, near, far, None)
        st = self._forward_packed(rays_o, rays_d, m, cos_anneal_ratio, background_rgb, want_nmap, infer_only=True)
        return st.color, st.nmap

    # occupancy-grid state rides in the checkpoint (Runner adds it under `dynhor_occgrid`): a resumed run keeps the decayed-max
    # history instead of restarting it, a validate-only run marches through the trained grid
    def sampler_state_dict(self):
        return None if self.grid is None else {"grid": self.grid.state_dict(), "march_iter": self._march_iter}

    def load_sampler_state_dict(self, sd):
        if sd is not None and self.grid is not None:
            self.grid.load_state_dict(sd["grid"])
            self._march_iter = int(sd["march_iter"])


def build_hash_models(seed=1234, device="cuda"):
    """Same seeded construction order as oracle/hashgrid_oracle.py:build_models (tests copy weights across anyway)."""
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(seed)
        sdf = HashSDFNetwork()
        col = SHRenderingNetwork()
    return sdf, SingleVarianceNetwork(0.3), col
