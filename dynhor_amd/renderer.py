"""NeuSRenderer -- host-side mirror of upstream ``models/renderer.py:NeuSRenderer``
whose every stage runs as hand-written HIP on gfx950 through the C ABI (include/dynhor_hip.h).

    renderer = NeuSRenderer(None, sdf_network, deviation_network, color_network,
                            n_samples=64, n_importance=64, n_outside=0, up_sample_steps=4, perturb=1.0)
    out = renderer.render(rays_o, rays_d, near, far, cos_anneal_ratio=r)     # dict, same keys as upstream

Two ways to train through it:
  * autograd (drop-in): the tensors returned by render() take part in autograd; ``loss.backward()`` runs the HIP
    backward (render-scan adjoint -> colour MLP -> SDF MLP incl. the second-order path through d sdf/d x ->
    weight-norm fold) and deposits gradients on the modules' parameters as views of ParamStore.grad_flat.
  * fused (Runner hot loop): ``train_step_core`` = sample -> render -> HIP loss kernel -> HIP backward with no
    autograd graph and no host synchronisation; the flat gradient is then all-reduced (RCCL) and fed to the fused Adam.
"""
from __future__ import annotations

import ctypes
from types import SimpleNamespace

import torch

from . import _lib
from .fields import ParamStore, RenderingNetwork, SDFNetwork, SingleVarianceNetwork

_NULL = ctypes.c_void_p(0)


def _p(t):
    return _NULL if t is None else _lib.ptr(t)


class StageTimer:
    """Calls a C-ABI stage; when enabled, brackets it with HIP events on the launch stream (torch's current stream) so
    bench.py can report per-kernel durations measured inside the timed region.  markers (Runner conf train.roctx, a flag, not an
    environment variable): every stage call sits in a roctx range named after the stage, so that a `rocprofv3 --marker-trace
    --kernel-trace` timeline reads as the stages of DESIGN.md section 1 (tracing)."""

    def __init__(self):
        self.enabled = False
        self.records = {}
        self.markers = False
        self._roctx = None

    def set_markers(self, on: bool):
        self._roctx = _lib.roctx() if on else None
        self.markers = bool(on) and self._roctx is not None
        return self.markers

    def __call__(self, name, fn, *args):
        if self.markers:
            self._roctx.roctxRangePushA(name.encode())
            try:
                self._call(name, fn, *args)
            finally:
                self._roctx.roctxRangePop()
        else:
            self._call(name, fn, *args)

    def _call(self, name, fn, *args):
        if self.enabled:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _lib.check(fn(*args))
            b.record()
            self.records.setdefault(name, []).append((a, b))
        else:
            _lib.check(fn(*args))

    def summary(self):
        """name -> (mean ms, count); call after a device synchronise."""
        return {k: (sum(a.elapsed_time(b) for a, b in v) / len(v), len(v)) for k, v in self.records.items()}

    def reset(self):
        self.records = {}


class _RenderCoreFn(torch.autograd.Function):
    """render_core (App. A.7) on fixed z_vals.  Inputs: every parameter (flat order) so autograd routes grads."""

    @staticmethod
    def forward(ctx, renderer, rays_o, rays_d, z_vals, cos_anneal_ratio, background_rgb, *params):
        s = renderer._forward_core(rays_o, rays_d, z_vals, cos_anneal_ratio, background_rgb, want_nmap=False)
        s.eik_sum = s.eik.sum(dim=0)              # the dict API's gradient_error; the fused step's loss kernel reads s.eik itself
        s.gradient_error = s.eik_sum[0] / (s.eik_sum[1] + 1e-5)
        ctx.set_materialize_grads(False)
        ctx.state = s
        ctx.renderer = renderer
        ctx.mark_non_differentiable(s.wmax, s.cdf, s.inside)
        B, n = z_vals.shape
        return (s.color, s.wsum, s.gradient_error, s.weights, s.normals.view(B, n, 3), s.wmax, s.cdf, s.inside,
                s.sdf.view(B * n, 1))

    @staticmethod
    def backward(ctx, d_color, d_wsum, d_ge, d_weights, d_gradients, _wmax, _cdf, _inside, d_sdf_out):
        r = ctx.renderer
        s = ctx.state
        dev = s.color.device
        eik_coef = ((d_ge if d_ge is not None else torch.zeros((), device=dev)) / (s.eik_sum[1] + 1e-5)).reshape(1)
        grad = r._backward_core(s, d_color, d_wsum, d_weights, d_gradients, None, eik_coef.contiguous(), d_sdf_out)
        return (None, None, None, None, None, None) + tuple(grad[off:off + cnt].view(p.shape)
                                                           for p, off, cnt in r.store.slices)


class NeuSRenderer:
    def __init__(self, nerf, sdf_network: SDFNetwork, deviation_network: SingleVarianceNetwork,
                 color_network: RenderingNetwork, n_samples, n_importance, n_outside, up_sample_steps, perturb,
                 store: ParamStore | None = None, device="cuda", arithmetic: int | None = None):
        if n_outside != 0:
            raise ValueError("n_outside > 0 (NeRF++ background) is out of scope with masks (SURVEY.md App. A.5)")
        if n_importance % max(up_sample_steps, 1) != 0:
            raise ValueError("n_importance must be divisible by up_sample_steps")
        if n_samples + n_importance > 128 or n_samples % 2 or (n_importance // max(up_sample_steps, 1)) % 2:
            raise ValueError("gfx950 per-ray kernels need an even sample count and n_samples + n_importance <= 128")
        _lib.lib()   # fail loudly if the HIP extension is missing
        self.nerf = nerf
        self.sdf_network = sdf_network
        self.deviation_network = deviation_network
        self.color_network = color_network
        self.n_samples = n_samples
        self.n_importance = n_importance
        self.n_outside = n_outside
        self.up_sample_steps = up_sample_steps
        self.perturb = perturb
        self.store = store if store is not None else self._make_store(sdf_network, deviation_network, color_network, device)
        self._ws = None
        self._ws_token = 0
        self.timer = StageTimer()
        # the arithmetic of this renderer's MLP stages (_lib.ARITH_SPLIT_F16 / ARITH_SPLIT_BF16 / ARITH_FP32_MFMA), passed with
        # every launch through the `_ex` entry points: no process-global word is read, so renderers with different arithmetics
        # can run side by side, also from different host threads.  None = the library's default at the time of each call.
        self.arithmetic = arithmetic

    def _make_store(self, sdf_network, deviation_network, color_network, device):
        return ParamStore(sdf_network, deviation_network, color_network, device)

    # ------------------------------------------------------------------ workspace (caller-owned, reused)
    def _workspace_need(self, npts: int, infer_only: bool) -> int:
        infer, _, total = _lib.workspace_floats(npts)
        return infer if infer_only else total

    def _workspace(self, npts: int, infer_only: bool = False) -> torch.Tensor:
        need = self._workspace_need(npts, infer_only)
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, device=self.store.device, dtype=torch.float32)
        return self._ws

    # ------------------------------------------------------------------ network stages (the model-family hooks;
    # hash_fields.HashNeuSRenderer overrides these three + _workspace_need)
    def _arith(self) -> int:
        return _lib.get_arithmetic() if self.arithmetic is None else int(self.arithmetic)

    def _net_sdf_nograd(self, tag, pts, n, out):
        self.timer(tag, _lib.lib().dh_sdf_nograd_ex, self._arith(), _p(self.store.packed), _p(pts), n, _p(out), _lib.stream())

    def _net_forward(self, s, packed):
        """pts [P,3] -> s.sdf, s.normals (d sdf / d x), s.colors; saves what _net_backward needs in s.ws."""
        L, T = _lib.lib(), self.timer
        P = s.B * s.n
        ar = s.arith = self._arith()          # the backward of this forward runs in the same arithmetic
        T("sdf_forward", L.dh_sdf_forward_ex, ar, _p(packed), _p(s.pts), P, _p(s.ws), _p(s.sdf), _lib.stream())
        save = 0 if s.infer_only else (2 if getattr(s, "ray_grads", False) else 1)
        T("sdf_gradient", L.dh_sdf_gradient_ex, ar, _p(packed), _p(s.pts), P, _p(s.ws), _p(s.normals), save, _lib.stream())
        T("color_forward", L.dh_color_forward_ex, ar, _p(packed), _p(s.pts), _p(s.rays_d), s.n, _p(s.normals), P, _p(s.ws),
          _p(s.colors), save, _lib.stream())

    def _net_backward(self, s, d_sdf, d_normals, d_colors, grad):
        """Adjoint of _net_forward into the flat gradient (every slot but `variance`).  With s.ray_grads (pose refinement) it
        also leaves s.d_pts [P,3] (d loss / d sample point) and s.d_dirs_pts [P,3] (d loss / d ray direction per point)."""
        L, T, st = _lib.lib(), self.timer, self.store
        P = s.B * s.n
        ar = s.arith
        if getattr(s, "ray_grads", False):
            s.d_pts = torch.empty(P, 3, device=s.pts.device)
            s.d_dirs_pts = torch.empty(P, 3, device=s.pts.device)
            T("color_backward", L.dh_color_backward_rays_ex, ar, _p(st.packed), _p(s.colors), _p(d_colors), _p(s.rays_d), s.n, P, _p(s.ws),
              _p(d_normals), _p(s.d_pts), _p(s.d_dirs_pts), _lib.stream())
            T("sdf_tangent", L.dh_sdf_tangent_ex, ar, _p(st.packed), _p(s.pts), _p(d_normals), P, _p(s.ws), _lib.stream())
            T("sdf_backward", L.dh_sdf_backward_rays_ex, ar, _p(st.packed), _p(d_sdf), _p(s.pts), _p(d_normals), P, _p(s.ws), _p(s.d_pts),
              _lib.stream())
        else:
            T("color_backward", L.dh_color_backward_ex, ar, _p(st.packed), _p(s.colors), _p(d_colors), P, _p(s.ws), _p(d_normals),
              _lib.stream())
            T("sdf_tangent", L.dh_sdf_tangent_ex, ar, _p(st.packed), _p(s.pts), _p(d_normals), P, _p(s.ws), _lib.stream())
            T("sdf_backward", L.dh_sdf_backward_ex, ar, _p(st.packed), _p(d_sdf), P, _p(s.ws), _lib.stream())
        T("weight_grads_gemm", L.dh_weight_grads_gemm_ex, ar, P, _p(s.ws), _lib.stream())
        T("weight_grads_fold", L.dh_weight_grads_fold, _p(st.packed), _p(st.flat), P, _p(s.ws), _p(grad), _lib.stream())

    # ------------------------------------------------------------------ range watch of the two-piece fp16 arithmetic
    def range_status(self, state=None):
        """(max softplus activation of the last SPLIT_F16 forward, limit) -- ONE device read, nothing raised; None for the other
        arithmetics (no limit) or before the first step."""
        s = state if state is not None else getattr(self, "last_state", None)
        if s is None or getattr(s, "arith", None) != _lib.ARITH_SPLIT_F16:
            return None
        a, _, lim = _lib.range_words()
        return float(s.ws[a]), lim

    def check_range(self, state=None, max_over_ranks=None):
        """The SPLIT_F16 forward chain carries its softplus activations at a constant fp16 scale (overflow beyond 4094:
        include/dynhor_hip.h dh_range_words); the input-gradient stage posts the largest activation of the step into the workspace.
        ONE device read: call at report iterations, not per step.  Returns (max activation, limit) of the last forward, raises
        DynhorHipError beyond the limit (the step's results are NaN there): switch to arithmetic 'split_bf16', which has no limit.
        max_over_ranks: a callable reducing one float with MAX over the data-parallel ranks (Runner.report passes dist.max_over_ranks)
        so that EVERY rank raises when any rank overflowed -- a rank raising alone leaves the others blocked in the next all-reduce."""
        st = self.range_status(state)
        if st is None:
            if max_over_ranks is not None:
                max_over_ranks(0.0)                        # (keep the collective matched across ranks)
            return None
        m, lim = st
        g = max_over_ranks(m) if max_over_ranks is not None else m
        if not g < lim:                                    # (NaN compares false)
            where = "" if g == m or max_over_ranks is None else f" (this rank: {m:.4g}; the maximum is another rank's)"
            raise _lib.DynhorHipError(
                f"split_f16 range exceeded: max softplus activation of the SDF network = {g:.4g}{where}, limit {lim:.0f} (fp16 overflow at "
                "the forward chain's constant scale 16): this step's results are not valid; use model.arithmetic = 'split_bf16'")
        return m, lim

    # ------------------------------------------------------------------ no-grad SDF queries
    def sdf(self, pts: torch.Tensor) -> torch.Tensor:
        """sdf_network.sdf(pts) under no_grad: [N,3] -> [N,1]."""
        pts = pts.contiguous().float()
        out = torch.empty(pts.shape[0], device=pts.device)
        self.store.ensure_packed(self._arith())
        self._net_sdf_nograd("sdf_nograd", pts, pts.shape[0], out)
        return out.view(-1, 1)

    # ------------------------------------------------------------------ geometry extraction (upstream extract_geometry)
    @torch.no_grad()
    def extract_geometry(self, bound_min, bound_max, resolution, threshold=0.0, method="cubes"):
        """Upstream NeuSRenderer.extract_geometry(bound_min, bound_max, resolution, threshold): -sdf on a regular grid
        (HIP no-grad SDF kernel, 64^3-point chunks), iso-surface at `threshold`.  Returns (vertices [V,3], triangles [F,3])
        as device tensors.  method "cubes" (default since round 5): table-driven marching cubes, upstream's algorithm
        (mcubes.marching_cubes; PyMCubes itself is not installable here, the case table is generated in dynhor_amd/mesh.py);
        "tetrahedra": marching tetrahedra (rounds 1-4), three times the triangles."""
        from .mesh import marching_cubes, marching_tetrahedra
        if method not in ("cubes", "tetrahedra"):
            raise ValueError("extract_geometry: method must be 'cubes' or 'tetrahedra'")
        dev = self.store.device
        N = int(resolution)
        ax = [torch.linspace(float(bound_min[i]), float(bound_max[i]), N, device=dev) for i in range(3)]
        u = torch.empty(N, N, N, device=dev)
        step = 64
        for xi in range(0, N, step):
            for yi in range(0, N, step):
                for zi in range(0, N, step):
                    gx, gy, gz = torch.meshgrid(ax[0][xi:xi + step], ax[1][yi:yi + step], ax[2][zi:zi + step], indexing="ij")
                    pts = torch.stack([gx, gy, gz], dim=-1).reshape(-1, 3).contiguous()
                    u[xi:xi + step, yi:yi + step, zi:zi + step] = -self.sdf(pts).reshape(gx.shape)
        if not bool(torch.isfinite(u).all()):
            # (the no-grad chain has no workspace to post a range status into: beyond the split_f16 range its outputs are NaN)
            raise _lib.DynhorHipError("extract_geometry: non-finite SDF values on the grid (split_f16 range exceeded, or the network "
                                      "has diverged); use arithmetic 'split_bf16' for queries this far out")
        return (marching_cubes if method == "cubes" else marching_tetrahedra)(u, threshold, bound_min, bound_max)

    # ------------------------------------------------------------------ hierarchical sampling (App. A.5/A.6)
    @torch.no_grad()
    def sample_z(self, rays_o, rays_d, near, far, perturb_overwrite=-1, t_rand=None):
        L = _lib.lib()
        packed = self.store.ensure_packed(self._arith())
        dev = rays_o.device
        B = rays_o.shape[0]
        ns = self.n_samples
        perturb = self.perturb if perturb_overwrite < 0 else perturb_overwrite
        if perturb > 0 and t_rand is None:
            t_rand = torch.rand([B, 1], device=dev)
        if perturb <= 0:
            t_rand = None
        near = near.contiguous().view(-1)
        far = far.contiguous().view(-1)
        z = torch.empty(B, ns, device=dev)
        pts = torch.empty(B * ns, 3, device=dev)
        _lib.check(L.dh_coarse_samples(_p(rays_o), _p(rays_d), _p(near), _p(far),
                                       _p(t_rand.contiguous().view(-1)) if t_rand is not None else _NULL,
                                       B, ns, _p(z), _p(pts), _lib.stream()))
        if self.n_importance > 0:
            sdf = torch.empty(B * ns, device=dev)
            self._net_sdf_nograd("sdf_nograd_coarse", pts, B * ns, sdf)
            n_new = self.n_importance // self.up_sample_steps
            n_cur = ns
            for i in range(self.up_sample_steps):
                last = (i + 1 == self.up_sample_steps)
                z_new = torch.empty(B, n_new, device=dev)
                pts_new = torch.empty(B * n_new, 3, device=dev)
                _lib.check(L.dh_upsample_step(_p(rays_o), _p(rays_d), _p(z), _p(sdf), B, n_cur, n_new,
                                              float(64 * 2 ** i), _p(z_new), _p(pts_new), _lib.stream()))
                z_out = torch.empty(B, n_cur + n_new, device=dev)
                if not last:
                    sdf_new = torch.empty(B * n_new, device=dev)
                    self._net_sdf_nograd("sdf_nograd_fine", pts_new, B * n_new, sdf_new)
                    sdf_out = torch.empty(B * (n_cur + n_new), device=dev)
                    _lib.check(L.dh_merge_samples(_p(z), _p(z_new), _p(sdf), _p(sdf_new), B, n_cur, n_new, _p(z_out),
                                                  _p(sdf_out), _lib.stream()))
                    sdf = sdf_out
                else:
                    _lib.check(L.dh_merge_samples(_p(z), _p(z_new), _NULL, _NULL, B, n_cur, n_new, _p(z_out), _NULL,
                                                  _lib.stream()))
                z = z_out
                n_cur += n_new
        return z

    # ------------------------------------------------------------------ render_core forward / backward (no autograd)
    @torch.no_grad()
    def _forward_core(self, rays_o, rays_d, z_vals, cos_anneal_ratio, background_rgb, want_nmap, infer_only=False,
                      ray_grads=False):
        L = _lib.lib()
        st = self.store
        packed = st.ensure_packed(self._arith())
        dev = rays_o.device
        B, n = z_vals.shape
        P = B * n
        s = SimpleNamespace()
        s.B, s.n, s.car, s.sample_dist = B, n, float(cos_anneal_ratio), 2.0 / self.n_samples
        s.rays_o, s.rays_d, s.z_vals, s.bg = rays_o, rays_d, z_vals, background_rgb
        s.ws = self._workspace(P, infer_only)
        s.infer_only = infer_only
        s.ray_grads = bool(ray_grads) and not infer_only
        self._ws_token += 1
        s.ws_token = self._ws_token
        s.pts = torch.empty(P, 3, device=dev)
        s.sdf = torch.empty(P, device=dev)
        s.normals = torch.empty(P, 3, device=dev)
        s.colors = torch.empty(P, 3, device=dev)
        _lib.check(L.dh_midpoints(_p(rays_o), _p(rays_d), _p(z_vals), B, n, s.sample_dist, _p(s.pts), _lib.stream()))
        self._net_forward(s, packed)
        s.inv_s = st.inv_s()
        s.weights = torch.empty(B, n, device=dev)
        s.color = torch.empty(B, 3, device=dev)
        s.wsum = torch.empty(B, 1, device=dev)
        s.wmax = torch.empty(B, 1, device=dev)
        s.cdf = torch.empty(B, n, device=dev)
        s.inside = torch.empty(B, n, device=dev)
        s.eik = torch.empty(B, 2, device=dev)
        s.nmap = torch.empty(B, 3, device=dev) if want_nmap else None
        _lib.check(L.dh_render_scan_fwd(_p(rays_o), _p(rays_d), _p(z_vals), _p(s.sdf), _p(s.normals), _p(s.colors),
                                        _p(s.inv_s), s.car, s.sample_dist, _p(background_rgb), B, n, _p(s.weights),
                                        _p(s.color), _p(s.wsum), _p(s.wmax), _p(s.cdf), _p(s.inside), _p(s.eik),
                                        _p(s.nmap), _lib.stream()))
        return s

    @torch.no_grad()
    def _backward_core(self, s, d_color, d_wsum, d_weights, d_gradients, d_nmap, eik_coef, d_sdf_out=None, persistent=False):
        """Adjoint of _forward_core: returns the flat parameter gradient (also kept as store.grad_flat).  persistent: write it
        into the store's one long-lived bucket (the fused training step: same address every iteration for the all-reduce)."""
        L = _lib.lib()
        st = self.store
        if s.infer_only:
            raise RuntimeError("this forward pass was run with infer_only=True: nothing was saved for backward")
        if s.ws_token != self._ws_token:
            raise RuntimeError("NeuSRenderer workspace was overwritten by a later render() call before backward(); "
                               "call backward() before rendering again (one live graph per renderer)")
        B, n = s.B, s.n
        P = B * n
        dev = s.color.device
        c = lambda t: None if t is None else t.contiguous()
        d_color = c(d_color) if d_color is not None else torch.zeros(B, 3, device=dev)
        d_sdf = torch.empty(P, device=dev)
        d_normals = torch.empty(P, 3, device=dev)
        d_colors = torch.empty(P, 3, device=dev)
        d_inv_s = torch.empty(B, device=dev)
        if s.ray_grads:
            d_rays_d = torch.empty(B, 3, device=dev)
            _lib.check(L.dh_render_scan_bwd_rays(_p(s.rays_o), _p(s.rays_d), _p(s.z_vals), _p(s.sdf), _p(s.normals), _p(s.colors),
                                                 _p(s.inv_s), s.car, s.sample_dist, _p(s.bg), B, n, _p(d_color), _p(c(d_wsum)),
                                                 _p(c(d_weights)), _p(c(d_gradients)), _p(c(d_nmap)), _p(eik_coef), _p(d_sdf),
                                                 _p(d_normals), _p(d_colors), _p(d_inv_s), _p(d_rays_d), _lib.stream()))
        else:
            _lib.check(L.dh_render_scan_bwd(_p(s.rays_o), _p(s.rays_d), _p(s.z_vals), _p(s.sdf), _p(s.normals), _p(s.colors),
                                            _p(s.inv_s), s.car, s.sample_dist, _p(s.bg), B, n, _p(d_color), _p(c(d_wsum)),
                                            _p(c(d_weights)), _p(c(d_gradients)), _p(c(d_nmap)), _p(eik_coef), _p(d_sdf),
                                            _p(d_normals), _p(d_colors), _p(d_inv_s), _lib.stream()))
        if d_sdf_out is not None:
            d_sdf = (d_sdf + d_sdf_out.reshape(-1)).contiguous()
        grad = st.grad_bucket() if persistent else torch.empty(st.n, device=dev)
        self._net_backward(s, d_sdf, d_normals, d_colors, grad)
        # variance: inv_s = clip(exp(10 v), 1e-6, 1e6)
        raw = torch.exp(st.flat[st.var_off] * 10.0)
        passthrough = ((raw >= 1e-6) & (raw <= 1e6)).float()
        grad[st.var_off] = d_inv_s.sum() * 10.0 * raw * passthrough
        st.grad_flat = grad
        if s.ray_grads:
            # per-ray reduction of the point adjoints (sample depths are constants): x = o + mid * d
            dz = torch.cat([s.z_vals[:, 1:] - s.z_vals[:, :-1], torch.full_like(s.z_vals[:, :1], s.sample_dist)], -1)
            mid = (s.z_vals + 0.5 * dz).unsqueeze(-1)
            dp = s.d_pts.view(B, n, 3)
            s.d_rays_o = dp.sum(dim=1)
            s.d_rays_d = (dp * mid).sum(dim=1) + s.d_dirs_pts.view(B, n, 3).sum(dim=1) + d_rays_d
        return grad

    @torch.no_grad()
    def render_rays(self, rays_o, rays_d, near, far, cos_anneal_ratio, background_rgb=None, want_nmap=True):
        """Forward-only colour (+ normal map) of a chunk of rays (validation frames): no perturbation, nothing saved."""
        z = self.sample_z(rays_o, rays_d, near, far, perturb_overwrite=0)
        st = self._forward_core(rays_o, rays_d, z, cos_anneal_ratio, background_rgb, want_nmap=want_nmap, infer_only=True)
        return st.color, st.nmap

    # ------------------------------------------------------------------ render (App. A.5)
    def render(self, rays_o, rays_d, near, far, perturb_overwrite=-1, background_rgb=None, cos_anneal_ratio=0.0,
               t_rand=None, z_vals=None):
        if rays_o.dim() != 2 or rays_o.shape[1] != 3 or rays_d.shape != rays_o.shape:
            raise ValueError("rays_o / rays_d must be [B,3]")
        if rays_o.dtype != torch.float32 or not rays_o.is_cuda:
            raise TypeError("rays must be fp32 tensors on the HIP device")
        rays_o = rays_o.contiguous()
        rays_d = rays_d.contiguous()
        B = rays_o.shape[0]
        if z_vals is None:
            z_vals = self.sample_z(rays_o, rays_d, near, far, perturb_overwrite, t_rand)
        z_vals = z_vals.contiguous()
        bg = None if background_rgb is None else background_rgb.reshape(-1).contiguous().float()
        outs = _RenderCoreFn.apply(self, rays_o, rays_d, z_vals, cos_anneal_ratio, bg, *self.store.params())
        color, wsum, gradient_error, weights, gradients, wmax, cdf, inside, sdf = outs
        inv_s = self.store.inv_s()
        s_val = (1.0 / inv_s).expand(B, 1)
        return {
            "color_fine": color, "s_val": s_val, "cdf_fine": cdf, "weight_sum": wsum, "weight_max": wmax,
            "gradients": gradients, "weights": weights, "gradient_error": gradient_error, "inside_sphere": inside,
            "z_vals": z_vals, "sdf": sdf,
        }

    # ------------------------------------------------------------------ fused training step (Runner hot loop)
    @torch.no_grad()
    def train_step_core(self, rays, near, far, R, cos_anneal_ratio, igr_weight=0.1, mask_weight=0.1, normal_weight=0.0,
                        background_rgb=None, t_rand=None, corr=None, corr_weight=0.0, corr_frames=None, corr_delta_px=4.0,
                        ray_grads=False):
        """rays [B,14] (dh_gen_rays layout), R [3,3] object->camera of the frame.  Returns stats [8] on device:
        loss, colour, eikonal, mask, normal, psnr, sum(obj*keep), sum(keep); leaves the flat gradient in
        store.grad_flat.  No host synchronisation.
        Full loss stack (BASELINE.json configs[4]): corr [B,4] = (u_j, v_j, certainty, frame_j) per ray with
        corr_frames = (R_all [F,3,3], T_all [F,3], K [3,3]) adds corr_weight * the dense-correspondence reprojection loss
        (dh_corr_loss); its statistics land in self.last_corr_stats [4] and self.last_corr_residual_px [B].
        ray_grads (pose refinement): additionally leaves d loss / d rays_o, d loss / d rays_d [B,3] (sample depths constant)
        and the normal loss's direct gradient w.r.t. R in self.last_ray_grads = (d_rays_o, d_rays_d, d_R or None); with the
        correspondence term active as well, self.last_partner_pose_grads = (d_R_all [F,3,3], d_T_all [F,3]) holds its gradient
        w.r.t. the partner frames' poses (None otherwise)."""
        L = _lib.lib()
        dev = rays.device
        B = rays.shape[0]
        rays_o = rays[:, 0:3].contiguous()
        rays_d = rays[:, 3:6].contiguous()
        z_vals = self.sample_z(rays_o, rays_d, near, far, t_rand=t_rand)
        bg = None if background_rgb is None else background_rgb.reshape(-1).contiguous().float()
        s = self._forward_core(rays_o, rays_d, z_vals, cos_anneal_ratio, bg, want_nmap=normal_weight > 0.0, ray_grads=ray_grads)
        stats = torch.empty(8, device=dev)
        d_color = torch.empty(B, 3, device=dev)
        d_wsum = torch.empty(B, device=dev)
        d_nmap = torch.empty(B, 3, device=dev) if normal_weight > 0.0 else None
        eik_coef = torch.empty(1, device=dev)
        Rc = R.contiguous().float() if R is not None else None
        _lib.check(L.dh_neus_loss(_p(s.color), _p(s.wsum), _p(s.nmap), _p(s.eik), _p(rays), _p(Rc), B, float(igr_weight),
                                  float(mask_weight), float(normal_weight), _p(stats), _p(d_color), _p(d_wsum),
                                  _p(d_nmap), _p(eik_coef), _lib.stream()))
        d_weights = None
        if corr is not None and corr_weight > 0.0:
            R_all, T_all, K = corr_frames
            cstats = torch.empty(4, device=dev)
            resid = torch.empty(B, device=dev)
            d_weights = torch.empty(B, s.n, device=dev)
            pose_adj = torch.empty(B, 7, device=dev) if ray_grads else None
            _lib.check(L.dh_corr_loss(_p(rays_o), _p(rays_d), _p(z_vals), _p(s.weights), _p(corr.contiguous()),
                                      _p(R_all.contiguous()), _p(T_all.contiguous()), int(R_all.shape[0]), _p(K.contiguous()),
                                      B, s.n, s.sample_dist, float(corr_delta_px), float(corr_weight), _p(cstats), _p(resid),
                                      _p(d_weights), _p(pose_adj), _lib.stream()))
            stats[0] += cstats[3]
            self.last_corr_stats, self.last_corr_residual_px = cstats, resid
        self._backward_core(s, d_color, d_wsum, d_weights, None, d_nmap, eik_coef, persistent=True)
        self.last_state = s
        if ray_grads:
            d_R = None
            if normal_weight > 0.0 and Rc is not None:
                # n_cam = R n_obj enters the normal loss directly: d loss / d R = sum_b (R d_nmap_b) nmap_b^T  (d_nmap = R^T d n_cam)
                d_R = (d_nmap @ Rc.T).T @ s.nmap
            self.last_partner_pose_grads = None
            if corr is not None and corr_weight > 0.0:
                # the correspondence term also depends on the rays DIRECTLY (x = o + t^ d at fixed t^) and on the partner
                # frames' poses (y = R_j x + T_j); dh_corr_loss returns those adjoints per ray (without them the pose
                # gradient of the combined configuration was incomplete)
                d_x, d_y, t_hat = pose_adj[:, 0:3], pose_adj[:, 3:6], pose_adj[:, 6:7]
                s.d_rays_o = s.d_rays_o + d_x
                s.d_rays_d = s.d_rays_d + t_hat * d_x
                x = rays_o + t_hat * rays_d
                j = corr[:, 3].long().clamp(0, R_all.shape[0] - 1)
                d_R_all = torch.zeros_like(R_all).index_add_(0, j, d_y[:, :, None] * x[:, None, :])
                d_T_all = torch.zeros_like(T_all).index_add_(0, j, d_y)
                self.last_partner_pose_grads = (d_R_all, d_T_all)
            self.last_ray_grads = (s.d_rays_o, s.d_rays_d, d_R)
        return stats
