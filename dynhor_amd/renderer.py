"""NeuSRenderer -- host-side mirror of upstream ``models/renderer.py:NeuSRenderer``
whose every stage runs as hand-written HIP on gfx950 through the C ABI (include/dynhor_hip.h).

    renderer = NeuSRenderer(None, sdf_network, deviation_network, color_network,
                            n_samples=64, n_importance=64, n_outside=0, up_sample_steps=4, perturb=1.0)
    out = renderer.render(rays_o, rays_d, near, far, cos_anneal_ratio=r)     # dict, same keys as upstream

The returned tensors take part in autograd: ``loss.backward()`` runs the HIP backward (render scan adjoint ->
colour MLP -> SDF MLP incl. the second-order path through d sdf/d x -> weight-norm fold) and deposits gradients on
the modules' parameters as views of one flat buffer (ParamStore.grad_flat).
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from .fields import ParamStore, RenderingNetwork, SDFNetwork, SingleVarianceNetwork

_NULL = ctypes.c_void_p(0)


def _p(t):
    return _NULL if t is None else _lib.ptr(t)


class _RenderCoreFn(torch.autograd.Function):
    """render_core (App. A.7) on fixed z_vals.  Inputs: every parameter (flat order) so autograd routes grads."""

    @staticmethod
    def forward(ctx, renderer, rays_o, rays_d, z_vals, cos_anneal_ratio, background_rgb, *params):
        L = _lib.lib()
        st = renderer.store
        packed = st.ensure_packed()
        dev = rays_o.device
        B, n = z_vals.shape
        P = B * n
        sample_dist = 2.0 / renderer.n_samples
        ws = renderer._workspace(P)
        pts = torch.empty(P, 3, device=dev)
        sdf = torch.empty(P, device=dev)
        normals = torch.empty(P, 3, device=dev)
        colors = torch.empty(P, 3, device=dev)
        _lib.check(L.dh_midpoints(_p(rays_o), _p(rays_d), _p(z_vals), B, n, sample_dist, _p(pts), _lib.stream()))
        _lib.check(L.dh_mlp_forward(_p(packed), _p(pts), _p(rays_d), n, P, _p(ws), _p(sdf), _p(normals), _p(colors),
                                    _lib.stream()))
        inv_s = st.inv_s()
        weights = torch.empty(B, n, device=dev)
        color = torch.empty(B, 3, device=dev)
        wsum = torch.empty(B, 1, device=dev)
        wmax = torch.empty(B, 1, device=dev)
        cdf = torch.empty(B, n, device=dev)
        inside = torch.empty(B, n, device=dev)
        eik = torch.empty(B, 2, device=dev)
        _lib.check(L.dh_render_scan_fwd(_p(rays_o), _p(rays_d), _p(z_vals), _p(sdf), _p(normals), _p(colors), _p(inv_s),
                                        float(cos_anneal_ratio), sample_dist, _p(background_rgb), B, n, _p(weights),
                                        _p(color), _p(wsum), _p(wmax), _p(cdf), _p(inside), _p(eik), _lib.stream()))
        eik_sum = eik.sum(dim=0)
        gradient_error = eik_sum[0] / (eik_sum[1] + 1e-5)
        gradients = normals.view(B, n, 3)
        ctx.set_materialize_grads(False)
        renderer._ws_token += 1
        ctx.ws_token = renderer._ws_token
        ctx.renderer = renderer
        ctx.meta = (B, n, float(cos_anneal_ratio), sample_dist)
        ctx.save_for_backward(rays_o, rays_d, z_vals, pts, sdf, normals, colors, inv_s, eik_sum, background_rgb
                              if background_rgb is not None else torch.empty(0, device=dev))
        ctx.has_bg = background_rgb is not None
        ctx.ws = ws
        ctx.mark_non_differentiable(wmax, cdf, inside)
        return color, wsum, gradient_error, weights, gradients, wmax, cdf, inside, sdf.view(P, 1)

    @staticmethod
    def backward(ctx, d_color, d_wsum, d_ge, d_weights, d_gradients, _wmax, _cdf, _inside, d_sdf_out):
        renderer = ctx.renderer
        return (None, None, None, None, None, None) + renderer._backward(ctx, d_color, d_wsum, d_ge, d_weights,
                                                                         d_gradients, d_sdf_out)


class NeuSRenderer:
    def __init__(self, nerf, sdf_network: SDFNetwork, deviation_network: SingleVarianceNetwork,
                 color_network: RenderingNetwork, n_samples, n_importance, n_outside, up_sample_steps, perturb,
                 store: ParamStore | None = None, device="cuda"):
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
        self.store = store if store is not None else ParamStore(sdf_network, deviation_network, color_network, device)
        self._ws = None
        self._ws_token = 0

    # ------------------------------------------------------------------ workspace (caller-owned, reused)
    def _workspace(self, npts: int) -> torch.Tensor:
        _, total = _lib.workspace_floats(npts)
        if self._ws is None or self._ws.numel() < total:
            self._ws = None
            self._ws = torch.empty(total, device=self.store.device, dtype=torch.float32)
        return self._ws

    # ------------------------------------------------------------------ no-grad SDF queries
    def sdf(self, pts: torch.Tensor) -> torch.Tensor:
        """sdf_network.sdf(pts) under no_grad: [N,3] -> [N,1]."""
        pts = pts.contiguous().float()
        out = torch.empty(pts.shape[0], device=pts.device)
        _lib.check(_lib.lib().dh_sdf_nograd(_p(self.store.ensure_packed()), _p(pts), pts.shape[0], _p(out), _lib.stream()))
        return out.view(-1, 1)

    # ------------------------------------------------------------------ hierarchical sampling (App. A.5/A.6)
    @torch.no_grad()
    def sample_z(self, rays_o, rays_d, near, far, perturb_overwrite=-1, t_rand=None):
        L = _lib.lib()
        packed = self.store.ensure_packed()
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
            _lib.check(L.dh_sdf_nograd(_p(packed), _p(pts), B * ns, _p(sdf), _lib.stream()))
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
                    _lib.check(L.dh_sdf_nograd(_p(packed), _p(pts_new), B * n_new, _p(sdf_new), _lib.stream()))
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
        n = z_vals.shape[1]
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

    # ------------------------------------------------------------------ backward (filled in by renderer_bwd)
    def _backward(self, ctx, d_color, d_wsum, d_ge, d_weights, d_gradients, d_sdf_out):
        L = _lib.lib()
        st = self.store
        if ctx.ws_token != self._ws_token:
            raise RuntimeError("NeuSRenderer workspace was overwritten by a later render() call before backward(); "
                               "call backward() before rendering again (one live graph per renderer)")
        rays_o, rays_d, z_vals, pts, sdf, normals, colors, inv_s, eik_sum, bg = ctx.saved_tensors
        bg = bg if ctx.has_bg else None
        B, n, car, sample_dist = ctx.meta
        P = B * n
        dev = rays_o.device
        zero = lambda *s: torch.zeros(*s, device=dev)
        d_color = d_color.contiguous() if d_color is not None else zero(B, 3)
        d_wsum = d_wsum.contiguous() if d_wsum is not None else None
        d_weights = d_weights.contiguous() if d_weights is not None else None
        d_gradients = d_gradients.contiguous() if d_gradients is not None else None
        d_ge = d_ge if d_ge is not None else zero(())
        eik_coef = (d_ge / (eik_sum[1] + 1e-5)).reshape(1).contiguous()
        d_sdf = torch.empty(P, device=dev)
        d_normals = torch.empty(P, 3, device=dev)
        d_colors = torch.empty(P, 3, device=dev)
        d_inv_s = torch.empty(B, device=dev)
        _lib.check(L.dh_render_scan_bwd(_p(rays_o), _p(rays_d), _p(z_vals), _p(sdf), _p(normals), _p(colors), _p(inv_s),
                                        car, sample_dist, _p(bg), B, n, _p(d_color), _p(d_wsum), _p(d_weights),
                                        _p(d_gradients), _p(eik_coef), _p(d_sdf), _p(d_normals), _p(d_colors),
                                        _p(d_inv_s), _lib.stream()))
        if d_sdf_out is not None:
            d_sdf = d_sdf + d_sdf_out.reshape(-1)
        grad = torch.empty(st.n, device=dev)
        _lib.check(L.dh_mlp_backward(_p(st.packed), _p(st.flat), _p(pts), P, _p(ctx.ws), _p(colors), _p(d_sdf),
                                     _p(d_normals), _p(d_colors), _p(grad), _lib.stream()))
        # variance: inv_s = clip(exp(10 v), 1e-6, 1e6)
        raw = torch.exp(st.flat[st.var_off] * 10.0)
        passthrough = ((raw >= 1e-6) & (raw <= 1e6)).float()
        grad[st.var_off] = d_inv_s.sum() * 10.0 * raw * passthrough
        st.grad_flat = grad
        return tuple(grad[off:off + cnt].view(p.shape) for p, off, cnt in st.slices)
