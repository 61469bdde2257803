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
        for net, mod, n_layers in ((0, sdf_network, 2), (2, color_network, 3)):
            for l in range(n_layers):
                b, g, v, out_dim, in_dim = hash_param_layout(net, l)
                lin = getattr(mod, "lin" + str(l))
                assert tuple(lin.weight_v.shape) == (out_dim, in_dim), (net, l, lin.weight_v.shape)
                slices += [(lin.bias, b, out_dim), (lin.weight_g, g, out_dim), (lin.weight_v, v, out_dim * in_dim)]
        _, _, voff, _, _ = hash_param_layout(1, 0)
        slices.append((deviation_network.variance, voff, 1))
        return int(L.dh_hash_num_params()), slices, voff, int(L.dh_hash_packed_floats())

    def _pack(self):
        _lib.check(_lib.lib().dh_hash_pack_weights(_lib.ptr(self.flat), _lib.ptr(self.packed), _lib.stream()))


class HashNeuSRenderer(NeuSRenderer):
    """NeuSRenderer over the hash-grid networks.  Same constructor, render() dict and train_step_core()."""

    def __init__(self, nerf, sdf_network: HashSDFNetwork, deviation_network: SingleVarianceNetwork,
                 color_network: SHRenderingNetwork, *args, **kwargs):
        if not isinstance(sdf_network, HashSDFNetwork) or not isinstance(color_network, SHRenderingNetwork):
            raise TypeError("HashNeuSRenderer needs HashSDFNetwork + SHRenderingNetwork")
        super().__init__(nerf, sdf_network, deviation_network, color_network, *args, **kwargs)
        self.radius = sdf_network.radius
        self.fd_eps = sdf_network.fd_eps

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
          _p(s.ws), 0 if s.infer_only else 1, _p(s.sdf), _p(s.feat), _p(s.normals), _lib.stream())
        T("hash_color_forward", L.dh_hash_color_forward, _p(packed), _p(s.feat), _p(s.normals), _p(s.rays_d), s.n, P,
          _p(s.colors), _lib.stream())

    def _net_backward(self, s, d_sdf, d_normals, d_colors, grad):
        L, T, st = _lib.lib(), self.timer, self.store
        P = s.B * s.n
        d_feat = torch.empty(P, 13, device=s.pts.device)
        T("hash_color_backward", L.dh_hash_color_backward, _p(st.packed), _p(s.feat), _p(s.normals), _p(s.rays_d),
          _p(d_colors), s.n, P, _p(s.ws), _p(d_feat), _p(d_normals), _lib.stream())
        T("hash_geo_backward", L.dh_hash_geo_backward, _p(st.flat), _p(st.packed), _p(s.pts), _p(d_sdf), _p(d_feat),
          _p(d_normals), P, self.radius, self.fd_eps, _p(s.ws), _lib.stream())
        T("hash_weight_grads", L.dh_hash_weight_grads, _p(st.flat), _p(st.packed), P, _p(s.ws), _p(grad), _lib.stream())


def build_hash_models(seed=1234, device="cuda"):
    """Same seeded construction order as oracle/hashgrid_oracle.py:build_models (tests copy weights across anyway)."""
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(seed)
        sdf = HashSDFNetwork()
        col = SHRenderingNetwork()
    return sdf, SingleVarianceNetwork(0.3), col
