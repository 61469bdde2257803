"""Network parameter containers with the upstream-NeuS constructor signatures and state_dict keys
(SURVEY.md App. A.2/A.3, §5 checkpoint row): ``lin{l}.bias / lin{l}.weight_g / lin{l}.weight_v`` (legacy
weight_norm names), ``variance``.

These modules hold parameters only: every evaluation goes through the HIP kernels (dynhor_amd.renderer); there is
no eager forward.  All parameters of a model live in ONE flat fp32 device vector (ParamStore) in the order of
include/dynhor_hip.h:dh_param_layout, so the weight packer, the fused Adam and the RCCL gradient all-reduce each
touch a single contiguous buffer.
"""
from __future__ import annotations


import numpy as np
import torch
import torch.nn as nn

from . import _lib


class _WNLinearParams(nn.Module):
    """bias, weight_g, weight_v of a weight-normed Linear (registration order == legacy weight_norm state_dict)."""

    def __init__(self, weight: torch.Tensor, bias: torch.Tensor):
        super().__init__()
        self.bias = nn.Parameter(bias.clone())
        self.weight_g = nn.Parameter(weight.norm(dim=1, keepdim=True).clone())
        self.weight_v = nn.Parameter(weight.clone())

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("dynhor_amd modules have no eager forward; use NeuSRenderer (HIP path)")


class SDFNetwork(nn.Module):
    """Upstream SDFNetwork(d_in, d_out, d_hidden, n_layers, skip_in, multires, bias, scale, geometric_init,
    weight_norm, inside_outside) -- parameters + geometric init only (App. A.2)."""

    def __init__(self, d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=(4,), multires=6, bias=0.5, scale=1.0,
                 geometric_init=True, weight_norm=True, inside_outside=False):
        super().__init__()
        if (d_in, d_out, d_hidden, n_layers, tuple(skip_in), multires, scale, weight_norm, inside_outside) != \
                (3, 257, 256, 8, (4,), 6, 1.0, True, False):
            raise ValueError("unsupported SDFNetwork configuration for the gfx950 kernels (DH_ERR_UNSUPPORTED): "
                             "only the NeuS wmask configuration d_hidden=256, n_layers=8, skip_in=(4,), multires=6")
        d0 = d_in + 2 * d_in * multires
        dims = [d0] + [d_hidden] * n_layers + [d_out]
        self.num_layers = len(dims)
        self.skip_in = tuple(skip_in)
        self.scale = scale
        for l in range(self.num_layers - 1):
            out_dim = dims[l + 1] - dims[0] if (l + 1) in self.skip_in else dims[l + 1]
            w = torch.empty(out_dim, dims[l])
            b = torch.empty(out_dim)
            if geometric_init:
                if l == self.num_layers - 2:
                    nn.init.normal_(w, mean=np.sqrt(np.pi) / np.sqrt(dims[l]), std=0.0001)
                    nn.init.constant_(b, -bias)
                elif l == 0:
                    nn.init.constant_(b, 0.0)
                    nn.init.constant_(w[:, 3:], 0.0)
                    nn.init.normal_(w[:, :3], 0.0, np.sqrt(2) / np.sqrt(out_dim))
                elif l in self.skip_in:
                    nn.init.constant_(b, 0.0)
                    nn.init.normal_(w, 0.0, np.sqrt(2) / np.sqrt(out_dim))
                    nn.init.constant_(w[:, -(dims[0] - 3):], 0.0)
                else:
                    nn.init.constant_(b, 0.0)
                    nn.init.normal_(w, 0.0, np.sqrt(2) / np.sqrt(out_dim))
            else:
                lin = nn.Linear(dims[l], out_dim)
                w, b = lin.weight.data, lin.bias.data
            setattr(self, "lin" + str(l), _WNLinearParams(w, b))


class RenderingNetwork(nn.Module):
    """Upstream RenderingNetwork(d_feature, mode, d_in, d_out, d_hidden, n_layers, weight_norm, multires_view,
    squeeze_out) -- parameters only (App. A.3)."""

    def __init__(self, d_feature=256, mode="idr", d_in=9, d_out=3, d_hidden=256, n_layers=4, weight_norm=True,
                 multires_view=4, squeeze_out=True):
        super().__init__()
        if (d_feature, mode, d_in, d_out, d_hidden, n_layers, weight_norm, multires_view, squeeze_out) != \
                (256, "idr", 9, 3, 256, 4, True, 4, True):
            raise ValueError("unsupported RenderingNetwork configuration for the gfx950 kernels (DH_ERR_UNSUPPORTED)")
        d0 = d_in + d_feature + 2 * 3 * multires_view
        dims = [d0] + [d_hidden] * n_layers + [d_out]
        self.num_layers = len(dims)
        for l in range(self.num_layers - 1):
            lin = nn.Linear(dims[l], dims[l + 1])
            setattr(self, "lin" + str(l), _WNLinearParams(lin.weight.data, lin.bias.data))


class SingleVarianceNetwork(nn.Module):
    def __init__(self, init_val=0.3):
        super().__init__()
        self.register_parameter("variance", nn.Parameter(torch.tensor(float(init_val))))


class ParamStore:
    """One flat fp32 device vector holding every parameter (views handed back to the modules), its packed
    MFMA-operand image, the last flat gradient and the fused-Adam moments."""

    def __init__(self, sdf_network: SDFNetwork, deviation_network: SingleVarianceNetwork, color_network: RenderingNetwork,
                 device):
        _lib.lib()
        self.device = torch.device(device)
        self.modules = (sdf_network, deviation_network, color_network)
        self.n, self.slices, self.var_off, packed_floats = self._layout(sdf_network, deviation_network, color_network)
        self.flat = torch.empty(self.n, device=self.device, dtype=torch.float32)
        self.slices.sort(key=lambda s: s[1])
        end = 0
        for p, off, n in self.slices:
            assert off == end, "flat layout must be dense"
            end = off + n
            self.flat[off:off + n].copy_(p.detach().reshape(-1).to(self.device, torch.float32))
            p.data = self.flat[off:off + n].view(p.shape)
        assert end == self.n
        self.packed = torch.empty(packed_floats, device=self.device, dtype=torch.float32)
        self._packed_version = None
        self.grad_flat = None
        self._grad_bucket = None
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.step_count = 0
        self._manual_version = 0

    # -- model-family hooks (overridden by hash_fields.HashParamStore)
    def _layout(self, sdf_network, deviation_network, color_network):
        """(n_params, [(param, offset, numel)], variance offset, packed floats) per include/dynhor_hip.h:dh_param_layout."""
        L = _lib.lib()
        slices = []
        for net, mod in ((0, sdf_network), (2, color_network)):
            n_layers = 9 if net == 0 else 5
            for l in range(n_layers):
                b, g, v, out_dim, in_dim = _lib.param_layout(net, l)
                lin = getattr(mod, "lin" + str(l))
                assert tuple(lin.weight_v.shape) == (out_dim, in_dim), (net, l, lin.weight_v.shape)
                slices += [(lin.bias, b, out_dim), (lin.weight_g, g, out_dim), (lin.weight_v, v, out_dim * in_dim)]
        _, _, voff, _, _ = _lib.param_layout(1, 0)
        slices.append((deviation_network.variance, voff, 1))
        return int(L.dh_num_params()), slices, voff, int(L.dh_packed_floats())

    def _pack(self, arithmetic=None):
        """arithmetic None: the operands of all three arithmetics (dh_pack_weights); else only that one's (dh_pack_weights_ex)."""
        if arithmetic is None:
            _lib.check(_lib.lib().dh_pack_weights(_lib.ptr(self.flat), _lib.ptr(self.packed), _lib.stream()))
        else:
            _lib.check(_lib.lib().dh_pack_weights_ex(int(arithmetic), _lib.ptr(self.flat), _lib.ptr(self.packed), _lib.stream()))

    def params(self):
        return [p for p, _, _ in self.slices]

    def grad_bucket(self) -> torch.Tensor:
        """The ONE persistent flat gradient buffer of the fused training step: the data-parallel all-reduce always sees the
        same device address (RCCL registers a buffer once instead of once per iteration) and the hot loop allocates nothing
        of the gradient's size.  The autograd path (render() + loss.backward()) keeps allocating a fresh vector per backward,
        because autograd may keep views of it alive as the parameters' .grad."""
        if self._grad_bucket is None:
            self._grad_bucket = torch.empty(self.n, device=self.device, dtype=torch.float32)
        return self._grad_bucket

    def bump(self):
        """Call after any write to ``flat`` that bypasses torch (the fused Adam kernel)."""
        self._manual_version += 1

    def ensure_packed(self, arithmetic=None):
        """The packed-weights buffer, current for ``arithmetic`` (None: for every arithmetic).  The renderers ask for their own
        arithmetic only, so a training step packs one operand set, not three; which sets are current is tracked per parameter
        version, so a caller that switches arithmetic on unchanged parameters gets the missing set packed then."""
        # p.data aliases flat but keeps its own version counter, so fold every counter in
        ver = (self.flat._version, self._manual_version, sum(p._version for p, _, _ in self.slices))
        if ver != self._packed_version:
            self._packed_version = ver
            self._packed_sets = set()
        want = "all" if arithmetic is None else int(arithmetic)
        have = getattr(self, "_packed_sets", set())
        if "all" not in have and want not in have:
            self._pack(arithmetic)
            have.add(want)
            self._packed_sets = have
        return self.packed

    def inv_s(self) -> torch.Tensor:
        """clip(exp(10 variance), 1e-6, 1e6) as a 1-element device tensor (App. A.7)."""
        return torch.exp(self.flat[self.var_off:self.var_off + 1] * 10.0).clip(1e-6, 1e6)

    # ------------------------------------------------------------------ fused Adam (App. A.8)
    def adam_step(self, lr: float, betas=(0.9, 0.999), eps=1e-8, grad: torch.Tensor | None = None, grad_scale=1.0):
        """One torch.optim.Adam-equivalent step on the flat vector using the flat gradient of the last backward."""
        g = grad if grad is not None else self.grad_flat
        if g is None:
            raise RuntimeError("adam_step: no gradient (call backward() first)")
        self.step_count += 1
        _lib.check(_lib.lib().dh_adam_step(_lib.ptr(self.flat), _lib.ptr(g), _lib.ptr(self.exp_avg),
                                           _lib.ptr(self.exp_avg_sq), self.n, float(lr), float(betas[0]), float(betas[1]),
                                           float(eps), self.step_count, float(grad_scale), _lib.stream()))
        self.bump()

    def zero_grad(self):
        for p, _, _ in self.slices:
            p.grad = None
        self.grad_flat = None

    # torch.optim.Adam-compatible optimizer state (checkpoint key 'optimizer', App. A.8)
    def optimizer_state_dict(self, lr: float, betas=(0.9, 0.999), eps=1e-8):
        state = {}
        for i, (p, off, cnt) in enumerate(self.slices):
            state[i] = {"step": torch.tensor(float(self.step_count)),
                        "exp_avg": self.exp_avg[off:off + cnt].view(p.shape).clone(),
                        "exp_avg_sq": self.exp_avg_sq[off:off + cnt].view(p.shape).clone()}
        group = {"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(self.slices)))}
        return {"state": state if self.step_count > 0 else {}, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd):
        st = sd.get("state", {})
        if not st:
            self.exp_avg.zero_(); self.exp_avg_sq.zero_(); self.step_count = 0
            return
        for i, (p, off, cnt) in enumerate(self.slices):
            s = st[i] if i in st else st[str(i)]
            self.exp_avg[off:off + cnt].copy_(s["exp_avg"].reshape(-1))
            self.exp_avg_sq[off:off + cnt].copy_(s["exp_avg_sq"].reshape(-1))
            self.step_count = int(float(s["step"]))
