"""GPU: range behaviour of the two-piece fp16 arithmetic (include/dynhor_hip.h dh_range_words).

Every operand class of the MLP GEMMs is scaled dynamically -- per 64-point tile in the chains, per launch or per tile in the
weight-gradient kernel -- EXCEPT the softplus activations (and the unit-sphere embedding) of the register-resident SDF forward chain,
which ride at the constant scale 16: correct up to |activation| = 4094, and LOUD beyond (the input-gradient stage posts the largest
activation; NeuSRenderer.check_range / the Runner's report raise DynhorHipError; the step's numbers are NaN, never finite garbage).

  * features of 1e4 and colour ReLU activations of 1e5 (weights scaled up): loss and every parameter gradient match the fp64 oracle at
    the usual tolerances;
  * SDF activations pushed beyond 4094: check_range raises;
  * a SPLIT_F16 weight-gradient launch behind a forward of another arithmetic: NaN slabs (poisoned), not stale-scaled numbers."""
import pytest
import torch

from oracle import neus_oracle as O
from tests.test_gpu_render_backward import _loss_inputs
from tests.test_gpu_render_forward import make_pair, make_rays

pytestmark = pytest.mark.gpu


def _scaled_pair(scale_fn, seed=31):
    o_r, p_r = make_pair(seed=seed, jitter=0.05, n_samples=32, n_importance=32)
    with torch.no_grad():
        scale_fn(o_r.sdf_network, o_r.color_network)
    p_r.sdf_network.load_state_dict(o_r.sdf_network.state_dict())
    p_r.color_network.load_state_dict(o_r.color_network.state_dict())
    return o_r, p_r


def _grads(o_r, p_r, B=64, car=0.5, nw=0.05):
    dev = "cuda:0"
    o, d, near, far, t_rand = make_rays(B, seed=77)
    z = o_r.sample_z(o, d, near, far, t_rand=t_rand)
    rgb, obj, keep, mono, R = _loss_inputs(B, dev, seed=B)

    def run(renderer, dt):
        c = lambda t: t.to(dt)
        out = renderer.render(c(o), c(d), c(near), c(far), cos_anneal_ratio=car, z_vals=c(z))
        return out, O.neus_losses(out, c(rgb), c(obj), c(keep), igr_weight=0.1, mask_weight=0.1, normal_weight=nw, mono_normal=c(mono), R=c(R))["loss"]
    mods = (o_r.sdf_network, o_r.deviation_network, o_r.color_network)
    for m in mods:
        m.double(); m.zero_grad()
    out64, l64 = run(o_r, torch.float64)
    l64.backward()
    gref = [p.grad.detach().clone() for m in mods for p in m.parameters()]
    for m in mods:
        m.float()
    outp, lp = run(p_r, torch.float32)
    lp.backward()
    torch.cuda.synchronize()
    got = [p.grad.detach().double() for m in (p_r.sdf_network, p_r.deviation_network, p_r.color_network) for p in m.parameters()]
    return out64, l64.item(), gref, outp, lp.item(), got


def test_features_of_1e4_and_relu_activations_of_1e5_match_the_fp64_oracle():
    def scale(sdf, col):
        # feature vector = lin8 rows 1..256: x 1e4; colour lin0's outputs (ReLU activations): x 3e4 on top, lin1 damped so that the
        # network's output stays a usable colour
        sdf.lin8.weight_g[1:] *= 1.0e4
        sdf.lin8.bias[1:] *= 1.0e4
        col.lin0.weight_g *= 10.0
        col.lin1.weight_g *= 1.0 / 1.0e5
    o_r, p_r = _scaled_pair(scale)
    out64, l64, gref, outp, lp, got = _grads(o_r, p_r)
    # the magnitudes the test is about
    with torch.no_grad():
        pts = (torch.rand(4096, 3, device="cuda:0") - 0.5)
        o_r.sdf_network.double()
        f = o_r.sdf_network(pts.double())[:, 1:]
        o_r.sdf_network.float()
    fmax = f.abs().max().item()
    s = p_r.last_state if hasattr(p_r, "last_state") else None
    print(f"max |feature| = {fmax:.3g}; loss hip {lp:.8f} fp64 {l64:.8f}")
    assert fmax > 5e3, "the weights were meant to push the feature vector to 1e4"
    assert abs(lp - l64) < 1e-5 * max(1.0, abs(l64))
    assert torch.isfinite(outp["color_fine"]).all()
    assert (outp["color_fine"].double() - out64["color_fine"]).abs().max().item() < 2e-5
    worst = 0.0
    for g, r in zip(got, gref):
        if r.norm().item() > 0:
            worst = max(worst, ((g - r).norm() / r.norm()).item())
    flat_g = torch.cat([g.reshape(-1) for g in got]); flat_r = torch.cat([r.reshape(-1) for r in gref])
    rel = ((flat_g - flat_r).norm() / flat_r.norm()).item()
    print(f"flat gradient rel L2 vs fp64 {rel:.2e}; worst single parameter {worst:.2e}")
    assert rel < 2e-4 and worst < 2e-3


def test_relu_activations_of_1e5_are_inside_the_colour_chains_range():
    def scale(sdf, col):
        col.lin0.weight_g *= 3.0e4
        col.lin0.bias *= 3.0e4
        col.lin1.weight_g *= 1.0 / 3.0e4
    o_r, p_r = _scaled_pair(scale, seed=33)
    out64, l64, gref, outp, lp, got = _grads(o_r, p_r)
    flat_g = torch.cat([g.reshape(-1) for g in got]); flat_r = torch.cat([r.reshape(-1) for r in gref])
    rel = ((flat_g - flat_r).norm() / flat_r.norm()).item()
    print(f"ReLU x 3e4: loss hip {lp:.8f} fp64 {l64:.8f}; flat gradient rel L2 {rel:.2e}")
    assert abs(lp - l64) < 1e-5 * max(1.0, abs(l64)) and rel < 2e-4


def test_sdf_activations_beyond_the_constant_scales_range_are_reported_not_hidden():
    from dynhor_amd import _lib
    def scale(sdf, col):
        sdf.lin1.weight_g *= 3.0e4          # softplus(lin1) reaches ~1e4-1e5: beyond 4094
    o_r, p_r = _scaled_pair(scale, seed=35)
    o, d, near, far, t_rand = make_rays(64, seed=5)
    z = o_r.sample_z(o, d, near, far, t_rand=t_rand)
    s = p_r._forward_core(o, d, z, 0.5, None, want_nmap=False)
    torch.cuda.synchronize()
    a, _, lim = _lib.range_words()
    m = float(s.ws[a])
    print(f"posted max activation {m:.4g}, limit {lim:.0f}")
    assert m > lim
    with pytest.raises(_lib.DynhorHipError, match="split_f16 range exceeded"):
        p_r.check_range(s)
    # the same network in the three-piece bf16 arithmetic has no such limit and no range word to consult
    p_r.arithmetic = _lib.ARITH_SPLIT_BF16
    s2 = p_r._forward_core(o, d, z, 0.5, None, want_nmap=False)
    assert p_r.check_range(s2) is None and torch.isfinite(s2.color).all()
    # an ordinary network: a finite maximum well inside the range
    o_r2, p_r2 = make_pair(seed=5, n_samples=32, n_importance=32)
    s3 = p_r2._forward_core(o, d, z, 0.5, None, want_nmap=False)
    m3, lim3 = p_r2.check_range(s3)
    assert 0.0 < m3 < 100.0 and lim3 == lim


def test_weight_gradients_behind_a_forward_of_another_arithmetic_are_poisoned_not_stale():
    """include/dynhor_hip.h ONE ARITHMETIC PER STEP: the SPLIT_F16 weight-gradient kernel needs the scale tables its own forward /
    backward stages write; behind a bf16 step it finds no tag and writes NaN."""
    from dynhor_amd import _lib
    from dynhor_amd.renderer import _p
    o_r, p_r = make_pair(seed=7, n_samples=32, n_importance=32)
    o, d, near, far, t_rand = make_rays(64, seed=9)
    rays = torch.cat([o, d, torch.rand(64, 3, device="cuda:0"), torch.ones(64, 2, device="cuda:0"), torch.zeros(64, 3, device="cuda:0")], -1).contiguous()
    p_r.arithmetic = _lib.ARITH_SPLIT_BF16
    p_r.train_step_core(rays, near, far, None, 0.5, 0.1, 0.1, 0.0, t_rand=t_rand)
    torch.cuda.synchronize()
    assert torch.isfinite(p_r.store.grad_flat).all()
    s = p_r.last_state
    P = s.B * s.n
    L = _lib.lib()
    _lib.check(L.dh_weight_grads_gemm_ex(_lib.ARITH_SPLIT_F16, P, _p(s.ws), _lib.stream()))
    grad = torch.empty_like(p_r.store.grad_flat)
    _lib.check(L.dh_weight_grads_fold(_p(p_r.store.packed), _p(p_r.store.flat), P, _p(s.ws), _p(grad), _lib.stream()))
    torch.cuda.synchronize()
    assert torch.isnan(grad).any(), "a SPLIT_F16 weight-gradient launch on a bf16 step's workspace must not look valid"
    # and the regular SPLIT_F16 step on the same workspace is finite again
    p_r.arithmetic = _lib.ARITH_SPLIT_F16
    p_r.train_step_core(rays, near, far, None, 0.5, 0.1, 0.1, 0.0, t_rand=t_rand)
    torch.cuda.synchronize()
    assert torch.isfinite(p_r.store.grad_flat).all()
