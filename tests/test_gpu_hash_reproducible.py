"""GPU: the hash family's table scatter in its reproducible form (include/dynhor_hip.h dh_hash_weight_grads_parts, parts bit 4;
VERDICT r4 next #4 "an optional fully deterministic mode tested bitwise").

Float atomics give sums that depend on the order in which the memory side sees the requests (the only order-dependent sums of the
library).  With bit 4 (the default of dh_hash_weight_grads and of the Python mirror: it costs nothing) every contribution is converted
to 2^-48 fixed point and added by an INTEGER atomic (associative: any order gives the same int64), converted to float once.

  * the table gradient of the bench-sized step, re-launched on one workspace: bit-identical (the float form is shown to differ);
  * against the float-atomic form and the fp64 oracle: the same gradient (the fixed-point form is the closer of the two);
  * two training runs with the same seeds: bit-identical parameters, for both samplers;
  * a non-finite adjoint or a single contribution beyond 64: the WHOLE table gradient is NaN;
  * many contributions BELOW that limit whose sum on one entry passes 16,384 (round 6, ADVICE r5: round 5's limit of 16,384 per
    contribution let three adds wrap the int64): that entry is NaN, never a finite wrong value."""
import pytest
import torch

from tests.test_gpu_hash_family import _oracle_grads, _rays, make_hash_pair

pytestmark = pytest.mark.gpu


def _runner(root, tag, reproducible, sampler="hierarchical", frames=8):
    from dynhor_amd.runner import Runner
    conf = {"seq_name": "rep", "exp_name": tag, "data_info": {"synthetic": {"n_frames": frames, "H": 128, "W": 128, "seed": 77}},
            "train": {"batch_size": 2048, "learning_rate": 5e-3, "normal_weight": 0.05, "report_freq": 10 ** 9, "save_freq": 10 ** 9,
                      "val_freq": 0, "warm_up_end": 50, "end_iter": 1000},
            "model": {"family": "hash", "hash_renderer": {"sampler": sampler, "reproducible_table_grad": reproducible}}}
    return Runner(conf=conf, device="cuda:0", exp_root=str(root))


def test_table_gradient_relaunched_on_one_workspace_is_bit_identical_and_equals_the_float_form(tmp_path):
    from dynhor_amd import _lib
    from dynhor_amd.renderer import _p
    r = _runner(tmp_path, "one", True)
    for _ in range(5):
        r.train_iteration()
    torch.cuda.synchronize()
    L, st, s = _lib.lib(), r.store, r.renderer.last_state
    P = s.B * s.n
    ntab = st.table_floats

    def table_grad(parts):
        g = torch.full_like(st.grad_flat, float("nan"))
        _lib.check(L.dh_hash_weight_grads_parts(_p(st.flat), _p(st.packed), P, _p(s.ws), _p(g), None, parts, _lib.stream()))
        torch.cuda.synchronize()
        return g[:ntab].clone()
    ref = table_grad(5)
    assert torch.isfinite(ref).all() and ref.abs().max().item() > 0
    for rep in range(30):
        assert torch.equal(table_grad(5).view(torch.int32), ref.view(torch.int32)), f"launch {rep} differs"
    fl = [table_grad(1) for _ in range(6)]
    differing = sum(int(not torch.equal(f, fl[0])) for f in fl[1:])
    rel = ((fl[0].double() - ref.double()).norm() / ref.double().norm()).item()
    worst = (fl[0].double() - ref.double()).abs().max().item() / ref.abs().max().item()
    print(f"float-atomic launches differing from the first: {differing} of 5; float vs fixed point: rel L2 {rel:.2e}, max |d| / max |g| {worst:.2e}")
    assert rel < 1e-6 and worst < 1e-6
    # parts 7 = 5 then 2: the whole gradient vector, the small linears untouched by the bit
    g7 = torch.zeros_like(st.grad_flat)                      # (the vector's tail -- the variance parameter -- is not this stage's)
    g3 = torch.zeros_like(st.grad_flat)
    _lib.check(L.dh_hash_weight_grads_parts(_p(st.flat), _p(st.packed), P, _p(s.ws), _p(g7), None, 7, _lib.stream()))
    _lib.check(L.dh_hash_weight_grads_parts(_p(st.flat), _p(st.packed), P, _p(s.ws), _p(g3), None, 3, _lib.stream()))
    torch.cuda.synchronize()
    assert torch.equal(g7[:ntab], ref) and torch.equal(g7[ntab:], g3[ntab:])
    assert L.dh_hash_weight_grads_parts(_p(st.flat), _p(st.packed), P, _p(s.ws), _p(g3), None, 4, _lib.stream()) == -1                 # DH_ERR_BAD_ARG
    assert L.dh_hash_weight_grads_parts(_p(st.flat), _p(st.packed), P, _p(s.ws), _p(g3), None, 6, _lib.stream()) == -1                 # DH_ERR_BAD_ARG


def test_fixed_point_table_gradient_matches_the_fp64_oracle():
    o_r, p_r = make_hash_pair(seed=5)
    p_r.reproducible_table_grad = True
    B, car = 48, 0.6
    rays_o, rays_d, near, far = _rays(B, seed=2)
    g = torch.Generator(device="cpu").manual_seed(9)
    t_rand = torch.rand(B, 1, generator=g).cuda()
    tgt = torch.rand(B, 3, generator=g).cuda()
    with torch.no_grad():
        z = o_r.sample_z(rays_o, rays_d, near, far, t_rand=t_rand)

    def loss_fn(out):
        t = tgt.to(out["color_fine"].dtype)
        return (out["color_fine"] - t).abs().mean() + 0.1 * out["gradient_error"]
    ref, ref_loss, gref = _oracle_grads(o_r, rays_o, rays_d, near, far, z, car, loss_fn, torch.float64)
    eager, eager_loss, geager = _oracle_grads(o_r, rays_o, rays_d, near, far, z, car, loss_fn, torch.float32)
    errs = {}
    for rep_mode in (True, False):
        p_r.reproducible_table_grad = rep_mode
        out = p_r.render(rays_o, rays_d, near, far, cos_anneal_ratio=car, z_vals=z)
        loss_fn(out).backward()
        torch.cuda.synchronize()
        ntab = p_r.store.table_floats
        got = p_r.store.grad_flat[:ntab].double()
        errs[rep_mode] = ((got - gref[:ntab]).norm() / gref[:ntab].norm()).item()
    e_eager = ((geager[:p_r.store.table_floats] - gref[:p_r.store.table_floats]).norm() / gref[:p_r.store.table_floats].norm()).item()
    print(f"table gradient rel L2 vs fp64: fixed point {errs[True]:.2e}, float atomics {errs[False]:.2e}, eager fp32 oracle {e_eager:.2e}")
    assert errs[True] < max(1e-3, 3 * e_eager) and errs[True] < 1.5 * errs[False] + 1e-7


@pytest.mark.parametrize("sampler", ["hierarchical", "occgrid"])
def test_two_hash_training_runs_with_the_same_seeds_are_bitwise_identical(tmp_path, sampler):
    a = _runner(tmp_path, "a" + sampler, True, sampler); a.train(n_iters=80)
    b = _runner(tmp_path, "b" + sampler, True, sampler); b.train(n_iters=80)
    torch.cuda.synchronize()
    assert torch.isfinite(a.store.flat).all()
    assert torch.equal(a.store.flat, b.store.flat)


def test_concurrent_and_serial_weight_gradient_parts_give_identical_parameters(tmp_path):
    """The table scatter and the small linears' weight-gradient GEMMs run on two streams (HashNeuSRenderer._weight_grads): disjoint
    outputs, joined before the optimiser -- the same bits as one stream."""
    a = _runner(tmp_path, "conc", True)
    b = _runner(tmp_path, "serial", True)
    b.renderer.concurrent_weight_grads = False
    a.train(n_iters=40); b.train(n_iters=40)
    torch.cuda.synchronize()
    assert getattr(a.renderer, "_dw_stream", None) is not None and getattr(b.renderer, "_dw_stream", None) is None
    assert torch.equal(a.store.flat, b.store.flat)


def test_non_finite_or_out_of_range_contributions_turn_the_whole_table_gradient_into_nan():
    o_r, p_r = make_hash_pair(seed=6)
    p_r.reproducible_table_grad = True
    B = 16
    rays_o, rays_d, near, far = _rays(B, seed=4)
    ntab = p_r.store.table_floats

    def table_grad(scale):
        out = p_r.render(rays_o, rays_d, near, far, cos_anneal_ratio=0.5)
        (out["color_fine"].sum() * scale).backward()
        torch.cuda.synchronize()
        return p_r.store.grad_flat[:ntab].clone()
    g = table_grad(1.0)
    assert torch.isfinite(g).all() and g.abs().max().item() > 0
    big = table_grad(1.0e12)                                   # finite, but single contributions beyond 64
    assert torch.isnan(big).all()
    assert torch.isfinite(table_grad(1.0)).all()               # the flag is cleared by the next launch
    nan = table_grad(float("inf"))
    assert torch.isnan(nan).all()


def test_many_sub_limit_contributions_on_one_entry_end_in_the_guard_band_not_in_a_wrapped_sum():
    """16,384 samples at ONE position: every level's eight corner entries collect all of them.  Consecutive samples of a wave are merged
    before they reach memory (16 per add), so an entry's sum is 1,024 adds of equal size.  Scale the adjoint so that the largest
    entry's SUM passes 2^14 = 16,384 while every single add stays below the per-contribution limit of 64: the entry must be NaN (a
    wrapped int64 -- what round 5's limit of 16,384 per contribution allowed -- would be finite and wrong by ~2^16); every entry whose
    sum stays below the guard band must be finite and equal to the scaled float-atomic value."""
    from dynhor_amd import _lib
    from dynhor_amd.renderer import _p
    o_r, p_r = make_hash_pair(seed=7)
    L, st = _lib.lib(), p_r.store
    N = 16384
    pts = torch.tensor([[0.1234, -0.2345, 0.3456]], device="cuda").repeat(N, 1).contiguous()
    ws = p_r._workspace(N, infer_only=False)
    sdf = torch.empty(N, device="cuda"); feat = torch.empty(N, 13, device="cuda"); nrm = torch.empty(N, 3, device="cuda")
    packed = st.ensure_packed()
    stream = _lib.stream()
    _lib.check(L.dh_hash_geo_forward(_p(st.flat), _p(packed), _p(pts), N, p_r.radius, p_r.fd_eps, _p(ws), 1, _p(sdf), _p(feat), _p(nrm), None, stream))
    ntab = st.table_floats
    d_feat = torch.zeros(N, 13, device="cuda"); d_n = torch.zeros(N, 3, device="cuda")

    def table_grad(scale, parts):
        d_sdf = torch.full((N,), float(scale), device="cuda")
        _lib.check(L.dh_hash_geo_backward(_p(st.flat), _p(packed), _p(pts), _p(d_sdf), _p(d_feat), _p(d_n), N, p_r.radius, p_r.fd_eps, _p(ws),
                                          None, stream))
        g = torch.full((st.n,), float("nan"), device="cuda")
        _lib.check(L.dh_hash_weight_grads_parts(_p(st.flat), _p(packed), N, _p(ws), _p(g), None, parts, stream))
        torch.cuda.synchronize()
        return g[:ntab].clone()
    g1 = table_grad(1.0, 1).double()                               # float atomics, unit adjoint
    gmax = g1.abs().max().item()
    assert torch.isfinite(g1).all() and gmax > 0
    for target in (1.2 * 2 ** 14, 2.0 * 2 ** 14, 2.9 * 2 ** 14):
        s = target / gmax
        got = table_grad(s, 5)
        want = g1 * s
        over = want.abs() >= 2 ** 14 * 1.001
        under = want.abs() <= 2 ** 14 * 0.999
        assert over.any()
        assert not torch.isnan(got).all(), f"target {target:.3g}: a single add of {target / 1024:.1f} tripped the per-contribution limit of 64"
        assert torch.isnan(got[over]).all(), "an entry whose sum passed 2^14 came back finite"
        fin = torch.isfinite(got)
        assert fin[under].all()
        rel = ((got[fin].double() - want[fin]).abs().max() / want.abs().max()).item()
        print(f"target {target:.3g}: {int(over.sum())} entries in the guard band are NaN, the other {int(fin.sum())} finite, max |d| / max |g| {rel:.2e}")
        assert rel < 1e-5
    assert torch.isfinite(table_grad(1.0, 5)).all()
