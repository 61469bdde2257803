"""The register-resident chains' weight streams (csrc/layout.h PACKT: three bf16 pieces; PACKH: two fp16 pieces of S_w W): every
16-byte fragment of them, decoded on the host, must hold the
effective weights W = g v / |v| at the position the header documents -- stage order lin0 (3 k-steps), lin1-3 (16 each), lin4 (14
of lin3's output + 3 of the embedding, both scaled 1/sqrt 2), lin5-7, lin8 rows 1..256; bf16x8 index ((stage*8 + M)*3 + piece)*64 +
lane = W[32 M + (lane & 31)][k], slot i <-> input feature 16 s + 8 (i/4) + 4 (lane >> 5) + (i % 4); then the ten bias rows."""
import math

import pytest
import torch

from tests.util import flat_from_oracle, randomized_models

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("arith", ["bf16x3", "f16x2"])
def test_tstream_holds_the_effective_weights_where_layout_h_says(hiplib, arith):
    from dynhor_amd import _lib
    dev = torch.device("cuda:0")
    sdf, col, var = randomized_models(seed=5, device=dev, jitter=0.05)
    flat = flat_from_oracle(sdf, var, col)
    packed = torch.empty(hiplib.dh_packed_floats(), device=dev)
    _lib.check(hiplib.dh_pack_weights(_lib.ptr(flat), _lib.ptr(packed), _lib.stream()))
    torch.cuda.synchronize()
    sd = {k: v.detach().double().cpu() for k, v in sdf.state_dict().items()}
    W = {}
    for l in range(9):
        v, g = sd["lin%d.weight_v" % l], sd["lin%d.weight_g" % l].reshape(-1)
        W[l] = g[:, None] * v / v.norm(dim=1, keepdim=True)
    n_stage = 132
    if arith == "bf16x3":
        stream0, n = _lib.packed_section(0)
        bias0, _ = _lib.packed_section(1)
        pieces = packed[stream0:stream0 + n].view(torch.bfloat16).view(n_stage, 8, 3, 64, 8).float()      # stage, M, piece, lane, slot
        w = (pieces[:, :, 0] + pieces[:, :, 1] + pieces[:, :, 2]).cpu()                              # [stage, M, lane, slot]
        wscale = {l: 1.0 for l in range(9)}
    else:
        stream0, n = _lib.packed_section(2)
        bias0, _ = _lib.packed_section(3)
        pieces = packed[stream0:stream0 + n].view(torch.float16).view(n_stage, 8, 2, 64, 8).double()
        w = (pieces[:, :, 0] + pieces[:, :, 1]).cpu()                                                 # = S_w W
        wabs0, _ = _lib.packed_section(4)
        wabs = packed[wabs0:wabs0 + 16].view(torch.int32).cpu()
        wscale = {}
        for l in range(9):
            m = torch.tensor([int(wabs[l])], dtype=torch.int32).view(torch.float32).item()
            assert abs(m - W[l].abs().max().item()) <= 1e-6 * m, (l, m, W[l].abs().max().item())     # max |W| of the linear
            wscale[l] = 2.0 ** (3 - math.floor(math.log2(m)))                                          # scaled maximum in [8, 16)
            assert 8.0 <= wscale[l] * m < 16.0
    # (layer, k-steps, first row, first input column, valid input columns, scale) in stream order
    jobs = [(0, 3, 0, 0, 39, 1.0), (1, 16, 0, 0, 256, 1.0), (2, 16, 0, 0, 256, 1.0), (3, 16, 0, 0, 256, 1.0),
            (4, 14, 0, 0, 217, 1 / math.sqrt(2)), (4, 3, 0, 217, 39, 1 / math.sqrt(2)),
            (5, 16, 0, 0, 256, 1.0), (6, 16, 0, 0, 256, 1.0), (7, 16, 0, 0, 256, 1.0), (8, 16, 1, 0, 256, 1.0)]
    lane = torch.arange(64)
    slot = torch.arange(8)
    stage = 0
    worst = 0.0
    for l, nk, row0, col0, ncol, scale in jobs:
        rows_valid = W[l].shape[0] - row0
        for s in range(nk):
            k = 16 * s + 8 * (slot[None, :] // 4) + 4 * (lane[:, None] // 32) + (slot[None, :] % 4)             # [lane, slot]
            for M in range(8):
                o = 32 * M + (lane % 32)                                                                        # [lane]
                ref = torch.zeros(64, 8, dtype=torch.float64)
                ok = (o[:, None] < rows_valid) & (k < ncol)
                oo = (o[:, None] + row0).expand(64, 8)[ok]
                kk = (k + col0)[ok]
                ref[ok] = scale * W[l][oo, kk]
                err = (w[stage, M].double() / wscale[l] - ref).abs().max().item()
                worst = max(worst, err)
                assert err < 2e-6, (l, s, M, err)
            stage += 1
    assert stage == n_stage
    print("max |stream - W| = %.2e" % worst)
    # bias rows 0..7 (the fp16 chain carries activations scaled by 16: its rows are 16 x bias), lin8's effective row 0, lin8's bias
    # rows 1..256; the fp16 table's row 10: 1 / S_w of lin0..8
    nrow = 10 if arith == "bf16x3" else 11
    b = packed[bias0:bias0 + nrow * 256].view(nrow, 256).double().cpu()
    xs = 1.0 if arith == "bf16x3" else 16.0
    for l in range(8):
        ref = torch.zeros(256, dtype=torch.float64)
        ref[:sd["lin%d.bias" % l].numel()] = sd["lin%d.bias" % l]
        assert (b[l] - xs * ref).abs().max() < 2e-6
    if arith == "f16x2":
        for l in range(9):
            assert b[10, l].item() == 1.0 / wscale[l]
    assert (b[8] - W[8][0]).abs().max() < 2e-6
    assert (b[9] - sd["lin8.bias"][1:]).abs().max() < 1e-7


@pytest.mark.parametrize("ar", [0, 1, 2])
def test_pack_ex_serves_its_arithmetic_exactly_like_the_full_packer(hiplib, ar):
    """dh_pack_weights_ex(a) writes the common tables and arithmetic a's operands only: a forward + backward run from it must equal,
    bit for bit, the run from dh_pack_weights' buffer (a NaN-poisoned buffer shows anything the selective packer forgot)."""
    from dynhor_amd import _lib
    dev = torch.device("cuda:0")
    sdf, col, var = randomized_models(seed=7, device=dev, jitter=0.05)
    flat = flat_from_oracle(sdf, var, col)
    full = torch.empty(hiplib.dh_packed_floats(), device=dev)
    _lib.check(hiplib.dh_pack_weights(_lib.ptr(flat), _lib.ptr(full), _lib.stream()))
    part = torch.full((hiplib.dh_packed_floats(),), float("nan"), device=dev)
    _lib.check(hiplib.dh_pack_weights_ex(ar, _lib.ptr(flat), _lib.ptr(part), _lib.stream()))
    g = torch.Generator(device="cpu").manual_seed(3)
    nrays, n_per_ray = 32, 64
    npts = nrays * n_per_ray
    pts = ((torch.rand(npts, 3, generator=g) * 2 - 1) * 0.9).to(dev)
    dirs = torch.nn.functional.normalize(torch.randn(nrays, 3, generator=g), dim=-1).to(dev)
    d_sdf = torch.randn(npts, generator=g).to(dev) * 1e-3
    d_n = torch.randn(npts, 3, generator=g).to(dev) * 1e-3
    d_c = torch.randn(npts, 3, generator=g).to(dev) * 1e-3
    total = _lib.workspace_floats(npts)[2]
    outs = []
    for packed in (full, part):
        ws = torch.zeros(total, device=dev)
        o_sdf = torch.empty(npts, device=dev); o_n = torch.empty(npts, 3, device=dev); o_c = torch.empty(npts, 3, device=dev)
        _lib.check(hiplib.dh_mlp_forward_ex(ar, _lib.ptr(packed), _lib.ptr(pts), _lib.ptr(dirs), n_per_ray, npts, _lib.ptr(ws),
                                            _lib.ptr(o_sdf), _lib.ptr(o_n), _lib.ptr(o_c), _lib.stream()))
        grad = torch.zeros(flat.numel(), device=dev)
        dn = d_n.clone()
        _lib.check(hiplib.dh_mlp_backward_ex(ar, _lib.ptr(packed), _lib.ptr(flat), _lib.ptr(pts), npts, _lib.ptr(ws), _lib.ptr(o_c),
                                             _lib.ptr(d_sdf), _lib.ptr(dn), _lib.ptr(d_c), _lib.ptr(grad), _lib.stream()))
        nog = torch.empty(npts, device=dev)
        _lib.check(hiplib.dh_sdf_nograd_ex(ar, _lib.ptr(packed), _lib.ptr(pts), npts, _lib.ptr(nog), _lib.stream()))
        torch.cuda.synchronize()
        outs.append((o_sdf, o_n, o_c, grad, nog))
    for a, b in zip(*outs):
        assert torch.isfinite(b).all() and torch.equal(a, b)
    with pytest.raises(_lib.DynhorHipError):
        _lib.check(hiplib.dh_pack_weights_ex(5, _lib.ptr(flat), _lib.ptr(part), _lib.stream()))


def test_param_store_packs_only_what_a_renderer_asks_for_and_tracks_it(tmp_path):
    from dynhor_amd import _lib
    from dynhor_amd.runner import Runner
    conf = {"seq_name": "pk", "exp_name": "e", "data_info": {"synthetic": {"n_frames": 4, "H": 64, "W": 64, "seed": 3}},
            "train": {"batch_size": 256, "report_freq": 10 ** 9, "save_freq": 10 ** 9, "val_freq": 0}}
    r = Runner(conf=conf, device="cuda:0", exp_root=str(tmp_path))
    st = r.store
    calls = []
    orig = st._pack
    st._pack = lambda arithmetic=None: (calls.append(arithmetic), orig(arithmetic))[1]
    r.train_iteration()
    assert calls == [_lib.ARITH_DEFAULT]                       # one operand set per step
    st.ensure_packed(_lib.ARITH_DEFAULT); assert len(calls) == 2   # (the optimiser step changed the parameters)
    st.ensure_packed(_lib.ARITH_DEFAULT); assert len(calls) == 2
    st.ensure_packed(_lib.ARITH_FP32_MFMA); assert calls[-1] == _lib.ARITH_FP32_MFMA and len(calls) == 3
    st.ensure_packed(); assert calls[-1] is None and len(calls) == 4            # everything
    st.ensure_packed(_lib.ARITH_SPLIT_BF16); assert len(calls) == 4              # covered by "everything"
