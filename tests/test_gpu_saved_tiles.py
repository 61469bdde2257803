"""The training forward's saved activation tiles (dh_sdf_forward, default arithmetic), read back directly.

Round 6 moved the register-resident chain's transposition (lane-per-point -> native tile) onto the matrix pipe: the m-tile's two
fp16 pieces against a constant selector (csrc/chain_t.hip, T_SAVE_MFMA).  What a saved activation IS changed with it: (hi + lo) / 16,
the value the next layer's GEMM consumed, instead of the fp32 activation (2^-22 apart).  Two checks:
  * against the fp64 oracle's activations, layer by layer, at the tolerance of the sdf itself (2e-5 absolute);
  * the defining property, size-independent: 16 x every saved value is EXACTLY the sum of its own two fp16 pieces
    (hi = fp16(X), lo = fp16(X - hi): X == hi + lo in fp32) -- an fp32 activation with 24 significant bits is not.
"""
import pytest
import torch

from tests.util import flat_from_oracle, randomized_models

pytestmark = pytest.mark.gpu

TILE_F, TM = 16384, 64
ABSMAX_FLOATS, TMAX_N = 4096, 21


def _native_to_rows(t, nt):
    """[nt * TILE_F] native tiles (csrc/tile.h: float4 index (((w*2 + m)*2 + t)*4 + r4)*64 + lane) -> [nt * 64 points, 256 features]."""
    x = t.view(nt, 4, 2, 2, 4, 64, 4)                                      # tile, w, m, t, r4, lane, rr
    lane = torch.arange(64, device=t.device)
    rr = torch.arange(4, device=t.device)
    out = torch.empty(nt, 64, 256, device=t.device, dtype=t.dtype)
    for w in range(4):
        for m in range(2):
            for tt in range(2):
                for r4 in range(4):
                    rows = (m * 32 + 8 * r4 + 4 * (lane >> 5))[:, None] + rr[None, :]
                    cols = (64 * w + 32 * tt + (lane & 31))[:, None].expand(64, 4)
                    out[:, rows, cols] = x[:, w, m, tt, r4]
    return out.view(nt * 64, 256)


def _oracle_activations(sdf, pts):
    """softplus(lin_l) for l = 0..7 in fp64, formed as oracle/neus_oracle.py SDFNetwork.forward forms them."""
    import math
    from oracle import neus_oracle as O
    net = sdf.double()
    inputs = pts.double() * net.scale
    e = O.embed(inputs, net.multires) if net.multires > 0 else inputs
    x = e
    acts = []
    for l in range(net.num_layers - 1):
        lin = getattr(net, "lin" + str(l))
        if l in net.skip_in:
            x = torch.cat([x, e], 1) / math.sqrt(2)
        x = lin(x)
        if l < net.num_layers - 2:
            x = net.activation(x)
            acts.append(x)
    return acts


@pytest.mark.parametrize("npts", [64, 129, 1000, 8192 + 37])
def test_saved_activation_tiles_match_the_oracle_and_are_two_piece_sums(hiplib, npts):
    from dynhor_amd import _lib
    dev = torch.device("cuda:0")
    sdf, col, var = randomized_models(seed=23, device=dev, jitter=0.05)
    flat = flat_from_oracle(sdf, var, col)
    g = torch.Generator(device="cpu").manual_seed(npts)
    pts = ((torch.rand(npts, 3, generator=g) * 2 - 1) * 0.9).to(dev)
    packed = torch.empty(hiplib.dh_packed_floats(), device=dev)
    _lib.check(hiplib.dh_pack_weights(_lib.ptr(flat), _lib.ptr(packed), _lib.stream()))
    fwd_floats = _lib.workspace_floats(npts)[1]
    ws = torch.zeros(fwd_floats, device=dev)
    out = torch.full((npts,), float("nan"), device=dev)
    _lib.check(hiplib.dh_sdf_forward(_lib.ptr(packed), _lib.ptr(pts), npts, _lib.ptr(ws), _lib.ptr(out), _lib.stream()))
    torch.cuda.synchronize()
    nt = (npts + TM - 1) // TM
    act0 = ABSMAX_FLOATS + (TMAX_N * nt + 3) // 4 * 4                       # csrc/workspace.h carve_workspace
    with torch.no_grad():
        ref = _oracle_activations(sdf, pts)
        sdf.float()
    assert len(ref) == 8
    worst = 0.0
    for l in range(8):
        tiles = ws[act0 + l * nt * TILE_F: act0 + (l + 1) * nt * TILE_F]
        got = _native_to_rows(tiles, nt)[:npts]
        width = ref[l].shape[1]                                              # 217 for lin3 (the skip layer's input), 256 otherwise
        err = (got[:, :width].double() - ref[l]).abs().max().item()
        worst = max(worst, err)
        assert err < 2e-5, (l, err)
        # the defining property of the matrix-pipe transposition: X = 16 * saved is hi + lo of its own two fp16 pieces, exactly
        X = got[:, :width] * 16.0
        hi = X.half().float()
        lo = (X - hi).half().float()
        exact = (hi + lo) == X
        assert bool(exact.all()), (l, int((~exact).sum()), float((hi + lo - X).abs().max()))
    print(f"npts={npts}: max |act - fp64 oracle| over the 8 layers = {worst:.2e}")


def test_relaunch_of_the_training_forward_is_bitwise_stable(hiplib):
    from dynhor_amd import _lib
    dev = torch.device("cuda:0")
    sdf, col, var = randomized_models(seed=5, device=dev, jitter=0.05)
    flat = flat_from_oracle(sdf, var, col)
    npts = 32768 + 64
    pts = ((torch.rand(npts, 3, generator=torch.Generator().manual_seed(1)) * 2 - 1) * 0.9).to(dev)
    packed = torch.empty(hiplib.dh_packed_floats(), device=dev)
    _lib.check(hiplib.dh_pack_weights(_lib.ptr(flat), _lib.ptr(packed), _lib.stream()))
    fwd_floats = _lib.workspace_floats(npts)[1]
    first = None
    for _ in range(50):
        ws = torch.zeros(fwd_floats, device=dev)
        out = torch.empty(npts, device=dev)
        _lib.check(hiplib.dh_sdf_forward(_lib.ptr(packed), _lib.ptr(pts), npts, _lib.ptr(ws), _lib.ptr(out), _lib.stream()))
        torch.cuda.synchronize()
        if first is None:
            first = (ws.clone(), out.clone())
        else:
            assert torch.equal(ws, first[0]) and torch.equal(out, first[1])
