"""GPU: multiresolution hash-grid encoding (BASELINE.json configs[3] building block) against oracle/hashgrid_oracle.py
(parity unpinned: restatement of Mueller et al. 2022 §3, see the oracle header)."""
import ctypes

import pytest
import torch

from oracle import hashgrid_oracle as H

pytestmark = pytest.mark.gpu


def _encode(hiplib, table, x01):
    from dynhor_amd import _lib
    out = torch.full((x01.shape[0], 32), float("nan"), device=x01.device)
    _lib.check(hiplib.dh_hashgrid_encode(_lib.ptr(table), _lib.ptr(x01), x01.shape[0], _lib.ptr(out), _lib.stream()))
    return out


def test_level_layout_matches_oracle(hiplib):
    enc = H.HashGridEncoding()
    assert hiplib.dh_hashgrid_entries() == enc.n_entries
    for l in range(16):
        s, r, o, d = ctypes.c_float(), ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        assert hiplib.dh_hashgrid_level(l, ctypes.byref(s), ctypes.byref(r), ctypes.byref(o), ctypes.byref(d)) == 0
        assert r.value == enc.resolutions[l] and o.value == enc.offsets[l] and bool(d.value) == enc.dense[l]
        assert abs(s.value - enc.scales[l]) < 1e-3 * max(1.0, enc.scales[l])
    assert hiplib.dh_hashgrid_level(16, ctypes.byref(s), ctypes.byref(r), ctypes.byref(o), ctypes.byref(d)) == -1


@pytest.mark.parametrize("n", [1, 17, 4096, 100003])
def test_encode_forward_matches_oracle(hiplib, n):
    dev = torch.device("cuda:0")
    torch.manual_seed(n)
    enc = H.HashGridEncoding().to(dev)
    with torch.no_grad():
        enc.table.copy_(torch.randn_like(enc.table))                 # O(1) entries so every corner matters
    x01 = torch.rand(n, 3, device=dev)
    x01[: min(n, 8)] = torch.tensor([[0, 0, 0], [1, 1, 1], [1, 0, 0.5], [0.5, 1, 0], [0.25, 0.75, 1], [1, 1, 0], [0, 1, 1],
                                     [0.999999, 0.5, 0.5]], device=dev)[: min(n, 8)]      # faces / corners of the box
    if n >= 17:      # outside the box (the renderer's last mid-point can overshoot): uint32 wrap, same rows as the oracle
        x01[8:16] = torch.rand(8, 3, device=dev) * 1.2 - 0.1
        x01[16] = torch.tensor([-0.05, 0.5, 1.08], device=dev)
    got = _encode(hiplib, enc.table.detach().contiguous(), x01)
    with torch.no_grad():
        ref = enc(x01)
        ref64 = enc.double()(x01.double())
    e_hip = (got.double() - ref64).abs().max().item()
    e_t32 = (ref.double() - ref64).abs().max().item()
    print(f"n={n}: |hip-f64|={e_hip:.2e} |torch32-f64|={e_t32:.2e}")
    # indices must agree exactly (a wrong corner is an O(1) error); values to fp32 rounding of the position scaling
    assert e_hip < 2e-3 and e_hip < 20 * e_t32 + 1e-5


def test_encode_backward_matches_oracle_autograd(hiplib):
    from dynhor_amd import _lib
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    enc = H.HashGridEncoding().to(dev)
    with torch.no_grad():
        enc.table.copy_(torch.randn_like(enc.table))
    n = 20000
    x01 = torch.rand(n, 3, device=dev) * 0.2 + 0.4             # concentrated: many points share corners (atomic sums)
    d_out = torch.randn(n, 32, device=dev)
    d_table = torch.zeros_like(enc.table)
    _lib.check(hiplib.dh_hashgrid_encode_backward(_lib.ptr(x01), _lib.ptr(d_out.contiguous()), n, _lib.ptr(d_table),
                                                  _lib.stream()))
    enc64 = enc.double()
    out = enc64(x01.double())
    (out * d_out.double()).sum().backward()
    ref = enc64.table.grad
    err = (d_table.double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    print("table-gradient max err", err, "of max", scale)
    assert err < 1e-4 * scale
    assert ((d_table != 0) == (ref != 0)).float().mean().item() > 0.9999, "same table rows touched"


def test_hash_collisions_and_empty_input(hiplib):
    from dynhor_amd import _lib
    dev = torch.device("cuda:0")
    t = torch.zeros(hiplib.dh_hashgrid_entries(), 2, device=dev)
    x = torch.zeros(1, 3, device=dev)
    o = torch.zeros(1, 32, device=dev)
    assert hiplib.dh_hashgrid_encode(_lib.ptr(t), _lib.ptr(x), 0, _lib.ptr(o), _lib.stream()) == 0
    assert hiplib.dh_hashgrid_encode(_lib.ptr(t), _lib.ptr(x), -3, _lib.ptr(o), _lib.stream()) == -1
    # finest level: 2049^3 grid points share 2^19 rows -> many-to-one; encoding stays a convex blend of table rows
    t.uniform_(0.0, 1.0)
    x = torch.rand(5000, 3, device=dev)
    out = torch.empty(5000, 32, device=dev)
    _lib.check(hiplib.dh_hashgrid_encode(_lib.ptr(t), _lib.ptr(x), 5000, _lib.ptr(out), _lib.stream()))
    assert (out >= -1e-6).all() and (out <= 1 + 1e-6).all()
