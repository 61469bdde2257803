"""Host logic of the occupancy grid's refresh schedules (dynhor_amd/hash_fields.py refresh_mask / OccupancyGrid.update) against the
oracle restatement of nerfacc 0.3's rule (oracle/occgrid_oracle.py quarter_refresh_mask), on CPU tensors: same mask from the same
uniforms, nerfacc's inclusion rates, untouched cells keep their occupancy, the every-cell default equals the oracle's update."""
import math

import pytest
import torch

from oracle import occgrid_oracle as G


def _grids(res, occupied_fraction, seed):
    from dynhor_amd.hash_fields import OccupancyGrid
    g = torch.Generator().manual_seed(seed)
    occ = torch.rand(res ** 3, generator=g)
    binary = occ > (1.0 - occupied_fraction)
    p = OccupancyGrid(res=res, radius=1.0, device="cpu")
    o = G.OccupancyGrid(res=res, radius=1.0, device="cpu")
    p.occ, p.binary = occ.clone(), binary.to(torch.uint8)
    o.occ, o.binary = occ.clone(), binary.clone()
    return p, o, g


@pytest.mark.parametrize("occupied_fraction", [0.05, 0.6])
def test_quarter_refresh_mask_matches_the_oracle_and_has_nerfaccs_inclusion_rates(occupied_fraction):
    from dynhor_amd.hash_fields import refresh_mask
    p, o, g = _grids(32, occupied_fraction, seed=3)
    cells = 32 ** 3
    u = torch.rand(cells, generator=g)
    mask = refresh_mask(p.binary, u)
    assert torch.equal(mask, o.quarter_refresh_mask(u))
    n, k = cells // 4, int(o.binary.sum())
    p_uni = 1.0 - math.exp(-n / cells)
    p_occ = 1.0 if k <= n else 1.0 - math.exp(-n / k)
    free = mask[~o.binary].float().mean().item()
    held = mask[o.binary].float().mean().item()
    assert abs(free - p_uni) < 0.02                                   # a free cell is refreshed only by the uniform draw
    assert abs(held - (p_uni + (1 - p_uni) * p_occ)) < 0.02           # an occupied one by either draw


def test_grid_updates_match_the_oracle_for_both_schedules():
    p, o, g = _grids(16, 0.3, seed=5)
    cells = 16 ** 3
    jit = torch.rand(cells, 3, generator=g)
    sdf_fn = lambda x: x.norm(dim=-1) - 0.4
    inv_s, step = torch.tensor(64.0), 0.02
    before = p.occ.clone()
    u = torch.rand(cells, generator=g)
    mask = o.quarter_refresh_mask(u)
    p.update(sdf_fn, inv_s, step, jitter=jit, refresh="quarter", select=u)
    o.update(G.occ_alpha(sdf_fn(o.cell_points(jit)), inv_s, step), mask=mask)
    assert torch.equal(p.occ, o.occ) and torch.equal(p.binary.bool(), o.binary)
    assert torch.equal(p.occ[~mask], before[~mask]) and not torch.equal(p.occ[mask], before[mask])
    p.update(sdf_fn, inv_s, step, jitter=jit)                         # the default: every cell
    o.update(G.occ_alpha(sdf_fn(o.cell_points(jit)), inv_s, step))
    assert torch.equal(p.occ, o.occ) and torch.equal(p.binary.bool(), o.binary) and p.updates == 2
    with pytest.raises(ValueError):
        p.update(sdf_fn, inv_s, step, jitter=jit, refresh="half")
