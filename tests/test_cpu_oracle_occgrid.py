"""CPU: the occupancy-grid marching oracle (oracle/occgrid_oracle.py) against closed forms: a ball of occupied cells crossed
by rays through its centre yields diameter / step samples, packed front to back; the per-ray cap keeps the first ones; the
packed compositing equals NeuS' dense render_core formula on the same samples; the grid update rule is nerfacc's."""
import torch

from oracle import neus_oracle as O
from oracle import occgrid_oracle as G


def _ball_grid(res=64, r=0.4):
    g = G.OccupancyGrid(res=res, radius=1.0)
    ax = (torch.arange(res) + 0.5) / res * 2 - 1
    x, y, z = torch.meshgrid(ax, ax, ax, indexing="ij")
    g.binary = ((x * x + y * y + z * z) < r * r).reshape(-1)
    return g


def test_ball_crossing_counts_order_and_cap():
    grid = _ball_grid()
    B = 8
    ang = torch.linspace(0, 3.0, B)
    o = torch.stack([2.2 * torch.cos(ang), 2.2 * torch.sin(ang), torch.zeros(B)], -1)
    d = -o / o.norm(dim=-1, keepdim=True)                       # through the centre
    near, far = O.near_far_from_sphere(o, d)
    step = 0.01
    u = torch.linspace(0.05, 0.95, B)
    m = G.march(o, d, near, far, u, grid, step, max_samples=128)
    # chord through the centre of a radius-0.4 ball = 0.8 -> 80 steps, up to the cell quantisation (cell = 1/32)
    assert ((m["cnt"] - 80).abs() <= 7).all(), m["cnt"]
    assert torch.equal(m["off"], torch.cumsum(m["cnt"], 0) - m["cnt"])
    for r in range(B):
        t = m["t_start"][m["off"][r]:m["off"][r] + m["cnt"][r]]
        assert (t[1:] > t[:-1]).all(), "front to back"
        k = (t - near[r]) / step - u[r]
        assert (k - k.round()).abs().max() < 1e-3, "t_k = near + (k + u) step"
        mid = o[r] + d[r] * (t[:, None] + 0.5 * step)
        assert (mid.norm(dim=-1) < 0.4 + 2.0 / 64 * 1.8).all() and grid.query(mid).all()
    capped = G.march(o, d, near, far, u, grid, step, max_samples=16)
    assert (capped["cnt"] == 16).all() and capped["truncated"].all()
    for r in range(B):
        full = m["t_start"][m["off"][r]:m["off"][r] + 16]
        assert torch.equal(capped["t_start"][capped["off"][r]:capped["off"][r] + 16], full), "the cap keeps the FIRST samples"
    # a ray that misses the ball has no samples; a point outside the cube is never occupied
    miss = G.march(torch.tensor([[2.0, 0.9, 0.0]]), torch.tensor([[-1.0, 0.0, 0.0]]), *O.near_far_from_sphere(
        torch.tensor([[2.0, 0.9, 0.0]]), torch.tensor([[-1.0, 0.0, 0.0]])), torch.tensor([0.5]), grid, step)
    assert int(miss["cnt"][0]) == 0
    full_grid = G.OccupancyGrid(res=8)
    assert not full_grid.query(torch.tensor([[1.5, 0.0, 0.0]])).any() and full_grid.query(torch.tensor([[0.99, -0.99, 0.0]])).all()


def test_packed_compositing_equals_dense_neus_formula():
    g = torch.Generator().manual_seed(0)
    B, cnt = 5, torch.tensor([7, 0, 12, 1, 30])
    off = torch.cumsum(cnt, 0) - cnt
    N = int(cnt.sum())
    ray_idx = torch.repeat_interleave(torch.arange(B), cnt)
    sdf = torch.randn(N, generator=g, dtype=torch.float64) * 0.05
    nrm = torch.nn.functional.normalize(torch.randn(N, 3, generator=g, dtype=torch.float64), dim=-1)
    col = torch.rand(N, 3, generator=g, dtype=torch.float64)
    dirs = torch.nn.functional.normalize(torch.randn(B, 3, generator=g, dtype=torch.float64), dim=-1)
    pts = torch.randn(N, 3, generator=g, dtype=torch.float64) * 0.4
    step, inv_s, car = 0.01, torch.tensor(40.0, dtype=torch.float64), 0.25
    out = G.render_packed(pts, sdf, nrm, col, dirs, ray_idx, off, cnt, step, inv_s, car)
    for r in range(B):
        sl = slice(int(off[r]), int(off[r] + cnt[r]))
        tc = (dirs[r] * nrm[sl]).sum(-1)
        ic = -(torch.relu(-tc * 0.5 + 0.5) * (1 - car) + torch.relu(-tc) * car)          # render_core, App. A.7
        prev = torch.sigmoid((sdf[sl] - ic * step * 0.5) * inv_s); nxt = torch.sigmoid((sdf[sl] + ic * step * 0.5) * inv_s)
        alpha = ((prev - nxt + 1e-5) / (prev + 1e-5)).clip(0, 1)
        w = alpha * torch.cumprod(torch.cat([torch.ones(1, dtype=torch.float64), 1 - alpha + 1e-7]), 0)[:-1]
        assert torch.allclose(out["weights"][sl], w, atol=1e-14)
        assert torch.allclose(out["color_fine"][r], (w[:, None] * col[sl]).sum(0), atol=1e-14)
        assert abs(out["weight_sum"][r, 0] - w.sum()) < 1e-14
    assert out["weight_sum"][1, 0] == 0 and (out["color_fine"][1] == 0).all()


def test_grid_update_rule():
    g = G.OccupancyGrid(res=4, decay=0.5, thre=0.01)
    a1 = torch.zeros(64); a1[:8] = 0.2
    g.update(a1)
    assert int(g.binary.sum()) == 8                      # mean = 0.025 -> threshold min(mean, 0.01) = 0.01
    g.update(torch.zeros(64))
    assert torch.allclose(g.occ[:8], torch.full((8,), 0.1)) and int(g.binary.sum()) == 8      # decayed, still above
    for _ in range(5):
        g.update(torch.zeros(64))
    assert int(g.binary.sum()) == 0 or g.occ.max() <= 0.01 + 1e-9                     # 0.1 * 0.5^5 = 0.003 < 0.01: pruned
    # alpha of one marching step: monotone in |sdf|, ~ step * inv_s / 2 at the surface for small step * inv_s
    a = G.occ_alpha(torch.tensor([0.0, 0.05, -0.05, 0.5]), torch.tensor(20.0), 0.01)
    assert a[0] > a[1] > a[3] and abs(a[0].item() - 0.0997) < 5e-3
