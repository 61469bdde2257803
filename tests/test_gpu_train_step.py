import pytest
import torch

from oracle import neus_oracle as O
from tests.test_gpu_render_forward import make_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny_dataset():
    from dynhor_amd.dataset import Dataset
    return Dataset.from_synthetic(n_frames=3, H=96, W=96, seed=7, device="cuda:0")


def _oracle_frames(ds):
    return {"rgb": ds.rgb, "label": ds.label, "normal": ds.normal, "R": ds.R, "T": ds.T, "K": ds.K}


def test_gen_rays_matches_oracle(tiny_dataset):
    ds = tiny_dataset
    g = torch.Generator(device="cpu").manual_seed(0)
    px = torch.randint(0, ds.W, [500], generator=g).cuda()
    py = torch.randint(0, ds.H, [500], generator=g).cuda()
    for f in range(ds.n_images):
        got = ds.gen_rays_at_pixels(f, px, py)
        ref = O.gather_rays(_oracle_frames(ds), f, px, py)
        assert got.shape == ref.shape == (500, 14)
        assert (got[:, 9:11] == ref[:, 9:11]).all(), "mask decode (obj / keep) must be exact"
        # u8/255: torch's GPU scalar division multiplies by a rounded reciprocal, the kernel divides: <= 1 ulp apart
        assert (got[:, 6:9] - ref[:, 6:9]).abs().max().item() <= 6e-8
        assert (got - ref).abs().max().item() < 2e-6
        near, far = ds._last_near_far
        rn, rf = O.near_far_from_sphere(ref[:, :3], ref[:, 3:6])
        assert (near - rn).abs().max().item() < 1e-5 and (far - rf).abs().max().item() < 1e-5
    # labels present: object, background and hand
    lab = ds.label
    assert (lab == 1).any() and (lab == 0).any() and (lab == -1).any()


@pytest.mark.parametrize("normal_w", [0.0, 0.05])
def test_fused_train_step_matches_oracle(tiny_dataset, normal_w):
    ds = tiny_dataset
    o_r, p_r = make_pair(seed=33, jitter=0.05, n_samples=32, n_importance=32)
    B, frame, car = 96, 1, 0.4
    g = torch.Generator(device="cpu").manual_seed(5)
    px = torch.randint(0, ds.W, [B], generator=g).cuda()
    py = torch.randint(0, ds.H, [B], generator=g).cuda()
    t_rand = torch.rand(B, 1, generator=g).cuda()
    rays = ds.gen_rays_at_pixels(frame, px, py)
    near, far = ds._last_near_far
    R = ds.R[frame]

    # oracle: same z (sampling parity is covered elsewhere), fp64 autograd
    z = o_r.sample_z(rays[:, :3], rays[:, 3:6], near, far, t_rand=t_rand)
    mods = (o_r.sdf_network, o_r.deviation_network, o_r.color_network)
    for m in mods:
        m.double(); m.zero_grad()
    r64 = rays.double()
    out = o_r.render(r64[:, :3], r64[:, 3:6], near.double(), far.double(), cos_anneal_ratio=car, z_vals=z.double())
    ref = O.neus_losses(out, r64[:, 6:9], r64[:, 9:10], r64[:, 10:11], 0.1, 0.1, normal_w, r64[:, 11:14], R.double())
    ref["loss"].backward()
    gref = torch.cat([p.grad.reshape(-1) for m in mods for p in m.parameters()])
    for m in mods:
        m.float()

    # product: fused path on the same z (monkeypatch sampler to the oracle's z)
    p_r.sample_z = lambda *a, **k: z
    stats = p_r.train_step_core(rays, near, far, R, car, 0.1, 0.1, normal_w)
    torch.cuda.synchronize()
    got = p_r.store.grad_flat.double()
    names = ["loss", "color_loss", "eikonal_loss", "mask_loss", "normal_loss", "psnr"]
    for i, k in enumerate(names):
        if k in ref:
            e = abs(stats[i].item() - ref[k].item())
            print(f"{k}: hip={stats[i].item():.7f} ref={ref[k].item():.7f}")
            assert e < 2e-5 * max(1.0, abs(ref[k].item())), k
    rel = (got - gref).norm().item() / gref.norm().item()
    print("flat grad rel err", rel)
    assert rel < 1e-4


def test_fused_adam_matches_torch_adam():
    from dynhor_amd.fields import ParamStore, RenderingNetwork, SDFNetwork, SingleVarianceNetwork
    torch.manual_seed(0)
    st = ParamStore(SDFNetwork(), SingleVarianceNetwork(0.3), RenderingNetwork(), "cuda:0")
    ref_p = st.flat.clone().double().requires_grad_(True)
    opt = torch.optim.Adam([ref_p], lr=5e-4)
    for it in range(5):
        g = torch.randn(st.n, device="cuda:0") * (10.0 ** (it - 2))
        ref_p.grad = g.double()
        for grp in opt.param_groups:
            grp["lr"] = 5e-4 * (it + 1) / 5
        opt.step()
        st.adam_step(5e-4 * (it + 1) / 5, grad=g)
    torch.cuda.synchronize()
    err = (st.flat.double() - ref_p.detach()).abs().max().item()
    print("adam max err", err)
    assert err < 1e-6
    sd = st.optimizer_state_dict(5e-4)
    assert len(sd["state"]) == len(st.slices) and sd["param_groups"][0]["betas"] == (0.9, 0.999)
