"""The shipping kernels form every fp32 product from bf16 pieces on the bf16 matrix cores;
dh_set_arithmetic(DH_ARITH_FP32_MFMA) selects the native fp32-MFMA twin of every kernel.  Both must produce the same
full-size training step: losses to 5e-6, flat gradient to 1e-5 relative (each is 3.7e-6 from the eager-fp32 oracle,
tests/test_gpu_fullsize_and_runner.py).  Every arithmetic mode the ABI can select is exercised here."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_split_bf16_and_fp32_mfma_kernels_agree(tmp_path):
    from dynhor_amd import _lib
    from dynhor_amd.runner import Runner
    conf = {"seq_name": "t", "exp_name": "modes", "data_info": {"synthetic": {"n_frames": 4, "H": 128, "W": 128, "seed": 11}},
            "train": {"batch_size": 2048, "normal_weight": 0.05, "report_freq": 10 ** 9, "save_freq": 10 ** 9, "val_freq": 0}}
    r = Runner(conf=conf, device="cuda:0", exp_root=str(tmp_path))
    g = torch.Generator(device="cuda:0"); g.manual_seed(5)
    rays = r.dataset.gen_random_rays_at(1, 2048, generator=g)
    near, far = r.dataset._last_near_far
    t_rand = torch.rand(2048, 1, device="cuda:0", generator=g)
    res = {}
    assert _lib.get_arithmetic() == _lib.ARITH_SPLIT_BF16, "split-bf16 is the default arithmetic"
    try:
        for mode in (_lib.ARITH_SPLIT_BF16, _lib.ARITH_FP32_MFMA):
            _lib.set_arithmetic(mode)
            assert _lib.get_arithmetic() == mode
            stats = r.renderer.train_step_core(rays, near, far, r.dataset.R[1], 0.3, 0.1, 0.1, 0.05, t_rand=t_rand)
            torch.cuda.synchronize()
            res[mode] = (stats.clone(), r.store.grad_flat.clone(), r.renderer.last_state.z_vals.clone())
        # a full-frame forward-only chunk in both modes (the save = 0 variants)
        o, d = rays[:, :3].contiguous(), rays[:, 3:6].contiguous()
        cols = {}
        for mode in (_lib.ARITH_SPLIT_BF16, _lib.ARITH_FP32_MFMA):
            _lib.set_arithmetic(mode)
            st = r.renderer._forward_core(o, d, res[0][2], 0.3, None, want_nmap=True, infer_only=True)
            cols[mode] = (st.color.clone(), st.nmap.clone())
    finally:
        _lib.set_arithmetic(_lib.ARITH_SPLIT_BF16)
    with pytest.raises(_lib.DynhorHipError):
        _lib.set_arithmetic(7)
    a, b = res[_lib.ARITH_SPLIT_BF16], res[_lib.ARITH_FP32_MFMA]
    ds = (a[0][:6] - b[0][:6]).abs().max().item()
    rel = ((a[1].double() - b[1].double()).norm() / b[1].double().norm()).item()
    dz = (a[2] - b[2]).abs()
    print(f"split-bf16 vs fp32-MFMA: loss/stat max abs diff {ds:.2e}; flat gradient rel diff {rel:.2e}; "
          f"sampled z: {float((dz > 1e-4).float().mean()):.2e} of samples differ by > 1e-4")
    assert ds < 5e-6
    assert rel < 1e-5
    assert float((dz > 1e-4).float().mean()) < 2e-3        # ill-conditioned inverse-CDF samples only
    assert (cols[0][0] - cols[1][0]).abs().max().item() < 2e-5 and (cols[0][1] - cols[1][1]).abs().max().item() < 2e-4


def test_two_renderers_of_one_process_can_run_different_arithmetics(tmp_path):
    """the arithmetic is a library-wide word, but the Python mirror sets it per renderer before each
    stage group, so interleaved steps of a split-bf16 Runner and an fp32-MFMA Runner reproduce their own solo results."""
    from dynhor_amd import _lib
    from dynhor_amd.runner import Runner
    base = {"seq_name": "t", "data_info": {"synthetic": {"n_frames": 4, "H": 96, "W": 96, "seed": 11}},
            "train": {"batch_size": 512, "normal_weight": 0.05, "report_freq": 10 ** 9, "save_freq": 10 ** 9, "val_freq": 0}}
    mk = lambda name, ar: Runner(conf={**base, "exp_name": name, "model": {"arithmetic": ar}}, device="cuda:0", exp_root=str(tmp_path))
    try:
        solo = {}
        for ar in ("split_bf16", "fp32_mfma"):
            r = mk("solo_" + ar, ar)
            for _ in range(3):
                r.train_iteration()
            solo[ar] = r.store.flat.clone()
        a, b = mk("a", "split_bf16"), mk("b", "fp32_mfma")
        for _ in range(3):
            a.train_iteration()
            b.train_iteration()
        torch.cuda.synchronize()
        assert torch.equal(a.store.flat, solo["split_bf16"]) and torch.equal(b.store.flat, solo["fp32_mfma"])
        assert not torch.equal(a.store.flat, b.store.flat), "the two arithmetics differ in the last bits"
    finally:
        _lib.set_arithmetic(_lib.ARITH_SPLIT_BF16)
