"""The shipping kernels form every fp32 product from two fp16 pieces per operand and three matrix-core products (round 4;
DESIGN.md section 3); DH_ARITH_SPLIT_BF16 is the three-piece bf16 / six-product form of rounds 1-3 and DH_ARITH_FP32_MFMA the
native fp32-MFMA twin of every kernel.  All three must produce the same full-size training step: losses to 5e-6, flat gradient
to 1e-5 relative (each is ~4e-6 from the eager-fp32 oracle, tests/test_gpu_fullsize_and_runner.py).  Every arithmetic mode the
ABI can select is exercised here, through the stateless `_ex` entry points (the renderer passes its arithmetic per call)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_the_three_arithmetics_agree_on_a_full_size_step(tmp_path):
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
    assert _lib.get_arithmetic() == _lib.ARITH_SPLIT_F16 == _lib.ARITH_DEFAULT, "the two-piece fp16 split is the default arithmetic"
    assert r.renderer.arithmetic is None
    modes = (_lib.ARITH_SPLIT_F16, _lib.ARITH_SPLIT_BF16, _lib.ARITH_FP32_MFMA)
    try:
        for mode in modes:
            r.renderer.arithmetic = mode            # passed with every launch; the library's default word is not touched
            stats = r.renderer.train_step_core(rays, near, far, r.dataset.R[1], 0.3, 0.1, 0.1, 0.05, t_rand=t_rand)
            torch.cuda.synchronize()
            res[mode] = (stats.clone(), r.store.grad_flat.clone(), r.renderer.last_state.z_vals.clone())
        # a full-frame forward-only chunk in every mode (the save = 0 variants)
        o, d = rays[:, :3].contiguous(), rays[:, 3:6].contiguous()
        cols = {}
        for mode in modes:
            r.renderer.arithmetic = mode
            st = r.renderer._forward_core(o, d, res[_lib.ARITH_FP32_MFMA][2], 0.3, None, want_nmap=True, infer_only=True)
            cols[mode] = (st.color.clone(), st.nmap.clone())
        assert _lib.get_arithmetic() == _lib.ARITH_DEFAULT
        # the default word is honoured by the entry points without an arithmetic argument and validated
        for mode in modes:
            _lib.set_arithmetic(mode)
            assert _lib.get_arithmetic() == mode
        with pytest.raises(_lib.DynhorHipError):
            _lib.set_arithmetic(7)
    finally:
        r.renderer.arithmetic = None
        _lib.set_arithmetic(_lib.ARITH_DEFAULT)
    b = res[_lib.ARITH_FP32_MFMA]
    for name, mode in (("split-f16", _lib.ARITH_SPLIT_F16), ("split-bf16", _lib.ARITH_SPLIT_BF16)):
        a = res[mode]
        ds = (a[0][:6] - b[0][:6]).abs().max().item()
        rel = ((a[1].double() - b[1].double()).norm() / b[1].double().norm()).item()
        dz = (a[2] - b[2]).abs()
        print(f"{name} vs fp32-MFMA: loss/stat max abs diff {ds:.2e}; flat gradient rel diff {rel:.2e}; "
              f"sampled z: {float((dz > 1e-4).float().mean()):.2e} of samples differ by > 1e-4")
        assert ds < 5e-6
        assert rel < 1e-5
        assert float((dz > 1e-4).float().mean()) < 2e-3        # ill-conditioned inverse-CDF samples only
        assert (cols[mode][0] - cols[_lib.ARITH_FP32_MFMA][0]).abs().max().item() < 2e-5
        assert (cols[mode][1] - cols[_lib.ARITH_FP32_MFMA][1]).abs().max().item() < 2e-4


def test_renderers_with_different_arithmetics_run_side_by_side_also_from_two_threads(tmp_path):
    """no launch depends on a process-global word -- every renderer passes its arithmetic through the `_ex`
    entry points.  Interleaved steps of three Runners (one per arithmetic) reproduce their solo results bit for bit, and so do
    two Runners stepped concurrently from two host threads (each on its own HIP stream)."""
    import threading
    from dynhor_amd import _lib
    from dynhor_amd.runner import Runner
    base = {"seq_name": "t", "data_info": {"synthetic": {"n_frames": 4, "H": 96, "W": 96, "seed": 11}},
            "train": {"batch_size": 512, "normal_weight": 0.05, "report_freq": 10 ** 9, "save_freq": 10 ** 9, "val_freq": 0}}
    mk = lambda name, ar: Runner(conf={**base, "exp_name": name, "model": {"arithmetic": ar}}, device="cuda:0", exp_root=str(tmp_path))
    names = ("split_f16", "split_bf16", "fp32_mfma")
    solo = {}
    for ar in names:
        r = mk("solo_" + ar, ar)
        for _ in range(3):
            r.train_iteration()
        solo[ar] = r.store.flat.clone()
    trio = {ar: mk("i_" + ar, ar) for ar in names}
    for _ in range(3):
        for ar in names:
            trio[ar].train_iteration()
    torch.cuda.synchronize()
    for ar in names:
        assert torch.equal(trio[ar].store.flat, solo[ar]), ar
    assert not torch.equal(solo["split_f16"], solo["fp32_mfma"]) and not torch.equal(solo["split_f16"], solo["split_bf16"]), \
        "the arithmetics differ in the last bits"
    assert _lib.get_arithmetic() == _lib.ARITH_DEFAULT, "nobody touched the default word"
    # two host threads, two streams, two arithmetics
    pair = {ar: mk("t_" + ar, ar) for ar in ("split_f16", "fp32_mfma")}
    torch.cuda.synchronize()          # the Runners' buffers were filled on the default stream
    errs = []

    def work(ar):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(3):
                    pair[ar].train_iteration()
                torch.cuda.current_stream().synchronize()
        except Exception as e:          # noqa: BLE001
            errs.append((ar, repr(e)))

    ts = [threading.Thread(target=work, args=(ar,)) for ar in pair]
    [t.start() for t in ts]
    [t.join() for t in ts]
    torch.cuda.synchronize()
    assert not errs, errs
    for ar in pair:
        assert torch.equal(pair[ar].store.flat, solo[ar]), ar
