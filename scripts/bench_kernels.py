"""Micro-benchmarks of individual C-ABI entry points (HIP events on torch's current stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynhor_amd import _lib
from tests.util import flat_from_oracle, randomized_models


def timeit(fn, warm=3, it=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it


def main():
    dev = torch.device("cuda:0")
    L = _lib.lib()
    sdf, col, var = randomized_models(seed=3, device=dev)
    flat = flat_from_oracle(sdf, var, col)
    packed = torch.empty(L.dh_packed_floats(), device=dev)
    ms = timeit(lambda: _lib.check(L.dh_pack_weights(_lib.ptr(flat), _lib.ptr(packed), _lib.stream())))
    print(f"pack_weights: {ms*1e3:.1f} us")
    for n in (32768, 131072, 262144):
        pts = (torch.rand(n, 3, device=dev) * 2 - 1)
        out = torch.empty(n, device=dev)
        ms = timeit(lambda: _lib.check(L.dh_sdf_nograd(_lib.ptr(packed), _lib.ptr(pts), n, _lib.ptr(out), _lib.stream())))
        fl = n * 918016.0
        print(f"sdf_nograd n={n}: {ms:.3f} ms  {fl/ms/1e9:.1f} TFLOP/s ({fl/ms/1e9/157.3*100:.1f}% of fp32 MFMA peak)")


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def bench_hashgrid():
    """Hash-grid encode fwd/bwd: algorithmic bytes = 16 levels x 8 corners x 8 B gathered + 12 B read + 128 B written per
    point (forward); backward 1 KiB of float atomics per point.  Points along rays (coherent) vs uniformly random."""
    dev = torch.device("cuda:0")
    L = _lib.lib()
    ne = L.dh_hashgrid_entries()
    table = torch.randn(ne, 2, device=dev)
    n = 262144
    g = torch.Generator(device="cpu").manual_seed(0)
    rnd = torch.rand(n, 3, generator=g).to(dev)
    o = torch.nn.functional.normalize(torch.randn(2048, 3, generator=g), dim=-1) * 2.2
    d = torch.nn.functional.normalize(-o + (torch.rand(2048, 3, generator=g) - 0.5) * 0.6, dim=-1)
    t = torch.linspace(1.4, 3.0, 128)[None, :, None]
    ray_pts = ((o[:, None] + d[:, None] * t).reshape(-1, 3).clamp(-1, 1) * 0.5 + 0.5).contiguous().to(dev)
    out = torch.empty(n, 32, device=dev)
    dout = torch.randn(n, 32, device=dev)
    dtab = torch.zeros(ne, 2, device=dev)
    for name, pts in (("uniform random points", rnd), ("2048 rays x 128 samples", ray_pts)):
        ms = timeit(lambda: _lib.check(L.dh_hashgrid_encode(_lib.ptr(table), _lib.ptr(pts), n, _lib.ptr(out), _lib.stream())))
        by = n * (16 * 8 * 8 + 12 + 128)
        print(f"hashgrid fwd, {name}: {ms*1e3:.1f} us  {by/ms/1e6:.0f} GB/s algorithmic ({by/ms/1e6/8000*100:.1f}% of 8 TB/s HBM)")
        ms = timeit(lambda: _lib.check(L.dh_hashgrid_encode_backward(_lib.ptr(pts), _lib.ptr(dout), n, _lib.ptr(dtab), _lib.stream())))
        by = n * (16 * 8 * 8 + 12 + 128)
        print(f"hashgrid bwd, {name}: {ms*1e3:.1f} us  {by/ms/1e6:.0f} GB/s of atomic+read bytes (float-atomic rate ~1.3 TB/s)")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "hashgrid":
    bench_hashgrid()
