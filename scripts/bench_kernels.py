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


if __name__ == "__main__":
    main()
