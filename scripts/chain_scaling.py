"""How far is each chain kernel from its own steady state?  Times the C-ABI MLP stages on the bench's point counts and on 4-8x
as many (HIP events, random weights and points): if TFLOP/s rises with the size, launch ramp / tail / per-launch effects cost
time at the bench's size; if it does not, the loss is inside the per-tile loop (round 3)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dynhor_amd import _lib
from tests.util import flat_from_oracle, randomized_models

MACS = {"sdf_nograd": 459008, "sdf_forward": 524544, "sdf_gradient": 459008, "color_forward": 271360}


def timeit(fn, warm=3, it=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it


def main():
    dev = torch.device("cuda:0")
    L = _lib.lib()
    sdf, col, var = randomized_models(seed=3, device=dev)
    flat = flat_from_oracle(sdf, var, col)
    packed = torch.empty(L.dh_packed_floats(), device=dev)
    _lib.check(L.dh_pack_weights(_lib.ptr(flat), _lib.ptr(packed), _lib.stream()))
    res = {}
    for n in (32768, 131072, 262144, 1048576, 2097152):
        pts = (torch.rand(n, 3, device=dev) * 1.4 - 0.7)
        out = torch.empty(n, device=dev)
        ms = timeit(lambda: _lib.check(L.dh_sdf_nograd(_lib.ptr(packed), _lib.ptr(pts), n, _lib.ptr(out), _lib.stream())))
        res[f"sdf_nograd@{n}"] = {"ms": round(ms, 4), "tflops": round(2 * MACS["sdf_nograd"] * n / ms / 1e9, 1)}
        print(f"sdf_nograd   n={n:8d}: {ms:8.3f} ms  {2 * MACS['sdf_nograd'] * n / ms / 1e9:6.1f} TFLOP/s", flush=True)
    for n in (262144, 1048576):
        _, _, total = _lib.workspace_floats(n)
        ws = torch.empty(total, device=dev)
        pts = (torch.rand(n, 3, device=dev) * 1.4 - 0.7)
        dirs = torch.nn.functional.normalize(torch.randn(n // 128, 3, device=dev), dim=-1)
        sdfo = torch.empty(n, device=dev); nrm = torch.empty(n, 3, device=dev); colr = torch.empty(n, 3, device=dev)
        st = _lib.stream()
        for name, fn in (("sdf_forward", lambda: L.dh_sdf_forward(_lib.ptr(packed), _lib.ptr(pts), n, _lib.ptr(ws), _lib.ptr(sdfo), st)),
                         ("sdf_gradient", lambda: L.dh_sdf_gradient(_lib.ptr(packed), _lib.ptr(pts), n, _lib.ptr(ws), _lib.ptr(nrm), 1, st)),
                         ("color_forward", lambda: L.dh_color_forward(_lib.ptr(packed), _lib.ptr(pts), _lib.ptr(dirs), 128, _lib.ptr(nrm), n,
                                                                      _lib.ptr(ws), _lib.ptr(colr), 1, st))):
            ms = timeit(lambda: _lib.check(fn()), it=10)
            res[f"{name}@{n}"] = {"ms": round(ms, 4), "tflops": round(2 * MACS[name] * n / ms / 1e9, 1)}
            print(f"{name:13s}n={n:8d}: {ms:8.3f} ms  {2 * MACS[name] * n / ms / 1e9:6.1f} TFLOP/s", flush=True)
        del ws
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
