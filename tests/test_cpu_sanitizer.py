"""SURVEY.md section 5 (race detection / sanitizers) -- VERDICT r3 next #8: the HOST side of libdynhor_hip.so under
AddressSanitizer + UndefinedBehaviorSanitizer.  GPU sanitizers are not available on this pool, and the device code needs none
for what is checked here: argument validation of every entry point, carve_workspace over a sweep of sizes, the parameter / packed
layouts, and the host tables the launch functions build before they enqueue anything (pack job lists of all three arithmetics,
the dW job list and its job-group partition, the fold table) -- without a GPU the launches themselves fail with DH_ERR_LAUNCH
AFTER that host logic has run, and no device pointer is ever dereferenced on the host.

The instrumented library is built here (hipcc -Xarch_host -fsanitize=address,undefined: host objects only, the gfx950 code objects
are the normal ones) into dynhor_amd/csrc/build_asan/ and reused while the sources are unchanged; it is loaded by a child python
with the sanitizer runtime preloaded."""
import glob
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dynhor_amd", "csrc")
OUT = os.path.join(CSRC, "build_asan")
LIB = os.path.join(OUT, "libdynhor_hip_asan.so")

DRIVER = r'''
import ctypes, sys
sys.path.insert(0, %(root)r)
from dynhor_amd import _lib
_lib.LIB_PATH = %(lib)r
L = _lib.lib()
vp = ctypes.c_void_p
null, fake = vp(0), vp(0x7f0000000000)          # a non-null, 16-byte aligned address that is never dereferenced on the host
i64 = ctypes.c_int64
assert L.dh_version() >= 3 and L.dh_num_params() == 802491
# layouts
for net, n in ((0, 9), (1, 1), (2, 5)):
    for l in range(n):
        _lib.param_layout(net, l)
for bad in ((0, 9), (0, -1), (2, 5), (3, 0)):
    try:
        _lib.param_layout(*bad); raise SystemExit("bad layer accepted")
    except _lib.DynhorHipError:
        pass
secs = [_lib.packed_section(i) for i in range(5)]
assert L.dh_packed_section(5, ctypes.byref(i64()), ctypes.byref(i64())) == -1 and L.dh_packed_section(0, None, None) == -1
# carve_workspace over ragged sizes (the tile count rounds up; every block stays 16-byte aligned)
prev = None
for n in (0, 1, 63, 64, 65, 127, 128, 4095, 4096, 100003, 262144, 1 << 22):
    a, b, c = _lib.workspace_floats(n)
    assert 0 < a <= b <= c and a %% 4 == 0 and b %% 4 == 0 and c %% 4 == 0
    assert prev is None or c >= prev
    prev = c
assert L.dh_workspace_floats(-1, ctypes.byref(i64()), ctypes.byref(i64()), ctypes.byref(i64())) == -1
# argument validation: null pointers, negative sizes, misaligned buffers, unknown arithmetic
mis = vp(0x7f0000000004)
assert L.dh_pack_weights(null, fake, None) == -1 and L.dh_pack_weights(fake, mis, None) == -1
assert L.dh_pack_weights_ex(9, fake, fake, None) == -1 and L.dh_pack_weights_ex(2, null, fake, None) == -1
for ar in (0, 1, 2):
    assert L.dh_sdf_nograd_ex(ar, null, null, 0, null, None) == 0
    assert L.dh_sdf_nograd_ex(ar, null, fake, 5, fake, None) == -1 and L.dh_sdf_nograd_ex(ar, fake, fake, -1, fake, None) == -1
    assert L.dh_sdf_forward_ex(ar, fake, fake, 100, mis, fake, None) == -1
    assert L.dh_sdf_gradient_ex(ar, fake, fake, 100, fake, fake, 3, None) == -1
    assert L.dh_color_forward_ex(ar, fake, fake, fake, 0, fake, 100, fake, fake, 1, None) == -1
    assert L.dh_weight_grads_gemm_ex(ar, 0, fake, None) == -1 and L.dh_weight_grads_gemm_ex(ar, 100, null, None) == -1
assert L.dh_sdf_nograd_ex(7, fake, fake, 5, fake, None) == -1 and L.dh_mlp_backward_ex(-1, *([fake] * 3), 100, *([fake] * 7)) == -1
assert L.dh_upsample_step(null, null, null, null, 4, 200, 16, 64.0, null, null, null) == -2
assert L.dh_set_arithmetic(9) == -1 and L.dh_hash_set_scatter_mode(5) == -1
# host tables behind the launches: without a GPU every launch fails, after the job lists / partitions / descriptors were built.
# The launch-type calls below pass FAKE device pointers: they may only run when no device can execute them.  The parent hides
# every device from this process (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES = "") and skips this part where /dev/kfd exists; the
# probe here is the last line of defence (hipGetDeviceCount through the runtime the library itself links).
if not %(launch_part)d:
    print("sanitized host paths ok (argument validation only: a GPU device node is present)")
    raise SystemExit(0)
hip = ctypes.CDLL("libamdhip64.so")
cnt = ctypes.c_int(-1)
rc = hip.hipGetDeviceCount(ctypes.byref(cnt))
assert rc != 0 or cnt.value == 0, "a HIP device is visible (%%d): refusing to enqueue kernels on fake pointers" %% cnt.value
no_gpu = L.dh_pack_weights(fake, fake, None)
assert no_gpu in (-3, -2), no_gpu
for ar in (0, 1, 2):
    assert L.dh_pack_weights_ex(ar, fake, fake, None) in (-3, -2)
for ar in (0, 1, 2):
    for n in (64, 100003, 262144):
        for rc in (L.dh_sdf_nograd_ex(ar, fake, fake, n, fake, None), L.dh_sdf_forward_ex(ar, fake, fake, n, fake, fake, None),
                   L.dh_sdf_gradient_ex(ar, fake, fake, n, fake, fake, 1, None),
                   L.dh_color_forward_ex(ar, fake, fake, fake, 128, fake, n, fake, fake, 1, None),
                   L.dh_color_backward_ex(ar, fake, fake, fake, n, fake, fake, None), L.dh_sdf_tangent_ex(ar, fake, fake, fake, n, fake, None),
                   L.dh_sdf_backward_ex(ar, fake, fake, n, fake, None), L.dh_weight_grads_gemm_ex(ar, n, fake, None),
                   L.dh_color_backward_rays_ex(ar, fake, fake, fake, fake, 128, n, fake, fake, fake, fake, None),
                   L.dh_sdf_backward_rays_ex(ar, fake, fake, fake, fake, n, fake, fake, None)):
            assert rc in (-3, -2), rc
assert L.dh_weight_grads_fold(fake, fake, 262144, fake, fake, None) in (-3, -2)
assert L.dh_hash_pack_weights(fake, fake, None) in (-3, -2)
hp = [i64() for _ in range(3)]
for net, layer in ((0, 0), (0, 1), (1, 0), (2, 0), (2, 1), (2, 2), (3, 0)):
    o, i = ctypes.c_int(), ctypes.c_int()
    assert L.dh_hash_param_layout(net, layer, ctypes.byref(hp[0]), ctypes.byref(hp[1]), ctypes.byref(hp[2]), ctypes.byref(o), ctypes.byref(i)) == 0
print("sanitized host paths ok")
'''


def _runtime():
    try:
        out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True,
                             text=True, timeout=60).stdout.strip()
    except (OSError, subprocess.TimeoutExpired):
        return None
    return out if out and os.path.isabs(out) and os.path.exists(out) else None


def _build():
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    newest = max(os.path.getmtime(d) for d in deps)
    if os.path.exists(LIB) and os.path.getmtime(LIB) >= newest:
        return
    os.makedirs(OUT, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fPIC", "-Xarch_host", "-fsanitize=address,undefined",
             "-Xarch_host", "-fno-omit-frame-pointer"]
    procs, objs = [], []
    for s in srcs:
        o = os.path.join(OUT, os.path.basename(s) + ".o")
        objs.append(o)
        procs.append(subprocess.Popen(["hipcc"] + flags + ["-c", s, "-o", o]))
    for p in procs:
        assert p.wait() == 0, "sanitizer build failed"
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address,undefined", "-shared-libsan",
                           "-o", LIB] + objs)


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not installed")
def test_host_side_of_the_library_is_clean_under_asan_and_ubsan(tmp_path):
    rt = _runtime()
    if rt is None:
        pytest.skip("no shared AddressSanitizer runtime in this toolchain")
    _build()
    script = tmp_path / "drive.py"
    # CPU-only by construction: the launch-type part enqueues on fake device pointers, so it runs only where no GPU
    # device node exists, and the child sees no device either way
    launch_part = 0 if (os.path.exists("/dev/kfd") or glob.glob("/dev/dri/renderD*")) else 1
    script.write_text(DRIVER % {"root": ROOT, "lib": LIB, "launch_part": launch_part})
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="",
               CUDA_VISIBLE_DEVICES="")
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0 and "sanitized host paths ok" in p.stdout, (p.stdout[-1500:] + p.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error:" not in p.stderr, p.stderr[-4000:]
