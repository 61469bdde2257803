"""dev: time the hash family's table-gradient stage with a variant library (probe builds of csrc/hash_mlp.hip: HASH_PROBE_*).
   python scripts/hash_probe.py dynhor_amd/libdynhor_hip_X.so  ->  one line: stage ms of hash_weight_grads and the step"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(ROOT, sys.argv[1])
code = ("import sys; sys.path.insert(0, %r); from dynhor_amd import _lib; _lib.LIB_PATH = %r; import bench; "
        "sys.argv = ['bench.py', '--family', 'hash', '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-secondary']; bench.main()" % (ROOT, lib))
p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
line = [l for l in p.stdout.strip().split("\n") if l.startswith("{")][-1]
d = json.loads(line)
print(os.path.basename(lib), "step ms", d["ms_per_step"], {k: round(v["ms"], 4) for k, v in d["kernels"].items() if "ms" in v})
