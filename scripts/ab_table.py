"""Development tool: table of gpurun_out/ab_<lib>_r<round>.json medians (scripts/ab_libs.sh)."""
import glob, json, os, re, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = sys.argv[1:]
stages = ["sdf_nograd_coarse", "sdf_nograd_fine", "sdf_forward", "sdf_gradient", "color_forward", "color_backward", "sdf_tangent", "sdf_backward", "weight_grads_gemm"]
print("%-8s" % "lib" + "".join("%10s" % s.replace("weight_grads_gemm", "dW").replace("color_", "col_").replace("sdf_nograd_", "ng_").replace("sdf_", "") for s in stages) + "%10s" % "sum")
for n in names:
    rows = []
    for f in sorted(glob.glob(os.path.join(R, "gpurun_out", "ab_%s_r*.json" % n))):
        d = json.load(open(f))["stages"]
        rows.append([d[s]["median_ms"] for s in stages])
    if not rows:
        continue
    avg = [sum(r[i] for r in rows) / len(rows) for i in range(len(stages))]
    print("%-8s" % n + "".join("%10.3f" % v for v in avg) + "%10.3f" % sum(avg))
