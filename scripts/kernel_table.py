"""Per-kernel roofline table of DESIGN.md section 4 from the committed artefacts (markdown on stdout).

    python scripts/kernel_table.py [profiles/r06_bench_n1.json profiles/r06_pmc_summary.json]

ms (HIP events), TFLOP/s and fraction come from the bench line; ms (rocprofv3) from the --kernel-trace --stats pass; MFMA-busy =
SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); clock = GRBM_GUI_ACTIVE / 8 / rocprofv3 time (a profiled pass runs
slower than the un-profiled bench: guide 'DVFS give-back' item 2); HBM = FETCH_SIZE x 2 + WRITE_SIZE (KB, separate passes);
L2 -> CU = TCP_TCC_READ_REQ_sum x 128 B; hit rate = TCC_HIT / (TCC_HIT + TCC_MISS).
"""
import json, os, re, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bench = sys.argv[1] if len(sys.argv) > 1 else os.path.join(R, "profiles", "r06_bench_n1.json")
pmc = sys.argv[2] if len(sys.argv) > 2 else os.path.join(R, "profiles", "r06_pmc_summary.json")
b = json.loads(open(bench).read().strip().split("\n")[-1])
d = json.load(open(pmc))
ks = {}
for r in d["kernel_stats"]:
    n = re.sub(r"[<(].*", "", re.sub(r"^dh::", "", re.sub(r"^void ", "", r["Name"])))
    ks[n] = float(r["AverageNs"]) * 1e-6
print("| stage | kernel | ms (HIP events) | ms (rocprofv3) | TFLOP/s | frac of ceiling | MFMA-busy | clock GHz | HBM GB / launch (TB/s) | L2 → CU GB (hit rate) |")
print("|---|---|---|---|---|---|---|---|---|---|")
order = ["weight_grads_gemm", "sdf_forward", "sdf_gradient", "color_forward", "color_backward", "sdf_tangent", "sdf_backward", "sdf_nograd_coarse"]
for st in order:
    v = b["kernels"][st]
    k = v["kernel"]
    g = lambda p, c: d[p][k][c]["mean_per_dispatch"]
    if st == "sdf_nograd_coarse":
        # four launches per step of different sizes share the kernel: the PMC means mix them, only the bench columns are per stage
        print("| %s | `%s` | %.2f | — | %.1f | %.2f | — | — | — | — |" % (st, k, v["ms"], v["tflops"], v["frac_of_peak"]))
        continue
    gui = g("prof_pmc1", "GRBM_GUI_ACTIVE")
    busy = g("prof_pmc1", "SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * gui / 8)
    clk = gui / 8 / (ks[k] * 1e-3) * 1e-9
    hbm = (g("prof_pmc2", "FETCH_SIZE") * 2 + g("prof_pmc3", "WRITE_SIZE")) * 1024
    l2 = g("prof_pmc4", "TCP_TCC_READ_REQ_sum") * 128
    hit = g("prof_pmc4", "TCC_HIT_sum") / (g("prof_pmc4", "TCC_HIT_sum") + g("prof_pmc4", "TCC_MISS_sum"))
    print("| %s | `%s` | %.2f | %.2f | %.1f | %.2f | %.2f | %.2f | %.1f (%.1f) | %.1f (%.2f) |" % (
        st, k, v["ms"], ks[k], v["tflops"], v["frac_of_peak"], busy, clk, hbm * 1e-9, hbm * 1e-12 / (v["ms"] * 1e-3), l2 * 1e-9, hit))
