"""CPU: the schema of the bench line the driver records, checked on the verbatim lines committed under profiles/.

`roofline.frac` follows SURVEY.md section 8(d): ALGORITHMIC fp32-product FLOPs of the dominant kernel / its launch time / the ceiling of
its instruction mix (2500 / 3 TFLOP/s for the two-piece fp16 arithmetic), with `peak_basis` naming that ceiling; the HBM view of this
design's save-everything data flow rides beside it as `design_floor` (design bytes, NOT section 8(d)'s algorithmic bytes) and
`traffic_ratio` = counter bytes per step / section 8(d) bytes per step.  `psnr_at_2k` carries the window AND the final-checkpoint
statistic; `lib_sha16` / `build` tie the line to a library build."""
import glob
import json
import math
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r05_bench_n1*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r06_bench_n1*.json")))


def _load(path):
    txt = open(path).read().strip().splitlines()[-1]
    return json.loads(txt)


def test_a_round_5_bench_line_is_committed():
    assert LINES, "profiles/r05_bench_n1*.json: the verbatim line(s) of `python bench.py` on one MI355X"


def test_the_round_6_headline_carries_its_own_parity_measurement_and_names_the_matrix_pipe():
    """Round 6 (VERDICT r5 next #5a, #8): `parity_check` is MEASURED by the run that printed the line (a 50-iteration lock-step segment
    against the GPU-eager oracle, outside the timed region), `psnr_at_2k` says that it is a committed quote, `mfma_dtype` names the pipe
    beside `dtype`, `bound_in_this_design` rides beside section 8(d)'s `bound`, and the occupancy-grid line carries its PSNR deficit."""
    path = os.path.join(ROOT, "profiles", "r06_bench_n1.json")
    if not os.path.exists(path):
        pytest.skip("profiles/r06_bench_n1.json not committed yet")
    d = _load(path)
    assert d["dtype"] == "f32" and d["mfma_dtype"].startswith("f16 (2-piece split, 3 products")
    pc = d["parity_check"]
    assert pc["measured_in_this_run"] is True and pc["iters"] == 50 and pc["pass"] is True
    assert abs(pc["first_step_loss_diff"]) <= 5e-6 and abs(pc["signed_segment_mean"]) <= 1e-4
    assert d["psnr_at_2k"]["measured_in_this_run"] is False and "COMMITTED" in d["psnr_at_2k"]["label"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["bound_in_this_design"] in ("hbm", "mfma") and "bound_basis" in r
    assert (r["bound_in_this_design"] == "hbm") == ("HBM-bound" in r["design_floor"]["why"])
    occ = d["secondary"]["hash_occgrid"]["psnr_vs_hierarchical_db"]
    assert occ["measured_in_this_run"] is False and occ["mean"] < -0.3 and occ["seeds"] >= 8
    assert d["kernels"]["color_forward"]["kernel"] == "color_fwd_p_kernel"


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_bench_line_schema(path):
    d = _load(path)
    # the driver's contract
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["metric"] == "training rays/sec" and d["unit"] == "rays/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["n_gpus"] * d["config"]["rays_per_rank"] / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    r = d["roofline"]
    if d["config"].get("family") != "neus":
        return
    # section 8(d): the MLP GEMMs are priced in FLOPs against the matrix pipe
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-4
    if "split_f16" in path or "2-way fp16" in d["config"]["arithmetic"]:
        assert abs(r["peak"] - 2500.0 / 3.0) < 0.1 and "three-product" in r["peak_basis"]
    assert abs(r["achieved"] - r["algorithmic_flop"] / (r["avg_launch_ms"] * 1e-3) / 1e12) < 0.02 * r["achieved"]
    assert r["algorithmic_flop"] == 2 * 1254656 * d["config"]["rays_per_rank"] * 128 or r["stage"] != "weight_grads_gemm"
    # the design's own HBM floor beside it, under its own name
    f = r["design_floor"]
    assert set(f) >= {"design_bytes", "achieved_gbps", "frac_of_hbm_peak_8000", "frac_of_achievable_6290", "why"}
    assert "algorithmic_bytes" not in r and abs(f["frac_of_achievable_6290"] - f["achieved_gbps"] / 6290.0) < 2e-4
    assert r["traffic"] is None or r["traffic"] >= 0.9 * f["design_bytes"]
    if r["traffic"] is not None:
        assert r["traffic_ratio"] > 100 and "section 8(d)" in r["traffic_ratio_basis"]
    # which library ran
    assert isinstance(d["lib_sha16"], str) and len(d["lib_sha16"]) == 16
    assert d["build"] is None or d["build"]["mode"] in ("compiled", "reused") or "unknown" in d["build"]["mode"]
    if d["build"] and "flags" in d["build"]:
        assert "-packed-fp32-ops" in d["build"]["flags"], "the shipping build switches packed fp32 off (csrc/layout.h)"
    # PSNR: window and final checkpoint side by side
    p = d.get("psnr_at_2k")
    if p is not None and "window" in p:
        for key in ("window", "final_checkpoint"):
            s = p[key]
            assert set(s) >= {"mean", "se", "seeds", "median", "worst_seed"} and math.isfinite(s["mean"]) and s["seeds"] >= 2
        assert p["delta"] == p["window"]["mean"]
    # cpu baseline and secondary lines of the default run
    if "cpu_baseline" in d:
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["unit"] == "rays/s" and c["cores"] >= 1 and "sample" in c
    for name, s in d.get("secondary", {}).items():
        assert "error" not in s, (name, s)
        assert s["value"] > 0 and "roofline" in s


def test_the_committed_psnr_record_feeds_both_statistics():
    import sys
    sys.path.insert(0, ROOT)
    import bench
    p = bench.committed_psnr_record()
    assert p is not None and p["window"]["seeds"] == p["seeds"] and p["final_checkpoint"]["seeds"] == p["seeds"]
    assert p["final_checkpoint"]["worst_seed"] <= p["final_checkpoint"]["median"] <= p["final_checkpoint"]["best_seed"]
