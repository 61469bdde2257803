"""profiles/pmc_traffic.json from a rocprofv3 PMC summary (scripts/prof_summarize.py output).
HBM bytes per launch = FETCH_SIZE[KB]*1024*2 + WRITE_SIZE[KB]*1024: on gfx950 FETCH_SIZE reports exactly half the bytes
of 16-B-per-lane coalesced streaming reads (MI355X_MICROARCH.md §HBM) -- every large read in these kernels is of that
form (native tiles, float4 per lane); WRITE_SIZE is exact for 16-B-per-lane streaming stores.  FETCH_SIZE and WRITE_SIZE
come from separate --pmc passes (TCC slots)."""
import json, sys
src, dst = sys.argv[1], sys.argv[2]
p = json.load(open(src))
stage_of = {"dw_lds_kernel": "weight_grads_gemm", "dw_bf16x3_kernel": "weight_grads_gemm",
            "sdf_fwd_train_kernel": "sdf_forward", "sdf_fwd_train16_kernel": "sdf_forward",
            "sdf_grad_kernel": "sdf_gradient", "sdf_grad16_kernel": "sdf_gradient",
            "color_fwd_kernel": "color_forward", "color_fwd16_kernel": "color_forward",
            "color_bwd_kernel": "color_backward", "color_bwd16_kernel": "color_backward",
            "sdf_tangent_kernel": "sdf_tangent", "sdf_tangent16_kernel": "sdf_tangent",
            "sdf_bwd_kernel": "sdf_backward", "sdf_bwd16_kernel": "sdf_backward",
            "sdf_fwd_train_s_kernel": "sdf_forward", "sdf_grad_s_kernel": "sdf_gradient", "color_fwd_s_kernel": "color_forward",
            "color_bwd_s_kernel": "color_backward", "sdf_tangent_s_kernel": "sdf_tangent", "sdf_bwd_s_kernel": "sdf_backward"}
out = {}
for kern, stage in stage_of.items():
    if kern not in p["prof_pmc2"] or kern not in p["prof_pmc3"]:
        continue          # the variant that did not run in this profile
    f = p["prof_pmc2"][kern]["FETCH_SIZE"]["mean_per_dispatch"]
    w = p["prof_pmc3"][kern]["WRITE_SIZE"]["mean_per_dispatch"]
    out[stage] = {"kernel": kern, "hbm_bytes_per_launch": f * 1024 * 2 + w * 1024, "fetch_size_kb_raw": f, "write_size_kb_raw": w,
                  "correction": "FETCH_SIZE x2 (gfx950 16-B/lane streaming reads), WRITE_SIZE x1"}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
