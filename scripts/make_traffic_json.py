"""profiles/pmc_traffic.json from a rocprofv3 PMC summary (scripts/prof_summarize.py output).
HBM bytes per launch = FETCH_SIZE[KB]*1024*2 + WRITE_SIZE[KB]*1024: on gfx950 FETCH_SIZE reports exactly half the bytes
of 16-B-per-lane coalesced streaming reads (MI355X_MICROARCH.md section HBM) -- every large read in these kernels is of that
form (native tiles, float4 per lane); WRITE_SIZE is exact for 16-B-per-lane streaming stores.  FETCH_SIZE and WRITE_SIZE
come from separate --pmc passes (TCC slots).

The stage -> kernel map is dynhor_amd/_lib.py:STAGE_KERNELS (the table bench.py reads): a kernel of the current shipping set
that is missing from the profile is an ERROR (the profile is stale), so a renamed kernel cannot leave old numbers behind."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynhor_amd import _lib
src, dst = sys.argv[1], sys.argv[2]
mode = _lib.ARITH_NAMES.get(sys.argv[3], _lib.ARITH_DEFAULT) if len(sys.argv) > 3 else _lib.ARITH_DEFAULT
p = json.load(open(src))
out = {}
missing = []
if len(sys.argv) > 3 and sys.argv[3] == "hash":
    # hash family (BASELINE.json configs[3]): one timed stage = several kernels (dynhor_amd/_lib.py:HASH_STAGE_LAUNCHES); the
    # table scatter's WRITE_SIZE is the bytes its atomics send to memory (64-byte requests; exact for one dword per lane, guide section HBM)
    for stage, kerns in _lib.HASH_STAGE_LAUNCHES.items():
        tot, parts = 0.0, {}
        for kern in kerns:
            if kern not in p.get("prof_pmc2", {}) or kern not in p.get("prof_pmc3", {}):
                missing.append(kern)
                continue
            f = p["prof_pmc2"][kern]["FETCH_SIZE"]["mean_per_dispatch"]
            w = p["prof_pmc3"][kern]["WRITE_SIZE"]["mean_per_dispatch"]
            parts[kern] = {"fetch_size_kb_raw": f, "write_size_kb_raw": w, "hbm_bytes_per_launch": f * 1024 * 2 + w * 1024}
            tot += f * 1024 * 2 + w * 1024
        out[stage] = {"kernel": _lib.HASH_STAGE_KERNELS[stage], "kernels": parts, "hbm_bytes_per_launch": tot,
                      "atomic_bytes_per_launch": (parts.get(_lib.HASH_STAGE_KERNELS[stage], {}).get("write_size_kb_raw", 0.0) * 1024
                                                  if stage == "hash_weight_grads" else None),
                      "correction": "FETCH_SIZE x2 (gfx950 16-B/lane streaming reads), WRITE_SIZE x1 (exact for float atomics)"}
    if missing:
        sys.exit(f"stale or incomplete PMC profile: no counters for the shipping kernels {missing}")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))
    sys.exit(0)
for stage, kern in _lib.STAGE_KERNELS[mode].items():
    if stage == "sdf_nograd_fine":
        continue          # same kernel as sdf_nograd_coarse; the PMC means are over all its launches
    if kern not in p.get("prof_pmc2", {}) or kern not in p.get("prof_pmc3", {}):
        missing.append(kern)
        continue
    f = p["prof_pmc2"][kern]["FETCH_SIZE"]["mean_per_dispatch"]
    w = p["prof_pmc3"][kern]["WRITE_SIZE"]["mean_per_dispatch"]
    out[stage] = {"kernel": kern, "hbm_bytes_per_launch": f * 1024 * 2 + w * 1024, "fetch_size_kb_raw": f, "write_size_kb_raw": w,
                  "correction": "FETCH_SIZE x2 (gfx950 16-B/lane streaming reads), WRITE_SIZE x1"}
if missing:
    sys.exit(f"stale or incomplete PMC profile: no counters for the shipping kernels {missing}")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
