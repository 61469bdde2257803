"""Condense rocprofv3 CSV output (kernel_stats / counter_collection) into small per-kernel summaries."""
import csv, glob, json, os, re, sys, collections

out_dir = sys.argv[1]
summary = {}
for f in glob.glob(os.path.join(out_dir, "prof_stats", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    summary["kernel_stats"] = [r for r in rows if "dh" in r.get("Name", "")]
    summary["kernel_stats_top_other"] = [r for r in rows if "dh" not in r.get("Name", "")][:8]
for d in sorted(glob.glob(os.path.join(out_dir, "prof_pmc*"))):
    if not os.path.isdir(d):
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if "dh" not in name:
                continue
            short = re.sub(r"<.*", "", name.split("(")[0].replace("dh::", "").replace("void ", ""))     # template instances -> kernel name
            a = agg[short][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    summary[os.path.basename(d)] = {k: {c: {"mean_per_dispatch": v[0] / v[1], "dispatches": v[1]} for c, v in cs.items()}
                                    for k, cs in agg.items()}
json.dump(summary, open(os.path.join(out_dir, "prof_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1)[:6000])
