#!/usr/bin/env python3
"""tools/summarize_slices.py <round>: gpurun_out/slices.jsonl (written by tools/gpu_slices.sh) -> profiles/<round>_slices.json,
adding to every row its rate over the full-size rate of its configuration"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
rows = [json.loads(l) for l in open(os.path.join(ROOT, "gpurun_out", "slices.jsonl")) if l.strip()]
full = {r["config"]: r["site_updates_per_s"] for r in rows if r["share_of_full"] == "1/1"}
for r in rows:
    r["rate_over_full_size_rate"] = round(r["site_updates_per_s"] / full[r["config"]], 4)
out = {"what": "strong-scaling proxy on ONE GPU: the per-GPU slice of a 1/2/4/8-way split of every configuration "
               "(tools/gpu_r3_part2.sh / tools/gpu_r4_final_c.sh -> tools/gpu_slices.sh: bench.py --config <cfg> --sites <full/N> --steps 20 --warmup 3 "
               "--no-cpu-baseline), written by the tool, one box, one run", "rows": rows}
json.dump(out, open(os.path.join(ROOT, "profiles", rnd + "_slices.json"), "w"), indent=1)
for r in rows:
    print(r["config"], r["share_of_full"], r["sites"], round(r["ms_per_step"], 3), r["rate_over_full_size_rate"])
