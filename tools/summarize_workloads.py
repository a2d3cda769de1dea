#!/usr/bin/env python3
"""tools/summarize_workloads.py <round>: what tools/gpu_r3_part3.sh left in gpurun_out/ -> profiles/<round>_workloads.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
out = json.load(open(os.path.join(G, "wl_r3.json")))
host = json.load(open(os.path.join(G, "wl_r3_hostnewton.json")))
for k, v in host.items():
    out[k + "_host_newton_loop"] = v
out["NEWTON_PROBE"] = {"text": [l.rstrip() for l in open(os.path.join(G, "newton_probe_r3.txt")) if l.strip()],
                       "what": "tools/gpu_newton_probe.py: pllhip_newton_branch (one launch per branch) against one blocking "
                               "pll_compute_likelihood_derivatives call per iterate (ctypes)"}
out["_provenance"] = "tools/gpu_r3_part3.sh on one MI355X box, written by the tools (no hand-typed values)"
for l in open(os.path.join(G, "manypart_r3.jsonl")):
    if not l.strip():
        continue
    j = json.loads(l)
    key = f"MANYPART_{j['states']}states_{j['partitions']}x{j['sites_each']}" + ("_round2_form" if "variant" in j else "")
    out[key] = j
json.dump(out, open(os.path.join(ROOT, "profiles", rnd + "_workloads.json"), "w"), indent=1)
print(len(out), "entries")
