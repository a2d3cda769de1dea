#!/usr/bin/env python3
"""tools/summarize_workloads.py <round>: what tools/gpu_r3_part3.sh (r03) / tools/gpu_r4_final_c.sh (r04) left in gpurun_out/
-> profiles/<round>_workloads.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
tag = "r3" if rnd == "r03" else "r" + str(int(rnd[1:]))
out = json.load(open(os.path.join(G, f"wl_{tag}.json")))
host = json.load(open(os.path.join(G, f"wl_{tag}_hostnewton.json")))
for k, v in host.items():
    out[k + "_host_newton_loop"] = v
out["NEWTON_PROBE"] = {"text": [l.rstrip() for l in open(os.path.join(G, f"newton_probe_{tag}.txt")) if l.strip()],
                       "what": "tools/gpu_newton_probe.py: pllhip_newton_branch (one launch per branch) against one blocking "
                               "pll_compute_likelihood_derivatives call per iterate (ctypes)"}
out["_provenance"] = ("tools/gpu_r3_part3.sh" if tag == "r3" else f"tools/gpu_{tag}_final_c.sh") + " on one MI355X box, written by the tools (no hand-typed values)"
for l in open(os.path.join(G, f"manypart_{tag}.jsonl")):
    if not l.strip():
        continue
    j = json.loads(l)
    key = f"MANYPART_{j['states']}states_{j['partitions']}x{j['sites_each']}" + ("_round2_form" if "variant" in j else "")
    out[key] = j
json.dump(out, open(os.path.join(ROOT, "profiles", rnd + "_workloads.json"), "w"), indent=1)
print(len(out), "entries")
