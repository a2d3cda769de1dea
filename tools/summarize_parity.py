#!/usr/bin/env python3
"""gpurun_out/parity_measured.jsonl (written by the -m gpu tests) -> profiles/<round>_parity.json:
the measured deviations behind the parity claims, worst case per test kind."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
    rows = [json.loads(l) for l in open(os.path.join(ROOT, "gpurun_out", "parity_measured.jsonl"))]
    out = {"source": "pytest -m gpu on one MI355X; every engine with its own eigen-solver", "fixtures": {},
           "hip_vs_oracle_61_states": {}, "baseline_tiles": {}}
    for r in rows:
        if r["test"] == "expm_fixture":
            key = f'{r["engine"]}:{r["case"]}'
            cur = out["fixtures"].setdefault(key, {})
            for k in ("dP", "dCLV_rel_site_max", "max_persite_dlnl", "dlnl_per_site"):
                cur[k] = max(cur.get(k, 0.0), r[k])
        elif r["test"] == "full_traversal_61":
            cur = out["hip_vs_oracle_61_states"]
            cur["cases"] = cur.get("cases", 0) + 1
            for k in ("dlnl_per_site", "clv_site_err"):
                cur[k] = max(cur.get(k, 0.0), r[k])
        elif r["test"] == "baseline_tile":
            out["baseline_tiles"][f'c{int(r["config"])}'] = {"tile_sites": int(r["tile_sites"]),
                                                             "dlnl_per_site": r["dlnl_per_site"],
                                                             "lnl_per_site": r["lnl_per_site"]}
    path = os.path.join(ROOT, "profiles", f"{rnd}_parity.json")
    json.dump(out, open(path, "w"), indent=1)
    print(open(path).read())


if __name__ == "__main__":
    main()
