#!/usr/bin/env python3
"""Condense the SQ counter pass of tools/gpu_pmc_sq.sh (gpurun_out/pmc_sq_<cfg>/) into
profiles/<round>_<cfg>_sq.json: per kernel, summed over its dispatches,

  mfma_util        SQ_VALU_MFMA_BUSY_CYCLES / (kernel time x 1024 SIMDs x 2.4 GHz): share of the
                   matrix-pipe cycles of the whole chip at the peak clock (the chip clocks lower
                   under a full FP64 matrix load, so 1.0 is not reachable: DESIGN.md section 3)
  lds_busy         SQ_LDS_IDX_ACTIVE / (kernel time x 256 CUs x 2.4 GHz): share of the LDS-array cycles
  lds_conflict     SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE   (conflict cycles per active LDS cycle)
  wait_any         SQ_WAIT_ANY / SQ_WAVE_CYCLES               (share of wave-cycles spent waiting)
  wait_inst_any    SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  active_inst_any  SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES

usage: tools/summarize_sq.py <round> <cfg>
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    rnd, cfg = sys.argv[1], sys.argv[2]
    files = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_sq_{cfg}", "*", "*_counter_collection.csv")),
                   key=os.path.getmtime)
    if not files:
        raise SystemExit("no counter collection found: run tools/gpu_pmc_sq.sh on the GPU box first")
    acc = {}
    for row in csv.DictReader(open(files[-1])):
        name = row["Kernel_Name"].split("(")[0]
        if name.startswith("__amd_rocclr"):
            continue
        k = acc.setdefault(name, {"dispatches": set(), "ns": 0.0})
        if row["Dispatch_Id"] not in k["dispatches"]:
            k["ns"] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
        k["dispatches"].add(row["Dispatch_Id"])
        k[row["Counter_Name"]] = k.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    out = {}
    for name, k in sorted(acc.items()):
        def ratio(a, b):
            return round(k[a] / k[b], 4) if k.get(b) else None
        out[name] = {
            "dispatches": len(k["dispatches"]),
            "avg_us": round(k["ns"] / len(k["dispatches"]) / 1e3, 1),
            "mfma_util": round(k.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (k["ns"] * 2.4 * 1024), 4) if k["ns"] else None,
            "lds_busy": round(k.get("SQ_LDS_IDX_ACTIVE", 0.0) / (k["ns"] * 2.4 * 256), 4) if k["ns"] else None,
            "lds_conflict": ratio("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"),
            "wait_any": ratio("SQ_WAIT_ANY", "SQ_WAVE_CYCLES"),
            "wait_inst_any": ratio("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"),
            "active_inst_any": ratio("SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES"),
            "raw": {c: v for c, v in k.items() if c not in ("dispatches", "ns")},
        }
    path = os.path.join(ROOT, "profiles", f"{rnd}_{cfg}_sq.json")
    json.dump({"command": f"rocprofv3 --pmc <8 SQ counters> -- python3 bench.py --config {cfg} --steps 1 --warmup 1 "
                          "--no-cpu-baseline (tools/gpu_pmc_sq.sh)", "kernels": out}, open(path, "w"), indent=1)
    for name, v in out.items():
        print(f"{name[:44]:44s} n={v['dispatches']:3d} {v['avg_us']:8.1f} us  mfma_util={v['mfma_util']} lds_busy={v['lds_busy']} "
              f"lds_conflict={v['lds_conflict']} wait_any={v['wait_any']} active={v['active_inst_any']}")


if __name__ == "__main__":
    main()
