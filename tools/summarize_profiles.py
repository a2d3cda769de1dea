#!/usr/bin/env python3
"""Condense the raw rocprofv3 output that tools/gpu_profile.sh leaves in
gpurun_out/ into small tracked files under profiles/:

  profiles/rNN_<cfg>_kernel_stats.csv   the --kernel-trace --stats summary, as is
  profiles/rNN_<cfg>_pmc.json           FETCH_SIZE / WRITE_SIZE per kernel
  profiles/traffic.json                 per-launch HBM bytes of the dominant kernel,
                                        read by bench.py for `roofline.traffic`

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md section HBM: the counters
are in KiB; WRITE_SIZE is exact for 16-B-per-lane streaming stores; FETCH_SIZE
reports exactly half of the bytes of a wide coalesced streaming read on gfx950,
so it is doubled.  Each counter comes from its own --pmc pass.
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles")


def newest(pattern):
    """gpurun merges every run into gpurun_out/: keep the most recent file only"""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:]


def pmc_per_kernel(directory, counter):
    files = newest(os.path.join(directory, "*", "*_counter_collection.csv"))
    acc = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"].split("(")[0]
            a = acc.setdefault(name, [0, 0.0])
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    return acc


def main():
    rnd, cfg = sys.argv[1], sys.argv[2]
    kernel_key = sys.argv[3] if len(sys.argv) > 3 else None
    os.makedirs(OUT, exist_ok=True)
    g = os.path.join(ROOT, "gpurun_out")
    stats = newest(os.path.join(g, f"prof_{cfg}", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(OUT, f"{rnd}_{cfg}_kernel_stats.csv"))
    bench = os.path.join(g, f"bench_{cfg}.json")
    if os.path.exists(bench):
        shutil.copy(bench, os.path.join(OUT, f"{rnd}_{cfg}_bench.json"))
    fetch = pmc_per_kernel(os.path.join(g, f"pmc_fetch_{cfg}"), "FETCH_SIZE")
    write = pmc_per_kernel(os.path.join(g, f"pmc_write_{cfg}"), "WRITE_SIZE")
    summary = {}
    for name in sorted(set(fetch) | set(write)):
        nf, vf = fetch.get(name, [0, 0.0])
        nw, vw = write.get(name, [0, 0.0])
        n = max(nf, nw)
        if not n:
            continue
        rd = 2.0 * vf * 1024.0 / max(1, nf)      # gfx950: FETCH_SIZE counts half
        wr = vw * 1024.0 / max(1, nw)
        summary[name] = {"dispatches": n, "fetch_size_kib_per_dispatch_raw": vf / max(1, nf),
                         "write_size_kib_per_dispatch_raw": vw / max(1, nw),
                         "hbm_read_bytes_per_dispatch": rd, "hbm_write_bytes_per_dispatch": wr,
                         "hbm_bytes_per_dispatch": rd + wr}
    json.dump(summary, open(os.path.join(OUT, f"{rnd}_{cfg}_pmc.json"), "w"), indent=1)
    if kernel_key:
        # all partials launches of the run -- whole traversals (k_traverse_*), rounds of chains
        # (k_chain_*) and per-operation launches (k_partials_*, plus the 61-state fix-up and
        # cherry-table kernels that belong to them) -- averaged per launch, the unit of
        # bench.py's roofline.algorithmic_bytes_per_launch
        dom = [v for k, v in summary.items() if "k_partials" in k or "k_chain" in k or "k_traverse" in k]
        extra = [v for k, v in summary.items() if "k_s61_scale_fixup" in k or "k_s61_cherry_scale" in k]
        if dom:
            total = sum(v["dispatches"] * v["hbm_bytes_per_dispatch"] for v in dom + extra)
            launches = sum(v["dispatches"] for v in dom)
            # one edge-lnL kernel per evaluation = the number of steps of the PMC run
            steps = sum(v["dispatches"] for k, v in summary.items() if "k_edge_lnl" in k) or 1
            tpath = os.path.join(OUT, "traffic.json")
            t = json.load(open(tpath)) if os.path.exists(tpath) else {}
            t[f"{cfg}:{kernel_key}"] = round(total / launches)
            # per step (traversal): what bench.py divides by its own launch count, so that a change of the
            # schedule (launches per traversal) does not invalidate the committed figure
            t[f"{cfg}:{kernel_key}:per_step"] = round(total / steps)
            json.dump(t, open(tpath, "w"), indent=1)
    for k, v in summary.items():
        print(f"{k:40s} n={v['dispatches']:4d} read={v['hbm_read_bytes_per_dispatch']/1e6:10.1f} MB "
              f"write={v['hbm_write_bytes_per_dispatch']/1e6:10.1f} MB")


if __name__ == "__main__":
    main()
