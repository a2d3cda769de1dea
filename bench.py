#!/usr/bin/env python3
"""bench.py -- CLV site-updates/sec of the likelihood hot path on MI355X.

One "step" = one full likelihood evaluation of the synthetic partition
(workload W1 of SURVEY.md section 8d, the call pattern of
treeinfo_compute_loglh, src/tree/treeinfo.c:946-1079): all 2n-3 P-matrices (one
pll_update_prob_matrices call per branch, as treeinfo issues them), the n-2
partial-likelihood operations of a full post-order traversal, one edge
log-likelihood and -- with more than one rank -- the all-reduce of the summed
lnL through the reference's reduce-callback interface (RCCL over xGMI).  The
step loop runs in C (include/pllhip_eval.h); Python only starts and times it.

Default workload: BASELINE.json's target configuration C3 (20 states, Gamma4,
200 taxa, 1M sites; the "LG-shaped" seeded model because the real LG table is
unavailable offline).  `--config c2` selects the DNA configuration.

Multi-GPU: one process per GPU (torch.distributed.run); sites shard across the
ranks.  `--scaling weak` (default) keeps the per-GPU slice at the configured
site count; `--scaling strong` splits the configured site count over the ranks.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pll-modules_amd"))

import numpy as np  # noqa: E402,F401

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix peak (= FP64 vector peak)
FP64_MFMA_SUSTAINED_TFLOPS = 47.5    # measured, tools/micro/mfma_f64.hip (all 256 CUs busy)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c3", choices=["c2", "c3", "c4", "c5"])
    ap.add_argument("--sites", type=int, default=0, help="override the per-configuration site count")
    ap.add_argument("--taxa", type=int, default=0)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--pmatrix-calls", default="per-branch", choices=["per-branch", "batched"],
                    help="per-branch = one pll_update_prob_matrices call per branch, as treeinfo issues "
                         "them (src/tree/treeinfo.c:845-865); batched = one call with count = 2n-3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sites", type=int, default=0, help="sites of the CPU baseline sample")
    return ap.parse_args()


def model_of(pc, states):
    if states == 4:
        return pc.DNA_GTR_RATES, pc.DNA_FREQS, 0.841
    if states == 20:
        r, f = pc.protein_model()
        return r, f, 0.5
    r, f = pc.codon_model()
    return r, f, 0.5


# C4 (BASELINE.json configs[3]): mixed DNA + protein partitions under one tree with
# linked branch lengths: (states, share of the configured site count)
C4_PARTS = [(4, 0.25), (4, 0.25), (20, 0.125), (20, 0.125)]


def partition_plan(config, states, nsites):
    if config != "c4":
        return [(states, nsites)]
    return [(s, max(1, int(round(nsites * f)))) for s, f in C4_PARTS]


def make_evaluation(pc, lib, tree, plan, rate_cats, seed, per_branch):
    """one Evaluation (C driver) over the partitions of `plan` = [(states, sites), ...]"""
    ev = pc.Evaluation(lib, tree.newick(), flags=1 if per_branch else 0, nparts=len(plan))
    insts = []
    for k, (states, nsites) in enumerate(plan):
        subst, freqs, alpha = model_of(pc, states)
        codes = pc.random_codes(tree.ntips, nsites, states, seed + 101 * k)
        insts.append(ev.add_partition(k, states, nsites, rate_cats, codes, subst, freqs, alpha, coded=True))
    return ev, insts


def cpu_baseline(pc, tree, config, states, rate_cats, nsites_gpu, cpu_sites, per_branch):
    """the oracle (a from-scratch CPU port, NOT libpll) timed on the host cores
    on a bounded sample of the same workload: same tree, model, tip generator and
    C driver, fewer sites, sized for ~10-30 s of CPU work"""
    threads = min(os.cpu_count() or 1, 16)      # the 1-GPU box share of host cores
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("ORC_FAST", "1")      # the oracle's AVX2-vectorised partials (4 / 20 / 61 states)
    oracle = pc.PllLib(os.path.join(ROOT, "oracle", "_build", "libpll_oracle.so"))
    ntips = tree.ntips
    if not cpu_sites:
        per_update = (4.0 * states * states + states) / 1.5e9      # ~scalar flop rate of one core
        cpu_sites = int(15.0 * threads / (per_update * rate_cats * (ntips - 2)))
        cpu_sites = max(1000, min(nsites_gpu, cpu_sites // 1000 * 1000))
    plan = partition_plan(config, states, cpu_sites)
    cpu_sites = sum(n for _, n in plan)
    ev, _ = make_evaluation(pc, oracle, tree, plan, rate_cats, 44, per_branch)
    with ev:
        ev.loglh()                                  # warm-up
        t0 = time.perf_counter()
        reps = 0
        while True:
            lnl = ev.loglh()
            reps += 1
            if time.perf_counter() - t0 > 10.0 or reps >= 5:
                break
        dt = time.perf_counter() - t0
    updates = reps * (ntips - 2) * cpu_sites * rate_cats
    return {"value": updates / dt, "unit": "CLV site-updates/s", "cores": threads, "kind": "port",
            "sample": f"{reps} full evaluations of the same tree/model with {cpu_sites} sites "
                      f"(oracle/, plain C + OpenMP over sites, AVX2-vectorised partials, "
                      f"{threads} threads); lnL/site "
                      f"{lnl / cpu_sites:.6f}"}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import pllhip_ctypes as pc
    product = pc.PllLib(pc.PRODUCT_LIB)
    if product.lib.pllhip_device_count() < 1:
        raise SystemExit("bench.py: no HIP device visible; the engine has no CPU fallback")

    dist = None
    comm = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not product.lib.pllhip_set_device(local_rank):
        raise SystemExit(product.errmsg)

    states, rate_cats, ntips, nsites = pc.CONFIGS[args.config]
    if args.sites:
        nsites = args.sites
    if args.taxa:
        ntips = args.taxa
    if args.scaling == "strong" and world > 1:
        local_sites = nsites * (rank + 1) // world - nsites * rank // world
        total_sites = nsites
    else:
        local_sites = nsites
        total_sites = nsites * world
    per_branch = args.pmatrix_calls == "per-branch"

    tree = pc.Tree(ntips, 42, 43)
    # every rank owns a different slice of the alignment (different tip seed)
    plan = partition_plan(args.config, states, local_sites)
    local_sites = sum(n for _, n in plan)
    total_sites = local_sites * world if not (args.scaling == "strong" and world > 1) else total_sites
    ev, insts = make_evaluation(pc, product, tree, plan, rate_cats, 44 + 7919 * rank, per_branch)
    inst = max(insts, key=lambda i: i.N * i.S)        # the partition that dominates the traffic

    if world > 1:
        # the native reduce callback (RCCL in C, behind the reference's parallel_reduce_cb
        # signature) gets its unique id through torch's store; the driver calls it after
        # every edge log-likelihood, exactly where treeinfo does (src/tree/treeinfo.c:1061)
        idbuf = C.create_string_buffer(128)
        if rank == 0 and not product.lib.pllhip_comm_get_unique_id(idbuf):
            raise SystemExit(product.errmsg)
        obj = [idbuf.raw]
        dist.broadcast_object_list(obj, src=0)
        comm = product.lib.pllhip_comm_create(obj[0], rank, world, local_rank)
        if not comm:
            raise SystemExit(product.errmsg)
        cb = C.cast(product.lib.pllhip_reduce_cb, C.c_void_p)
        product.lib.pllhip_eval_set_parallel_context(ev.ev, comm, cb)

    nops = ntips - 2

    def barrier():
        for i in insts:
            product.lib.pllhip_synchronize(i.p)
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        ev.loglh()
    for i in insts:
        product.lib.pllhip_profile_partials(i.p, 1)
    barrier()
    t0 = time.perf_counter()
    lnl = 0.0
    for _ in range(args.steps):
        lnl = ev.loglh()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = pc.Profile()           # summed over the partitions of the evaluation
    for i in insts:
        one = pc.Profile()
        product.lib.pllhip_profile_read(i.p, C.byref(one))
        product.lib.pllhip_profile_partials(i.p, 0)
        for f, _ in pc.Profile._fields_:
            setattr(prof, f, getattr(prof, f) + getattr(one, f))
    counters = inst.counters()

    if dist is not None:
        import torch
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    kernel = product.lib.pllhip_partials_kernel_name(inst.p).decode()
    updates_per_step = nops * total_sites * rate_cats
    value = updates_per_step * args.steps / elapsed

    # roofline of the dominant kernel (pll_update_partials), from the HIP events
    # recorded around its launches during the timed region on rank 0
    achieved = prof.algorithmic_bytes / (prof.kernel_ms * 1e-3) / 1e9 if prof.kernel_ms > 0 else 0.0
    if len(insts) > 1:
        # partitions run on their own streams and overlap on the device: the summed
        # event times exceed the wall time, so the kernels are priced against the
        # whole step instead (a lower bound of what they achieve)
        achieved = prof.algorithmic_bytes / elapsed / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(f"{args.config}:{kernel}")
        except Exception:
            traffic = None
    tflops = prof.algorithmic_flops / (prof.kernel_ms * 1e-3) / 1e12 if prof.kernel_ms > 0 else 0.0
    # the kernel is priced against the roof that bounds it: HBM for 4 and 20 states
    # (0.7 and 3.4 flop/B), the FP64 matrix pipe for 61 states (10.2 flop/B ~ the ridge)
    mfma_bound = states > 32
    roofline = {
        "bound": "mfma" if mfma_bound else "hbm", "kernel": kernel,
        "achieved": round(tflops, 2) if mfma_bound else round(achieved, 1),
        "peak": FP64_MFMA_PEAK_TFLOPS if mfma_bound else HBM_PEAK_GBS,
        "unit": "TFLOP/s" if mfma_bound else "GB/s",
        "frac": round(tflops / FP64_MFMA_PEAK_TFLOPS, 4) if mfma_bound else round(achieved / HBM_PEAK_GBS, 4),
        "traffic": traffic, "algorithmic_GBps": round(achieved, 1), "algorithmic_TFLOPs": round(tflops, 2),
        "launches": int(prof.launches), "ops": int(prof.ops),
        "avg_launch_ms": round(prof.kernel_ms / max(1, prof.launches), 4),
        "algorithmic_bytes_per_launch": round(prof.algorithmic_bytes / max(1, prof.launches)),
        "kernel_share_of_step": round(prof.kernel_ms * 1e-3 / elapsed, 4),
        # tools/micro/mfma_f64.hip: v_mfma_f64_16x16x4_f64 back to back on every SIMD sustains
        # 47.5 TFLOP/s on this chip (29.9 ns per MFMA per SIMD with <= 128 CUs busy, 44.1 ns with
        # all 256: clocks drop under a full-chip FP64 matrix load), not the 78.6 TFLOP/s data-sheet peak
        "mfma_sustained_measured_TFLOPs": FP64_MFMA_SUSTAINED_TFLOPS if mfma_bound else None,
        "frac_of_sustained": round(tflops / FP64_MFMA_SUSTAINED_TFLOPS, 4) if mfma_bound else None,
        # HBM rate implied by the PMC traffic of the committed profile (same command): with
        # operation chains the carried child of a link is not re-read, so the real traffic
        # is BELOW the algorithmic bytes (SURVEY.md 8d counts every child as one read)
        "traffic_GBps": (round(traffic / (prof.kernel_ms / max(1, prof.launches) * 1e-3) / 1e9, 1)
                         if traffic and prof.kernel_ms > 0 and len(insts) == 1 else None),
        "traffic_over_algorithmic": (round(traffic / (prof.algorithmic_bytes / max(1, prof.launches)), 3)
                                     if traffic and prof.algorithmic_bytes > 0 and len(insts) == 1 else None),
        "method": ("algorithmic bytes of all partials launches / step wall time (partitions overlap on "
                   "concurrent streams)") if len(insts) > 1 else
                  "algorithmic bytes of the partials launches / their HIP-event time",
    }

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(pc, tree, args.config, states, rate_cats, local_sites, args.cpu_sites, per_branch)

    ev.close()
    if comm:
        product.lib.pllhip_comm_destroy(comm)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()

    if rank == 0:
        names = {"c2": "C2 DNA GTR+G4", "c4": "C4 mixed: 2 DNA GTR+G4 + 2 protein GTR20+G4 partitions, linked branch lengths",
                 "c3": "C3 protein 'LG-shaped' GTR20+G4 (real LG table unavailable offline)",
                 "c5": "C5 codon GY94-shaped+G4"}
        evals = max(1, args.steps + args.warmup)
        out = {
            "metric": "CLV site-updates/sec (sites x rates x edges)",
            "value": value, "unit": "CLV site-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"{names[args.config]}: {states} states, {rate_cats} rate cats, {ntips} taxa, "
                            f"{total_sites} sites total ({local_sites} per GPU), full traversal "
                            f"({nops} ops + {tree.nedges} P-matrices + edge lnL) per step",
                "config": args.config, "states": states, "rate_cats": rate_cats, "taxa": ntips,
                "sites_total": total_sites, "sites_per_gpu": local_sites, "ops_per_step": nops,
                "partitions": [{"states": s, "sites_per_gpu": n} for s, n in plan],
                "tips": "1-byte codes", "scalers": "per-site, one per inner node",
                "pmatrix_calls": args.pmatrix_calls,
                "pmatrix_launches_per_step": counters.pmatrix_launches // evals,
                "partial_launches_per_step": counters.partial_launches // evals,
            },
            "lnl": lnl, "lnl_per_site": lnl / total_sites,
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))


if __name__ == "__main__":
    main()
