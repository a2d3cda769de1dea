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

Multi-GPU: one process per GPU; the sites of ONE alignment shard across the
ranks (rank r owns a contiguous site range, so the summed lnL is the same number
at every N).  `--scaling strong` (default; the north star's "1 M-site partition
... >= 6x at 8 GPUs") splits the configured site count over the ranks;
`--scaling weak` keeps the configured site count per GPU.  `python bench.py
--gpus N` with no rank environment starts the N ranks itself (a
torch.distributed.run child, created before anything in this process touches
the GPU) and forwards the child's JSON line and exit code; under a launcher
(RANK / WORLD_SIZE set) it is one of the ranks.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
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
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"])
    ap.add_argument("--topology", default="ranks", choices=["ranks", "internal"],
                    help="ranks: one process per GPU, lnL all-reduced over RCCL (the contract's launch); "
                         "internal: ONE process, every partition spread over --gpus devices inside the engine "
                         "(pllhip_set_sharding: what an unmodified single-treeinfo client gets)")
    ap.add_argument("--pmatrix-calls", default="per-branch", choices=["per-branch", "batched"],
                    help="per-branch = one pll_update_prob_matrices call per branch, as treeinfo issues "
                         "them (src/tree/treeinfo.c:845-865); batched = one call with count = 2n-3")
    ap.add_argument("--rate-scalers", action="store_true",
                    help="PLL_ATTRIB_RATE_SCALERS: one scaling count per (site, rate) instead of per site")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "torch"],
                    help="N > 1: rccl = the library's own RCCL communicator, lnL reduced on the device (default); "
                         "torch = the reduce callback through torch.distributed (a rehearsal path: with "
                         "PLLHIP_BENCH_DIST_BACKEND=gloo and PLLHIP_ALLOW_DEVICE_WRAP=1 all ranks can share one GPU)")
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


ATTRIBUTES = 0      # extra pll_partition_create attributes of every partition (--rate-scalers)


def make_evaluation(pc, lib, tree, plan, rate_cats, seed, per_branch, first_sites=None):
    """one Evaluation (C driver) over the partitions of `plan` = [(states, sites), ...];
    partition k holds sites first_sites[k] .. of the alignment that seed + 101 k defines"""
    ev = pc.Evaluation(lib, tree.newick(), flags=1 if per_branch else 0, nparts=len(plan))
    insts = []
    for k, (states, nsites) in enumerate(plan):
        subst, freqs, alpha = model_of(pc, states)
        codes = pc.random_codes(tree.ntips, nsites, states, seed + 101 * k,
                                first_site=first_sites[k] if first_sites else 0)
        insts.append(ev.add_partition(k, states, nsites, rate_cats, codes, subst, freqs, alpha, coded=True,
                                      attributes=ATTRIBUTES))
    return ev, insts


def evaluate_with_persite(ev, plan):
    """lnL and the per-site lnL of every partition, concatenated"""
    lnl = ev.loglh()
    persite = np.concatenate([ev.persite_lnl(k)[1] for k in range(len(plan))])
    return lnl, persite


def cpu_baseline(pc, product, tree, config, states, rate_cats, nsites_gpu, cpu_sites, per_branch):
    """the oracle (a from-scratch CPU port, NOT libpll) timed on the host cores
    on a bounded sample of the same workload: same tree, model, C driver and the
    FIRST cpu_sites sites of the same alignment, sized for ~10-30 s of CPU work.
    The same sample is then evaluated on the GPU: |dlnL| per site on identical
    inputs (BASELINE.json's metric, second half)."""
    host_cores = os.cpu_count() or 1
    threads = min(host_cores, 16)      # the 1-GPU box share of host cores
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
        lnl_cpu, persite_cpu = evaluate_with_persite(ev, plan)
    gev, _ = make_evaluation(pc, product, tree, plan, rate_cats, 44, per_branch)
    with gev:
        lnl_gpu, persite_gpu = evaluate_with_persite(gev, plan)
    updates = reps * (ntips - 2) * cpu_sites * rate_cats
    parity = {
        "dlnl_per_site": abs(lnl_gpu - lnl_cpu) / cpu_sites,
        "max_persite_dlnl": float(np.max(np.abs(persite_gpu - persite_cpu))),
        "sample_sites": cpu_sites, "lnl_gpu": lnl_gpu, "lnl_cpu": lnl_cpu,
        "against": "oracle/ (CPU restatement pinned by the reference's golden files; libpll itself is "
                   "not available offline) on the first sample_sites sites of the benchmark alignment",
        "tolerance_per_site": 1e-6,
    }
    base = {"value": updates / dt, "unit": "CLV site-updates/s", "cores": threads, "kind": "port",
            "host_cores": host_cores, "threads": threads, "sample_sites": cpu_sites, "sample_evaluations": reps,
            "lnl_per_site": lnl / cpu_sites,
            "sample": f"{reps} full evaluations of the same tree/model on the first {cpu_sites} sites of the "
                      f"benchmark alignment (oracle/, plain C + OpenMP over sites, AVX2-vectorised partials, "
                      f"{threads} threads on a host with {host_cores} cores)"}
    return base, parity


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args):
    """`bench.py --gpus N` without a rank environment: start the N ranks as ONE child
    process (torch.distributed.run -> one process per GPU).  Nothing in this process has
    touched HIP or torch.cuda at this point, and nothing will: the parent only forwards
    the child's JSON line and exit code."""
    probe = os.environ.get("PLLHIP_BENCH_LAUNCH_PROBE") == "1"
    if not probe:
        import torch
        have = torch.cuda.device_count()          # counting devices does not initialise the GPU
        if have < args.gpus and os.environ.get("PLLHIP_ALLOW_DEVICE_WRAP") != "1":
            raise SystemExit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        raise SystemExit(proc.returncode or 1)
    if json.loads(line).get("n_gpus") != args.gpus:
        raise SystemExit(f"bench.py: the ranks report n_gpus != {args.gpus}")
    print(line)
    raise SystemExit(0)


def launch_probe(args, rank, world):
    """PLLHIP_BENCH_LAUNCH_PROBE=1: the ranks only prove that they exist (gloo, no GPU);
    tests/test_bench_launcher.py runs this where there is no GPU"""
    import torch.distributed as dist
    dist.init_process_group("gloo")
    seen = [None] * world
    dist.all_gather_object(seen, (rank, int(os.environ.get("LOCAL_RANK", "-1")), os.getpid()))
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "launch-probe", "n_gpus": world, "ranks": sorted(r for r, _, _ in seen),
                          "local_ranks": sorted(l for _, l, _ in seen),
                          "distinct_processes": len({p for _, _, p in seen})}))


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    in_rank_env = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    internal = args.topology == "internal" and args.gpus > 1
    if internal:
        import torch
        if torch.cuda.device_count() < args.gpus and os.environ.get("PLLHIP_ALLOW_DEVICE_WRAP") != "1":
            raise SystemExit(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) visible")
    elif args.gpus > 1 and not in_rank_env:
        launch_ranks(args)                      # never returns
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if internal:
        rank, world, local_rank = 0, 1, 0
    elif world != args.gpus:
        # never report a run of `world` ranks as a run of --gpus ranks
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("PLLHIP_BENCH_LAUNCH_PROBE") == "1":
        return launch_probe(args, rank, world)

    # ONE line on stdout: libraries that talk on file descriptor 1 (RCCL prints a banner when a
    # communicator is created) are pointed at stderr; the JSON line goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    if world > 1:
        # torch brings its own HIP and RCCL runtimes (same SONAMEs as /opt/rocm's): whichever is
        # loaded first serves the whole process, and torch does not find the GPU on the system's.
        # Load torch first, so that the engine binds to the runtime torch.cuda uses.
        import torch  # noqa: F401
    import pllhip_ctypes as pc
    global ATTRIBUTES
    if args.rate_scalers:
        ATTRIBUTES = pc.PLL_ATTRIB_RATE_SCALERS
    product = pc.PllLib(pc.PRODUCT_LIB)
    if product.lib.pllhip_device_count() < 1:
        raise SystemExit("bench.py: no HIP device visible; the engine has no CPU fallback")

    dist = None
    comm = None
    if world > 1:
        import torch
        import torch.distributed as dist
        backend = os.environ.get("PLLHIP_BENCH_DIST_BACKEND", "nccl")
        if os.environ.get("PLLHIP_ALLOW_DEVICE_WRAP") == "1":
            local_rank %= max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if not product.lib.pllhip_set_device(local_rank):
        raise SystemExit(product.errmsg)
    if internal and not product.lib.pllhip_set_sharding(args.gpus, None):
        raise SystemExit(product.errmsg)

    states, rate_cats, ntips, nsites = pc.CONFIGS[args.config]
    if args.sites:
        nsites = args.sites
    if args.taxa:
        ntips = args.taxa
    per_branch = args.pmatrix_calls == "per-branch"
    tree = pc.Tree(ntips, 42, 43)

    # ONE alignment for the whole job; rank r owns a contiguous site range of every partition
    # (strong: the configured site count is split; weak: it is the per-GPU share)
    job_plan = partition_plan(args.config, states, nsites * (world if args.scaling == "weak" else 1))
    plan, first_sites = [], []
    for s_, n_ in job_plan:
        lo, hi = n_ * rank // world, n_ * (rank + 1) // world
        plan.append((s_, hi - lo))
        first_sites.append(lo)
    local_sites = sum(n for _, n in plan)
    total_sites = sum(n for _, n in job_plan)
    if min(n for _, n in plan) < 1:
        raise SystemExit("bench.py: fewer sites than ranks")
    ev, insts = make_evaluation(pc, product, tree, plan, rate_cats, 44, per_branch, first_sites)
    inst = max(insts, key=lambda i: i.N * i.S)        # the partition that dominates the traffic

    reduce_cb = None
    if world > 1 and args.comm == "rccl":
        # one RCCL communicator per rank (C, behind the reference's parallel_reduce_cb
        # semantics); its unique id travels through torch's store.  The driver leaves every
        # partition's lnL in a device-resident slot and all-reduces the slots in place
        # (src/tree/treeinfo.c:1061 is where the reference reduces).
        idbuf = C.create_string_buffer(128)
        if rank == 0 and not product.lib.pllhip_comm_get_unique_id(idbuf):
            raise SystemExit(product.errmsg)
        obj = [idbuf.raw]
        dist.broadcast_object_list(obj, src=0)
        comm = product.lib.pllhip_comm_create(obj[0], rank, world, local_rank)
        ok = bool(comm) and bool(product.lib.pllhip_eval_attach_comm(ev.ev, comm))
        # every rank must take the same path: agree on the outcome before going on
        import torch
        flag = torch.tensor([1.0 if ok else 0.0], device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if flag.item() < 1.0:
            print(f"bench.py: rank {rank}: the library's communicator is not available ({product.errmsg}); "
                  "falling back to the reduce hook over torch.distributed", file=sys.stderr)
            product.lib.pllhip_eval_set_fused(ev.ev, None)      # back to blocking calls + the reduce hook
            if comm:
                product.lib.pllhip_comm_destroy(comm)
            comm = None
            args.comm = "torch"
    if world > 1 and args.comm == "torch":
        # rehearsal path: the reference's reduce hook served by torch.distributed (any backend)
        import torch
        ops = {0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MAX, 2: dist.ReduceOp.MIN}    # include/pllhip.h PLLHIP_REDUCE_*
        on_gpu = dist.get_backend() == "nccl"

        def _reduce(ctx, data, n, op):
            arr = np.ctypeslib.as_array(data, shape=(n,))
            t = torch.from_numpy(arr.copy())
            if on_gpu:
                t = t.cuda()
            dist.all_reduce(t, op=ops[op])
            arr[:] = t.cpu().numpy()

        reduce_cb = pc.REDUCE_CB(_reduce)
        ev.set_parallel_context(reduce_cb)

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
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
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
    traffic_source = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and world == 1 and not args.sites and not args.taxa and not args.rate_scalers:
        try:
            tj = json.load(open(tpath))
            per_step = tj.get(f"{args.config}:{kernel}:per_step")
            if per_step and prof.launches:
                traffic = round(per_step * args.steps / prof.launches)      # per launch, like `achieved`
            else:
                traffic = tj.get(f"{args.config}:{kernel}")
            traffic_source = ("profiles/traffic.json: committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                              "this command (tools/gpu_profile.sh), NOT measured in this run")
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
        "traffic": traffic, "traffic_source": traffic_source,
        "algorithmic_GBps": round(achieved, 1), "algorithmic_TFLOPs": round(tflops, 2),
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
                  "algorithmic bytes of the partials launches / their HIP-event time (rank 0)",
    }

    ev.close()                   # frees the CLVs: the parity sample below gets its own partitions
    if comm:
        product.lib.pllhip_comm_destroy(comm)

    cpu = parity = None
    if internal:
        product.lib.pllhip_set_sharding(0, None)
    if rank == 0 and world == 1 and not internal and not args.no_cpu_baseline:
        cpu, parity = cpu_baseline(pc, product, tree, args.config, states, rate_cats, local_sites,
                                   args.cpu_sites, per_branch)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()

    if rank == 0:
        names = {"c2": "C2 DNA GTR+G4", "c4": "C4 mixed: 2 DNA GTR+G4 + 2 protein GTR20+G4 partitions, linked branch lengths",
                 "c3": "C3 protein 'LG-shaped' GTR20+G4 (real LG table unavailable offline)",
                 "c5": "C5 codon GY94-shaped+G4"}
        evals = max(1, args.steps + args.warmup)
        out = {
            "metric": "CLV site-updates/sec (sites x rates x edges); |dlnL| vs ref",
            "value": value, "unit": "CLV site-updates/s",
            "n_gpus": args.gpus if internal else world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"{names[args.config]}: {states} states, {rate_cats} rate cats, {ntips} taxa, "
                            f"{total_sites} sites total ({local_sites} on rank 0), full traversal "
                            f"({nops} ops + {tree.nedges} P-matrices + edge lnL"
                            f"{(' + RCCL all-reduce of the lnL on the device' if args.comm == 'rccl' else ' + lnL summed through the reduce hook (torch.distributed)') if world > 1 else ''}) per step",
                "config": args.config, "states": states, "rate_cats": rate_cats, "taxa": ntips,
                "sites_total": total_sites, "sites_per_gpu": local_sites, "ops_per_step": nops,
                "partitions": [{"states": s, "sites_per_gpu": n} for s, n in plan],
                "tips": "1-byte codes",
                "scalers": ("per (site, rate)" if args.rate_scalers else "per-site") + ", one buffer per inner node",
                "pmatrix_calls": args.pmatrix_calls,
                "pmatrix_launches_per_step": counters.pmatrix_launches // evals,
                "partial_launches_per_step": counters.partial_launches // evals,
                "parallelism": (f"one process, every partition spread over {args.gpus} devices inside the engine, "
                                f"lnL summed on the host" if internal else
                                f"sites sharded over {world} GPU(s), lnL all-reduced" if world > 1 else "1 GPU"),
            },
            # the same alignment at every N: the summed lnL must not depend on n_gpus
            "lnl": lnl, "lnl_per_site": lnl / total_sites,
            "dlnl_per_site": parity["dlnl_per_site"] if parity else None,
            "max_persite_dlnl": parity["max_persite_dlnl"] if parity else None,
            "parity": parity,
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
