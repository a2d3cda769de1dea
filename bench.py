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

The default run (one GPU, no overrides) times C3 -- the configuration the north star quotes -- and then, in
the same process, a few steps of C2, C4 and C5, one Newton-Raphson smoothing pass on the 125 k-site slice of
C3 and one SPR round on the 25 k-site slice of C5 (the per-GPU shares of an 8-way split): their figures go
into `also` of the same line (`--no-also` skips them).

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
    ap.add_argument("--site-repeats", action="store_true",
                    help="PLL_ATTRIB_SITE_REPEATS (first step: cherries are computed per class of sites, not per site)")
    ap.add_argument("--clv-tips", action="store_true",
                    help="partitions WITHOUT PLL_ATTRIB_PATTERN_TIP (tips are vectors set through pll_set_tip_states): the form "
                         "libpll combines with PLL_ATTRIB_SITE_REPEATS, and the one the reference's harness runs under 'sr'")
    ap.add_argument("--data", default="random", choices=["random", "simulated", "tiled"],
                    help="random: iid uniform tip states (seed 44); simulated: states evolved along the tree (SURVEY.md 8d, seed 45); "
                         "tiled: copies of one random 1000-site tile (the extreme of site repeats: no node has more than 1000 classes)")
    ap.add_argument("--rate-scalers", action="store_true",
                    help="PLL_ATTRIB_RATE_SCALERS: one scaling count per (site, rate) instead of per site")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "torch"],
                    help="N > 1: rccl = the library's own RCCL communicator, lnL reduced on the device (default); "
                         "torch = the reduce callback through torch.distributed (a rehearsal path: with "
                         "PLLHIP_BENCH_DIST_BACKEND=gloo and PLLHIP_ALLOW_DEVICE_WRAP=1 all ranks can share one GPU)")
    ap.add_argument("--transient", action="store_true",
                    help="evaluate-only traversals (pllhip_eval_set_transient): the vectors inside operation chains stay in "
                         "registers; what a model-parameter optimiser's full evaluations need")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pmc", default="auto", choices=["auto", "on", "off"],
                    help="roofline.traffic from two rocprofv3 --pmc child passes of this very command (FETCH_SIZE, WRITE_SIZE), "
                         "started before this process touches the GPU; auto: for the default line only")
    ap.add_argument("--tree", default="random", choices=["random", "ladder", "balanced"],
                    help="tree shape: random stepwise addition (seed 42, the benchmark's), a caterpillar, a complete binary tree")
    ap.add_argument("--as-rank", default="",
                    help="R/N: run, on ONE GPU and without a reduction, the share rank R of N would hold (the cost-balanced "
                         "partition assignment of --config c4); `value` then counts that share only")
    ap.add_argument("--no-also", action="store_true",
                    help="default run only: skip the C2 / C4 / C5 / branch-length / SPR legs behind the C3 line")
    ap.add_argument("--also-steps", type=int, default=5)
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
CODED_TIPS = True   # PLL_ATTRIB_PATTERN_TIP (--clv-tips: without)


def make_evaluation(pc, lib, tree, plan, rate_cats, seed, per_branch, first_sites=None):
    """one Evaluation (C driver) over the partitions of `plan` = [(states, sites), ...];
    partition k holds sites first_sites[k] .. of the alignment that seed + 101 k defines"""
    ev = pc.Evaluation(lib, tree.newick(), flags=1 if per_branch else 0, nparts=len(plan))
    insts = []
    for k, (states, nsites) in enumerate(plan):
        subst, freqs, alpha = model_of(pc, states)
        codes = pc.random_codes(tree.ntips, nsites, states, seed + 101 * k,
                                first_site=first_sites[k] if first_sites else 0)
        insts.append(ev.add_partition(k, states, nsites, rate_cats, codes, subst, freqs, alpha, coded=CODED_TIPS,
                                      attributes=ATTRIBUTES))
    return ev, insts


def evaluate_with_persite(ev, plan):
    """lnL and the per-site lnL of every partition, concatenated"""
    lnl = ev.loglh()
    persite = np.concatenate([ev.persite_lnl(k)[1] for k in range(len(plan))])
    return lnl, persite


def usable_cores():
    """cores this process may really use: the affinity mask and the cgroup CPU quota (a one-GPU box gets a
    16-core share of a 256-core host; 256 OpenMP threads on it run slower than 16)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                quota = int(parts[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = int(f.read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def _omp_set_threads(n):
    try:
        C.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
        return True
    except OSError:
        return False


def _time_oracle(pc, oracle, tree, plan, rate_cats, per_branch, budget_s, max_reps=5):
    """full evaluations of `plan` on the oracle until budget_s has passed: (updates/s, reps, lnl, persite)"""
    ev, _ = make_evaluation(pc, oracle, tree, plan, rate_cats, 44, per_branch)
    with ev:
        ev.loglh()                                  # warm-up
        t0 = time.perf_counter()
        reps = 0
        while True:
            ev.loglh()
            reps += 1
            if time.perf_counter() - t0 > budget_s or reps >= max_reps:
                break
        dt = time.perf_counter() - t0
        lnl, persite = evaluate_with_persite(ev, plan)
    sites = sum(n for _, n in plan)
    return reps * (tree.ntips - 2) * sites * rate_cats / dt, reps, lnl, persite


def cpu_baseline(pc, product, tree, config, states, rate_cats, nsites_gpu, cpu_sites, per_branch, quick=False):
    """the oracle (a from-scratch CPU port, NOT libpll) timed on the host cores
    on a bounded sample of the same workload: same tree, model, C driver and the
    FIRST cpu_sites sites of the same alignment, sized for ~10 s of CPU work at 16 threads;
    also with ONE thread and with ALL cores of the host, each on a sample
    sized for a few seconds.  The 16-thread sample is then evaluated on the GPU: |dlnL| per site on
    identical inputs (BASELINE.json's metric, second half).  quick: parity sample only (no timing legs)."""
    host_cores = os.cpu_count() or 1
    usable = usable_cores()
    threads = min(usable, 16)          # the 1-GPU box share of host cores
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("ORC_FAST", "1")      # the oracle's AVX2-vectorised partials (4 / 20 / 61 states)
    oracle = pc.PllLib(os.path.join(ROOT, "oracle", "_build", "libpll_oracle.so"))
    can_set = _omp_set_threads(threads)
    ntips = tree.ntips
    per_update = (4.0 * states * states + states) / 1.5e9      # ~scalar flop rate of one core

    def sample(seconds, nthreads):
        n = int(seconds * nthreads / (per_update * rate_cats * (ntips - 2)))
        return max(1000, min(nsites_gpu, n // 1000 * 1000))
    if not cpu_sites:
        cpu_sites = sample(1.0 if quick else 15.0, threads)
    plan = partition_plan(config, states, cpu_sites)
    cpu_sites = sum(n for _, n in plan)
    rate, reps, lnl_cpu, persite_cpu = _time_oracle(pc, oracle, tree, plan, rate_cats, per_branch,
                                                    0.5 if quick else 10.0, 1 if quick else 5)
    gev, _ = make_evaluation(pc, product, tree, plan, rate_cats, 44, per_branch)
    with gev:
        lnl_gpu, persite_gpu = evaluate_with_persite(gev, plan)
    parity = {
        "dlnl_per_site": abs(lnl_gpu - lnl_cpu) / cpu_sites,
        "max_persite_dlnl": float(np.max(np.abs(persite_gpu - persite_cpu))),
        "sample_sites": cpu_sites, "lnl_gpu": lnl_gpu, "lnl_cpu": lnl_cpu,
        "against": "oracle/ (CPU restatement pinned by the reference's golden files; libpll itself is "
                   "not available offline) on the first sample_sites sites of the benchmark alignment",
        "tolerance_per_site": 1e-6,
    }
    if quick:
        return None, parity
    base = {"value": rate, "unit": "CLV site-updates/s", "cores": threads, "kind": "port",
            "host_cores": host_cores, "usable_cores": usable, "threads": threads, "sample_sites": cpu_sites,
            "sample_evaluations": reps,
            "lnl_per_site": lnl_cpu / cpu_sites,
            "sample": f"{reps} full evaluations of the same tree/model on the first {cpu_sites} sites of the "
                      f"benchmark alignment (oracle/, plain C + OpenMP over sites, AVX2-vectorised partials, "
                      f"{threads} threads on a host with {host_cores} cores)"}
    # one thread, and every core of the host: a few seconds each, own sample sizes
    if can_set:
        for label, nthreads in (("one_thread", 1), ("all_cores", usable)):
            if label == "all_cores" and usable == threads:
                base[label] = {"value": rate, "threads": threads, "sample_sites": cpu_sites, "same_as": "value"}
                continue
            _omp_set_threads(nthreads)
            n = sample(4.0, nthreads)
            sub = partition_plan(config, states, n)
            r, k, _, _ = _time_oracle(pc, oracle, tree, sub, rate_cats, per_branch, 4.0, 3)
            base[label] = {"value": r, "threads": nthreads, "sample_sites": sum(x for _, x in sub),
                           "sample_evaluations": k}
        _omp_set_threads(threads)
    return base, parity


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def kfd_gpu_count():
    """GPU nodes of /sys/class/kfd/kfd/topology (no HIP call, no torch import); None if unreadable"""
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(base):
            with open(os.path.join(base, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
        if vis:
            n = min(n, len([v for v in vis.split(",") if v.strip() != ""]))
        return n
    except (OSError, ValueError):
        return None


def launch_ranks(args):
    """`bench.py --gpus N` without a rank environment: start the N ranks as ONE child
    process (torch.distributed.run -> one process per GPU).  Nothing in this process has
    touched HIP or torch.cuda at this point, and nothing will: the parent only forwards
    the child's JSON line and exit code."""
    probe = os.environ.get("PLLHIP_BENCH_LAUNCH_PROBE") == "1"
    if not probe:
        # the parent stays free of torch and HIP (a device count through either may initialise the runtime
        # in the process the ranks are forked from): GPUs are counted from the KFD topology, and a rank that
        # finds no device for its LOCAL_RANK fails on pllhip_set_device anyway
        have = kfd_gpu_count()
        if have is not None and have < args.gpus and os.environ.get("PLLHIP_ALLOW_DEVICE_WRAP") != "1":
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


def partition_cost(states, rate_cats):
    """relative cost of one site of a partition: bytes per site across one operation for the HBM-bound
    families (24 S per rate; 20 % on top where the matrix pipe works next to the memory system:
    measured 6.0 : 1 between 20 and 4 states at the slice sizes of an 8-way split), flops over the ridge
    (~10 flop/B) where the matrix pipe bounds it"""
    per_rate = 24.0 * states * (1.2 if states > 16 else 1.0)
    return rate_cats * max(per_rate, 2.0 * states * states / 10.2)


# what a further slice on a rank costs (its own launches, tables, lnL kernel), in units of partition_cost(4, 4) sites:
# measured at the C4 slices of an 8-way split (two slices of one family on a rank: +0.15 ms of 1.04)
SLICE_OVERHEAD_SITES = 30000.0


def assign_partitions(job_plan, rate_cats, rank, world):
    """Cost-balanced assignment of the partitions of a job to the ranks: all partitions are
    laid end to end, every site weighted with its partition's cost, and cut into `world` consecutive pieces --
    a rank gets the (few, large) partition slices inside its piece and NULL slots for the rest
    (src/tree/treeinfo.c:1024-1031), instead of a 1/world slice of EVERY partition.  The cuts minimise the load
    of the busiest rank, a slice costing a fixed overhead on top of its sites (so that no rank is left with a
    sliver of a second partition): the smallest budget with which a greedy walk places everything on `world`
    ranks, found by bisection.  Slice borders sit on whole 32-site blocks.  Returns [(first_site, sites) or None
    per partition]."""
    unit = partition_cost(4, rate_cats)
    cost = [partition_cost(s_, rate_cats) / unit for s_, _ in job_plan]
    sizes = [n_ for _, n_ in job_plan]

    def walk(budget):
        """greedy: [per rank: [(partition, first, sites)]] or None if `world` ranks do not suffice"""
        ranks, cur, load = [], [], 0.0
        for k, n_ in enumerate(sizes):
            first = 0
            while first < n_:
                room = budget - load - (SLICE_OVERHEAD_SITES if cur else 0.0)
                take = int(room / cost[k]) // 32 * 32
                if take >= n_ - first:
                    take = n_ - first
                if take <= 0 or (take < 32 and take < n_ - first):
                    if not cur:
                        return None                    # the budget does not even hold one block
                    ranks.append(cur)
                    cur, load = [], 0.0
                    if len(ranks) >= world:
                        return None
                    continue
                load += (SLICE_OVERHEAD_SITES if cur else 0.0) + take * cost[k]
                cur.append((k, first, take))
                first += take
        if cur:
            ranks.append(cur)
        return ranks if len(ranks) <= world else None

    total = sum(c * n_ for c, n_ in zip(cost, sizes))
    lo, hi = total / world, total + SLICE_OVERHEAD_SITES * len(sizes)
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if walk(mid) is None:
            lo = mid
        else:
            hi = mid
    ranks = walk(hi) or []
    out = [None] * len(sizes)
    if rank < len(ranks):
        for k, first, take in ranks[rank]:
            out[k] = (first, take)
    return out


# ---------------------------------------------------------------------------
# roofline.traffic, measured in the run that reports it: HBM bytes of the partials kernels from the PMC counters,
# collected and corrected as /opt/skills/guides/MI355X_MICROARCH.md (section HBM) prescribes -- FETCH_SIZE and WRITE_SIZE
# in separate `rocprofv3 --pmc` passes (they do not fit one), no trace domains next to them, the program directly
# after `--`; the counters are in KiB, FETCH_SIZE reports half of the bytes of a wide coalesced streaming read on
# gfx950 (doubled here), WRITE_SIZE is exact for 16-B-per-lane streaming stores.  The two passes are CHILD processes
# of a parent that has not touched the GPU yet, each running this command for two evaluations (one warm-up, one step).
# ---------------------------------------------------------------------------
PARTIALS_KERNELS = ("k_traverse", "k_chain", "k_partials", "k_s61_scale_fixup", "k_s61_cherry_scale", "k_cherry_build",
                    "k_pair_lut", "k_class_")


def _pmc_pass(counter, child_args, outdir):
    import csv
    import glob
    import shutil
    import subprocess
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", outdir, "--", sys.executable,
           os.path.abspath(__file__)] + child_args
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
    except (OSError, subprocess.TimeoutExpired) as exc:
        return None, f"{type(exc).__name__}: {exc}"
    files = glob.glob(os.path.join(outdir, "**", "*_counter_collection.csv"), recursive=True)
    if r.returncode != 0 or not files:
        return None, f"rc {r.returncode}: {r.stderr[-300:]}"
    per_kernel, steps = {}, 0
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"].split("(")[0]
            a = per_kernel.setdefault(name, [0, 0.0])
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    steps = sum(v[0] for k, v in per_kernel.items() if "k_edge_lnl" in k)
    return {"per_kernel": per_kernel, "evaluations": steps}, None


def measure_traffic(args):
    """HBM bytes the partials kernels of ONE evaluation move (dict), or a dict with only `error`"""
    import shutil
    import tempfile
    child = ["--config", args.config, "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-also", "--pmc", "off",
             "--pmatrix-calls", args.pmatrix_calls, "--data", args.data, "--tree", args.tree]
    if args.sites:
        child += ["--sites", str(args.sites)]
    if args.taxa:
        child += ["--taxa", str(args.taxa)]
    for flag in ("site_repeats", "rate_scalers", "transient", "clv_tips"):
        if getattr(args, flag, False):
            child.append("--" + flag.replace("_", "-"))
    out = {}
    t0 = time.perf_counter()
    for counter, factor in (("FETCH_SIZE", 2.0 * 1024.0), ("WRITE_SIZE", 1024.0)):
        d = tempfile.mkdtemp(prefix="pllhip_pmc_", dir="/tmp")
        try:
            res, err = _pmc_pass(counter, child, d)
        finally:
            shutil.rmtree(d, ignore_errors=True)
        if err:
            return {"error": f"{counter}: {err}"}
        evals = max(1, res["evaluations"])
        tot = sum(v[1] for k, v in res["per_kernel"].items() if any(x in k for x in PARTIALS_KERNELS))
        n = sum(v[0] for k, v in res["per_kernel"].items() if any(x in k for x in ("k_traverse", "k_chain", "k_partials")))
        out[counter] = {"bytes_per_evaluation": tot * factor / evals, "raw_kib_per_evaluation": tot / evals,
                        "partials_launches_per_evaluation": n / evals, "evaluations": evals,
                        "kernels": {k: v[0] for k, v in res["per_kernel"].items() if any(x in k for x in PARTIALS_KERNELS)}}
    out["read_bytes_per_evaluation"] = out["FETCH_SIZE"]["bytes_per_evaluation"]
    out["write_bytes_per_evaluation"] = out["WRITE_SIZE"]["bytes_per_evaluation"]
    out["bytes_per_evaluation"] = out["read_bytes_per_evaluation"] + out["write_bytes_per_evaluation"]
    out["wall_s"] = round(time.perf_counter() - t0, 1)
    out["source"] = ("measured in this run: two `rocprofv3 --pmc` child passes of this command (FETCH_SIZE, WRITE_SIZE; "
                     "one warm-up + one step each), started before this process touched the GPU; KiB -> bytes, "
                     "FETCH_SIZE doubled (gfx950: MI355X_MICROARCH.md, HBM)")
    return out


class Ctx:
    """what every leg of a run shares"""
    def __init__(self, **kw):
        self.__dict__.update(kw)


def run_leg(ctx, config, sites=0, taxa=0, steps=10, warmup=2, cpu="full"):
    """one timed workload (W1 on `config`); returns the JSON object of its line.
    cpu: "full" = CPU baseline + parity sample, "parity" = parity sample only, None = neither"""
    pc, product, args = ctx.pc, ctx.product, ctx.args
    rank, world, dist, comm = ctx.rank, ctx.world, ctx.dist, ctx.comm
    states, rate_cats, ntips, nsites = pc.CONFIGS[config]
    if sites:
        nsites = sites
    if taxa:
        ntips = taxa
    per_branch = args.pmatrix_calls == "per-branch"
    tree = pc.Tree(ntips, 42, 43, ladder=(args.tree == "ladder"), balanced=(args.tree == "balanced"))

    # ONE alignment for the whole job (strong: the configured site count is split; weak: it is the per-GPU share).
    # One partition: rank r owns a contiguous site range.  Several partitions on several ranks: whole partitions /
    # large slices go to few ranks each, balanced by cost (assign_partitions); the others are NULL slots there.
    job_plan = partition_plan(config, states, nsites * (world if args.scaling == "weak" else 1))
    balanced = world > 1 and len(job_plan) > 1 and os.environ.get("PLLHIP_BENCH_BALANCE", "1") != "0"
    emulate = None
    if args.as_rank and world == 1:
        er, en = (int(x) for x in args.as_rank.split("/"))
        emulate = (er, en)
        balanced = len(job_plan) > 1 and os.environ.get("PLLHIP_BENCH_BALANCE", "1") != "0"
    if emulate and balanced:
        mine = assign_partitions(job_plan, rate_cats, emulate[0], emulate[1])
    elif emulate:
        mine = [(n_ * emulate[0] // emulate[1], n_ * (emulate[0] + 1) // emulate[1] - n_ * emulate[0] // emulate[1]) for _, n_ in job_plan]
    elif balanced:
        mine = assign_partitions(job_plan, rate_cats, rank, world)
    else:
        mine = [(n_ * rank // world, n_ * (rank + 1) // world - n_ * rank // world) for _, n_ in job_plan]
    plan = [(s_, m[1]) if m else None for (s_, _), m in zip(job_plan, mine)]
    first_sites = [m[0] if m else 0 for m in mine]
    local_sites = sum(p_[1] for p_ in plan if p_)
    total_sites = sum(n for _, n in job_plan)
    if not balanced and min(p_[1] for p_ in plan) < 1:
        raise SystemExit("bench.py: fewer sites than ranks")
    ev = pc.Evaluation(product, tree.newick(), flags=1 if per_branch else 0, nparts=len(plan))
    insts = []
    for k, p_ in enumerate(plan):
        if p_ is None:
            ev.add_remote_partition(k)
            continue
        subst, freqs, alpha = model_of(pc, p_[0])
        if args.data == "simulated":
            if world > 1:
                raise SystemExit("bench.py: --data simulated is a one-GPU data set")
            codes = pc.simulated_codes(tree, p_[1], p_[0], seed=45 + k)
        elif args.data == "tiled":
            if world > 1:
                raise SystemExit("bench.py: --data tiled is a one-GPU data set")
            tile = pc.random_codes(tree.ntips, 1000, p_[0], 44 + 101 * k)
            codes = np.tile(tile, (1, (p_[1] + 999) // 1000))[:, :p_[1]].copy()
        else:
            codes = pc.random_codes(tree.ntips, p_[1], p_[0], 44 + 101 * k, first_site=first_sites[k])
        insts.append(ev.add_partition(k, p_[0], p_[1], rate_cats, codes, subst, freqs, alpha, coded=CODED_TIPS,
                                      attributes=ATTRIBUTES))
        del codes
    if not insts:
        raise SystemExit("bench.py: a rank without work")
    inst = max(insts, key=lambda i: i.N * i.S)        # the partition that dominates the traffic
    if getattr(args, "transient", False):
        ev.set_transient(1)

    comm_mode = ctx.comm_mode
    reduce_cb = None
    if world > 1 and comm_mode == "rccl":
        if not product.lib.pllhip_eval_attach_comm(ev.ev, comm):
            raise SystemExit(product.errmsg)
    if world > 1 and comm_mode == "torch":
        # rehearsal path: the reference's reduce hook served by torch.distributed (any backend)
        import torch
        ops = {0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MAX, 2: dist.ReduceOp.MIN}    # include/pllhip.h PLLHIP_REDUCE_*
        on_gpu = dist.get_backend() == "nccl"

        def _reduce(c_, data, n, op):
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

    for _ in range(warmup):
        ev.loglh()
    for i in insts:
        product.lib.pllhip_profile_partials(i.p, 1)
    barrier()
    t0 = time.perf_counter()
    lnl = 0.0
    for _ in range(steps):
        lnl = ev.loglh()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = pc.Profile()           # summed over the partitions of the evaluation
    families = {}                 # kernel family -> its launches (partitions of a family share theirs: one stream's events)
    for i in insts:
        one = pc.Profile()
        product.lib.pllhip_profile_read(i.p, C.byref(one))
        product.lib.pllhip_profile_partials(i.p, 0)
        for f, _ in pc.Profile._fields_:
            setattr(prof, f, getattr(prof, f) + getattr(one, f))
        fam = families.setdefault(product.lib.pllhip_partials_kernel_name(i.p).decode(), pc.Profile())
        for f, _ in pc.Profile._fields_:
            setattr(fam, f, getattr(fam, f) + getattr(one, f))
    counters = [i.counters() for i in insts]
    repeats = None
    if args.site_repeats:
        st = [i.repeat_stats() for i in insts]
        repeats = {"class_operations_per_step": sum(x.cherries for x in st) // max(1, steps + warmup),
                   "operations_per_step": nops,
                   "classes": sum(x.classes for x in st), "sites": sum(x.sites for x in st),
                   "classes_over_sites": (sum(x.classes for x in st) / max(1, sum(x.sites for x in st))),
                   "expansions": sum(x.expansions for x in st),
                   "what": "operations computed per class of sites (cherries: pairs of tip codes; nodes above them: pairs of "
                           "their children's classes) instead of per site"}
    partial_launches = sum(c.partial_launches for c in counters)
    pmatrix_launches = sum(c.pmatrix_launches for c in counters)

    if dist is not None:
        import torch
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    kernel = product.lib.pllhip_partials_kernel_name(inst.p).decode()
    updates_per_step = nops * (local_sites if emulate else total_sites) * rate_cats
    value = updates_per_step * steps / elapsed

    # roofline of the dominant kernel (pll_update_partials), from the HIP events
    # recorded around its launches during the timed region on rank 0
    several = len(insts) > 1
    kernel_s = prof.kernel_ms * 1e-3
    achieved = prof.algorithmic_bytes / kernel_s / 1e9 if kernel_s > 0 else 0.0
    minimum = prof.minimum_bytes / kernel_s / 1e9 if kernel_s > 0 else 0.0
    per_family = None
    if several:
        # partitions of different families run on their own streams and overlap on the device: the summed
        # event times exceed the wall time, so the kernels are priced against the
        # whole step instead (a lower bound of what they achieve) ...
        achieved = prof.algorithmic_bytes / elapsed / 1e9
        minimum = prof.minimum_bytes / elapsed / 1e9
        # ... and per family against the HIP events of its own launches (the partitions of a family share one
        # launch per round on one stream; while it runs the other family's launches share the chip with it)
        per_family = []
        for name, fp in sorted(families.items()):
            if not fp.launches or fp.kernel_ms <= 0:
                continue
            fs = fp.kernel_ms * 1e-3
            per_family.append({"kernel": name, "launches": int(fp.launches), "ops": int(fp.ops),
                               "avg_launch_ms": round(fp.kernel_ms / fp.launches, 4),
                               "algorithmic_GBps": round(fp.algorithmic_bytes / fs / 1e9, 1),
                               "frac": round(fp.algorithmic_bytes / fs / 1e9 / HBM_PEAK_GBS, 4),
                               "frac_minimum": round(fp.minimum_bytes / fs / 1e9 / HBM_PEAK_GBS, 4),
                               "kernel_share_of_step": round(fs / elapsed, 4)})
    traffic = None
    traffic_source = None
    measured = getattr(ctx, "traffic", None) if config == args.config and not sites and not taxa else None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if measured and "bytes_per_evaluation" in measured and prof.launches:
        traffic = round(measured["bytes_per_evaluation"] * steps / prof.launches)       # per launch, like `achieved`
        traffic_source = measured["source"]
    elif os.path.exists(tpath) and world == 1 and not sites and not taxa and not args.rate_scalers and not args.site_repeats \
            and args.data == "random" and args.tree == "random" and not getattr(args, "transient", False):
        try:
            tj = json.load(open(tpath))
            per_step = tj.get(f"{config}:{kernel}:per_step")
            if per_step and prof.launches:
                traffic = round(per_step * steps / prof.launches)      # per launch, like `achieved`
            else:
                traffic = tj.get(f"{config}:{kernel}")
            traffic_source = ("profiles/traffic.json: committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                              "this command (tools/gpu_profile.sh), NOT measured in this run")
        except Exception:
            traffic = None
    tflops = prof.algorithmic_flops / kernel_s / 1e12 if kernel_s > 0 else 0.0
    # the kernel is priced against the roof that bounds it: HBM for 4 and 20 states
    # (0.7 and 3.4 flop/B), the FP64 matrix pipe for 61 states (10.2 flop/B ~ the ridge)
    mfma_bound = states > 32
    launches = max(1, prof.launches)
    traffic_gbps = (traffic / (prof.kernel_ms / launches * 1e-3) / 1e9
                    if traffic and prof.kernel_ms > 0 and not several else None)
    roofline = {
        "bound": "mfma" if mfma_bound else "hbm", "kernel": kernel,
        "achieved": round(tflops, 2) if mfma_bound else round(achieved, 1),
        "peak": FP64_MFMA_PEAK_TFLOPS if mfma_bound else HBM_PEAK_GBS,
        "unit": "TFLOP/s" if mfma_bound else "GB/s",
        "frac": round(tflops / FP64_MFMA_PEAK_TFLOPS, 4) if mfma_bound else round(achieved / HBM_PEAK_GBS, 4),
        "traffic": traffic, "traffic_source": traffic_source,
        "traffic_measured": ({k: measured[k] for k in ("read_bytes_per_evaluation", "write_bytes_per_evaluation",
                                                       "bytes_per_evaluation", "wall_s")} | {"passes": {c: measured[c] for c in ("FETCH_SIZE", "WRITE_SIZE")}}
                             if measured and "bytes_per_evaluation" in measured else
                             ({"error": measured["error"]} if measured else None)),
        "algorithmic_GBps": round(achieved, 1), "algorithmic_TFLOPs": round(tflops, 2),
        # `frac` prices SURVEY.md 8d's algorithmic bytes (every child vector one read).  Operation chains hand
        # the carried child over in registers, so the schedule has to move less than that:
        #   frac_minimum   the bytes the schedule as planned must move (carried children and their scaler
        #                  counts not read; every vector still written) / time / peak: the rate an HBM-bound
        #                  kernel can be held to -- this is the figure that still discriminates
        #   frac_physical  the PMC traffic of the committed profile of this command / time / peak
        "minimum_GBps": round(minimum, 1),
        "frac_minimum": round(minimum / HBM_PEAK_GBS, 4),
        "minimum_over_algorithmic": (round(prof.minimum_bytes / prof.algorithmic_bytes, 4)
                                     if prof.algorithmic_bytes > 0 else None),
        "frac_physical": round(traffic_gbps / HBM_PEAK_GBS, 4) if traffic_gbps else None,
        "hbm_copy_rate_measured_GBps": 6290.0,      # MI355X_MICROARCH.md: what a plain copy reaches of the 8 TB/s
        "launches": int(prof.launches), "ops": int(prof.ops),
        "avg_launch_ms": round(prof.kernel_ms / launches, 4),
        "algorithmic_bytes_per_launch": round(prof.algorithmic_bytes / launches),
        "minimum_bytes_per_launch": round(prof.minimum_bytes / launches),
        # (several families: the share of the family whose launches take longest -- they overlap on their streams)
        "kernel_share_of_step": round((max(f_["kernel_share_of_step"] for f_ in per_family) if per_family else kernel_s / elapsed), 4),
        "families": per_family,
        # tools/micro/mfma_f64.hip: v_mfma_f64_16x16x4_f64 back to back on every SIMD sustains
        # 47.5 TFLOP/s on this chip (29.9 ns per MFMA per SIMD with <= 128 CUs busy, 44.1 ns with
        # all 256: clocks drop under a full-chip FP64 matrix load), not the 78.6 TFLOP/s data-sheet peak
        "mfma_sustained_measured_TFLOPs": FP64_MFMA_SUSTAINED_TFLOPS if mfma_bound else None,
        "frac_of_sustained": round(tflops / FP64_MFMA_SUSTAINED_TFLOPS, 4) if mfma_bound else None,
        "traffic_GBps": round(traffic_gbps, 1) if traffic_gbps else None,
        "traffic_over_algorithmic": (round(traffic / (prof.algorithmic_bytes / launches), 3)
                                     if traffic and prof.algorithmic_bytes > 0 and not several else None),
        "method": ("algorithmic bytes of all partials launches / step wall time (partitions overlap on "
                   "concurrent streams)") if several else
                  "algorithmic bytes of the partials launches / their HIP-event time (rank 0)",
    }

    ev.close()                   # frees the CLVs: the parity sample below gets its own partitions
    cpu_base = parity = None
    if rank == 0 and world == 1 and not ctx.internal and cpu and not emulate:
        cpu_base, parity = cpu_baseline(pc, product, tree, config, states, rate_cats, local_sites,
                                        args.cpu_sites, per_branch, quick=(cpu == "parity"))
    names = {"c2": "C2 DNA GTR+G4", "c4": "C4 mixed: 2 DNA GTR+G4 + 2 protein GTR20+G4 partitions, linked branch lengths",
             "c3": "C3 protein 'LG-shaped' GTR20+G4 (real LG table unavailable offline)",
             "c5": "C5 codon GY94-shaped+G4"}
    evals = max(1, steps + warmup)
    comm_note = ""
    if world > 1:
        comm_note = (" + RCCL all-reduce of the lnL on the device" if comm_mode == "rccl"
                     else " + lnL summed through the reduce hook (torch.distributed)")
    return {
        "metric": "CLV site-updates/sec (sites x rates x edges); |dlnL| vs ref",
        "value": value, "unit": "CLV site-updates/s",
        "n_gpus": args.gpus if ctx.internal else world, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": f"{names[config]}: {states} states, {rate_cats} rate cats, {ntips} taxa, "
                        f"{total_sites} sites total ({local_sites} on rank 0), full traversal "
                        f"({nops} ops + {tree.nedges} P-matrices + edge lnL{comm_note}) per step",
            "config": config, "states": states, "rate_cats": rate_cats, "taxa": ntips,
            "sites_total": total_sites, "sites_per_gpu": local_sites, "ops_per_step": nops,
            "partitions": [({"states": p_[0], "first_site": f_, "sites_on_rank0": p_[1]} if p_ else
                            {"states": job_plan[k][0], "sites_on_rank0": 0, "remote": True})
                           for k, (p_, f_) in enumerate(zip(plan, first_sites))],
            "emulated_rank": (f"{emulate[0]}/{emulate[1]}" if emulate else None),
            "partition_assignment": ("cost-balanced: whole partitions / large slices per rank, NULL slots elsewhere"
                                     if balanced else "every rank holds a contiguous 1/N site range of every partition"),
            "tips": "1-byte codes" if CODED_TIPS else "vectors (no PLL_ATTRIB_PATTERN_TIP), set through pll_set_tip_states",
            "scalers": ("per (site, rate)" if args.rate_scalers else "per-site") + ", one buffer per inner node",
            "pmatrix_calls": args.pmatrix_calls, "alignment": args.data, "tree": args.tree,
            "site_repeats": repeats,
            "transient": bool(getattr(args, "transient", False)),
            "pmatrix_launches_per_step": pmatrix_launches // evals,
            "partial_launches_per_step": partial_launches // evals,
            "parallelism": (f"one process, every partition spread over {args.gpus} devices inside the engine, "
                            f"lnL summed on the host" if ctx.internal else
                            f"sites sharded over {world} GPU(s), lnL all-reduced" if world > 1 else "1 GPU"),
        },
        # the same alignment at every N: the summed lnL must not depend on n_gpus
        "lnl": lnl, "lnl_per_site": lnl / total_sites,
        "dlnl_per_site": parity["dlnl_per_site"] if parity else None,
        "max_persite_dlnl": parity["max_persite_dlnl"] if parity else None,
        "parity": parity,
        "roofline": roofline,
        "cpu_baseline": cpu_base,
    }


def also_legs(ctx):
    """behind the default C3 line, in the same process: the other BASELINE configurations (a few steps each, with a
    small parity sample) and the two secondary workloads VERDICT r2 asks for at the per-GPU slice of an 8-way split"""
    also = {}
    steps = ctx.args.also_steps
    for cfg in ("c2", "c4", "c5"):
        t0 = time.perf_counter()
        d = run_leg(ctx, cfg, steps=steps, warmup=2, cpu="parity")
        r = d["roofline"]
        also[cfg] = {
            "workload": d["config"]["workload"], "value": d["value"], "ms_per_step": d["ms_per_step"], "steps": steps,
            "roofline": {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_minimum",
                                           "frac_physical", "frac_of_sustained", "traffic_over_algorithmic",
                                           "avg_launch_ms", "launches", "kernel_share_of_step", "families")},
            "traffic_over_algorithmic": r["traffic_over_algorithmic"],
            "dlnl_per_site": d["dlnl_per_site"], "max_persite_dlnl": d["max_persite_dlnl"],
            "parity_sample_sites": d["parity"]["sample_sites"] if d["parity"] else None,
            "partial_launches_per_step": d["config"]["partial_launches_per_step"],
            "leg_wall_s": round(time.perf_counter() - t0, 2),
        }
    # the headline configuration once more with PLL_ATTRIB_SITE_REPEATS: operations computed per class of sites
    # (same iid-uniform data as the headline: its worst case), with the same parity sample against the oracle
    global ATTRIBUTES
    saved = (ATTRIBUTES, ctx.args.site_repeats)
    ATTRIBUTES |= ctx.pc.PLL_ATTRIB_SITE_REPEATS
    ctx.args.site_repeats = True
    try:
        t0 = time.perf_counter()
        d = run_leg(ctx, "c3", steps=steps, warmup=2, cpu="parity")
        also["c3_site_repeats"] = {
            "workload": d["config"]["workload"], "value": d["value"], "ms_per_step": d["ms_per_step"], "steps": steps,
            "site_repeats": d["config"].get("site_repeats"),
            "dlnl_per_site": d["dlnl_per_site"], "max_persite_dlnl": d["max_persite_dlnl"],
            "leg_wall_s": round(time.perf_counter() - t0, 2),
        }
    finally:
        ATTRIBUTES, ctx.args.site_repeats = saved
    # the headline configuration as a model-parameter optimiser evaluates it: full evaluations whose vectors nobody
    # reads before the next one recomputes them (src/algorithm/algo_callback.c:338, 465, 568, 678) -- evaluate-only
    # traversals (include/pllhip.h, pllhip_set_transient): same lnL to the last bit, the vectors inside the operation
    # chains are not stored
    ctx.args.transient = True
    try:
        t0 = time.perf_counter()
        d = run_leg(ctx, "c3", steps=steps, warmup=2, cpu="parity")
        r = d["roofline"]
        also["c3_transient"] = {
            "workload": d["config"]["workload"] + "; evaluate-only", "value": d["value"], "ms_per_step": d["ms_per_step"],
            "steps": steps, "lnl": d["lnl"],
            "roofline": {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_minimum", "minimum_GBps",
                                           "minimum_bytes_per_launch", "avg_launch_ms", "launches", "kernel_share_of_step")},
            "dlnl_per_site": d["dlnl_per_site"], "max_persite_dlnl": d["max_persite_dlnl"],
            "leg_wall_s": round(time.perf_counter() - t0, 2),
        }
    finally:
        ctx.args.transient = False
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gpu_workloads as gw
    out = {}
    t0 = time.perf_counter()
    gw.blo(ctx.product, "c3", out, nsites=125_000)
    b = out["BLO_c3_125000"]
    also["blo_c3_125k_us_per_iterate"] = b["us_per_derivative_call_incl_everything"]
    also["blo_c3_125k"] = {k: b[k] for k in ("s_per_smoothing_pass", "newton_iterations", "sumtable_scans",
                                             "single_op_updates", "pmatrix_updates", "branches", "lnl_before", "lnl_after")}
    also["blo_c3_125k"]["leg_wall_s"] = round(time.perf_counter() - t0, 2)
    # ... and over C4's four partitions (linked branch lengths) at the per-GPU slice of an 8-way split: the partitions'
    # derivative sums meet between two iterates -- on the device (include/pllhip.h, pllhip_newton_branch_multi: one launch
    # for all four partitions, k_newton_multi)
    t0 = time.perf_counter()
    gw.blo_c4(ctx.product, out, nsites=125_000)
    b = out["BLO_c4_125000"]
    also["blo_c4_125k_us_per_iterate"] = b["us_per_derivative_call_incl_everything"]
    also["blo_c4_125k"] = {k: b[k] for k in ("s_per_smoothing_pass", "newton_iterations", "single_op_updates", "pmatrix_updates",
                                             "branches", "partitions", "lnl_before", "lnl_after", "device_newton")}
    also["blo_c4_125k"]["leg_wall_s"] = round(time.perf_counter() - t0, 2)
    t0 = time.perf_counter()
    gw.spr(ctx.product, out, nsites=25_000)
    sp = out["SPR_c5_50x25000_fast"]
    also["spr_c5_25k_s"] = sp["s_per_round"]
    also["spr_c5_25k"] = {k: sp[k] for k in ("prunings", "insertions", "moves_applied", "rescored", "clv_ops",
                                             "pmatrix_updates", "derivative_calls", "lnl_before", "lnl_after")}
    also["spr_c5_25k"]["leg_wall_s"] = round(time.perf_counter() - t0, 2)
    return also


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    in_rank_env = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    internal = args.topology == "internal" and args.gpus > 1
    if internal:
        have = kfd_gpu_count()
        if have is not None and have < args.gpus and os.environ.get("PLLHIP_ALLOW_DEVICE_WRAP") != "1":
            raise SystemExit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible")
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

    # kernel arguments in device memory (3 - 5 us per short launch, pll_core.hip): decided HERE, before any
    # library that initialises the HIP runtime is loaded, so that N = 1 and N > 1 run under the same setting
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    default_shape = (world == 1 and not internal and args.config == "c3" and not args.sites and not args.taxa and
                     not args.rate_scalers and not args.no_also and not args.no_cpu_baseline and not args.site_repeats and
                     args.data == "random" and args.tree == "random" and not args.transient and not args.clv_tips)
    traffic_measured = None
    if world == 1 and not internal and not args.as_rank and (args.pmc == "on" or (args.pmc == "auto" and default_shape)):
        traffic_measured = measure_traffic(args)      # child processes; nothing here has touched the GPU yet
    if world > 1:
        # torch brings its own HIP and RCCL runtimes (same SONAMEs as /opt/rocm's): whichever is
        # loaded first serves the whole process, and torch does not find the GPU on the system's.
        # Load torch first, so that the engine binds to the runtime torch.cuda uses.
        import torch  # noqa: F401
    import pllhip_ctypes as pc
    global ATTRIBUTES, CODED_TIPS
    CODED_TIPS = not args.clv_tips
    if args.rate_scalers:
        ATTRIBUTES = pc.PLL_ATTRIB_RATE_SCALERS
    if args.site_repeats:
        ATTRIBUTES |= pc.PLL_ATTRIB_SITE_REPEATS
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

    comm_mode = args.comm
    if world > 1 and comm_mode == "rccl":
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
        # every rank must take the same path: agree on the outcome before going on
        import torch
        flag = torch.tensor([1.0 if comm else 0.0], device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if flag.item() < 1.0:
            print(f"bench.py: rank {rank}: the library's communicator is not available ({product.errmsg}); "
                  "falling back to the reduce hook over torch.distributed", file=sys.stderr)
            if comm:
                product.lib.pllhip_comm_destroy(comm)
            comm = None
            comm_mode = "torch"

    ctx = Ctx(pc=pc, product=product, args=args, rank=rank, world=world, dist=dist, comm=comm,
              comm_mode=comm_mode, internal=internal, traffic=traffic_measured)
    out = run_leg(ctx, args.config, sites=args.sites, taxa=args.taxa, steps=args.steps, warmup=args.warmup,
                  cpu=None if args.no_cpu_baseline else "full")
    out["runtime"] = {"HIP_FORCE_DEV_KERNARG": os.environ.get("HIP_FORCE_DEV_KERNARG"),
                      "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
                      "torch_loaded_before_engine": world > 1}
    default_run = (world == 1 and not internal and args.config == "c3" and not args.sites and not args.taxa and
                   not args.rate_scalers and not args.no_also and not args.no_cpu_baseline and not args.site_repeats and
                   args.data == "random" and args.tree == "random" and not args.transient)
    if default_run:
        t0 = time.perf_counter()
        out["also"] = also_legs(ctx)
        out["also"]["wall_s"] = round(time.perf_counter() - t0, 1)

    if comm:
        product.lib.pllhip_comm_destroy(comm)
    if internal:
        product.lib.pllhip_set_sharding(0, None)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
