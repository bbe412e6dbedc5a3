#!/usr/bin/env python3
"""Benchmark of the MI355X GraphChainer hot path (BASELINE.json metric: reads/s and Gbp/s aligned).

One "step" = one pass of the hot path (whole-read GraphAligner pass; seed lookup -> seed ordering -> fragment
seed-extension -> anchors -> co-linear chaining -> chain stitching; NW edit distances, the chained-vs-whole-read
decision and the chained alignment's trace for the reads that take it) over the rank's reads, which are already
resident in HBM when the timed region starts.

--config 2 (default) = BASELINE.json configs[1]: chr22-like graph (50.8 Mbp backbone, SNP/indel bubbles every ~45 bp,
SURVEY.md §8d), 10 000 simulated 10 kb ONT-like reads, reference defaults; one batch per step.
--config 3 = configs[2]: the same graph, 100 000 reads in batches of 10 000, --colinear-split-gap 18 (the reference's
spelling of --sampling-step 0.5): one step = all ten batches.
For N>1 (weak scaling) every rank aligns its own reads against its own replica of the graph, no data-path collective;
--strong divides one read set over the ranks instead (BASELINE config 4's shape) with the product's work queue
(graphchainer_amd/workqueue.py): length-sorted batches handed out dynamically.

Usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|5] [--sv-fraction F]
       --gpus N > 1 run plainly (no WORLD_SIZE in the environment) starts the N ranks itself: a fresh child process
       `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py <same arguments>`, before this process has
       touched the GPU; rank 0's JSON line comes through and the exit code is the child's. Fewer than N visible devices: a loud refusal, not an N=1 number.
       (Under torch.distributed.run - the driver's form for N > 1 - the ranks are already there and nothing is started.)
"""
import argparse
import glob
import json
import os
import sys
import tempfile
import threading
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_TILE = 200           # algorithmic bytes of one (node, 64-row slice) tile
BYTES_PER_BACKTRACE_TILE = 176  # 80 + 80 + 16 per (slice, node) visited by the backtrace
BYTES_PER_TRACE_ITEM = 32


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--distinct-batches", type=int, default=None, help="read sets the timed steps cycle through (config 2: 4 - step s aligns set s mod 4, every set drawn with its own seed and checked "
                    "against the oracle's rows for its first reads; 1: every step re-aligns the same reads, as up to r5, whose graph lines then stay warm in L2 / Infinity Cache from step to step)")
    ap.add_argument("--dry-launch", action="store_true", help="--gpus N > 1 without WORLD_SIZE: print the command that would start the N ranks, and stop")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=int(os.environ.get("GC_BENCH_CONFIG", 2)), choices=[2, 3, 5])
    ap.add_argument("--chromosomes", type=int, default=24, help="config 5: weakly connected chromosome graphs in the GFA (two components each, one per strand)")
    ap.add_argument("--backbone", type=int, default=int(os.environ.get("GC_BENCH_BACKBONE", 0)), help="backbone bp (config 2 / 3: 50.8 Mbp; config 5: 8 Mbp per chromosome)")
    ap.add_argument("--reads", type=int, default=None, help="reads per rank and step (default: 10 000 for config 2, 100 000 for config 3)")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("GC_BENCH_BATCH", 10_000)), help="reads per gc_align_batch call")
    ap.add_argument("--read-len", type=int, default=None)
    ap.add_argument("--colinear-gap", type=int, default=None)
    ap.add_argument("--split-gap", type=int, default=None)
    ap.add_argument("--sv-fraction", type=float, default=float(os.environ.get("GC_BENCH_SV_FRACTION", 0.0)),
                    help="fraction of the reads that carry a 1.5 kb deletion the graph does not hold: the reads whose chained alignment wins (decision.chained_better > 0)")
    ap.add_argument("--cpu-sample", type=int, default=int(os.environ.get("GC_BENCH_CPU_SAMPLE", 600)), help="reads of the same workload timed on ONE thread of the CPU oracle (rank 0, N=1)")
    ap.add_argument("--cpu-threads", type=int, default=int(os.environ.get("GC_BENCH_CPU_THREADS", 0)), help="worker threads of the all-core CPU leg (0 = all host cores)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-long-pass", action="store_true", help="skip the whole-read GraphAligner pass (src/Aligner.cpp:630-654)")
    ap.add_argument("--strong", action="store_true", help="N>1: one read set divided over the ranks through the work queue (strong scaling)")
    ap.add_argument("--e2e-steps", type=int, default=int(os.environ.get("GC_BENCH_E2E_STEPS", 16)),
                    help="after the timed steps (N=1): this many steps with the read upload (gc_reads_upload) and the GAF encoding of every batch inside the step "
                         "(the whole boundary: host bases in, GAF text out); 0 skips it")
    ap.add_argument("--e2e-formats", default=os.environ.get("GC_BENCH_E2E_FORMATS", "gaf,gam,json"), help="end-to-end legs to run: gaf, gam, json (comma separated)")
    ap.add_argument("--sv-leg-steps", type=int, default=int(os.environ.get("GC_BENCH_SV_STEPS", 3)),
                    help="after the timed steps (N=1, config 2): this many steps over reads of which 20 %% carry a 1.5 kb deletion - the reads whose chained alignment wins "
                         "(k_edit_path, chained traces, the winners' output), with the oracle's summary of the first 1 000 for the parity check; 0 skips it")
    ap.add_argument("--repeats-leg-steps", type=int, default=int(os.environ.get("GC_BENCH_REPEATS_STEPS", 3)),
                    help="after that: this many steps on a second graph with pasted repeats (several seeds per fragment window), a quarter of the backbone; 0 skips it")
    ap.add_argument("--setup-dir", default=os.environ.get("GC_BENCH_SETUP_DIR"),
                    help="keep the GFA, the reads and the index cache of this workload in this directory and reuse them when a later run asks for the same workload "
                         "(N=1; profiling runs of one gpurun call: a 960 Mbp graph costs five minutes to generate, build and save, 85 s to load)")
    ap.add_argument("--no-index-cache", action="store_true", help="one rank: align on the graph and index as built, without writing the index cache and loading it back (config 5 at 3.1 Gbp: "
                    "a 59 GB file, ~8 min of set-up)")
    ap.add_argument("--host-memory-cap-gb", type=float, default=float(os.environ.get("GC_BENCH_HOST_CAP_GB", 0)), help="stop the run (exit 3) when the process's resident memory passes this: "
                    "a run that would otherwise take its host out of memory ends by itself (0, the default: no cap; config 5 at 3.1 Gbp was run with 290 on a 300 GiB host)")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("GC_BENCH_INFLIGHT", 5)),
                    help="batches in flight per GPU, each on its own gc_stream and host thread (like the reference's -t worker threads): one batch's seeding, fragment "
                         "pipeline, distances and assembly run beside another's whole-read pass (r5, ms per 10 k x 10 kb batch: " + ", ".join(f"{k} -> {v:.0f}" for k, v in sorted(BATCH_MS_BY_INFLIGHT.items())) + "; "
                         "a batch in flight holds 19 GB of device memory, the whole-read scratch of 27 GB is shared per device)")
    args = ap.parse_args()
    # config 5 on one GPU (BASELINE configs[4] is the whole genome over eight): 24 chromosome graphs, 2 000 CLR-like 50 kb reads, --colinear-gap 50000
    if args.reads is None:
        args.reads = 100_000 if args.config == 3 else 2_000 if args.config == 5 else 10_000
    if args.read_len is None:
        args.read_len = 50_000 if args.config == 5 else 10_000
    if args.colinear_gap is None:
        args.colinear_gap = 50_000 if args.config == 5 else 10_000
    if not args.backbone:
        args.backbone = 8_000_000 if args.config == 5 else 50_800_000
    if args.config == 5 and args.batch == 10_000:
        args.batch = 2_000
    if args.config == 5 and "GC_BENCH_INFLIGHT" not in os.environ and not any(a.startswith("--inflight") for a in sys.argv[1:]):
        args.inflight = 5                                  # 2 000 x 50 kb batches: 42 GB each on a 960 Mbp graph + the device's one 21 GB whole-read scratch + 31 GB of graph and index (r5, 960 Mbp: 4 -> 663, 5 -> 649 ms per batch; a sixth does not fit)
    if args.split_gap is None:
        args.split_gap = 18 if args.config == 3 else 35
    return args


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cpus():
    """CPUs this process can actually use: the affinity mask, cut to the cgroup's CPU bandwidth quota when there is one (a container may see
    all 256 hardware threads of the box and be allowed 16 CPUs' worth of time: cpu.max = "1600000 100000")."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
        except (OSError, ValueError):
            pass
    return max(1, n)


# Host CPU a batch costs and the batch period by batches in flight (r5 measurements on this pool's boxes, 10 k x 10 kb: host_cpu_s_per_step and the --inflight sweep,
# `gpurun_out/r5_inflight`; r3: 0.41 CPU-s and 204 / 189 / 173 / 165 / 157 ms): a rank keeps k batches in flight fed when it has HOST_CPU_S_PER_BATCH / period(k) CPUs to itself.
HOST_CPU_S_PER_BATCH = 0.24
BATCH_MS_BY_INFLIGHT = {1: 157.0, 2: 148.0, 3: 149.0, 4: 145.0, 5: 146.0}


def choose_inflight(asked, cpus, world):
    """Batches in flight per GPU from the CPU budget of a rank (usable CPUs / ranks): the largest count whose host work fits, at least 2 when that
    leaves the host as the bound anyway (the second batch costs no more CPU per batch - its waits sleep - and lets host work overlap device work)."""
    per_rank = cpus / max(1, world)
    demand = lambda k: HOST_CPU_S_PER_BATCH / (BATCH_MS_BY_INFLIGHT[min(k, 5)] / 1e3)
    fit = [k for k in range(1, asked + 1) if demand(k) <= per_rank]
    chosen = max(fit) if fit else min(asked, 2)
    why = (f"{chosen} in flight need {demand(chosen):.1f} CPUs of the {per_rank:.1f} a rank has ({cpus} usable / {world} ranks)" if fit else
           f"host-bound: even one batch in flight needs {demand(1):.1f} CPUs and a rank has {per_rank:.1f} ({cpus} usable / {world} ranks); {chosen} in flight so that host work overlaps device work")
    return chosen, {"asked": asked, "chosen": chosen, "cpus_per_rank": round(per_rank, 2), "cpu_demand_at_chosen": round(demand(chosen), 2), "why": why}


def cpu_baseline_leg(args, gfa, reads, long_pass):
    """The CPU restatement (oracle/) timed on the host cores of this box, BEFORE this process touches the GPU: one thread on a
    bounded sample, then one worker per usable CPU (affinity mask and cgroup quota) over a shared read queue (the reference's -t model,
    src/Aligner.cpp:1267-1270)."""
    from oracle import Oracle   # the CPU baseline leg is the one place bench.py may touch the oracle
    threads = args.cpu_threads or usable_cpus()
    ora = Oracle(gfa, long_pass=long_pass, split_gap=args.split_gap, colinear_gap=args.colinear_gap)
    n1 = min(args.cpu_sample, len(reads))
    if args.config == 5:
        n1 = min(n1, 40)                                   # (50 kb reads: a core does one or two per second)
    wall1, stage1 = ora.align_timed(reads[:n1], 1)
    n_all = min(len(reads), max(n1, (20 if args.config == 5 else 250) * threads))   # ~10-15 s at the ~20 reads/s a core does (10 kb reads)
    # the same run keeps 12 values per read (chain, chain score, both NW distances, the decision, the whole-read alignments and the
    # selection): main() compares them with the timed GPU output after the timed region ("parity_check")
    wall_all, _, summary = ora.align_summary(reads[:n_all], threads, gaf_hash=True)   # (last column: the hash of the read's GAF lines, for the end-to-end leg)
    extra = {}
    for name, more in (getattr(args, "extra_cpu_reads", None) or {}).items():           # the same oracle over the first reads of the SV leg (same graph)
        extra[name] = ora.align_summary(more, threads, gaf_hash=True)[2]
    ora.close()
    for name, (other_gfa, more) in (getattr(args, "extra_cpu_graphs", None) or {}).items():   # and an oracle of its own for the leg on the graph with repeats
        other = Oracle(other_gfa, long_pass=long_pass, split_gap=args.split_gap, colinear_gap=args.colinear_gap)
        extra[name] = other.align_summary(more, threads, gaf_hash=True)[2]
        other.close()
    args.extra_cpu_summaries = extra
    stage_names = ["seeding", "whole_read_pass", "fragment_extension+anchors", "chaining", "stitching+edlib"]
    total = float(stage1.sum()) or 1.0
    return {"value": round(n_all / wall_all, 2), "unit": "reads/s", "cores": threads, "kind": "port",
            "sample": f"first {n_all} reads of the same workload, same stages, {threads} worker threads over a shared read queue, {wall_all:.1f} s; one thread: first {n1} reads, {wall1:.1f} s",
            "cpu_model": cpu_model(), "host_hardware_threads": os.cpu_count(), "usable_cpus": usable_cpus(),
            "single_thread_reads_per_s": round(n1 / wall1, 2),
            "build": "g++ -O3; the restatement holds no assert() - the reference's assertions are restated as its throwing checks, which stay in - so an -DNDEBUG build is the same code (BASELINE.md §3 planned both figures)",
            "includes_output": "the GAF lines of every read are formatted and hashed inside the timed loop (the reference writes its output in the worker, src/Aligner.cpp:1003-1049)",
            "single_thread_stage_share": {k: round(float(v) / total, 3) for k, v in zip(stage_names, stage1)}}, summary


def launch_ranks(args, argv):
    """--gpus N > 1 without a launcher around us: the reference is ONE command (`-t N`, src/Aligner.cpp:1267-1270), so is this. Starts
    `python -m torch.distributed.run` as a child process (never an exec: this process may not have touched the GPU yet, and must not replace itself
    after it has) and returns its exit code; None when there is nothing to start (N == 1, or the ranks exist already)."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return None
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    forwarded = [a for a in argv if a != "--dry-launch"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + forwarded
    if args.dry_launch:
        print(" ".join(cmd))
        return 0
    import torch   # (counting devices does not initialise the GPU)
    visible = torch.cuda.device_count()
    if visible < args.gpus:
        print(f"bench.py: --gpus {args.gpus} asked, {visible} device(s) visible: refusing to measure fewer GPUs than asked", file=sys.stderr)
        return 2
    return subprocess.run(cmd).returncode


def resident_bytes():
    """What counts against that limit and cannot be reclaimed: the cgroup's anonymous + shared memory when memory.stat is readable, else this process's VmRSS."""
    try:
        stat = dict(line.split() for line in open("/sys/fs/cgroup/memory.stat"))
        return int(stat["anon"]) + int(stat.get("shmem", 0))
    except (OSError, KeyError, ValueError):
        pass
    try:
        for line in open("/proc/self/status"):
            if line.startswith("VmRSS:"):
                return int(line.split()[1]) * 1024
    except OSError:
        pass
    return 0


def note_memory(what):
    """GC_DEBUG_TIMES: one line per set-up stage with the host memory in use (what bounds config 5's graph size)."""
    if os.environ.get("GC_DEBUG_TIMES"):
        sys.stderr.write(f"[bench] {what}: {resident_bytes() / 2**30:.1f} GiB of host memory resident\n")
        sys.stderr.flush()


def start_host_memory_watchdog(cap_gb):
    """A thread that ends the process (exit 3, one line on stderr) when resident memory passes the cap: the kernel's OOM kill takes more than the process with it on a shared box."""
    if cap_gb <= 0:
        return None
    cap = int(cap_gb * 2**30)
    state = {"peak": 0}

    def watch():
        while True:
            now = resident_bytes()
            state["peak"] = max(state["peak"], now)
            if now > cap:
                sys.stderr.write(f"[bench] host memory {now / 2**30:.1f} GiB passed the cap of {cap / 2**30:.1f} GiB: stopping\n")
                sys.stderr.flush()
                os._exit(3)
            time.sleep(0.25)

    threading.Thread(target=watch, daemon=True).start()
    state["cap_gb"] = round(cap / 2**30, 1)
    return state


def main():
    args = parse_args()
    rc = launch_ranks(args, sys.argv[1:])
    if rc is not None:
        sys.exit(rc)
    memory_watch = start_host_memory_watchdog(args.host_memory_cap_gb)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # before anything touches HIP: a batch in flight uses a dozen streams, five batches share the device
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    long_pass = not args.no_long_pass

    from graphchainer_amd.synth import SynthGenome, SynthGraph
    t0 = time.time()
    tmp = tempfile.mkdtemp(prefix="gcbench_")
    gfa = os.path.join(tmp, "graph.gfa")
    strong = args.strong and world > 1
    # --setup-dir: GFA, reads and index cache kept between runs of the same workload (one rank; nothing of a run's RESULTS is kept)
    setup_key = {"config": args.config, "chromosomes": args.chromosomes, "backbone": args.backbone, "reads": args.reads, "read_len": args.read_len, "sv_fraction": args.sv_fraction,
                 # (what the generators below are called with, and a version of this layout: a change of either must not find a stale graph, reads or index - ADVICE r5)
                 "generator": "SynthGenome(seed=7, multi_allelic=0.1, nested=0.1, minus_links=0.3, repeats=4, repeat_len=3000) / SynthGraph(seed=7); setup format 2"}
    setup_reused = False
    if args.setup_dir and world == 1:
        os.makedirs(args.setup_dir, exist_ok=True)
        gfa = os.path.join(args.setup_dir, "graph.gfa")
        try:
            setup_reused = json.load(open(os.path.join(args.setup_dir, "key.json"))) == setup_key and os.path.exists(os.path.join(args.setup_dir, "index.gcidx"))
        except (OSError, ValueError):
            setup_reused = False
    if setup_reused:
        blob = np.load(os.path.join(args.setup_dir, "reads.npy"))
        offs = np.load(os.path.join(args.setup_dir, "read_offsets.npy"))
        reads = [blob[offs[i]:offs[i + 1]].tobytes() for i in range(len(offs) - 1)]
        sg = None
    elif args.config == 5:
        # several chromosomes in one GFA (the cross-component rule of src/AlignmentGraph.cpp:1722-1733), multi-allelic and nested sites (cover width > 2),
        # reverse-strand links, repeats; reads at PacBio-CLR-like error rates
        sg = SynthGenome(args.chromosomes, args.backbone, seed=7, multi_allelic=0.1, nested=0.1, minus_links=0.3, repeats=4, repeat_len=3000)
        sg.write_gfa(gfa)
        reads = sg.sample_reads(args.reads, args.read_len, seed=11 + (0 if strong else rank), p_del=0.04, p_sub=0.02, p_ins=0.09)
        sg = None                                          # (3.5 bytes per graph base: nothing below draws from it again)
    else:
        sg = SynthGraph(args.backbone, seed=7)
        if rank == 0:
            sg.write_gfa(gfa)                              # only rank 0 builds from the GFA; the others load its index cache
        # weak scaling: every rank draws its own reads; strong scaling: all ranks draw the same set and the work queue divides it
        reads = sg.sample_reads(args.reads, args.read_len, seed=11 + (0 if strong else rank), sv_fraction=args.sv_fraction)
    if args.setup_dir and world == 1 and not setup_reused:
        np.save(os.path.join(args.setup_dir, "reads.npy"), np.frombuffer(b"".join(reads), dtype=np.uint8))
        np.save(os.path.join(args.setup_dir, "read_offsets.npy"), np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.int64))
    t_gen = time.time() - t0
    note_memory("GFA written, reads drawn")
    # two short legs after the headline (N=1, config 2, outside `value`): reads whose chained alignment wins, and a graph with repeats
    legs = world == 1 and args.config == 2 and long_pass and not args.no_cpu_baseline and sg is not None
    sv_reads = rep_reads = rep_gfa = None
    # r6: the timed steps cycle through several read sets instead of re-aligning one (same shape, own seeds; each set's first reads go through the oracle too)
    if args.distinct_batches is None:
        args.distinct_batches = 4 if (args.config == 2 and sg is not None) else 1
    if sg is None or args.config != 2:
        args.distinct_batches = 1
    extra_sets = [sg.sample_reads(args.reads, args.read_len, seed=11 + 1000 * d + (0 if strong else rank), sv_fraction=args.sv_fraction) for d in range(1, args.distinct_batches)]
    args.extra_cpu_reads = {f"set{d + 1}": rs[:1000] for d, rs in enumerate(extra_sets)} if (rank == 0 and world == 1) else {}
    if legs and args.sv_leg_steps > 0:
        sv_reads = sg.sample_reads(args.reads, args.read_len, seed=13, sv_fraction=0.2)
        args.extra_cpu_reads["sv"] = sv_reads[:1000]
    if legs and args.repeats_leg_steps > 0:
        rep_sg = SynthGraph(max(2_000_000, args.backbone // 4), seed=9, repeats=600, repeat_len=3000)
        rep_gfa = os.path.join(tmp, "repeats.gfa")
        rep_sg.write_gfa(rep_gfa)
        rep_reads = rep_sg.sample_reads(args.reads, args.read_len, seed=17)
        args.extra_cpu_graphs = {"repeats": (rep_gfa, rep_reads[:1000])}
        del rep_sg

    cpu_baseline = cpu_summary = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline, cpu_summary = cpu_baseline_leg(args, gfa, reads, long_pass)   # before any HIP call of this process

    dist = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import torch
        import torch.distributed as dist_mod
        dist = dist_mod
        # reads shard embarrassingly: the only cross-rank traffic is the barrier, the max-over-ranks of the step time and (strong) the queue cursor
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    if world > 1 and "GC_HOST_THREADS" not in os.environ:
        # ranks share the host: split its cores between their worker pools (read by the library when it first loads)
        os.environ["GC_HOST_THREADS"] = str(max(4, min(96, 2 * usable_cpus() // world)))
    inflight_choice = None
    if world > 1:
        # ranks share the host's CPUs: the waits sleep between polls (GC_SPIN_SYNC=2, the library's default) and the number of batches in flight per GPU follows
        # from what a batch costs the host (r3: 0.41 CPU-s per 10 k x 10 kb batch) - unless the caller chose
        os.environ.setdefault("GC_SPIN_SYNC", "2")
        if "GC_BENCH_INFLIGHT" not in os.environ and not any(a.startswith("--inflight") for a in sys.argv[1:]):
            args.inflight, inflight_choice = choose_inflight(max(1, args.inflight), usable_cpus(), world)
    import graphchainer_amd as gca
    from graphchainer_amd.workqueue import SUMMARY_FIELDS, SUMMARY_WIDTH, ReadQueue, gaf_read_hashes, length_sorted_batches, read_summary, run_queue

    if gca.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    device = local_rank
    if device >= gca.device_count():
        # dry run of the multi-rank path on a box with fewer GPUs than ranks (not a measurement): ranks share devices
        if not os.environ.get("GC_BENCH_ALLOW_SHARED_GPU"):
            raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU of its own ({gca.device_count()} visible)")
        device = local_rank % gca.device_count()
    gca.set_device(device)
    if dist is not None:
        import torch
        torch.cuda.set_device(device)

    # Start-up: rank 0 builds the graph, MPC index and minimizer index from the GFA and writes the index cache;
    # the other ranks load that file instead of repeating the build. Rank 0 also loads it once, to report the load time.
    cache = os.path.join(tempfile.gettempdir(), f"gcbench_{os.environ.get('MASTER_PORT', 'single')}_{os.getuid()}.gcidx")
    if args.setup_dir and world == 1:
        cache = os.path.join(args.setup_dir, "index.gcidx")
    t_graph = t_index = t_save = t_load = 0.0
    cache_bytes = 0
    use_cache = not (args.no_index_cache and world == 1 and not setup_reused)
    built = None
    if rank == 0 and not setup_reused:
        t0 = time.time()
        graph = gca.AlignmentGraph(gfa)
        t_graph = time.time() - t0
        note_memory("graph built and uploaded")
        if not use_cache:
            graph.trim_host()   # before the index is built beside it (its hash table and the sorted pairs are ~20 B of host memory per graph base while they are made)
            note_memory("host's MPC copy released")
        t0 = time.time()
        seeder = gca.MinimizerSeeder(graph)
        t_index = time.time() - t0
        note_memory("minimizer index built")
        if use_cache:
            t0 = time.time()
            gca.api.save_index_cache(graph, seeder, cache)
            t_save = time.time() - t0
            cache_bytes = os.path.getsize(cache)
            # ONE host copy of graph and index at a time (r6): up to r5 the built objects stayed while the cache was loaded beside them to be compared, and that - 2 x 74 B of host
            # memory per graph base + the upload's staging copies - was the "175 B per base" that kept config 5 at 1.25 Gbp on a 300 GiB host. What the comparison needs is kept instead.
            positions = seeder.array("positions")
            built = (graph.NodeSize(), len(positions), zlib.crc32(positions.tobytes() if len(positions) < (1 << 27) else positions[:: max(1, len(positions) >> 26)].tobytes()))
            del positions
            seeder.close(); graph.close()
            graph = seeder = None
    if dist is not None:
        dist.barrier()
    if use_cache:
        t0 = time.time()
        graph, seeder = gca.api.load_index_cache(cache)
        t_load = time.time() - t0
        if built is not None:
            positions = seeder.array("positions")
            loaded = (graph.NodeSize(), len(positions), zlib.crc32(positions.tobytes() if len(positions) < (1 << 27) else positions[:: max(1, len(positions) >> 26)].tobytes()))
            del positions
            if loaded != built:
                raise SystemExit("index cache does not reproduce the built index")
    note_memory("before the host's MPC copy is released")
    graph.trim_host()   # nothing below writes the cache again: the host copy of the MPC index (33 B per graph base) goes back to the system
    note_memory("set-up done")
    if dist is not None:
        dist.barrier()
    if rank == 0 and use_cache and not (args.setup_dir and world == 1):
        os.remove(cache)
    if args.setup_dir and world == 1 and not setup_reused:
        json.dump(setup_key, open(os.path.join(args.setup_dir, "key.json"), "w"))

    inflight = max(1, args.inflight)
    mem_free_start, mem_total = gca.device_memory()
    # batches in flight that fit the device beside the graph and the index. A batch in flight holds (r6, measured with pools sized by use) 17 GB for 10 k x 10 kb on 51 Mbp,
    # and for 2 k x 50 kb: 28 GB on 192 Mbp, 41 on 960 Mbp, 46 on 1.25 Gbp, 81 on 3.1 Gbp - linear in the graph's size, because chance minimizer hits (and the fragment
    # extensions run from them) are: 245 + 183 x Gbp bytes per read base. Beside them the device's one whole-read scratch and the result blocks, ~20 GB. Streams whose
    # first batches do not fit after all are dropped one at a time in the warm-up below
    graph_gbp = (args.chromosomes if args.config == 5 else 1) * args.backbone / 1e9
    batch_bytes = int((245 + 183 * graph_gbp) * min(args.batch, args.reads) * args.read_len)
    fit = int((mem_free_start - (20 << 30)) // max(1, batch_bytes))
    memory_choice = None
    if fit < inflight:
        memory_choice = {"asked": inflight, "chosen": max(1, fit), "free_gb_after_graph_and_index": round(mem_free_start / 2**30, 1), "estimated_gb_per_batch_in_flight": round(batch_bytes / 2**30, 1)}
        inflight = max(1, fit)
    aligners = [gca.Aligner(graph, seeder, split_gap=args.split_gap, colinear_gap=args.colinear_gap, long_pass=long_pass) for _ in range(inflight)]
    # the rank's reads as length-sorted batches (one batch for config 2), uploaded before the timed region; the upload itself
    # (2-bit packing, reverse complement, match-mask bit vectors, PCIe) is timed here and reported beside the step time
    chunks = length_sorted_batches(reads, args.batch)
    t0 = time.perf_counter()
    batches = [gca.ReadBatch([reads[i] for i in idx]) for idx in chunks]
    upload_s = time.perf_counter() - t0
    total_bases = int(sum(int(b.lengths.sum()) for b in batches))
    queue = ReadQueue(len(batches), rank, world, dist if strong else None)
    # the read sets the timed steps cycle through, flat: set d's batches at [d * per_step, (d + 1) * per_step), each with the oracle's rows for ITS set's first reads
    per_step = len(batches)
    extra_summaries = getattr(args, "extra_cpu_summaries", None) or {}
    flat_chunks, flat_batches, flat_summaries = list(chunks), list(batches), [cpu_summary] * per_step
    for d, rs in enumerate(extra_sets):
        set_chunks = length_sorted_batches(rs, args.batch)
        flat_chunks += set_chunks
        flat_batches += [gca.ReadBatch([rs[i] for i in idx]) for idx in set_chunks]
        flat_summaries += [extra_summaries.get(f"set{d + 1}")] * len(set_chunks)
    n_sets = 1 + len(extra_sets)

    def sync():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    def run_steps(count, sets=None):
        """`count` passes over the rank's batches (step s: read set s mod `sets`): `inflight` host threads (one gc_stream each, like the reference's -t workers) pull
        (step, batch) items from one queue, so with more than one batch in flight consecutive steps overlap - a stream of batches."""
        sets = n_sets if sets is None else sets
        queue.reset(count * per_step)
        flat = lambda item: ((item // per_step) % sets) * per_step + item % per_step
        return run_queue(queue, lambda i, item: (flat(item), aligners[i].align_batch(flat_batches[flat(item)])), inflight)   # align_batch returns after its streams are drained

    def cpu_seconds():
        """CPU time of this container so far (cgroup v2 cpu.stat), or of this process when there is no such file."""
        try:
            for line in open("/sys/fs/cgroup/cpu.stat"):
                if line.startswith("usage_usec"):
                    return int(line.split()[1]) / 1e6
        except OSError:
            pass
        return time.process_time()

    # every stream in flight allocates its buffers (tens of GB of extension scratch, pinned staging) in its first batch: with W > 0 each of
    # them gets a warm-up batch, or the second stream's allocations would fall into the timed steps (W=1 with two streams measured 875 ms
    # per batch instead of 230)
    warmup_done = max(args.warmup, inflight) if args.warmup else 0
    if warmup_done:
        # static hand-out, outside the (possibly cross-rank) queue: stream i of this rank runs warm-up items i, i + inflight, ... so that
        # every stream of every rank has done its first-batch allocations before the timed steps, whatever the dynamic queue would do
        from concurrent.futures import ThreadPoolExecutor
        while True:
            # (config 5: two batches per stream - a stream's pools reach their size with its second batch, and at 3.1 Gbp, where ONE 81 GB batch fits beside 119 GB of graph
            # and index, a second stream got through a single warm-up batch and ran out of memory in the timed steps)
            n_items = max(args.warmup, (2 if args.config == 5 else 1) * inflight) * (1 if strong else len(batches))
            try:
                with ThreadPoolExecutor(max_workers=inflight) as warm:
                    list(warm.map(lambda i: [aligners[i].align_batch(batches[(item + rank) % len(batches)]) and None for item in range(i, n_items, inflight)], range(inflight)))
                break
            except RuntimeError as e:
                # the streams' first batches size their buffers: when they do not all fit the device beside the graph and the index, one batch fewer is kept in flight
                # (config 5 on a 960 Mbp graph: 31 GB of graph and index, 42 GB per 2 000 x 50 kb batch in flight, the 21 GB whole-read scratch)
                if "out of memory" not in str(e) or inflight <= 1:
                    raise
                for a in aligners:
                    a.close()
                gca.load_library().gc_result_cache_trim()
                memory_choice = {"asked": (memory_choice or {}).get("asked", inflight), "chosen": inflight - 1, "why": "the first batches of that many streams did not fit the device's memory"}
                inflight -= 1
                aligners[:] = [gca.Aligner(graph, seeder, split_gap=args.split_gap, colinear_gap=args.colinear_gap, long_pass=long_pass) for _ in range(inflight)]
    def thread_cpu():
        """{(tid, name): CPU seconds} of this process's threads (GC_DEBUG_TIMES: which threads the host CPU per step goes to)"""
        out = {}
        tick = os.sysconf("SC_CLK_TCK")
        for tid in os.listdir("/proc/self/task"):
            try:
                fields = open(f"/proc/self/task/{tid}/stat").read().rsplit(")", 1)
                name = fields[0].split("(", 1)[1]
                rest = fields[1].split()
                out[(int(tid), name)] = (int(rest[11]) + int(rest[12])) / tick
            except (OSError, IndexError, ValueError):
                pass
        return out

    sync()
    threads_start = thread_cpu() if os.environ.get("GC_DEBUG_TIMES") else None
    cpu_start = cpu_seconds()
    rank_cpu_start = time.process_time()
    t_start = time.perf_counter()
    outs = run_steps(args.steps)
    sync()
    elapsed = time.perf_counter() - t_start
    host_cpu_s = cpu_seconds() - cpu_start
    rank_cpu_s = time.process_time() - rank_cpu_start               # this rank's process alone (the cgroup figure above is the whole container's)
    if threads_start is not None:
        now = thread_cpu()
        used = sorted(((now[k] - threads_start.get(k, 0.0), k) for k in now), reverse=True)
        print(f"[bench cpu] pid {os.getpid()}; per step, by thread (s): " + ", ".join(f"{name}/{tid} {d / args.steps:.3f}" for d, (tid, name) in used[:14] if d > 0)
              + f"; all threads {sum(d for d, _ in used) / args.steps:.3f}, process {rank_cpu_s / args.steps:.3f}, container {host_cpu_s / args.steps:.3f}", file=sys.stderr)
    mem_free_end, _ = gca.device_memory()
    # what re-aligning ONE read set step after step (the bench up to r5) is worth: a few steps of it, timed the same way, outside `value`
    same_batch = None
    if n_sets > 1 and dist is None:
        k = max(1, args.steps)                             # (as many steps as the timed region: a short run would be mostly pipeline fill)
        sync()
        t_same = time.perf_counter()
        run_steps(k, sets=1)
        sync()
        same_batch = {"ms_per_step": round((time.perf_counter() - t_same) / k * 1e3, 2), "steps": k,
                      "what": "the same steps with ONE read set re-aligned every step (its graph lines stay in L2 / Infinity Cache between steps); `value` cycles through distinct_read_sets sets"}
    # Parity of the timed mode (src/Aligner.cpp:630-654,735,901-905): the reads the CPU leg aligned with the oracle are compared, value
    # for value, with what EVERY timed batch returned for them - chain, chain score, both NW distances, the decision, the whole-read
    # alignments and the selection. Outside the timed region; a mismatch fails the run.
    def summary_check(results, chunk_list, summary):
        """results: [(item, (batch index, out))] as run_queue returns them; summary: the oracle's rows for the first reads of the set the chunks index"""
        mismatches, checked, fields_bad = 0, 0, {}
        for _item, (b, out) in results:
            original = np.asarray(chunk_list[b], dtype=np.int64)
            rows_of = summary[b] if isinstance(summary, list) else summary      # (the timed steps: one summary per read set)
            if rows_of is None:
                continue
            rows = np.nonzero(original < len(rows_of))[0]
            if not len(rows):
                continue
            got = read_summary(out)[rows]
            bad = got != rows_of[original[rows], :SUMMARY_WIDTH]
            checked += len(rows)
            mismatches += int(bad.any(axis=1).sum())
            for k in np.nonzero(bad.any(axis=0))[0]:
                fields_bad[SUMMARY_FIELDS[k]] = fields_bad.get(SUMMARY_FIELDS[k], 0) + int(bad[:, k].sum())
        rec = {"reads": int(sum(len(x) for x in {id(x): x for x in summary if x is not None}.values())) if isinstance(summary, list) else int(len(summary)), "timed_batches_checked": len(results), "read_results_compared": checked, "mismatches": mismatches,
               "fields": "anchors, chain, chain score, whole-read and chain NW distance, chained_better, whole-read alignments (start, end, score), selection, failed_assertion, flatten ties (fragments, whole read)",
               "against": "oracle (CPU leg of this run), same reads"}
        if mismatches:
            rec["fields_with_mismatches"] = fields_bad
        return rec

    def gaf_check(text, out, chunk, summary):
        """(reads compared, reads whose GAF lines differ from the oracle's): the text gc_format_gaf returned for a batch against the 13th column of the CPU leg's summary"""
        original = np.asarray(chunk, dtype=np.int64)
        rows = np.nonzero(original < len(summary))[0]
        if not len(rows):
            return 0, 0
        hashes = gaf_read_hashes(text, np.diff(np.asarray(out["read_out_off"]).astype(np.int64)))
        return len(rows), int((hashes[rows] != summary[original[rows], SUMMARY_WIDTH]).sum())

    parity_check = summary_check(outs, flat_chunks, flat_summaries) if cpu_summary is not None else None
    failures = []
    if parity_check is not None and parity_check["mismatches"]:
        failures.append(f"parity check failed: {parity_check['mismatches']} of {parity_check['read_results_compared']} timed read results differ from the oracle")

    # End to end (outside the headline figure): host bases in, GAF text / GAM bytes out - gc_reads_upload, the hot path with the final alignments encoded on the
    # device (gc_params::device_output: no trace comes down), and gc_format_gaf / gc_format_gam of every batch inside the step, `inflight` batches overlapping
    # as in the timed region (src/Aligner.cpp:261-311). The GAF leg's text is compared with the oracle's, read by read (hash of its lines).
    e2e = None
    if world == 1 and args.e2e_steps > 0 and long_pass:
        e2e = {}
        import ctypes
        import queue as queue_mod
        # the read ids as the C arrays gc_format_* take (input, like the bases: built once)
        names = [(ctypes.c_char_p * len(idx))(*[f"read{i}".encode() for i in idx]) for idx in chunks]
        # host threads: as many streams as in the timed region, and three more threads than streams - a thread holds a stream only while gc_align_batch runs, so
        # one batch's upload and formatting overlap another's alignment (the reference keeps reader and writer threads beside its aligner threads, src/Aligner.cpp:1230-1300)
        e2e_workers = inflight + 3
        legs = [("gaf", "gaf", None)] if "gaf" in args.e2e_formats.split(",") else []
        if "gam" in args.e2e_formats.split(","):
            # (r6: gc_format_gam's own default is the device's LZ77 + Huffman deflate; level -1 = zlib's default on the host, the reference's setting)
            legs += [("gam", "gam", None), ("gam_zlib_default", "gam", -1), ("gam_level1", "gam", 1), ("gam_device_huffman", "gam", gca.GAM_DEVICE_HUFFMAN)]
        if "json" in args.e2e_formats.split(","):
            legs += [("json", "json", None)]                          # (r6; src/Aligner.cpp:286-293: one vg::Alignment per line through protobuf's JSON mapping)
        json_seen = {}
        gam_inflated = {}                                             # leg -> (bytes, CRC-32) of batch 0's inflated GAM stream: the same at every level
        for leg, fmt, level in legs:
            for a in aligners:
                a.params.device_output = 1 if fmt == "gaf" else 4   # (GAM and JSON: the vg::Path bytes of every final alignment)
            free_streams = queue_mod.SimpleQueue()
            for a in aligners:
                free_streams.put(a)
            kept, checking = {}, [True]

            spent = np.zeros(4)                                       # seconds in: upload, waiting for a stream, gc_align_batch, formatting (summed over the timed batches)

            def e2e_item(worker, item, leg=leg, fmt=fmt, level=level, kept=kept, checking=checking, free_streams=free_streams, spent=spent):
                b = item % len(chunks)
                t_a = time.perf_counter()
                batch = gca.ReadBatch([reads[i] for i in chunks[b]])
                t_b = time.perf_counter()
                aligner = free_streams.get()
                t_c = time.perf_counter()
                try:
                    out = aligner.align_batch(batch)
                finally:
                    free_streams.put(aligner)
                t_d = time.perf_counter()
                texts, skipped = aligner.format_batch(out, batch, names[b], formats=(fmt,), gam_level=level)
                t_e = time.perf_counter()
                if checking[0] and fmt == "gaf" and cpu_summary is not None:
                    kept[item] = gaf_check(texts["gaf"], out, chunks[b], cpu_summary)
                elif checking[0] and fmt == "json" and item == 0:
                    lines = texts["json"].split(b"\n")[:-1]
                    first = json.loads(lines[0]) if lines else {}
                    json_seen["lines"], json_seen["final_alignments"] = len(lines), int(np.asarray(out["read_out_off"])[-1])
                    json_seen["every_line_parses"] = all(isinstance(json.loads(l), dict) for l in lines[:2000])
                    json_seen["first_line_keys"] = sorted(first.keys())
                elif checking[0] and fmt == "gam" and item == 0:
                    import gzip
                    import zlib
                    inflated = gzip.decompress(texts["gam"])
                    gam_inflated[leg] = (len(inflated), zlib.crc32(inflated))
                elif not checking[0]:
                    spent += (t_b - t_a, t_c - t_b, t_d - t_c, t_e - t_d)   # (float adds under the GIL)
                batch.close()
                return len(texts[fmt]), skipped
            queue.reset(e2e_workers * len(chunks))
            run_queue(queue, e2e_item, e2e_workers)                   # first-batch allocations of the encoder's buffers; these batches' text is what the check reads
            checking[0] = False                                       # (the same batches overlapping as in the timed steps, the hashing kept out of the timing)
            queue.reset(args.e2e_steps * len(chunks))
            cpu0 = cpu_seconds()
            t0 = time.perf_counter()
            done = run_queue(queue, e2e_item, e2e_workers)
            dt = time.perf_counter() - t0
            rec = {"reads_per_s": round(args.e2e_steps * len(reads) / dt, 2), "ms_per_step": round(dt / args.e2e_steps * 1e3, 2), "steps": args.e2e_steps,
                   "host_cpu_s_per_step": round((cpu_seconds() - cpu0) / args.e2e_steps, 3), "host_threads": e2e_workers, "streams": inflight,
                   "bytes_per_step": int(sum(n for _i, (n, _s) in done) / args.e2e_steps), "chained_winners_without_trace": int(sum(s for _i, (_n, s) in done)),
                   "ms_per_batch_in": dict(zip(("upload", "waiting_for_a_stream", "gc_align_batch", "format"), (np.round(spent / max(1, len(done)) * 1e3, 1)).tolist())),
                   "includes": f"gc_reads_upload (PCIe + packing kernels) + hot path + output encoding on the device (k_out_encode) + gc_format_{fmt}" + (f"_level(level {level})" if level is not None else "") + " of every batch"}
            if fmt == "json":
                rec["json_check"] = dict(json_seen, what="first batch: one line per final alignment, every line a JSON object (the GPU tests hold the lines to the reference-decoded fixtures: tests/test_gpu_parity.py)")
                if json_seen.get("lines") != json_seen.get("final_alignments") or not json_seen.get("every_line_parses"):
                    failures.append(f"e2e json: {json_seen}")
            if fmt == "gam":
                if level == gca.GAM_DEVICE_HUFFMAN:
                    rec["gzip"] = "deflated on the device (hip/gc_deflate.hip: one dynamic-Huffman block of literals per read, no LZ77 matches); the host frames the members and computes their CRC-32s"
                elif level is None:
                    rec["gzip"] = "gc_format_gam's default (r6): deflated on the device with LZ77 matches in front of the Huffman stage (GC_GAM_DEVICE_LZ, hip/gc_deflate.hip); the host frames the members and computes their CRC-32s"
                else:
                    rec["gzip"] = "zlib level " + ("default (6), as the reference's GzipOutputStream" if level == -1 else str(level)) + ": deflate is host work the reference pays too, ~1 ms of CPU per 10 kb read at the default level"
                if leg in gam_inflated:
                    rec["inflated_bytes_batch0"], rec["inflated_crc32_batch0"] = gam_inflated[leg]
            if fmt == "gaf" and kept:
                compared, bad = sum(c for c, _ in kept.values()), sum(m for _, m in kept.values())
                rec["gaf_check"] = {"reads_compared": compared, "reads_with_different_lines": bad, "against": "the oracle's GAF lines of the same reads (hash per read, CPU leg of this run)"}
                if bad:
                    failures.append(f"end-to-end GAF check failed: the lines of {bad} of {compared} reads differ from the oracle's")
            e2e[leg] = rec
        if len(gam_inflated) > 1:
            same = len(set(gam_inflated.values())) == 1
            e2e["gam_check"] = {"legs": sorted(gam_inflated), "same_inflated_stream": same, "what": "batch 0's GAM bytes of every leg inflated with zlib: length and CRC-32 of the stream"}
            if not same:
                failures.append("end-to-end GAM check failed: the legs' inflated streams differ")
        for a in aligners:
            a.params.device_output = 0
        if "gaf" in e2e:                                              # (the r3 line's keys, for the GAF leg)
            e2e.update({k: e2e["gaf"][k] for k in ("reads_per_s", "ms_per_step", "steps", "host_cpu_s_per_step")})

    # The chained branch under the driver: the same graph, reads of which 20 % carry a 1.5 kb deletion - k_edit_path, the chained traces and the
    # winners' output run - a few steps with every batch compared with the oracle's summary of the first 1 000 reads, then one batch written as GAF and compared line by line
    def side_leg(leg_reads, leg_aligners, summary, steps, label):
        leg_chunks = length_sorted_batches(leg_reads, args.batch)
        leg_batches = [gca.ReadBatch([leg_reads[i] for i in idx]) for idx in leg_chunks]
        n_streams = len(leg_aligners)
        leg_queue = ReadQueue(len(leg_batches), 0, 1, None)
        run = lambda count: (leg_queue.reset(count * len(leg_batches)), run_queue(leg_queue, lambda i, item: (item % len(leg_batches), leg_aligners[i].align_batch(leg_batches[item % len(leg_batches)])), n_streams))[1]
        run(n_streams)                                                # every stream once: buffers sized for these reads
        t0 = time.perf_counter()
        leg_outs = run(steps)
        dt = time.perf_counter() - t0
        rec = {"reads_per_s": round(steps * len(leg_reads) / dt, 2), "ms_per_step": round(dt / steps * 1e3, 2), "steps": steps, "batches_in_flight": n_streams,
               "chained_better": int(sum(int(np.sum(o["chained_better"])) for _i, (_b, o) in leg_outs) / steps),
               "seeds_extended_per_read_fragment_pass": round(float(np.mean(np.concatenate([np.asarray(o["seeds_extended"], dtype=np.float64) for _i, (_b, o) in leg_outs]))), 1),
               "parity_check": summary_check(leg_outs, leg_chunks, summary)}
        if rec["parity_check"]["mismatches"]:
            failures.append(f"{label}: {rec['parity_check']['mismatches']} read results differ from the oracle")
        leg_aligners[0].params.device_output = 1
        out = leg_aligners[0].align_batch(leg_batches[0], gaf_names=[f"read{i}" for i in leg_chunks[0]], formats=("gaf",))
        leg_aligners[0].params.device_output = 0
        compared, bad = gaf_check(out["gaf"], out, leg_chunks[0], summary)
        rec["gaf_check"] = {"reads_compared": compared, "reads_with_different_lines": bad, "chained_winners_without_trace": int(out["gaf_chained_skipped"])}
        if bad:
            failures.append(f"{label}: the GAF lines of {bad} of {compared} reads differ from the oracle's")
        for b in leg_batches:
            b.close()
        leg_queue.close()
        return rec

    extra = getattr(args, "extra_cpu_summaries", {})
    sv_leg = repeats_leg = None
    if sv_reads is not None and "sv" in extra:
        sv_leg = side_leg(sv_reads, aligners, extra["sv"], args.sv_leg_steps, "sv leg")
        sv_leg["workload"] = f"{len(sv_reads)} x {args.read_len} bp reads on the same graph, 20 % with a 1.5 kb deletion the graph does not hold"
    kernel_us = np.zeros(8)
    host_us = np.zeros(4)
    counters = np.zeros(8, dtype=np.float64)
    counters_long = np.zeros(8, dtype=np.float64)
    reads_done = aligned_bases = chained_better = reads_with_chain = reads_with_long = 0
    long_ed, chain_ed, seeds_ext_long = [], [], []
    tie_reads = tie_reads_output = tie_extensions = 0
    for _item, (b, out) in outs:
        kernel_us += out["kernel_us"]
        host_us += out["host_us"]
        counters += out["counters"].astype(np.float64)
        counters_long += out["counters_long"].astype(np.float64)
        chain_len = np.diff(out["read_chain_off"])
        n_long = np.diff(out["read_longall_off"])
        reads_done += len(chain_len)
        aligned_bases += int(flat_batches[b].lengths[(chain_len > 0) | (n_long > 0)].sum())
        chained_better += int(np.sum(out["chained_better"]))
        reads_with_chain += int((chain_len > 0).sum())
        reads_with_long += int((n_long > 0).sum())
        long_ed.append(out["long_edit_distance"][out["long_edit_distance"] >= 0].astype(np.float64))
        chain_ed.append(out["chain_edit_distance"][out["chain_edit_distance"] >= 0].astype(np.float64))
        seeds_ext_long.append(np.asarray(out["seeds_extended_long"], dtype=np.float64))
        # the one rule this build defines instead of reproducing (flattenLastSliceEnd's tie order): reads that met it at all, and reads whose OUTPUT can depend on it -
        # a tie in the whole-read pass when the whole-read alignments are the output, a tie among the fragments' extensions when the chained alignment is
        ties_frag, ties_long = np.asarray(out["flatten_ties"], dtype=np.int64), np.asarray(out["flatten_ties_long"], dtype=np.int64)
        tie_reads += int(((ties_frag + ties_long) > 0).sum())
        tie_reads_output += int(np.where(np.asarray(out["chained_better"]) > 0, ties_frag > 0, ties_long > 0).sum())
        tie_extensions += int(ties_frag.sum() + ties_long.sum())
    per_rank = None
    if dist is not None:
        mine = {"rank": rank, "ms_per_step": round(elapsed / max(1, args.steps) * 1e3, 2), "host_cpu_s_per_step": round(rank_cpu_s / max(1, args.steps), 3),
                "reads": int(reads_done), "batches_in_flight": inflight}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        per_rank = gathered                                         # (a host-bound curve shows here: CPU seconds per step against the step time, rank by rank)
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([reads_done, aligned_bases], dtype=torch.float64)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        reads_total, aligned_total = float(tot[0].item()), float(tot[1].item())
    else:
        reads_total, aligned_total = float(reads_done), float(aligned_bases)
    steps = max(1, args.steps)
    kernel_us /= steps
    host_us /= steps
    counters /= steps
    counters_long /= steps
    reads_per_s = reads_total / elapsed
    gbp_per_s = aligned_total / elapsed / 1e9

    # roofline of the dominant kernel (most device time per step: k_long_extend when the whole-read pass runs, else k_extend):
    # algorithmic bytes per launch (SURVEY.md §8d unit x the counts the kernel reports) / its HIP-event duration
    def kernel_roofline(name, cnt, us, launches=1.0, reference_share=1.0):
        # `us` = the kernel's HIP-event time summed over its launches of one step (k_long_extend: one launch per round and batch)
        # reference_share (r6): the part of the kernel's extensions the reference would run - the whole-read pass extends some seeds speculatively (late rounds; on
        # config 5 a third more than the reference's count), and work the reference does not do is not algorithmic work: its bytes are left out of `achieved`
        dp_tiles, recompute_tiles, column_steps, trace_items, _ext, backtrace_tiles = cnt[:6]
        nbytes = (BYTES_PER_TILE * (dp_tiles + recompute_tiles) + BYTES_PER_BACKTRACE_TILE * backtrace_tiles + BYTES_PER_TRACE_ITEM * trace_items) * reference_share
        seconds = us * 1e-6
        achieved = nbytes / seconds / 1e9 if seconds > 0 else 0.0
        launches = max(1.0, launches)
        return {"bound": "hbm", "kernel": name, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
                "launches_per_step": round(launches, 2),
                "algorithmic_bytes_per_launch": int(nbytes / launches), "avg_launch_ms": round(us / 1e3 / launches, 3),
                "tiles_per_launch": int((dp_tiles + recompute_tiles) / launches), "column_steps_per_s_G": round(column_steps / seconds / 1e9, 3) if seconds > 0 else 0.0}

    def measured_traffic(kernel, launches):
        """HBM bytes per launch from the newest committed PMC pass (profiles/rNN_pmc_traffic.json: rocprofv3 FETCH_SIZE + WRITE_SIZE,
        separate passes over this same bench command); None when no profile is committed."""
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
        if not files:
            return None, None
        with open(files[-1]) as f:
            prof = json.load(f)
        for name, rec in prof.get("kernels", {}).items():
            if name.split("<")[0] == kernel and "FETCH_SIZE_KB_per_step" in rec and "WRITE_SIZE_KB_per_step" in rec:
                # (by the pass's own launch count: an r6 pass over `--steps 1` holds two batches - the timed step and the same-read-set leg)
                return int((rec["FETCH_SIZE_KB_per_step"] + rec["WRITE_SIZE_KB_per_step"]) * 1024 / max(1.0, float(rec.get("launches_per_step", launches)))), os.path.basename(files[-1])
        return None, None

    n_batches_step = len(outs) / steps
    roof_extend = kernel_roofline("k_extend", counters, kernel_us[1], n_batches_step)
    # (two extensions - backward, forward - per seed the reference extends: gc_result::seeds_extended_long)
    long_reference_share = min(1.0, 2.0 * float(sum(float(x.sum()) for x in seeds_ext_long)) / steps / max(1.0, counters_long[4])) if long_pass else 1.0
    roof_long = kernel_roofline("k_long_extend", counters_long, kernel_us[4], counters_long[6], long_reference_share) if long_pass else None
    if roof_long is not None:
        roof_long["extensions_the_reference_runs_share"] = round(long_reference_share, 4)
    roofline = roof_long if (long_pass and kernel_us[4] >= kernel_us[1]) else roof_extend
    if args.config == 2 and args.reads == 10_000:
        roofline["traffic"], roofline["traffic_source"] = measured_traffic(roofline["kernel"], roofline["launches_per_step"])

    graph_nodes = graph.NodeSize()
    if rep_reads is not None and "repeats" in extra:
        # a second graph (pasted repeats: several seeds per fragment window, clusters to order): this run's streams and graph go first - HBM holds one set
        for a in aligners:
            a.close()
        for b in flat_batches:
            b.close()
        seeder.close()
        graph.close()
        t0 = time.time()
        rep_graph = gca.AlignmentGraph(rep_gfa)
        rep_seeder = gca.MinimizerSeeder(rep_graph)
        rep_setup = time.time() - t0
        rep_aligners = [gca.Aligner(rep_graph, rep_seeder, split_gap=args.split_gap, colinear_gap=args.colinear_gap, long_pass=long_pass) for _ in range(min(inflight, 3))]
        repeats_leg = side_leg(rep_reads, rep_aligners, extra["repeats"], args.repeats_leg_steps, "repeats leg")
        repeats_leg["workload"] = f"{len(rep_reads)} x {args.read_len} bp reads on a {max(2_000_000, args.backbone // 4)} bp backbone with 600 pasted 3 kb repeats ({rep_graph.NodeSize()} split nodes); set-up {rep_setup:.1f} s"
        for a in rep_aligners:
            a.close()
        rep_seeder.close()
        rep_graph.close()

    if rank == 0:
        cat = lambda parts: np.concatenate(parts) if parts else np.zeros(0)
        long_ed, chain_ed, seeds_ext_long = cat(long_ed), cat(chain_ed), cat(seeds_ext_long)
        line = {
            "metric": "reads_per_sec", "value": round(reads_per_s, 2), "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": warmup_done,
            "ms_per_step": round(elapsed / steps * 1e3, 2), "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "gbp_per_sec_aligned": round(gbp_per_s, 5),
            "config": {"workload": (f"BASELINE configs[4] sized for one GPU: {args.chromosomes} chromosome graphs x {args.backbone} bp ({graph_nodes} split nodes, {2 * args.chromosomes} components), CLR-like errors, colinear_gap {args.colinear_gap}, " if args.config == 5 else
                                    f"BASELINE configs[{args.config - 1}]: chr22-like synthetic DAG ({args.backbone} bp backbone, {graph_nodes} split nodes), ")
                                   + f"{args.reads} x {args.read_len} bp {'CLR' if args.config == 5 else 'ONT'}-like reads {'in total' if strong else 'per GPU'} in batches of {args.batch}, split_len 35 split_gap {args.split_gap} bandwidth 10"
                                   + (f", {args.sv_fraction:.0%} of the reads with a 1.5 kb deletion" if args.sv_fraction > 0 else ""),
                       "stages": ("whole-read GraphAligner pass + selection + " if long_pass else "") + "seed lookup + seed ordering + fragment seed-extension + anchors + co-linear chaining + chain stitching + NW edit distances + chained-vs-whole-read decision + chained alignment trace (edlib path) for its winners",
                       "reads_per_gpu": args.reads if not strong else None, "read_len": args.read_len, "batch": args.batch, "batches_in_flight_per_gpu": inflight,
                       "parallelism": f"read-sharded x{world} ({'dynamic work queue over one read set' if strong else 'own reads per rank'}), graph replicated, no collective"},
            "roofline": roofline,
            "roofline_other": roof_extend if roofline is roof_long else roof_long,
            "cpu_baseline": cpu_baseline,
            "parity_check": parity_check,
            "unpinned_tie_reads": {"reads_with_a_tie_per_step": int(tie_reads / steps), "reads_whose_output_pass_had_a_tie_per_step": int(tie_reads_output / steps), "tied_extensions_per_step": int(tie_extensions / steps),
                                   "of_reads": int(reads_done / steps), "of_extensions": int(counters[4] + counters_long[4]),
                                   "what": "extensions whose backtrace started from a last-slice minimum attained in more than one node: the reference picks by parallel-hashmap iteration order (absent here), this build by band-entry order (DESIGN.md §7); compared with the oracle's count in parity_check"},
            "e2e": e2e,
            "sv_leg": sv_leg,
            "repeats_leg": repeats_leg,
            "inflight_choice": inflight_choice,
            "per_rank": per_rank,
            # inputs are resident before the timed region; what putting them there costs (host-side packing + PCIe), and the rate with it included
            "distinct_read_sets": n_sets, "same_read_set_every_step": same_batch,
            "reads_upload": {"ms_per_step": round(upload_s * 1e3, 2), "bases": total_bases, "reads_per_s_including_upload": round(reads_total / (elapsed + upload_s * steps * (1 if not strong else 1)), 2)},
            "host_cpu_s_per_step": round(host_cpu_s / max(1, args.steps), 3),   # CPU time the container spent per step (all threads, this rank's box)
            "stage_ms": {"k_seed_probe+compact": round(kernel_us[0] / 1e3, 3), "k_extend": round(kernel_us[1] / 1e3, 3), "k_build_anchors": round(kernel_us[2] / 1e3, 3),
                         "k_chain": round(kernel_us[3] / 1e3, 3), "k_long_extend_all_rounds": round(kernel_us[4] / 1e3, 3), "whole_read_pass_wall": round(kernel_us[5] / 1e3, 3), "seed_glue_wall": round(host_us[0] / 1e3, 3), "host_result_assembly": round(host_us[1] / 1e3, 3),
                         "wall_seed_lookup_and_copies": round(host_us[2] / 1e3, 3), "wall_extend_to_chain_and_copies": round(host_us[3] / 1e3, 3)},
            "device_memory_gb": {"total": round(mem_total / 2**30, 1), "graph_and_index": round((mem_total - mem_free_start) / 2**30, 1), "in_use_after_timed_steps": round((mem_total - mem_free_end) / 2**30, 1)},
            "setup_s": {"generate": round(t_gen, 1), "graph_build_upload": round(t_graph, 1), "minimizer_index": round(t_index, 1),
                        "index_cache_save": round(t_save, 1), "index_cache_load_upload": round(t_load, 1), "index_cache_bytes": cache_bytes or (os.path.getsize(cache) if os.path.exists(cache) else 0),
                        "reused_from_setup_dir": setup_reused},
            "inflight_for_device_memory": memory_choice,
            "host_memory_watch": memory_watch and {"cap_gb": memory_watch["cap_gb"], "peak_gb": round(memory_watch["peak"] / 2**30, 1), "what": "anonymous + shared memory of the cgroup (memory.stat), else VmRSS, sampled every 0.25 s"},
            "host_peak_rss_gb": round(__import__("resource").getrusage(__import__("resource").RUSAGE_SELF).ru_maxrss / 2**20, 1),   # (this process: the GFA in memory, the host graph and index, the batches)
            "reads_with_chain": int(reads_with_chain / steps), "extensions_per_step": int(counters[4]),
            "decision": {"chained_better": int(chained_better / steps), "mean_long_edit_distance": round(float(long_ed.mean()), 1) if len(long_ed) else None,
                         "mean_chain_edit_distance": round(float(chain_ed.mean()), 1) if len(chain_ed) else None},
            "long_pass": {"reads_with_alignment": int(reads_with_long / steps), "extensions_per_step": int(counters_long[4]), "rounds": int(counters_long[6]), "plain_layout_reruns": int(counters_long[7]),
                          "seeds_extended_mean": round(float(seeds_ext_long.mean()), 2) if len(seeds_ext_long) else None, "seeds_extended_max": int(seeds_ext_long.max()) if len(seeds_ext_long) else None} if long_pass else None,
        }
        print(json.dumps(line))
    queue.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if failures:
        raise SystemExit("; ".join(failures))


if __name__ == "__main__":
    main()
