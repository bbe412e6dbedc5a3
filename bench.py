#!/usr/bin/env python3
"""Benchmark of the MI355X GraphChainer hot path (BASELINE.json metric: reads/s and Gbp/s aligned).

One "step" = one pass of the hot path (whole-read GraphAligner pass; seed lookup -> seed ordering -> fragment
seed-extension -> anchors -> co-linear chaining -> chain stitching; NW edit distances and the chained-vs-whole-read
decision) over one batch of synthetic reads that is already resident in HBM.

Workload at N=1 = BASELINE.json configs[1]: chr22-like graph (50.8 Mbp backbone, SNP/indel bubbles every ~45 bp,
SURVEY.md §8d) and 10 000 simulated 10 kb ONT-like reads, reference defaults. For N>1 every rank aligns its own
10 000-read shard against its own replica of the graph (read-parallel, no data-path collective): weak scaling.

Usage: python bench.py [--gpus N] [--steps K] [--warmup W]
       (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)
"""
import argparse
import glob
import json
import os
import sys
import tempfile
import time

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
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--backbone", type=int, default=int(os.environ.get("GC_BENCH_BACKBONE", 50_800_000)))
    ap.add_argument("--reads", type=int, default=int(os.environ.get("GC_BENCH_READS", 10_000)))
    ap.add_argument("--read-len", type=int, default=10_000)
    ap.add_argument("--split-gap", type=int, default=35)
    ap.add_argument("--cpu-sample", type=int, default=int(os.environ.get("GC_BENCH_CPU_SAMPLE", 600)), help="reads of the same workload timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-long-pass", action="store_true", help="skip the whole-read GraphAligner pass (src/Aligner.cpp:630-654)")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("GC_BENCH_INFLIGHT", 1)),
                    help="batches in flight per GPU, each on its own gc_stream and host thread (like the reference's -t worker threads); every step still runs the whole hot path on the whole batch. Measured on cfg2: 1 -> 339, 2 -> 332, 3 -> 337 ms/step: the GPU is the bottleneck, so the default stays 1")
    return ap.parse_args()


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import torch
        import torch.distributed as dist_mod
        dist = dist_mod
        # reads shard embarrassingly: the only cross-rank traffic is the barrier and the max-over-ranks of the step time
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    if world > 1 and "GC_HOST_THREADS" not in os.environ:
        # ranks share the host: split its cores between their worker pools (read by the library when it first loads)
        os.environ["GC_HOST_THREADS"] = str(max(8, min(96, (os.cpu_count() or 8) // world)))
    import graphchainer_amd as gca
    from graphchainer_amd.synth import SynthGraph

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

    t0 = time.time()
    tmp = tempfile.mkdtemp(prefix="gcbench_")
    gfa = os.path.join(tmp, "graph.gfa")
    sg = SynthGraph(args.backbone, seed=7)
    sg.write_gfa(gfa)
    reads = sg.sample_reads(args.reads, args.read_len, seed=11 + rank)   # every rank draws its own shard
    t_gen = time.time() - t0
    # Start-up: rank 0 builds the graph, MPC index and minimizer index from the GFA and writes the index cache;
    # the other ranks load that file instead of repeating the build. Rank 0 also loads it once, to report the load time.
    cache = os.path.join(tempfile.gettempdir(), f"gcbench_{os.environ.get('MASTER_PORT', 'single')}_{os.getuid()}.gcidx")
    t_graph = t_index = t_save = t_load = 0.0
    if rank == 0:
        t0 = time.time()
        graph = gca.AlignmentGraph(gfa)
        t_graph = time.time() - t0
        t0 = time.time()
        seeder = gca.MinimizerSeeder(graph)
        t_index = time.time() - t0
        t0 = time.time()
        gca.api.save_index_cache(graph, seeder, cache)
        t_save = time.time() - t0
        cache_bytes = os.path.getsize(cache)
    if dist is not None:
        dist.barrier()
    t0 = time.time()
    loaded_graph, loaded_seeder = gca.api.load_index_cache(cache)
    t_load = time.time() - t0
    if rank == 0:
        if loaded_graph.NodeSize() != graph.NodeSize() or not np.array_equal(loaded_seeder.array("positions"), seeder.array("positions")):
            raise SystemExit("index cache does not reproduce the built index")
        loaded_seeder.close()
        loaded_graph.close()
    else:
        graph, seeder = loaded_graph, loaded_seeder
    if dist is not None:
        dist.barrier()
    if rank == 0:
        os.remove(cache)
    long_pass = not args.no_long_pass
    inflight = max(1, min(args.inflight, max(1, args.steps)))
    aligners = [gca.Aligner(graph, seeder, split_gap=args.split_gap, long_pass=long_pass) for _ in range(inflight)]
    batch = gca.ReadBatch(reads)          # inputs resident in HBM before the timed region
    total_bases = int(batch.lengths.sum())

    def sync():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=inflight)

    def run_steps(count):
        """`count` passes over the batch; worker i takes passes i, i+inflight, ... on its own stream. Returns the per-pass results."""
        def worker(i):
            return [aligners[i].align_batch(batch) for _ in range(i, count, inflight)]   # returns after its streams are drained
        outs = []
        for part in pool.map(worker, range(inflight)):
            outs.extend(part)
        return outs

    warmup_done = max(args.warmup, inflight if args.warmup else 0)   # every stream allocates its arenas before the timed region
    run_steps(warmup_done)
    sync()
    t_start = time.perf_counter()
    outs = run_steps(args.steps)
    sync()
    elapsed = time.perf_counter() - t_start
    kernel_us = np.zeros(8)
    host_us = np.zeros(4)
    counters = np.zeros(8, dtype=np.float64)
    counters_long = np.zeros(8, dtype=np.float64)
    for out in outs:
        kernel_us += out["kernel_us"]
        host_us += out["host_us"]
        counters += out["counters"].astype(np.float64)
        counters_long += out["counters_long"].astype(np.float64)
    out = outs[-1]
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_us /= max(1, args.steps)
    host_us /= max(1, args.steps)
    counters /= max(1, args.steps)
    counters_long /= max(1, args.steps)

    chain_len = np.diff(out["read_chain_off"])
    n_long = np.diff(out["read_longall_off"])
    aligned_bases = int(batch.lengths[(chain_len > 0) | (n_long > 0)].sum())
    reads_total = args.reads * world * args.steps
    reads_per_s = reads_total / elapsed
    gbp_per_s = aligned_bases * world * args.steps / elapsed / 1e9

    # roofline of the dominant kernel (most device time per step: k_long_extend when the whole-read pass runs, else k_extend):
    # algorithmic bytes per launch (SURVEY.md §8d unit x the counts the kernel reports) / its HIP-event duration
    def kernel_roofline(name, cnt, us, launches=1.0):
        # `us` = the kernel's HIP-event time summed over its launches of one step (k_long_extend: one launch per round)
        dp_tiles, recompute_tiles, column_steps, trace_items, _ext, backtrace_tiles = cnt[:6]
        nbytes = BYTES_PER_TILE * (dp_tiles + recompute_tiles) + BYTES_PER_BACKTRACE_TILE * backtrace_tiles + BYTES_PER_TRACE_ITEM * trace_items
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
            return None
        with open(files[-1]) as f:
            prof = json.load(f)
        for name, rec in prof.get("kernels", {}).items():
            if name.split("<")[0] == kernel and "FETCH_SIZE_KB_per_step" in rec and "WRITE_SIZE_KB_per_step" in rec:
                return int((rec["FETCH_SIZE_KB_per_step"] + rec["WRITE_SIZE_KB_per_step"]) * 1024 / max(1.0, launches))
        return None

    roof_extend = kernel_roofline("k_extend", counters, kernel_us[1])
    roof_long = kernel_roofline("k_long_extend", counters_long, kernel_us[4], counters_long[6]) if long_pass else None
    roofline = roof_long if (long_pass and kernel_us[4] >= kernel_us[1]) else roof_extend
    roofline["traffic"] = measured_traffic(roofline["kernel"], roofline["launches_per_step"])
    extensions = counters[4]

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import Oracle   # the CPU baseline leg is the one place bench.py may touch the oracle
        n_sample = min(args.cpu_sample, len(reads))
        ora = Oracle(gfa, long_pass=long_pass, split_gap=args.split_gap)
        t0 = time.perf_counter()
        ora.align(reads[:n_sample])
        cpu_t = time.perf_counter() - t0
        cpu_baseline = {"value": round(n_sample / cpu_t, 2), "unit": "reads/s", "cores": 1, "kind": "port",
                        "sample": f"first {n_sample} reads of the same workload, same stages ({'whole-read pass + selection, ' if long_pass else ''}seeding, fragment extension, anchors, chaining, chain stitching, NW edit distances), 1 thread, {cpu_t:.1f} s"}

    if rank == 0:
        line = {
            "metric": "reads_per_sec", "value": round(reads_per_s, 2), "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": warmup_done,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "gbp_per_sec_aligned": round(gbp_per_s, 5),
            "config": {"workload": f"BASELINE configs[1]: chr22-like synthetic DAG ({args.backbone} bp backbone, {graph.NodeSize()} split nodes), "
                                   f"{args.reads} x {args.read_len} bp ONT-like reads per GPU, split_len 35 split_gap {args.split_gap} bandwidth 10",
                       "stages": ("whole-read GraphAligner pass + selection + " if long_pass else "") + "seed lookup + seed ordering + fragment seed-extension + anchors + co-linear chaining + chain stitching + NW edit distances + chained-vs-whole-read decision",
                       "reads_per_gpu": args.reads, "read_len": args.read_len, "batches_in_flight_per_gpu": inflight, "parallelism": f"read-sharded x{world}, graph replicated, no collective"},
            "roofline": roofline,
            "roofline_other": roof_extend if roofline is roof_long else roof_long,
            "cpu_baseline": cpu_baseline,
            "stage_ms": {"k_seed_probe+compact": round(kernel_us[0] / 1e3, 3), "k_extend": round(kernel_us[1] / 1e3, 3), "k_build_anchors": round(kernel_us[2] / 1e3, 3),
                         "k_chain": round(kernel_us[3] / 1e3, 3), "k_long_extend_all_rounds": round(kernel_us[4] / 1e3, 3), "whole_read_pass_wall": round(kernel_us[5] / 1e3, 3), "host_seed_glue": round(host_us[0] / 1e3, 3), "host_result_assembly": round(host_us[1] / 1e3, 3),
                         "wall_seed_lookup_and_copies": round(host_us[2] / 1e3, 3), "wall_extend_to_chain_and_copies": round(host_us[3] / 1e3, 3)},
            "setup_s": {"generate": round(t_gen, 1), "graph_build_upload": round(t_graph, 1), "minimizer_index": round(t_index, 1),
                        "index_cache_save": round(t_save, 1), "index_cache_load_upload": round(t_load, 1), "index_cache_bytes": cache_bytes},
            "reads_with_chain": int((chain_len > 0).sum()), "extensions_per_step": int(extensions),
            "decision": {"chained_better": int(np.sum(out["chained_better"])), "mean_long_edit_distance": round(float(np.mean(out["long_edit_distance"][out["long_edit_distance"] >= 0])), 1) if long_pass and (out["long_edit_distance"] >= 0).any() else None,
                         "mean_chain_edit_distance": round(float(np.mean(out["chain_edit_distance"][out["chain_edit_distance"] >= 0])), 1) if (out["chain_edit_distance"] >= 0).any() else None},
            "long_pass": {"reads_with_alignment": int((n_long > 0).sum()), "extensions_per_step": int(counters_long[4]), "rounds": int(counters_long[6]), "plain_layout_reruns": int(counters_long[7]),
                          "seeds_extended_mean": round(float(out["seeds_extended_long"].mean()), 2), "seeds_extended_max": int(out["seeds_extended_long"].max())} if long_pass else None,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
