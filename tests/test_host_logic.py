"""Host-side logic that needs no GPU: synthetic generator determinism, shard bounds, a 2-rank gloo run of
the read-sharded layout (each rank aligns its shard with the oracle; the union must equal the unsharded run)."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from graphchainer_amd.sharding import shard_bounds
from graphchainer_amd.synth import SynthGraph

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_partition():
    for n in (0, 1, 7, 10, 1001):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = shard_bounds(n, r, world)
                cover.extend(range(lo, hi))
            assert cover == list(range(n))
            sizes = [shard_bounds(n, r, world)[1] - shard_bounds(n, r, world)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_synthetic_generator_is_deterministic(tmp_path):
    a = SynthGraph(50_000, seed=7)
    b = SynthGraph(50_000, seed=7)
    pa, pb = str(tmp_path / "a.gfa"), str(tmp_path / "b.gfa")
    a.write_gfa(pa)
    b.write_gfa(pb)
    assert open(pa, "rb").read() == open(pb, "rb").read()
    ra, rb = a.sample_reads(5, 1000, seed=3), b.sample_reads(5, 1000, seed=3)
    assert ra == rb and all(len(r) == 1000 for r in ra)
    assert set(b"".join(ra)) <= set(b"ACGT")


def test_two_rank_gloo_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {ROOT!r})
        import numpy as np
        import torch.distributed as dist
        from graphchainer_amd.sharding import shard_bounds, sum_over_ranks, max_over_ranks
        from oracle import Oracle
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        gold = os.path.join({ROOT!r}, "tests", "golden")
        reads = [l.strip() for l in open(os.path.join(gold, "syn20k.fa")) if not l.startswith(">")]
        lo, hi = shard_bounds(len(reads), rank, world)
        res = Oracle(os.path.join(gold, "syn20k.gfa"), long_pass=False).align(reads[lo:hi])
        dist.barrier()
        total = sum_over_ranks(int(res["chain_score"].sum()), dist)
        n = sum_over_ranks(hi - lo, dist)
        slowest = max_over_ranks(float(rank + 1), dist)
        if rank == 0:
            want = np.load(os.path.join(gold, "syn20k.expected.npz"))
            assert total == int(want["chain_score"].sum()), (total, int(want["chain_score"].sum()))
            assert n == len(reads) and slowest == float(world)
            print("SHARD_OK")
        dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29571", str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert "SHARD_OK" in out.stdout, out.stdout + out.stderr


def test_two_rank_index_cache_handoff(tmp_path):
    """bench.py's start-up on N > 1 ranks: rank 0 builds the index cache on the host, the others wait at a barrier and take the
    file (here they verify it on the host; loading it needs a GPU and is covered by the gpu tests)."""
    script = tmp_path / "worker.py"
    cache = tmp_path / "shared.gcidx"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {ROOT!r})
        import torch.distributed as dist
        import graphchainer_amd as gca
        from graphchainer_amd.sharding import sum_over_ranks
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        gfa = os.path.join({ROOT!r}, "tests", "golden", "syn20k.gfa")
        if rank == 0:
            gca.api.build_index_cache(gfa, {str(cache)!r}, 15, 20)
        dist.barrier()
        info = gca.api.check_index_cache({str(cache)!r})
        nodes = sum_over_ranks(info["nodes"], dist)
        if rank == 0:
            assert nodes == world * info["nodes"] and info["has_seeder"] == 1
            print("CACHE_OK")
        dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29573")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29573", str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert "CACHE_OK" in out.stdout, out.stdout + out.stderr
