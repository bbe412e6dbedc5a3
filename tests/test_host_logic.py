"""Host-side logic that needs no GPU: synthetic generator determinism, the multi-GPU work queue (in one process, and across two
gloo ranks with the aligner mocked: the queue, the batch hand-out and the result merge are the product's own code paths)."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from graphchainer_amd.synth import SynthGraph
from graphchainer_amd.workqueue import ReadQueue, length_sorted_batches, merge_read_results, run_queue

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_length_sorted_batches_cover_every_read_once():
    import random
    rng = random.Random(3)
    for n, batch in ((0, 4), (1, 4), (10, 3), (1001, 64), (50, 100)):
        reads = [b"A" * rng.randint(1, 500) for _ in range(n)]
        batches = length_sorted_batches(reads, batch)
        flat = [i for b in batches for i in b]
        assert sorted(flat) == list(range(n)) and all(len(b) <= batch for b in batches)
        lens = [len(reads[i]) for i in flat]
        assert lens == sorted(lens, reverse=True)                       # longest first: a batch holds reads of similar length
    same = length_sorted_batches([b"ACGT"] * 9, 4)
    assert same == [[0, 1, 2, 3], [4, 5, 6, 7], [8]]                    # equal lengths keep their order (BASELINE config 2)


def test_queue_hands_every_batch_to_one_worker():
    reads = [bytes([65 + i % 4]) * (1 + i % 37) for i in range(500)]
    batches = length_sorted_batches(reads, 16)
    queue = ReadQueue(len(batches))
    for _ in range(3):                                                   # three "steps"
        queue.reset()
        parts = run_queue(queue, lambda worker, b: {"len": [len(reads[i]) for i in batches[b]], "worker": worker}, workers=4)
        assert sorted(b for b, _ in parts) == list(range(len(batches)))
        merged = merge_read_results(parts, batches, len(reads), "len", fill=-1)
        assert list(merged) == [len(r) for r in reads]
    assert queue.next() is None


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


def test_two_rank_gloo_work_queue(tmp_path):
    """BASELINE config 4's shape on two gloo ranks: ONE read set, the product's queue hands its length-sorted batches to whichever
    rank asks next (flock'ed counter file, no collective on the data path), every rank runs run_queue with two worker threads
    (gc_streams) and a mocked aligner, the per-read results merged over the ranks equal the unsharded run, over several steps."""
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys, time
        sys.path.insert(0, {ROOT!r})
        import numpy as np
        import torch
        import torch.distributed as dist
        from graphchainer_amd.workqueue import ReadQueue, length_sorted_batches, merge_read_results, run_queue
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        rng = np.random.default_rng(5)                                   # the same read set on every rank
        reads = [bytes(rng.integers(65, 69, size=int(n)).astype(np.uint8)) for n in rng.integers(20, 400, size=333)]
        batches = length_sorted_batches(reads, 16)
        queue = ReadQueue(len(batches), rank, world, dist, path={str(tmp_path / "queue.bin")!r})

        def mock_align(worker, b):                                       # stands in for Aligner.align_batch on a CPU-only box
            time.sleep(0.002 * (1 + rank))                               # rank 1 is slower: the queue gives it fewer batches
            return {{"score": [sum(reads[i]) % 1009 for i in batches[b]]}}
        for step in range(3):
            queue.reset()
            parts = run_queue(queue, mock_align, workers=2)
            mine = merge_read_results(parts, batches, len(reads), "score", fill=0)
            count = torch.tensor([len(parts)], dtype=torch.int64)
            merged = torch.from_numpy(mine)
            dist.all_reduce(count); dist.all_reduce(merged)
            got_batches = [None] * world
            dist.all_gather_object(got_batches, sorted(b for b, _ in parts))
            if rank == 0:
                assert int(count.item()) == len(batches), (int(count.item()), len(batches))
                assert sorted(b for part in got_batches for b in part) == list(range(len(batches)))      # each batch exactly once
                assert merged.tolist() == [sum(r) % 1009 for r in reads]
                assert len(got_batches[0]) > len(got_batches[1])           # dynamic: the faster rank took more
        queue.close()
        if rank == 0:
            print("QUEUE_OK")
        dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29571", str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert "QUEUE_OK" in out.stdout, out.stdout + out.stderr


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
        import torch
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        gfa = os.path.join({ROOT!r}, "tests", "golden", "syn20k.gfa")
        if rank == 0:
            gca.api.build_index_cache(gfa, {str(cache)!r}, 15, 20)
        dist.barrier()
        info = gca.api.check_index_cache({str(cache)!r})
        t = torch.tensor([int(info["nodes"])], dtype=torch.int64)
        dist.all_reduce(t)
        nodes = int(t.item())
        if rank == 0:
            assert nodes == world * info["nodes"] and info["has_seeder"] == 1
            print("CACHE_OK")
        dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29573")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29573", str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert "CACHE_OK" in out.stdout, out.stdout + out.stderr


def test_stdsort_clone_equals_libstdcxx(tmp_path):
    """The device replays the reference's three unstable std::sort calls with libstdc++'s own algorithm (graphchainer_amd/csrc/hip/gc_stdsort.hpp);
    tests/stdsort/stdsort_test.cpp compiles that header with g++ and compares it, permutation for permutation, with the local libstdc++ on
    20 000 arrays with many ties, on saw-tooth inputs of up to 100 000 elements, and - through libstdc++'s own __introsort_loop with a small depth
    limit - on the heapsort path."""
    exe = tmp_path / "stdsort_test"
    src = os.path.join(ROOT, "tests", "stdsort", "stdsort_test.cpp")
    subprocess.run(["g++", "-std=c++17", "-O2", src, "-o", str(exe)], check=True, timeout=300)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "STDSORT_OK" in out.stdout, out.stdout + out.stderr


def test_hash_order_replays_the_libstdcxx_containers(tmp_path):
    """gc::HashOrder (csrc/host/gc_hashorder.hpp) gives the iteration order of a libstdc++ unordered container from the sequence of its keys' hash values: the split-node numbering,
    the edge order and the minimizer index's node order all follow such an order (src/BigraphToDigraph.cpp:229,251, src/MinimizerSeeder.cpp:354-357). tests/hashorder/hashorder_test.cpp
    compares it with the real std::unordered_map for the three key types, 0 to 2 M keys (ascending, shuffled, strided, clustered, negative, the 2 id / 2 id + 1 pairs), and with erased keys."""
    exe = tmp_path / "hashorder_test"
    host = os.path.join(ROOT, "graphchainer_amd", "csrc", "host")
    subprocess.run(["g++", "-std=c++17", "-O2", "-I" + host, os.path.join(ROOT, "tests", "hashorder", "hashorder_test.cpp"), "-o", str(exe)], check=True, timeout=600)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.startswith("ok "), out.stdout + out.stderr


def test_encoder_letter_cursor_equals_the_graph_lookups(tmp_path):
    """The output encoders read the graph letter under every trace cell through gc::GraphLetters (csrc/host/gc_output.hpp), a cursor that remembers the last cell's split
    node; tests/output_host/letters_test.cpp compares it with GetUnitigNode + NodeSequences (what the reference does per cell, src/GraphAlignerCommon.h:148-153) on every
    letter of every original node - forwards, backwards, and in random jumps with node changes - of the golden graphs and of a graph with IUPAC letters and long nodes."""
    exe = tmp_path / "letters_test"
    host = os.path.join(ROOT, "graphchainer_amd", "csrc", "host")
    subprocess.run(["g++", "-std=c++17", "-O2", "-I" + host, os.path.join(ROOT, "tests", "output_host", "letters_test.cpp"), os.path.join(host, "gc_output.cpp"), os.path.join(host, "gc_graph.cpp"),
                    "-o", str(exe), "-lpthread", "-lz"], check=True, timeout=600)
    rng = np.random.default_rng(5)
    mixed = tmp_path / "iupac.gfa"
    segs = ["".join(rng.choice(list("ACGTACGTACGTNRYKM"), size=int(n))) for n in (1, 63, 64, 65, 200, 1000, 129, 5)]
    mixed.write_text("".join(f"S\t{i + 1}\t{s}\n" for i, s in enumerate(segs)) + "".join(f"L\t{i + 1}\t+\t{i + 2}\t+\t0M\n" for i in range(len(segs) - 1)))
    for gfa in (os.path.join(ROOT, "tests", "golden", "syn20k.gfa"), os.path.join(ROOT, "tests", "golden", "ref_test_graph.gfa"), str(mixed)):
        out = subprocess.run([str(exe), gfa], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and out.stdout.startswith("OK"), (gfa, out.stdout + out.stderr)


def test_gaf_and_json_encoders_equal_the_oracle_encoders_on_the_cpu(tmp_path):
    """The product's GAF encoder (gc::formatGafLine, csrc/host/gc_output.cpp: the reference's GraphAlignerGAFAlignment::traceToAlignment, src/GraphAlignerGAFAlignment.h:38-196) is
    host code: tests/output_host/gaf_test.cpp runs it on the CPU over the oracle's whole-read alignments (their traces are what gc_align_batch returns, array for array, in the GPU
    tests) and the text must equal the oracle's own encoder's - both CIGAR styles. (JSON and GAM share its trace walk through buildVgAlignment; their bytes are compared in the GPU tests.)"""
    from oracle import Oracle
    exe = tmp_path / "gaf_test"
    host = os.path.join(ROOT, "graphchainer_amd", "csrc", "host")
    subprocess.run(["g++", "-std=c++17", "-O2", "-I" + host, os.path.join(ROOT, "tests", "output_host", "gaf_test.cpp"), os.path.join(host, "gc_output.cpp"), os.path.join(host, "gc_graph.cpp"),
                    "-o", str(exe), "-lpthread", "-lz"], check=True, timeout=600)
    gold = os.path.join(ROOT, "tests", "golden")
    gfa = os.path.join(gold, "syn20k.gfa")
    reads = [l.strip() for l in open(os.path.join(gold, "syn20k.fa")) if not l.startswith(">")]
    oracle = Oracle(gfa, long_pass=True)
    w = oracle.align(reads)
    for merge in (0, 1, 2, 3, 4):       # 3 / 4 (r4): the GAF line / the vg message put together from pieces, as gc_format_* does for alignments the device encoded
        dump, expected_reads = [], 0
        for r, read in enumerate(reads):
            all_lo, all_hi = int(w["read_longall_off"][r]), int(w["read_longall_off"][r + 1])
            picked = []
            for k in range(int(w["read_long_off"][r]), int(w["read_long_off"][r + 1])):      # the selected alignments, found again in the read's full list
                key = (int(w["long_start"][k]), int(w["long_end"][k]), int(w["long_score"][k]))
                match = [a for a in range(all_lo, all_hi) if (int(w["longall_start"][a]), int(w["longall_end"][a]), int(w["longall_score"][a])) == key and a not in picked]
                assert match, (r, key)
                picked.append(match[0])
            if int(w["chained_better"][r]):
                continue                                                                      # (the chained alignment replaces them in the output; none on this fixture)
            picked.sort(key=lambda a: int(w["longall_start"][a]))                             # src/Aligner.cpp:1003,1023
            expected_reads += bool(picked)
            for a in picked:
                t0, t1 = int(w["long_trace_off"][a]), int(w["long_trace_off"][a + 1])
                rows = " ".join(f"{int(w['long_trace_node'][i])} {int(w['long_trace_offset'][i])} {int(w['long_trace_seqpos'][i])} {int(w['long_trace_switch'][i])}" for i in range(t0, t1))
                dump.append(f"r{r} {merge} {t1 - t0} {int(w['longall_score'][a])} {int(w['longall_start'][a])} {int(w['longall_end'][a])} {read}\n{rows}\n")
        path = tmp_path / f"alignments{merge}.txt"
        path.write_text("".join(dump))
        out = subprocess.run([str(exe), gfa, str(path)], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr
        want = oracle.json().decode() if merge in (2, 4) else oracle.gaf(merge=merge == 1).decode()
        assert expected_reads >= 5 and out.stdout == want, (merge, out.stdout[:200], want[:200])


def test_state_machine_extension_core_equals_oracle(tmp_path):
    """The per-lane state machine of the experimental kernel k_long_extend_sm (graphchainer_amd/csrc/hip/gc_sm_core.hpp) is plain C++:
    tests/sm_host/sm_host_test.cpp compiles its phase functions with g++, drives ONE lane on the CPU and compares status, score and every trace
    cell of several hundred extensions (10 kb ONT-like reads on a graph with repeats and multi-allelic sites, both directions, two band
    widths) with the oracle's getReverseTraceFromSeed."""
    exe = tmp_path / "sm_host_test"
    src = os.path.join(ROOT, "tests", "sm_host", "sm_host_test.cpp")
    host = os.path.join(ROOT, "graphchainer_amd", "csrc", "host")
    subprocess.run(["g++", "-std=c++17", "-O2", "-w", "-I/opt/rocm/include", src, os.path.join(host, "gc_graph.cpp"), os.path.join(host, "gc_minimizer.cpp"), "-o", str(exe), "-lpthread"], check=True, timeout=600)
    sg = SynthGraph(300_000, seed=43, repeats=4, repeat_len=2000, multi_allelic=0.1)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(10, 8000, seed=8) + sg.sample_reads(3, 8000, seed=9, sv_fraction=1.0)
    (tmp_path / "reads.txt").write_bytes(b"\n".join(reads) + b"\n")
    for args in (["4"], ["3", "5"]):
        out = subprocess.run([str(exe), gfa, str(tmp_path / "reads.txt")] + args, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and "SM_HOST_OK" in out.stdout, out.stdout + out.stderr
        fields = out.stdout.split()
        equal = int(fields[fields.index("equal") + 1])
        assert equal > 60, out.stdout


def test_fragment_extension_core_equals_oracle(tmp_path):
    """The per-lane state machine of the fragment extension kernel k_extend (graphchainer_amd/csrc/hip/gc_frag_core.hpp, r6) is plain C++ over a memory policy:
    tests/frag_host/frag_host_test.cpp compiles its phase functions with g++, drives ONE lane on the CPU over the kernel's own word layout (queue and ring overlaid,
    poisoned at the hand-over) and compares status, score, every trace cell, the tie flag and the work counters of tens of thousands of extensions of 1..64 rows
    (fragment-sized and longer, both directions, graphs with and without multi-allelic sites, two band widths) with the oracle's getReverseTraceFromSeed; an
    extension the core hands to the plain-layout kernel must have a reason (more than 64 rows, more tiles or pending nodes than its tables hold)."""
    exe = tmp_path / "frag_host_test"
    src = os.path.join(ROOT, "tests", "frag_host", "frag_host_test.cpp")
    host = os.path.join(ROOT, "graphchainer_amd", "csrc", "host")
    subprocess.run(["g++", "-std=c++17", "-O2", "-w", "-I/opt/rocm/include", src, os.path.join(host, "gc_graph.cpp"), os.path.join(host, "gc_minimizer.cpp"), "-o", str(exe), "-lpthread"], check=True, timeout=600)
    cases = [(SynthGraph(300_000, seed=43, repeats=4, repeat_len=2000, multi_allelic=0.1), ["300", "10"], 2000), (SynthGraph(400_000, seed=5), ["150", "5"], 3000)]
    for i, (sg, args, least) in enumerate(cases):
        gfa = str(tmp_path / f"g{i}.gfa")
        sg.write_gfa(gfa)
        reads = sg.sample_reads(12, 8000, seed=8 + i) + sg.sample_reads(3, 8000, seed=9, sv_fraction=1.0)
        (tmp_path / f"reads{i}.txt").write_bytes(b"\n".join(reads) + b"\n")
        out = subprocess.run([str(exe), gfa, str(tmp_path / f"reads{i}.txt")] + args, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and "FRAG_HOST_OK" in out.stdout, out.stdout + out.stderr
        fields = out.stdout.split()
        equal, declined, total = (int(fields[fields.index(k) + 1]) for k in ("equal", "declined", "extensions"))
        assert equal > least and declined < total // 8, out.stdout


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus N` run plainly must measure N GPUs or say why not (the flag was parsed and never read, so the plain command reported
    n_gpus 1). --dry-launch prints the child command - torch.distributed.run on 127.0.0.1 with the same arguments; without enough devices the real launch refuses
    with a non-zero exit code and no JSON line; under a launcher (WORLD_SIZE set) nothing is started."""
    bench = os.path.join(ROOT, "bench.py")
    out = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-launch", "--steps", "3", "--strong"], capture_output=True, text=True, timeout=300)
    words = out.stdout.split()
    assert out.returncode == 0 and "torch.distributed.run" in words and "--nproc-per-node=2" in words and "127.0.0.1" in words, out.stdout + out.stderr
    assert words[-5:] == ["--gpus", "2", "--steps", "3", "--strong"] and "--dry-launch" not in words[words.index(bench):]
    import torch
    if torch.cuda.device_count() < 2:
        out = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1"], capture_output=True, text=True, timeout=300)
        assert out.returncode != 0 and "refusing" in out.stderr and "n_gpus" not in out.stdout, out.stdout + out.stderr
    env = dict(os.environ, WORLD_SIZE="2")
    probe = "import sys, os; sys.path.insert(0, %r); sys.argv = ['bench.py', '--gpus', '2']; import bench; print(bench.launch_ranks(bench.parse_args(), sys.argv[1:]))" % ROOT
    out = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, timeout=300, env=env)
    assert out.stdout.strip() == "None", out.stdout + out.stderr


def test_bench_inflight_rule_follows_the_cpu_budget():
    """bench.py's batches-in-flight rule: from the host CPU a batch costs and the batch period, not from a fixed ranks-to-CPUs ratio.
    r5 (0.24 CPU-s per batch, 146 ms per batch): eight ranks on a 16-CPU box keep five batches in flight each (1.6 CPUs of the 2 a rank has; r3's 0.41 CPU-s made them
    host-bound); eight ranks on 8 CPUs are host-bound whatever they do and keep two (host work overlapping device work) instead of one; the record says why."""
    sys.path.insert(0, ROOT)
    import bench
    k, rec = bench.choose_inflight(5, 16, 2)
    assert k == 5 and rec["chosen"] == 5 and rec["cpus_per_rank"] == 8.0 and "5 in flight need" in rec["why"]
    k, rec = bench.choose_inflight(5, 16, 8)
    assert k == 5 and "host-bound" not in rec["why"]
    k, rec = bench.choose_inflight(5, 8, 8)
    assert k == 2 and "host-bound" in rec["why"]
    k, rec = bench.choose_inflight(5, 13, 8)          # 1.625 CPUs per rank: one or two in flight need 1.53 / 1.62, three 1.61, four 1.66
    assert k == 3 and "host-bound" not in rec["why"]
    k, rec = bench.choose_inflight(1, 256, 1)
    assert k == 1


def test_gaf_hash_of_the_bench_check_equals_the_oracles():
    """The end-to-end leg of bench.py compares, per read, a hash of the GAF lines the product wrote with the one the CPU leg's oracle run kept
    (gco_align_summary2). Here both sides are the oracle's own text: the numpy hash over the text blob must equal the C one, and a one-letter change in
    one CIGAR must change exactly that read's hash."""
    from graphchainer_amd.workqueue import gaf_read_hashes
    from oracle import Oracle
    gold = os.path.join(ROOT, "tests", "golden")
    reads = [l.strip() for l in open(os.path.join(gold, "syn20k.fa")) if not l.startswith(">")]
    oracle = Oracle(os.path.join(gold, "syn20k.gfa"), long_pass=True)
    w = oracle.align(reads)
    text = oracle.gaf(False)
    _, _, summary = oracle.align_summary(reads, 2, gaf_hash=True)
    lines = np.where(w["chained_better"] > 0, 1, np.diff(w["read_long_off"]))
    hashes = gaf_read_hashes(text, lines)
    assert summary.shape[1] == 15 and np.array_equal(hashes, summary[:, 14]) and len(set(hashes.tolist())) > len(reads) // 2
    # names do not enter: other read ids, same hashes
    renamed = b"\n".join(b"someothername" + l[l.index(b"\t"):] for l in text.split(b"\n") if l) + b"\n"
    assert np.array_equal(gaf_read_hashes(renamed, lines), hashes)
    at = text.index(b"cg:Z:") + 6
    changed = text[:at] + (b"9" if text[at:at + 1] != b"9" else b"8") + text[at + 1:]
    diff = np.nonzero(gaf_read_hashes(changed, lines) != hashes)[0]
    first_with_lines = int(np.nonzero(lines > 0)[0][0])
    assert diff.tolist() == [first_with_lines]


def test_genome_gfa_is_written_chromosome_by_chromosome_with_the_same_bytes(tmp_path):
    """SynthGenome.write_gfa keeps one chromosome's lines in memory at a time (config 5 at 3.1 Gbp); the file is what joining all S lines, then all L lines gives."""
    from graphchainer_amd.synth import SynthGenome
    sg = SynthGenome(3, 20_000, seed=7, multi_allelic=0.1, nested=0.1, minus_links=0.3, repeats=4, repeat_len=3000)
    path = str(tmp_path / "g.gfa")
    last = sg.write_gfa(path)
    segs, links, first = [], [], 1
    for chrom in sg.chromosomes:
        s, l, first = chrom.gfa_lines(first)
        segs += s
        links += l
    assert open(path, "rb").read() == b"".join(segs) + b"".join(links)
    assert last == first - 1 == len(segs) and not os.path.exists(path + ".links")


def test_bench_host_memory_cap_ends_the_process_itself():
    """bench.py --host-memory-cap-gb (config 5 at 3.1 Gbp was run with 290 on a 300 GiB host): a watchdog thread ends the process with exit code 3 and one line on stderr when
    resident memory passes the cap - before the kernel's OOM kill would; without the flag there is no watchdog."""
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "assert bench.start_host_memory_watchdog(0) is None\n"
            "state = bench.start_host_memory_watchdog(0.05)\n"
            "block = bytearray(200 << 20)\n"
            "time.sleep(5)\n"
            "print('still here')\n") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 3 and "passed the cap" in out.stderr and "still here" not in out.stdout, out.stdout + out.stderr
