"""Index cache: host-side build / verify on CPU, and load parity on the GPU.

The reference has no cache (saveMPC/loadMPC are empty, src/AlignmentGraph.h:96-97), so the property tested is that a graph
and minimizer index loaded from the file are indistinguishable from the ones built from the GFA: same arrays, same
alignments, and the same minimizer index when it is rebuilt on the loaded graph."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
GRAPH_ARRAYS = ["nodeLength", "nodeOffset", "nodeIDs", "reverse", "componentNumber", "chainNumber", "chainApproxPos",
                "component_map", "out_off", "out_adj", "in_off", "in_adj", "mpc_width"]


@pytest.fixture(scope="module")
def gca():
    import graphchainer_amd as gca
    if not os.path.exists(gca.api.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    gca.load_library()
    return gca


@pytest.mark.parametrize("gfa", ["ref_test_graph.gfa", "syn20k.gfa"])
def test_build_is_deterministic_and_verifies(gca, tmp_path, gfa):
    a, b = str(tmp_path / "a.gcidx"), str(tmp_path / "b.gcidx")
    gca.api.build_index_cache(os.path.join(GOLD, gfa), a, 15, 20)
    gca.api.build_index_cache(os.path.join(GOLD, gfa), b, 15, 20)
    assert open(a, "rb").read() == open(b, "rb").read()
    info = gca.api.check_index_cache(a)
    assert info["version"] == 1 and info["has_seeder"] == 1 and (info["k"], info["w"]) == (15, 20)
    assert info["nodes"] > 0 and info["bp"] > 0 and info["positions"] >= info["kmers"] > 0
    # against the oracle's own graph build (same host code, separate library): node and base-pair counts
    from oracle import Oracle
    lengths = Oracle(os.path.join(GOLD, gfa)).graph_array("nodeLength")
    assert info["nodes"] == len(lengths) and info["bp"] == int(np.sum(lengths))


def test_graph_only_cache(gca, tmp_path):
    path = str(tmp_path / "g.gcidx")
    gca.api.build_index_cache(os.path.join(GOLD, "ref_test_graph.gfa"), path, 0, 0)
    info = gca.api.check_index_cache(path)
    assert info["has_seeder"] == 0 and info["kmers"] == 0


def test_damaged_files_are_refused(gca, tmp_path):
    path = str(tmp_path / "a.gcidx")
    gca.api.build_index_cache(os.path.join(GOLD, "ref_test_graph.gfa"), path, 15, 20)
    data = bytearray(open(path, "rb").read())
    cases = {
        "flip": bytes(data[:len(data) // 2]) + bytes([data[len(data) // 2] ^ 0x40]) + bytes(data[len(data) // 2 + 1:]),
        "truncated": bytes(data[:-9]),
        "magic": b"NOTCACHE" + bytes(data[8:]),
        "short": b"GCAMDIDX",
        "version": bytes(data[:8]) + bytes([data[8] + 1]) + bytes(data[9:]),
    }
    for name, blob in cases.items():
        bad = str(tmp_path / (name + ".gcidx"))
        open(bad, "wb").write(blob)
        with pytest.raises(RuntimeError):
            gca.api.check_index_cache(bad)
    with pytest.raises(RuntimeError):
        gca.api.check_index_cache(str(tmp_path / "missing.gcidx"))
    with pytest.raises(RuntimeError):
        gca.api.build_index_cache(os.path.join(GOLD, "ref_test_graph.gfa"), str(tmp_path / "k.gcidx"), 32, 40)   # (minimizer lengths run to 31, src/MinimizerSeeder.cpp:63)


def test_load_needs_a_device(gca, tmp_path):
    if gca.device_count() > 0:
        pytest.skip("a GPU is present")
    path = str(tmp_path / "a.gcidx")
    gca.api.build_index_cache(os.path.join(GOLD, "ref_test_graph.gfa"), path, 15, 20)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        gca.api.load_index_cache(path)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["ref_test", "syn20k"])
def test_loaded_index_matches_built_index(gca, tmp_path, case):
    from test_oracle_golden import read_fasta
    gfa = os.path.join(GOLD, "ref_test_graph.gfa" if case == "ref_test" else "syn20k.gfa")
    fasta = os.path.join(GOLD, "ref_test_read.fa" if case == "ref_test" else "syn20k.fa")
    built_graph = gca.AlignmentGraph(gfa)
    built_seeder = gca.MinimizerSeeder(built_graph, 15, 20)
    host_path, dev_path = str(tmp_path / "host.gcidx"), str(tmp_path / "dev.gcidx")
    gca.api.build_index_cache(gfa, host_path, 15, 20)      # host-only builder
    gca.api.save_index_cache(built_graph, built_seeder, dev_path)   # from the live objects
    assert open(host_path, "rb").read() == open(dev_path, "rb").read()
    graph, seeder = gca.api.load_index_cache(host_path)
    assert seeder is not None
    for name in GRAPH_ARRAYS:
        assert np.array_equal(graph.array(name), built_graph.array(name)), name
    for name in ["kmers", "start", "positions", "maxcount"]:
        assert np.array_equal(seeder.array(name), built_seeder.array(name)), name
    # a minimizer index rebuilt on the loaded graph follows the same node order as on the built one
    rebuilt = gca.MinimizerSeeder(graph, 15, 20)
    assert np.array_equal(rebuilt.array("positions"), built_seeder.array("positions"))
    # a saved loaded index is the same file again
    again = str(tmp_path / "again.gcidx")
    gca.api.save_index_cache(graph, seeder, again)
    assert open(again, "rb").read() == open(host_path, "rb").read()
    # and the alignments are the same, field for field
    reads = read_fasta(fasta)
    names = [f"r{i}" for i in range(len(reads))]
    results = []
    for g, s in ((built_graph, built_seeder), (graph, seeder)):
        aligner = gca.Aligner(g, s, long_pass=True, keep_traces=True)
        results.append(aligner.align_batch(gca.ReadBatch(reads), gaf_names=names))
    a, b = results
    assert a["gaf"] == b["gaf"]
    for key in a:
        if isinstance(a[key], np.ndarray) and key not in ("kernel_us", "host_us"):
            assert np.array_equal(a[key], b[key]), key
    # graph-only cache: the seeder comes back as None
    only = str(tmp_path / "only.gcidx")
    gca.api.save_index_cache(built_graph, None, only)
    g2, s2 = gca.api.load_index_cache(only)
    assert s2 is None and g2.NodeSize() == built_graph.NodeSize()


DESC_ARRAYS = ["nodeLength", "nodeOffset", "nodeIDs", "nodeSeq", "ambiguousSeq", "in_off", "in_adj", "out_off", "out_adj", "componentNumber",
               "chainNumber", "chainApproxPos", "lookupOrder", "firstAmbiguous"]


@pytest.mark.gpu
def test_graph_from_host_arrays_matches_graph_from_gfa(gca, tmp_path):
    """gc_graph_create (the entry point for a host that keeps its own AlignmentGraph): handing over the arrays of a graph gives
    the same MPC index, the same minimizer index (thanks to lookup_order) and the same alignments as building from the GFA.
    The graph carries IUPAC letters, so both sequence encodings are exercised."""
    from graphchainer_amd.synth import SynthGraph
    sg = SynthGraph(60_000, seed=23)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    lines = open(gfa).read().split("\n")
    touched = 0
    for i, line in enumerate(lines):
        if line.startswith("S\t") and touched < 10 and i % 5 == 0:
            f = line.split("\t")
            if len(f[2]) >= 20:
                seq = bytearray(f[2].encode())
                seq[len(seq) // 3] = ord("NRYKMSW"[touched % 7])
                f[2] = seq.decode()
                lines[i] = "\t".join(f)
                touched += 1
    assert touched > 0
    open(gfa, "w").write("\n".join(lines))
    reads = sg.sample_reads(10, 3000, seed=3)
    built = gca.AlignmentGraph(gfa)
    arrays = {name: built.array(name) for name in DESC_ARRAYS}
    assert int(arrays["firstAmbiguous"][0]) < built.NodeSize()
    handed = gca.AlignmentGraph.from_arrays(arrays)
    for name in GRAPH_ARRAYS + DESC_ARRAYS:
        assert np.array_equal(handed.array(name), built.array(name)), name
    built_seeder, handed_seeder = gca.MinimizerSeeder(built, 15, 20), gca.MinimizerSeeder(handed, 15, 20)
    for name in ["kmers", "start", "positions", "maxcount"]:
        assert np.array_equal(handed_seeder.array(name), built_seeder.array(name)), name
    results = []
    for g, s in ((built, built_seeder), (handed, handed_seeder)):
        results.append(gca.Aligner(g, s, long_pass=True, keep_traces=True).align_batch(gca.ReadBatch(reads)))
    a, b = results
    assert int(a["read_chain_off"][-1]) > 0
    for key in a:
        if isinstance(a[key], np.ndarray) and key not in ("kernel_us", "host_us"):
            assert np.array_equal(a[key], b[key]), key
    # without lookup_order the graph and the k-mer set are still the same; only the order inside position lists may differ
    plain = gca.AlignmentGraph.from_arrays(arrays, with_lookup_order=False)
    plain_seeder = gca.MinimizerSeeder(plain, 15, 20)
    assert np.array_equal(plain_seeder.array("kmers"), built_seeder.array("kmers"))
    assert np.array_equal(np.sort(plain_seeder.array("positions")), np.sort(built_seeder.array("positions")))
    # a malformed order is refused
    bad = dict(arrays)
    bad["lookupOrder"] = arrays["lookupOrder"][:-1]
    with pytest.raises(RuntimeError):
        gca.AlignmentGraph.from_arrays(bad)


def test_corrupt_payload_with_valid_checksum_never_crashes(gca, tmp_path):
    """The checksum stops accidental damage; behind it the parser bounds every count by the file size and the validator checks
    every stored index against the array it points into. Files with a mutated payload and a recomputed checksum must be
    refused or - when the mutation happens to be consistent - accepted, but never crash or hang the process. Run in a child
    process so that a crash is a test failure, not a dead test runner."""
    import subprocess
    import sys
    import textwrap
    path = str(tmp_path / "a.gcidx")
    gca.api.build_index_cache(os.path.join(GOLD, "ref_test_graph.gfa"), path, 15, 20)
    script = tmp_path / "fuzz.py"
    script.write_text(textwrap.dedent(f"""
        import random, sys
        sys.path.insert(0, {ROOT!r})
        import graphchainer_amd as gca
        data = bytearray(open({path!r}, "rb").read())
        def fnv(b):
            h = 0xcbf29ce484222325
            for x in b:
                h = ((h ^ x) * 0x100000001b3) & 0xffffffffffffffff
            return h
        assert fnv(data[:-8]) == int.from_bytes(data[-8:], "little")
        rng = random.Random(1)
        refused = accepted = 0
        for trial in range(300):
            blob = bytearray(data[:-8])
            kind = trial % 3
            at = rng.randrange(9, len(blob))
            if kind == 0:
                blob[at] ^= 1 << rng.randrange(8)
            elif kind == 1:
                blob[at] = rng.choice([0, 0x7f, 0x80, 0xff])
            else:
                del blob[at:at + rng.randrange(1, 9)]
            blob += fnv(blob).to_bytes(8, "little")
            open({str(tmp_path / "m.gcidx")!r}, "wb").write(blob)
            try:
                gca.api.check_index_cache({str(tmp_path / "m.gcidx")!r})
                accepted += 1
            except RuntimeError:
                refused += 1
        print("FUZZ_OK", refused, accepted)
    """))
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "FUZZ_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    refused, accepted = int(out.stdout.split()[-2]), int(out.stdout.split()[-1])
    # most mutations break a structural invariant; the rest change data no invariant covers (a base, a chain label, a k-mer that
    # stays sorted) - catching those is the checksum's job
    assert refused + accepted == 300 and refused >= 150


@pytest.mark.gpu
def test_accepted_mutated_caches_load_and_align_without_crashing(gca, tmp_path):
    """A mutated file that passes validation (recomputed checksum, consistent structure) is then USED: loaded, uploaded to the device and
    aligned against. The validator promises that nothing it accepts can index out of bounds later (twin lookups, minimizer positions,
    adjacency, path cover); here that promise is exercised end to end, in a child process so that a crash is a test failure."""
    import subprocess
    import sys
    import textwrap
    path = str(tmp_path / "a.gcidx")
    gfa = os.path.join(GOLD, "syn20k.gfa")
    gca.api.build_index_cache(gfa, path, 15, 20)
    reads = [l.strip() for l in open(os.path.join(GOLD, "syn20k.fa")) if not l.startswith(">")][:3]
    script = tmp_path / "fuzz_use.py"
    script.write_text(textwrap.dedent(f"""
        import random, sys
        sys.path.insert(0, {ROOT!r})
        import graphchainer_amd as gca
        data = bytearray(open({path!r}, "rb").read())
        def fnv(b):
            h = 0xcbf29ce484222325
            for x in b:
                h = ((h ^ x) * 0x100000001b3) & 0xffffffffffffffff
            return h
        rng = random.Random(7)
        reads = {[r.encode() for r in reads]!r}
        used = refused = 0
        for trial in range(120):
            blob = bytearray(data[:-8])
            at = rng.randrange(9, len(blob))
            if trial % 2 == 0:
                blob[at] ^= 1 << rng.randrange(8)
            else:
                blob[at] = rng.choice([0, 1, 0x3f, 0x40, 0x7f])
            blob += fnv(blob).to_bytes(8, "little")
            m = {str(tmp_path / "m.gcidx")!r}
            open(m, "wb").write(blob)
            try:
                graph, seeder = gca.api.load_index_cache(m)
            except RuntimeError:
                refused += 1
                continue
            try:
                out = gca.Aligner(graph, seeder, long_pass=True).align_reads(reads)   # may differ from the true result, must not crash
                assert len(out["read_chain_off"]) == len(reads) + 1
            except RuntimeError:
                pass                                                                  # a clean error is fine too
            used += 1
            seeder.close(); graph.close()
        print("FUZZ_USE_OK", used, refused)
    """))
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0 and "FUZZ_USE_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    assert int(out.stdout.split()[-2]) >= 5


def test_threaded_build_equals_serial_build(gca, tmp_path, monkeypatch):
    """The start-up builders run MPC components and minimizer node chunks on several threads; the cache they produce is
    byte-identical to the single-threaded one (a 1.5 Mbp graph: two components, ~130 k bigraph nodes, all threads used)."""
    from graphchainer_amd.synth import SynthGraph
    sg = SynthGraph(1_500_000, seed=29)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    monkeypatch.setenv("GC_BUILD_THREADS", "1")
    gca.api.build_index_cache(gfa, str(tmp_path / "serial.gcidx"), 15, 20)
    blobs = [open(str(tmp_path / "serial.gcidx"), "rb").read()]
    for threads in ("2", "5", "8"):
        monkeypatch.setenv("GC_BUILD_THREADS", threads)
        gca.api.build_index_cache(gfa, str(tmp_path / "threaded.gcidx"), 15, 20)
        blobs.append(open(str(tmp_path / "threaded.gcidx"), "rb").read())
    assert all(b == blobs[0] for b in blobs[1:])
    assert gca.api.check_index_cache(str(tmp_path / "serial.gcidx"))["kmers"] > 100_000


@pytest.mark.gpu
def test_device_built_minimizer_index_equals_host_built(gca, tmp_path, monkeypatch):
    """gc_seeder_create builds the minimizer index on the device (gc_minimizer.hip: window scan per bigraph node + one radix sort);
    gc_index_build and GC_SEEDER_BUILD=host run the host builder (host/gc_minimizer.cpp). Same k-mers, same lists in the same order,
    same frequency cutoff - over the golden graphs, graphs with IUPAC letters (windows restart after them), short nodes (< w), and
    window shapes from w = k to the longest deque the kernel holds; beyond that the library falls back to the host builder."""
    import random
    from test_graph_model import random_dag_gfa
    from graphchainer_amd.synth import SynthGenome
    paths = [os.path.join(GOLD, "ref_test_graph.gfa"), os.path.join(GOLD, "syn20k.gfa")]
    p = str(tmp_path / "genome.gfa")
    SynthGenome(2, 30000, seed=5, multi_allelic=0.3, nested=0.5, minus_links=0.4, repeats=3, repeat_len=500).write_gfa(p)
    paths.append(p)
    rng = random.Random(77)
    for i in range(4):
        p = str(tmp_path / f"rand{i}.gfa")
        open(p, "w").write(random_dag_gfa(rng, 150, iupac=True))
        paths.append(p)
    # one long segment with runs of IUPAC letters: restarts in the middle of a window, at the node's end, back to back
    seq = "".join(rng.choice("ACGT") for _ in range(5000))
    seq = bytearray(seq.encode())
    for at in (0, 1, 17, 18, 19, 40, 700, 701, 702, 760, 2000, 4979, 4999):
        seq[at] = ord("N")
    p = str(tmp_path / "longn.gfa")
    open(p, "w").write("S\t1\t" + seq.decode() + "\nS\t2\tACGTNACGTACGTACGTACGTACGATCGATCGACTAGCTAGCATCGACTAGCTACGACTAGCATCGACTAC\nL\t1\t+\t2\t+\t0M\n")
    paths.append(p)
    shapes = [(15, 20), (11, 15), (5, 5), (15, 44), (3, 8), (15, 60)]     # (15, 60): deque of 47 > 32, host fallback on both sides
    total = 0
    for path in paths:
        graph = gca.AlignmentGraph(path)
        for k, w in shapes:
            monkeypatch.delenv("GC_SEEDER_BUILD", raising=False)
            dev = gca.MinimizerSeeder(graph, k, w)
            monkeypatch.setenv("GC_SEEDER_BUILD", "host")
            host = gca.MinimizerSeeder(graph, k, w)
            for name in ["kmers", "start", "positions", "maxcount"]:
                a, b = dev.array(name), host.array(name)
                assert np.array_equal(a, b), (path, k, w, name, len(a), len(b))
            total += len(host.array("positions"))
    assert total > 50000
