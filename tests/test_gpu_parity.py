"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, bit-exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COMPARE_KEYS = [
    "read_seed_off", "seed_node", "seed_offset", "seed_seqpos", "seed_goodness",
    "read_anchor_off", "anchor_x", "anchor_y", "anchor_path_off", "anchor_path",
    "anchor_first_node", "anchor_first_offset", "anchor_first_seqpos",
    "anchor_last_node", "anchor_last_offset", "anchor_last_seqpos", "anchor_score",
    "anchor_trace_off", "anchor_trace_node", "anchor_trace_offset", "anchor_trace_seqpos", "anchor_trace_switch",
    "read_chain_off", "chain", "chain_score", "failed_assertion", "seeds_extended",
    "read_path_off", "path_node", "path_offset",
    "flatten_ties",      # r5: per read, the fragment extensions whose backtrace started from a last-slice minimum tied between nodes (the one rule whose order this build defines)
] + ["chain_edit_distance", "chained_better"] + [
    # the chained alignment's trace = edlib's alignment path walked over the stitched path (src/Aligner.cpp:845-897); run_case asks for it for every read
    "read_chain_trace_off", "chain_trace_node", "chain_trace_offset", "chain_trace_seqpos", "chain_trace_switch", "chain_aln_start", "chain_aln_end",
]


def compare(got, want, keys=COMPARE_KEYS):
    for key in keys:
        g = np.asarray(got[key], dtype=np.int64)
        w = np.asarray(want[key], dtype=np.int64)
        assert g.shape == w.shape, f"{key}: shape {g.shape} vs oracle {w.shape}"
        if not np.array_equal(g, w):
            bad = np.nonzero(g != w)[0]
            raise AssertionError(f"{key}: {len(bad)} mismatches, first at {bad[0]}: got {g[bad[0]]} oracle {w[bad[0]]}")


@pytest.fixture(scope="module")
def gca():
    import graphchainer_amd as g
    assert g.device_count() >= 1, "GPU tests need a device"
    return g


LONG_KEYS = ["read_long_off", "long_start", "long_end", "long_score", "long_edit_distance",
             "read_longall_off", "longall_start", "longall_end", "longall_score",
             "long_trace_off", "long_trace_node", "long_trace_offset", "long_trace_seqpos", "long_trace_switch", "flatten_ties_long"]


def run_case(gca, gfa, reads, long_pass=False, **kw):
    from oracle import Oracle
    graph = gca.AlignmentGraph(gfa)
    seeder = gca.MinimizerSeeder(graph)
    capacities = kw.pop("capacities", None)
    keep_traces = kw.pop("keep_traces", True)
    aligner = gca.Aligner(graph, seeder, keep_traces=keep_traces, keep_seeds=True, long_pass=long_pass, chain_traces=2, capacities=capacities, **kw)
    got = {k: (v.astype(np.int64) if v.dtype.kind in "ui" and k not in ("counters", "counters_long") else v) for k, v in aligner.align_reads(reads).items()}
    expand_stitched_path(got, graph.array("nodeLength"))
    mark_missing_chain_alignments(got)
    # selected whole-read alignments come back as indices into the read's longall list
    sel = np.repeat(got["read_longall_off"][:-1], np.diff(got["read_long_off"])) + got["long_index"]
    for key in ("start", "end", "score"):
        got["long_" + key] = got["longall_" + key][sel]
    want = Oracle(gfa, long_pass=long_pass, **kw).align(reads)
    return got, want


def mark_missing_chain_alignments(got):
    """The C ABI reports alignmentStart / alignmentEnd as 0, 0 for a read without a chained alignment; the oracle says -1."""
    none = np.diff(got["read_chain_trace_off"]) == 0
    for key in ("chain_aln_start", "chain_aln_end"):
        got[key] = np.where(none, -1, got[key])


def expand_stitched_path(got, node_length):
    """The C-ABI returns the stitched path as its node list + end offsets; expand it the way the reference's
    pathToTrace does (src/Aligner.cpp:409-424: one (node, offset) cell per base, first node from its start offset,
    last node - if it is not also the first - up to its last offset) so it compares with the oracle's `longest`."""
    off = got["read_path_off"]
    nodes_out, offs_out, read_off = [], [], [0]
    for r in range(len(off) - 1):
        path = got["path_node"][off[r]:off[r + 1]]
        cells = 0
        for node in path:
            s, l = 0, int(node_length[node])
            if node == path[0]:
                s = int(got["path_first_offset"][r])
            elif node == path[-1]:
                l = int(got["path_last_offset"][r]) + 1
            if l > s:
                nodes_out.append(np.full(l - s, node, dtype=np.int64))
                offs_out.append(np.arange(s, l, dtype=np.int64))
                cells += l - s
        assert cells == int(got["path_cells"][r])
        read_off.append(read_off[-1] + cells)
    got["path_nodes_raw"] = got["path_node"]
    got["read_path_off"] = np.array(read_off, dtype=np.int64)
    got["path_node"] = np.concatenate(nodes_out) if nodes_out else np.zeros(0, dtype=np.int64)
    got["path_offset"] = np.concatenate(offs_out) if offs_out else np.zeros(0, dtype=np.int64)


def test_reference_fixture(gca, golden_dir):
    """BASELINE config 1: the reference's own test/graph.gfa + test/read.fa (committed copies of the data files)."""
    gfa = os.path.join(golden_dir, "ref_test_graph.gfa")
    read = open(os.path.join(golden_dir, "ref_test_read.fa")).read().split("\n")[1]
    got, want = run_case(gca, gfa, [read])
    compare(got, want)
    # the anchor SURVEY.md §8c recorded from the reference sources on this input: x=35, y=69, score=3, path 6,4,0, chain [0]
    assert list(got["anchor_x"]) == [35] and list(got["anchor_y"]) == [69] and list(got["anchor_score"]) == [3]
    assert list(got["anchor_path"]) == [6, 4, 0] and list(got["chain"]) == [0]


def test_host_copy_of_the_mpc_index_can_be_released(gca, tmp_path):
    """gc_graph_trim_host: aligning reads nothing of the host's MPC copy (the device has its own); saving the cache and the MPC views refuse afterwards."""
    from graphchainer_amd.synth import SynthGenome
    sg = SynthGenome(3, 30_000, seed=7, multi_allelic=0.1, nested=0.1, minus_links=0.3, repeats=4, repeat_len=3000)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(9, 4000, seed=11)
    graph = gca.AlignmentGraph(gfa)
    seeder = gca.MinimizerSeeder(graph)
    aligner = gca.Aligner(graph, seeder, keep_traces=True, long_pass=True, chain_traces=2, colinear_gap=50000)
    before = {k: np.array(v) for k, v in aligner.align_reads(reads).items()}   # (copies: the arrays are views of the result block)
    width = graph.array("mpc_width")
    graph.trim_host()
    after = {k: np.array(v) for k, v in aligner.align_reads(reads).items()}
    fresh = gca.Aligner(graph, seeder, keep_traces=True, long_pass=True, chain_traces=2, colinear_gap=50000).align_reads(reads)
    assert len(width) == 6 and int(before["read_chain_off"][-1]) > 0
    for key in before:
        if key in ("counters", "counters_long", "kernel_us", "host_us"):
            continue
        assert np.array_equal(before[key], after[key]), key
        assert np.array_equal(before[key], fresh[key]), key
    for name in ("mpc_width", "paths", "back_node", "topo_id", "mpc_path_nodes"):
        with pytest.raises(Exception, match="released"):
            graph.array(name)
    assert len(graph.array("nodeLength")) == graph.NodeSize()
    with pytest.raises(Exception, match="released"):
        gca.api.save_index_cache(graph, seeder, str(tmp_path / "x.gcidx"))
    assert not os.path.exists(str(tmp_path / "x.gcidx"))


@pytest.mark.parametrize("backbone,n_reads,read_len,kw", [
    (40_000, 6, 2000, {}),
    (120_000, 12, 5000, {}),
    (120_000, 6, 5000, {"split_gap": 18}),      # BASELINE config 3 spelling of --sampling-step 0.5
    (80_000, 6, 3000, {"bandwidth": 5}),
])
def test_synthetic_parity(gca, tmp_path, backbone, n_reads, read_len, kw):
    from graphchainer_amd.synth import SynthGraph
    sg = SynthGraph(backbone, seed=7)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(n_reads, read_len, seed=11)
    got, want = run_case(gca, gfa, reads, **kw)
    compare(got, want)
    assert int(got["read_chain_off"][-1]) > 0


def test_edge_cases(gca, tmp_path):
    """Empty read, read shorter than a fragment, read with N runs, homopolymer read, lower-case read."""
    from graphchainer_amd.synth import SynthGraph
    sg = SynthGraph(30_000, seed=3)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    base = sg.sample_reads(3, 1500, seed=5)
    with_n = bytearray(base[0])
    with_n[200:230] = b"N" * 30
    with_n[700] = ord("n")
    reads = [b"", b"ACGTACGTAC", bytes(with_n), b"A" * 500, base[1], base[2].lower()]
    got, want = run_case(gca, gfa, reads)
    compare(got, want)


def test_counters_cover_oracle_work(gca, tmp_path):
    """The work counters that price the roofline (tiles, column steps) use the oracle's own unit."""
    from graphchainer_amd.synth import SynthGraph
    sg = SynthGraph(60_000, seed=9)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(4, 3000, seed=2)
    got, want = run_case(gca, gfa, reads)
    compare(got, want)
    # the device extends every fragment seed (filtered ones are discarded afterwards), the oracle only the accepted
    # ones, so device counts are >= oracle counts
    assert int(got["counters"][4]) >= int(want["counters"][4])
    assert int(got["counters"][2]) >= int(want["counters"][2])


@pytest.mark.parametrize("backbone,n_reads,read_len", [(60_000, 6, 3000), (150_000, 8, 10_000)])
def test_whole_read_pass_parity(gca, tmp_path, backbone, n_reads, read_len):
    """The whole-read GraphAligner pass (multi-slice extensions, sloppy seed rules) against the oracle,
    traces included; the fragment pass must be unaffected by running both."""
    from graphchainer_amd.synth import SynthGraph
    sg = SynthGraph(backbone, seed=5)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(n_reads, read_len, seed=21)
    reads.append(reads[0][:700] + reads[1][300:1500])      # a chimeric read: more than one alignment
    got, want = run_case(gca, gfa, reads, long_pass=True)
    compare(got, want, COMPARE_KEYS + LONG_KEYS)
    assert int(got["read_longall_off"][-1]) >= n_reads


def test_whole_read_pass_plain_layout_fallback(gca, tmp_path, monkeypatch):
    """Reads whose band does not fit the LDS tables are rerun with the plain-layout kernel; force that path for all."""
    from graphchainer_amd.synth import SynthGraph
    monkeypatch.setenv("GC_TEST_LONG_FORCE_FALLBACK", "1")
    sg = SynthGraph(60_000, seed=8)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(5, 4000, seed=4)
    got, want = run_case(gca, gfa, reads, long_pass=True)
    compare(got, want, COMPARE_KEYS + LONG_KEYS)
    assert int(got["counters_long"][7]) == len(reads)


def test_whole_read_pass_speculative_rounds(gca, tmp_path, monkeypatch):
    """Tail rounds extend several seeds of a read at once and re-check them in order; force that from round 0."""
    from graphchainer_amd.synth import SynthGraph
    monkeypatch.setenv("GC_TEST_LONG_SPECULATE", "2")
    sg = SynthGraph(100_000, seed=12)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(10, 6000, seed=6)
    reads.append(reads[2][:2500] + reads[5][1000:4000])
    got, want = run_case(gca, gfa, reads, long_pass=True)
    compare(got, want, COMPARE_KEYS + LONG_KEYS)


def test_whole_read_assertion_drops_the_read(gca, tmp_path, monkeypatch):
    """`cont` is one flag per read in the reference (src/Aligner.cpp:529): set by the whole-read pass's catch (:591) it
    makes the fragment loop skip every anchor (:702), so such a read has no anchors, no chain and no alignment at all.
    GC_TEST_FAIL_LONG=<read> makes that read's whole-read pass "assert" in the product and in the oracle."""
    from graphchainer_amd.synth import SynthGraph
    sg = SynthGraph(60_000, seed=21)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(5, 3000, seed=2)
    monkeypatch.setenv("GC_TEST_FAIL_LONG", "3")
    got, want = run_case(gca, gfa, reads, long_pass=True)
    compare(got, want, COMPARE_KEYS + LONG_KEYS)
    for key in ("read_anchor_off", "read_chain_off", "read_longall_off", "read_long_off", "read_path_off"):
        assert got[key][4] == got[key][3], key            # nothing for read 3 ...
        assert got[key][3] > got[key][2] and got[key][5] > got[key][4], key   # ... and its neighbours are untouched
    assert list(got["failed_assertion"]) == [0, 0, 0, 1, 0]
    assert got["chained_better"][3] == 0 and got["chain_edit_distance"][3] == -1 and got["long_edit_distance"][3] == -1


@pytest.mark.parametrize("env", [{"GC_TEST_LONG_TEAM": "8"}, {"GC_TEST_LONG_TEAM": "64", "GC_TEST_LONG_ORDER": "0"}, {"GC_TEST_LONG_MAX_BLOCKS": "7"}, {"GC_TEST_LONG_MAX_BLOCKS": "3", "GC_TEST_LONG_TEAM": "4"}, {"GC_TEST_LONG_REG_CAP": "3"}, {"GC_TEST_LONG_REG_CAP": "6", "GC_TEST_LONG_MAX_BLOCKS": "5"}])
def test_whole_read_pass_launch_shapes(gca, tmp_path, monkeypatch, env):
    """Launch-shape knobs of the whole-read pass (concurrent read groups, lanes per wave, execution order, persistent waves, register-table cap -> LDS-table retry) never change results."""
    from graphchainer_amd.synth import SynthGraph
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    sg = SynthGraph(120_000, seed=3)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(140, 2500, seed=9)
    got, want = run_case(gca, gfa, reads, long_pass=True)
    compare(got, want, COMPARE_KEYS + LONG_KEYS)


def _mutate(rng, s, rate):
    out = bytearray()
    for ch in s:
        x = rng.random()
        if x < rate / 3:
            continue                                   # deletion
        if x < 2 * rate / 3:
            out.append(rng.choice(b"ACGT"))            # substitution
        else:
            out.append(ch)
        if rng.random() < rate / 3:
            out.append(rng.choice(b"ACGT"))            # insertion
    return bytes(out)


def test_edit_distance_kernel(gca, monkeypatch):
    """NW edit distance kernel (banded Myers wavefront, k doubling, unit escalation) against the oracle's plain DP value,
    which tests/test_oracle_units.py pins to edlib."""
    import random
    from oracle import Oracle  # noqa: F401  (builds the oracle library)
    from oracle.binding import load_oracle_lib
    import ctypes as C
    lib = load_oracle_lib()
    lib.gco_edit_distance.restype = C.c_uint64
    lib.gco_edit_distance.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]
    rng = random.Random(5)
    pairs = []
    # (r4: the 5-10 kb pairs below ~12 % share waves three at a time - teams of 21 lanes, bands below 1290 -, the 10 kb pair at 12.6 % is handed on to the two-per-wave kernel,
    # the one at 30 % from there to the one-pair kernels)
    for length, rate in [(1, 0.0), (63, 0.1), (64, 0.3), (65, 0.0), (700, 0.15), (3000, 0.05), (3000, 0.4), (10000, 0.12), (10000, 0.3), (20000, 0.25),
                         (5000, 0.1), (8000, 0.11), (10000, 0.05), (10000, 0.126), (9000, 0.02), (1400, 0.5), (12000, 0.1)]:
        a = bytes(rng.choice(b"ACGT") for _ in range(length))
        pairs.append((a, _mutate(rng, a, rate)))
    pairs.append((b"ACGTNNRYACGT" * 30, b"ACGTNARYACGA" * 29))                     # letters outside ACGT compare by equality
    pairs.append((bytes(rng.choice(b"ACGT") for _ in range(5000)), bytes(rng.choice(b"ACGT") for _ in range(4000))))   # unrelated: distance ~ half the length
    pairs.append((b"A" * 300, b"ACGT" * 2000))                                      # very different lengths
    pairs.append((b"", b"ACGT"))
    pairs.append((b"ACGT", b""))
    got = gca.edit_distance([p[0] for p in pairs], [p[1] for p in pairs])
    want = [len(a) + len(b) if (not a or not b) else int(lib.gco_edit_distance(a, len(a), b, len(b))) for a, b in pairs]
    assert list(map(int, got)) == want
    # r4: a first band of half the read or more sends the pair to k_edit_distance_block (a workgroup per pair, the whole matrix): here every pair, both ways round
    monkeypatch.setenv("GC_ED_FIRST_K", "30000")
    assert list(map(int, gca.edit_distance([p[0] for p in pairs], [p[1] for p in pairs]))) == want
    assert list(map(int, gca.edit_distance([p[1] for p in pairs], [p[0] for p in pairs]))) == want


def test_edit_distance_long_low_error_pairs(gca, monkeypatch):
    """pairs of 131-150 kb at < 1 % error entered the two-pairs-per-wave kernel with k < 1900 (a whole-read alignment's own bound)
    and read letters from ring slots already overwritten (units beyond 2048 behind a refill that runs 2048 ahead); k_edit_distance<1> had the
    same window with 8192 slots. GC_ED_FIRST_K makes gc_edit_distance start from such a small band."""
    import random
    from oracle.binding import RefUnits, load_oracle_lib
    try:
        ref = RefUnits().lib.ref_edit_distance          # the real edlib, compiled from /root/reference where it lies
    except (FileNotFoundError, OSError):
        ref = load_oracle_lib().gco_edit_distance
    rng = random.Random(17)
    pairs = []
    # (r4: the three-pairs-per-wave kernel's ring holds 2048 letters and is refilled 1024 ahead: reads up to 65536 bases go there, the next length must not)
    for length, rate in [(140_000, 0.006), (149_000, 0.004), (131_072, 0.008), (120_000, 0.01), (400_000, 0.004), (60_000, 0.012), (65_536, 0.01), (65_537, 0.01)]:
        a = bytes(rng.choice(b"ACGT") for _ in range(length))
        pairs.append((a, _mutate(rng, a, rate)))
    want = [int(ref(a, len(a), b, len(b))) for a, b in pairs]
    for first_k in ("1200", "1850", "3900"):
        monkeypatch.setenv("GC_ED_FIRST_K", first_k)
        got = gca.edit_distance([p[0] for p in pairs], [p[1] for p in pairs])
        assert list(map(int, got)) == want, first_k
        got = gca.edit_distance([p[1] for p in pairs], [p[0] for p in pairs])
        assert list(map(int, got)) == want, first_k
    # r4: a first band of half the read or more sends a read of up to 65536 bases to k_edit_distance_block (a workgroup per pair: 938 and 1024 threads here), the next length to the widest unit
    monkeypatch.setenv("GC_ED_FIRST_K", "40000")
    short = [i for i, (a, b) in enumerate(pairs) if len(a) <= 65_537]
    assert len(short) == 3
    got = gca.edit_distance([pairs[i][0] for i in short], [pairs[i][1] for i in short])
    assert list(map(int, got)) == [want[i] for i in short]
    got = gca.edit_distance([pairs[i][1] for i in short], [pairs[i][0] for i in short])
    assert list(map(int, got)) == [want[i] for i in short]


def _revcomp(s):
    return s[::-1].translate(bytes.maketrans(b"ACGTacgtNn", b"TGCAtgcaNn"))


def test_edge_cases_whole_read(gca, tmp_path):
    """Whole-read pass, stitching and the decision on awkward input: empty / tiny reads, N runs, homopolymers, lower case,
    reverse-strand reads, and a graph whose segments carry IUPAC letters (ambiguous split nodes)."""
    from graphchainer_amd.synth import SynthGraph
    sg = SynthGraph(50_000, seed=13)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    # sprinkle IUPAC letters over a few segments
    lines = open(gfa).read().split("\n")
    touched = 0
    for i, line in enumerate(lines):
        if line.startswith("S\t") and touched < 12:
            f = line.split("\t")
            if len(f[2]) >= 20 and i % 7 == 0:
                seq = bytearray(f[2].encode())
                seq[len(seq) // 2] = ord("NRYKMSW"[touched % 7])
                f[2] = seq.decode()
                lines[i] = "\t".join(f)
                touched += 1
    assert touched > 0
    open(gfa, "w").write("\n".join(lines))
    base = sg.sample_reads(6, 2500, seed=17)
    with_n = bytearray(base[0])
    with_n[300:340] = b"N" * 40
    reads = [b"", b"ACGTACGTAC", bytes(with_n), b"A" * 600, base[1], base[2].lower(), _revcomp(base[3]), _revcomp(base[4])[:1800], base[5][:40]]
    got, want = run_case(gca, gfa, reads, long_pass=True)
    compare(got, want, COMPARE_KEYS + LONG_KEYS)


@pytest.mark.parametrize("merge", [False, True])
def test_gaf_output(gca, tmp_path, merge):
    """GAF lines of the final alignments (node path with original names, cigar, NM/dv/id tags, per-read order) against the
    oracle's restatement of GraphAlignerGAFAlignment.h; graph with IUPAC letters, reverse-strand and chimeric reads."""
    from graphchainer_amd.synth import SynthGraph
    from oracle import Oracle
    sg = SynthGraph(80_000, seed=21)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    lines = open(gfa).read().split("\n")
    touched = 0
    for i, line in enumerate(lines):
        if line.startswith("S\t") and touched < 8 and i % 11 == 0:
            f = line.split("\t")
            if len(f[2]) >= 20:
                seq = bytearray(f[2].encode())
                seq[len(seq) // 3] = ord("NRYK"[touched % 4])
                f[2] = seq.decode()
                lines[i] = "\t".join(f)
                touched += 1
    open(gfa, "w").write("\n".join(lines))
    reads = sg.sample_reads(8, 3000, seed=2)
    reads.append(_revcomp(reads[0]))
    reads.append(reads[1][:900] + reads[2][400:1700])
    graph = gca.AlignmentGraph(gfa)
    seeder = gca.MinimizerSeeder(graph)
    aligner = gca.Aligner(graph, seeder, keep_traces=True, long_pass=True)
    got = aligner.align_reads(reads, gaf_names=[f"r{i}" for i in range(len(reads))], cigar_match_mismatch_merge=merge)
    ora = Oracle(gfa, long_pass=True)
    ora.align(reads)
    want = ora.gaf(merge)
    assert got["gaf"] == want
    assert got["gaf"].count(b"\n") >= len(reads) - got["gaf_chained_skipped"] - 1
    assert b"cg:Z:" in got["gaf"]
    # r4: the same lines put together from what the device wrote (k_out_encode: path and CIGAR text, the counts of the other columns) - no trace comes down
    dev = gca.Aligner(graph, seeder, long_pass=True, device_output=2 if merge else 1)
    got_dev = dev.align_reads(reads, gaf_names=[f"r{i}" for i in range(len(reads))], cigar_match_mismatch_merge=merge)
    assert "long_trace_off" not in got_dev
    assert got_dev["gaf"] == want
    assert int(got_dev["read_out_off"][-1]) == want.count(b"\n")


def test_chained_alignment_wins(gca, tmp_path):
    """Reads with a 1.5 kb deletion: the whole-read aligner stops at the breakpoint, the chain bridges it (stitching with a
    BFS bridge), and the edit distances make the chained alignment the read's result (src/Aligner.cpp:901-905)."""
    from graphchainer_amd.synth import SynthGraph
    from oracle import Oracle
    sg = SynthGraph(200_000, seed=19)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    bb = sg.backbone.tobytes()
    reads = [bb[x:x + 3000] + bb[x + 4500:x + 7500] for x in (10_000, 60_000, 120_000)]
    reads += [_revcomp(reads[0])]
    reads += sg.sample_reads(3, 4000, seed=8)
    graph = gca.AlignmentGraph(gfa)
    seeder = gca.MinimizerSeeder(graph)
    aligner = gca.Aligner(graph, seeder, keep_traces=True, keep_seeds=True, long_pass=True)     # chain_traces = 1: winners only
    names = [f"r{i}" for i in range(len(reads))]
    raw = aligner.align_reads(reads, gaf_names=names, other_formats=True)
    got = {k: (v.astype(np.int64) if isinstance(v, np.ndarray) and v.dtype.kind in "ui" and k not in ("counters", "counters_long") else v) for k, v in raw.items()}
    expand_stitched_path(got, graph.array("nodeLength"))
    mark_missing_chain_alignments(got)
    sel = np.repeat(got["read_longall_off"][:-1], np.diff(got["read_long_off"])) + got["long_index"]
    for key in ("start", "end", "score"):
        got["long_" + key] = got["longall_" + key][sel]
    ora = Oracle(gfa, long_pass=True)
    want = ora.align(reads)
    winners = want["chained_better"].astype(bool)
    # the oracle traces every read, this run only the winners: compare the winners' traces, everything else as usual
    trace_keys = [k for k in COMPARE_KEYS if k.startswith(("read_chain_trace", "chain_trace", "chain_aln"))]
    compare(got, want, [k for k in COMPARE_KEYS + LONG_KEYS if k not in trace_keys])
    assert np.array_equal(np.diff(got["read_chain_trace_off"]) > 0, winners)
    for r in np.nonzero(winners)[0]:
        g0, g1 = got["read_chain_trace_off"][r], got["read_chain_trace_off"][r + 1]
        w0, w1 = want["read_chain_trace_off"][r], want["read_chain_trace_off"][r + 1]
        for k in ("chain_trace_node", "chain_trace_offset", "chain_trace_seqpos", "chain_trace_switch"):
            assert np.array_equal(got[k][g0:g1], want[k][w0:w1]), (k, r)
        assert got["chain_aln_start"][r] == want["chain_aln_start"][r] and got["chain_aln_end"][r] == want["chain_aln_end"][r]
    assert int(np.sum(got["chained_better"][:4])) >= 3          # the deletion reads
    assert int(np.sum(got["chained_better"][4:])) == 0           # ordinary reads keep their whole-read alignment
    # output: every read is written, the winners from their chained alignment (src/Aligner.cpp:901-920)
    assert raw["gaf_chained_skipped"] == 0
    assert raw["gaf"] == ora.gaf(False)
    assert raw["json"] == ora.json()
    assert raw["gaf"].count(b"\n") >= len(reads)
    import gzip
    assert gzip.decompress(raw["gam"])   # framed messages decode (content is checked against the JSON in test_json_and_gam_output)
    # r4: with the output encoded on the device the winners' entries are marked for the host (their trace is built there), everything else comes as pieces
    dev = gca.Aligner(graph, seeder, long_pass=True, device_output=1 | 4).align_reads(reads, gaf_names=names, other_formats=True)
    assert dev["gaf"] == raw["gaf"] and dev["json"] == raw["json"] and gzip.decompress(dev["gam"]) == gzip.decompress(raw["gam"])
    assert int(np.sum(dev["out_source"])) == int(np.sum(got["chained_better"]))


@pytest.mark.parametrize("upload_slice", [None, "7"])
def test_config5_shape(gca, tmp_path, monkeypatch, upload_slice):
    """(upload_slice: the MPC index's backward links reach the device in slices of nodes - 64 M links each; the hook makes them 7.)
    BASELINE config 5 in miniature: several chromosomes in one GFA (6 weakly connected components: the cross-component rule of
    src/AlignmentGraph.cpp:1722-1733), path-cover width > 2 (multi-allelic and nested bubbles), links written from the reverse
    strand, repeats, 50 kb reads at PacBio-CLR-like error rates (4 % deletions, 2 % substitutions, 9 % insertions),
    --colinear-gap 50000, whole-read pass on."""
    from graphchainer_amd.synth import SynthGenome
    gen = SynthGenome(3, 300_000, seed=41, multi_allelic=0.25, nested=0.3, minus_links=0.3, repeats=8, repeat_len=3000)
    gfa = str(tmp_path / "g.gfa")
    gen.write_gfa(gfa)
    reads = gen.sample_reads(5, 50_000, seed=6, p_del=0.04, p_sub=0.02, p_ins=0.09)
    reads.append(reads[0][:20_000] + reads[1][:25_000])      # a read spanning two chromosomes
    if upload_slice:
        monkeypatch.setenv("GC_TEST_UPLOAD_SLICE", upload_slice)
    got, want = run_case(gca, gfa, reads, long_pass=True, colinear_gap=50000)
    compare(got, want, COMPARE_KEYS + LONG_KEYS)
    graph = gca.AlignmentGraph(gfa)
    assert len(graph.array("mpc_width")) == 6 and int(graph.array("mpc_width").max()) >= 3
    assert int(got["read_chain_off"][-1]) > 1000 and int(got["read_longall_off"][-1]) >= 5


def test_device_output_equals_host_encoders_on_varied_inputs(gca, tmp_path):
    """k_out_encode (gc_params::device_output) against the host encoders walking the downloaded traces, byte for byte in all three formats and both CIGAR styles, on inputs that
    stretch the walk: several chromosomes with long unsplit segments (mappings spanning many 64-cell chunks), multi-allelic and nested bubbles (one-base nodes: many path steps per
    chunk), links written from the reverse strand, repeats, non-numeric segment names, IUPAC letters in the graph, N runs and lower-case letters in the reads, reverse-strand and
    chimeric reads, reads of 1-70 bases, 50 kb CLR-like reads with long insertion / deletion runs."""
    import gzip
    from graphchainer_amd.synth import SynthGenome
    gen = SynthGenome(2, 150_000, seed=47, multi_allelic=0.25, nested=0.3, minus_links=0.3, repeats=4, repeat_len=2000)
    gfa = str(tmp_path / "g.gfa")
    gen.write_gfa(gfa)
    # rename a third of the segments (names are what the path column and the mapping positions print) and put IUPAC letters into some
    lines = open(gfa).read().split("\n")
    names, touched = {}, 0
    for i, line in enumerate(lines):
        if line.startswith("S\t"):
            f = line.split("\t")
            if i % 3 == 0:
                names[f[1]] = "seg_" + f[1] + ("x" * (i % 5))
            if len(f[2]) >= 30 and i % 17 == 0 and touched < 40:
                seq = bytearray(f[2].encode())
                seq[len(seq) // 2] = ord("NRYKMSW"[touched % 7])
                f[2] = seq.decode()
                touched += 1
            lines[i] = "\t".join(f)
    for i, line in enumerate(lines):
        f = line.split("\t")
        if f[0] == "S":
            f[1] = names.get(f[1], f[1])
        elif f[0] == "L":
            f[1], f[3] = names.get(f[1], f[1]), names.get(f[3], f[3])
        lines[i] = "\t".join(f)
    open(gfa, "w").write("\n".join(lines))
    reads = gen.sample_reads(6, 6000, seed=3) + gen.sample_reads(2, 50_000, seed=4, p_del=0.04, p_sub=0.02, p_ins=0.09)
    with_n = bytearray(reads[0])
    with_n[700:760] = b"N" * 60
    reads += [bytes(with_n), reads[1].lower(), _revcomp(reads[2]), reads[3][:2500] + _revcomp(reads[4])[1000:4000], reads[5][:70], reads[5][100:135], b"A", b"ACGTTGCA"]
    graph = gca.AlignmentGraph(gfa)
    seeder = gca.MinimizerSeeder(graph)
    ids = [f"read/{i}" for i in range(len(reads))]
    host = gca.Aligner(graph, seeder, keep_traces=2, long_pass=True, colinear_gap=50000).align_reads(reads, gaf_names=ids, other_formats=True)
    dev = gca.Aligner(graph, seeder, long_pass=True, colinear_gap=50000, device_output=1 | 4).align_reads(reads, gaf_names=ids, other_formats=True)
    assert dev["gaf"] == host["gaf"] and dev["gaf"].count(b"\n") >= 10
    assert dev["json"] == host["json"]
    assert gzip.decompress(dev["gam"]) == gzip.decompress(host["gam"])
    assert b"seg_" in dev["gaf"] and b"<" in dev["gaf"]
    host_m = gca.Aligner(graph, seeder, keep_traces=2, long_pass=True, colinear_gap=50000).align_reads(reads, gaf_names=ids, cigar_match_mismatch_merge=True)
    dev_m = gca.Aligner(graph, seeder, long_pass=True, colinear_gap=50000, device_output=2).align_reads(reads, gaf_names=ids, cigar_match_mismatch_merge=True)
    assert dev_m["gaf"] == host_m["gaf"] and dev_m["gaf"] != dev["gaf"]
    # the counts the device reports are the lines' own columns
    numbers = np.asarray(dev["out_numbers"]).reshape(-1, 12)
    lines_out = [l for l in dev["gaf"].decode().splitlines() if l]
    assert len(lines_out) == len(numbers)
    for line, num, source in zip(lines_out, numbers, np.asarray(dev["out_source"])):
        if source != 0:
            continue                                                        # (a chained winner: written by the host encoder, no counts from the device)
        col = line.split("\t")
        assert (int(col[2]), int(col[3]), int(col[6]), int(col[7]), int(col[8]), int(col[9]), int(col[10])) == (int(num[8]), int(num[9]), int(num[0]), int(num[1]), int(num[2]), int(num[3]), int(num[7]))


def test_config3_shape_with_whole_read_pass(gca, tmp_path):
    """BASELINE config 3 in miniature: 10 kb reads, --colinear-split-gap 18 (the reference's spelling of --sampling-step 0.5:
    overlapping fragments, twice the anchors, the overlap branch of the chaining DP), whole-read pass on, a graph with repeats."""
    from graphchainer_amd.synth import SynthGraph
    sg = SynthGraph(400_000, seed=43, repeats=6, repeat_len=2500, multi_allelic=0.1)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(24, 10_000, seed=8)
    got, want = run_case(gca, gfa, reads, long_pass=True, split_gap=18)
    compare(got, want, COMPARE_KEYS + LONG_KEYS)
    assert int(got["read_anchor_off"][-1]) > 24 * 200


@pytest.mark.parametrize("host_anchors", ["0", "1"])
def test_anchor_arrays_made_on_the_device(gca, tmp_path, monkeypatch, host_anchors):
    """r5: without the anchors' traces (keep_traces 0 or 2: what bench.py and the output encoders run) the result's anchor_* arrays are compacted per read on the
    device (gc_results.hip: every valid anchor up to the first fragment that threw, in fragment and seed order) and come down as they are; GC_HOST_ANCHORS=1 keeps
    the host's walk over the slots. Both against the oracle: 10 kb reads with the whole-read pass, overlapping fragments, chimeric reads, a read too short for a fragment,
    an empty read, a read of N."""
    from graphchainer_amd.synth import SynthGraph
    monkeypatch.setenv("GC_HOST_ANCHORS", host_anchors)
    sg = SynthGraph(300_000, seed=17, repeats=4, repeat_len=2000, multi_allelic=0.1)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(40, 10_000, seed=3)
    reads += [reads[0][:3000] + reads[1][2000:6000], "ACGT" * 5, "", "N" * 400, reads[2][:36]]
    keys = [k for k in COMPARE_KEYS + LONG_KEYS if "trace" not in k]
    for kw in ({}, {"split_gap": 18}):
        got, want = run_case(gca, gfa, reads, long_pass=True, keep_traces=False, **kw)
        compare(got, want, keys)
        assert int(got["read_anchor_off"][-1]) > 40 * 100
    got, want = run_case(gca, gfa, reads[:12], long_pass=False, keep_traces=False)
    compare(got, want, [k for k in COMPARE_KEYS if "trace" not in k])


@pytest.mark.parametrize("plain_scan", ["0", "1"])
def test_chain_kernel_scratch_path(gca, tmp_path, monkeypatch, plain_scan):
    """k_chain keeps a read's anchors and path entries in LDS; reads beyond those tables (or on a cover wider than the LDS threshold
    table) run in a second launch with the same code on HBM scratch. Forced for every read here. r4: that launch scans the entries grouped by
    weakly connected component (the default) or all earlier entries (GC_CHAIN_PLAIN_SCAN=1: what a read touching more than 256 components gets);
    chimeric reads put anchors of several components into one read."""
    from graphchainer_amd.synth import SynthGenome
    monkeypatch.setenv("GC_TEST_CHAIN_FORCE_SCRATCH", "1")
    monkeypatch.setenv("GC_CHAIN_PLAIN_SCAN", plain_scan)
    gen = SynthGenome(6, 20_000, seed=29, multi_allelic=0.3, nested=0.3, repeats=4, repeat_len=1500)
    gfa = str(tmp_path / "g.gfa")
    gen.write_gfa(gfa)
    reads = gen.sample_reads(300, 1200, seed=5)          # more reads than the scratch launch has blocks
    reads += [reads[i][:500] + reads[i + 1][200:700] + reads[i + 2][:400] for i in range(0, 60, 3)]   # pieces of three places: several components per read
    got, want = run_case(gca, gfa, reads, split_gap=18)
    compare(got, want)


def test_chain_kernel_against_bruteforce(gca, tmp_path):
    """k_chain's chains against the quadratic DP over BFS reachability of tests/graph_model.py (independent of the oracle and of
    the MPC index): several components, wide covers, overlapping fragments."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from graph_model import GraphModel, chain_bruteforce
    from graphchainer_amd.synth import SynthGenome
    gen = SynthGenome(2, 50_000, seed=23, multi_allelic=0.3, nested=0.4, minus_links=0.3, repeats=4, repeat_len=1500)
    gfa = str(tmp_path / "g.gfa")
    gen.write_gfa(gfa)
    reads = gen.sample_reads(6, 3000, seed=3)
    reads.append(reads[0][:1500] + reads[1][:1500])
    model = GraphModel(open(gfa).read())
    comp, _, _ = model.weak_components()
    graph = gca.AlignmentGraph(gfa)
    assert list(graph.array("nodeIDs")) == model.ids           # the product's numbering is the model's
    seeder = gca.MinimizerSeeder(graph)
    for kw in ({}, {"split_gap": 18}):
        res = gca.Aligner(graph, seeder, **kw).align_reads(reads)
        for r in range(len(reads)):
            a0, a1 = int(res["read_anchor_off"][r]), int(res["read_anchor_off"][r + 1])
            anchors = [([int(v) for v in res["anchor_path"][int(res["anchor_path_off"][a]):int(res["anchor_path_off"][a + 1])]], int(res["anchor_x"][a]), int(res["anchor_y"][a])) for a in range(a0, a1)]
            if not anchors:
                continue
            chain, score = chain_bruteforce(model.out, model.inn, comp, anchors)
            assert [int(c) for c in res["chain"][int(res["read_chain_off"][r]):int(res["read_chain_off"][r + 1])]] == chain, (kw, r)
            assert int(res["chain_score"][r]) == score


def _per_read(res, off_key, key, r):
    return np.asarray(res[key][int(res[off_key][r]):int(res[off_key][r + 1])], dtype=np.int64)


def test_fragment_overflow_is_retried(gca, tmp_path):
    """Slabs far too small for most fragment extensions (gc_params::capacity: 8 tiles, 8 queue entries): the retry launch with 16x the room recovers every
    one of them - same results as the oracle, nothing flagged."""
    from graphchainer_amd.synth import SynthGraph
    sg = SynthGraph(80_000, seed=51, multi_allelic=0.3, nested=0.3)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(8, 3000, seed=2)
    got, want = run_case(gca, gfa, reads, capacities={"ext_max_items": 8, "ext_max_pending": 8})
    compare(got, want)
    assert not got["capacity_exceeded"].any()


def test_capacity_block_is_validated_and_the_column_store_is_optional(gca, tmp_path):
    """gc_params::capacity: nonsense is refused before anything runs; long_column_store = -1 (the backtrace recomputes its tiles, as in r2) and a store too small for
    most extensions (they fall back to the recomputing layout) give the same answers as the default."""
    from graphchainer_amd.synth import SynthGraph
    sg = SynthGraph(60_000, seed=52)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(6, 4000, seed=3)
    graph = gca.AlignmentGraph(gfa)
    seeder = gca.MinimizerSeeder(graph)
    with pytest.raises(RuntimeError):
        gca.Aligner(graph, seeder, capacities={"ext_max_items": -5}).align_reads(reads)
    with pytest.raises(ValueError):
        gca.Aligner(graph, seeder, capacities={"no_such_table": 1})
    base = gca.Aligner(graph, seeder, long_pass=True, keep_traces=True).align_reads(reads)
    for cap in ({"long_column_store": -1}, {"long_column_store": 700}):
        got = gca.Aligner(graph, seeder, long_pass=True, keep_traces=True, capacities=cap).align_reads(reads)
        for key in base:
            if key not in ("counters", "counters_long", "kernel_us", "host_us"):
                assert np.array_equal(np.asarray(got[key]), np.asarray(base[key])), (cap, key)


def test_lazy_fragment_extension_equals_eager(gca, tmp_path, monkeypatch):
    """The fragment pass extends a seed only when the reference would (lazy rounds: first seeds, then the seeds parked fragments ask for);
    GC_EXT_LAZY=0 extends every seed of every window up front and filters afterwards. Same anchors, chains, traces and statistics either
    way - on a graph with repeats and nested bubbles (windows with many seeds, several of them needing an extension) - and fewer
    extensions run in the lazy mode; the lazy run also equals the oracle."""
    from graphchainer_amd.synth import SynthGenome
    gfa = str(tmp_path / "g.gfa")
    genome = SynthGenome(2, 60_000, seed=9, multi_allelic=0.3, nested=0.5, minus_links=0.3, repeats=6, repeat_len=400, repeat_divergence=0.03)
    genome.write_gfa(gfa)
    reads = genome.sample_reads(60, 4000, seed=3)
    graph = gca.AlignmentGraph(gfa)
    seeder = gca.MinimizerSeeder(graph)
    runs = {}
    for mode in ("0", "1"):
        with monkeypatch.context() as m:
            m.setenv("GC_EXT_LAZY", mode)
            runs[mode] = gca.Aligner(graph, seeder, keep_traces=True, keep_seeds=True, long_pass=False).align_reads(reads)
    eager, lazy = runs["0"], runs["1"]
    for key in eager:
        if key in ("kernel_us", "host_us", "counters", "counters_long"):
            continue
        a, b = eager[key], lazy[key]
        if isinstance(a, np.ndarray):
            assert np.array_equal(a, b), key
    assert int(lazy["counters"][4]) < int(eager["counters"][4])          # extensions run
    assert int(lazy["counters"][4]) >= 2 * int(lazy["seeds_extended"].sum())   # every seed the reference extends ran both directions
    got, want = run_case(gca, gfa, reads)
    compare(got, want)


def test_capacity_overflow_flags_the_read_not_the_batch(gca, tmp_path, monkeypatch):
    """Capacities shrunk until even the retries overflow (fragment slabs, whole-read extension scratch, the whole-read cell pool): the
    call succeeds, the reads that hit a limit are flagged in capacity_exceeded, and every other read has exactly the oracle's
    results (the reference's convention: a problem with one read never costs the others, src/Aligner.cpp:585-592,695-703)."""
    from graphchainer_amd.synth import SynthGraph
    from oracle import Oracle
    sg = SynthGraph(80_000, seed=53, multi_allelic=0.3, nested=0.3)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(10, 3000, seed=4)
    reads.append(reads[0][:300])                     # a short read: few fragments, likely untouched by the fragment limits
    graph = gca.AlignmentGraph(gfa)
    seeder = gca.MinimizerSeeder(graph)
    want = Oracle(gfa, long_pass=True).align(reads)
    flagged_total = 0
    # (gc_params::capacity where the knob is public; the retry's own size is a test hook in the environment)
    for env, cap in (({"GC_TEST_EXT_RETRY_MAX_ITEMS": "10"}, {"ext_max_items": 8}), ({}, {"long_max_items": 64}), ({}, {"long_cells_per_base": 2})):
        with monkeypatch.context() as m:
            for k, v in env.items():
                m.setenv(k, v)
            got = gca.Aligner(graph, seeder, long_pass=True, capacities=cap).align_reads(reads)     # must not raise
        env = (env, cap)
        flagged = np.asarray(got["capacity_exceeded"]).astype(bool)
        flagged_total += int(flagged.sum())
        for r in np.nonzero(~flagged)[0]:
            for off, key in (("read_anchor_off", "anchor_x"), ("read_anchor_off", "anchor_score"), ("read_chain_off", "chain"), ("read_longall_off", "longall_start"), ("read_longall_off", "longall_score")):
                assert np.array_equal(_per_read(got, off, key, r), _per_read(want, off, key, r)), (env, r, key)
            assert int(got["chain_edit_distance"][r]) == int(want["chain_edit_distance"][r]) and int(got["long_edit_distance"][r]) == int(want["long_edit_distance"][r])
    assert flagged_total >= 3


def test_shim_replays_the_reference_call_sequence(gca, tmp_path, golden_dir):
    """tests/shim/shim_test.cpp drives getSeeds -> OrderSeeds -> AlignOneWay (whole read, then every fragment) -> colinearChaining with the
    reference's signatures through include/graphchainer_amd_shim.hpp; what it sees equals the batch API's result for the same reads."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_library_exports import _build_shim_test
    exe = _build_shim_test(tmp_path)
    gfa = os.path.join(golden_dir, "syn20k.gfa")
    reads = [l.strip() for l in open(os.path.join(golden_dir, "syn20k.fa")) if not l.startswith(">")][:4]
    out = subprocess.run([exe, gfa] + reads, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    graph = gca.AlignmentGraph(gfa)
    res = gca.Aligner(graph, gca.MinimizerSeeder(graph), long_pass=True, keep_seeds=True).align_reads([r.encode() for r in reads])
    all_lines = out.stdout.strip().splitlines()
    # r4: AddGAFLine / AddAlignment / AddCorrected on the shim's items (src/GraphAlignerWrapper.h:43-45): every whole-read alignment's GAF line through the shim; the ones
    # the reference would write (the selection) must be the lines gc_format_gaf returns for the batch, and the vg message / corrected sequence exist for each
    gaf_lines = [l for l in all_lines if l.startswith("gaf ")]
    lines = [l for l in all_lines if l.startswith("read ")]
    traced = gca.Aligner(graph, gca.MinimizerSeeder(graph), long_pass=True, keep_traces=True).align_reads([r.encode() for r in reads], gaf_names=[f"r{i}" for i in range(len(reads))])
    shim_text = {l.split("\t", 1)[1] for l in gaf_lines}
    batch_lines = [l for l in traced["gaf"].decode().splitlines() if l]
    assert batch_lines and all(l in shim_text for l in batch_lines)
    assert len(gaf_lines) == int(traced["read_longall_off"][-1])
    for l in gaf_lines:
        head = l.split("\t", 1)[0].split()
        assert int(head[6]) > 0 and len(head[5]) == 16           # corrected letters; hash of the message bytes
    # r5: AddAlignment leaves DIGRAPH node ids and no names, as the reference's does (src/GraphAligner.h:205-212); the caller's own
    # replaceDigraphNodeIdsWithOriginalNodeIds (src/Aligner.cpp:152-165, called at :1009) - restated here on the decoded message - must give exactly the message the batch's GAM holds
    import gzip
    from vg_descriptor import alignment_class, decode_gam_stream
    Alignment = alignment_class()
    seg_names = [l.split("\t")[1] for l in open(gfa) if l.startswith("S\t")]             # first-appearance order = segment index (src/GfaGraph.cpp:146-174)
    shim_messages = {}
    for l in all_lines:
        if not l.startswith("vg "):
            continue
        _, r, i, hexbytes = l.split()
        msg = Alignment()
        msg.ParseFromString(bytes.fromhex(hexbytes))
        for m in msg.path.mapping:
            assert m.position.name == "" and (m.position.node_id & 1) == int(m.position.is_reverse)      # digraph ids: 2 x segment index + strand
            digraph_id = m.position.node_id                                                             # src/Aligner.cpp:156-162
            m.position.node_id = digraph_id // 2
            m.position.name = seg_names[digraph_id // 2]
        shim_messages.setdefault(int(r), []).append(msg.SerializeToString())
    gam = gca.Aligner(graph, gca.MinimizerSeeder(graph), long_pass=True, keep_traces=True).align_reads([r.encode() for r in reads], gaf_names=[f"r{i}" for i in range(len(reads))], formats=("gam",))["gam"]
    raw, at, batch_messages = gzip.decompress(gam), 0, []
    while at < len(raw):
        count, at = _read_varint(raw, at)
        for _ in range(count):
            size, at = _read_varint(raw, at)
            batch_messages.append(raw[at:at + size])
            at += size
    shim_all = {m for ms in shim_messages.values() for m in ms}
    assert batch_messages and all(m in shim_all for m in batch_messages)                                  # (the shim encodes every whole-read alignment, the batch writes the selected ones)
    assert len(lines) == len(reads)
    for r, line in enumerate(lines):
        head, chain = line.split(":", 2)[1:]
        f = head.split()
        n_seeds, n_long, n_anchors, n_chain = int(f[1]), int(f[3]), int(f[5]), int(f[7])
        assert n_seeds == res["read_seed_off"][r + 1] - res["read_seed_off"][r]
        assert n_long == res["read_longall_off"][r + 1] - res["read_longall_off"][r]
        assert n_anchors == res["read_anchor_off"][r + 1] - res["read_anchor_off"][r]
        assert [int(x) for x in chain.split()] == [int(c) for c in res["chain"][int(res["read_chain_off"][r]):int(res["read_chain_off"][r + 1])]] and n_chain == len(chain.split())


def test_single_process_two_devices_four_streams(gca, tmp_path):
    """the reference is ONE process with -t worker threads (src/Aligner.cpp:1267-1270); on a node with several GPUs that is a worker thread per gc_stream, a
    replica of the graph per device (gc_set_device + gc_index_load on the device's first thread) and one shared atomic batch cursor - examples/multi_gpu_host.cpp, C ABI only.
    Here: two LOGICAL devices (both on the box's one GPU: two gc_graph handles in one process) x two threads each, batches of 7 reads; every read's anchors, chain, chain score,
    whole-read alignments, both NW distances, decision and tie counts equal the oracle's, and both devices took batches."""
    import subprocess
    import sys
    from graphchainer_amd.synth import SynthGraph
    from oracle import Oracle
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_library_exports import _build_multi_gpu_host
    exe = _build_multi_gpu_host(tmp_path)
    sg = SynthGraph(200_000, seed=21)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(60, 3000, seed=5) + sg.sample_reads(10, 6000, seed=6, sv_fraction=0.5)
    with open(tmp_path / "reads.txt", "wb") as f:
        f.write(b"\n".join(reads) + b"\n")
    out = subprocess.run([exe, gfa, str(tmp_path / "reads.txt"), "2", "2", "7"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    assert len(lines) == len(reads) + 1
    want = Oracle(gfa, long_pass=True).align(reads)
    for r, line in enumerate(lines[:-1]):
        f = line.split()
        assert f[0] == "read" and int(f[1]) == r
        chain = want["chain"][int(want["read_chain_off"][r]):int(want["read_chain_off"][r + 1])]
        h = 0
        for c in chain:
            h = (h * 1000003 + int(c) + 1) % (1 << 64)
        expected = [int(want["read_anchor_off"][r + 1] - want["read_anchor_off"][r]), len(chain), h, int(want["chain_score"][r]), int(want["read_longall_off"][r + 1] - want["read_longall_off"][r]),
                    int(want["long_edit_distance"][r]), int(want["chain_edit_distance"][r]), int(want["chained_better"][r]), int(want["flatten_ties"][r]), int(want["flatten_ties_long"][r])]
        got = [int(f[3]), int(f[5]), int(f[6]), int(f[8]), int(f[10]), int(f[12]), int(f[13]), int(f[15]), int(f[17]), int(f[18])]
        assert got == expected, (r, line)
    per_device = [int(x) for x in lines[-1].split()[1:]]
    assert sum(per_device) == 10 and all(x > 0 for x in per_device), per_device


def _path_pairs(rng):
    """(path letters, read) shapes the chained alignment meets: similar strings, a path covering only part of the read, a path
    with a stretch the read lacks, unrelated strings, repeats (many optimal alignments), other letters, tiny / empty sides,
    and sizes on both sides of edlib's 1 MB traceback limit (Hirschberg splits, edlib/src/edlib.cpp:1204-1212)."""
    rand = lambda n, alphabet=b"ACGT": bytes(rng.choice(alphabet) for _ in range(n))
    for n in (0, 1, 2, 63, 64, 65, 200):
        q = rand(n)
        yield q, _mutate(rng, q, 0.15)
        yield q, rand(rng.randint(0, 130))
    for _ in range(12):
        q = rand(rng.randint(100, 900))
        yield q, _mutate(rng, q, rng.choice([0.02, 0.1, 0.3]))
        yield q, rand(400) + _mutate(rng, q, 0.1)
        yield q, _mutate(rng, q, 0.1) + rand(350)
        yield q, _mutate(rng, q[:len(q) // 3], 0.1) + _mutate(rng, q[2 * len(q) // 3:], 0.1)
    unit = rand(5)
    q = (unit * 200)[:700]
    yield q, _mutate(rng, q, 0.1)
    yield rand(300, b"AC"), rand(280, b"AC")
    q = rand(500, b"ACGTNRY")
    yield q, _mutate(rng, q, 0.1)
    yield rand(400), _mutate(rng, rand(400), 0.1).lower()
    for qn in (1500, 2600, 4200, 10_000):
        q = rand(qn)
        yield q, _mutate(rng, q, 0.12)
        yield q, rand(3000) + _mutate(rng, q, 0.1)
        yield q, _mutate(rng, q, 0.1) + rand(3300)
        yield q, rand(2000) + _mutate(rng, q, 0.08) + rand(2500)
        yield q, _mutate(rng, q[:qn // 3], 0.1) + _mutate(rng, q[2 * qn // 3:], 0.1)
        yield q[:300], rand(9000)
        yield (b"ACG" * 4000)[:qn], _mutate(rng, (b"ACG" * 4000)[:qn], 0.05)


def test_edit_path_kernel(gca):
    """k_edit_path (gc_edit_path) against the real edlib's EDLIB_TASK_PATH output (oracle/_ref) and the oracle's restatement:
    same distance, same op string, op for op."""
    import random
    from oracle import RefUnits
    from oracle.binding import oracle_edit_path
    rng = random.Random(99)
    pairs = list(_path_pairs(rng))
    dist, ops = gca.api.edit_path([a for a, _ in pairs], [b for _, b in pairs])
    try:
        ref = RefUnits()
    except (FileNotFoundError, OSError):
        ref = None
    hirschberg = 0
    for (a, b), d, o in zip(pairs, dist, ops):
        want_d, want_ops = oracle_edit_path(a, b)
        assert d == want_d, (len(a), len(b))
        assert np.array_equal(o, want_ops), (len(a), len(b), d, len(o), len(want_ops))
        if ref is not None:
            ref_d, ref_ops = ref.edit_path(a, b)
            assert d == ref_d and np.array_equal(o, ref_ops), (len(a), len(b))
        hirschberg += (20 * ((len(a) + 63) // 64) + 8) * len(b) >= 1 << 20
    assert hirschberg >= 15 and ref is not None


def _vg_alignment_class():
    """vg::Alignment (the fields the reference sets, src/vg.proto:52-154) built for the protobuf Python runtime: tests/vg_descriptor.py, checked on the CPU against the
    field table of the reference's own generated descriptor (tests/golden/vg_schema.expected.json)."""
    from vg_descriptor import alignment_class
    return alignment_class()


def _read_varint(buf, at):
    from vg_descriptor import read_varint
    return read_varint(buf, at)


def test_json_and_gam_output(gca, tmp_path):
    """vg::Alignment output: JSON lines against the oracle's restatement, and the GAM bytes (gzip members of framed proto3
    messages) decoded with the protobuf Python runtime against those same JSON objects."""
    import gzip
    import json
    from google.protobuf import json_format
    from graphchainer_amd.synth import SynthGraph
    from oracle import Oracle
    sg = SynthGraph(80_000, seed=23)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(6, 2500, seed=4)
    reads.append(_revcomp(reads[0]))
    reads.append(reads[1][:800] + reads[2][300:1500])
    reads.append(b"ACGTACGT")                      # no alignment: no output
    graph = gca.AlignmentGraph(gfa)
    seeder = gca.MinimizerSeeder(graph)
    aligner = gca.Aligner(graph, seeder, keep_traces=True, long_pass=True)
    got = aligner.align_reads(reads, gaf_names=[f"r{i}" for i in range(len(reads))], other_formats=True)
    ora = Oracle(gfa, long_pass=True)
    ora.align(reads)
    assert got["json"] == ora.json()
    objects = [json.loads(line) for line in got["json"].decode().splitlines()]
    assert len(objects) >= 7 and all(o["path"]["mapping"] for o in objects)
    # GAM: inflate all members, walk the groups, parse every message with protobuf itself
    Alignment = _vg_alignment_class()
    raw = gzip.decompress(got["gam"])
    decoded, at = [], 0
    while at < len(raw):
        count, at = _read_varint(raw, at)
        assert count >= 1
        for _ in range(count):
            size, at = _read_varint(raw, at)
            msg = Alignment()
            msg.ParseFromString(raw[at:at + size])
            at += size
            assert msg.SerializeToString() == raw[at - size:at]          # canonical proto3 bytes, nothing unknown
            decoded.append(json_format.MessageToDict(msg, preserving_proto_field_name=True))
    assert decoded == objects
    # r4: vg::Path bytes written by the device (gc_params::device_output & 4), wrapped by the host: the same inflated GAM stream byte for byte, the same JSON
    dev = gca.Aligner(graph, seeder, long_pass=True, device_output=1 | 4)
    got_dev = dev.align_reads(reads, gaf_names=[f"r{i}" for i in range(len(reads))], other_formats=True)
    assert gzip.decompress(got_dev["gam"]) == raw
    assert got_dev["json"] == got["json"]
    assert got_dev["gaf"] == got["gaf"]


def test_device_deflate_streams(gca):
    """hip/gc_deflate.hip through gc_gzip_streams: every member inflates (zlib's own inflate, CRC and length checked by gzip) to the stream it was made from - empty and
    one-byte streams, one repeated symbol, text-like and base-like streams (dynamic Huffman block), random bytes (stored blocks, more than one above 65535 bytes), and
    Fibonacci-weighted symbols whose Huffman code would be deeper than deflate's 15 bits (stored)."""
    import gzip
    import zlib
    rng = np.random.default_rng(5)
    fib = [1, 1]
    while len(fib) < 24:
        fib.append(fib[-1] + fib[-2])
    deep = b"".join(bytes([i]) * f for i, f in enumerate(fib))
    streams = [b"", b"A", b"AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA", b"AB" * 40, bytes(rng.integers(0, 256, 70_000, dtype=np.uint8)), bytes(rng.integers(0, 256, 131_071, dtype=np.uint8)),
               bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 50_001)), b"the quick brown fox jumps over the lazy dog " * 300, deep, bytes(range(256)) * 3,
               bytes(rng.choice(np.arange(256, dtype=np.uint8), 200_003, p=np.arange(1, 257) / np.arange(1, 257).sum()))]
    streams += [bytes(rng.choice(np.frombuffer(b"ACGTN\x08\x10\x12\x1a", dtype=np.uint8), int(n))) for n in rng.integers(1, 3000, 300)]
    members = gca.gzip_streams(streams)
    assert len(members) == len(streams)
    for raw, member in zip(streams, members):
        assert gzip.decompress(member) == raw
        d = zlib.decompressobj(31)
        assert d.decompress(member) == raw and d.eof and not d.unused_data     # exactly one member, nothing after it
    assert gzip.decompress(b"".join(members)) == b"".join(streams)              # concatenated members: what the GAM file is
    # the Huffman stage earns its keep on bases (2 bits per letter, 3 for the one that shares its subtree with the end-of-block symbol), random bytes are stored at 5 bytes per 65535
    assert len(members[6]) < 0.3 * 50_001 + 200 and len(members[4]) <= 70_000 + 10 + 18
    assert len(members[8]) <= len(deep) + 5 * (len(deep) // 65535 + 1) + 18      # the too-deep code fell back to stored blocks


def test_gam_deflated_on_device(gca, tmp_path):
    """gam_level = GC_GAM_DEVICE_HUFFMAN: the same inflated stream as the default level's, for host-encoded and device-encoded alignments and a chained winner."""
    import gzip
    from graphchainer_amd.synth import SynthGraph
    sg = SynthGraph(80_000, seed=23)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(12, 2500, seed=4)
    reads.append(reads[1][:800] + reads[2][300:1500])
    reads.append(b"ACGTACGT")                      # no alignment: no member
    names = [f"r{i}" for i in range(len(reads))]
    graph = gca.AlignmentGraph(gfa)
    seeder = gca.MinimizerSeeder(graph)
    for kw in ({"keep_traces": True}, {"device_output": 1 | 4}):
        aligner = gca.Aligner(graph, seeder, long_pass=True, **kw)
        want = aligner.align_reads(reads, gaf_names=names, formats=("gam",), gam_level=-1)["gam"]      # zlib's default level on the host, the reference's own setting
        got = aligner.align_reads(reads, gaf_names=names, formats=("gam",), gam_level=gca.GAM_DEVICE_HUFFMAN)["gam"]
        assert got != want and gzip.decompress(got) == gzip.decompress(want) and len(gzip.decompress(got)) > 30_000
        assert len(got) < 0.6 * len(gzip.decompress(got))
        # r6: LZ77 matches in front of the Huffman stage - the same inflated stream, within 1.35 x of zlib's default level where literal-only blocks are ~2 x
        lz = aligner.align_reads(reads, gaf_names=names, formats=("gam",), gam_level=gca.GAM_DEVICE_LZ)["gam"]
        assert gzip.decompress(lz) == gzip.decompress(want) and len(lz) < 0.8 * len(got) and len(lz) < 1.35 * len(want), (len(lz), len(got), len(want))
        assert aligner.align_reads(reads, gaf_names=names, formats=("gam",))["gam"] == lz                  # ... and what gc_format_gam gives when no level is asked for


def test_device_lz_deflate_equals_its_model(gca):
    """The device's LZ77 + dynamic-Huffman deflate (GC_GAM_DEVICE_LZ, hip/gc_deflate.hip) against tests/deflate_model.py, the same algorithm in plain Python whose output zlib
    inflates: every stream's member inflates to the stream, and its deflate bytes are the model's byte for byte - empty and tiny streams (stored blocks), one letter repeated
    (overlapping matches of the maximum length), random bytes (no matches: an empty distance alphabet), text with repeats, lengths around the 64-position chunks, a stream
    beyond one stored block's 65 535 bytes."""
    import gzip
    import random
    import sys
    import zlib
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from deflate_model import deflate
    rnd = random.Random(5)
    text = b"Mapping{position{node_id:%d offset:%d} edit{from_length:%d to_length:%d}} "
    streams = [b"", b"a", b"abc", b"abcd", b"A" * 5000, bytes(rnd.randrange(256) for _ in range(3000)),
               b"".join(text % (rnd.randrange(900), rnd.randrange(64), rnd.randrange(40), rnd.randrange(40)) for _ in range(400)),
               bytes(rnd.choice(b"ACGT") for _ in range(10_000)), b"xyz" * 21 + b"q", b"0123456789abcdef" * 4, b"0123456789abcdef" * 4 + b"0",
               bytes(rnd.choice(b"ACGT") for _ in range(40_000)) + b"".join(text % (i % 7, i % 5, 3, 3) for i in range(600))]
    for lz in (True, False):
        members = gca.gzip_streams(streams, lz=lz)
        assert len(members) == len(streams)
        for raw, member in zip(streams, members):
            assert gzip.decompress(member) == raw
    members = gca.gzip_streams(streams, lz=True)
    for raw, member in zip(streams, members):
        body = member[10:-8]                               # between the ten-byte gzip header and CRC-32 + length
        assert zlib.decompress(body, -15) == raw
        want, _tokens = deflate(raw) if raw else (b"", None)
        stored = len(raw) + 5 * max(1, (len(raw) + 65534) // 65535)
        if raw and len(want) < stored:
            assert body == want, (len(raw), len(body), len(want))
        else:
            assert len(body) == stored


def test_long_reads(gca, tmp_path, monkeypatch):
    """30 kb reads: per-extension scratch scales with the read length, persistent waves under a small scratch budget,
    edit-distance bands that need more than one block per lane."""
    from graphchainer_amd.synth import SynthGraph
    monkeypatch.setenv("GC_TEST_LONG_SCRATCH_GB", "1")
    sg = SynthGraph(300_000, seed=29)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(3, 30_000, seed=3, p_del=0.05, p_sub=0.06, p_ins=0.05)
    reads.append(sg.sample_reads(1, 30_000, seed=5)[0][:21_000])
    got, want = run_case(gca, gfa, reads, long_pass=True)
    compare(got, want, COMPARE_KEYS + LONG_KEYS)
    assert int(np.max(got["long_edit_distance"])) > 2016          # a band wider than one 64-row block per lane can hold


def test_output_against_golden_files(gca, golden_dir):
    """The encoders' output for the committed synthetic fixture against the committed text (tests/golden/syn20k.expected.*)."""
    reads = []
    for line in open(os.path.join(golden_dir, "syn20k.fa")):
        if not line.startswith(">"):
            reads.append(line.strip().encode())
    graph = gca.AlignmentGraph(os.path.join(golden_dir, "syn20k.gfa"))
    seeder = gca.MinimizerSeeder(graph)
    names = [f"r{i}" for i in range(len(reads))]
    out = gca.Aligner(graph, seeder, keep_traces=True, long_pass=True).align_reads(reads, gaf_names=names, other_formats=True)
    assert out["gaf"] == open(os.path.join(golden_dir, "syn20k.expected.gaf"), "rb").read()
    assert out["json"] == open(os.path.join(golden_dir, "syn20k.expected.json"), "rb").read()
    merged = gca.Aligner(graph, seeder, keep_traces=True, long_pass=True).align_reads(reads, gaf_names=names, cigar_match_mismatch_merge=True)
    assert merged["gaf"] == open(os.path.join(golden_dir, "syn20k.expected.merged.gaf"), "rb").read()
    # r4: encoded on the device
    import gzip
    dev = gca.Aligner(graph, seeder, long_pass=True, device_output=1 | 4).align_reads(reads, gaf_names=names, other_formats=True)
    assert dev["gaf"] == out["gaf"] and dev["json"] == out["json"] and gzip.decompress(dev["gam"]) == gzip.decompress(out["gam"])
    dev_merged = gca.Aligner(graph, seeder, long_pass=True, device_output=2).align_reads(reads, gaf_names=names, cigar_match_mismatch_merge=True)
    assert dev_merged["gaf"] == merged["gaf"]
    # r5: the pieces' CIGAR style is fixed when the batch is aligned; asking gc_format_gaf for the other one is refused, not answered with mixed styles
    for mode, merge in ((1, True), (2, False)):
        with pytest.raises(RuntimeError, match="cigar_match_mismatch_merge"):
            gca.Aligner(graph, seeder, long_pass=True, device_output=mode).align_reads(reads, gaf_names=names, cigar_match_mismatch_merge=merge)


def test_wave_sort_gives_libstdcxx_permutations(gca, tmp_path):
    """r5: the reference's three order-critical UNSTABLE std::sort calls (matches by count, seeds by goodness, seeds by read position) are replayed on the 64 lanes of a wave
    (csrc/hip/gc_stdsort_wave.hpp: stop-list partitions by the whole wave, independent ranges on one lane each, leaf-wise insertion) instead of on lane 0. Through the test
    entry gc_std_sort_permutations: the permutation equals the REAL std::sort's (tests/stdsort/std_sort_perm.cpp, built here with the local g++) on keys with many ties, saw-tooth
    and organ-pipe inputs, sizes on both sides of the LDS image (1 024) and of the cooperative threshold, and with small depth limits (introsort's heapsort path against
    libstdc++'s own __introsort_loop is the CPU test's part: tests/stdsort/stdsort_test.cpp checks the same decomposition step for step)."""
    import ctypes
    import subprocess
    so = str(tmp_path / "libstd_sort_perm.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(os.path.dirname(os.path.abspath(__file__)), "stdsort", "std_sort_perm.cpp")])
    ref = ctypes.CDLL(so)
    ref.std_sort_perm.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]

    def reference(keys):
        k = np.ascontiguousarray(keys, dtype=np.uint64)
        perm = np.zeros(len(k), dtype=np.int64)
        ref.std_sort_perm(k.ctypes.data, len(k), perm.ctypes.data)
        return perm

    rng = np.random.default_rng(9)
    arrays = []
    for n in list(range(0, 40)) + [63, 64, 65, 255, 256, 257, 258, 511, 700, 1023, 1024, 1025, 1500, 4097, 18_000, 70_000]:
        for mode in range(6):
            if mode == 0:
                k = np.ones(n)
            elif mode == 1:
                k = rng.integers(0, 3, n)
            elif mode == 2:
                k = rng.integers(0, 50, n)
            elif mode == 3:
                k = rng.integers(0, 2**32, n)
            elif mode == 4:
                k = np.arange(n) // 7
            else:
                k = np.minimum(np.arange(n), n - np.arange(n))          # organ pipe
            arrays.append(k.astype(np.uint32))
    for n in (5000, 20000):
        i = np.arange(n)
        arrays.append(np.where(i % 2 == 1, i, n - i).astype(np.uint32))   # saw-tooth: bad pivots, deep recursion
    got = gca.std_sort_permutations(arrays)
    for k, g in zip(arrays, got):
        want = reference(k)
        assert np.array_equal(g.astype(np.int64), want), (len(k), int(np.flatnonzero(g != want)[0]) if len(k) else -1)


def test_fragment_pools_sized_by_use_rerun_when_too_small(gca, tmp_path, monkeypatch):
    """r5: the fragment pipeline's trace pool and anchor path pool are sized by what the stream's batches have used (not by every slot's worst case: 26 GB per batch on a 960 Mbp
    graph); a batch that outgrows them runs the stage again with the room it asked for. GC_TEST_POOL_FIRST_GUESS makes a stream's first batch far too small: same results as the
    oracle, counters[6] says the stage ran again, and the stream's next batch (same Aligner) fits at once."""
    from graphchainer_amd.synth import SynthGraph
    from oracle import Oracle
    sg = SynthGraph(150_000, seed=5)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(40, 3000, seed=2)
    monkeypatch.setenv("GC_TEST_POOL_FIRST_GUESS", "0.5")
    monkeypatch.setenv("GC_TEST_POOL_SHRINK_FLOOR", "0")      # (the batch after a rerun gives back what the rerun's sizing overshot: here small pools do too)
    graph = gca.AlignmentGraph(gfa)
    seeder = gca.MinimizerSeeder(graph)
    aligner = gca.Aligner(graph, seeder, keep_traces=True, keep_seeds=True, long_pass=True, chain_traces=2)
    want = Oracle(gfa, long_pass=True).align(reads)
    reruns = []
    for _ in range(2):
        got = {k: (v.astype(np.int64) if v.dtype.kind in "ui" and k not in ("counters", "counters_long") else v) for k, v in aligner.align_reads(reads).items()}
        reruns.append(int(got["counters"][6]))
        expand_stitched_path(got, graph.array("nodeLength"))
        mark_missing_chain_alignments(got)
        compare(got, want)
        assert not got["capacity_exceeded"].any()
    assert reruns[0] >= 1 and reruns[1] == 0, reruns
    # a new stream whose first batch reruns, then a much smaller batch (the pools come back to what it needs), then the big one again (they grow): every time the oracle's results
    aligner = gca.Aligner(graph, seeder, keep_traces=True, keep_seeds=True, long_pass=True, chain_traces=2)
    few = reads[:3]
    want_few = Oracle(gfa, long_pass=True).align(few)
    for batch, expect in ((reads, want), (few, want_few), (reads, want), (few, want_few)):
        got = {k: (v.astype(np.int64) if v.dtype.kind in "ui" and k not in ("counters", "counters_long") else v) for k, v in aligner.align_reads(batch).items()}
        expand_stitched_path(got, graph.array("nodeLength"))
        mark_missing_chain_alignments(got)
        compare(got, expect)
        assert not got["capacity_exceeded"].any()


def test_flatten_tie_counts_equal_the_oracles(gca, tmp_path):
    """r5: both extension cores count the extensions whose backtrace started from a flattenLastSliceEnd minimum attained in more than one node -
    the only place where the reference's parallel-hashmap iteration order (absent here; band-entry order stands in) can choose another cell. gc_result::flatten_ties /
    flatten_ties_long per read equal the oracle's, in the lazy and the eager fragment pipeline and through the plain-layout fallback; the counts are not zero on 10 kb reads
    (about one tie per hundred extensions), so the comparison has something to compare."""
    from graphchainer_amd.synth import SynthGraph
    sg = SynthGraph(300_000, seed=7)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(48, 10_000, seed=11)
    got, want = run_case(gca, gfa, reads, long_pass=True)
    compare(got, want)
    compare(got, want, LONG_KEYS)
    assert int(want["flatten_ties"].sum()) >= 20 and int(want["flatten_ties_long"].sum()) >= 1
    assert int((got["flatten_ties"] + got["flatten_ties_long"] > 0).sum()) >= 10      # most 10 kb reads meet the rule at least once
    for env in ({"GC_EXT_LAZY": "0"}, {"GC_TEST_LONG_FORCE_FALLBACK": "1"}):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            again, _ = run_case(gca, gfa, reads[:16], long_pass=True)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        assert np.array_equal(again["flatten_ties"], want["flatten_ties"][:16]), env
        assert np.array_equal(again["flatten_ties_long"], want["flatten_ties_long"][:16]), env


@pytest.mark.parametrize("case", ["ref_test", "syn20k", "syn20k_more"])
def test_gam_against_the_reference_decoded_fixture(gca, case):
    """r5: the product's GAM - the host encoder over the traces, the device encoder's vg::Path bytes (k_out_encode) wrapped by the host, and the members deflated on the
    device - inflates to the stream the REFERENCE'S OWN descriptor and reader decoded (tests/golden/make_gam_golden.py: /root/reference/scripts/vg_pb2.py through the
    reader of scripts/summary.py:63-75), byte for byte, and decodes to the committed messages; one gzip member per read with output, as writeGAMToQueue frames them."""
    import gzip
    import zlib
    from vg_descriptor import decode_gam_stream, golden_case
    gfa, reads, want_groups, want_stream, want_better = golden_case(case)
    graph = gca.AlignmentGraph(gfa)
    seeder = gca.MinimizerSeeder(graph)
    names = [f"r{i}" for i in range(len(reads))]
    legs = {
        "host encoder": (dict(keep_traces=True), {}),
        "device encoder": (dict(device_output=1 | 4), {}),
        "device encoder, device deflate": (dict(device_output=1 | 4), dict(gam_level=gca.GAM_DEVICE_HUFFMAN)),
        "host encoder, device deflate": (dict(keep_traces=True), dict(gam_level=gca.GAM_DEVICE_HUFFMAN)),
    }
    for leg, (akw, fkw) in legs.items():
        out = gca.Aligner(graph, seeder, long_pass=True, **akw).align_reads(reads, gaf_names=names, formats=("gam",), **fkw)
        assert [int(x) for x in out["chained_better"]] == want_better, leg
        raw = gzip.decompress(out["gam"])
        assert raw == want_stream, leg
        assert decode_gam_stream(raw) == want_groups, leg
        # the framing: as many gzip members as reads with output, each inflating to one group
        members, rest = 0, out["gam"]
        while rest:
            d = zlib.decompressobj(31)
            group = d.decompress(rest)
            assert d.eof and len(decode_gam_stream(group)) == 1, leg
            rest = d.unused_data
            members += 1
        assert members == len(want_groups), leg


@pytest.mark.parametrize("case", ["ref_test", "syn20k", "syn20k_more"])
def test_json_lines_against_the_reference_decoded_fixture(gca, case):
    """r6: the JSON output (src/Aligner.cpp:286-293: one vg::Alignment per line through protobuf's JSON mapping) of the host encoder over the traces and of the device
    path (k_out_encode's vg::Path bytes, device_output 4) parses - with the field names and types of the descriptor - into the messages the reference's own descriptor
    decoded from the GAM of the same alignments (tests/golden/*.expected.gam.json), line for line in output order. What the fixture cannot hold is the C++ printer's
    spelling of `identity` (compared as a parsed double)."""
    from google.protobuf import json_format
    from vg_descriptor import alignment_class, golden_case
    gfa, reads, want_groups, _, want_better = golden_case(case)
    Alignment = alignment_class()
    graph = gca.AlignmentGraph(gfa)
    seeder = gca.MinimizerSeeder(graph)
    names = [f"r{i}" for i in range(len(reads))]
    want = [m for g in want_groups for m in g]
    for leg, akw in (("host encoder", dict(keep_traces=True)), ("device encoder", dict(device_output=1 | 4))):
        out = gca.Aligner(graph, seeder, long_pass=True, **akw).align_reads(reads, gaf_names=names, formats=("json",))
        lines = out["json"].split(b"\n")[:-1]
        assert len(lines) == len(want) >= 1, leg
        for line, message in zip(lines, want):
            got = json_format.MessageToDict(json_format.Parse(line.decode(), Alignment()), preserving_proto_field_name=True)
            assert got == message, (leg, line[:120])
        assert [int(x) for x in out["chained_better"]] == want_better, leg


@pytest.mark.parametrize("case", ["ref_test", "syn20k", "syn20k_more"])
def test_gam_paths_spelled_through_the_gfa_give_the_reported_distances(gca, case):
    """r6: the product's GAM read back the way the reference's harness reads it (scripts/summary.py:77-91; tests/golden/make_gam_golden.py committed what the restated
    reader saw through the reference's own descriptor): every alignment's path spelled through the GFA equals the fixture's, the NW edit distance (the product's own
    gc_edit_distance) of the part the alignment covers to the read equals the fixture's, and the distance the result reports for the read is that of its first
    selected / its chained alignment - output content, not schema."""
    import gzip
    from vg_descriptor import decode_gam_stream, golden_case, golden_paths, load_gfa_segments, spell_alignment
    gfa, reads, _, _, _ = golden_case(case)
    graph = gca.AlignmentGraph(gfa)
    seeder = gca.MinimizerSeeder(graph)
    names = [f"r{i}" for i in range(len(reads))]
    out = gca.Aligner(graph, seeder, long_pass=True, device_output=1 | 4).align_reads(reads, gaf_names=names, formats=("gam",))
    groups = decode_gam_stream(gzip.decompress(out["gam"]))
    want = golden_paths(case)
    VL = load_gfa_segments(gfa)
    assert len(groups) == len(want)
    parts, part_reads = [], []
    for group, row in zip(groups, want):
        assert len(group) == len(row["alignments"])
        for aln, want_aln in zip(group, row["alignments"]):
            got, part = spell_alignment(aln, VL)
            assert got == {k: want_aln[k] for k in got}, (case, row["read"])
            parts.append(part.encode())
            part_reads.append(reads[row["read"]])
    distances = [int(d) for d in gca.edit_distance(parts, part_reads)]
    assert distances == [a["nw_distance_to_read"] for row in want for a in row["alignments"]]
    for row in want:
        r = row["read"]
        reported = int(out["chain_edit_distance"][r]) if out["chained_better"][r] else int(out["long_edit_distance"][r])
        assert reported == row["reported_distance"], (case, r)


@pytest.mark.parametrize("env,kw,host_expected", [
    ({}, {}, "none"),
    ({"GC_STITCH_CLASS": "3"}, {}, "none"),          # r5: the long-read class (node set in HBM scratch behind an LDS filter) forced on short reads
    ({"GC_STITCH_CLASS": "3", "GC_TEST_STITCH_BFS_CAP": "2"}, {}, "none"),   # ... with an LDS search that may visit 2 nodes: every longer bridge search runs again in the scratch, none on the host
    ({"GC_STITCH_CLASS": "3", "GC_TEST_STITCH_BFS_CAP": "6"}, {"colinear_gap": 150}, "none"),
    ({"GC_STITCH_CLASS": "3"}, {"colinear_gap": -1}, "any"),
    ({"GC_HOST_STITCH": "1"}, {}, "all"),
    ({"GC_TEST_STITCH_BFS_CAP": "2"}, {}, "some"),        # a bridge search may visit 2 nodes: most reads fall back to the host
    ({"GC_TEST_STITCH_SET_MAX": "40"}, {}, "some"),       # a piece may hold 40 nodes
    ({}, {"colinear_gap": 150}, "none"),             # small --colinear-gap: bridges fail, chains break into pieces
    ({}, {"colinear_gap": -1}, "any"),               # no limit: a search for an unreachable anchor walks the whole graph downstream, on the host
    ({"GC_TEST_STITCH_BFS_CAP": "6"}, {"colinear_gap": 150}, "some"),
])
def test_chain_stitching_device_and_host(gca, tmp_path, monkeypatch, env, kw, host_expected):
    """Row f3: k_stitch against the oracle's stitching (src/Aligner.cpp:754-822); the host path that takes the reads the
    kernel's tables cannot hold gives the same paths, whichever reads it gets. Chimeric reads and a small --colinear-gap
    produce chains that break into several pieces."""
    from graphchainer_amd.synth import SynthGraph
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    sg = SynthGraph(150_000, seed=17)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(24, 4000, seed=31)
    reads.append(reads[0][:1500] + reads[1][500:2500])
    reads.append(reads[2][:900] + reads[3][2000:3500] + reads[4][100:1200])
    got, want = run_case(gca, gfa, reads, **kw)
    compare(got, want)
    chained = int(np.count_nonzero(np.diff(got["read_chain_off"])))
    on_host = int(got["counters"][7])
    assert chained >= 24
    if host_expected == "none":
        assert on_host == 0
    elif host_expected == "few":
        assert on_host <= 2
    elif host_expected == "all":
        assert on_host == chained
    elif host_expected == "some":
        assert 0 < on_host <= chained


def _normalise(out, node_length):
    got = {k: (v.astype(np.int64) if v.dtype.kind in "ui" and k not in ("counters", "counters_long") else v) for k, v in out.items()}
    expand_stitched_path(got, node_length)
    mark_missing_chain_alignments(got)
    sel = np.repeat(got["read_longall_off"][:-1], np.diff(got["read_long_off"])) + got["long_index"]
    for key in ("start", "end", "score"):
        got["long_" + key] = got["longall_" + key][sel]
    return got


@pytest.mark.parametrize("token", ["0", "1", "one", "two"])
def test_batches_in_flight_equal_serial_and_oracle(gca, tmp_path, monkeypatch, token):
    """The mode bench.py times: several gc_align_batch calls in flight on ONE device, each on its own gc_stream and host thread
    (run_queue with workers > 1, the reference's -t workers over one queue, src/Aligner.cpp:1267-1270), whole-read pass on, with the
    settings of the per-device whole-read token: none ("0"), per pass with the r5 rule ("1": batches this small may run two passes side by side, each in a scratch of its
    own), and the rule overridden to one or two tokens ("one", "two"; the per-round token lives in the experiments build). Four different read sets go through three
    Aligners concurrently, twice each; every result array must equal the oracle's AND the same Aligner's serial answer."""
    from graphchainer_amd.synth import SynthGraph
    from graphchainer_amd.workqueue import ReadQueue, run_queue
    from oracle import Oracle
    monkeypatch.setenv("GC_LONG_TOKEN", "0" if token == "0" else "1")
    if token in ("one", "two"):
        monkeypatch.setenv("GC_LONG_TOKENS", "1" if token == "one" else "2")
    sg = SynthGraph(300_000, seed=17, repeats=3, repeat_len=1500)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    read_sets = [sg.sample_reads(90, 4000, seed=31), sg.sample_reads(70, 6000, seed=32),
                 sg.sample_reads(40, 10_000, seed=33, sv_fraction=0.3), sg.sample_reads(120, 2500, seed=34)]
    read_sets[1].append(read_sets[1][0][:2000] + read_sets[1][3][1000:4000])      # a chimera: several whole-read alignments
    graph = gca.AlignmentGraph(gfa)
    seeder = gca.MinimizerSeeder(graph)
    node_length = graph.array("nodeLength")
    kw = dict(keep_traces=True, keep_seeds=True, long_pass=True, chain_traces=2)
    ora = Oracle(gfa, long_pass=True)
    want = [ora.align(rs) for rs in read_sets]
    batches = [gca.ReadBatch(rs) for rs in read_sets]
    aligners = [gca.Aligner(graph, seeder, **kw) for _ in range(3)]
    serial = [_normalise(aligners[0].align_batch(b), node_length) for b in batches]
    for s, w in zip(serial, want):
        compare(s, w, COMPARE_KEYS + LONG_KEYS)
    items = 2 * len(batches)
    queue = ReadQueue(items)
    outs = run_queue(queue, lambda worker, item: aligners[worker].align_batch(batches[item % len(batches)]), workers=3)
    assert sorted(i for i, _ in outs) == list(range(items))
    for item, out in outs:
        got = _normalise(out, node_length)
        compare(got, want[item % len(batches)], COMPARE_KEYS + LONG_KEYS)
        compare(got, serial[item % len(batches)], COMPARE_KEYS + LONG_KEYS)
    assert sum(int(np.sum(w["chained_better"])) for w in want) > 0


@pytest.mark.parametrize("n_ranks", [2, 4])
def test_two_ranks_real_aligner_strong_queue(gca, tmp_path, n_ranks):
    """BASELINE configs[3]'s shape with the REAL aligner: two (r4: and four) ranks (torch.distributed.run, gloo for the barrier and the queue
    reset only), ONE read set divided by the product's flock'ed work queue, every rank running gc_align_batch on its own two
    gc_streams (the ranks share this box's one GPU: a dry run of the launch path, not a measurement), per-read results merged
    over the ranks and compared with the oracle's. The launcher starts before anything touches the GPU."""
    import subprocess
    import sys
    import textwrap
    from graphchainer_amd.synth import SynthGraph
    from oracle import Oracle
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sg = SynthGraph(200_000, seed=23)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(60, 3000, seed=5) + sg.sample_reads(40, 6000, seed=6, sv_fraction=0.25)
    want = Oracle(gfa, long_pass=True).align(reads)
    np.savez(tmp_path / "reads.npz", blob=np.frombuffer(b"".join(reads), dtype=np.uint8), lens=np.array([len(r) for r in reads]))
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {root!r})
        import numpy as np
        import torch
        import torch.distributed as dist
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        import graphchainer_amd as gca
        from graphchainer_amd.workqueue import ReadQueue, length_sorted_batches, merge_read_results, run_queue
        gca.set_device(0)
        z = np.load({str(tmp_path / "reads.npz")!r})
        cuts = np.concatenate([[0], np.cumsum(z["lens"])])
        blob = z["blob"].tobytes()
        reads = [blob[cuts[i]:cuts[i + 1]] for i in range(len(z["lens"]))]
        graph = gca.AlignmentGraph({gfa!r})
        seeder = gca.MinimizerSeeder(graph)
        aligners = [gca.Aligner(graph, seeder, long_pass=True) for _ in range(2)]
        chunks = length_sorted_batches(reads, 16)
        batches = [gca.ReadBatch([reads[i] for i in idx]) for idx in chunks]
        queue = ReadQueue(len(batches), rank, world, dist, path={str(tmp_path / "queue.bin")!r})
        keys = ["chain_score", "chain_edit_distance", "long_edit_distance", "chained_better", "n_chain", "n_anchor", "n_long", "long_sum"]
        for step in range(2):
            queue.reset()
            parts = []
            for b, out in run_queue(queue, lambda w, b: aligners[w].align_batch(batches[b]), workers=2):
                rec = {{k: np.asarray(out[k]).astype(np.int64) for k in keys[:4]}}
                rec["n_chain"] = np.diff(out["read_chain_off"].astype(np.int64))
                rec["n_anchor"] = np.diff(out["read_anchor_off"].astype(np.int64))
                off = out["read_longall_off"].astype(np.int64)
                rec["n_long"] = np.diff(off)
                csum = np.concatenate([[0], np.cumsum(out["longall_start"].astype(np.int64) * 3 + out["longall_end"].astype(np.int64) * 5 + out["longall_score"].astype(np.int64) * 7)])
                rec["long_sum"] = csum[off[1:]] - csum[off[:-1]]
                parts.append((b, rec))
            merged = np.stack([merge_read_results(parts, chunks, len(reads), k, fill=0) for k in keys])
            # every read is aligned by exactly one rank: the other rank's rows are the fill value 0, so a sum over ranks reassembles
            # them (the -1 "no distance" values survive: one rank reports -1, the other 0)
            t = torch.from_numpy(merged.copy())
            count = torch.tensor([len(parts)])
            dist.all_reduce(t); dist.all_reduce(count)
            if rank == 0:
                assert int(count.item()) == len(batches)
                np.save({str(tmp_path)!r} + f"/merged{{step}}.npy", t.numpy())
        queue.close()
        dist.barrier()
        if rank == 0:
            print("RANKS_OK")
        dist.destroy_process_group()
    """))
    port = str(29577 + n_ranks)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, GC_HOST_THREADS="4", GPU_MAX_HW_QUEUES="16")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1", "--master-port", port, str(script)],
                         capture_output=True, text=True, env=env, timeout=900)
    assert "RANKS_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
    off = want["read_longall_off"]
    csum = np.concatenate([[0], np.cumsum(want["longall_start"] * 3 + want["longall_end"] * 5 + want["longall_score"] * 7)])
    expect = np.stack([want["chain_score"], want["chain_edit_distance"], want["long_edit_distance"], want["chained_better"],
                       np.diff(want["read_chain_off"]), np.diff(want["read_anchor_off"]), np.diff(off), csum[off[1:]] - csum[off[:-1]]])
    for step in range(2):
        merged = np.load(tmp_path / f"merged{step}.npy")
        assert np.array_equal(merged, expect), f"step {step}: rows {np.nonzero((merged != expect).any(axis=1))[0]} differ from the oracle"
    assert int(np.sum(want["chained_better"])) > 0 and int(np.diff(want["read_chain_off"]).min()) >= 0


@pytest.mark.parametrize("k,w", [(19, 30), (31, 31), (16, 25)])
def test_minimizer_lengths_above_15(gca, tmp_path, k, w):
    """--seeds-minimizer-length goes up to 31 in the reference (src/AlignerMain.cpp:221,390, src/MinimizerSeeder.cpp:63): k-mers beyond 30 bits are
    looked up through the 32-bit tag of the hash slot plus a check of the key's whole k-mer; the index for them is built on the host. Seeds,
    anchors, chains and whole-read alignments against the oracle run with the same k and w."""
    from graphchainer_amd.synth import SynthGraph
    from oracle import Oracle
    sg = SynthGraph(150_000, seed=19)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(10, 6000, seed=4, p_del=0.01, p_sub=0.015, p_ins=0.01)   # (long k-mers need cleaner reads to hit)
    graph = gca.AlignmentGraph(gfa)
    seeder = gca.MinimizerSeeder(graph, minimizer_length=k, window_size=w)
    aligner = gca.Aligner(graph, seeder, keep_traces=True, keep_seeds=True, long_pass=True, chain_traces=2)
    got = _normalise(aligner.align_reads(reads), graph.array("nodeLength"))
    want = Oracle(gfa, k=k, w=w, long_pass=True).align(reads)
    compare(got, want, COMPARE_KEYS + LONG_KEYS)
    assert int(got["read_seed_off"][-1]) > 50
