"""The product's first build of a graph (AlignmentGraph::BuildFromGFAFile, csrc/host/gc_graph_fast.cpp: flat arrays filled by several threads, the reference's hash-table
iteration orders replayed by gc::HashOrder) against the literal builder (GfaGraph::LoadFromFile + AlignmentGraph::BuildFromGFA, csrc/host/gc_graph.cpp: the reference's own
container types and insertion sequences, src/GfaGraph.cpp:212-370, src/BigraphToDigraph.cpp:215-267, src/AlignmentGraph.cpp:51-307) - which tests/test_graph_model.py and the
oracle pin from their side. tests/graph_build/build_compare.cpp builds a file both ways and compares every array, or the two error messages when the file is refused."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "graphchainer_amd", "csrc", "host")


@pytest.fixture(scope="module")
def compare(tmp_path_factory):
    exe = tmp_path_factory.mktemp("graph_build") / "build_compare"
    subprocess.run(["g++", "-std=c++17", "-O2", "-I" + HOST, os.path.join(ROOT, "tests", "graph_build", "build_compare.cpp"), os.path.join(HOST, "gc_graph.cpp"),
                    os.path.join(HOST, "gc_graph_fast.cpp"), "-o", str(exe), "-lpthread"], check=True, timeout=900)

    def run(path, threads=None):
        env = dict(os.environ)
        if threads is not None:
            env["GC_BUILD_THREADS"] = str(threads)
        out = subprocess.run([str(exe), str(path)], capture_output=True, text=True, timeout=900, env=env)
        assert out.returncode == 0 and out.stdout.startswith("SAME"), (str(path), out.stdout + out.stderr)
        return out.stdout
    return run


def test_golden_and_synthetic_graphs(compare, tmp_path):
    from graphchainer_amd.synth import SynthGenome, SynthGraph
    for name in ("syn20k.gfa", "ref_test_graph.gfa"):
        assert "SAME graph" in compare(os.path.join(ROOT, "tests", "golden", name))
    sg = SynthGraph(400_000, seed=5, repeats=3, repeat_len=900)
    sg.write_gfa(str(tmp_path / "a.gfa"))
    genome = SynthGenome(5, 150_000, seed=9, multi_allelic=0.1, nested=0.1, minus_links=0.3, repeats=2, repeat_len=1200)
    genome.write_gfa(str(tmp_path / "b.gfa"))
    for gfa in ("a.gfa", "b.gfa"):
        for threads in (1, 3, 8):          # the result does not depend on the number of threads
            assert "SAME graph" in compare(tmp_path / gfa, threads)


def _write(path, lines, newline="\n", last_newline=True):
    text = newline.join(lines) + (newline if last_newline else "")
    path.write_bytes(text.encode())
    return path


def test_awkward_files(compare, tmp_path):
    rng = np.random.default_rng(3)

    def seq(n, letters="ACGT"):
        return "".join(rng.choice(list(letters), size=int(n)))
    cases = {
        # names: words, numbers with leading zeros (a different segment than the plain number), beyond 2^31, signs; fields after the sequence; P / H lines; blank lines
        "names": ["H\tVN:Z:1.0", "S\tchr1_a\t" + seq(70) + "\tLN:i:70", "S\t007\t" + seq(5), "S\t7\t" + seq(64), "S\t99999999999\t" + seq(65), "", "S\t-5\t" + seq(3), "S\t+\t" + seq(2),
                  "L\tchr1_a\t+\t007\t+\t0M", "L\t007\t+\t7\t-\t0M", "L\t7\t-\t99999999999\t+\t0M\tXX:i:3", "L\t-5\t-\t+\t-\t0M", "P\tpath\tchr1_a+,007+\t*", "L\tchr1_a\t+\t7\t+\t*"],
        # links before their segments, links of segments that never come (source, target, both), the same link twice, a segment given twice (the second sequence counts)
        "order": ["L\t1\t+\t2\t+\t0M", "L\t9\t+\t1\t+\t0M", "L\t2\t+\t8\t-\t0M", "L\t8\t+\t9\t+\t0M", "S\t2\t" + seq(130), "S\t1\t" + seq(64), "L\t1\t+\t2\t+\t0M", "L\t2\t-\t1\t-\t0M",
                  "S\t3\tACGT", "S\t3\t" + seq(200), "L\t2\t+\t3\t+\t0M", "L\t3\t+\t3\t+\t0M", "L\t1\t-\t3\t+\t0M"],
        # letters: lower case, U, every IUPAC code (ambiguous split nodes move to the end), pieces of exactly 64 / 65 / 128 / 129 letters
        "letters": ["S\t1\t" + seq(64).lower(), "S\t2\t" + seq(65, "ACGTUacgtu"), "S\t3\t" + seq(128, "ACGTRYSWKMBDHVN"), "S\t4\t" + seq(129, "acgtryswkmbdhvn"), "S\t5\tN", "S\t6\t" + seq(300),
                    "L\t1\t+\t2\t+\t0M", "L\t2\t+\t3\t-\t0M", "L\t3\t-\t4\t+\t0M", "L\t4\t+\t5\t+\t0M", "L\t5\t+\t6\t-\t0M", "L\t6\t+\t1\t+\t0M"],
        # spaces and form feeds between fields, a tab at the end of a line
        "blanks": ["S 1  " + seq(10) + "\t", "S\t\t2\t" + seq(80), "L 1 +\t2\f+ 0M", "S\t3\t" + seq(3) + " ", "L\t2\t+\t3\t+\t0M\t"],
        # a cigar without a number / with a sign / with other units: overlap 0 as a failed extraction leaves it
        "cigars": ["S\t1\tACGT", "S\t2\tACGT", "S\t3\tACGT", "L\t1\t+\t2\t+\tM", "L\t2\t+\t3\t+\t+0M", "L\t1\t+\t3\t+\t0S", "L\t3\t+\t1\t-"],
        "only_links": ["L\t1\t+\t2\t+\t0M", "L\t2\t+\t3\t+\t0M"],
        "empty": [],
    }
    for name, lines in cases.items():
        assert "SAME graph" in compare(_write(tmp_path / (name + ".gfa"), lines)), name
    # Windows line ends; a last line without a line end is not read (src/GfaGraph.cpp:219-223)
    assert "SAME graph" in compare(_write(tmp_path / "crlf.gfa", cases["order"], newline="\r\n"))
    out = compare(_write(tmp_path / "cut.gfa", ["S\t1\tACGT", "S\t2\tACGTA", "L\t1\t+\t2\t+\t0M"], last_newline=False))
    assert "SAME graph: 4 split nodes" in out        # (the link's line is the one that is not read: two segments, two strands each)
    # many segments with shuffled, sparse numbers: the container orders across several rehashes
    ids = rng.permutation(200_000)[:30_000] + 1
    lines = [f"S\t{i}\t{seq(rng.integers(1, 90))}" for i in ids] + [f"L\t{a}\t{'+-'[int(rng.integers(2))]}\t{b}\t{'+-'[int(rng.integers(2))]}\t0M" for a, b in zip(ids[:-1], ids[1:])]
    order = rng.permutation(len(lines))
    assert "SAME graph" in compare(_write(tmp_path / "shuffled.gfa", [lines[i] for i in order]))


def test_refused_files_are_refused_alike(compare, tmp_path):
    good = ["S\t1\tACGT", "S\t2\tACGT", "L\t1\t+\t2\t+\t0M"]
    cases = {
        "star": good + ["S\t3\t*"],
        "no_sequence": good + ["S\t3"],
        "no_name": ["S"] + good,
        "orientation": good + ["L\t1\tx\t2\t+\t0M"],
        "orientation_missing": good + ["L\t1\t+\t2"],
        "negative_overlap": good + ["L\t2\t+\t1\t+\t-3M"],
        "overlap": good + ["L\t2\t+\t1\t+\t3M"],
        "letter": good + ["S\t3\tACGTXACGT", "S\t4\tAC-GT"],
        "two_errors": ["L\t1\t+\t2\t+\t-1M", "S\t3\t*"],               # the first one in the file is reported
        "error_then_overlap": ["L\t1\t+\t2\t+\t5M", "S\t3\t*"],         # parse errors come before the overlap check of the build
        "overlap_and_letter": good + ["S\t3\tAXGT", "L\t2\t+\t3\t+\t1M"],
    }
    for name, lines in cases.items():
        out = compare(_write(tmp_path / (name + ".gfa"), lines))
        assert out.startswith("SAME error"), (name, out)
