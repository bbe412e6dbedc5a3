"""The independent seeding model (tests/seeding_model.py: rows A1, A2 of SURVEY.md §8 read a second time, from the reference) against the oracle's seeds:
same seeds in the same order with the same goodness on the golden graphs, a graph with repeats (many hits per k-mer, ties) and reads with N runs."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from graphchainer_amd.synth import SynthGraph   # noqa: E402  (the generator of test inputs, not the product path)
import seeding_model as model                   # noqa: E402


@pytest.fixture(scope="module")
def std_sort(tmp_path_factory):
    """libstdc++'s std::sort as a permutation function (tests/stdsort/std_sort_perm.cpp, built here with the local g++)."""
    so = str(tmp_path_factory.mktemp("stdsort") / "libstd_sort_perm.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "stdsort", "std_sort_perm.cpp")])
    lib = ctypes.CDLL(so)
    lib.std_sort_perm.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]

    def sort(keys):
        k = np.asarray(keys, dtype=np.uint64)
        perm = np.zeros(len(k), dtype=np.int64)
        if len(k):
            lib.std_sort_perm(k.ctypes.data, len(k), perm.ctypes.data)
        return perm.tolist()
    return sort


def _inputs(oracle):
    graph = {name: oracle.graph_array(name).tolist() for name in ("nodeIDs", "nodeOffset", "reverse", "chainNumber", "chainApproxPos")}
    index = {"kmers": oracle.graph_array("index_kmers").astype(np.uint64).tolist(), "start": oracle.graph_array("index_start").tolist(),
             "positions": oracle.graph_array("index_positions").astype(np.uint64).tolist(), "maxcount": int(oracle.graph_array("index_maxcount")[0])}
    return graph, index


def _check(gfa, reads, std_sort, k=15, w=20):
    from oracle import Oracle
    oracle = Oracle(gfa, long_pass=False, k=k, w=w)
    want = oracle.align(reads)
    graph, index = _inputs(oracle)
    compared = ties = 0
    for r, read in enumerate(reads):
        seeds = model.get_seeds(read, index, graph, k, w, 10.0, std_sort)
        seeds = model.fragment_order(model.order_seeds_by_chaining(seeds, graph, std_sort), std_sort)
        s0, s1 = int(want["read_seed_off"][r]), int(want["read_seed_off"][r + 1])
        assert len(seeds) == s1 - s0, (r, len(seeds), s1 - s0)
        got = np.array([[s["agNode"], s["agOffset"], s["seqPos"], s["goodness"]] for s in seeds], dtype=np.int64).reshape(-1, 4)
        exp = np.stack([want["seed_node"][s0:s1], want["seed_offset"][s0:s1], want["seed_seqpos"][s0:s1], want["seed_goodness"][s0:s1]], axis=1)
        assert np.array_equal(got, exp), (r, np.nonzero((got != exp).any(axis=1))[0][:5])
        compared += len(seeds)
        ties += len(seeds) - len({(s["seqPos"]) for s in seeds})
    return compared, ties


def test_kmer_iteration_rules():
    """iterateKmers: every k-mer is reported unless it repeats the last reported one inside the window; a non-ACGT letter restarts; a read shorter than k reports nothing."""
    assert model.iterate_kmers(b"ACG", 4, 6) == []
    assert [p for p, _ in model.iterate_kmers(b"ACGTACGT", 4, 6)] == [3, 4, 5, 6, 7]
    same = model.iterate_kmers(b"A" * 12, 4, 6)                 # one k-mer all along: reported again once the last report has left the window of w - k + 1
    assert [p for p, _ in same] == [3, 6, 9]
    with_n = model.iterate_kmers(b"ACGTNACGTA", 4, 6)
    assert [p for p, _ in with_n] == [3, 8, 9] and with_n[0][1] == with_n[1][1]
    assert model.iterate_kmers(b"acgt", 4, 6) == model.iterate_kmers(b"ACGT", 4, 6)


def test_seeds_equal_the_oracle_on_the_golden_graphs(std_sort):
    gold = os.path.join(ROOT, "tests", "golden")
    reads = [l.strip().encode() for l in open(os.path.join(gold, "syn20k.fa")) if not l.startswith(">")]
    compared, _ = _check(os.path.join(gold, "syn20k.gfa"), reads, std_sort)
    assert compared > 400
    ref_reads = [l.strip().encode() for l in open(os.path.join(gold, "ref_test_read.fa")) if not l.startswith(">")]
    _check(os.path.join(gold, "ref_test_graph.gfa"), ref_reads, std_sort)


def test_seeds_equal_the_oracle_with_repeats_ties_and_n_runs(tmp_path, std_sort):
    """Repeats put several positions behind one k-mer (the count sort's ties, the density cut-off inside a group of equal counts, clusters on several chains); N runs restart the
    k-mer scan; a second minimizer shape moves the thinning window."""
    sg = SynthGraph(150_000, seed=31, repeats=9, repeat_len=1500)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(12, 4000, seed=4)
    with_n = bytearray(reads[0])
    with_n[500:520] = b"N" * 20
    with_n[1500] = ord("N")
    reads += [bytes(with_n), reads[1].lower(), reads[2][:40], b"ACGT" * 30]
    compared, ties = _check(gfa, reads, std_sort)
    assert compared > 2500 and ties > 50, (compared, ties)
    _check(gfa, reads[:6], std_sort, k=11, w=15)


def test_the_tie_order_of_std_sort_is_what_the_check_sees(tmp_path, std_sort):
    """Mutation check: with a stable sort in place of libstdc++'s std::sort the model's seed order differs from the oracle's on the repeat graph - the comparison above
    does exercise the tie orders (the count sort, the goodness sort, the position sort), it does not pass by having none."""
    from oracle import Oracle
    sg = SynthGraph(150_000, seed=31, repeats=9, repeat_len=1500)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(12, 4000, seed=4)
    oracle = Oracle(gfa, long_pass=False)
    want = oracle.align(reads)
    graph, index = _inputs(oracle)
    stable = lambda keys: sorted(range(len(keys)), key=lambda i: keys[i])
    differing = 0
    for r, read in enumerate(reads):
        seeds = model.fragment_order(model.order_seeds_by_chaining(model.get_seeds(read, index, graph, 15, 20, 10.0, stable), graph, stable), stable)
        s0, s1 = int(want["read_seed_off"][r]), int(want["read_seed_off"][r + 1])
        got = [(s["agNode"], s["agOffset"], s["seqPos"]) for s in seeds]
        exp = list(zip(want["seed_node"][s0:s1].tolist(), want["seed_offset"][s0:s1].tolist(), want["seed_seqpos"][s0:s1].tolist()))
        differing += got != exp
    assert differing >= 3
