"""The independent model of AlignOneWay, the two-directional trace and the anchors (tests/alignment_model.py over tests/extension_model.py and tests/seeding_model.py:
rows A1-A11 of SURVEY.md §8 read a second time, from the reference) against the oracle: the whole-read alignments with every trace cell, the seeds extended, and the
anchors with their paths, end cells and scores."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import seeding_model                                        # noqa: E402
from alignment_model import AlignmentModel                  # noqa: E402
from extension_model import ExtensionModel, Graph           # noqa: E402
from graphchainer_amd.synth import SynthGraph              # noqa: E402  (test inputs)
from test_seeding_model import _inputs, std_sort            # noqa: E402,F401  (the fixture)


def _models(oracle, bandwidth):
    length = oracle.graph_array("nodeLength").tolist()
    flat = oracle.graph_array("sequence")
    seq, at = [], 0
    for n in length:
        seq.append("".join(chr(c) for c in flat[at:at + n]))
        at += n

    def csr(off, adj):
        off, adj = oracle.graph_array(off).tolist(), oracle.graph_array(adj).tolist()
        return [adj[off[i]:off[i + 1]] for i in range(len(length))]
    node_ids, node_offset = oracle.graph_array("nodeIDs").tolist(), oracle.graph_array("nodeOffset").tolist()
    g = Graph(length, seq, csr("out_off", "out_adj"), csr("in_off", "in_adj"), oracle.graph_array("componentNumber").tolist(),
              [bool(x) for x in oracle.graph_array("linearizable")], node_ids, node_offset)
    original_size = {}
    for v, big in enumerate(node_ids):
        original_size[big] = max(original_size.get(big, 0), node_offset[v] + length[v])
    return AlignmentModel(ExtensionModel(g, bandwidth), g, original_size)


def _check(gfa, reads, std_sort, whole_read=True):
    from oracle import Oracle
    oracle = Oracle(gfa, long_pass=whole_read)
    want = oracle.align(reads)
    graph, index = _inputs(oracle)
    model = _models(oracle, 10)
    alignments = anchors = cells = 0
    for r, read in enumerate(reads):
        if want["failed_assertion"][r]:
            continue
        seeds = seeding_model.order_seeds_by_chaining(seeding_model.get_seeds(read, index, graph, 15, 20, 10.0, std_sort), graph, std_sort)
        if whole_read and seeds:
            got, extended = model.align_one_way(read, seeds, True)
            a0, a1 = int(want["read_longall_off"][r]), int(want["read_longall_off"][r + 1])
            assert len(got) == a1 - a0, (r, len(got), a1 - a0)
            for k, aln in enumerate(got):
                a = a0 + k
                assert (aln["start"], aln["end"], aln["score"]) == (int(want["longall_start"][a]), int(want["longall_end"][a]), int(want["longall_score"][a])), (r, k)
                t0, t1 = int(want["long_trace_off"][a]), int(want["long_trace_off"][a + 1])
                exp = list(zip(want["long_trace_node"][t0:t1].tolist(), want["long_trace_offset"][t0:t1].tolist(), want["long_trace_seqpos"][t0:t1].tolist(), [bool(x) for x in want["long_trace_switch"][t0:t1]]))
                assert [tuple(c) for c in aln["trace"]] == exp, (r, k)
                cells += len(exp)
            alignments += len(got)
        by_position = seeding_model.fragment_order(seeds, std_sort)
        found = model.anchors_of_read(read, by_position)
        b0, b1 = int(want["read_anchor_off"][r]), int(want["read_anchor_off"][r + 1])
        assert len(found) == b1 - b0, (r, len(found), b1 - b0)
        for k, (x, y, path, first, last, score) in enumerate(found):
            b = b0 + k
            assert (x, y, score) == (int(want["anchor_x"][b]), int(want["anchor_y"][b]), int(want["anchor_score"][b])), (r, k)
            assert path == want["anchor_path"][int(want["anchor_path_off"][b]):int(want["anchor_path_off"][b + 1])].tolist(), (r, k)
            assert first == (int(want["anchor_first_node"][b]), int(want["anchor_first_offset"][b]), int(want["anchor_first_seqpos"][b])), (r, k)
            assert last == (int(want["anchor_last_node"][b]), int(want["anchor_last_offset"][b]), int(want["anchor_last_seqpos"][b])), (r, k)
        anchors += len(found)
    return alignments, cells, anchors


def test_whole_read_alignments_and_anchors_on_the_golden_graph(std_sort):   # noqa: F811
    gold = os.path.join(ROOT, "tests", "golden")
    reads = [l.strip().encode() for l in open(os.path.join(gold, "syn20k.fa")) if not l.startswith(">")][:4]
    alignments, cells, anchors = _check(os.path.join(gold, "syn20k.gfa"), reads, std_sort)
    assert alignments >= 4 and cells > 3000 and anchors > 40, (alignments, cells, anchors)


def test_reverse_strand_chimeric_and_noisy_reads(tmp_path, std_sort):   # noqa: F811
    """Reverse-strand reads (the backward extension runs on the forward strand), a chimeric read (two alignments, the overlap rule), reads with a deletion the graph does not
    hold (the whole-read pass stops at the breakpoint and other seeds are extended), noise (seeds off the first alignment's trace)."""
    sg = SynthGraph(40_000, seed=37)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(3, 1200, seed=6)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    bb = sg.backbone.tobytes()
    reads += [reads[0].translate(comp)[::-1], reads[1][:500] + reads[2][300:900], bb[5000:5600] + bb[7100:7700]]
    reads += sg.sample_reads(2, 900, seed=8, p_del=0.06, p_sub=0.06, p_ins=0.06)
    alignments, cells, anchors = _check(gfa, reads, std_sort)
    assert alignments >= 9 and cells > 6000 and anchors > 100, (alignments, cells, anchors)
