"""Generates the committed golden fixtures from the CPU oracle (run here, results committed):

  ref_test_*      - the reference's own data files test/graph.gfa + test/read.fa (data, copied verbatim)
  syn20k.gfa/.fa  - a 20 kbp synthetic SNP/indel-bubble graph with 6 reads of 2 kb (seeded generator)
  *.expected.npz  - the oracle's flat result arrays for those inputs
  syn20k.expected.{gaf,merged.gaf,json} - the oracle's output-encoder text for the synthetic reads

The reference ships no expected outputs and cannot be built here, so these vectors pin the
oracle against regressions, not against the reference; the one reference-derived value is the anchor recorded
in SURVEY.md §8c for test/graph.gfa (checked in tests/test_oracle_golden.py).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from graphchainer_amd.synth import SynthGraph, write_fasta  # noqa: E402
from oracle import Oracle  # noqa: E402


def main():
    sg = SynthGraph(20_000, seed=7)
    sg.write_gfa(os.path.join(HERE, "syn20k.gfa"))
    reads = sg.sample_reads(6, 2000, seed=11)
    write_fasta(os.path.join(HERE, "syn20k.fa"), reads)
    ora = Oracle(os.path.join(HERE, "syn20k.gfa"))
    res = ora.align(reads)
    np.savez_compressed(os.path.join(HERE, "syn20k.expected.npz"), **{k: v for k, v in res.items() if k != "stage_microseconds"})
    # the output encoders' text for the same reads (read ids r0..r5): GAF with =/X and with M cigars, protobuf-JSON lines
    with open(os.path.join(HERE, "syn20k.expected.gaf"), "wb") as f:
        f.write(ora.gaf(False))
    with open(os.path.join(HERE, "syn20k.expected.merged.gaf"), "wb") as f:
        f.write(ora.gaf(True))
    with open(os.path.join(HERE, "syn20k.expected.json"), "wb") as f:
        f.write(ora.json())
    read = open(os.path.join(HERE, "ref_test_read.fa")).read().split("\n")[1]
    res = Oracle(os.path.join(HERE, "ref_test_graph.gfa")).align([read])
    np.savez_compressed(os.path.join(HERE, "ref_test.expected.npz"), **{k: v for k, v in res.items() if k != "stage_microseconds"})
    o = Oracle(os.path.join(HERE, "ref_test_graph.gfa"))
    np.savez_compressed(os.path.join(HERE, "ref_test.graph.npz"), **{k: o.graph_array(k) for k in
                        ["nodeLength", "nodeOffset", "nodeIDs", "reverse", "componentNumber", "chainNumber", "chainApproxPos", "out_off", "out_adj", "in_off", "in_adj", "mpc_width",
                         "index_kmers", "index_start", "index_positions", "index_maxcount"]})


if __name__ == "__main__":
    main()
