#!/usr/bin/env python3
"""Larger one-off parity run of the HIP path against the oracle (every result array, whole-read pass included).
Usage: python scripts/parity_sweep.py [backbone_bp] [n_reads] [read_len] [seed] [colinear_gap]"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import graphchainer_amd as gca  # noqa: E402
from graphchainer_amd.synth import SynthGraph  # noqa: E402
import test_gpu_parity as T  # noqa: E402


def main():
    backbone = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
    n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    read_len = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 31
    kw = {"colinear_gap": int(sys.argv[5])} if len(sys.argv) > 5 else {}
    with tempfile.TemporaryDirectory() as tmp:
        gfa = os.path.join(tmp, "g.gfa")
        sg = SynthGraph(backbone, seed=seed)
        sg.write_gfa(gfa)
        reads = sg.sample_reads(n_reads, read_len, seed=seed + 1)
        rc = bytes.maketrans(b"ACGT", b"TGCA")
        reads += [r[::-1].translate(rc) for r in reads[: n_reads // 10]]              # reverse-strand reads
        reads += [reads[i][: read_len // 3] + reads[i + 1][read_len // 2:] for i in range(0, n_reads // 10, 2)]   # chimeras
        t0 = time.time()
        got, want = T.run_case(gca, gfa, reads, long_pass=True, **kw)
        print(f"{len(reads)} reads, {time.time() - t0:.1f} s (mostly the oracle)")
        T.compare(got, want, T.COMPARE_KEYS + T.LONG_KEYS)
        print("parity ok:", {k: int(np.asarray(got[k]).size) for k in ("anchor_x", "chain", "longall_start", "long_trace_node", "path_node")},
              "chained_better", int(np.sum(got["chained_better"])), "chains stitched on the host", int(got["counters"][7]))


if __name__ == "__main__":
    main()
