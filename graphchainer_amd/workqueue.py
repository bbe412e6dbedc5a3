"""Read-parallel multi-GPU driver: one process per GPU, graph and index replicated, reads handed out by a work queue.

The hot path shards embarrassingly (every read is independent, src/Aligner.cpp:492-1062), so there is no data-path collective
and no RCCL: the reference's own model is N workers pulling reads from one queue (src/Aligner.cpp:1267-1270, the moodycamel
queue filled by the reader thread). Here the queue hands out BATCHES of reads of similar length (a batch is one
gc_align_batch call; similar lengths keep the lanes of the extension kernels and the whole-read rounds balanced), dynamically,
so a rank that finishes early takes more:

  length_sorted_batches(reads, batch)  -> the batches, longest reads first
  ReadQueue(n, rank, world, dist)      -> next() returns the next batch index for this process or None
        one process (weak scaling, or a single GPU): an in-process counter shared by the worker threads of the rank;
        several processes on one node (strong scaling): a counter file advanced under flock - every batch index is handed
        to exactly one rank; torch.distributed (gloo) only synchronises the reset between steps.
  run_queue(queue, align_fn, workers)  -> drives `workers` threads (one gc_stream each) over the queue and returns
        [(batch index, result)] - what bench.py times, and what tests drive with a mocked align_fn on CPU-only boxes.
  merge_read_results(parts, batches, n_reads, key) -> per-read values back in the caller's read order.
  read_summary(result)                 -> 12 values per read of a batch result (what must match the CPU path read for read:
        anchors, chain, chain score, both NW distances, the decision, the whole-read alignments and the selection).
"""
import fcntl
import os
import struct
import tempfile
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def length_sorted_batches(reads, batch):
    """Indices of `reads` cut into batches of at most `batch` reads, longest reads first (stable: equal lengths keep their order)."""
    lengths = np.array([len(r) for r in reads], dtype=np.int64)
    order = np.argsort(-lengths, kind="stable")
    return [order[i:i + batch].tolist() for i in range(0, len(order), batch)] or [[]]


class ReadQueue:
    def __init__(self, n_items, rank=0, world=1, dist=None, path=None):
        self.n = int(n_items)
        self.rank, self.world, self.dist = rank, world, dist
        self.lock = threading.Lock()
        self.cursor = 0
        self.shared = dist is not None and world > 1
        self.fd = None
        if self.shared:
            # one counter file per job on the node-local temp dir: rank 0 creates it exclusively under a fresh name (mkstemp: O_EXCL, so
            # nothing pre-placed at a guessable path is written through) and tells the others the path; `path` is for tests
            names = [None]
            if rank == 0:
                if path is None:
                    fd, path = tempfile.mkstemp(prefix="gc_queue_", suffix=".bin")
                else:
                    fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
                os.pwrite(fd, struct.pack("<q", 0), 0)
                os.close(fd)
                names = [path]
            dist.broadcast_object_list(names, src=0)
            self.path = names[0]
            self.fd = os.open(self.path, os.O_RDWR | getattr(os, "O_NOFOLLOW", 0))

    def reset(self, n_items=None):
        """Start of a pass: every rank has drained the queue (barrier), rank 0 rewinds it, everyone sees the rewound cursor.
        n_items: a new item count (e.g. steps x batches when consecutive steps may overlap)."""
        if n_items is not None:
            self.n = int(n_items)
        if self.shared:
            self.dist.barrier()
            if self.rank == 0:
                os.pwrite(self.fd, struct.pack("<q", 0), 0)
            self.dist.barrier()
        else:
            with self.lock:
                self.cursor = 0

    def next(self):
        if self.shared:
            with self.lock:   # one thread of this process at a time; flock arbitrates between the processes
                fcntl.flock(self.fd, fcntl.LOCK_EX)
                try:
                    (cur,) = struct.unpack("<q", os.pread(self.fd, 8, 0))
                    if cur >= self.n:
                        return None
                    os.pwrite(self.fd, struct.pack("<q", cur + 1), 0)
                    return cur
                finally:
                    fcntl.flock(self.fd, fcntl.LOCK_UN)
        with self.lock:
            if self.cursor >= self.n:
                return None
            self.cursor += 1
            return self.cursor - 1

    def close(self):
        if self.shared and self.fd is not None:
            os.close(self.fd)
            self.fd = None
            if self.rank == 0:
                try:
                    os.remove(self.path)
                except OSError:
                    pass


def run_queue(queue, align_fn, workers=1):
    """`workers` host threads (one gc_stream each: align_fn(worker, batch_index) -> result) drain the queue. Returns
    [(batch_index, result)] in completion order per worker."""
    def work(i):
        got = []
        while True:
            b = queue.next()
            if b is None:
                return got
            got.append((b, align_fn(i, b)))
    if workers <= 1:
        return work(0)
    out = []
    with ThreadPoolExecutor(max_workers=workers) as pool:
        for part in pool.map(work, range(workers)):
            out.extend(part)
    return out


def merge_read_results(parts, batches, n_reads, key, fill=0):
    """Per-read array `key` of the batch results in `parts` ([(batch index, result dict)]) scattered back to the original read
    order (batches[b] = original indices of batch b's reads). Reads of batches this process did not run keep `fill`."""
    out = np.full(n_reads, fill, dtype=np.int64)
    for b, res in parts:
        out[np.asarray(batches[b], dtype=np.int64)] = np.asarray(res[key], dtype=np.int64)
    return out


SUMMARY_FIELDS = ["anchors", "chain_len", "chain_hash", "chain_score", "long_edit_distance", "chain_edit_distance", "chained_better",
                  "longall", "longall_hash", "selected", "selected_hash", "failed_assertion", "flatten_ties", "flatten_ties_long"]
SUMMARY_WIDTH = len(SUMMARY_FIELDS)


def _list_hash(values, offsets):
    """Per segment [offsets[r], offsets[r+1]) of `values`: sum (v[i] + 1) * (i + 1) * 2654435761 mod 2^64, i counted inside the segment."""
    offsets = np.asarray(offsets, dtype=np.int64)
    n = len(offsets) - 1
    v = np.asarray(values, dtype=np.int64).astype(np.uint64)
    if len(v) == 0:
        return np.zeros(n, dtype=np.int64)
    local = np.arange(len(v), dtype=np.int64) - np.repeat(offsets[:-1], np.diff(offsets))
    with np.errstate(over="ignore"):
        terms = (v + np.uint64(1)) * ((local.astype(np.uint64) + np.uint64(1)) * np.uint64(2654435761))
        csum = np.concatenate([np.zeros(1, dtype=np.uint64), np.cumsum(terms, dtype=np.uint64)])
        return (csum[offsets[1:]] - csum[offsets[:-1]]).view(np.int64)


def gaf_read_hashes(text, lines_per_read):
    """Per read the hash of its GAF lines as the CPU leg's summary holds it (oracle: gco_align_summary2): every line from its first TAB to its
    newline (the read name left out), line hash = list hash of the bytes, read hash = list hash of its lines' hashes. `text` = what
    gc_format_gaf returned for a batch, lines_per_read[r] = final alignments of read r (its lines are consecutive, reads in batch order)."""
    lines_per_read = np.asarray(lines_per_read, dtype=np.int64)
    a = np.frombuffer(text, dtype=np.uint8)
    nl = np.flatnonzero(a == 10)
    if len(nl) != int(lines_per_read.sum()):
        raise ValueError(f"GAF text has {len(nl)} lines, the result says {int(lines_per_read.sum())}")
    read_off = np.concatenate([np.zeros(1, dtype=np.int64), np.cumsum(lines_per_read)])
    if len(nl) == 0:
        return np.zeros(len(lines_per_read), dtype=np.int64)
    starts = np.concatenate([np.zeros(1, dtype=np.int64), nl[:-1] + 1])
    tabs = np.flatnonzero(a == 9)
    first_tab = tabs[np.searchsorted(tabs, starts)]
    seg_len = nl - first_tab + 1
    seg_off = np.concatenate([np.zeros(1, dtype=np.int64), np.cumsum(seg_len)])
    idx = np.repeat(first_tab - seg_off[:-1], seg_len) + np.arange(int(seg_off[-1]), dtype=np.int64)
    line_hash = _list_hash(a[idx], seg_off)
    return _list_hash(line_hash, read_off)


def read_summary(out):
    """[n, 14] int64: SUMMARY_FIELDS of every read of a batch result (a dict as Aligner.align_batch returns it; the selected whole-read
    alignments either as `long_index` into the read's alignment list or as `long_start/end/score`). The chain and alignment lists enter
    as order-sensitive 64-bit hashes, so two results agree on a read iff its chain, scores, distances, decision and alignments agree."""
    i64 = lambda k: np.asarray(out[k]).astype(np.int64)
    chain_off, anchor_off, all_off, sel_off = i64("read_chain_off"), i64("read_anchor_off"), i64("read_longall_off"), i64("read_long_off")
    n = len(chain_off) - 1
    triples = lambda s, e, c: np.stack([s, e, c], axis=1).reshape(-1)
    everything = triples(i64("longall_start"), i64("longall_end"), i64("longall_score"))
    if "long_index" in out:
        sel = np.repeat(all_off[:-1], np.diff(sel_off)) + i64("long_index")
        selected = triples(i64("longall_start")[sel], i64("longall_end")[sel], i64("longall_score")[sel])
    else:
        selected = triples(i64("long_start"), i64("long_end"), i64("long_score"))
    s = np.zeros((n, SUMMARY_WIDTH), dtype=np.int64)
    s[:, 0] = np.diff(anchor_off)
    s[:, 1] = np.diff(chain_off)
    s[:, 2] = _list_hash(i64("chain"), chain_off)
    s[:, 3] = i64("chain_score")
    s[:, 4] = i64("long_edit_distance")
    s[:, 5] = i64("chain_edit_distance")
    s[:, 6] = i64("chained_better")
    s[:, 7] = np.diff(all_off)
    s[:, 8] = _list_hash(everything, 3 * all_off)
    s[:, 9] = np.diff(sel_off)
    s[:, 10] = _list_hash(selected, 3 * sel_off)
    s[:, 11] = i64("failed_assertion")
    s[:, 12] = i64("flatten_ties")            # r5: extensions whose last-slice minimum is attained in more than one node (the tie whose order this build defines)
    s[:, 13] = i64("flatten_ties_long")
    return s
