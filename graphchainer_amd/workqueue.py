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
  merge_read_results(parts, n_reads, order) -> per-read values back in the caller's read order.
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
        if self.shared:
            # one counter file per job on the node-local temp dir; rank 0 creates it, the others open it after a barrier
            self.path = path or os.path.join(tempfile.gettempdir(), f"gc_queue_{os.environ.get('MASTER_PORT', 'x')}_{os.getuid()}.bin")
            if rank == 0:
                with open(self.path, "wb") as f:
                    f.write(struct.pack("<q", 0))
            dist.barrier()
            self.fd = os.open(self.path, os.O_RDWR)

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
        if self.shared:
            os.close(self.fd)
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
