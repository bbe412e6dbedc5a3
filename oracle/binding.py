"""ctypes bindings for the parity oracle. See oracle/__init__.py for who may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def build():
    """Compile liboracle.so (and _ref/libref_units.so when /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


def load_oracle_lib():
    path = os.environ.get("GC_ORACLE_LIBRARY")   # a sanitizer build of the oracle (scripts/sanitize_host.sh)
    if not path:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
    lib = C.CDLL(path)
    lib.gco_create.restype = C.c_void_p
    lib.gco_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_int, C.c_int]
    lib.gco_error.restype = C.c_char_p
    lib.gco_error.argtypes = [C.c_void_p]
    lib.gco_destroy.argtypes = [C.c_void_p]
    lib.gco_align.restype = C.c_int
    lib.gco_align.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
    lib.gco_array.restype = C.POINTER(C.c_int64)
    lib.gco_array.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64)]
    lib.gco_graph_array.restype = C.c_int
    lib.gco_graph_array.argtypes = [C.c_void_p, C.c_char_p]
    u64, i32 = C.c_uint64, C.c_int32
    lib.gco_merge.argtypes = [u64, u64, i32, u64, u64, i32, C.POINTER(u64), C.POINTER(u64), C.POINTER(i32)]
    lib.gco_changed_min_score.restype = i32
    lib.gco_changed_min_score.argtypes = [u64, u64, i32, u64, u64, i32]
    lib.gco_get_value.restype = i32
    lib.gco_get_value.argtypes = [u64, u64, i32, C.c_int]
    lib.gco_score_before_start.restype = i32
    lib.gco_score_before_start.argtypes = [u64, u64, i32]
    lib.gco_next_slice.argtypes = [u64, u64, u64, i32, u64, u64, C.POINTER(u64), C.POINTER(u64), C.POINTER(i32), C.POINTER(u64), C.POINTER(u64)]
    lib.gco_correctness_series.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gco_edit_distance.restype = u64
    lib.gco_edit_distance.argtypes = [C.c_char_p, u64, C.c_char_p, u64]
    lib.gco_minimizer_hash.restype = u64
    lib.gco_minimizer_hash.argtypes = [u64]
    lib.gco_edit_path.restype = C.c_longlong
    lib.gco_edit_path.argtypes = [C.c_char_p, u64, C.c_char_p, u64, C.c_void_p, u64, C.POINTER(C.c_longlong)]
    lib.gco_evalue.argtypes = [C.c_double, u64, u64, u64, u64, C.c_void_p]
    lib.gco_set_e_cutoff.argtypes = [C.c_void_p, C.c_double]
    lib.gco_set_tie_order.argtypes = [C.c_void_p, C.c_int]
    lib.gco_extend.restype = C.c_int
    lib.gco_extend.argtypes = [C.c_void_p, C.c_char_p, u64, C.c_int, u64]
    lib.gco_align_timed.restype = C.c_double
    lib.gco_align_timed.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.gco_align_summary.restype = C.c_double
    lib.gco_align_summary.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.gco_align_summary2.restype = C.c_double
    lib.gco_align_summary2.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


RESULT_ARRAYS = [
    "read_seed_off", "seed_node", "seed_offset", "seed_seqpos", "seed_goodness",
    "read_frag_off", "frag_l", "frag_sl", "frag_sr",
    "read_anchor_off", "anchor_x", "anchor_y", "anchor_path_off", "anchor_path",
    "anchor_first_node", "anchor_first_offset", "anchor_first_seqpos",
    "anchor_last_node", "anchor_last_offset", "anchor_last_seqpos", "anchor_score",
    "anchor_trace_off", "anchor_trace_node", "anchor_trace_offset", "anchor_trace_seqpos", "anchor_trace_switch",
    "read_chain_off", "chain", "chain_score",
    "read_long_off", "long_start", "long_end", "long_score",
    "read_longall_off", "longall_start", "longall_end", "longall_score",
    "long_trace_off", "long_trace_node", "long_trace_offset", "long_trace_seqpos", "long_trace_switch",
    "read_path_off", "path_node", "path_offset",
    "long_edit_distance", "chain_edit_distance", "chained_better", "failed_assertion", "seeds_extended",
    "read_chain_ops_off", "chain_ops", "read_chain_trace_off", "chain_trace_node", "chain_trace_offset", "chain_trace_seqpos", "chain_trace_switch",
    "chain_aln_start", "chain_aln_end", "flatten_ties", "flatten_ties_long", "flatten_counters",
    "counters", "stage_microseconds",
]


class Oracle:
    """CPU restatement of the per-read hot path (reference defaults: src/AlignerMain.cpp:186-209)."""

    def __init__(self, gfa_path, k=15, w=20, density=10.0, discard_fraction=0.001, bandwidth=10,
                 split_len=35, split_gap=35, colinear_gap=10000, long_pass=True, shrink_mpc=True, e_cutoff=-1.0, tie_order=0):
        self.lib = load_oracle_lib()
        self.h = self.lib.gco_create(gfa_path.encode(), k, w, density, discard_fraction, bandwidth,
                                     split_len, split_gap, colinear_gap, int(long_pass), int(shrink_mpc))
        err = self.lib.gco_error(self.h).decode()
        if err:
            raise RuntimeError(err)
        self.lib.gco_set_e_cutoff(self.h, float(e_cutoff))
        # tie_order: the node order in which flattenLastSliceEnd takes its strict minimum (oracle/bitvector_aligner.hpp header): 0 = band-entry order, the order this build
        # DEFINES in place of the reference's parallel-hashmap iteration order; 1 = the reverse, for the sensitivity runs that say how many outputs depend on the definition
        self.lib.gco_set_tie_order(self.h, int(tie_order))

    def close(self):
        if self.h:
            self.lib.gco_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _array(self, name):
        n = C.c_uint64()
        p = self.lib.gco_array(self.h, name.encode(), C.byref(n))
        if not p or n.value == 0:
            return np.zeros(0, dtype=np.int64)
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy()

    def graph_array(self, name):
        if self.lib.gco_graph_array(self.h, name.encode()) != 0:
            raise KeyError(name)
        return self._array("graph_" + name)

    def align(self, reads):
        """reads: list of str/bytes. Returns dict of flat int64 arrays (CSR layout, see oracle_capi.cpp)."""
        bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
        off = np.zeros(len(bs) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(b) for b in bs])
        blob = b"".join(bs)
        rc = self.lib.gco_align(self.h, blob, off.ctypes.data, len(bs))
        if rc != 0:
            raise RuntimeError(self.lib.gco_error(self.h).decode())
        return {name: self._array(name) for name in RESULT_ARRAYS}

    def align_timed(self, reads, threads=1):
        """CPU baseline: aligns `reads` with `threads` workers over a shared queue (the reference's -t model), results discarded.
        Returns (wall seconds, per-stage CPU seconds [seed, whole-read pass, fragments, chaining, stitch + edlib])."""
        bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
        off = np.zeros(len(bs) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(b) for b in bs])
        stage = np.zeros(5, dtype=np.float64)
        wall = self.lib.gco_align_timed(self.h, b"".join(bs), off.ctypes.data, len(bs), int(threads), stage.ctypes.data)
        return float(wall), stage

    SUMMARY_FIELDS = ["anchors", "chain_len", "chain_hash", "chain_score", "long_edit_distance", "chain_edit_distance", "chained_better",
                      "longall", "longall_hash", "selected", "selected_hash", "failed_assertion", "flatten_ties", "flatten_ties_long"]

    def align_summary(self, reads, threads=1, gaf_hash=False):
        """align_timed that keeps 14 values per read (SUMMARY_FIELDS; see gco_align_summary) for bench.py's parity sample.
        Returns (wall seconds, stage seconds, int64 array [n, 14]); with gaf_hash a 15th column: the hash of the GAF lines the
        reference would write for the read (gco_align_summary2), which bench.py's end-to-end leg compares with the product's text."""
        bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
        off = np.zeros(len(bs) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(b) for b in bs])
        stage = np.zeros(5, dtype=np.float64)
        summary = np.zeros((len(bs), 14), dtype=np.int64)
        hashes = np.zeros(len(bs), dtype=np.int64)
        wall = self.lib.gco_align_summary2(self.h, b"".join(bs), off.ctypes.data, len(bs), int(threads), stage.ctypes.data, summary.ctypes.data, hashes.ctypes.data if gaf_hash else None)
        if gaf_hash:
            summary = np.concatenate([summary, hashes[:, None]], axis=1)
        return float(wall), stage, summary

    def extend(self, sequence, bigraph_node_id, node_offset):
        """One seed extension (src/GraphAlignerBitvectorBanded.h:46-71) laid open: per kept slice its minimum, minimum cell and node set, and the trace.
        Returns None when the extension trips one of the reference's assertions."""
        b = sequence.encode() if isinstance(sequence, str) else bytes(sequence)
        rc = self.lib.gco_extend(self.h, b, len(b), int(bigraph_node_id), int(node_offset))
        if rc == 2:
            return None
        off = self._array("ext_slice_off")
        nodes = self._array("ext_slice_nodes")
        return {
            "failed": rc == 1,
            "slice_min": self._array("ext_slice_min").tolist(),
            "slice_min_cell": list(zip(self._array("ext_slice_minnode").tolist(), self._array("ext_slice_minoffset").tolist())),
            "slice_nodes": [nodes[off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)],
            "score": None if rc == 1 else int(self._array("ext_score")[0]),
            "trace": [] if rc == 1 else [tuple(x) for x in self._array("ext_trace").reshape(-1, 3).tolist()],
        }

    def gaf(self, merge=False):
        """GAF text of the last align() call (read ids r0, r1, ...), the reference's writer restated (oracle/output.hpp)."""
        self.lib.gco_gaf.restype = C.c_char_p
        self.lib.gco_gaf.argtypes = [C.c_void_p, C.c_int]
        return self.lib.gco_gaf(self.h, int(merge))

    def json(self):
        """JSON lines (vg::Alignment via protobuf's JSON mapping) of the last align() call's final alignments."""
        self.lib.gco_json.restype = C.c_char_p
        self.lib.gco_json.argtypes = [C.c_void_p]
        return self.lib.gco_json(self.h)


    def gam_groups(self):
        """The final alignments of the last align() call as the reference's GAM stream before the gzip layer (src/Aligner.cpp:261-281): one bytes object
        per read with output, holding varint count + (varint size, vg::Alignment proto3 bytes) per alignment. The reference deflates each into its own gzip member."""
        n, ng, off = C.c_uint64(0), C.c_uint64(0), C.POINTER(C.c_uint64)()
        self.lib.gco_gam.restype = C.POINTER(C.c_char)
        self.lib.gco_gam.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_uint64)]
        ptr = self.lib.gco_gam(self.h, C.byref(n), C.byref(off), C.byref(ng))
        raw = C.string_at(ptr, n.value)
        return [raw[off[i]:off[i + 1]] for i in range(ng.value)]


class RefUnits:
    """The reference's own WordSlice.h / AlignmentCorrectnessEstimation.cpp / edlib, compiled unmodified."""

    def __init__(self):
        path = os.path.join(_HERE, "_ref", "libref_units.so")
        if not os.path.exists(path):
            build()
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        lib = C.CDLL(path)
        u64, i32 = C.c_uint64, C.c_int32
        lib.ref_merge.argtypes = [u64, u64, i32, u64, u64, i32, C.POINTER(u64), C.POINTER(u64), C.POINTER(i32)]
        lib.ref_changed_min_score.restype = i32
        lib.ref_changed_min_score.argtypes = [u64, u64, i32, u64, u64, i32]
        lib.ref_get_value.restype = i32
        lib.ref_get_value.argtypes = [u64, u64, i32, C.c_int]
        lib.ref_score_before_start.restype = i32
        lib.ref_score_before_start.argtypes = [u64, u64, i32]
        lib.ref_correctness_series.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ref_edit_distance.restype = C.c_longlong
        lib.ref_edit_distance.argtypes = [C.c_char_p, u64, C.c_char_p, u64]
        lib.ref_edit_path.restype = C.c_longlong
        lib.ref_edit_path.argtypes = [C.c_char_p, u64, C.c_char_p, u64, C.c_void_p, u64, C.POINTER(C.c_longlong)]
        lib.ref_evalue.argtypes = [C.c_double, u64, u64, u64, u64, C.c_void_p]
        self.lib = lib

    def edit_path(self, a, b):
        """edlibAlign(a, b, NW, PATH) of the real edlib: (distance, op string as uint8 array)."""
        return _edit_path(self.lib.ref_edit_path, a, b)

    def evalue(self, min_identity, database_size, query_size, alignment_length, num_edits):
        out = np.zeros(2, dtype=np.float64)
        self.lib.ref_evalue(min_identity, database_size, query_size, alignment_length, num_edits, out.ctypes.data)
        return out


def _edit_path(fn, a, b):
    buf = np.zeros(len(a) + len(b) + 8, dtype=np.uint8)
    d = C.c_longlong(-1)
    n = fn(a, len(a), b, len(b), buf.ctypes.data, len(buf), C.byref(d))
    if n < 0:
        raise RuntimeError(f"edit path failed ({n})")
    return int(d.value), buf[:n].copy()


def oracle_edit_path(a, b):
    """The oracle's restatement of the same call (oracle/edlib_path.hpp)."""
    return _edit_path(load_oracle_lib().gco_edit_path, a, b)


def oracle_evalue(min_identity, database_size, query_size, alignment_length, num_edits):
    out = np.zeros(2, dtype=np.float64)
    load_oracle_lib().gco_evalue(min_identity, database_size, query_size, alignment_length, num_edits, out.ctypes.data)
    return out
