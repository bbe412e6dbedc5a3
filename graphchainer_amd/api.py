"""Python host side of the MI355X GraphChainer hot path.

Thin ctypes layer over the C ABI (include/graphchainer_amd.h, built into graphchainer_amd/libgraphchainer_amd.so).
Class and method names mirror the reference objects they stand in for:

  AlignmentGraph   <- AlignmentGraph + buildMPC          (src/AlignmentGraph.h, src/Aligner.cpp:1137-1156)
  MinimizerSeeder  <- MinimizerSeeder                    (src/MinimizerSeeder.h:32)
  Aligner.align_reads(reads) <- the per-read body of runComponentMappings (src/Aligner.cpp:601-922), batched

There is no CPU fallback: if the HIP library is missing or no GPU is visible these raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GC_LIBRARY") or os.path.join(_HERE, "libgraphchainer_amd.so")   # GC_LIBRARY: profiling builds (make stamps)

EXPORTED_SYMBOLS = [
    "gc_params_default", "gc_graph_create_from_gfa", "gc_graph_create", "gc_graph_destroy", "gc_graph_num_nodes",
    "gc_graph_size_bp", "gc_graph_array", "gc_graph_trim_host", "gc_seeder_create", "gc_seeder_destroy", "gc_seeder_array",
    "gc_stream_create", "gc_stream_destroy", "gc_reads_upload", "gc_reads_destroy", "gc_align_batch",
    "gc_result_free", "gc_last_error", "gc_free", "gc_device_count", "gc_set_device", "gc_device_memory", "gc_edit_distance", "gc_edit_path", "gc_evalue", "gc_format_gaf", "gc_format_json", "gc_format_gam", "gc_format_gam_level", "gc_gzip_streams", "gc_gzip_streams_lz", "gc_format_gaf_trace", "gc_format_vg_trace", "gc_format_vg_trace_digraph", "gc_graph_letters", "gc_std_sort_permutations",
    "gc_index_build", "gc_index_save", "gc_index_load", "gc_index_check", "gc_result_cache_trim",
]


class GcCapacities(C.Structure):
    """gc_capacities (include/graphchainer_amd.h): sizes of the device-side tables; 0 = automatic."""
    _fields_ = [("ext_max_items", C.c_int64), ("ext_max_pending", C.c_int64), ("ext_max_trace", C.c_int64), ("long_max_items", C.c_int64), ("long_column_store", C.c_int64),
                ("long_cells_per_base", C.c_int64), ("long_scratch_bytes", C.c_int64), ("stitch_set_max", C.c_int64), ("stitch_bfs_cap", C.c_int64), ("reserved", C.c_int64 * 3)]


class GcParams(C.Structure):
    _fields_ = [("bandwidth", C.c_int32), ("split_len", C.c_int32), ("split_gap", C.c_int32), ("colinear_gap", C.c_int64),
                ("seed_density", C.c_double), ("min_cluster_size", C.c_int32), ("long_pass", C.c_int32), ("keep_traces", C.c_int32), ("keep_seeds", C.c_int32), ("stitch", C.c_int32), ("edit_distances", C.c_int32),
                ("chain_traces", C.c_int32), ("device_output", C.c_int32), ("e_cutoff", C.c_double), ("capacity", GcCapacities)]


_P = C.POINTER


class GcResult(C.Structure):
    _fields_ = [
        ("n_reads", C.c_uint64),
        ("read_seed_off", _P(C.c_uint64)), ("seed_node", _P(C.c_uint32)), ("seed_offset", _P(C.c_uint32)), ("seed_seqpos", _P(C.c_uint32)), ("seed_goodness", _P(C.c_uint64)),
        ("read_anchor_off", _P(C.c_uint64)), ("anchor_x", _P(C.c_uint32)), ("anchor_y", _P(C.c_uint32)),
        ("anchor_path_off", _P(C.c_uint64)), ("anchor_path", _P(C.c_uint32)),
        ("anchor_first_node", _P(C.c_uint32)), ("anchor_first_offset", _P(C.c_uint32)), ("anchor_first_seqpos", _P(C.c_uint32)),
        ("anchor_last_node", _P(C.c_uint32)), ("anchor_last_offset", _P(C.c_uint32)), ("anchor_last_seqpos", _P(C.c_uint32)),
        ("anchor_score", _P(C.c_int32)),
        ("anchor_trace_off", _P(C.c_uint64)), ("anchor_trace_node", _P(C.c_int32)), ("anchor_trace_offset", _P(C.c_uint32)),
        ("anchor_trace_seqpos", _P(C.c_uint32)), ("anchor_trace_switch", _P(C.c_uint8)),
        ("read_chain_off", _P(C.c_uint64)), ("chain", _P(C.c_uint32)), ("chain_score", _P(C.c_uint64)),
        ("read_longall_off", _P(C.c_uint64)), ("longall_start", _P(C.c_uint32)), ("longall_end", _P(C.c_uint32)), ("longall_score", _P(C.c_uint32)),
        ("long_trace_off", _P(C.c_uint64)), ("long_trace_node", _P(C.c_int32)), ("long_trace_offset", _P(C.c_uint32)),
        ("long_trace_seqpos", _P(C.c_uint32)), ("long_trace_switch", _P(C.c_uint8)),
        ("failed_assertion", _P(C.c_uint8)), ("capacity_exceeded", _P(C.c_uint8)), ("seeds_extended", _P(C.c_uint64)), ("seeds_extended_long", _P(C.c_uint64)),
        ("read_path_off", _P(C.c_uint64)), ("path_node", _P(C.c_uint32)), ("path_first_offset", _P(C.c_uint32)), ("path_last_offset", _P(C.c_uint32)), ("path_cells", _P(C.c_uint64)),
        ("read_long_off", _P(C.c_uint64)), ("long_index", _P(C.c_uint32)), ("long_edit_distance", _P(C.c_int64)), ("chain_edit_distance", _P(C.c_int64)), ("chained_better", _P(C.c_uint8)),
        ("read_chain_trace_off", _P(C.c_uint64)), ("chain_trace_node", _P(C.c_int32)), ("chain_trace_offset", _P(C.c_uint32)), ("chain_trace_seqpos", _P(C.c_uint32)), ("chain_trace_switch", _P(C.c_uint8)),
        ("chain_aln_start", _P(C.c_uint32)), ("chain_aln_end", _P(C.c_uint32)),
        ("counters", C.c_uint64 * 8), ("counters_long", C.c_uint64 * 8), ("kernel_us", C.c_double * 8), ("host_us", C.c_double * 4),
        ("read_out_off", _P(C.c_uint64)), ("out_source", _P(C.c_uint8)), ("out_numbers", _P(C.c_uint64)),
        ("out_path_off", _P(C.c_uint64)), ("out_path_text", _P(C.c_char)), ("out_cigar_off", _P(C.c_uint64)), ("out_cigar_text", _P(C.c_char)),
        ("out_vg_off", _P(C.c_uint64)), ("out_vg_path", _P(C.c_uint8)),
        ("flatten_ties", _P(C.c_uint32)), ("flatten_ties_long", _P(C.c_uint32)), ("device_output", C.c_int32),
    ]


_lib = None


def load_library():
    """Loads the in-tree HIP library; fails loudly if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C graphchainer_amd/csrc` (or __graft_entry__.build())")
    lib = C.CDLL(LIB_PATH)
    lib.gc_last_error.restype = C.c_char_p
    lib.gc_graph_create_from_gfa.argtypes = [C.c_char_p, _P(C.c_void_p)]
    lib.gc_graph_destroy.argtypes = [C.c_void_p]
    lib.gc_graph_num_nodes.restype = C.c_uint64
    lib.gc_graph_num_nodes.argtypes = [C.c_void_p]
    lib.gc_graph_size_bp.restype = C.c_uint64
    lib.gc_graph_size_bp.argtypes = [C.c_void_p]
    lib.gc_graph_array.argtypes = [C.c_void_p, C.c_char_p, _P(_P(C.c_int64)), _P(C.c_uint64)]
    lib.gc_graph_trim_host.argtypes = [C.c_void_p]
    lib.gc_seeder_create.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_double, _P(C.c_void_p)]
    lib.gc_seeder_destroy.argtypes = [C.c_void_p]
    lib.gc_seeder_array.argtypes = [C.c_void_p, C.c_char_p, _P(_P(C.c_int64)), _P(C.c_uint64)]
    lib.gc_stream_create.argtypes = [_P(C.c_void_p)]
    lib.gc_stream_destroy.argtypes = [C.c_void_p]
    lib.gc_reads_upload.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, _P(C.c_void_p)]
    lib.gc_reads_destroy.argtypes = [C.c_void_p]
    lib.gc_align_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _P(GcParams), _P(_P(GcResult))]
    lib.gc_result_free.argtypes = [_P(GcResult)]
    lib.gc_params_default.argtypes = [_P(GcParams)]
    lib.gc_free.argtypes = [C.c_void_p]
    lib.gc_set_device.argtypes = [C.c_int]
    lib.gc_index_build.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_double, C.c_char_p]
    lib.gc_index_save.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p]
    lib.gc_index_load.argtypes = [C.c_char_p, _P(C.c_void_p), _P(C.c_void_p)]
    lib.gc_index_check.argtypes = [C.c_char_p, _P(C.c_uint64)]
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise RuntimeError(f"graphchainer_amd error {rc}: {load_library().gc_last_error().decode()}")


def device_count():
    return load_library().gc_device_count()


def device_memory():
    """(free, total) bytes of the current device's memory."""
    lib = load_library()
    free, total = C.c_uint64(), C.c_uint64()
    lib.gc_device_memory.argtypes = [_P(C.c_uint64), _P(C.c_uint64)]
    _check(lib.gc_device_memory(C.byref(free), C.byref(total)))
    return int(free.value), int(total.value)


def set_device(index):
    _check(load_library().gc_set_device(index))


def edit_distance(a_list, b_list):
    """Global (NW) edit distance of each (a, b) pair of byte strings on the GPU (edlib NW distance, src/Aligner.cpp:645,845)."""
    assert len(a_list) == len(b_list)
    lib = load_library()
    n = len(a_list)
    a = b"".join(a_list)
    b = b"".join(b_list)
    a_off = np.zeros(n + 1, dtype=np.uint64)
    b_off = np.zeros(n + 1, dtype=np.uint64)
    a_off[1:] = np.cumsum([len(x) for x in a_list], dtype=np.uint64) if n else []
    b_off[1:] = np.cumsum([len(x) for x in b_list], dtype=np.uint64) if n else []
    out = np.zeros(max(n, 1), dtype=np.int64)
    lib.gc_edit_distance.restype = C.c_int
    lib.gc_edit_distance.argtypes = [C.c_char_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p]
    _check(lib.gc_edit_distance(a, a_off.ctypes.data, b, b_off.ctypes.data, n, out.ctypes.data))
    return out[:n]


def edit_path(a_list, b_list):
    """edlibAlign(a, b, NW, EDLIB_TASK_PATH) of each pair on the GPU (src/Aligner.cpp:845): returns (distances, [op arrays]);
    ops: 0 match, 1 letter of a alone, 2 letter of b alone, 3 mismatch."""
    assert len(a_list) == len(b_list)
    lib = load_library()
    n = len(a_list)
    a, b = b"".join(a_list), b"".join(b_list)
    a_off = np.zeros(n + 1, dtype=np.uint64)
    b_off = np.zeros(n + 1, dtype=np.uint64)
    a_off[1:] = np.cumsum([len(x) for x in a_list], dtype=np.uint64) if n else []
    b_off[1:] = np.cumsum([len(x) for x in b_list], dtype=np.uint64) if n else []
    ops_off = (a_off + b_off).astype(np.uint64)
    ops = np.zeros(max(int(ops_off[-1]), 1), dtype=np.uint8)
    ops_len = np.zeros(max(n, 1), dtype=np.uint32)
    dist = np.zeros(max(n, 1), dtype=np.int64)
    lib.gc_edit_path.restype = C.c_int
    lib.gc_edit_path.argtypes = [C.c_char_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _check(lib.gc_edit_path(a, a_off.ctypes.data, b, b_off.ctypes.data, n, ops_off.ctypes.data, ops.ctypes.data, ops_len.ctypes.data, dist.ctypes.data))
    return dist[:n], [ops[int(ops_off[i]):int(ops_off[i]) + int(ops_len[i])].copy() for i in range(n)]


GAM_DEVICE_HUFFMAN = 100   # GC_GAM_DEVICE_HUFFMAN: gam_level value that has the gzip members deflated on the device
GAM_DEVICE_LZ = 101        # GC_GAM_DEVICE_LZ (r6): ... with LZ77 matches in front of the Huffman stage


def std_sort_permutations(arrays, depth_limit=-1):
    """Test entry: the permutations the device's wave-cooperative replay of libstdc++'s std::sort (csrc/hip/gc_stdsort_wave.hpp) gives for arrays of uint32 keys."""
    lib = load_library()
    arrays = [np.ascontiguousarray(a, dtype=np.uint32) for a in arrays]
    off = np.zeros(len(arrays) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(a) for a in arrays])
    keys = np.concatenate(arrays) if arrays else np.zeros(0, dtype=np.uint32)
    perm = np.zeros(len(keys), dtype=np.uint32)
    lib.gc_std_sort_permutations.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64, C.c_void_p]
    _check(lib.gc_std_sort_permutations(keys.ctypes.data, off.ctypes.data, len(arrays), int(depth_limit), perm.ctypes.data))
    return [perm[int(off[i]):int(off[i + 1])] for i in range(len(arrays))]


def gzip_streams(streams, lz=False):
    """One gzip member per byte string, deflated on the GPU as one dynamic-Huffman block of literals (what gam_level=GAM_DEVICE_HUFFMAN does with a batch's GAM groups);
    lz: with LZ77 matches in front of the Huffman stage (GAM_DEVICE_LZ)."""
    lib = load_library()
    n = len(streams)
    raw = b"".join(streams)
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in streams], dtype=np.uint64) if n else []
    out_off = np.zeros(n + 1, dtype=np.uint64)
    ptr = C.c_void_p()
    fn = lib.gc_gzip_streams_lz if lz else lib.gc_gzip_streams
    fn.restype = C.c_int
    fn.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, _P(C.c_void_p), C.c_void_p]
    _check(fn(raw, off.ctypes.data, n, C.byref(ptr), out_off.ctypes.data))
    data = C.string_at(ptr.value, int(out_off[n]))
    lib.gc_free(ptr)
    return [data[int(out_off[i]):int(out_off[i + 1])] for i in range(n)]


def evalue(min_identity, database_size, query_size, alignment_length, num_edits):
    """{alignment score, E-value} of the --E-cutoff model (host only)."""
    lib = load_library()
    out = np.zeros(2, dtype=np.float64)
    lib.gc_evalue.restype = C.c_int
    lib.gc_evalue.argtypes = [C.c_double, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
    _check(lib.gc_evalue(min_identity, database_size, query_size, alignment_length, num_edits, out.ctypes.data))
    return out


def _fetch_array(fn, handle, name):
    lib = load_library()
    ptr = _P(C.c_int64)()
    n = C.c_uint64()
    _check(fn(handle, name.encode(), C.byref(ptr), C.byref(n)))
    out = np.ctypeslib.as_array(ptr, shape=(n.value,)).copy() if n.value else np.zeros(0, dtype=np.int64)
    lib.gc_free(ptr)
    return out


class GcGraphDesc(C.Structure):
    _fields_ = [("n_nodes", C.c_uint64), ("first_ambiguous", C.c_uint64), ("node_length", C.c_void_p), ("node_offset", C.c_void_p),
                ("node_ids", C.c_void_p), ("node_seq", C.c_void_p), ("ambiguous_seq", C.c_void_p), ("in_off", C.c_void_p), ("in_adj", C.c_void_p),
                ("out_off", C.c_void_p), ("out_adj", C.c_void_p), ("component_number", C.c_void_p), ("chain_number", C.c_void_p),
                ("chain_approx_pos", C.c_void_p), ("lookup_order", C.c_void_p), ("n_lookup", C.c_uint64)]


def build_index_cache(gfa_path, cache_path, minimizer_length=15, window_size=20, keep_least_frequent_fraction=1 - 0.001):
    """Builds graph + MPC index (+ minimizer index unless minimizer_length is 0) on the host and writes the cache file;
    needs no GPU (the reference's saveMPC, src/AlignmentGraph.h:96, is an empty stub)."""
    _check(load_library().gc_index_build(os.fsencode(gfa_path), minimizer_length, window_size, keep_least_frequent_fraction, os.fsencode(cache_path)))


def save_index_cache(graph, seeder, cache_path):
    _check(load_library().gc_index_save(graph.handle, seeder.handle if seeder is not None else None, os.fsencode(cache_path)))


def check_index_cache(cache_path):
    """Verifies a cache file on the host and returns what it holds; raises RuntimeError for a damaged or foreign file."""
    info = (C.c_uint64 * 8)()
    _check(load_library().gc_index_check(os.fsencode(cache_path), info))
    keys = ["version", "nodes", "bp", "has_seeder", "k", "w", "kmers", "positions"]
    return {k: int(v) for k, v in zip(keys, info)}


def load_index_cache(cache_path):
    """-> (AlignmentGraph, MinimizerSeeder or None) uploaded to the current device from a cache file."""
    lib = load_library()
    g, s = C.c_void_p(), C.c_void_p()
    _check(lib.gc_index_load(os.fsencode(cache_path), C.byref(g), C.byref(s)))
    graph = AlignmentGraph(None, _handle=g)
    seeder = MinimizerSeeder(graph, _handle=s) if s else None
    return graph, seeder


class AlignmentGraph:
    """Split-node DAG + MPC index resident in HBM (reference: AlignmentGraph, src/AlignmentGraph.h)."""

    def __init__(self, gfa_path, _handle=None):
        self.lib = load_library()
        self.handle = C.c_void_p()
        if _handle is not None:
            self.handle = _handle
            return
        _check(self.lib.gc_graph_create_from_gfa(os.fsencode(gfa_path), C.byref(self.handle)))

    @classmethod
    def from_arrays(cls, arrays, with_lookup_order=True):
        """gc_graph_create: a graph from the arrays a host that keeps its own AlignmentGraph would hand over (the names are
        those of gc_graph_array)."""
        lib = load_library()
        keep = {
            "node_length": np.ascontiguousarray(arrays["nodeLength"], dtype=np.uint8),
            "node_offset": np.ascontiguousarray(arrays["nodeOffset"], dtype=np.uint32),
            "node_ids": np.ascontiguousarray(arrays["nodeIDs"], dtype=np.int32),
            "node_seq": np.ascontiguousarray(arrays["nodeSeq"]).view(np.uint64) if len(arrays["nodeSeq"]) else np.zeros(1, dtype=np.uint64),
            "ambiguous_seq": np.ascontiguousarray(arrays["ambiguousSeq"]).view(np.uint64) if len(arrays["ambiguousSeq"]) else np.zeros(1, dtype=np.uint64),
            "in_off": np.ascontiguousarray(arrays["in_off"], dtype=np.uint64), "in_adj": np.ascontiguousarray(arrays["in_adj"], dtype=np.uint32),
            "out_off": np.ascontiguousarray(arrays["out_off"], dtype=np.uint64), "out_adj": np.ascontiguousarray(arrays["out_adj"], dtype=np.uint32),
            "component_number": np.ascontiguousarray(arrays["componentNumber"], dtype=np.uint32),
            "chain_number": np.ascontiguousarray(arrays["chainNumber"], dtype=np.uint32),
            "chain_approx_pos": np.ascontiguousarray(arrays["chainApproxPos"], dtype=np.uint64),
            "lookup_order": np.ascontiguousarray(arrays["lookupOrder"], dtype=np.int32),
        }
        desc = GcGraphDesc()
        desc.n_nodes = len(keep["node_length"])
        desc.first_ambiguous = int(arrays["firstAmbiguous"][0])
        for name, arr in keep.items():
            setattr(desc, name, arr.ctypes.data)
        desc.n_lookup = len(keep["lookup_order"])
        if not with_lookup_order:
            desc.lookup_order, desc.n_lookup = None, 0
        handle = C.c_void_p()
        lib.gc_graph_create.argtypes = [_P(GcGraphDesc), _P(C.c_void_p)]
        _check(lib.gc_graph_create(C.byref(desc), C.byref(handle)))
        return cls(None, _handle=handle)

    def NodeSize(self):
        return self.lib.gc_graph_num_nodes(self.handle)

    def SizeInBP(self):
        return self.lib.gc_graph_size_bp(self.handle)

    def array(self, name):
        return _fetch_array(self.lib.gc_graph_array, self.handle, name)

    def trim_host(self):
        """Release the host copy of the MPC index (gc_graph_trim_host): aligning does not read it, saving the index cache does."""
        _check(self.lib.gc_graph_trim_host(self.handle))

    def close(self):
        if self.handle:
            self.lib.gc_graph_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MinimizerSeeder:
    """reference: MinimizerSeeder(graph, k, w, threads, 1 - discardMostNumerousFraction), src/Aligner.cpp:1162"""

    def __init__(self, graph, minimizer_length=15, window_size=20, keep_least_frequent_fraction=1 - 0.001, _handle=None):
        self.lib = load_library()
        self.graph = graph
        self.handle = C.c_void_p()
        if _handle is not None:
            self.handle = _handle
            return
        _check(self.lib.gc_seeder_create(graph.handle, minimizer_length, window_size, keep_least_frequent_fraction, C.byref(self.handle)))

    def array(self, name):
        return _fetch_array(self.lib.gc_seeder_array, self.handle, name)

    def close(self):
        if self.handle:
            self.lib.gc_seeder_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ReadBatch:
    """A batch of reads resident in HBM."""

    def __init__(self, reads):
        self.lib = load_library()
        bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
        self.lengths = np.array([len(b) for b in bs], dtype=np.uint64)
        self.offsets = np.zeros(len(bs) + 1, dtype=np.uint64)
        self.offsets[1:] = np.cumsum(self.lengths)
        blob = b"".join(bs)
        self.blob = blob               # host copy: the output encoders read bases next to the device results
        self.handle = C.c_void_p()
        _check(self.lib.gc_reads_upload(blob, self.offsets.ctypes.data, len(bs), C.byref(self.handle)))

    def close(self):
        if self.handle:
            self.lib.gc_reads_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_RESULT_FIELDS = {
    # name: (count expression)
    "read_seed_off": "n+1", "seed_node": "seeds", "seed_offset": "seeds", "seed_seqpos": "seeds", "seed_goodness": "seeds",
    "read_anchor_off": "n+1", "anchor_x": "anchors", "anchor_y": "anchors", "anchor_path_off": "anchors+1", "anchor_path": "paths",
    "anchor_first_node": "anchors", "anchor_first_offset": "anchors", "anchor_first_seqpos": "anchors",
    "anchor_last_node": "anchors", "anchor_last_offset": "anchors", "anchor_last_seqpos": "anchors", "anchor_score": "anchors",
    "read_chain_off": "n+1", "chain": "chains", "chain_score": "n",
    "failed_assertion": "n", "capacity_exceeded": "n", "seeds_extended": "n", "seeds_extended_long": "n", "flatten_ties": "n", "flatten_ties_long": "n",
    "read_longall_off": "n+1", "longall_start": "longs", "longall_end": "longs", "longall_score": "longs",
}


class _ResultHolder:
    """Owns one gc_result; frees it when the last reference goes away."""

    def __init__(self, lib, res):
        self.lib, self.res = lib, res

    def __del__(self):
        try:
            if self.res:
                self.lib.gc_result_free(self.res)
                self.res = None
        except Exception:
            pass


class _ResultArray(np.ndarray):
    """ndarray view of C memory that keeps the owning gc_result alive: every array (and every slice or view made from one, through
    numpy's .base chain) references the holder, so `x = aligner.align_batch(b)["chain"]` stays valid after the dict is gone."""
    _holder = None


class BatchResult(dict):
    """dict of numpy arrays that are VIEWS of the C result (no copies on the hot path). Each array keeps the C result alive
    (see _ResultArray); the memory is freed when the last of them goes away."""
    _holder = None


class Aligner:
    """Batched stand-in for the reference's per-read hot path (src/Aligner.cpp:601-922)."""

    def __init__(self, graph, seeder, bandwidth=10, split_len=35, split_gap=35, colinear_gap=10000, seed_density=10.0, keep_traces=False, keep_seeds=False, long_pass=False, stitch=True, edit_distances=True, chain_traces=None, e_cutoff=-1.0, capacities=None, device_output=0):
        """capacities: {field of gc_capacities: value} for the device-side tables (default: all automatic).
        device_output: gc_params::device_output - 1 / 2: the final alignments' GAF path and CIGAR text (= / X or M items), + 4: their vg::Path bytes, written
        by the device from the traces it holds; align_batch(gaf_names=...) then needs no keep_traces."""
        self.lib = load_library()
        self.graph = graph
        self.seeder = seeder
        self.params = GcParams()
        self.lib.gc_params_default(C.byref(self.params))
        self.params.bandwidth = bandwidth
        self.params.split_len = split_len
        self.params.split_gap = split_gap
        self.params.colinear_gap = colinear_gap
        self.params.seed_density = seed_density
        self.params.keep_traces = int(keep_traces)
        self.params.keep_seeds = int(keep_seeds)
        self.params.long_pass = int(long_pass)
        self.params.stitch = int(stitch)
        self.params.edit_distances = int(edit_distances)
        # the chained alignment's trace costs a k_edit_path run, the op strings coming down and ~13 B per trace cell of result memory; without
        # the whole-read pass EVERY read with a stitched path "wins" (src/Aligner.cpp:905: nothing to beat), so a caller that only wants anchors
        # and chains (long_pass=False) gets none unless it asks (chain_traces=1: winners, 2: every read)
        self.params.chain_traces = int(chain_traces) if chain_traces is not None else (1 if long_pass else 0)
        self.params.e_cutoff = float(e_cutoff)
        self.params.device_output = int(device_output)
        for name, value in (capacities or {}).items():
            if name not in dict(GcCapacities._fields_) or name == "reserved":
                raise ValueError("no such capacity: " + name)
            setattr(self.params.capacity, name, int(value))
        self.stream = C.c_void_p()
        _check(self.lib.gc_stream_create(C.byref(self.stream)))

    def align_batch(self, batch, gaf_names=None, cigar_match_mismatch_merge=False, other_formats=False, formats=None, gam_level=None):
        """Runs the hot path for a ReadBatch; returns a dict of arrays. With gaf_names (one id per read; needs long_pass and
        keep_traces or device_output) the dict also holds "gaf" (bytes: the reference's GAF lines) and "gaf_chained_skipped"; with other_formats also "json" (JSON
        lines) and "gam" (gzip members of framed vg::Alignment messages; gam_level: their zlib level, or GAM_DEVICE_HUFFMAN for the device's deflate)."""
        res = _P(GcResult)()
        _check(self.lib.gc_align_batch(self.graph.handle, self.seeder.handle, self.stream, batch.handle, C.byref(self.params), C.byref(res)))
        holder = _ResultHolder(self.lib, res)
        if True:
            gaf = None
            if gaf_names is not None:
                # formats: which of "gaf", "json", "gam" to produce (default: GAF, all three with other_formats)
                want = tuple(formats) if formats is not None else (("gaf", "json", "gam") if other_formats else ("gaf",))
                gaf = self._format(res, batch, gaf_names, want, cigar_match_mismatch_merge, gam_level)
            r = res.contents
            n = int(r.n_reads)

            def arr(ptr, count):
                if not count:
                    return np.zeros(0, dtype=np.int64)
                a = np.ctypeslib.as_array(ptr, shape=(count,)).view(_ResultArray)
                a._holder = holder
                return a

            out = BatchResult()
            out._holder = holder
            out["read_seed_off"] = arr(r.read_seed_off, n + 1)
            seeds = int(out["read_seed_off"][-1])
            out["read_anchor_off"] = arr(r.read_anchor_off, n + 1)
            anchors = int(out["read_anchor_off"][-1])
            out["anchor_path_off"] = arr(r.anchor_path_off, anchors + 1)
            paths = int(out["anchor_path_off"][-1])
            out["read_chain_off"] = arr(r.read_chain_off, n + 1)
            chains = int(out["read_chain_off"][-1])
            out["read_longall_off"] = arr(r.read_longall_off, n + 1)
            longs = int(out["read_longall_off"][-1])
            out["read_path_off"] = arr(r.read_path_off, n + 1)
            cells_path = int(out["read_path_off"][-1])
            out["path_node"] = arr(r.path_node, cells_path)
            out["path_first_offset"] = arr(r.path_first_offset, n)
            out["path_last_offset"] = arr(r.path_last_offset, n)
            out["path_cells"] = arr(r.path_cells, n)
            out["read_long_off"] = arr(r.read_long_off, n + 1)
            out["long_index"] = arr(r.long_index, int(out["read_long_off"][-1]))
            out["long_edit_distance"] = arr(r.long_edit_distance, n)
            out["chain_edit_distance"] = arr(r.chain_edit_distance, n)
            out["chained_better"] = arr(r.chained_better, n)
            out["read_chain_trace_off"] = arr(r.read_chain_trace_off, n + 1)
            ctrace = int(out["read_chain_trace_off"][-1])
            for name in ("chain_trace_node", "chain_trace_offset", "chain_trace_seqpos", "chain_trace_switch"):
                out[name] = arr(getattr(r, name), ctrace)
            out["chain_aln_start"] = arr(r.chain_aln_start, n)
            out["chain_aln_end"] = arr(r.chain_aln_end, n)
            counts = {"n": n, "n+1": n + 1, "seeds": seeds, "anchors": anchors, "paths": paths, "chains": chains, "longs": longs}
            for name, expr in _RESULT_FIELDS.items():
                if name in out:
                    continue
                out[name] = arr(getattr(r, name), counts[expr])
            if self.params.keep_traces:
                out["long_trace_off"] = arr(r.long_trace_off, longs + 1)
                lcells = int(out["long_trace_off"][-1]) if longs else 0
                for name in ("long_trace_node", "long_trace_offset", "long_trace_seqpos", "long_trace_switch"):
                    out[name] = arr(getattr(r, name), lcells)
            if self.params.keep_traces == 1:   # (2: the alignments' traces only)
                out["anchor_trace_off"] = arr(r.anchor_trace_off, anchors + 1)
                cells = int(out["anchor_trace_off"][-1])
                for name in ("anchor_trace_node", "anchor_trace_offset", "anchor_trace_seqpos", "anchor_trace_switch"):
                    out[name] = arr(getattr(r, name), cells)
            if self.params.device_output:
                out["read_out_off"] = arr(r.read_out_off, n + 1)
                n_out = int(out["read_out_off"][-1])
                out["out_source"] = arr(r.out_source, n_out)
                out["out_numbers"] = arr(r.out_numbers, 12 * n_out)
                for name in ("out_path_off", "out_cigar_off", "out_vg_off"):
                    out[name] = arr(getattr(r, name), n_out + 1)
            out["counters"] = np.array(list(r.counters), dtype=np.uint64)
            out["counters_long"] = np.array(list(r.counters_long), dtype=np.uint64)
            out["kernel_us"] = np.array(list(r.kernel_us))
            out["host_us"] = np.array(list(r.host_us))
            if gaf is not None:
                out.update(gaf[0])
                out["gaf_chained_skipped"] = gaf[1]
            return out   # arrays keep the C ABI's dtypes (uint32/uint64/...) and are views: no copies on the hot path

    def _format(self, res, batch, names, formats, cigar_match_mismatch_merge=False, gam_level=None):
        if not isinstance(names, C.Array):
            names = (C.c_char_p * len(names))(*[n.encode() if isinstance(n, str) else bytes(n) for n in names])

        def encode(fn, *extra):
            text, length, skipped = C.c_void_p(), C.c_uint64(), C.c_uint64()
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p] + [C.c_int] * len(extra) + [C.c_void_p, C.c_void_p, C.c_void_p]
            _check(fn(self.graph.handle, res, names, batch.blob, batch.offsets.ctypes.data, *extra, C.byref(text), C.byref(length), C.byref(skipped)))
            data = C.string_at(text.value, length.value)
            self.lib.gc_free(text)
            return data, int(skipped.value)

        texts, skipped = {}, 0
        for fmt in formats:
            if fmt == "gaf":
                texts[fmt], skipped = encode(self.lib.gc_format_gaf, int(cigar_match_mismatch_merge))
            elif fmt == "gam" and gam_level is not None:
                texts[fmt], skipped = encode(self.lib.gc_format_gam_level, int(gam_level))
            else:
                texts[fmt], skipped = encode(self.lib.gc_format_json if fmt == "json" else self.lib.gc_format_gam)
        return texts, skipped

    def format_batch(self, out, batch, names, formats=("gaf",), cigar_match_mismatch_merge=False, gam_level=None):
        """The writers on a result align_batch returned earlier (gc_format_gaf / _json / _gam): needs the graph and the result, not the stream - a host can
        format one batch while the stream aligns the next. names: one id per read, or a prebuilt ctypes array of them. gam_level: zlib level of the GAM's gzip
        members (default: the reference's, Z_DEFAULT_COMPRESSION). Returns ({format: bytes}, chained winners the result held no trace for)."""
        return self._format(out._holder.res, batch, names, tuple(formats), cigar_match_mismatch_merge, gam_level)

    def align_reads(self, reads, **kw):
        batch = ReadBatch(reads)
        try:
            return self.align_batch(batch, **kw)
        finally:
            batch.close()

    def close(self):
        if self.stream:
            self.lib.gc_stream_destroy(self.stream)
            self.stream = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
