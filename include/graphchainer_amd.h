/* graphchainer_amd - C ABI of the MI355X-native GraphChainer hot path.
 *
 * The reference (algbio/GraphChainer) has no plugin/FFI layer; its de-facto seam is the free-function
 * facade src/GraphAlignerWrapper.h:39-47 plus MinimizerSeeder::getSeeds (src/MinimizerSeeder.h:33) and
 * AlignmentGraph::colinearChaining (src/AlignmentGraph.h:121), all called per read from
 * runComponentMappings (src/Aligner.cpp:538,560,565,660,666,691,735). This library replaces that
 * per-read call sequence with one batched call; every entry point cites what it stands in for.
 * INTEGRATION.md shows the shim a maintainer would add to src/Aligner.cpp.
 *
 * Conventions: plain pointers and sizes, no C++/torch types. Every function returns 0 on success and
 * a negative gc_status otherwise; gc_last_error() gives the message for the calling thread. Handles are
 * immutable after creation and may be shared between threads; one gc_align_batch per gc_stream at a
 * time (the reference's rule of one AlignerGraphsizedState per worker, src/Aligner.cpp:469).
 * Per-read failures (the reference's per-read catch of AssertionFailure, src/Aligner.cpp:585-592) and per-read
 * capacity overflows of this library are reported in the result arrays (failed_assertion, capacity_exceeded),
 * never as a call failure.
 */
#ifndef GRAPHCHAINER_AMD_H
#define GRAPHCHAINER_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gc_status {
	GC_OK = 0,
	GC_ERR_INVALID = -1,      /* bad argument */
	GC_ERR_GRAPH = -2,        /* invalid graph (reference: InvalidGraphException, cyclic graph) */
	GC_ERR_DEVICE = -3,       /* HIP error or no device: the product path never falls back to the CPU */
	GC_ERR_INTERNAL = -4
} gc_status;

typedef struct gc_graph gc_graph;     /* AlignmentGraph + MPC index, resident in HBM */
typedef struct gc_seeder gc_seeder;   /* MinimizerSeeder index, resident in HBM */
typedef struct gc_stream gc_stream;   /* per-worker reusable state (HIP stream + arenas) */
typedef struct gc_reads gc_reads;     /* a batch of reads uploaded to HBM */

/* Capacities of the device-side tables one batch uses. The reference grows std::vectors and hash maps as it goes; a kernel works in tables sized before the launch.
 * Every table is PER EXTENSION or PER READ, never per batch, and what outgrows one never fails the call: the work is rerun with more room (second launch, other
 * kernel layout, larger pool) and, only if that overflows too, the READ is flagged in gc_result::capacity_exceeded and reported without the affected part.
 * 0 = automatic (the values in brackets, derived from the batch's longest read L); set a field only to trade memory against reruns on unusual inputs
 * (dense variant clusters, very noisy long reads). The same knobs exist as GC_* environment variables for experiments; the environment wins. */
typedef struct gc_capacities {
	int64_t ext_max_items;          /* fragment extensions (k_extend): (slice, node) tiles per extension [72; the retry launch has 16x] (GC_TEST_EXT_MAX_ITEMS) */
	int64_t ext_max_pending;        /* ... entries of the per-slice node queue [48; retry 16x] (GC_TEST_EXT_MAX_PENDING) */
	int64_t ext_max_trace;          /* ... trace cells per extension [192; retry 16x] (GC_TEST_EXT_MAX_TRACE) */
	int64_t long_max_items;         /* whole-read extensions (k_long_extend): tiles per extension [max(8192, 24 per slice)] (GC_TEST_LONG_MAX_ITEMS) */
	int64_t long_column_store;      /* ... DP columns kept for the backtrace per extension [3 L + 4096]; -1: keep none, the backtrace recomputes its tiles (GC_TEST_LONG_MAX_COLS) */
	int64_t long_cells_per_base;    /* merged trace cells per read base in the batch's pool [8, tripled and the pass rerun when a batch overflows it; the stream remembers]
	                                 * (GC_TEST_LONG_CELLS_PER_BASE; setting it pins the pool: overflowing reads are flagged instead) */
	int64_t long_scratch_bytes;     /* upper bound of the whole-read pass's per-wave scratch [48 GiB; 0.8 MB per resident wave at L = 10 kb, 2.4 MB at 50 kb] (GC_TEST_LONG_SCRATCH_GB) */
	int64_t stitch_set_max;         /* chain stitching (k_stitch): nodes of one bridged region searched on the device [automatic]; larger regions go to the host (GC_TEST_STITCH_SET_MAX) */
	int64_t stitch_bfs_cap;         /* ... frontier entries of one bridge search [automatic] (GC_TEST_STITCH_BFS_CAP) */
	int64_t reserved[3];            /* must be 0 */
} gc_capacities;

/* Alignment parameters. Defaults = the reference's chaining-mode presets, src/AlignerMain.cpp:186-209. */
typedef struct gc_params {
	int32_t bandwidth;          /* -b, initialBandwidth (10) */
	int32_t split_len;          /* --colinear-split-len (35) */
	int32_t split_gap;          /* --colinear-split-gap (35); --sampling-step 0.5 == 18 */
	int64_t colinear_gap;       /* --colinear-gap (10000) */
	double  seed_density;       /* --seeds-minimizer-density (10) */
	int32_t min_cluster_size;   /* --seeds-clustersize (1) */
	int32_t long_pass;          /* 1: also run the whole-read GraphAligner pass (src/Aligner.cpp:630-654) */
	int32_t keep_traces;        /* 1: return the traces of every anchor and of every whole-read alignment (parity tests); 2: the alignments' traces only
	                             *    (long_trace_*: what gc_format_gaf / _json / _gam read; the anchors' traces - 13 B x ~37 cells per anchor - stay on the device) */
	int32_t keep_seeds;         /* 1: return the ordered seed list of every read (seed_* arrays; else they are empty) */
	int32_t stitch;             /* 1: stitch the chain into one path (src/Aligner.cpp:754-822): read_path_off / path_* */
	int32_t edit_distances;     /* 1: GreedyLength selection of the whole-read alignments and the two NW edit distances that pick
	                             *    the winner (src/Aligner.cpp:636-654,845,901-905): read_long_off / long_index / *_edit_distance / chained_better */
	int32_t chain_traces;       /* the chained alignment's trace (edlibAlign(pathseq, read, NW, EDLIB_TASK_PATH) walked over the stitched path,
	                             *    src/Aligner.cpp:845-897): 0 none, 1 for the reads whose chained alignment wins (default; what the output
	                             *    encoders need), 2 for every read with a stitched path. Needs stitch and edit_distances. Cost: one k_edit_path
	                             *    pair per traced read plus ~13 B of result memory per trace cell; with long_pass == 0 every read with a stitched
	                             *    path counts as a winner (nothing to beat, src/Aligner.cpp:905), i.e. mode 1 then traces them all - pass 0 when
	                             *    only anchors and chains are wanted. */
	int32_t device_output;      /* r4: encode the final alignments on the device, where their traces already are, instead of bringing the traces down (keep_traces)
	                             *    for gc_format_*: bit 0 (1) the GAF path and cg:Z: CIGAR text with = / X items, bit 1 (2) the same with M items
	                             *    (--cigar-match-mismatch merge; give one of the two), bit 2 (4) the vg::Path wire bytes that gc_format_gam / _json wrap.
	                             *    Fills gc_result::read_out_off / out_*; needs long_pass and edit_distances. (Sits in what used to be padding: the layout of the
	                             *    other fields is unchanged.) */
	double  e_cutoff;           /* --E-cutoff (src/AlignerMain.cpp:159,271-274; SelectECutoff src/AlignmentSelection.cpp:57-61,91-99):
	                             *    alignments with a larger E-value are dropped before selection; -1 (default) keeps all */
	gc_capacities capacity;     /* sizes of the device-side tables (all 0 = automatic: sized from the batch) */
} gc_params;

void gc_params_default(gc_params* p);

/* ---- graph (replaces getGraph + AlignmentGraph::buildMPC, src/Aligner.cpp:1137-1156) -------------- */

/* Loads a GFA (S/L lines, 0M overlaps), builds the split-node DAG with the reference's node numbering,
 * the MPC index, and uploads everything to the current HIP device. */
int gc_graph_create_from_gfa(const char* gfa_path, gc_graph** out);

/* Same, from arrays the existing C++ host already holds (src/AlignmentGraph.h:145-172): for a host that
 * keeps its own AlignmentGraph. Adjacency is CSR in the reference's neighbour order.
 * lookup_order (optional): the bigraph node ids in the iteration order of the host's nodeLookup hash map. The reference's
 * minimizer index enumerates nodes in that order (src/MinimizerSeeder.cpp:299-365), and the order of a k-mer's position
 * list - and with it tie-breaks among equally good seeds - follows from it; with it the index built on this graph is
 * identical to the one built on the gc_graph_create_from_gfa graph (tested). Without it (NULL) nodes are enumerated in
 * ascending id, which can order position lists differently. Node names are only known to gc_graph_create_from_gfa (the
 * output encoders print numeric ids otherwise). */
typedef struct gc_graph_desc {
	uint64_t n_nodes;                 /* split nodes */
	uint64_t first_ambiguous;         /* nodes >= this index use ambiguous_seq */
	const uint8_t*  node_length;      /* [n] 1..64 */
	const uint32_t* node_offset;      /* [n] offset inside the original (bigraph) node */
	const int32_t*  node_ids;         /* [n] bigraph node id (2*gfa_id + strand) */
	const uint64_t* node_seq;         /* [2*first_ambiguous] 2 bits/bp */
	const uint64_t* ambiguous_seq;    /* [4*(n-first_ambiguous)] one-hot A,T,C,G words */
	const uint64_t* in_off;  const uint32_t* in_adj;    /* CSR [n+1], [m] */
	const uint64_t* out_off; const uint32_t* out_adj;
	const uint32_t* component_number; /* [n] topological rank (src/AlignmentGraph.cpp:1008) */
	const uint32_t* chain_number;     /* [n] */
	const uint64_t* chain_approx_pos; /* [n] */
	const int32_t*  lookup_order;     /* [n_lookup] or NULL, see above */
	uint64_t n_lookup;
} gc_graph_desc;
int gc_graph_create(const gc_graph_desc* desc, gc_graph** out);
void gc_graph_destroy(gc_graph* g);

/* Read-only views of the host copy (for parity tests and for hosts that want the numbering). */
uint64_t gc_graph_num_nodes(const gc_graph* g);
uint64_t gc_graph_size_bp(const gc_graph* g);
/* name in {"nodeLength","nodeOffset","nodeIDs","reverse","componentNumber","chainNumber","chainApproxPos",
 * "component_map","out_off","out_adj","in_off","in_adj","mpc_width","firstAmbiguous","nodeSeq","ambiguousSeq" (64-bit
 * patterns, gc_graph_desc layout),"lookupOrder", and the MPC index in global node ids: "component_idx","topo_id","mpc_path_comp",
 * "mpc_path_off","mpc_path_nodes","paths_off","paths","back_off","back_node","back_path"}; returns a malloc'd int64 array
 * (free with gc_free). */
int gc_graph_array(const gc_graph* g, const char* name, int64_t** out, uint64_t* count);
/* Frees the HOST copy of the MPC index (path cover, per-node path lists, backward links, topological orders: ~33 bytes of host memory per graph base, 100 GB at 3.1 Gbp);
 * the device keeps its own and aligning reads nothing of it. For hosts that do not write the index cache afterwards: gc_index_save and the MPC arrays of gc_graph_array
 * ("mpc_*", "paths*", "back_*", "topo_id") return GC_ERR_INVALID on a trimmed graph. No reference counterpart (the reference's AlignmentGraph keeps everything). */
int gc_graph_trim_host(gc_graph* g);

/* ---- seeder (replaces MinimizerSeeder::MinimizerSeeder, src/Aligner.cpp:1162) --------------------- */
int gc_seeder_create(const gc_graph* g, int32_t k, int32_t w, double keep_least_frequent_fraction, gc_seeder** out);
void gc_seeder_destroy(gc_seeder* s);
int gc_seeder_array(const gc_seeder* s, const char* name, int64_t** out, uint64_t* count);   /* "kmers","start","positions","maxcount","k","w" */

/* ---- index cache --------------------------------------------------------------
 * The start-up work above (GFA parse, node splitting, topological order, greedy path cover + max-flow shrink, MPC index,
 * minimizer scan: src/AlignmentGraph.cpp:1267-1391,1465-1495, src/MinimizerSeeder.cpp:299-492) written once to a file and
 * loaded instead of recomputed. The reference declares saveMPC/loadMPC (src/AlignmentGraph.h:96-97) with empty bodies and
 * rebuilds everything on every run; these entry points are what those two would bind. The file is checksummed and
 * versioned; a damaged, truncated or other-version file is refused (GC_ERR_GRAPH), never used.
 *   gc_index_build  host only, no HIP device needed: builds graph + MPC (+ minimizer index when k > 0) and writes the cache.
 *   gc_index_save   writes the cache from objects already built (seeder may be NULL).
 *   gc_index_load   reads the cache and uploads to the current device; *seeder_out is NULL when the file holds no
 *                   minimizer index (seeder_out may itself be NULL to skip it). Results are identical to the built objects'.
 *   gc_index_check  host only: verifies the file (checksum, bounds, re-serialises to the same bytes) and reports
 *                   info8 = {version, split nodes, bp, has seeder, k, w, distinct k-mers, positions} (info8 may be NULL). */
int gc_index_build(const char* gfa_path, int32_t k, int32_t w, double keep_least_frequent_fraction, const char* cache_path);
int gc_index_save(const gc_graph* g, const gc_seeder* s, const char* cache_path);
int gc_index_load(const char* cache_path, gc_graph** graph_out, gc_seeder** seeder_out);
int gc_index_check(const char* cache_path, uint64_t* info8);

/* ---- streams and read batches --------------------------------------------------------------------- */
int gc_stream_create(gc_stream** out);
void gc_stream_destroy(gc_stream* st);

/* Uploads n reads (ASCII, concatenated; read i = bases[offsets[i] .. offsets[i+1])) to HBM. */
int gc_reads_upload(const char* bases, const uint64_t* offsets, uint64_t n, gc_reads** out);
void gc_reads_destroy(gc_reads* r);

/* ---- the hot path ---------------------------------------------------------------------------------- */

/* Flat (CSR) result of one batch. All arrays are host memory owned by the result; free with
 * gc_result_free. Node ids are split-node indices unless stated otherwise. */
typedef struct gc_result {
	uint64_t n_reads;
	/* seeds in fragment-pass order: MinimizerSeeder::getSeeds + OrderSeeds + sort by seqPos
	 * (src/Aligner.cpp:660-667) */
	uint64_t* read_seed_off;      /* [n_reads+1] */
	uint32_t* seed_node; uint32_t* seed_offset; uint32_t* seed_seqpos; uint64_t* seed_goodness;
	/* anchors = fragment alignments (src/Aligner.cpp:706-729) */
	uint64_t* read_anchor_off;    /* [n_reads+1] */
	uint32_t* anchor_x; uint32_t* anchor_y;
	uint64_t* anchor_path_off;    /* [n_anchors+1] */
	uint32_t* anchor_path;
	uint32_t* anchor_first_node; uint32_t* anchor_first_offset; uint32_t* anchor_first_seqpos;
	uint32_t* anchor_last_node;  uint32_t* anchor_last_offset;  uint32_t* anchor_last_seqpos;
	int32_t*  anchor_score;
	/* optional traces (keep_traces): bigraph node id, offset in original node, seqPos in the fragment */
	uint64_t* anchor_trace_off;   /* [n_anchors+1] or NULL */
	int32_t*  anchor_trace_node; uint32_t* anchor_trace_offset; uint32_t* anchor_trace_seqpos; uint8_t* anchor_trace_switch;
	/* chain = AlignmentGraph::colinearChaining (src/Aligner.cpp:735): anchor indices local to the read */
	uint64_t* read_chain_off;     /* [n_reads+1] */
	uint32_t* chain;
	uint64_t* chain_score;        /* [n_reads] covered read bases */
	/* whole-read pass (long_pass): all alignments of AlignOneWay(sloppy), src/Aligner.cpp:565 */
	uint64_t* read_longall_off;   /* [n_reads+1] */
	uint32_t* longall_start; uint32_t* longall_end; uint32_t* longall_score;
	uint64_t* long_trace_off;     /* [n_longall+1] */
	int32_t*  long_trace_node; uint32_t* long_trace_offset; uint32_t* long_trace_seqpos; uint8_t* long_trace_switch;
	/* per read flags */
	uint8_t*  failed_assertion;   /* [n_reads] the reference would have thrown on this read */
	uint8_t*  capacity_exceeded;  /* [n_reads] 1: a capacity of THIS library was exceeded for the read (an extension with more tiles than even the
	                               *    retry launch holds, more whole-read alignments than max(32, longest read / 512), a full cell pool, an NW band
	                               *    beyond the kernel's range): its results are incomplete. The reference has no such limits; the batch always
	                               *    completes and every other read is unaffected. */
	uint64_t* seeds_extended;     /* [n_reads] fragment pass (stats.seedsExtended, src/Aligner.cpp:705) */
	uint64_t* seeds_extended_long; /* [n_reads] whole-read pass (AlignmentResult::seedsExtended) */
	/* chain stitching (stitch): the longest stitched piece of the chain (src/Aligner.cpp:754-822) as its split-node
	 * path and the offsets of its first and last base. pathToTrace (src/Aligner.cpp:409-424) expands this to the
	 * reference's `longest` vector, one (node, offset) cell per graph base; path_cells is that vector's size. */
	uint64_t* read_path_off;      /* [n_reads+1] into path_node */
	uint32_t* path_node;
	uint32_t* path_first_offset; uint32_t* path_last_offset;   /* [n_reads] */
	uint64_t* path_cells;         /* [n_reads] */
	/* decision (edit_distances): whole-read alignments kept by SelectAlignments(GreedyLength) (src/Aligner.cpp:636-639) as
	 * indices into the read's longall_* list, best first; the NW edit distance of the best one's path against the read
	 * (:645) and of the stitched path against the read (:845), -1 where there is none; and the test of :901-905 */
	uint64_t* read_long_off;      /* [n_reads+1] */
	uint32_t* long_index;
	int64_t*  long_edit_distance; int64_t* chain_edit_distance;   /* [n_reads] */
	uint8_t*  chained_better;     /* [n_reads] 1: the chained alignment is the read's result, 0: the selected whole-read alignments are */
	/* the chained alignment (chain_traces): its trace in output coordinates like long_trace_* (bigraph node id, offset in the original
	 * node, read position, "next cell is in another split node"), one cell per op of edlib's alignment (src/Aligner.cpp:855-887);
	 * alignmentStart / alignmentEnd (:894-895). Its alignmentScore is chain_edit_distance, its trace score 0 (never set, :739,891). */
	uint64_t* read_chain_trace_off; /* [n_reads+1] */
	int32_t*  chain_trace_node; uint32_t* chain_trace_offset; uint32_t* chain_trace_seqpos; uint8_t* chain_trace_switch;
	uint32_t* chain_aln_start; uint32_t* chain_aln_end;   /* [n_reads]; 0,0 where there is no chained alignment */
	/* work counters of the fragment pass: [0] dp tiles, [1] recompute tiles (last-slice flatten + backtrace),
	 * [2] column steps, [3] trace items, [4] extensions, [5] backtrace tiles (subset of [1]);
	 * [6] (r5) times the batch's fragment pipeline ran again because its trace pool or anchor path pool - sized by what the stream's earlier batches used - was too small;
	 * [7] reads whose chain was stitched on the host because it did not fit the stitching kernel's tables */
	uint64_t counters[8];
	uint64_t counters_long[8];    /* the same for the whole-read pass */
	/* device time of each kernel of this batch in microseconds (HIP events on the stream):
	 * [0] seed lookup, [1] fragment extension (k_extend: every lazy round's launches with their retry launches, each between an event pair of its own), [2] anchor build
	 * (k_build_anchors, every round, likewise), [3] chaining, [4] whole-read extension kernels, summed over all
	 * rounds of all read groups (the groups run concurrently on their own streams, so this can exceed the wall clock),
	 * [5] whole-read pass wall clock, first group's start to last group's end (host clock; overlaps 1-3) */
	double kernel_us[8];
	double host_us[4];            /* wall time: [0] host seed glue, [1] result assembly, [2] seed lookup + transfers, [3] extension..chaining + transfers */
	/* Final alignments encoded on the device (params->device_output; NULL otherwise): one entry per alignment the reference would write, in its order
	 * (src/Aligner.cpp:901-920,1003-1023: the chained alignment when it won, else the selected whole-read alignments sorted by alignmentStart).
	 * out_source[k] = 0: the pieces below hold it (GraphAlignerGAFAlignment::traceToAlignment / GraphAlignerVGAlignment::traceToAlignment run on the device);
	 * 1: it is the read's chained alignment, whose trace the host builds (chain_trace_*): the gc_format_* functions encode it from there.
	 * out_numbers[12 k ..]: node path length, start, end (GAF columns 7-9), matches, mismatches, insertions, deletions, trace cells (column 11), alignmentStart,
	 * alignmentEnd, path steps (mappings), alignmentScore. */
	uint64_t* read_out_off;       /* [n_reads+1] */
	uint8_t*  out_source;         /* [n_out] */
	uint64_t* out_numbers;        /* [12 * n_out] */
	uint64_t* out_path_off;  char* out_path_text;     /* [n_out+1]; GAF column 6, e.g. ">12>13<7" (device_output & 3) */
	uint64_t* out_cigar_off; char* out_cigar_text;    /* [n_out+1]; the cg:Z: value (device_output & 3) */
	uint64_t* out_vg_off;    uint8_t* out_vg_path;    /* [n_out+1]; the alignment's vg::Path in proto3 wire format, src/vg.proto:52-109 (device_output & 4) */
	/* r5 (appended: the layout above is unchanged). How often this read exercised the ONE rule this library defines instead of reproducing: the reference's
	 * flattenLastSliceEnd (src/GraphAlignerBitvectorCommon.h:1170-1229) takes the last partial slice's minimum cell with a strict '<' in the iteration order of a
	 * parallel-hashmap (src/NodeSlice.h:54) - a library absent from the reference tree here - so when the minimum is attained in MORE THAN ONE NODE the cell the
	 * reference starts its backtrace from depends on that order; this library takes the node that entered the slice's band first. flatten_ties[r] /
	 * flatten_ties_long[r] = the extensions of read r (fragment pass / whole-read pass; only extensions the reference would have run) with such a tie.
	 * 0 for a read means: nothing in its result depends on the defined order. DESIGN.md §7 says how often a tie changes an output. */
	uint32_t* flatten_ties;       /* [n_reads] */
	uint32_t* flatten_ties_long;  /* [n_reads] */
	int32_t device_output;        /* the gc_params::device_output the batch ran with: gc_format_gaf refuses a cigar_match_mismatch_merge that differs from the pieces' style */
} gc_result;

/* Runs seeding, fragment seed-extension, anchor construction and co-linear chaining (and, with
 * params->long_pass, the whole-read pass) for every read of the batch. Replaces, per read:
 * MinimizerSeeder::getSeeds, OrderSeeds, AlignOneWay, AlignmentGraph::colinearChaining. */
int gc_align_batch(const gc_graph* g, const gc_seeder* s, gc_stream* st, const gc_reads* reads, const gc_params* params, gc_result** out);
void gc_result_free(gc_result* r);

/* gc_result_free keeps the large arrays of freed results (trace cells, output text: up to 24 GB in all) for the next batch's result - fresh memory of that size is
 * mapped and zero-filled page by page every batch otherwise; gc_reads_destroy and the device deflate keep their device / pinned blocks the same way (up to 24 GB and 4 GB per cache).
 * gc_result_cache_trim gives everything that is held back to the allocator (e.g. when a host stops aligning). */
void gc_result_cache_trim(void);

const char* gc_last_error(void);
void gc_free(void* p);
/* Global (NW) edit distance of each pair (a[a_off[i]..a_off[i+1]), b[b_off[i]..b_off[i+1])) on the GPU: the value
 * edlibAlign(a, |a|, b, |b|, edlibNewAlignConfig(-1, EDLIB_MODE_NW, EDLIB_TASK_DISTANCE, NULL, 0)).editDistance returns at
 * src/Aligner.cpp:645 and :845 (characters compare by equality, as in edlib's default alphabet handling). */
int gc_edit_distance(const char* a, const uint64_t* a_off, const char* b, const uint64_t* b_off, uint64_t n_pairs, int64_t* out);
/* The alignment edlibAlign(a, |a|, b, |b|, edlibNewAlignConfig(-1, EDLIB_MODE_NW, EDLIB_TASK_PATH, NULL, 0)) returns at src/Aligner.cpp:845
 * for each pair, on the GPU: distance[i] = editDistance, ops (0 match, 1 letter of a alone, 2 letter of b alone, 3 mismatch) of pair i at
 * ops[ops_off[i] .. ops_off[i] + ops_len[i]); the caller provides ops_off with room for |a_i| + |b_i| ops per pair. ops_len[i] = 0 where
 * edlib returns no alignment (an empty side). edlib's own choice among the optimal alignments is reproduced (Hirschberg split order and
 * traceback preference, edlib/src/edlib.cpp:917-1419). */
int gc_edit_path(const char* a, const uint64_t* a_off, const char* b, const uint64_t* b_off, uint64_t n_pairs, const uint64_t* ops_off, uint8_t* ops, uint32_t* ops_len, int64_t* distance);

/* The E-value --E-cutoff compares (EValueCalculator, src/EValue.cpp; host only, no device needed): out2 = {alignment score, E-value}
 * of an alignment of alignment_length read bases with num_edits edits, for a graph of database_size bp and a read of query_size bp. */
int gc_evalue(double min_identity, uint64_t database_size, uint64_t query_size, uint64_t alignment_length, uint64_t num_edits, double* out2);

/* ---- output ------------------------------------------------------------------------ */

/* GAF text of the batch's final alignments, one line per alignment in the reference's order (AddGAFLine +
 * GraphAlignerGAFAlignment::traceToAlignment, src/GraphAlignerGAFAlignment.h:38-196; the per-read list sorted by
 * alignmentStart, src/Aligner.cpp:1022, written by writeGAFToQueue :300-311). `result` must come from gc_align_batch with
 * long_pass, keep_traces, edit_distances and chain_traces >= 1; bases/offsets are the read batch as given to gc_reads_upload;
 * (or, r4, with device_output instead of keep_traces: the lines are then put together from the pieces the device wrote - no trace comes down);
 * with device_output the pieces' CIGAR style is the one the batch was aligned with: cigar_match_mismatch_merge must agree with it (GC_ERR_INVALID otherwise);
 * read_names[i] is the FASTQ id. A read whose chained alignment won (chained_better) is written from its chain_trace_*
 * (src/Aligner.cpp:901-920); n_chained_skipped counts winners the result holds no trace for (0 unless chain_traces was 0).
 * *out_text is malloc'd (gc_free), NUL-terminated, *out_len bytes long. */
int gc_format_gaf(const gc_graph* g, const gc_result* result, const char* const* read_names, const char* bases, const uint64_t* offsets,
                  int cigar_match_mismatch_merge, char** out_text, uint64_t* out_len, uint64_t* n_chained_skipped);

/* The same alignments as vg::Alignment messages (GraphAlignerVGAlignment::traceToAlignment + AddAlignment +
 * replaceDigraphNodeIdsWithOriginalNodeIds): JSON lines as MessageToJsonString(preserve_proto_field_names) prints them
 * (writeJSONToQueue, src/Aligner.cpp:283-298), or GAM: per read one gzip member holding varint count + (varint size, proto3
 * bytes) per alignment (writeGAMToQueue, src/Aligner.cpp:261-281; the gzip bytes depend on the zlib build, the inflated
 * stream is the reference's). r6: gc_format_gam deflates the members on the device (GC_GAM_DEVICE_LZ below; needs the GPU);
 * gc_format_gam_level(-1) is zlib's default level on the host, the reference's own setting. */
int gc_format_json(const gc_graph* g, const gc_result* result, const char* const* read_names, const char* bases, const uint64_t* offsets,
                   char** out_text, uint64_t* out_len, uint64_t* n_chained_skipped);
int gc_format_gam(const gc_graph* g, const gc_result* result, const char* const* read_names, const char* bases, const uint64_t* offsets,
                  char** out_bytes, uint64_t* out_len, uint64_t* n_chained_skipped);

/* One alignment at a time, for a host that keeps the reference's per-alignment output calls (src/GraphAlignerWrapper.h:43-44; include/graphchainer_amd_shim.hpp
 * forwards AddGAFLine / AddAlignment here). The trace is in output coordinates (bigraph node id, offset in the original node, read position, "the next cell is in
 * another split node": long_trace_* / chain_trace_* / anchor_trace_* of a gc_result, or a reference OnewayTrace). Host code, no device work; free the output with gc_free.
 *   gc_format_gaf_trace  GraphAlignerGAFAlignment::traceToAlignment (src/GraphAlignerGAFAlignment.h:38-196): the GAF line, no newline.
 *   gc_format_vg_trace   GraphAlignerVGAlignment::traceToAlignment + AddAlignment's sequence / query_position + replaceDigraphNodeIdsWithOriginalNodeIds
 *                        (src/GraphAlignerVGAlignment.h:36-163, src/GraphAligner.h:205-212, src/Aligner.cpp:152-165): the vg::Alignment message, proto3 wire bytes.
 *   gc_format_vg_trace_digraph   the same message as AddAlignment alone leaves it (src/GraphAligner.h:205-212): position.node_id = the digraph node id (2 x segment index
 *                        + strand) and no position.name - for a host that keeps the reference's own call to replaceDigraphNodeIdsWithOriginalNodeIds right after AddAlignment
 *                        (src/Aligner.cpp:1009; include/graphchainer_amd_shim.hpp's AddAlignment goes here, so that an unchanged src/Aligner.cpp:1006-1012 gives the reference's ids).
 *   gc_graph_letters     the graph letter under each (node id, offset): TraceItem's graphCharacter (src/GraphAlignerCommon.h:148-153). */
int gc_format_gaf_trace(const gc_graph* g, const char* read_name, const char* sequence, uint64_t sequence_len, const int32_t* node, const uint32_t* offset, const uint32_t* seqpos,
                        const uint8_t* node_switch, uint64_t n, int cigar_match_mismatch_merge, char** out_text, uint64_t* out_len);
int gc_format_vg_trace(const gc_graph* g, const char* read_name, const char* sequence, uint64_t sequence_len, const int32_t* node, const uint32_t* offset, const uint32_t* seqpos,
                       const uint8_t* node_switch, uint64_t n, int32_t score, uint64_t alignment_start, uint64_t alignment_end, char** out_bytes, uint64_t* out_len);
int gc_format_vg_trace_digraph(const gc_graph* g, const char* read_name, const char* sequence, uint64_t sequence_len, const int32_t* node, const uint32_t* offset, const uint32_t* seqpos,
                               const uint8_t* node_switch, uint64_t n, int32_t score, uint64_t alignment_start, uint64_t alignment_end, char** out_bytes, uint64_t* out_len);
int gc_graph_letters(const gc_graph* g, const int32_t* node, const uint32_t* offset, uint64_t n, char* out);

/* gc_format_gam with the zlib level of the gzip members chosen by the caller (-1 = Z_DEFAULT_COMPRESSION, what the reference's GzipOutputStream uses and gc_format_gam
 * gives; 0..9). The inflated stream is the same at every level; deflate at the default level costs ~1 ms of CPU per 10 kb read - more than the whole alignment costs the
 * GPU - so a host that writes GAM at the hot path's rate wants level 1 (or its own compressor on the bytes of level 0), or GC_GAM_DEVICE_HUFFMAN: the members are then
 * deflated on the device, every read's group as one dynamic-Huffman block of literals (no LZ77 matches: a larger file than zlib's, the same inflated stream, and the host
 * only frames the members and computes their CRC-32s). */
#define GC_GAM_DEVICE_HUFFMAN 100
/* r6: the same with LZ77 matches in front of the Huffman stage (one probe of a hash of 4-byte prefixes per position, greedy parse): ~1.2 x zlib's default bytes instead of 2 x,
 * the same inflated stream; what the reference's GzipOutputStream costs the host (src/Aligner.cpp:261-281) stays on the device. */
#define GC_GAM_DEVICE_LZ 101
int gc_format_gam_level(const gc_graph* g, const gc_result* result, const char* const* read_names, const char* bases, const uint64_t* offsets, int level,
                        char** out_bytes, uint64_t* out_len, uint64_t* n_chained_skipped);

/* The device deflate behind GC_GAM_DEVICE_HUFFMAN on the caller's own byte streams: stream i = bytes[offsets[i] .. offsets[i+1]) becomes the gzip member
 * out_bytes[out_offsets[i] .. out_offsets[i+1]) (out_offsets: n + 1 entries, caller's; out_bytes: gc_free). */
int gc_gzip_streams(const uint8_t* bytes, const uint64_t* offsets, uint64_t n, char** out_bytes, uint64_t* out_offsets);
/* ... and the deflate behind GC_GAM_DEVICE_LZ (r6; no counterpart in the reference: its gzip layer is protobuf's GzipOutputStream, src/Aligner.cpp:261-281) */
int gc_gzip_streams_lz(const uint8_t* bytes, const uint64_t* offsets, uint64_t n, char** out_bytes, uint64_t* out_offsets);

/* Test entry (no counterpart in the reference): the permutation the device's replay of libstdc++'s std::sort gives for arrays of 32-bit keys. The reference feeds three UNSTABLE
 * std::sort calls into order-sensitive logic (src/MinimizerSeeder.cpp:497, src/GraphAligner.h:293, src/Aligner.cpp:667), so the permutation libstdc++ produces for equal keys is
 * part of its behaviour; the seed kernels replay that algorithm on the 64 lanes of a wave (csrc/hip/gc_stdsort_wave.hpp) and the tests hold this entry against the local std::sort.
 * Array s = keys[offsets[s] .. offsets[s + 1]); perm_out[offsets[s] + i] = index inside its array of the element that ends at place i. depth_limit < 0: the reference's
 * 2 floor(log2 n); a small value forces introsort's heapsort path. */
int gc_std_sort_permutations(const uint32_t* keys, const uint64_t* offsets, uint64_t n_arrays, int64_t depth_limit, uint32_t* perm_out);

int gc_device_count(void);
int gc_set_device(int device);
/* free / total bytes of the current device's memory (a host that sizes its batches: a 10 k x 10 kb batch in flight holds ~19 GB, a 2 k x 50 kb batch 28-42 GB, beside the device's shared whole-read scratch of 21-27 GB - two of them when small batches run two passes side by side) */
int gc_device_memory(uint64_t* free_bytes, uint64_t* total_bytes);

#ifdef __cplusplus
}
#endif
#endif
