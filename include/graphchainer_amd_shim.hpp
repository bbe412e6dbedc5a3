// graphchainer_amd_shim.hpp - the reference's per-call surface over the batched C ABI.
//
// north_star keeps "the Aligner / GraphAlignerWrapper call surface": src/Aligner.cpp calls, per read,
//   MinimizerSeeder::getSeeds            src/MinimizerSeeder.h:33        (src/Aligner.cpp:538,660)
//   OrderSeeds                           src/GraphAlignerWrapper.h:46    (:560,666)
//   AlignOneWay (whole read, sloppy)     src/GraphAlignerWrapper.h:41    (:565)
//   AlignOneWay (fragment, l / r / offset)                               (:691)
//   AlignmentGraph::colinearChaining     src/AlignmentGraph.h:121        (:735)
//   AddAlignment / AddGAFLine / AddCorrected   src/GraphAlignerWrapper.h:43-45   (:1006-1019)   (r4)
// This header gives those calls, with the reference's signatures, on top of include/graphchainer_amd.h, so the
// reference's own src/Aligner.cpp can link against libgraphchainer_amd.so instead of GraphAlignerWrapper.cpp /
// MinimizerSeeder.cpp / the chaining part of AlignmentGraph.cpp where its tool-chain exists.
//
// How: the hot path is one GPU pipeline per read batch, not five separable calls. The first call that mentions a read
// (getSeeds) runs gc_align_batch for a batch of ONE read on the calling thread's gc_stream and keeps the result; the
// following calls of the same read replay their part of it:
//   getSeeds          -> the read's seeds as the fragment pass uses them (after OrderSeeds and the sort by seqPos; OrderSeeds and
//                        the caller's std::sort by seqPos then have nothing left to change but ties, which nothing reads)
//   AlignOneWay       -> whole read: every alignment of the sloppy pass with its trace; fragment (offset = l): the anchors of
//                        that fragment as alignment items with their traces
//   colinearChaining  -> the chain over exactly those anchors
// A batch of one read leaves the GPU mostly idle: this is the drop-in for linking and checking, INTEGRATION.md §3 (batching
// the dequeued reads) is the form to run. Error convention: a read the reference would have dropped with an
// AssertionFailure (src/Aligner.cpp:585-592,695-703) comes back with what the reference would have kept - nothing after
// the failing fragment - instead of a throw; gcshim::lastReadFailedAssertion() tells.
//
// The header is written against the reference's type NAMES. Include the reference's GraphAlignerWrapper.h first (or, as
// tests/shim/shim_test.cpp does, minimal definitions with the same members): SeedHit, AlignmentResult (+ ::AlignmentItem),
// AlignmentGraph (+ ::Anchor), GraphAlignerCommon<size_t, int32_t, uint64_t>::{OnewayTrace, TraceItem, MatrixPosition,
// AlignerGraphsizedState}. Define GC_SHIM_DEFINE_GLOBALS before including it in ONE translation unit to also get the global
// AlignOneWay / OrderSeeds with the reference's exact signatures.
#ifndef GRAPHCHAINER_AMD_SHIM_HPP
#define GRAPHCHAINER_AMD_SHIM_HPP

#include "graphchainer_amd.h"
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <limits>
#include <string>
#include <vector>

namespace gcshim {

typedef GraphAlignerCommon<size_t, int32_t, uint64_t> Common;

struct Binding {
	gc_graph* graph = nullptr;
	gc_seeder* seeder = nullptr;
	gc_params params {};
	std::vector<int64_t> nodeIDs, nodeOffset;   // split node -> bigraph node id, offset in the original node
	size_t minimizerLength = 15;                // the seeder's k: SeedHit::matchLen (src/MinimizerSeeder.cpp:549)
};
inline Binding& binding() { static Binding b; return b; }

inline std::vector<int64_t> graphArray(const gc_graph* g, const char* name)
{
	int64_t* p = nullptr; uint64_t n = 0;
	if (gc_graph_array(g, name, &p, &n) != GC_OK) throw std::runtime_error(gc_last_error());
	std::vector<int64_t> v(p, p + n);
	gc_free(p);
	return v;
}

// Once, after gc_graph_create* / gc_seeder_create (src/Aligner.cpp:1137-1162). `params` as INTEGRATION.md §2 fills them; the shim
// turns on what the replay needs (whole-read pass, traces, seeds).
inline void bind(gc_graph* graph, gc_seeder* seeder, gc_params params)
{
	Binding& b = binding();
	b.graph = graph; b.seeder = seeder;
	params.long_pass = 1; params.keep_traces = 1; params.keep_seeds = 1; params.stitch = 1; params.edit_distances = 1;
	b.params = params;
	b.nodeIDs = graphArray(graph, "nodeIDs"); b.nodeOffset = graphArray(graph, "nodeOffset");
	{
		int64_t* p = nullptr; uint64_t n = 0;
		if (gc_seeder_array(seeder, "k", &p, &n) != GC_OK || n != 1) throw std::runtime_error(gc_last_error());
		b.minimizerLength = (size_t)p[0];
		gc_free(p);
	}
}

// One per worker thread: the stand-in for AlignerGraphsizedState (src/Aligner.cpp:469).
class Session {
public:
	~Session() { drop(); if (stream) gc_stream_destroy(stream); }
	const gc_result& of(const std::string& sequence)
	{
		if (result && sequence == current) return *result;
		drop();
		Binding& b = binding();
		if (!b.graph || !b.seeder) throw std::logic_error("gcshim::bind was not called");
		if (!stream && gc_stream_create(&stream) != GC_OK) throw std::runtime_error(gc_last_error());
		const uint64_t off[2] = { 0, sequence.size() };
		if (gc_reads_upload(sequence.data(), off, 1, &reads) != GC_OK) throw std::runtime_error(gc_last_error());
		if (gc_align_batch(b.graph, b.seeder, stream, reads, &b.params, &result) != GC_OK) throw std::runtime_error(gc_last_error());
		current = sequence;
		return *result;
	}
	bool failedAssertion() const { return result && result->failed_assertion[0]; }
private:
	void drop() { if (result) gc_result_free(result); if (reads) gc_reads_destroy(reads); result = nullptr; reads = nullptr; }
	gc_stream* stream = nullptr;
	gc_reads* reads = nullptr;
	gc_result* result = nullptr;
	std::string current;
};
inline Session& session() { static thread_local Session s; return s; }
inline bool lastReadFailedAssertion() { return session().failedAssertion(); }

template <typename TraceVector>
inline void fillTrace(TraceVector& out, const std::string& sequence, const int32_t* node, const uint32_t* offset, const uint32_t* seqPos, const uint8_t* nodeSwitch, uint64_t n, size_t seqBase)
{
	out.reserve(n);
	std::vector<char> letters(n);   // TraceItem's graphCharacter (src/GraphAlignerCommon.h:148-153), so that the reference's own AddAlignment / AddGAFLine / AddCorrected can read a shim trace
	if (n && gc_graph_letters(binding().graph, node, offset, n, letters.data()) != GC_OK) throw std::runtime_error(gc_last_error());
	for (uint64_t i = 0; i < n; i++) {
		typename TraceVector::value_type item;
		item.DPposition.node = (size_t)node[i];
		item.DPposition.nodeOffset = offset[i];
		item.DPposition.seqPos = seqPos[i];
		item.nodeSwitch = nodeSwitch[i] != 0;
		const size_t at = seqBase + seqPos[i];
		item.sequenceCharacter = at < sequence.size() ? sequence[at] : '-';
		item.graphCharacter = letters[i];
		out.push_back(item);
	}
}

// a OnewayTrace as the four arrays the C ABI's per-trace encoders take
struct TraceArrays {
	std::vector<int32_t> node; std::vector<uint32_t> offset, seqPos; std::vector<uint8_t> nodeSwitch;
	template <typename TraceVector> explicit TraceArrays(const TraceVector& t)
	{
		node.reserve(t.size()); offset.reserve(t.size()); seqPos.reserve(t.size()); nodeSwitch.reserve(t.size());
		for (const auto& item : t) { node.push_back((int32_t)item.DPposition.node); offset.push_back((uint32_t)item.DPposition.nodeOffset); seqPos.push_back((uint32_t)item.DPposition.seqPos); nodeSwitch.push_back(item.nodeSwitch ? 1 : 0); }
	}
};

// AddGAFLine, src/GraphAlignerWrapper.h:44 (GraphAligner::AddGAFLine, src/GraphAligner.h:214-218): alignment.GAFline = the line of the alignment's trace
inline void AddGAFLine(const AlignmentGraph&, const std::string& seq_id, const std::string& sequence, AlignmentResult::AlignmentItem& alignment, bool cigarMatchMismatchMerge)
{
	if (!alignment.trace || alignment.trace->trace.empty()) throw std::logic_error("gcshim::AddGAFLine: the alignment has no trace");
	const TraceArrays t(alignment.trace->trace);
	char* text = nullptr; uint64_t len = 0;
	if (gc_format_gaf_trace(binding().graph, seq_id.c_str(), sequence.data(), sequence.size(), t.node.data(), t.offset.data(), t.seqPos.data(), t.nodeSwitch.data(), t.node.size(), cigarMatchMismatchMerge ? 1 : 0, &text, &len) != GC_OK)
		throw std::runtime_error(gc_last_error());
	alignment.GAFline.assign(text, len);
	gc_free(text);
}

// AddAlignment, src/GraphAlignerWrapper.h:43 (GraphAligner::AddAlignment, src/GraphAligner.h:205-212): alignment.alignment = the vg::Alignment of the trace, parsed from the message
// bytes the library builds (the reference's vg::Alignment is a protobuf message). As in the reference the positions hold DIGRAPH node ids and no names: the caller's own
// replaceDigraphNodeIdsWithOriginalNodeIds (src/Aligner.cpp:152-165), which src/Aligner.cpp:1009 calls right after AddAlignment, turns them into segment indices and names.
// (r4 returned the ids already replaced - an unchanged Aligner.cpp then halved them a second time: ADVICE r4.)
inline void AddAlignment(const std::string& seq_id, const std::string& sequence, AlignmentResult::AlignmentItem& alignment)
{
	if (!alignment.trace || alignment.trace->trace.empty()) throw std::logic_error("gcshim::AddAlignment: the alignment has no trace");
	const TraceArrays t(alignment.trace->trace);
	char* bytes = nullptr; uint64_t len = 0;
	if (gc_format_vg_trace_digraph(binding().graph, seq_id.c_str(), sequence.data(), sequence.size(), t.node.data(), t.offset.data(), t.seqPos.data(), t.nodeSwitch.data(), t.node.size(), (int32_t)alignment.trace->score,
			alignment.alignmentStart, alignment.alignmentEnd, &bytes, &len) != GC_OK)
		throw std::runtime_error(gc_last_error());
	alignment.alignment = std::make_shared<typename decltype(alignment.alignment)::element_type>();
	const bool parsed = alignment.alignment->ParseFromString(std::string(bytes, len));
	gc_free(bytes);
	if (!parsed) throw std::runtime_error("gcshim::AddAlignment: the vg::Alignment bytes did not parse");
}

// AddCorrected, src/GraphAlignerWrapper.h:45 (src/GraphAligner.h:220-231): the graph letters along the trace, one per graph position
inline void AddCorrected(AlignmentResult::AlignmentItem& alignment)
{
	if (!alignment.trace || alignment.trace->trace.empty()) throw std::logic_error("gcshim::AddCorrected: the alignment has no trace");
	const auto& t = alignment.trace->trace;
	alignment.corrected.assign(1, t[0].graphCharacter);
	for (size_t i = 1; i < t.size(); i++) {
		if (!t[i - 1].nodeSwitch && t[i].DPposition.nodeOffset == t[i - 1].DPposition.nodeOffset && t[i].DPposition.node == t[i - 1].DPposition.node) continue;
		alignment.corrected += t[i].graphCharacter;
	}
}

// MinimizerSeeder::getSeeds(sequence, density), src/MinimizerSeeder.h:33. The body a maintainer puts into src/MinimizerSeeder.cpp:522 is
// `return gcshim::getSeeds(sequence, density);`.
inline std::vector<SeedHit> getSeeds(const std::string& sequence, double /*density: fixed at bind() time*/)
{
	const gc_result& r = session().of(sequence);
	const Binding& b = binding();
	std::vector<SeedHit> seeds;
	for (uint64_t i = r.read_seed_off[0]; i < r.read_seed_off[1]; i++) {
		const uint32_t node = r.seed_node[i];
		const int bigraph = (int)b.nodeIDs[node];
		SeedHit s(bigraph / 2, (size_t)b.nodeOffset[node] + r.seed_offset[i], r.seed_seqpos[i], b.minimizerLength, 0, (bigraph & 1) != 0);
		s.alignmentGraphNodeId = node;
		s.alignmentGraphNodeOffset = r.seed_offset[i];
		s.seedGoodness = r.seed_goodness[i];
		s.seedClusterSize = 1;   // not part of the batch result; the replayed AlignOneWay does not read it
		seeds.push_back(s);
	}
	return seeds;
}

// OrderSeeds, src/GraphAlignerWrapper.h:46: the replayed seed list is already ordered (and sorted by seqPos as src/Aligner.cpp:667 does next).
inline void OrderSeeds(const AlignmentGraph&, std::vector<SeedHit>&) {}

// AlignOneWay, src/GraphAlignerWrapper.h:41. l < 0: the whole read (sloppy pass, src/Aligner.cpp:565); otherwise the fragment that
// starts at `offset` (src/Aligner.cpp:691): `sequence` is then the fragment, and the read it belongs to is the one getSeeds saw last.
inline AlignmentResult AlignOneWay(const AlignmentGraph&, const std::string& /*seq_id*/, const std::string& sequence, size_t initialBandwidth, size_t rampBandwidth, size_t maxCellsPerSlice, bool /*quietMode*/,
	bool /*sloppyOptimizations: implied by l < 0, as at src/Aligner.cpp:565,684*/, const std::vector<SeedHit>&, Common::AlignerGraphsizedState&, bool /*lowMemory*/, bool forceGlobal, bool preciseClipping,
	size_t /*minClusterSize, seedExtendDensity: fixed at bind() time*/, double, bool /*nondeterministicOptimizations*/, double /*preciseClippingIdentityCutoff*/, int Xdropcutoff,
	long long l = -1, long long /*r*/ = -1, long long offset = 0, const std::string* wholeRead = nullptr)
{
	// Options of this signature the kernels do not implement are refused, not ignored: a caller that asks for them would silently get the default behaviour otherwise.
	// (--ramp-bandwidth: src/GraphAlignerBitvectorBanded.h:544,608-644; --precise-clipping / X-drop: :61-68,703-; forceGlobal: :587; a finite maxCellsPerSlice: :405,581.)
	if (rampBandwidth > initialBandwidth) throw std::invalid_argument("gcshim::AlignOneWay: --ramp-bandwidth is not built (DESIGN.md section 9)");
	if (preciseClipping || Xdropcutoff > 0) throw std::invalid_argument("gcshim::AlignOneWay: --precise-clipping / X-drop are not built (DESIGN.md section 9)");
	if (forceGlobal) throw std::invalid_argument("gcshim::AlignOneWay: forced global alignment is not built (DESIGN.md section 9)");
	if (maxCellsPerSlice != std::numeric_limits<size_t>::max()) throw std::invalid_argument("gcshim::AlignOneWay: a cell limit per slice (--tangle-effort) is not built (DESIGN.md section 9)");
	if ((int64_t)initialBandwidth != (int64_t)binding().params.bandwidth) throw std::invalid_argument("gcshim::AlignOneWay: the bandwidth differs from the one given to gcshim::bind()");
	AlignmentResult out;
	if (l < 0) {
		const gc_result& r = session().of(sequence);
		for (uint64_t a = r.read_longall_off[0]; a < r.read_longall_off[1]; a++) {
			Common::OnewayTrace trace;
			const uint64_t t0 = r.long_trace_off[a], t1 = r.long_trace_off[a + 1];
			fillTrace(trace.trace, sequence, r.long_trace_node + t0, r.long_trace_offset + t0, r.long_trace_seqpos + t0, r.long_trace_switch + t0, t1 - t0, 0);
			trace.score = (int32_t)r.longall_score[a];
			AlignmentResult::AlignmentItem item(std::move(trace), 0, 0);
			item.alignmentScore = r.longall_score[a];
			item.alignmentStart = r.longall_start[a];
			item.alignmentEnd = r.longall_end[a];
			out.alignments.push_back(item);
		}
		out.seedsExtended = r.seeds_extended_long[0];
		return out;
	}
	if (!wholeRead) throw std::logic_error("gcshim::AlignOneWay(fragment): pass the read the fragment was cut from (the shim replays the read getSeeds saw)");
	const gc_result& r = session().of(*wholeRead);
	for (uint64_t a = r.read_anchor_off[0]; a < r.read_anchor_off[1]; a++) {
		if ((long long)r.anchor_x[a] != offset) continue;
		Common::OnewayTrace trace;
		const uint64_t t0 = r.anchor_trace_off[a], t1 = r.anchor_trace_off[a + 1];
		fillTrace(trace.trace, *wholeRead, r.anchor_trace_node + t0, r.anchor_trace_offset + t0, r.anchor_trace_seqpos + t0, r.anchor_trace_switch + t0, t1 - t0, (size_t)offset);
		trace.score = r.anchor_score[a];
		AlignmentResult::AlignmentItem item(std::move(trace), 0, 0);
		item.alignmentScore = (size_t)r.anchor_score[a];
		item.alignmentStart = t1 > t0 ? r.anchor_trace_seqpos[t0] : 0;
		item.alignmentEnd = t1 > t0 ? r.anchor_trace_seqpos[t1 - 1] + 1 : 0;
		out.alignments.push_back(item);
		out.seedsExtended++;
	}
	return out;
}

// AlignmentGraph::colinearChaining(anchors, sep_limit), src/AlignmentGraph.h:121: the chain over the anchors the replayed fragment calls
// returned, in their order (src/Aligner.cpp:706-729 appends them fragment by fragment, which is the batch result's order).
inline std::vector<size_t> colinearChaining(const std::string& wholeRead, const std::vector<AlignmentGraph::Anchor>& anchors, long long /*sep_limit: unused by the reference's DP as well*/)
{
	const gc_result& r = session().of(wholeRead);
	if (anchors.size() != r.read_anchor_off[1] - r.read_anchor_off[0]) throw std::logic_error("gcshim::colinearChaining: the anchors are not the ones the replayed AlignOneWay calls returned");
	std::vector<size_t> chain;
	for (uint64_t c = r.read_chain_off[0]; c < r.read_chain_off[1]; c++) chain.push_back(r.chain[c]);
	return chain;
}

} // namespace gcshim

#ifdef GC_SHIM_DEFINE_GLOBALS
// The reference's free functions (src/GraphAlignerWrapper.h:41,46), for a link without GraphAlignerWrapper.cpp. The fragment call needs the
// whole read, which the reference's signature does not carry: gcshim::currentRead() is set by the getSeeds forward in MinimizerSeeder.cpp.
namespace gcshim { inline std::string& currentRead() { static thread_local std::string s; return s; } }
AlignmentResult AlignOneWay(const AlignmentGraph& graph, const std::string& seq_id, const std::string& sequence, size_t initialBandwidth, size_t rampBandwidth, size_t maxCellsPerSlice, bool quietMode, bool sloppyOptimizations, const std::vector<SeedHit>& seedHits, GraphAlignerCommon<size_t, int32_t, uint64_t>::AlignerGraphsizedState& reusableState, bool lowMemory, bool forceGlobal, bool preciseClipping, size_t minClusterSize, double seedExtendDensity, bool nondeterministicOptimizations, double preciseClippingIdentityCutoff, int Xdropcutoff, long long l, long long r, long long offset)
{
	return gcshim::AlignOneWay(graph, seq_id, sequence, initialBandwidth, rampBandwidth, maxCellsPerSlice, quietMode, sloppyOptimizations, seedHits, reusableState, lowMemory, forceGlobal, preciseClipping, minClusterSize, seedExtendDensity, nondeterministicOptimizations, preciseClippingIdentityCutoff, Xdropcutoff, l, r, offset, l < 0 ? nullptr : &gcshim::currentRead());
}
void OrderSeeds(const AlignmentGraph& graph, std::vector<SeedHit>& seedHits) { gcshim::OrderSeeds(graph, seedHits); }
void AddAlignment(const std::string& seq_id, const std::string& sequence, AlignmentResult::AlignmentItem& alignment) { gcshim::AddAlignment(seq_id, sequence, alignment); }
void AddGAFLine(const AlignmentGraph& graph, const std::string& seq_id, const std::string& sequence, AlignmentResult::AlignmentItem& alignment, bool cigarMatchMismatchMerge) { gcshim::AddGAFLine(graph, seq_id, sequence, alignment, cigarMatchMismatchMerge); }
void AddCorrected(AlignmentResult::AlignmentItem& alignment) { gcshim::AddCorrected(alignment); }
#endif

#endif
