// Compiles include/graphchainer_amd_shim.hpp against minimal definitions of the reference's types (same names and members as
// src/GraphAlignerWrapper.h:11-37, src/GraphAlignerCommon.h:127-183,299-355, src/AlignmentGraph.h:112-115) and drives the
// reference's per-read call sequence (src/Aligner.cpp:538-735) through it for the reads given on the command line:
//   shim_test graph.gfa READ [READ ...]
// Prints, per read: seeds, whole-read alignments, anchors, chain length, chain (anchor indices). Without a GPU the library refuses
// to create the graph (no CPU fallback): prints NO_DEVICE and exits 0.
#include <cstdint>
#include <cstdio>
#include <limits>
#include <memory>
#include <string>
#include <vector>

template <typename LengthType, typename ScoreType, typename Word>
struct GraphAlignerCommon {
	struct MatrixPosition { size_t node = 0, nodeOffset = 0, seqPos = 0; };
	struct TraceItem { MatrixPosition DPposition; bool nodeSwitch = false; char sequenceCharacter = '-', graphCharacter = '-'; };
	struct OnewayTrace { std::vector<TraceItem> trace; ScoreType score = 0; };
	struct AlignerGraphsizedState {};
};
namespace vg { struct Alignment { std::string bytes; bool ParseFromString(const std::string& s) { bytes = s; return true; } }; }
struct SeedHit {
	SeedHit(int nodeID, size_t nodeOffset, size_t seqPos, size_t matchLen, size_t rawSeedGoodness, bool reverse) : nodeID(nodeID), nodeOffset(nodeOffset), seqPos(seqPos), matchLen(matchLen), reverse(reverse),
		alignmentGraphNodeId(std::numeric_limits<size_t>::max()), alignmentGraphNodeOffset(std::numeric_limits<size_t>::max()), rawSeedGoodness(rawSeedGoodness), seedGoodness(0), seedClusterSize(0) {}
	int nodeID; size_t nodeOffset, seqPos, matchLen; bool reverse; size_t alignmentGraphNodeId, alignmentGraphNodeOffset, rawSeedGoodness, seedGoodness, seedClusterSize;
};
struct AlignmentResult {
	struct AlignmentItem {
		AlignmentItem() {}
		AlignmentItem(GraphAlignerCommon<size_t, int32_t, uint64_t>::OnewayTrace&& t, size_t cells, size_t ms) : cellsProcessed(cells), elapsedMilliseconds(ms)
		{ trace = std::make_shared<GraphAlignerCommon<size_t, int32_t, uint64_t>::OnewayTrace>(); *trace = std::move(t); }
		bool alignmentFailed() const { return alignmentEnd == alignmentStart; }
		std::shared_ptr<GraphAlignerCommon<size_t, int32_t, uint64_t>::OnewayTrace> trace;
		std::shared_ptr<vg::Alignment> alignment;   // (the reference's is the protobuf message; here: a holder of its bytes)
		std::string GAFline, corrected;
		size_t seedGoodness = 0, cellsProcessed = 0, elapsedMilliseconds = 0, alignmentStart = 0, alignmentEnd = 0, alignmentScore = std::numeric_limits<size_t>::max();
	};
	std::vector<AlignmentItem> alignments;
	size_t seedsExtended = 0;
};
struct AlignmentGraph { struct Anchor { std::vector<size_t> path; size_t x, y; }; };

#define GC_SHIM_DEFINE_GLOBALS
#include "graphchainer_amd_shim.hpp"

int main(int argc, char** argv)
{
	if (argc < 3) { fprintf(stderr, "usage: shim_test graph.gfa READ...\n"); return 2; }
	gc_graph* graph = nullptr;
	gc_seeder* seeder = nullptr;
	int rc = gc_graph_create_from_gfa(argv[1], &graph);
	if (rc == GC_ERR_DEVICE) { printf("NO_DEVICE\n"); return 0; }
	if (rc != GC_OK || gc_seeder_create(graph, 15, 20, 1.0 - 0.001, &seeder) != GC_OK) { fprintf(stderr, "%s\n", gc_last_error()); return 1; }
	gc_params gp;
	gc_params_default(&gp);
	gcshim::bind(graph, seeder, gp);
	AlignmentGraph alignmentGraph;
	GraphAlignerCommon<size_t, int32_t, uint64_t>::AlignerGraphsizedState reusableState;
	for (int a = 2; a < argc; a++) {
		const std::string sequence = argv[a];
		// ---- the reference's per-read sequence, src/Aligner.cpp:538-735, with the shim's calls
		std::vector<SeedHit> seeds = gcshim::getSeeds(sequence, 10);                                   // :538 / :660
		gcshim::currentRead() = sequence;
		OrderSeeds(alignmentGraph, seeds);                                                             // :560 / :666
		AlignmentResult longAlignments = AlignOneWay(alignmentGraph, "r", sequence, 10, 0, std::numeric_limits<size_t>::max(), true, true, seeds, reusableState, true, false, false, 1, -1, false, 0.5, 0, -1, -1, 0);   // :565
		if (a == 2) {
			// options of the signature that are not built are refused, not ignored: --ramp-bandwidth, --precise-clipping, a forced global alignment, a cell limit per slice
			int refused = 0;
			const size_t unlimited = std::numeric_limits<size_t>::max();
			try { AlignOneWay(alignmentGraph, "r", sequence, 10, 20, unlimited, true, true, seeds, reusableState, true, false, false, 1, -1, false, 0.5, 0, -1, -1, 0); } catch (const std::invalid_argument&) { refused++; }
			try { AlignOneWay(alignmentGraph, "r", sequence, 10, 0, unlimited, true, true, seeds, reusableState, true, false, true, 1, -1, false, 0.5, 0, -1, -1, 0); } catch (const std::invalid_argument&) { refused++; }
			try { AlignOneWay(alignmentGraph, "r", sequence, 10, 0, unlimited, true, true, seeds, reusableState, true, true, false, 1, -1, false, 0.5, 0, -1, -1, 0); } catch (const std::invalid_argument&) { refused++; }
			try { AlignOneWay(alignmentGraph, "r", sequence, 10, 0, 5000, true, true, seeds, reusableState, true, false, false, 1, -1, false, 0.5, 0, -1, -1, 0); } catch (const std::invalid_argument&) { refused++; }
			try { AlignOneWay(alignmentGraph, "r", sequence, 35, 0, unlimited, true, true, seeds, reusableState, true, false, false, 1, -1, false, 0.5, 0, -1, -1, 0); } catch (const std::invalid_argument&) { refused++; }
			if (refused != 5) { fprintf(stderr, "the shim accepted %d of 5 calls with options that are not built\n", 5 - refused); return 1; }
		}
		std::vector<AlignmentGraph::Anchor> A;
		const size_t len = 35, sep = 35;
		size_t sl = 0, sr = 0;
		for (size_t l = 0; l + len <= sequence.size(); l += sep) {                                     // :672-730
			while (sr < seeds.size() && seeds[sr].seqPos + seeds[sr].matchLen <= l + len) sr++;
			while (sl < sr && seeds[sl].seqPos < l) sl++;
			if (sl >= sr) continue;
			AlignmentResult alignments = AlignOneWay(alignmentGraph, "f", sequence.substr(l, len), 10, 0, std::numeric_limits<size_t>::max(), true, false, seeds, reusableState, true, false, false, 1, -1, false, 0.5, 0, (long long)sl, (long long)sr, (long long)l);
			for (auto& alignment : alignments.alignments) {
				if (alignment.alignmentFailed() || alignment.trace->trace.empty()) continue;
				AlignmentGraph::Anchor anchor { {}, l, l + len - 1 };
				for (auto& t : alignment.trace->trace) if (anchor.path.empty() || t.DPposition.node != anchor.path.back()) anchor.path.push_back(t.DPposition.node);   // (bigraph ids here; the reference maps to split nodes with GetUnitigNode)
				A.push_back(anchor);
			}
		}
		// output of the whole-read alignments through the reference's calls (src/Aligner.cpp:1006-1019): AddGAFLine / AddAlignment on the shim's items
		for (size_t i = 0; i < longAlignments.alignments.size(); i++) {
			auto& item = longAlignments.alignments[i];
			AddGAFLine(alignmentGraph, "r" + std::to_string(a - 2), sequence, item, false);
			AddAlignment("r" + std::to_string(a - 2), sequence, item);
			AddCorrected(item);
			unsigned long long h = 1469598103934665603ull;
			for (unsigned char c : item.alignment->bytes) { h ^= c; h *= 1099511628211ull; }
			printf("gaf %d %zu %zu %zu %016llx %zu\t%s\n", a - 2, i, item.alignmentStart, item.alignmentEnd, h, item.corrected.size(), item.GAFline.c_str());
			// the vg::Alignment bytes AddAlignment left (digraph node ids, no names: the caller's replaceDigraphNodeIdsWithOriginalNodeIds comes next, src/Aligner.cpp:1009);
			// the GPU test decodes them, applies that step as the reference states it (:152-165) and compares with the batch's GAM
			printf("vg %d %zu ", a - 2, i);
			for (unsigned char c : item.alignment->bytes) printf("%02x", c);
			printf("\n");
		}
		std::vector<size_t> ids = gcshim::colinearChaining(sequence, A, 10000);                        // :735
		printf("read %d: seeds %zu long %zu anchors %zu chain %zu :", a - 2, seeds.size(), longAlignments.alignments.size(), A.size(), ids.size());
		for (size_t i : ids) printf(" %zu", i);
		printf("\n");
	}
	gc_seeder_destroy(seeder);
	gc_graph_destroy(graph);
	return 0;
}
