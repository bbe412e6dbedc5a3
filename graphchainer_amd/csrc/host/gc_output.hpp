// Output encoders of the hot path's results, host side. GAF first: one line per alignment.
#pragma once
#include "gc_graph.hpp"
#include <cstdint>
#include <string>
#include <vector>

namespace gc {

// One alignment trace in the reference's output coordinates (what GraphAligner's OnewayTrace holds after
// fixForwardTraceSeqPos / fixReverseTraceSeqPosAndOrder, src/GraphAligner.h:527-565): bigraph node id, offset in the
// original node, read position, and "the next cell is in another node" flag.
struct TraceView {
	const int32_t* node;
	const uint32_t* offset;
	const uint32_t* seqPos;
	const uint8_t* nodeSwitch;
	uint64_t size;
};

// The graph letter under a trace cell (bigraph node id, offset in the original node): GetUnitigNode + NodeSequences as the reference's TraceItem constructor does
// (src/GraphAlignerCommon.h:148-153), but remembering the split node of the last cell - consecutive cells of a trace sit in the same 64-letter split node or the next, and
// GetUnitigNode costs two hash look-ups and a division per call (r3: 105 M cells per 10 k reads went through it, most of gc_format_gaf's 9 CPU-seconds).
class GraphLetters {
public:
	explicit GraphLetters(const AlignmentGraph& graph) : graph(graph) {}
	char at(int nodeId, size_t offset)
	{
		if (nodeId != currentId || !known) {
			nodes = graph.nodeLookup.at(nodeId);
			known = true;
			currentId = nodeId;
			index = (size_t)(nodes.size() * ((double)offset / (double)graph.originalNodeSize.at(nodeId)));
			if (index >= nodes.size()) index = nodes.size() - 1;
		} else if (offset >= lo && offset < hi) {
			return graph.NodeSequences(split, offset - lo);
		}
		// the split nodes of one original node partition its letters: the one that holds `offset` is unique, whichever index the search starts from
		while (index < nodes.size() - 1 && graph.nodeOffset[nodes[index]] + graph.nodeLength[nodes[index]] <= offset) index++;
		while (index > 0 && graph.nodeOffset[nodes[index]] > offset) index--;
		split = nodes[index];
		lo = graph.nodeOffset[split];
		hi = lo + graph.nodeLength[split];
		return graph.NodeSequences(split, offset - lo);
	}
private:
	const AlignmentGraph& graph;
	NodeLookup::Span nodes { nullptr, nullptr };
	bool known = false;
	int currentId = 0;
	size_t index = 0, split = 0, lo = 0, hi = 0;
};

// IUPAC-aware base comparison, reference: GraphAlignerCommon::characterMatch, src/GraphAlignerCommon.h:190-297.
bool characterMatch(char sequenceCharacter, char graphCharacter);

// GAF line of one alignment (no trailing newline). reference: GraphAlignerGAFAlignment::traceToAlignment,
// src/GraphAlignerGAFAlignment.h:38-196 (called through AddGAFLine, src/GraphAlignerWrapper.cpp:38-43).
std::string formatGafLine(const AlignmentGraph& graph, const std::string& readName, const char* sequence, uint64_t readLength, const TraceView& trace, bool cigarMatchMismatchMerge);

// vg::Alignment as the reference fills it (GraphAlignerVGAlignment::traceToAlignment, src/GraphAlignerVGAlignment.h:36-163,
// AddAlignment src/GraphAligner.h:205-212, replaceDigraphNodeIdsWithOriginalNodeIds src/Aligner.cpp:152-165); only the
// fields the reference sets. Messages and field numbers: src/vg.proto:52-56 (Edit), :62-66 (Mapping), :89-94 (Position),
// :104-109 (Path), :113-154 (Alignment).
struct VgEdit { int32_t fromLength = 0, toLength = 0; std::string sequence; };
struct VgMapping { int64_t nodeId = 0, offset = 0; bool isReverse = false; std::string name; std::vector<VgEdit> edits; int64_t rank = 0; };
struct VgAlignment { std::string sequence, name; int32_t score = 0, queryPosition = 0; double identity = 0; std::vector<VgMapping> mappings; };

VgAlignment buildVgAlignment(const AlignmentGraph& graph, const std::string& readName, const char* sequence, uint64_t readLength, const TraceView& trace,
	int32_t score, uint64_t alignmentStart, uint64_t alignmentEnd);

// google::protobuf::util::MessageToJsonString with preserve_proto_field_names (src/Aligner.cpp:283-298): one compact JSON object.
std::string vgToJson(const VgAlignment& aln);
// proto3 wire format of the message (what Alignment::SerializeToString writes, src/Aligner.cpp:273).
std::string vgToProtobuf(const VgAlignment& aln);
// An alignment whose trace walk ran on the device (hip/gc_output.hip): the GAF path and CIGAR columns as text, or its vg::Path in wire format, and the counts
// the remaining columns / fields are made of. appendGafLine writes the line formatGafLine would (no newline); vgProtobufFromEncoded the message
// vgToProtobuf(buildVgAlignment(...)) would; vgFromEncoded decodes the path bytes back into the struct (for the JSON printer).
struct EncodedAlignment {
	const char* path = nullptr; uint64_t pathLen = 0;
	const char* cigar = nullptr; uint64_t cigarLen = 0;
	const uint8_t* vgPath = nullptr; uint64_t vgPathLen = 0;
	uint64_t nodePathLen = 0, nodePathStart = 0, nodePathEnd = 0, matches = 0, mismatches = 0, insertions = 0, deletions = 0, cells = 0, alignmentStart = 0, alignmentEnd = 0;
	int32_t score = 0;
};
void appendGafLine(std::string& out, const std::string& readName, uint64_t readLength, const EncodedAlignment& a);
std::string vgProtobufFromEncoded(const std::string& readName, const char* sequence, const EncodedAlignment& a);
VgAlignment vgFromEncoded(const std::string& readName, const char* sequence, const EncodedAlignment& a);

// One GAM group as writeGAMToQueue frames it (src/Aligner.cpp:261-281): varint64 message count, then per message varint32
// size + bytes, the whole group one gzip member.
std::string gamGroupRaw(const std::vector<std::string>& messages);   // the group before compression: varint count, then varint size + message each
std::string gzipMember(const uint8_t* deflated, size_t deflatedBytes, const std::string& raw);   // gzip framing of a deflate stream of `raw` made elsewhere
std::string gamGroup(const std::vector<std::string>& messages, int level = -1);   // level: zlib's (-1 = Z_DEFAULT_COMPRESSION, what GzipOutputStream uses)

} // namespace gc
