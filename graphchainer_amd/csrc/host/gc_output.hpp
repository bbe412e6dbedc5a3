// Output encoders of the hot path's results, host side. GAF first: one line per alignment.
#pragma once
#include "gc_graph.hpp"
#include <cstdint>
#include <string>

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

// IUPAC-aware base comparison, reference: GraphAlignerCommon::characterMatch, src/GraphAlignerCommon.h:190-297.
bool characterMatch(char sequenceCharacter, char graphCharacter);

// GAF line of one alignment (no trailing newline). reference: GraphAlignerGAFAlignment::traceToAlignment,
// src/GraphAlignerGAFAlignment.h:38-196 (called through AddGAFLine, src/GraphAlignerWrapper.cpp:38-43).
std::string formatGafLine(const AlignmentGraph& graph, const std::string& readName, const char* sequence, uint64_t readLength, const TraceView& trace, bool cigarMatchMismatchMerge);

} // namespace gc
