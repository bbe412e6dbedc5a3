// TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's GAF writer.
// reference: GraphAlignerGAFAlignment::traceToAlignment, src/GraphAlignerGAFAlignment.h:38-252, called through
// AddGAFLine (src/GraphAlignerWrapper.cpp:38-43, src/Aligner.cpp:1015-1019). Parity unpinned: the reference binary cannot
// be built here, so no reference-produced GAF exists to check this against.
#pragma once
#include "bitvector_aligner.hpp"
#include <sstream>
#include <string>

namespace oracle {

struct GafMergedNodePos { int nodeId; bool reverse; size_t nodeOffset; size_t seqPos; };   // :19-25
enum GafEditType { GafMatch, GafMismatch, GafMatchOrMismatch, GafInsertion, GafDeletion, GafEmpty };   // :26-34

inline void gafAddPosToString(std::stringstream& str, GafMergedNodePos pos, const AlignmentGraph& graph)   // :200-218
{
	str << (pos.reverse ? "<" : ">");
	std::string nodeName = graph.OriginalNodeName(pos.nodeId);
	if (nodeName == "") str << pos.nodeId / 2; else str << nodeName;
}

inline void gafAddCigarItem(std::stringstream& str, size_t editLength, GafEditType type)   // :220-248
{
	if (editLength == 0) return;
	str << editLength;
	switch (type) {
		case GafMatchOrMismatch: str << "M"; break;
		case GafMatch: str << "="; break;
		case GafMismatch: str << "X"; break;
		case GafInsertion: str << "I"; break;
		case GafDeletion: str << "D"; break;
		case GafEmpty: default: return;
	}
}

inline std::string traceToGaf(const AlignmentGraph& graph, const std::string& seq_id, const std::string& sequence, const OnewayTrace& tracePair, bool cigarMatchMismatchMerge)
{
	const auto& trace = tracePair.trace;
	if (trace.size() == 0) return std::string();
	std::stringstream cigar;
	size_t readLen = sequence.size();
	size_t readStart = trace[0].DPposition.seqPos;
	size_t readEnd = trace.back().DPposition.seqPos + 1;
	bool strand = true;
	std::stringstream nodePath;
	size_t nodePathLen = 0;
	size_t nodePathStart = trace[0].DPposition.nodeOffset;
	size_t nodePathEnd = 0;
	size_t matches = 0;
	size_t blockLength = trace.size();
	int mappingQuality = 255;
	GafMergedNodePos currentPos { (int)trace[0].DPposition.node, (trace[0].DPposition.node % 2) == 1, trace[0].DPposition.nodeOffset, trace[0].DPposition.seqPos };
	GafEditType currentEdit = GafEmpty;
	size_t mismatches = 0, deletions = 0, insertions = 0, editLength = 0;
	if (cigarMatchMismatchMerge) {
		currentEdit = GafMatchOrMismatch;
		editLength = 1;
		if (characterMatch(trace[0].sequenceCharacter, trace[0].graphCharacter)) matches += 1; else mismatches += 1;
	} else if (characterMatch(trace[0].sequenceCharacter, trace[0].graphCharacter)) {
		currentEdit = GafMatch; editLength = 1; matches += 1;
	} else {
		currentEdit = GafMismatch; editLength = 1; mismatches += 1;
	}
	gafAddPosToString(nodePath, currentPos, graph);
	nodePathLen += graph.originalNodeSize.at(currentPos.nodeId);
	for (size_t pos = 1; pos < trace.size(); pos++) {
		ORACLE_ASSERT(trace[pos].DPposition.seqPos < sequence.size());
		GafMergedNodePos newPos { (int)trace[pos].DPposition.node, (trace[pos].DPposition.node % 2) == 1, trace[pos].DPposition.nodeOffset, trace[pos].DPposition.seqPos };
		bool insideNode = !trace[pos - 1].nodeSwitch || (newPos.nodeId == currentPos.nodeId && newPos.reverse == currentPos.reverse && newPos.nodeOffset > currentPos.nodeOffset);
		ORACLE_ASSERT(newPos.seqPos >= currentPos.seqPos);
		if (!insideNode) {
			size_t skippedBefore = graph.originalNodeSize.at(currentPos.nodeId) - 1 - trace[pos - 1].DPposition.nodeOffset;
			currentPos = newPos;
			gafAddPosToString(nodePath, currentPos, graph);
			ORACLE_ASSERT(trace[pos].DPposition.nodeOffset < graph.originalNodeSize.at(currentPos.nodeId));
			size_t skippedAfter = trace[pos].DPposition.nodeOffset;
			nodePathLen += graph.originalNodeSize.at(currentPos.nodeId) - (skippedBefore + skippedAfter);
		}
		auto change = [&](GafEditType t) {
			if (currentEdit == GafEmpty) currentEdit = t;
			if (currentEdit != t) { gafAddCigarItem(cigar, editLength, currentEdit); currentEdit = t; editLength = 0; }
			editLength += 1;
		};
		if (trace[pos - 1].DPposition.seqPos == trace[pos].DPposition.seqPos) { change(GafDeletion); deletions += 1; }
		else if (insideNode && trace[pos - 1].DPposition.nodeOffset == trace[pos].DPposition.nodeOffset) { change(GafInsertion); insertions += 1; }
		else if (cigarMatchMismatchMerge) {
			change(GafMatchOrMismatch);
			if (characterMatch(trace[pos].sequenceCharacter, trace[pos].graphCharacter)) matches += 1; else mismatches += 1;
		}
		else if (characterMatch(trace[pos].sequenceCharacter, trace[pos].graphCharacter)) { change(GafMatch); matches += 1; }
		else { change(GafMismatch); mismatches += 1; }
	}
	ORACLE_ASSERT(matches + mismatches + deletions + insertions == trace.size());
	gafAddCigarItem(cigar, editLength, currentEdit);
	nodePathEnd = nodePathLen - (graph.originalNodeSize.at((int)trace.back().DPposition.node) - 1 - trace.back().DPposition.nodeOffset);
	std::stringstream sstr;
	sstr << seq_id << "\t" << readLen << "\t" << readStart << "\t" << readEnd << "\t" << (strand ? "+" : "-") << "\t" << nodePath.str() << "\t" << nodePathLen << "\t" << nodePathStart << "\t" << nodePathEnd << "\t" << matches << "\t" << blockLength << "\t" << mappingQuality;
	sstr << "\t" << "NM:i:" << (mismatches + deletions + insertions);
	sstr << "\t" << "dv:f:" << 1.0 - ((double)matches / (double)(matches + mismatches + deletions + insertions));
	sstr << "\t" << "id:f:" << ((double)matches / (double)(matches + mismatches + deletions + insertions));
	sstr << "\t" << "cg:Z:" << cigar.str();
	return sstr.str();
}

} // namespace oracle
