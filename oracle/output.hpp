// TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's GAF writer.
// reference: GraphAlignerGAFAlignment::traceToAlignment, src/GraphAlignerGAFAlignment.h:38-252, called through
// AddGAFLine (src/GraphAlignerWrapper.cpp:38-43, src/Aligner.cpp:1015-1019). Parity unpinned: the reference binary cannot
// be built here, so no reference-produced GAF exists to check this against.
#pragma once
#include "bitvector_aligner.hpp"
#include "pipeline.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

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

// ---- vg::Alignment -> JSON. reference: GraphAlignerVGAlignment::traceToAlignment (src/GraphAlignerVGAlignment.h:36-163),
// AddAlignment (src/GraphAligner.h:205-212), replaceDigraphNodeIdsWithOriginalNodeIds (src/Aligner.cpp:152-165),
// writeJSONToQueue (src/Aligner.cpp:283-298: MessageToJsonString, preserve_proto_field_names). The JSON text follows
// protobuf 3's json_util: fields in field-number order (src/vg.proto:52-154), proto3 defaults omitted, int64 as strings,
// doubles with the shortest of %.15g / %.17g that round-trips. Parity unpinned (no protobuf C++ runtime on this box); the
// tests cross-check the same alignments against the protobuf Python runtime through the GAM bytes.
struct OraEdit { int from_length = 0, to_length = 0; std::string sequence; };
struct OraMapping { long long node_id = 0, offset = 0; bool is_reverse = false; std::string name; std::vector<OraEdit> edit; long long rank = 0; };

inline std::string oraJsonEscape(const std::string& in)
{
	std::string out = "\"";
	char tmp[8];
	for (unsigned char c : in) {
		if (c == '"') out += "\\\"";
		else if (c == '\\') out += "\\\\";
		else if (c == '\n') out += "\\n";
		else if (c == '\r') out += "\\r";
		else if (c == '\t') out += "\\t";
		else if (c == '\b') out += "\\b";
		else if (c == '\f') out += "\\f";
		else if (c == '<' || c == '>' || c < 0x20 || c == 0x7f) { snprintf(tmp, sizeof tmp, "\\u%04x", c); out += tmp; }
		else out += (char)c;
	}
	return out + "\"";
}

// the vg::Alignment the reference holds for one final alignment after AddAlignment + replaceDigraphNodeIdsWithOriginalNodeIds: exactly the fields it sets
struct OraAlignment { std::string sequence; std::vector<OraMapping> mapping; std::string name; int score = 0; int query_position = 0; double identity = 0; };

inline OraAlignment buildVgAlignment(const AlignmentGraph& graph, const std::string& seq_id, const std::string& sequence, const AlignmentItem& item)
{
	const auto& trace = item.trace->trace;
	OraAlignment aln;
	std::vector<OraMapping>& mapping = aln.mapping;
	enum { Match, Mismatch, Insertion, Deletion, Empty } currentEdit = Empty;
	size_t mismatches = 0, deletions = 0, insertions = 0, matches = 0;
	int curNode = (int)trace[0].DPposition.node;
	size_t curOffset = trace[0].DPposition.nodeOffset;
	int rank = 0;
	mapping.push_back(OraMapping { curNode, (long long)curOffset, (curNode % 2) == 1, "", { OraEdit() }, rank });
	OraEdit* edit = &mapping.back().edit.back();
	if (characterMatch(trace[0].sequenceCharacter, trace[0].graphCharacter)) {
		currentEdit = Match; edit->from_length++; edit->to_length++; matches++;
	} else {
		currentEdit = Mismatch; edit->from_length++; edit->to_length++; edit->sequence = std::string { sequence[0] }; mismatches++;
	}
	for (size_t pos = 1; pos < trace.size(); pos++) {
		int newNode = (int)trace[pos].DPposition.node;
		size_t newOffset = trace[pos].DPposition.nodeOffset;
		bool insideNode = !trace[pos - 1].nodeSwitch || (newNode == curNode && newOffset > curOffset);
		if (!insideNode) {
			rank++;
			curNode = newNode; curOffset = newOffset;
			mapping.push_back(OraMapping { curNode, (long long)curOffset, (curNode % 2) == 1, "", { OraEdit() }, rank });
			edit = &mapping.back().edit.back();
			currentEdit = Empty;
		}
		if (trace[pos - 1].DPposition.seqPos == trace[pos].DPposition.seqPos) {
			if (currentEdit == Empty) currentEdit = Deletion;
			if (currentEdit != Deletion) { mapping.back().edit.push_back(OraEdit()); edit = &mapping.back().edit.back(); currentEdit = Deletion; }
			edit->from_length++; deletions++;
		} else if (insideNode && trace[pos - 1].DPposition.nodeOffset == trace[pos].DPposition.nodeOffset) {
			if (currentEdit == Empty) currentEdit = Insertion;
			if (currentEdit != Insertion) { mapping.back().edit.push_back(OraEdit()); edit = &mapping.back().edit.back(); currentEdit = Insertion; }
			edit->to_length++; edit->sequence += trace[pos].sequenceCharacter; insertions++;
		} else if (characterMatch(trace[pos].sequenceCharacter, trace[pos].graphCharacter)) {
			if (currentEdit == Empty) currentEdit = Match;
			if (currentEdit != Match) { mapping.back().edit.push_back(OraEdit()); edit = &mapping.back().edit.back(); currentEdit = Match; }
			edit->from_length++; edit->to_length++; matches++;
		} else {
			if (currentEdit == Empty) currentEdit = Mismatch;
			if (currentEdit != Mismatch) { mapping.back().edit.push_back(OraEdit()); edit = &mapping.back().edit.back(); currentEdit = Mismatch; }
			edit->from_length++; edit->to_length++; edit->sequence += trace[pos].sequenceCharacter; mismatches++;
		}
	}
	double identity = (double)matches / (double)(matches + mismatches + insertions + deletions);
	for (OraMapping& m : mapping) {
		int digraphNodeId = (int)m.node_id;
		m.node_id = digraphNodeId / 2;
		m.name = graph.OriginalNodeName(digraphNodeId);
	}
	aln.sequence = sequence.substr(item.alignmentStart, item.alignmentEnd - item.alignmentStart);   // src/GraphAligner.h:210
	aln.name = seq_id;                                  // src/GraphAlignerVGAlignment.h:42
	aln.score = item.trace->score;                      // :43
	aln.query_position = (int)item.alignmentStart;      // src/GraphAligner.h:211
	aln.identity = identity;                            // src/GraphAlignerVGAlignment.h:162
	return aln;
}

inline std::string vgAlignmentToJson(const OraAlignment& aln)
{
	const std::vector<OraMapping>& mapping = aln.mapping;
	const std::string& alignedSequence = aln.sequence;
	const std::string& seq_id = aln.name;
	double identity = aln.identity;
	std::string js = "{";
	auto field = [&](std::string& s, bool& first, const std::string& key) { if (!first) s += ","; first = false; s += "\"" + key + "\":"; };
	bool first = true;
	if (!alignedSequence.empty()) { field(js, first, "sequence"); js += oraJsonEscape(alignedSequence); }
	field(js, first, "path");
	js += "{";
	if (!mapping.empty()) {
		js += "\"mapping\":[";
		for (size_t i = 0; i < mapping.size(); i++) {
			const OraMapping& m = mapping[i];
			if (i) js += ",";
			js += "{\"position\":{";
			bool pf = true;
			if (m.node_id) { field(js, pf, "node_id"); js += "\"" + std::to_string(m.node_id) + "\""; }
			if (m.offset) { field(js, pf, "offset"); js += "\"" + std::to_string(m.offset) + "\""; }
			if (m.is_reverse) { field(js, pf, "is_reverse"); js += "true"; }
			if (!m.name.empty()) { field(js, pf, "name"); js += oraJsonEscape(m.name); }
			js += "}";
			if (!m.edit.empty()) {
				js += ",\"edit\":[";
				for (size_t e = 0; e < m.edit.size(); e++) {
					if (e) js += ",";
					js += "{";
					bool ef = true;
					if (m.edit[e].from_length) { field(js, ef, "from_length"); js += std::to_string(m.edit[e].from_length); }
					if (m.edit[e].to_length) { field(js, ef, "to_length"); js += std::to_string(m.edit[e].to_length); }
					if (!m.edit[e].sequence.empty()) { field(js, ef, "sequence"); js += oraJsonEscape(m.edit[e].sequence); }
					js += "}";
				}
				js += "]";
			}
			if (m.rank) js += ",\"rank\":\"" + std::to_string(m.rank) + "\"";
			js += "}";
		}
		js += "]";
	}
	js += "}";
	if (!seq_id.empty()) { field(js, first, "name"); js += oraJsonEscape(seq_id); }
	int score = aln.score;
	if (score) { field(js, first, "score"); js += std::to_string(score); }
	if (aln.query_position) { field(js, first, "query_position"); js += std::to_string(aln.query_position); }
	if (identity != 0) {
		char buf[40];
		snprintf(buf, sizeof buf, "%.15g", identity);
		if (strtod(buf, nullptr) != identity) snprintf(buf, sizeof buf, "%.17g", identity);
		field(js, first, "identity"); js += buf;
	}
	js += "}";
	return js;
}

inline std::string alignmentToJson(const AlignmentGraph& graph, const std::string& seq_id, const std::string& sequence, const AlignmentItem& item)
{
	return vgAlignmentToJson(buildVgAlignment(graph, seq_id, sequence, item));
}

// ---- vg::Alignment -> GAM. reference: writeGAMToQueue (src/Aligner.cpp:261-281): per read one gzip member holding
// varint64 count, then per alignment varint32 size + Alignment::SerializeToString. The message bytes are proto3's canonical
// serialisation of src/vg.proto:52-154: fields in field-number order, scalar fields with their default value omitted, a
// sub-message that was set (set_allocated_path, set_allocated_position, add_mapping, add_edit) written even when it is empty.
// r5: PINNED against the reference's own generated descriptor - tests/golden/make_gam_golden.py parses these bytes with
// /root/reference/scripts/vg_pb2.py through the reader of scripts/summary.py:63-75 and requires that every message
// re-serialises to itself; the decoded messages are the committed *.expected.gam.json fixtures.
inline void pbVarint(std::string& out, uint64_t v) { while (v >= 0x80) { out += (char)(v | 0x80); v >>= 7; } out += (char)v; }
inline void pbKey(std::string& out, int fieldNumber, int wireType) { pbVarint(out, ((uint64_t)fieldNumber << 3) | (uint64_t)wireType); }
inline void pbInt(std::string& out, int fieldNumber, long long v) { if (v == 0) return; pbKey(out, fieldNumber, 0); pbVarint(out, (uint64_t)v); }   // int32 / int64: negative values sign-extend to ten bytes
inline void pbBool(std::string& out, int fieldNumber, bool v) { if (!v) return; pbKey(out, fieldNumber, 0); out += (char)1; }
inline void pbString(std::string& out, int fieldNumber, const std::string& v) { if (v.empty()) return; pbKey(out, fieldNumber, 2); pbVarint(out, v.size()); out += v; }
inline void pbMessage(std::string& out, int fieldNumber, const std::string& body) { pbKey(out, fieldNumber, 2); pbVarint(out, body.size()); out += body; }
inline void pbDouble(std::string& out, int fieldNumber, double v)
{
	uint64_t bits;
	memcpy(&bits, &v, 8);
	if (bits == 0) return;
	pbKey(out, fieldNumber, 1);
	for (int i = 0; i < 8; i++) out += (char)(bits >> (8 * i));
}

inline std::string vgAlignmentToProto(const OraAlignment& aln)
{
	std::string path;                                   // Path { repeated Mapping mapping = 2 }
	for (const OraMapping& m : aln.mapping) {
		std::string position;                           // Position { node_id = 1, offset = 2, is_reverse = 4, name = 5 }
		pbInt(position, 1, m.node_id);
		pbInt(position, 2, m.offset);
		pbBool(position, 4, m.is_reverse);
		pbString(position, 5, m.name);
		std::string mapping;                            // Mapping { position = 1, repeated edit = 2, rank = 5 }
		pbMessage(mapping, 1, position);
		for (const OraEdit& e : m.edit) {
			std::string edit;                           // Edit { from_length = 1, to_length = 2, sequence = 3 }
			pbInt(edit, 1, e.from_length);
			pbInt(edit, 2, e.to_length);
			pbString(edit, 3, e.sequence);
			pbMessage(mapping, 2, edit);
		}
		pbInt(mapping, 5, m.rank);
		pbMessage(path, 2, mapping);
	}
	std::string out;                                    // Alignment { sequence = 1, path = 2, name = 3, score = 6, query_position = 7, identity = 16 }
	pbString(out, 1, aln.sequence);
	pbMessage(out, 2, path);
	pbString(out, 3, aln.name);
	pbInt(out, 6, aln.score);
	pbInt(out, 7, aln.query_position);
	pbDouble(out, 16, aln.identity);
	return out;
}

// one read's group as the reference frames it BEFORE the gzip layer (the inflated bytes of its member)
inline std::string gamGroup(const std::vector<std::string>& messages)
{
	std::string out;
	pbVarint(out, messages.size());
	for (const std::string& m : messages) { pbVarint(out, m.size()); out += m; }
	return out;
}

} // namespace oracle
