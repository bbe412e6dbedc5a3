#include "gc_output.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <stdexcept>
#include <zlib.h>

namespace gc {

namespace {

// set of plain bases a character stands for: bit 0 A, 1 C, 2 G, 3 T; 0 = not a base symbol
inline unsigned baseSet(char c)
{
	switch (c) {
		case 'A': case 'a': return 1;
		case 'C': case 'c': return 2;
		case 'G': case 'g': return 4;
		case 'T': case 't': case 'U': case 'u': return 8;
		case 'R': case 'r': return 1 | 4;
		case 'Y': case 'y': return 2 | 8;
		case 'K': case 'k': return 4 | 8;
		case 'M': case 'm': return 1 | 2;
		case 'S': case 's': return 2 | 4;
		case 'W': case 'w': return 1 | 8;
		case 'B': case 'b': return 2 | 4 | 8;
		case 'D': case 'd': return 1 | 4 | 8;
		case 'H': case 'h': return 1 | 2 | 8;
		case 'V': case 'v': return 1 | 2 | 4;
		case 'N': case 'n': return 15;
	}
	return 0;
}

enum class Edit { Match, Mismatch, MatchOrMismatch, Insertion, Deletion, Empty };

// decimal digits of v appended to out (what `stream << v` writes for an unsigned integer; a CIGAR of a 10 kb read has ~2 000 runs and a stream insertion costs ~20x this)
inline void appendUint(std::string& out, uint64_t v)
{
	char buf[24];
	int at = 24;
	do { buf[--at] = (char)('0' + v % 10); v /= 10; } while (v);
	out.append(buf + at, buf + 24);
}

inline void addCigarItem(std::string& str, uint64_t length, Edit type)
{
	if (length == 0) return;
	char op = 0;
	switch (type) {
		case Edit::MatchOrMismatch: op = 'M'; break;
		case Edit::Match: op = '='; break;
		case Edit::Mismatch: op = 'X'; break;
		case Edit::Insertion: op = 'I'; break;
		case Edit::Deletion: op = 'D'; break;
		case Edit::Empty: break;
	}
	appendUint(str, length);  // (the reference writes the length first and then returns without an op for Empty)
	if (op) str += op;
}

} // namespace

bool characterMatch(char sequenceCharacter, char graphCharacter)
{
	if (sequenceCharacter == graphCharacter) return true;
	if (sequenceCharacter == '-' || graphCharacter == '-') return false;
	return (baseSet(sequenceCharacter) & baseSet(graphCharacter)) != 0;
}

std::string formatGafLine(const AlignmentGraph& graph, const std::string& readName, const char* sequence, uint64_t readLength, const TraceView& trace, bool merge)
{
	if (trace.size == 0) return std::string();
	GraphLetters letters(graph);
	auto graphChar = [&](uint64_t i) { return letters.at(trace.node[i], trace.offset[i]); };
	auto readChar = [&](uint64_t i) { return trace.seqPos[i] < readLength ? sequence[trace.seqPos[i]] : '-'; };
	auto originalSize = [&](int nodeId) { return graph.originalNodeSize.at(nodeId); };
	std::string cigar, nodePath;
	cigar.reserve(trace.size / 4 + 16);
	auto addNode = [&](int nodeId) {
		nodePath += (nodeId % 2) == 1 ? '<' : '>';
		std::string name = graph.OriginalNodeName(nodeId);
		if (name.empty()) { const int64_t half = (int64_t)(nodeId / 2); if (half < 0) { nodePath += '-'; appendUint(nodePath, (uint64_t)(-half)); } else appendUint(nodePath, (uint64_t)half); } else nodePath += name;   // (`stream << nodeId / 2`)
	};
	const uint64_t readStart = trace.seqPos[0], readEnd = (uint64_t)trace.seqPos[trace.size - 1] + 1;
	uint64_t nodePathLen = 0, matches = 0, mismatches = 0, deletions = 0, insertions = 0, editLength = 1;
	const uint64_t nodePathStart = trace.offset[0];
	int currentNode = trace.node[0];
	uint32_t currentOffset = trace.offset[0];
	Edit currentEdit;
	const bool firstMatches = characterMatch(readChar(0), graphChar(0));
	if (merge) currentEdit = Edit::MatchOrMismatch;
	else currentEdit = firstMatches ? Edit::Match : Edit::Mismatch;
	if (firstMatches) matches++; else mismatches++;
	addNode(currentNode);
	nodePathLen += originalSize(currentNode);
	auto switchTo = [&](Edit e) {
		if (currentEdit == Edit::Empty) currentEdit = e;
		if (currentEdit != e) { addCigarItem(cigar, editLength, currentEdit); currentEdit = e; editLength = 0; }
		editLength++;
	};
	for (uint64_t pos = 1; pos < trace.size; pos++) {
		const int newNode = trace.node[pos];
		const uint32_t newOffset = trace.offset[pos];
		// a switch flag between two cells of the same node going forward is not a new path step (src/GraphAlignerGAFAlignment.h:101)
		const bool insideNode = !trace.nodeSwitch[pos - 1] || (newNode == currentNode && newOffset > currentOffset);
		if (!insideNode) {
			uint64_t skippedBefore = originalSize(currentNode) - 1 - trace.offset[pos - 1];
			currentNode = newNode;
			currentOffset = newOffset;
			addNode(currentNode);
			uint64_t skippedAfter = trace.offset[pos];
			nodePathLen += originalSize(currentNode) - (skippedBefore + skippedAfter);
		}
		if (trace.seqPos[pos - 1] == trace.seqPos[pos]) { switchTo(Edit::Deletion); deletions++; }
		else if (insideNode && trace.offset[pos - 1] == trace.offset[pos]) { switchTo(Edit::Insertion); insertions++; }
		else {
			const bool m = characterMatch(readChar(pos), graphChar(pos));
			switchTo(merge ? Edit::MatchOrMismatch : (m ? Edit::Match : Edit::Mismatch));
			if (m) matches++; else mismatches++;
		}
	}
	addCigarItem(cigar, editLength, currentEdit);
	const uint64_t nodePathEnd = nodePathLen - (originalSize(trace.node[trace.size - 1]) - 1 - trace.offset[trace.size - 1]);
	const uint64_t all = matches + mismatches + deletions + insertions;
	std::ostringstream out;
	out << readName << "\t" << readLength << "\t" << readStart << "\t" << readEnd << "\t" << "+" << "\t" << nodePath << "\t" << nodePathLen << "\t" << nodePathStart << "\t" << nodePathEnd
		<< "\t" << matches << "\t" << trace.size << "\t" << 255;
	out << "\t" << "NM:i:" << (mismatches + deletions + insertions);
	out << "\t" << "dv:f:" << 1.0 - ((double)matches / (double)all);
	out << "\t" << "id:f:" << ((double)matches / (double)all);
	out << "\t" << "cg:Z:" << cigar;
	return out.str();
}

VgAlignment buildVgAlignment(const AlignmentGraph& graph, const std::string& readName, const char* sequence, uint64_t readLength, const TraceView& trace,
	int32_t score, uint64_t alignmentStart, uint64_t alignmentEnd)
{
	VgAlignment aln;
	if (trace.size == 0) return aln;
	GraphLetters letters(graph);
	auto graphChar = [&](uint64_t i) { return letters.at(trace.node[i], trace.offset[i]); };
	auto readChar = [&](uint64_t i) { return trace.seqPos[i] < readLength ? sequence[trace.seqPos[i]] : '-'; };
	aln.name = readName;
	aln.score = score;
	uint64_t matches = 0, mismatches = 0, insertions = 0, deletions = 0;
	int currentNode = trace.node[0];
	uint32_t currentOffset = trace.offset[0];
	int64_t rank = 0;
	auto newMapping = [&](int node, uint32_t offset) {
		VgMapping m;
		m.nodeId = node;                 // digraph id for now; replaced below
		m.isReverse = (node % 2) == 1;
		m.offset = offset;
		m.rank = rank;
		m.edits.emplace_back();
		aln.mappings.push_back(m);
	};
	enum class E { Match, Mismatch, Insertion, Deletion, Empty } currentEdit;
	newMapping(currentNode, currentOffset);
	{
		VgEdit& e = aln.mappings.back().edits.back();
		e.fromLength++; e.toLength++;
		if (characterMatch(readChar(0), graphChar(0))) { currentEdit = E::Match; matches++; }
		else { currentEdit = E::Mismatch; e.sequence = std::string(1, sequence[0]); mismatches++; }   // (sic) sequence[0], src/GraphAlignerVGAlignment.h:73
	}
	for (uint64_t pos = 1; pos < trace.size; pos++) {
		const int newNode = trace.node[pos];
		const uint32_t newOffset = trace.offset[pos];
		const bool insideNode = !trace.nodeSwitch[pos - 1] || (newNode == currentNode && newOffset > currentOffset);
		if (!insideNode) {
			rank++;
			currentNode = newNode;
			currentOffset = newOffset;
			newMapping(currentNode, currentOffset);
			currentEdit = E::Empty;
		}
		auto edit = [&](E kind) -> VgEdit& {
			if (currentEdit == E::Empty) currentEdit = kind;
			if (currentEdit != kind) { aln.mappings.back().edits.emplace_back(); currentEdit = kind; }
			return aln.mappings.back().edits.back();
		};
		if (trace.seqPos[pos - 1] == trace.seqPos[pos]) { edit(E::Deletion).fromLength++; deletions++; }
		else if (insideNode && trace.offset[pos - 1] == trace.offset[pos]) { VgEdit& e = edit(E::Insertion); e.toLength++; e.sequence += readChar(pos); insertions++; }
		else if (characterMatch(readChar(pos), graphChar(pos))) { VgEdit& e = edit(E::Match); e.fromLength++; e.toLength++; matches++; }
		else { VgEdit& e = edit(E::Mismatch); e.fromLength++; e.toLength++; e.sequence += readChar(pos); mismatches++; }
	}
	aln.identity = (double)matches / (double)(matches + mismatches + insertions + deletions);
	// AddAlignment: the aligned part of the read and where it starts
	aln.sequence = std::string(sequence + alignmentStart, sequence + alignmentEnd);
	aln.queryPosition = (int32_t)alignmentStart;
	// replaceDigraphNodeIdsWithOriginalNodeIds
	for (VgMapping& m : aln.mappings) {
		int digraphNodeId = (int)m.nodeId;
		m.nodeId = digraphNodeId / 2;
		m.name = graph.OriginalNodeName(digraphNodeId);
	}
	return aln;
}

// ---- JSON (protobuf's json_util conventions: fields in number order, proto3 defaults omitted, 64-bit integers as strings)

namespace {

void jsonString(std::string& out, const std::string& s)
{
	static const char* hex = "0123456789abcdef";
	out += '"';
	for (unsigned char c : s) {
		switch (c) {
			case '"': out += "\\\""; break;
			case '\\': out += "\\\\"; break;
			case '\b': out += "\\b"; break;
			case '\f': out += "\\f"; break;
			case '\n': out += "\\n"; break;
			case '\r': out += "\\r"; break;
			case '\t': out += "\\t"; break;
			case '<': out += "\\u003c"; break;
			case '>': out += "\\u003e"; break;
			default:
				if (c < 0x20 || c == 0x7f) { out += "\\u00"; out += hex[c >> 4]; out += hex[c & 15]; }
				else out += (char)c;
		}
	}
	out += '"';
}

std::string jsonDouble(double v)
{
	char buf[40];
	snprintf(buf, sizeof buf, "%.15g", v);
	if (strtod(buf, nullptr) != v) snprintf(buf, sizeof buf, "%.17g", v);
	return buf;
}

struct JsonObject {   // "{" field, field, ... "}"
	std::string& out;
	bool first = true;
	explicit JsonObject(std::string& o) : out(o) { out += '{'; }
	void key(const char* k) { if (!first) out += ','; first = false; out += '"'; out += k; out += "\":"; }
	void close() { out += '}'; }
};

// wire format helpers
void putVarint(std::string& out, uint64_t v) { while (v >= 0x80) { out += (char)(v | 0x80); v >>= 7; } out += (char)v; }
void putTag(std::string& out, uint32_t field, uint32_t wireType) { putVarint(out, ((uint64_t)field << 3) | wireType); }
void putBytes(std::string& out, uint32_t field, const std::string& s) { putTag(out, field, 2); putVarint(out, s.size()); out += s; }

} // namespace

std::string vgToJson(const VgAlignment& aln)
{
	std::string out;
	JsonObject a(out);
	if (!aln.sequence.empty()) { a.key("sequence"); jsonString(out, aln.sequence); }
	{
		a.key("path");
		JsonObject p(out);
		if (!aln.mappings.empty()) {
			p.key("mapping");
			out += '[';
			for (size_t i = 0; i < aln.mappings.size(); i++) {
				const VgMapping& m = aln.mappings[i];
				if (i) out += ',';
				JsonObject mo(out);
				mo.key("position");
				{
					JsonObject po(out);
					if (m.nodeId != 0) { po.key("node_id"); out += '"'; out += std::to_string(m.nodeId); out += '"'; }
					if (m.offset != 0) { po.key("offset"); out += '"'; out += std::to_string(m.offset); out += '"'; }
					if (m.isReverse) { po.key("is_reverse"); out += "true"; }
					if (!m.name.empty()) { po.key("name"); jsonString(out, m.name); }
					po.close();
				}
				if (!m.edits.empty()) {
					mo.key("edit");
					out += '[';
					for (size_t e = 0; e < m.edits.size(); e++) {
						if (e) out += ',';
						JsonObject eo(out);
						if (m.edits[e].fromLength != 0) { eo.key("from_length"); out += std::to_string(m.edits[e].fromLength); }
						if (m.edits[e].toLength != 0) { eo.key("to_length"); out += std::to_string(m.edits[e].toLength); }
						if (!m.edits[e].sequence.empty()) { eo.key("sequence"); jsonString(out, m.edits[e].sequence); }
						eo.close();
					}
					out += ']';
				}
				if (m.rank != 0) { mo.key("rank"); out += '"'; out += std::to_string(m.rank); out += '"'; }
				mo.close();
			}
			out += ']';
		}
		p.close();
	}
	if (!aln.name.empty()) { a.key("name"); jsonString(out, aln.name); }
	if (aln.score != 0) { a.key("score"); out += std::to_string(aln.score); }
	if (aln.queryPosition != 0) { a.key("query_position"); out += std::to_string(aln.queryPosition); }
	if (aln.identity != 0) { a.key("identity"); out += jsonDouble(aln.identity); }
	a.close();
	return out;
}

std::string vgToProtobuf(const VgAlignment& aln)
{
	std::string path;
	for (const VgMapping& m : aln.mappings) {
		std::string pos;
		if (m.nodeId != 0) { putTag(pos, 1, 0); putVarint(pos, (uint64_t)m.nodeId); }
		if (m.offset != 0) { putTag(pos, 2, 0); putVarint(pos, (uint64_t)m.offset); }
		if (m.isReverse) { putTag(pos, 4, 0); putVarint(pos, 1); }
		if (!m.name.empty()) putBytes(pos, 5, m.name);
		std::string mapping;
		putBytes(mapping, 1, pos);          // position is always present (set_allocated_position), even when empty
		for (const VgEdit& e : m.edits) {
			std::string ed;
			if (e.fromLength != 0) { putTag(ed, 1, 0); putVarint(ed, (uint64_t)(int64_t)e.fromLength); }
			if (e.toLength != 0) { putTag(ed, 2, 0); putVarint(ed, (uint64_t)(int64_t)e.toLength); }
			if (!e.sequence.empty()) putBytes(ed, 3, e.sequence);
			putBytes(mapping, 2, ed);
		}
		if (m.rank != 0) { putTag(mapping, 5, 0); putVarint(mapping, (uint64_t)m.rank); }
		putBytes(path, 2, mapping);
	}
	std::string out;
	if (!aln.sequence.empty()) putBytes(out, 1, aln.sequence);
	putBytes(out, 2, path);                 // path is always present (set_allocated_path)
	if (!aln.name.empty()) putBytes(out, 3, aln.name);
	if (aln.score != 0) { putTag(out, 6, 0); putVarint(out, (uint64_t)(int64_t)aln.score); }
	if (aln.queryPosition != 0) { putTag(out, 7, 0); putVarint(out, (uint64_t)(int64_t)aln.queryPosition); }
	if (aln.identity != 0) { putTag(out, 16, 1); uint64_t bits; memcpy(&bits, &aln.identity, 8); for (int i = 0; i < 8; i++) out += (char)(bits >> (8 * i)); }
	return out;
}

// ---- alignments encoded on the device: the host's part of the line / message

void appendGafLine(std::string& out, const std::string& readName, uint64_t readLength, const EncodedAlignment& a)
{
	// the columns of formatGafLine above; `stream << double` is printf's %g
	const uint64_t all = a.matches + a.mismatches + a.deletions + a.insertions;
	const double identity = (double)a.matches / (double)all;
	char num[64];
	out += readName; out += '\t'; appendUint(out, readLength); out += '\t'; appendUint(out, a.alignmentStart); out += '\t'; appendUint(out, a.alignmentEnd); out += "\t+\t";
	out.append(a.path, a.pathLen);
	out += '\t'; appendUint(out, a.nodePathLen); out += '\t'; appendUint(out, a.nodePathStart); out += '\t'; appendUint(out, a.nodePathEnd);
	out += '\t'; appendUint(out, a.matches); out += '\t'; appendUint(out, a.cells); out += "\t255\tNM:i:"; appendUint(out, a.mismatches + a.deletions + a.insertions);
	snprintf(num, sizeof num, "%g", 1.0 - identity); out += "\tdv:f:"; out += num;
	snprintf(num, sizeof num, "%g", identity); out += "\tid:f:"; out += num;
	out += "\tcg:Z:";
	out.append(a.cigar, a.cigarLen);
}

std::string vgProtobufFromEncoded(const std::string& readName, const char* sequence, const EncodedAlignment& a)
{
	// vgToProtobuf's fields around the path the device wrote
	std::string out;
	out.reserve(a.vgPathLen + (a.alignmentEnd - a.alignmentStart) + readName.size() + 48);
	if (a.alignmentEnd > a.alignmentStart) { putTag(out, 1, 2); putVarint(out, a.alignmentEnd - a.alignmentStart); out.append(sequence + a.alignmentStart, sequence + a.alignmentEnd); }
	putTag(out, 2, 2); putVarint(out, a.vgPathLen); out.append((const char*)a.vgPath, a.vgPathLen);
	if (!readName.empty()) putBytes(out, 3, readName);
	if (a.score != 0) { putTag(out, 6, 0); putVarint(out, (uint64_t)(int64_t)a.score); }
	if ((int32_t)a.alignmentStart != 0) { putTag(out, 7, 0); putVarint(out, (uint64_t)(int64_t)(int32_t)a.alignmentStart); }
	const double identity = (double)a.matches / (double)(a.matches + a.mismatches + a.insertions + a.deletions);
	if (identity != 0) { putTag(out, 16, 1); uint64_t bits; memcpy(&bits, &identity, 8); for (int i = 0; i < 8; i++) out += (char)(bits >> (8 * i)); }
	return out;
}

namespace {
struct WireReader {
	const uint8_t* p; const uint8_t* end;
	bool done() const { return p >= end; }
	uint64_t varint() { uint64_t v = 0; int shift = 0; while (p < end) { uint8_t b = *p++; v |= (uint64_t)(b & 0x7f) << shift; if (!(b & 0x80)) return v; shift += 7; } throw std::runtime_error("truncated vg::Path bytes"); }
	WireReader sub() { uint64_t len = varint(); if ((uint64_t)(end - p) < len) throw std::runtime_error("truncated vg::Path bytes"); WireReader r { p, p + len }; p += len; return r; }
};
}

VgAlignment vgFromEncoded(const std::string& readName, const char* sequence, const EncodedAlignment& a)
{
	VgAlignment aln;
	aln.name = readName;
	aln.score = a.score;
	aln.identity = (double)a.matches / (double)(a.matches + a.mismatches + a.insertions + a.deletions);
	aln.sequence = std::string(sequence + a.alignmentStart, sequence + a.alignmentEnd);
	aln.queryPosition = (int32_t)a.alignmentStart;
	WireReader path { a.vgPath, a.vgPath + a.vgPathLen };
	while (!path.done()) {
		if (path.varint() != ((2u << 3) | 2u)) throw std::runtime_error("unexpected field in vg::Path bytes");
		WireReader m = path.sub();
		VgMapping mapping;
		while (!m.done()) {
			const uint64_t tag = m.varint();
			if (tag == ((1u << 3) | 2u)) {
				WireReader pos = m.sub();
				while (!pos.done()) {
					const uint64_t t = pos.varint();
					if (t == (1u << 3)) mapping.nodeId = (int64_t)pos.varint();
					else if (t == (2u << 3)) mapping.offset = (int64_t)pos.varint();
					else if (t == (4u << 3)) mapping.isReverse = pos.varint() != 0;
					else if (t == ((5u << 3) | 2u)) { WireReader nm = pos.sub(); mapping.name.assign((const char*)nm.p, (const char*)nm.end); }
					else throw std::runtime_error("unexpected field in vg::Position bytes");
				}
			} else if (tag == ((2u << 3) | 2u)) {
				WireReader e = m.sub();
				VgEdit edit;
				while (!e.done()) {
					const uint64_t t = e.varint();
					if (t == (1u << 3)) edit.fromLength = (int32_t)e.varint();
					else if (t == (2u << 3)) edit.toLength = (int32_t)e.varint();
					else if (t == ((3u << 3) | 2u)) { WireReader sq = e.sub(); edit.sequence.assign((const char*)sq.p, (const char*)sq.end); }
					else throw std::runtime_error("unexpected field in vg::Edit bytes");
				}
				mapping.edits.push_back(edit);
			} else if (tag == (5u << 3)) mapping.rank = (int64_t)m.varint();
			else throw std::runtime_error("unexpected field in vg::Mapping bytes");
		}
		aln.mappings.push_back(mapping);
	}
	return aln;
}

std::string gamGroupRaw(const std::vector<std::string>& messages)
{
	std::string raw;
	size_t bytes = 10;
	for (const std::string& m : messages) bytes += m.size() + 10;
	raw.reserve(bytes);
	putVarint(raw, messages.size());
	for (const std::string& m : messages) { putVarint(raw, m.size()); raw += m; }
	return raw;
}

// RFC 1952 framing of a deflate stream produced elsewhere (the device: hip/gc_deflate.hip): header without name or time, the deflate bytes, CRC-32 and length of the input
std::string gzipMember(const uint8_t* deflated, size_t deflatedBytes, const std::string& raw)
{
	std::string out;
	out.reserve(deflatedBytes + 18);
	const unsigned char header[10] = { 0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3 };
	out.append((const char*)header, 10);
	out.append((const char*)deflated, deflatedBytes);
	uLong crc = crc32(0L, Z_NULL, 0);
	crc = crc32(crc, (const Bytef*)raw.data(), (uInt)raw.size());
	for (int i = 0; i < 4; i++) out += (char)((crc >> (8 * i)) & 255);
	for (int i = 0; i < 4; i++) out += (char)((raw.size() >> (8 * i)) & 255);
	return out;
}

std::string gamGroup(const std::vector<std::string>& messages, int level)
{
	const std::string raw = gamGroupRaw(messages);
	// one gzip member (protobuf's GzipOutputStream defaults: gzip format, default compression level and strategy)
	z_stream zs;
	memset(&zs, 0, sizeof zs);
	if (deflateInit2(&zs, level, Z_DEFLATED, 15 | 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("zlib deflateInit2 failed");
	std::string out(deflateBound(&zs, raw.size()) + 64, '\0');
	zs.next_in = (Bytef*)raw.data(); zs.avail_in = (uInt)raw.size();
	zs.next_out = (Bytef*)&out[0]; zs.avail_out = (uInt)out.size();
	int rc = deflate(&zs, Z_FINISH);
	deflateEnd(&zs);
	if (rc != Z_STREAM_END) throw std::runtime_error("zlib deflate failed");
	out.resize(zs.total_out);
	return out;
}

} // namespace gc
