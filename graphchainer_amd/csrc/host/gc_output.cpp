#include "gc_output.hpp"
#include <sstream>

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

inline void addCigarItem(std::ostringstream& str, uint64_t length, Edit type)
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
	str << length;            // (the reference writes the length first and then returns without an op for Empty)
	if (op) str << op;
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
	auto graphChar = [&](uint64_t i) {
		size_t split = graph.GetUnitigNode(trace.node[i], trace.offset[i]);
		return graph.NodeSequences(split, trace.offset[i] - graph.NodeOffset(split));
	};
	auto readChar = [&](uint64_t i) { return trace.seqPos[i] < readLength ? sequence[trace.seqPos[i]] : '-'; };
	auto originalSize = [&](int nodeId) { return graph.originalNodeSize.at(nodeId); };
	std::ostringstream cigar, nodePath;
	auto addNode = [&](int nodeId) {
		nodePath << ((nodeId % 2) == 1 ? "<" : ">");
		std::string name = graph.OriginalNodeName(nodeId);
		if (name.empty()) nodePath << nodeId / 2; else nodePath << name;
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
	out << readName << "\t" << readLength << "\t" << readStart << "\t" << readEnd << "\t" << "+" << "\t" << nodePath.str() << "\t" << nodePathLen << "\t" << nodePathStart << "\t" << nodePathEnd
		<< "\t" << matches << "\t" << trace.size << "\t" << 255;
	out << "\t" << "NM:i:" << (mismatches + deletions + insertions);
	out << "\t" << "dv:f:" << 1.0 - ((double)matches / (double)all);
	out << "\t" << "id:f:" << ((double)matches / (double)all);
	out << "\t" << "cg:Z:" << cigar.str();
	return out.str();
}

} // namespace gc
