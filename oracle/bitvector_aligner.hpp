// ORACLE (test infrastructure, not product code): CPU restatement of the reference's banded
// bit-vector graph aligner (seed extension + backtrace).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// Follows, function by function:
//   src/GraphAlignerBitvectorBanded.h:46-71   getReverseTraceFromSeed
//   src/GraphAlignerBitvectorBanded.h:205-426 calculateSlice
//   src/GraphAlignerBitvectorBanded.h:428-476 fillDPSlice
//   src/GraphAlignerBitvectorBanded.h:513-701 getViterbiSlices
//   src/GraphAlignerBitvectorCommon.h:280-319 getEqVector
//   src/GraphAlignerBitvectorCommon.h:385-544 getReverseTraceFromTable (+ pickBacktrace* :556-804)
//   src/GraphAlignerBitvectorCommon.h:828-852 recalcNodeWordslice
//   src/GraphAlignerBitvectorCommon.h:885-1168 calculateNodeInner
//   src/GraphAlignerBitvectorCommon.h:1170-1229 flattenLastSliceEnd
//   src/GraphAlignerBitvectorCommon.h:1231-1279 removeWronglyAlignedEnd, getInitialSliceExactPosition
//   src/ComponentPriorityQueue.h (node scheduling), src/AlignmentCorrectnessEstimation.cpp
//
// Fixed configuration (the reference's defaults in chaining mode, src/AlignerMain.cpp:149,186-193):
// rampBandwidth 0, maxCellsPerSlice unlimited, preciseClipping off, forceGlobal off, Xdrop 0, low-memory
// slices. Options that change these are outside this build's scope.
//
// PARITY UNPINNED for one rule: flattenLastSliceEnd (:1184,1213) picks the minimum cell with a strict
// '<' in parallel-hashmap iteration order; that library is absent here (un-vendored submodule), so the
// order is DEFINED as the order in which nodes entered the slice's band (NodeSliceMap::order).
#pragma once
#include "wordslice.hpp"
#include "../graphchainer_amd/csrc/host/gc_graph.hpp"
#include <cassert>
#include <cmath>
#include <queue>
#include <stdexcept>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

namespace oracle {

using gc::AlignmentGraph;

// The reference's `assert` throws (src/ThreadReadAssertion.h:27) and a read/fragment that trips one is
// dropped by the caller (src/Aligner.cpp:585-592,695-701); asserts are on in its release build.
struct AssertionFailure : std::runtime_error { using std::runtime_error::runtime_error; };
#define ORACLE_ASSERT(cond) do { if (!(cond)) throw ::oracle::AssertionFailure(#cond); } while (0)

// ---- character matching. reference: src/GraphAlignerCommon.h:190-297
inline bool ambiguousMatch(char ambiguousChar, char exactChar)
{
	switch (ambiguousChar) {
		case '-': return false;
		case 'A': case 'a': return exactChar == 'A';
		case 'U': case 'u': case 'T': case 't': return exactChar == 'T';
		case 'C': case 'c': return exactChar == 'C';
		case 'G': case 'g': return exactChar == 'G';
		case 'N': case 'n': return true;
		case 'R': case 'r': return exactChar == 'A' || exactChar == 'G';
		case 'Y': case 'y': return exactChar == 'C' || exactChar == 'T';
		case 'K': case 'k': return exactChar == 'G' || exactChar == 'T';
		case 'M': case 'm': return exactChar == 'C' || exactChar == 'A';
		case 'S': case 's': return exactChar == 'C' || exactChar == 'G';
		case 'W': case 'w': return exactChar == 'A' || exactChar == 'T';
		case 'B': case 'b': return exactChar == 'C' || exactChar == 'G' || exactChar == 'T';
		case 'D': case 'd': return exactChar == 'A' || exactChar == 'G' || exactChar == 'T';
		case 'H': case 'h': return exactChar == 'A' || exactChar == 'C' || exactChar == 'T';
		case 'V': case 'v': return exactChar == 'A' || exactChar == 'C' || exactChar == 'G';
	}
	throw AssertionFailure("invalid character in sequence");
}
inline bool characterMatch(char sequenceCharacter, char graphCharacter)
{
	if (sequenceCharacter == graphCharacter) return true;
	switch (sequenceCharacter) {
		case 'a': case 'A': return ambiguousMatch(graphCharacter, 'A');
		case 'c': case 'C': return ambiguousMatch(graphCharacter, 'C');
		case 'g': case 'G': return ambiguousMatch(graphCharacter, 'G');
		case 't': case 'T': return ambiguousMatch(graphCharacter, 'T');
		case '-': return false;
	}
	return (ambiguousMatch(sequenceCharacter, 'A') && ambiguousMatch(graphCharacter, 'A'))
		|| (ambiguousMatch(sequenceCharacter, 'C') && ambiguousMatch(graphCharacter, 'C'))
		|| (ambiguousMatch(sequenceCharacter, 'G') && ambiguousMatch(graphCharacter, 'G'))
		|| (ambiguousMatch(sequenceCharacter, 'T') && ambiguousMatch(graphCharacter, 'T'));
}

// ---- 2-state Viterbi over per-slice mismatch counts. reference: src/AlignmentCorrectnessEstimation.cpp
struct CorrectnessTables {
	double correctOdds[64], wrongOdds[64];
	double f2c, f2f, c2f, c2c;
	CorrectnessTables()
	{
		const double correctMean = 0.1875, correctStddev = 0.0955, wrongMean = 0.5, wrongStddev = 0.0291;
		f2c = log(0.00001); f2f = log(1.0 - 0.00001); c2f = log(0.0000000001); c2c = log(1.0 - 0.0000000001);
		fill(correctOdds, correctMean * 64, correctStddev * 64);
		fill(wrongOdds, wrongMean * 64, wrongStddev * 64);
	}
	static void fill(double* out, double mean, double stddev)   // :20-54
	{
		std::vector<double> v;
		for (int i = 0; i <= 32; i++) v.push_back(-(i - mean) * (i - mean) / (2 * stddev * stddev));
		double sum = 0;
		for (double x : v) sum += exp(x);
		double add = log(1.0 / sum);
		for (double& x : v) x += add;
		for (int i = 32; i < 64; i++) v.push_back(v.back());
		for (int i = 0; i < 64; i++) out[i] = v[i];   // the reference's vector has 65 entries; index 64+ clamps to back() (== [63])
	}
};
inline const CorrectnessTables& correctnessTables() { static CorrectnessTables t; return t; }

struct CorrectnessState {
	double correctLogOdds = log(0.8), falseLogOdds = log(0.2);
	bool correctFromCorrectTrace = false, falseFromCorrectTrace = false;
	bool CurrentlyCorrect() const { return correctLogOdds > falseLogOdds; }
	bool CorrectFromCorrect() const { return correctFromCorrectTrace; }
	bool FalseFromCorrect() const { return falseFromCorrectTrace; }
	CorrectnessState NextState(int mismatches) const   // :105-129
	{
		ORACLE_ASSERT(mismatches >= 0);
		const CorrectnessTables& t = correctnessTables();
		CorrectnessState r;
		r.correctFromCorrectTrace = correctLogOdds + t.c2c >= falseLogOdds + t.f2c;
		r.falseFromCorrectTrace = correctLogOdds + t.c2f >= falseLogOdds + t.f2f;
		double newCorrect = std::max(correctLogOdds + t.c2c, falseLogOdds + t.f2c);
		double newFalse = std::max(correctLogOdds + t.c2f, falseLogOdds + t.f2f);
		int idx = mismatches < 64 ? mismatches : 63;
		r.correctLogOdds = newCorrect + t.correctOdds[idx];
		r.falseLogOdds = newFalse + t.wrongOdds[idx];
		return r;
	}
};

// ---- per-slice node map. reference: src/NodeSlice.h:15-47 (item), :50-427 (map)
struct NodeSliceItem {
	WordSlice startSlice, endSlice;
	bool exists = false;
	uint64_t HP = 0, HN = 0;
	int32_t minScore = 0;
};
class NodeSliceMap {
public:
	bool hasNode(size_t n) const { return index.count(n) != 0; }
	NodeSliceItem& node(size_t n) { return items[index.at(n)].second; }
	const NodeSliceItem& node(size_t n) const { return items[index.at(n)].second; }
	void addNode(size_t n)   // src/NodeSlice.h:244-253: a fresh item has minScore and both scoreEnds at INT_MAX
	{
		ORACLE_ASSERT(!hasNode(n));
		index[n] = items.size();
		NodeSliceItem item;
		item.minScore = INT32_MAX;
		item.startSlice = WordSlice(0, 0, INT32_MAX);
		item.endSlice = WordSlice(0, 0, INT32_MAX);
		items.emplace_back(n, item);
	}
	size_t size() const { return items.size(); }
	std::vector<std::pair<size_t, NodeSliceItem>> items;   // band-entry order (the defined iteration order)
private:
	std::unordered_map<size_t, size_t> index;
};

struct DPSlice {   // reference: src/GraphAlignerBitvectorCommon.h:138-214
	int32_t minScore = INT32_MAX;
	size_t minScoreNode = SIZE_MAX, minScoreNodeOffset = SIZE_MAX;
	NodeSliceMap scores;
	CorrectnessState correctness;
	size_t j = SIZE_MAX;
	size_t cellsProcessed = 0;
	size_t bandwidth = 0;
	bool flattenTie = false;   // (this build's bookkeeping, not the reference's) flattenLastSliceEnd found its minimum in more than one node: minScoreNode / minScoreNodeOffset depend on the iteration order
};
struct DPTable { std::vector<DPSlice> slices; };

struct MatrixPosition {   // reference: src/AlignmentGraph.h:70-78
	size_t node, nodeOffset, seqPos;
	bool operator==(const MatrixPosition& o) const { return node == o.node && nodeOffset == o.nodeOffset && seqPos == o.seqPos; }
	bool operator!=(const MatrixPosition& o) const { return !(*this == o); }
};
struct TraceItem {        // reference: src/GraphAlignerCommon.h:127-157
	MatrixPosition DPposition;
	bool nodeSwitch;
	char sequenceCharacter, graphCharacter;
};
struct OnewayTrace {      // reference: src/GraphAlignerCommon.h:158-183
	std::vector<TraceItem> trace;
	int32_t score = 0;
	static OnewayTrace TraceFailed() { OnewayTrace t; t.score = INT32_MAX; return t; }
	bool failed() const { return score == INT32_MAX; }
};

struct EqVector { uint64_t masks[4]; };   // A,C,G,T match masks of 64 read rows. reference: ...Common.h:37-46

struct EdgeWithPriority {   // reference: src/GraphAlignerCommon.h:30-50
	size_t target;
	int priority;
	WordSlice incoming;
	bool skipFirst;
};

// Work counters in the unit SURVEY.md §8d prices (tiles of one node x one 64-row slice).
struct AlignerCounters {
	uint64_t dpTiles = 0, recomputeTiles = 0, columnSteps = 0, traceItems = 0, extensions = 0;
	// r5: how often the one rule this build DEFINES instead of reproducing is exercised (the header's PARITY UNPINNED note): calls of flattenLastSliceEnd, and the
	// extensions whose backtrace STARTED from a flattened slice whose minimum is attained in more than one node - only there can the reference's parallel-hashmap
	// iteration order pick a different cell than band-entry order does
	uint64_t flattenCalls = 0, flattenTies = 0;
};

// Per-thread reusable state. reference: src/GraphAlignerCommon.h:51-93 (AlignerGraphsizedState)
struct AlignerState {
	std::vector<bool> currentBand, previousBand;
	AlignerCounters counters;
	int tieOrder = 0;   // flattenLastSliceEnd's iteration order over the slice's nodes: 0 band-entry order (the defined order), 1 the reverse (the sensitivity runs of tests/test_oracle_golden.py)
	explicit AlignerState(const AlignmentGraph& g) : currentBand(g.NodeSize(), false), previousBand(g.NodeSize(), false) {}
	void clear() { currentBand.assign(currentBand.size(), false); previousBand.assign(previousBand.size(), false); }
};

class BitvectorAligner {
public:
	BitvectorAligner(const AlignmentGraph& graph, size_t bandwidth) : graph(graph), initialBandwidth(bandwidth) {}

	// reference: src/GraphAlignerBitvectorBanded.h:46-71
	OnewayTrace getReverseTraceFromSeed(std::string_view sequence, int bigraphNodeId, size_t nodeOffset, AlignerState& state) const
	{
		state.counters.extensions++;
		size_t numSlices = (sequence.size() + 63) / 64;
		DPSlice initial = getInitialSliceExactPosition(bigraphNodeId, nodeOffset);
		DPTable table = getViterbiSlices(sequence, initial, numSlices, state);
		removeWronglyAlignedEnd(table);
		if (table.slices.size() <= 1) return OnewayTrace::TraceFailed();
		ORACLE_ASSERT(table.slices.back().minScore >= 0);
		ORACLE_ASSERT(table.slices.back().minScore <= (int32_t)sequence.size() + 128);
		// getReverseTraceFromTableStartLastRow, ...Common.h:385-390
		const DPSlice& last = table.slices.back();
		MatrixPosition startPos { last.minScoreNode, last.minScoreNodeOffset, std::min(last.j + 63, sequence.size() - 1) };
		OnewayTrace trace = getReverseTraceFromTable(sequence, table, startPos, last.minScore, state);
		// the tie counts where it can change an output: the flattened slice is the one the backtrace starts in (a slice the correctness estimate dropped or trimmed has
		// handed on nothing but its minimum SCORE, which no order changes) and the extension produced a trace
		if (last.flattenTie) state.counters.flattenTies++;
		return trace;
	}

	static EqVector getEqVector(std::string_view sequence, size_t j)   // ...Common.h:280-319
	{
		EqVector e { { 0, 0, 0, 0 } };
		for (int i = 0; i < 64 && j + i < sequence.size(); i++) {
			uint64_t mask = (uint64_t)1 << i;
			switch (sequence[j + i]) {
				case 'a': case 'A': e.masks[0] |= mask; break;
				case 'c': case 'C': e.masks[1] |= mask; break;
				case 'g': case 'G': e.masks[2] |= mask; break;
				case 't': case 'T': e.masks[3] |= mask; break;
				default:
					if (characterMatch(sequence[j + i], 'A')) e.masks[0] |= mask;
					if (characterMatch(sequence[j + i], 'C')) e.masks[1] |= mask;
					if (characterMatch(sequence[j + i], 'T')) e.masks[3] |= mask;
					if (characterMatch(sequence[j + i], 'G')) e.masks[2] |= mask;
			}
		}
		return e;
	}

private:
	const AlignmentGraph& graph;
	size_t initialBandwidth;

	uint64_t eqOfColumn(const EqVector& EqV, size_t node, size_t pos) const   // EqVector::getEqI, ...Common.h:47-56,114-118
	{
		if (node < graph.firstAmbiguous) return EqV.masks[(graph.nodeSequences[node][pos / 32] >> ((pos % 32) * 2)) & 3];
		const gc::AmbiguousSeq& s = graph.ambiguousNodeSequences[node - graph.firstAmbiguous];
		uint64_t r = 0;
		if ((s.A >> pos) & 1) r |= EqV.masks[0];
		if ((s.C >> pos) & 1) r |= EqV.masks[1];
		if ((s.G >> pos) & 1) r |= EqV.masks[2];
		if ((s.T >> pos) & 1) r |= EqV.masks[3];
		return r;
	}

	// reference: ...Common.h:1243-1279. Row -1 scores are |column - offsetInNode| on the seed's split node.
	DPSlice getInitialSliceExactPosition(int bigraphNodeId, size_t offset) const
	{
		DPSlice result;
		result.j = (size_t)-64;
		result.bandwidth = 1;
		result.minScore = 0;
		ORACLE_ASSERT(offset < graph.originalNodeSize.at(bigraphNodeId));
		size_t nodeIndex = graph.GetUnitigNode(bigraphNodeId, offset);
		size_t offsetInNode = offset - graph.nodeOffset[nodeIndex];
		ORACLE_ASSERT(offsetInNode < graph.NodeLength(nodeIndex));
		result.scores.addNode(nodeIndex);
		result.minScoreNode = nodeIndex;
		result.minScoreNodeOffset = offsetInNode;
		NodeSliceItem& node = result.scores.node(nodeIndex);
		node.startSlice = WordSlice(0, 0, (int32_t)offsetInNode);
		node.endSlice = WordSlice(0, 0, (int32_t)graph.NodeLength(nodeIndex) - 1 - (int32_t)offsetInNode);
		node.minScore = 0;
		node.exists = true;
		for (size_t i = 1; i <= offsetInNode; i++) node.HN |= (uint64_t)1 << i;
		for (size_t i = offsetInNode + 1; i < graph.NodeLength(nodeIndex); i++) node.HP |= (uint64_t)1 << i;
		return result;
	}

	struct NodeCalculationResult { int32_t minScore; size_t minScoreNode, minScoreNodeOffset, cellsProcessed; bool flattenTie = false; };

	// reference: ...Common.h:885-1168 with PreciseClipping=false. `columns` (non-null) collects every
	// column's WordSlice and corresponds to AllowEarlyLeave=false (recalcNodeWordslice).
	NodeCalculationResult calculateNodeInner(size_t i, NodeSliceItem& slice, const EqVector& EqV, NodeSliceItem previousSlice, const std::vector<EdgeWithPriority>& incoming, const std::vector<bool>* previousBand, std::vector<WordSlice>* columns, AlignerState& state) const
	{
		const bool allowEarlyLeave = columns == nullptr;
		ORACLE_ASSERT(incoming.size() > 0);
		WordSlice ws;
		bool hasWs = false;
		NodeCalculationResult result { INT32_MAX, SIZE_MAX, SIZE_MAX, 0 };
		size_t nodeLength = graph.NodeLength(i);
		uint64_t Eq = eqOfColumn(EqV, i, 0);
		bool hasSkipless = false;
		for (const EdgeWithPriority& inc : incoming) {
			result.cellsProcessed++;
			if (inc.skipFirst) {
				ws = hasWs ? mergeTwoSlices(ws, inc.incoming) : inc.incoming;
				hasWs = true;
				continue;
			}
			hasSkipless = true;
			uint64_t hinP, hinN;
			if (previousSlice.exists) {
				int32_t before = inc.incoming.getScoreBeforeStart();
				if (previousSlice.startSlice.scoreEnd < before) { hinP = 0; hinN = 1; }
				else if (previousSlice.startSlice.scoreEnd > before) { hinP = 1; hinN = 0; }
				else { hinP = 0; hinN = 0; }
			} else { hinP = 1; hinN = 0; }
			WordSlice stepped = getNextSlice(Eq, inc.incoming, hinP, hinN).ws;
			if (!previousSlice.exists || stepped.getScoreBeforeStart() < previousSlice.startSlice.scoreEnd) {
				stepped.VP &= ~(uint64_t)1;
				stepped.VN |= 1;
			}
			ws = hasWs ? mergeTwoSlices(ws, stepped) : stepped;
			hasWs = true;
		}
		ORACLE_ASSERT(hasWs);
		result.minScore = ws.scoreEnd;
		result.minScoreNode = i;
		result.minScoreNodeOffset = 0;

		if (slice.exists) {
			bool inBand = previousBand != nullptr && graph.inNeighbors[i].size() == 1 && (*previousBand)[graph.inNeighbors[i][0]];
			if (hasSkipless && graph.inNeighbors[i].size() == 1 && inBand) {
				if (ws.scoreEnd > slice.startSlice.scoreEnd) {
					if (allowEarlyLeave) return result;
				} else if (ws.scoreEnd < slice.startSlice.scoreEnd) {
				} else {
					uint64_t newBigger = (ws.VP & ~slice.startSlice.VP) | (slice.startSlice.VN & ~ws.VN);
					uint64_t oldBigger = (slice.startSlice.VP & ~ws.VP) | (ws.VN & ~slice.startSlice.VN);
					if (newBigger > oldBigger) {
					} else if (oldBigger > newBigger) {
						if (allowEarlyLeave) return result;
					} else if (newBigger == 0 && oldBigger == 0) {
						if (allowEarlyLeave) return result;
					} else {
						WordSlice test = mergeTwoSlices(ws, slice.startSlice);
						if (test == slice.startSlice) { if (allowEarlyLeave) return result; }
						ws = test;
					}
				}
			} else {
				WordSlice test = mergeTwoSlices(ws, slice.startSlice);
				// (sic) the reference compares test.VP with startSlice.VN, ...Common.h:1044
				if (test.scoreEnd == slice.startSlice.scoreEnd && test.VP == slice.startSlice.VP && test.VP == slice.startSlice.VN) { if (allowEarlyLeave) return result; }
				ws = test;
			}
		}
		if (previousSlice.exists && ws.getScoreBeforeStart() > previousSlice.startSlice.scoreEnd)
			ws = mergeTwoSlices(ws, getSourceSliceFromScore(previousSlice.startSlice.scoreEnd));

		slice.HP = 0;
		slice.HN = 0;
		size_t forceUntil = 0;
		if (previousSlice.exists) {
			// repair of the previous slice's last-row deltas where this column now enters cheaper, :1068-1104
			int32_t scoreBefore = ws.getScoreBeforeStart();
			int32_t scoreComparison = previousSlice.startSlice.scoreEnd;
			ORACLE_ASSERT(scoreBefore <= scoreComparison);
			if (scoreBefore < scoreComparison) {
				for (size_t fix = 1; fix < 64; fix++) {
					int32_t next = scoreComparison + (int32_t)((previousSlice.HP >> fix) & 1) - (int32_t)((previousSlice.HN >> fix) & 1);
					uint64_t mask = (uint64_t)1 << fix;
					ORACLE_ASSERT(scoreBefore <= next);
					if (scoreBefore < next) { previousSlice.HP |= mask; previousSlice.HN &= ~mask; forceUntil = fix; }
					if (scoreBefore == next) { previousSlice.HP &= ~mask; previousSlice.HN &= ~mask; }
					scoreBefore++;
					scoreComparison = next;
					if (scoreBefore >= scoreComparison) break;
				}
			}
		} else {
			forceUntil = nodeLength;
		}
		slice.startSlice = ws;
		if (columns) columns->push_back(ws);
		slice.exists = true;
		uint64_t forceEq = ~(uint64_t)0;
		if (!previousSlice.exists) forceEq ^= 1;
		size_t pos = 1;
		for (; pos < nodeLength; pos++) {
			Eq = eqOfColumn(EqV, i, pos) & forceEq;
			StepResult st = getNextSlice(Eq, ws, (previousSlice.HP >> pos) & 1, (previousSlice.HN >> pos) & 1);
			if (forceUntil >= pos) { st.ws.VP &= ~(uint64_t)1; st.ws.VN |= 1; }
			ws = st.ws;
			if (ws.scoreEnd < result.minScore) { result.minScore = ws.scoreEnd; result.minScoreNodeOffset = pos; }
			if (columns) columns->push_back(ws);
			slice.HP |= st.houtP << pos;
			slice.HN |= st.houtN << pos;
		}
		// the reference's chunk loop (:1118-1161) leaves pos == max(nodeLength, 1)
		result.cellsProcessed = pos;
		state.counters.columnSteps += pos;
		slice.endSlice = ws;
		return result;
	}

	// reference: ...Common.h:828-852
	std::vector<WordSlice> recalcNodeWordslice(size_t node, const NodeSliceItem& slice, const EqVector& EqV, const NodeSliceItem& previousSlice, AlignerState& state) const
	{
		state.counters.recomputeTiles++;
		std::vector<EdgeWithPriority> incoming { EdgeWithPriority { node, 0, slice.startSlice, true } };
		std::vector<WordSlice> result;
		result.reserve(graph.NodeLength(node));
		NodeSliceItem sliceCopy = slice;
		calculateNodeInner(node, sliceCopy, EqV, previousSlice, incoming, nullptr, &result, state);
		ORACLE_ASSERT(result.size() == graph.NodeLength(node));
		ORACLE_ASSERT(result[0] == slice.startSlice);
		ORACLE_ASSERT(result.back() == slice.endSlice);
		return result;
	}

	static NodeSliceItem absentPrevious()   // "not in previous band" stand-in, ...Banded.h:313-321
	{
		NodeSliceItem p;
		p.HP = ~(uint64_t)0;
		p.HN = 0;
		p.exists = false;
		return p;
	}

	// reference: ...Common.h:1170-1229
	void flattenLastSliceEnd(NodeSliceMap& slice, const NodeSliceMap& previousSlice, NodeCalculationResult& sliceCalc, size_t j, std::string_view sequence, AlignerState& state) const
	{
		ORACLE_ASSERT(j < sequence.size());
		ORACLE_ASSERT(sequence.size() - j < 64);
		sliceCalc.minScore = INT32_MAX;
		sliceCalc.minScoreNode = SIZE_MAX;
		sliceCalc.minScoreNodeOffset = SIZE_MAX;
		size_t offset = sequence.size() - j;
		EqVector EqV = getEqVector(sequence, j);
		size_t nodesAtMinimum = 0;
		for (size_t at = 0; at < slice.items.size(); at++) {   // DEFINED order: band-entry order (see header); state.tieOrder == 1 walks it backwards
			const auto& entry = slice.items[state.tieOrder == 1 ? slice.items.size() - 1 - at : at];
			NodeSliceItem old = previousSlice.hasNode(entry.first) ? previousSlice.node(entry.first) : absentPrevious();
			std::vector<WordSlice> cols = recalcNodeWordslice(entry.first, entry.second, EqV, old, state);
			int32_t nodeMinimum = INT32_MAX;
			for (size_t i = 0; i < cols.size(); i++) {
				WordSlice flat = flattenWordSlice(cols[i], offset);
				if (flat.scoreEnd < nodeMinimum) nodeMinimum = flat.scoreEnd;
				if (flat.scoreEnd < sliceCalc.minScore) {
					sliceCalc.minScore = flat.scoreEnd;
					sliceCalc.minScoreNode = entry.first;
					sliceCalc.minScoreNodeOffset = i;
					nodesAtMinimum = 0;
				}
			}
			if (nodeMinimum == sliceCalc.minScore) nodesAtMinimum++;
		}
		ORACLE_ASSERT(sliceCalc.minScore != INT32_MAX);
		state.counters.flattenCalls++;
		sliceCalc.flattenTie = nodesAtMinimum > 1;
	}

	// Node scheduling inside one slice. reference: src/ComponentPriorityQueue.h. Items are ordered by
	// (componentNumber, score of the first edge that activated the node); each node keeps the list of
	// incoming edges ("extras") gathered until it is popped.
	struct QueueItem {
		size_t component; int score; size_t index;
		bool operator>(const QueueItem& o) const { return component > o.component || (component == o.component && score > o.score); }
	};
	struct ComponentQueue {
		std::priority_queue<QueueItem, std::vector<QueueItem>, std::greater<QueueItem>> active;
		std::unordered_map<size_t, std::vector<EdgeWithPriority>> extras;
		std::unordered_map<size_t, bool> isActive;
		void insert(size_t component, int score, const EdgeWithPriority& e)
		{
			if (!isActive[e.target]) { active.push({ component, score, e.target }); isActive[e.target] = true; }
			extras[e.target].push_back(e);
		}
		size_t size() const { return active.size(); }
		size_t topIndex() const { return active.top().index; }
		void pop() { size_t idx = active.top().index; extras[idx].clear(); isActive[idx] = false; active.pop(); }
	};

	// reference: ...Banded.h:205-426 (ComponentPriorityQueue branch, always taken: SURVEY.md §3.3)
	NodeCalculationResult calculateSlice(std::string_view sequence, size_t j, NodeSliceMap& currentSlice, const NodeSliceMap& previousSlice, std::vector<bool>& currentBand, const std::vector<bool>& previousBand, int32_t previousQuitScore, int bandwidth, int32_t previousMinScore, AlignerState& state) const
	{
		NodeCalculationResult result;
		result.minScore = INT32_MAX - bandwidth - 1;
		result.minScoreNode = SIZE_MAX;
		result.minScoreNodeOffset = SIZE_MAX;
		result.cellsProcessed = 0;
		EqVector EqV = getEqVector(sequence, j);
		ORACLE_ASSERT(previousSlice.size() > 0);
		ComponentQueue queue;
		for (const auto& node : previousSlice.items) {
			if (j == 0) {
				ORACLE_ASSERT(node.second.minScore <= previousQuitScore);
			} else {
				ORACLE_ASSERT(node.second.exists);
				if (node.second.minScore > previousQuitScore) continue;
				if (graph.linearizable[node.first]) {
					size_t neighbor = graph.inNeighbors[node.first][0];
					if (previousBand[neighbor] && previousSlice.node(neighbor).endSlice.scoreEnd < previousQuitScore && previousSlice.node(neighbor).minScore < previousQuitScore) continue;
				}
			}
			WordSlice startSlice = getSourceSliceFromScore(node.second.startSlice.scoreEnd);
			queue.insert(graph.componentNumber[node.first], node.second.minScore, EdgeWithPriority { node.first, node.second.minScore - previousMinScore, startSlice, true });
		}
		ORACLE_ASSERT(queue.size() > 0);
		int32_t currentMinScoreAtEndRow = result.minScore;
		while (queue.size() > 0) {
			size_t i = queue.topIndex();
			if (queue.extras[i].empty()) { queue.pop(); continue; }
			if (!currentBand[i]) {
				currentSlice.addNode(i);
				currentBand[i] = true;
			}
			const std::vector<EdgeWithPriority> extras = queue.extras[i];
			NodeSliceItem& thisNode = currentSlice.node(i);
			WordSlice oldEnd = thisNode.endSlice;
			if (!thisNode.exists) oldEnd = WordSlice(0, 0, INT32_MAX);
			NodeSliceItem previousThisNode;
			if (previousBand[i]) {
				previousThisNode = previousSlice.node(i);
				ORACLE_ASSERT(previousThisNode.exists);
			} else {
				previousThisNode = absentPrevious();
			}
			state.counters.dpTiles++;
			NodeCalculationResult nodeCalc = calculateNodeInner(i, thisNode, EqV, previousThisNode, extras, &previousBand, nullptr, state);
			queue.pop();
			ORACLE_ASSERT(nodeCalc.minScore <= previousQuitScore + bandwidth + 64 + 64);
			currentMinScoreAtEndRow = std::min(currentMinScoreAtEndRow, nodeCalc.minScore);
			if (nodeCalc.minScore < thisNode.minScore) thisNode.minScore = nodeCalc.minScore;   // setMinScoreIfSmaller, src/NodeSlice.h:331-335
			WordSlice newEnd = thisNode.endSlice;
			if (newEnd.scoreEnd != oldEnd.scoreEnd || newEnd.VP != oldEnd.VP || newEnd.VN != oldEnd.VN) {
				int32_t newEndMinScore = changedMinScore(newEnd, oldEnd);
				ORACLE_ASSERT(newEndMinScore >= previousMinScore);
				ORACLE_ASSERT(newEndMinScore != INT32_MAX);
				if (newEndMinScore <= currentMinScoreAtEndRow + bandwidth)
					for (size_t neighbor : graph.outNeighbors[i])
						queue.insert(graph.componentNumber[neighbor], newEndMinScore, EdgeWithPriority { neighbor, newEndMinScore - previousMinScore, newEnd, false });
			}
			if (nodeCalc.minScore < result.minScore) {
				result.minScore = nodeCalc.minScore;
				result.minScoreNode = nodeCalc.minScoreNode;
				result.minScoreNodeOffset = nodeCalc.minScoreNodeOffset;
			}
			ORACLE_ASSERT(result.minScore == currentMinScoreAtEndRow);
			result.cellsProcessed += nodeCalc.cellsProcessed;
			ORACLE_ASSERT(nodeCalc.cellsProcessed > 0);
		}
		ORACLE_ASSERT(result.minScoreNode != SIZE_MAX);
		if (j + 64 > sequence.size()) flattenLastSliceEnd(currentSlice, previousSlice, result, j, sequence, state);
		return result;
	}

	typedef std::pair<MatrixPosition, bool> Step;               // (cell, nodeSwitch)
	TraceItem makeItem(MatrixPosition pos, bool nodeSwitch, std::string_view seq) const   // TraceItem ctor, src/GraphAlignerCommon.h:148-153
	{
		return TraceItem { pos, nodeSwitch, pos.seqPos < seq.size() ? seq[pos.seqPos] : '-', graph.NodeSequences(pos.node, pos.nodeOffset) };
	}
	std::vector<MatrixPosition> pickBacktraceInside(size_t verticalOffset, const std::vector<WordSlice>& nodeSlices, MatrixPosition pos, std::string_view sequence) const;
	std::pair<Step, Step> pickBacktraceHorizontalCrossing(const DPSlice& cur, const DPSlice& prev, size_t node, MatrixPosition pos, std::string_view sequence) const;
	std::pair<Step, Step> pickBacktraceVerticalCrossing(const DPSlice& cur, const DPSlice& prev, const std::vector<WordSlice>& nodeScores, size_t node, MatrixPosition pos, std::string_view sequence) const;
	Step pickBacktraceCorner(const DPSlice& cur, const DPSlice& prev, size_t node, std::string_view sequence) const;
	static void checkBacktraceCircularity(const OnewayTrace& result);

public:
	DPSlice initialSlice(int bigraphNodeId, size_t offset) const { return getInitialSliceExactPosition(bigraphNodeId, offset); }   // (for gco_extend)
	DPTable getViterbiSlices(std::string_view sequence, const DPSlice& initialSlice, size_t numSlices, AlignerState& state) const;
	static void removeWronglyAlignedEnd(DPTable& table);
	OnewayTrace getReverseTraceFromTable(std::string_view sequence, const DPTable& slice, MatrixPosition startPos, int32_t startScore, AlignerState& state) const;
};

// reference: src/GraphAlignerBitvectorBanded.h:513-701 with :428-498 (fillDPSlice / pickMethodAndExtendFill)
// folded in. The ramp-bandwidth branch (:608-644) needs rampBandwidth > initialBandwidth and is not built.
inline DPTable BitvectorAligner::getViterbiSlices(std::string_view sequence, const DPSlice& initialSlice, size_t numSlices, AlignerState& state) const
{
	DPTable result;
	result.slices.reserve(numSlices + 1);
	for (const auto& node : initialSlice.scores.items) state.previousBand[node.first] = true;
	DPSlice lastSlice = initialSlice;
	result.slices.push_back(initialSlice);
	ORACLE_ASSERT(lastSlice.correctness.CurrentlyCorrect());
	auto clearBand = [](std::vector<bool>& band, const DPSlice& s) { for (const auto& node : s.scores.items) band[node.first] = false; };
	try {
		for (size_t slice = 0; slice < numSlices; slice++) {
			int bandwidth = (int)initialBandwidth;
			DPSlice newSlice;
			newSlice.j = lastSlice.j + 64;
			newSlice.correctness = lastSlice.correctness;
			ORACLE_ASSERT(lastSlice.minScore < INT32_MAX - (int32_t)lastSlice.bandwidth);
			NodeCalculationResult r = calculateSlice(sequence, newSlice.j, newSlice.scores, lastSlice.scores, state.currentBand, state.previousBand, lastSlice.minScore + (int32_t)lastSlice.bandwidth, bandwidth, lastSlice.minScore, state);
			newSlice.cellsProcessed = r.cellsProcessed;
			newSlice.minScoreNode = r.minScoreNode;
			newSlice.minScoreNodeOffset = r.minScoreNodeOffset;
			newSlice.minScore = r.minScore;
			newSlice.flattenTie = r.flattenTie;
			ORACLE_ASSERT(newSlice.minScore >= lastSlice.minScore);
			newSlice.correctness = newSlice.correctness.NextState(newSlice.minScore - lastSlice.minScore);
			newSlice.bandwidth = bandwidth;
			ORACLE_ASSERT(newSlice.minScore != INT32_MAX);
			ORACLE_ASSERT(newSlice.scores.hasNode(newSlice.minScoreNode));
			ORACLE_ASSERT(newSlice.minScoreNodeOffset < graph.NodeLength(newSlice.minScoreNode));
			if (!newSlice.correctness.CorrectFromCorrect()) {   // :589-607
				clearBand(state.previousBand, lastSlice);
				clearBand(state.currentBand, newSlice);
				break;
			}
			result.slices.push_back(newSlice);
			clearBand(state.previousBand, lastSlice);
			if (slice == numSlices - 1) clearBand(state.currentBand, newSlice);
			else std::swap(state.previousBand, state.currentBand);
			lastSlice = std::move(newSlice);
		}
	} catch (...) {
		state.clear();   // the reference's caller does this after a failed assertion (src/Aligner.cpp:589,698)
		throw;
	}
	return result;
}

inline void BitvectorAligner::removeWronglyAlignedEnd(DPTable& table)   // ...Common.h:1231-1241
{
	if (table.slices.empty()) return;
	bool currentlyCorrect = table.slices.back().correctness.CurrentlyCorrect();
	while (!currentlyCorrect) {
		currentlyCorrect = table.slices.back().correctness.FalseFromCorrect();
		table.slices.pop_back();
		if (table.slices.empty()) break;
	}
}

// reference: ...Common.h:392-544. Walks from the start cell to row -1, recomputing a node's columns
// whenever (slice, node) changes.
inline OnewayTrace BitvectorAligner::getReverseTraceFromTable(std::string_view sequence, const DPTable& table, MatrixPosition startPos, int32_t startScore, AlignerState& state) const
{
	const std::vector<DPSlice>& slices = table.slices;
	ORACLE_ASSERT(slices.size() > 0);
	OnewayTrace result;
	result.score = startScore;
	result.trace.push_back(makeItem(startPos, false, sequence));
	size_t currentNode = SIZE_MAX;
	size_t currentSlice = slices.size();
	std::vector<WordSlice> nodeSlices;
	EqVector EqV = getEqVector(sequence, 0);
	while (result.trace.back().DPposition.seqPos != (size_t)-1) {
		MatrixPosition here = result.trace.back().DPposition;
		size_t newSlice = here.seqPos / 64 + 1;
		ORACLE_ASSERT(newSlice < slices.size());
		size_t newNode = here.node;
		if (newSlice != currentSlice || newNode != currentNode) {
			if (newSlice != currentSlice) EqV = getEqVector(sequence, slices[newSlice].j);
			currentSlice = newSlice;
			currentNode = newNode;
			ORACLE_ASSERT(slices[currentSlice].scores.hasNode(currentNode));
			NodeSliceItem previous = slices[currentSlice - 1].scores.hasNode(currentNode) ? slices[currentSlice - 1].scores.node(currentNode) : absentPrevious();
			if (!slices[currentSlice - 1].scores.hasNode(currentNode)) previous.exists = false;
			nodeSlices = recalcNodeWordslice(currentNode, slices[currentSlice].scores.node(currentNode), EqV, previous, state);
		}
		ORACLE_ASSERT(here.nodeOffset < graph.NodeLength(currentNode));
		const DPSlice& cur = slices[currentSlice];
		const DPSlice& prev = slices[currentSlice - 1];
		if (here.seqPos % 64 == 0 && here.nodeOffset == 0) {
			Step bt = pickBacktraceCorner(cur, prev, currentNode, sequence);
			result.trace.push_back(makeItem(bt.first, bt.second, sequence));
			checkBacktraceCircularity(result);
			continue;
		}
		if (here.seqPos % 64 == 0) {
			if (!prev.scores.hasNode(currentNode)) {
				result.trace.push_back(makeItem(MatrixPosition { currentNode, 0, here.seqPos }, false, sequence));
				continue;
			}
			auto crossing = pickBacktraceVerticalCrossing(cur, prev, nodeSlices, currentNode, here, sequence);
			ORACLE_ASSERT(crossing.first.first.nodeOffset <= here.nodeOffset);
			if (crossing.first.first.nodeOffset != here.nodeOffset)
				for (size_t off = here.nodeOffset - 1; off != crossing.first.first.nodeOffset; off--)
					result.trace.push_back(makeItem(MatrixPosition { crossing.first.first.node, off, crossing.first.first.seqPos }, false, sequence));
			if (crossing.first.first != result.trace.back().DPposition) result.trace.push_back(makeItem(crossing.first.first, crossing.first.second, sequence));
			ORACLE_ASSERT(crossing.second.first != result.trace.back().DPposition);
			result.trace.push_back(makeItem(crossing.second.first, crossing.second.second, sequence));
			continue;
		}
		if (here.nodeOffset == 0) {
			auto crossing = pickBacktraceHorizontalCrossing(cur, prev, currentNode, here, sequence);
			ORACLE_ASSERT(crossing.first.first.seqPos <= here.seqPos);
			if (crossing.first.first.seqPos != here.seqPos)
				for (size_t sp = here.seqPos - 1; sp != crossing.first.first.seqPos; sp--)
					result.trace.push_back(makeItem(MatrixPosition { crossing.first.first.node, crossing.first.first.nodeOffset, sp }, false, sequence));
			if (crossing.first.first != result.trace.back().DPposition) result.trace.push_back(makeItem(crossing.first.first, crossing.first.second, sequence));
			ORACLE_ASSERT(crossing.second.first != result.trace.back().DPposition);
			result.trace.push_back(makeItem(crossing.second.first, crossing.second.second, sequence));
			checkBacktraceCircularity(result);
			continue;
		}
		for (const MatrixPosition& p : pickBacktraceInside(cur.j, nodeSlices, here, sequence))
			result.trace.push_back(makeItem(p, false, sequence));
	}
	// row -1: walk left along the seed node while the initial ramp decreases (:508-542)
	{
		MatrixPosition here = result.trace.back().DPposition;
		ORACLE_ASSERT(slices[0].scores.hasNode(here.node));
		const NodeSliceItem& node = slices[0].scores.node(here.node);
		std::vector<int32_t> before(graph.NodeLength(here.node));
		before[0] = node.startSlice.scoreEnd;
		for (size_t i = 1; i < before.size(); i++) before[i] = before[i - 1] + (int32_t)((node.HP >> i) & 1) - (int32_t)((node.HN >> i) & 1);
		ORACLE_ASSERT(before.back() == node.endSlice.scoreEnd);
		while (true) {
			MatrixPosition p = result.trace.back().DPposition;
			if (!(before[p.nodeOffset] != 0 && p.nodeOffset > 0 && before[p.nodeOffset - 1] == before[p.nodeOffset] - 1)) break;
			result.trace.push_back(makeItem(MatrixPosition { p.node, p.nodeOffset - 1, p.seqPos }, false, sequence));
		}
		MatrixPosition p = result.trace.back().DPposition;
		if (p.nodeOffset == 0 && before[0] != 0) {
			for (size_t neighbor : graph.inNeighbors[p.node]) {
				if (slices[0].scores.hasNode(neighbor) && slices[0].scores.node(neighbor).endSlice.getScoreBeforeStart() == before[0] - 1) {
					result.trace.push_back(makeItem(MatrixPosition { neighbor, graph.NodeLength(neighbor) - 1, p.seqPos }, true, sequence));
					break;
				}
			}
		}
	}
	state.counters.traceItems += result.trace.size();
	return result;
}

inline void BitvectorAligner::checkBacktraceCircularity(const OnewayTrace& result)   // ...Common.h:546-554
{
	for (size_t i = result.trace.size() - 2; i < result.trace.size(); i--) {
		ORACLE_ASSERT(result.trace[i].DPposition != result.trace.back().DPposition);
		if (result.trace[i].DPposition.seqPos != result.trace.back().DPposition.seqPos) return;
	}
}

// reference: ...Common.h:556-597. Preference: vertical, then diagonal, then horizontal.
inline std::vector<MatrixPosition> BitvectorAligner::pickBacktraceInside(size_t verticalOffset, const std::vector<WordSlice>& nodeSlices, MatrixPosition pos, std::string_view sequence) const
{
	size_t hori = pos.nodeOffset;
	size_t vert = pos.seqPos - verticalOffset;
	ORACLE_ASSERT(vert < 64 && hori < nodeSlices.size());
	std::vector<MatrixPosition> result;
	while (hori > 0 && vert > 0) {
		int32_t scoreHere = nodeSlices[hori].getValue((int)vert);
		int32_t verticalScore = nodeSlices[hori].getValue((int)vert - 1);
		int32_t horizontalScore = nodeSlices[hori - 1].getValue((int)vert);
		int32_t diagonalScore = nodeSlices[hori - 1].getValue((int)vert - 1);
		bool eq = characterMatch(sequence[vert + verticalOffset], graph.NodeSequences(pos.node, hori));
		ORACLE_ASSERT(verticalScore >= scoreHere - 1);
		ORACLE_ASSERT(horizontalScore >= scoreHere - 1);
		ORACLE_ASSERT(diagonalScore >= scoreHere - (eq ? 0 : 1));
		if (verticalScore == scoreHere - 1) { vert--; result.push_back({ pos.node, hori, vert + verticalOffset }); continue; }
		if (diagonalScore == scoreHere - (eq ? 0 : 1)) { hori--; vert--; result.push_back({ pos.node, hori, vert + verticalOffset }); continue; }
		ORACLE_ASSERT(horizontalScore == scoreHere - 1);
		hori--;
		result.push_back({ pos.node, hori, vert + verticalOffset });
	}
	return result;
}

// reference: ...Common.h:599-663 (scoresNotValid is always false here: maxCellsPerSlice is unlimited)
inline std::pair<BitvectorAligner::Step, BitvectorAligner::Step> BitvectorAligner::pickBacktraceHorizontalCrossing(const DPSlice& curSlice, const DPSlice& prevSlice, size_t node, MatrixPosition pos, std::string_view sequence) const
{
	const NodeSliceMap& current = curSlice.scores;
	int32_t quitScore = curSlice.minScore + (int32_t)curSlice.bandwidth;
	ORACLE_ASSERT(current.hasNode(node));
	WordSlice startSlice = current.node(node).startSlice;
	while (pos.seqPos % 64 != 0 && (startSlice.VP & ((uint64_t)1 << (pos.seqPos % 64)))) pos.seqPos--;
	size_t offset = pos.seqPos % 64;
	if (offset == 0) return { { pos, false }, pickBacktraceCorner(curSlice, prevSlice, node, sequence) };
	bool eq = characterMatch(sequence[pos.seqPos], graph.NodeSequences(pos.node, pos.nodeOffset));
	int32_t scoreHere = startSlice.getValue((int)offset);
	if (scoreHere > quitScore) {
		int32_t smallestFound = startSlice.getValue((int)offset - 1);
		MatrixPosition smallestPos { node, 0, pos.seqPos - 1 };
		bool nodeChange = false;
		for (size_t neighbor : graph.inNeighbors[node]) {
			if (!current.hasNode(neighbor)) continue;
			WordSlice neighborSlice = current.node(neighbor).endSlice;
			if (neighborSlice.getValue((int)offset - 1) <= smallestFound) {
				smallestFound = neighborSlice.getValue((int)offset - 1);
				smallestPos = MatrixPosition { neighbor, graph.NodeLength(neighbor) - 1, pos.seqPos - 1 };
				nodeChange = true;
			}
			if (neighborSlice.getValue((int)offset) < smallestFound && neighbor != node) {
				smallestFound = neighborSlice.getValue((int)offset);
				smallestPos = MatrixPosition { neighbor, graph.NodeLength(neighbor) - 1, pos.seqPos };
				nodeChange = true;
			}
		}
		ORACLE_ASSERT(smallestPos != pos);
		return { { pos, false }, { smallestPos, nodeChange } };
	}
	for (size_t neighbor : graph.inNeighbors[node]) {
		if (!current.hasNode(neighbor)) continue;
		WordSlice neighborSlice = current.node(neighbor).endSlice;
		ORACLE_ASSERT(neighborSlice.getValue((int)offset) >= scoreHere - 1);
		ORACLE_ASSERT(neighborSlice.getValue((int)offset - 1) >= scoreHere - (eq ? 0 : 1));
		if (neighborSlice.getValue((int)offset) == scoreHere - 1) return { { pos, false }, { MatrixPosition { neighbor, graph.NodeLength(neighbor) - 1, pos.seqPos }, true } };
		if (neighborSlice.getValue((int)offset - 1) == scoreHere - (eq ? 0 : 1)) return { { pos, false }, { MatrixPosition { neighbor, graph.NodeLength(neighbor) - 1, pos.seqPos - 1 }, true } };
	}
	throw AssertionFailure("pickBacktraceHorizontalCrossing: no predecessor");
}

// reference: ...Common.h:665-708
inline std::pair<BitvectorAligner::Step, BitvectorAligner::Step> BitvectorAligner::pickBacktraceVerticalCrossing(const DPSlice& curSlice, const DPSlice& prevSlice, const std::vector<WordSlice>& nodeScores, size_t node, MatrixPosition pos, std::string_view sequence) const
{
	int32_t quitScore = curSlice.minScore + (int32_t)curSlice.bandwidth;
	int32_t previousQuitScore = prevSlice.minScore + (int32_t)prevSlice.bandwidth;
	ORACLE_ASSERT(pos.nodeOffset > 0 && pos.nodeOffset < nodeScores.size());
	while (pos.nodeOffset > 0 && nodeScores[pos.nodeOffset - 1].getValue(0) == nodeScores[pos.nodeOffset].getValue(0) - 1) pos.nodeOffset--;
	if (pos.nodeOffset == 0) return { { pos, false }, pickBacktraceCorner(curSlice, prevSlice, node, sequence) };
	ORACLE_ASSERT(prevSlice.scores.hasNode(node));
	bool eq = characterMatch(sequence[pos.seqPos], graph.NodeSequences(pos.node, pos.nodeOffset));
	const NodeSliceItem& previousNode = prevSlice.scores.node(node);
	int32_t scoreHere = nodeScores[pos.nodeOffset].getValue(0);
	int32_t scoreDiagonal = previousNode.startSlice.scoreEnd;
	for (size_t i = 1; i <= pos.nodeOffset - 1; i++) scoreDiagonal += (int32_t)((previousNode.HP >> i) & 1) - (int32_t)((previousNode.HN >> i) & 1);
	int32_t scoreUp = scoreDiagonal + (int32_t)((previousNode.HP >> pos.nodeOffset) & 1) - (int32_t)((previousNode.HN >> pos.nodeOffset) & 1);
	if (scoreHere > quitScore || scoreDiagonal > previousQuitScore || scoreUp > previousQuitScore) {
		if (scoreDiagonal < scoreUp) return { { pos, false }, { MatrixPosition { pos.node, pos.nodeOffset - 1, pos.seqPos - 1 }, false } };
		return { { pos, false }, { MatrixPosition { pos.node, pos.nodeOffset, pos.seqPos - 1 }, false } };
	}
	ORACLE_ASSERT(scoreUp >= scoreHere - 1);
	ORACLE_ASSERT(scoreDiagonal >= scoreHere - (eq ? 0 : 1));
	if (scoreUp == scoreHere - 1) return { { pos, false }, { MatrixPosition { pos.node, pos.nodeOffset, pos.seqPos - 1 }, false } };
	ORACLE_ASSERT(scoreDiagonal == scoreHere - (eq ? 0 : 1));
	return { { pos, false }, { MatrixPosition { pos.node, pos.nodeOffset - 1, pos.seqPos - 1 }, false } };
}

// reference: ...Common.h:710-804. In-neighbours are tried in adjacency order; first match wins.
inline BitvectorAligner::Step BitvectorAligner::pickBacktraceCorner(const DPSlice& curSlice, const DPSlice& prevSlice, size_t node, std::string_view sequence) const
{
	const NodeSliceMap& current = curSlice.scores;
	const NodeSliceMap& previous = prevSlice.scores;
	size_t j = curSlice.j;
	int32_t quitScore = curSlice.minScore + (int32_t)curSlice.bandwidth;
	int32_t previousQuitScore = prevSlice.minScore + (int32_t)prevSlice.bandwidth;
	int32_t scoreHere = current.node(node).startSlice.getValue(0);
	if (scoreHere > quitScore) {
		int32_t smallestFound = scoreHere + 1;
		MatrixPosition smallestPos { 0, 0, 0 };
		bool nodeChange = false;
		if (previous.hasNode(node)) {
			smallestFound = previous.node(node).startSlice.scoreEnd;
			smallestPos = MatrixPosition { node, 0, j - 1 };
		}
		for (size_t neighbor : graph.inNeighbors[node]) {
			if (previous.hasNode(neighbor)) {
				WordSlice neighborSlice = previous.node(neighbor).endSlice;
				if (neighborSlice.scoreEnd <= smallestFound) {
					smallestFound = neighborSlice.scoreEnd;
					smallestPos = MatrixPosition { neighbor, graph.NodeLength(neighbor) - 1, j - 1 };
					nodeChange = true;
				}
			}
			if (current.hasNode(neighbor) && neighbor != node) {
				WordSlice neighborSlice = current.node(neighbor).endSlice;
				if (neighborSlice.getValue(0) < smallestFound) {
					smallestFound = neighborSlice.getValue(0);
					smallestPos = MatrixPosition { neighbor, graph.NodeLength(neighbor) - 1, j };
					nodeChange = true;
				}
			}
		}
		return { smallestPos, nodeChange };
	}
	bool eq = characterMatch(sequence[j], graph.NodeSequences(node, 0));
	if (previous.hasNode(node)) {
		ORACLE_ASSERT(previous.node(node).startSlice.scoreEnd >= scoreHere - 1);
		if (previous.node(node).startSlice.scoreEnd == scoreHere - 1) return { MatrixPosition { node, 0, j - 1 }, false };
	}
	MatrixPosition bestInvalid { (size_t)-1, (size_t)-1, (size_t)-1 };
	int32_t bestInvalidScore = scoreHere + 1;
	for (size_t neighbor : graph.inNeighbors[node]) {
		if (current.hasNode(neighbor)) {
			ORACLE_ASSERT(current.node(neighbor).endSlice.getValue(0) >= scoreHere - 1);
			if (current.node(neighbor).endSlice.getValue(0) == scoreHere - 1) return { MatrixPosition { neighbor, graph.NodeLength(neighbor) - 1, j }, true };
		}
		if (previous.hasNode(neighbor)) {
			int32_t cornerScore = previous.node(neighbor).endSlice.scoreEnd;
			if (cornerScore > previousQuitScore) {
				if (cornerScore < bestInvalidScore) {
					bestInvalidScore = cornerScore;
					bestInvalid = MatrixPosition { neighbor, graph.NodeLength(neighbor) - 1, j - 1 };
				}
			} else {
				ORACLE_ASSERT(cornerScore >= scoreHere - (eq ? 0 : 1));
				if (cornerScore == scoreHere - (eq ? 0 : 1)) return { MatrixPosition { neighbor, graph.NodeLength(neighbor) - 1, j - 1 }, true };
			}
		}
	}
	if (bestInvalidScore < scoreHere + 1) return { bestInvalid, true };
	throw AssertionFailure("pickBacktraceCorner: no predecessor");
}

} // namespace oracle
