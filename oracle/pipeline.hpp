// ORACLE (test infrastructure, not product code): CPU restatement of the reference's per-read hot path:
// minimizer seeding -> seed ordering -> seed extension (whole read and 35 bp fragments) -> anchors ->
// co-linear chaining -> chain stitching.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// Follows: src/MinimizerSeeder.cpp:59-102,494-555 (getSeeds), src/GraphAligner.h:114-203,233-295,
// 407-461,480-626 (AlignOneWay, orderSeedsByChaining, exactAlignmentPart, traces),
// src/AlignmentGraph.cpp:1600-1863 (chaining), src/Aligner.cpp:601-922 (per-read logic).
//
// PARITY: partially pinned. WordSlice, the correctness HMM and edit distance are checked against the
// reference's own sources compiled unmodified (oracle/_ref). The rest of the reference cannot be built
// here (needs parallel-hashmap, protobuf-generated vg.pb.h, sdsl, BBHash, concurrentqueue - all absent
// and stand-ins are not allowed), and the reference ships no expected outputs for test/graph.gfa, so the
// graph walk, seeding and chaining are PARITY UNPINNED: they are a line-by-line restatement, cross-checked
// only by independent brute-force models in tests/.
#pragma once
#include "bitvector_aligner.hpp"
#include "edlib_path.hpp"
#include <algorithm>
#include <map>
#include <memory>
#include <unordered_set>

namespace oracle {

struct SeedHit {   // reference: src/GraphAlignerWrapper.h:11-37
	int nodeID;                 // GFA-order id (bigraph id / 2)
	size_t nodeOffset;          // offset in the original node
	size_t seqPos;              // read position of the LAST base of the k-mer
	size_t matchLen;
	bool reverse;
	size_t alignmentGraphNodeId;      // split node
	size_t alignmentGraphNodeOffset;  // offset in the split node
	size_t rawSeedGoodness;
	size_t seedGoodness = 0;
	size_t seedClusterSize = 0;
};

struct AlignmentItem {   // reference: src/GraphAlignerCommon.h:306-350
	std::shared_ptr<OnewayTrace> trace;
	size_t seedGoodness = 0;
	size_t alignmentStart = 0, alignmentEnd = 0;
	size_t alignmentScore = SIZE_MAX;
	bool alignmentFailed() const { return alignmentEnd == alignmentStart; }
};
struct AlignmentResult {
	std::vector<AlignmentItem> alignments;
	size_t seedsExtended = 0;
};

struct Params {   // defaults: src/AlignerMain.cpp:186-209
	size_t k = 15, w = 20;
	double seedDensity = 10;
	double discardMostNumerousFraction = 0.001;
	size_t bandwidth = 10;
	size_t minClusterSize = 1;
	size_t splitLen = 35, splitGap = 35;
	long long colinearGap = 10000;
	bool longPass = true;       // run the whole-read GraphAligner pass (src/Aligner.cpp:630-654)
	double eCutoff = -1;        // --E-cutoff (src/AlignerMain.cpp:159,271-274): -1 = keep every alignment
};

// ------------------------------------------------------------------ seeding (K1)

// reference: src/MinimizerSeeder.cpp:59-102
template <typename F>
inline void iterateKmers(const std::string& str, size_t k, size_t w, F&& callback)
{
	const size_t realWindow = w - k + 1;
	if (str.size() < k) return;
	const uint64_t mask = ~(~(uint64_t)0 << (k * 2));
	auto code = [](char c) { switch (c) { case 'a': case 'A': return 0; case 'c': case 'C': return 1; case 'g': case 'G': return 2; case 't': case 'T': return 3; } return -1; };
	size_t offset = 0;
	while (true) {
	restart:
		while (offset < str.size() && code(str[offset]) < 0) offset++;
		if (offset + k > str.size()) return;
		uint64_t kmer = 0;
		for (size_t i = 0; i < k; i++) {
			int c = code(str[offset + i]);
			if (c < 0) { offset += i; goto restart; }
			kmer = (kmer << 2) | (uint64_t)c;
		}
		callback(offset + k - 1, kmer);
		uint64_t lastKmer = kmer;
		size_t lastPos = offset + k - 1;
		for (size_t i = k; offset + i < str.size(); i++) {
			int c = code(str[offset + i]);
			if (c < 0) { offset += i; goto restart; }
			kmer = ((kmer << 2) & mask) | (uint64_t)c;
			if (lastKmer != kmer || lastPos <= offset + i - realWindow) {
				callback(offset + i, kmer);
				lastKmer = kmer;
				lastPos = offset + i;
			}
		}
		return;
	}
}

// reference: src/MinimizerSeeder.cpp:522-555 (+addMinimizers :494-520)
inline std::vector<SeedHit> getSeeds(const AlignmentGraph& graph, const gc::MinimizerIndex& index, const std::string& sequence, double density)
{
	struct Match { size_t pos, start, count; };
	std::vector<Match> matches;
	iterateKmers(sequence, index.k, index.w, [&](size_t pos, uint64_t kmer) {
		auto it = std::lower_bound(index.kmers.begin(), index.kmers.end(), kmer);
		if (it == index.kmers.end() || *it != kmer) return;
		size_t i = it - index.kmers.begin();
		size_t start = index.startPos[i], count = index.startPos[i + 1] - start;
		if (count >= index.maxCount) return;
		matches.push_back({ pos, start, count });
	});
	std::vector<SeedHit> result;
	size_t maxHits = (size_t)(sequence.size() * density);
	if (density == -1) maxHits = SIZE_MAX;
	std::sort(matches.begin(), matches.end(), [](const Match& l, const Match& r) { return l.count < r.count; });   // same std::sort, same input => same order
	size_t seedsHere = 0, allowedCount = 0;
	for (const Match& m : matches) {
		if (seedsHere >= maxHits && m.count > allowedCount) break;
		allowedCount = m.count;
		for (size_t i = m.start; i < m.start + m.count; i++) {
			size_t split = index.positions[i] >> 6, off = index.positions[i] & 63;
			SeedHit s;
			s.nodeID = graph.nodeIDs[split] / 2;
			s.nodeOffset = off + graph.nodeOffset[split];
			s.seqPos = m.pos;
			s.matchLen = index.k;
			s.rawSeedGoodness = index.maxCount - m.count;
			s.reverse = graph.reverse[split];
			s.alignmentGraphNodeId = split;
			s.alignmentGraphNodeOffset = off;
			result.push_back(s);
		}
		seedsHere += m.count;
	}
	return result;
}

// ------------------------------------------------------------------ seed ordering (K2)

// reference: src/GraphAligner.h:233-295
inline void orderSeedsByChaining(const AlignmentGraph& graph, std::vector<SeedHit>& seedHits)
{
	std::map<size_t, std::vector<std::pair<size_t, size_t>>> seedPoses;   // per-chain work is independent => map order is immaterial
	for (size_t i = 0; i < seedHits.size(); i++) {
		size_t nodeIndex = seedHits[i].alignmentGraphNodeId, realOffset = seedHits[i].alignmentGraphNodeOffset;
		ORACLE_ASSERT(graph.chainApproxPos[nodeIndex] + realOffset >= seedHits[i].seqPos);
		seedPoses[graph.chainNumber[nodeIndex]].emplace_back(i, graph.chainApproxPos[nodeIndex] + realOffset - seedHits[i].seqPos);
	}
	for (auto& pair : seedPoses) {
		auto& v = pair.second;
		std::sort(v.begin(), v.end(), [](std::pair<size_t, size_t> l, std::pair<size_t, size_t> r) { return l.second < r.second; });
		size_t clusterStart = 0;
		for (size_t i = 1; i <= v.size(); i++) {
			if (i < v.size() && v[i].second <= v[i - 1].second + 100) continue;
			std::sort(v.begin() + clusterStart, v.begin() + i, [&seedHits](std::pair<size_t, size_t> l, std::pair<size_t, size_t> r) { return seedHits[l.first].seqPos < seedHits[r.first].seqPos; });
			size_t matchingBps = 0;
			int lastEnd = INT_MIN;
			for (size_t j = clusterStart; j < i; j++) {
				int thisStart = (int)seedHits[v[j].first].seqPos - (int)seedHits[v[j].first].matchLen + 1;
				int thisEnd = (int)seedHits[v[j].first].seqPos;
				ORACLE_ASSERT(thisEnd >= lastEnd);
				ORACLE_ASSERT(thisEnd > thisStart);
				matchingBps += (thisEnd - std::max(thisStart, lastEnd));
				lastEnd = thisEnd;
			}
			for (size_t j = clusterStart; j < i; j++) {
				seedHits[v[j].first].seedGoodness = matchingBps + seedHits[v[j].first].rawSeedGoodness;
				seedHits[v[j].first].seedClusterSize = i - clusterStart;
			}
			clusterStart = i;
		}
	}
	std::sort(seedHits.begin(), seedHits.end(), [](const SeedHit& l, const SeedHit& r) { return l.seedGoodness < r.seedGoodness; });
	std::reverse(seedHits.begin(), seedHits.end());
}

// ------------------------------------------------------------------ seed extension driver (K3)

class GraphAligner {
public:
	GraphAligner(const AlignmentGraph& graph, const Params& params, bool sloppy) : graph(graph), params(params), sloppyOptimizations(sloppy), bv(graph, params.bandwidth) {}

	// reference: src/GraphAligner.h:114-203 with seedExtendDensity == -1 and nondeterministicOptimizations off
	AlignmentResult AlignOneWay(const std::string& sequence, const std::vector<SeedHit>& seedHits, AlignerState& state, size_t l, size_t r, size_t offset) const
	{
		AlignmentResult result;
		ORACLE_ASSERT(seedHits.size() > 0);
		size_t seedScoreForEndToEndAln = 0;
		size_t extendSeeds = seedHits.size();
		size_t worstExtendedSeedScore = 0;
		std::string revSequence = gc::ReverseComplement(sequence);
		for (size_t i = l; i < seedHits.size() && i < r; i++) {
			if (sloppyOptimizations && seedHits[i].seedGoodness < seedScoreForEndToEndAln) break;
			if (result.seedsExtended >= extendSeeds && seedHits[i].seedGoodness < worstExtendedSeedScore) break;
			SeedHit seed = seedHits[i];
			seed.seqPos -= offset;
			if (seed.seedClusterSize < params.minClusterSize) continue;
			if (sloppyOptimizations) {
				bool found = false;
				for (const auto& aln : result.alignments)
					if (aln.alignmentStart <= seed.seqPos && aln.alignmentEnd >= seed.seqPos && aln.seedGoodness > seed.seedGoodness) { found = true; break; }
				if (found) continue;
			}
			bool found = false;
			for (const auto& aln : result.alignments)
				if (exactAlignmentPart(aln, seed)) { found = true; break; }
			if (found) continue;
			worstExtendedSeedScore = seed.seedGoodness;
			result.seedsExtended += 1;
			AlignmentItem item = getAlignmentFromSeed(sequence, revSequence, seed, state);
			if (item.alignmentFailed()) continue;
			item.seedGoodness = seed.seedGoodness;
			result.alignments.push_back(std::move(item));
			if (sloppyOptimizations) {
				std::sort(result.alignments.begin(), result.alignments.end(), [](const AlignmentItem& l, const AlignmentItem& r) { return l.alignmentStart < r.alignmentStart; });
				if (result.alignments[0].alignmentStart == 0) {
					size_t minSeedGoodness = result.alignments[0].seedGoodness;
					size_t contiguousEnd = result.alignments[0].alignmentEnd;
					for (size_t a = 1; a < result.alignments.size(); a++)
						if (result.alignments[a].alignmentStart <= contiguousEnd) {
							minSeedGoodness = std::min(minSeedGoodness, result.alignments[a].seedGoodness);
							contiguousEnd = std::max(contiguousEnd, result.alignments[a].alignmentEnd);
						}
					if (contiguousEnd == sequence.size()) seedScoreForEndToEndAln = minSeedGoodness;
				}
			}
		}
		return result;
	}

private:
	const AlignmentGraph& graph;
	const Params& params;
	bool sloppyOptimizations;
	BitvectorAligner bv;

	// reference: src/GraphAligner.h:407-461. True if the seed's cell lies on the alignment's trace.
	bool exactAlignmentPart(const AlignmentItem& aln, const SeedHit& seedHit) const
	{
		const std::vector<TraceItem>& trace = aln.trace->trace;
		ORACLE_ASSERT(trace.size() > 0);
		ORACLE_ASSERT(trace.back().DPposition.seqPos > trace[0].DPposition.seqPos);
		if (trace.back().DPposition.seqPos < seedHit.seqPos) return false;
		if (trace[0].DPposition.seqPos > seedHit.seqPos) return false;
		size_t high = trace.size(), low = 0;
		size_t mid = (seedHit.seqPos - trace[0].DPposition.seqPos) / (trace.back().DPposition.seqPos - trace[0].DPposition.seqPos);
		while (trace[mid].DPposition.seqPos != seedHit.seqPos) {
			if (trace[mid].DPposition.seqPos < seedHit.seqPos) {
				low = mid;
				mid = (high + low) / 2;
				if (mid == low) mid += 1;
				ORACLE_ASSERT(mid < trace.size());
			}
			if (trace[mid].DPposition.seqPos > seedHit.seqPos) {
				high = mid;
				mid = (high + low) / 2;
				ORACLE_ASSERT(mid < trace.size());
			}
			ORACLE_ASSERT(low < mid);
			ORACLE_ASSERT(mid < high);
		}
		size_t compareNode = (size_t)seedHit.nodeID * 2 + (seedHit.reverse ? 1 : 0);
		for (size_t down = mid; trace[down].DPposition.seqPos == seedHit.seqPos;) {
			if (compareNode == trace[down].DPposition.node && seedHit.nodeOffset == trace[down].DPposition.nodeOffset) return true;
			if (down == 0) break;
			down -= 1;
		}
		for (size_t up = mid; trace[up].DPposition.seqPos == seedHit.seqPos;) {
			if (compareNode == trace[up].DPposition.node && seedHit.nodeOffset == trace[up].DPposition.nodeOffset) return true;
			up += 1;
			if (up == trace.size()) break;
		}
		return false;
	}

	// reference: src/GraphAligner.h:527-540. Split-node coordinates -> (bigraph node id, offset in original node).
	void fixForwardTraceSeqPos(std::vector<TraceItem>& trace, size_t start, const std::string& sequence) const
	{
		if (trace.empty()) return;
		for (size_t i = 0; i < trace.size(); i++) {
			trace[i].DPposition.seqPos += start;
			size_t nodeIndex = trace[i].DPposition.node;
			trace[i].DPposition.node = graph.nodeIDs[nodeIndex];
			trace[i].DPposition.nodeOffset += graph.nodeOffset[nodeIndex];
			ORACLE_ASSERT(trace[i].DPposition.seqPos < sequence.size());
			ORACLE_ASSERT(i == 0 || trace[i].DPposition.seqPos == trace[0].DPposition.seqPos || trace[i].sequenceCharacter == sequence[trace[i].DPposition.seqPos]);
		}
		trace[0].sequenceCharacter = sequence[trace[0].DPposition.seqPos];
	}

	// reference: src/GraphAligner.h:543-565
	void fixReverseTraceSeqPosAndOrder(std::vector<TraceItem>& trace, size_t end, const std::string& sequence) const
	{
		if (trace.empty()) return;
		std::reverse(trace.begin(), trace.end());
		for (size_t i = 0; i < trace.size(); i++) {
			ORACLE_ASSERT(trace[i].DPposition.seqPos <= end || trace[i].DPposition.seqPos == (size_t)-1);
			trace[i].DPposition.seqPos = end - trace[i].DPposition.seqPos;
			size_t offset = graph.nodeOffset[trace[i].DPposition.node] + trace[i].DPposition.nodeOffset;
			auto reversePos = graph.GetReversePosition(graph.nodeIDs[trace[i].DPposition.node], offset);
			trace[i].DPposition.node = reversePos.first;
			trace[i].DPposition.nodeOffset = reversePos.second;
			ORACLE_ASSERT(trace[i].DPposition.seqPos < sequence.size());
			trace[i].sequenceCharacter = sequence[trace[i].DPposition.seqPos];
			trace[i].graphCharacter = gc::Complement(trace[i].graphCharacter);
		}
		for (size_t i = 0; i + 1 < trace.size(); i++) trace[i].nodeSwitch = trace[i + 1].nodeSwitch;
		trace.back().nodeSwitch = false;
	}

	// reference: src/GraphAligner.h:480-525 + :567-626
	AlignmentItem getAlignmentFromSeed(const std::string& sequence, const std::string& revSequence, const SeedHit& seedHit, AlignerState& state) const
	{
		ORACLE_ASSERT(seedHit.seqPos < sequence.size());
		int forwardNodeId = seedHit.nodeID * 2 + (seedHit.reverse ? 1 : 0);
		int backwardNodeId = forwardNodeId ^ 1;
		OnewayTrace backward = OnewayTrace::TraceFailed(), forward = OnewayTrace::TraceFailed();
		if (seedHit.seqPos > 0) {
			std::string_view backwardPart(revSequence.data() + revSequence.size() - seedHit.seqPos, seedHit.seqPos);
			auto reversePos = graph.GetReversePosition(forwardNodeId, seedHit.nodeOffset);
			backward = bv.getReverseTraceFromSeed(backwardPart, backwardNodeId, reversePos.second, state);
		}
		if (seedHit.seqPos < sequence.size() - 1) {
			std::string_view forwardPart(sequence.data() + seedHit.seqPos + 1, sequence.size() - seedHit.seqPos - 1);
			forward = bv.getReverseTraceFromSeed(forwardPart, forwardNodeId, seedHit.nodeOffset, state);
		}
		if (!backward.failed()) {
			const MatrixPosition& p = backward.trace.back().DPposition;
			auto reversePos = graph.GetReversePosition(forwardNodeId, seedHit.nodeOffset);
			ORACLE_ASSERT(p.seqPos == (size_t)-1 && graph.nodeIDs[p.node] == backwardNodeId && graph.nodeOffset[p.node] + p.nodeOffset == reversePos.second);
			std::reverse(backward.trace.begin(), backward.trace.end());
		}
		if (!forward.failed()) {
			const MatrixPosition& p = forward.trace.back().DPposition;
			ORACLE_ASSERT(p.seqPos == (size_t)-1 && graph.nodeIDs[p.node] == forwardNodeId && graph.nodeOffset[p.node] + p.nodeOffset == seedHit.nodeOffset);
			std::reverse(forward.trace.begin(), forward.trace.end());
		}
		fixReverseTraceSeqPosAndOrder(backward.trace, seedHit.seqPos - 1, sequence);
		fixForwardTraceSeqPos(forward.trace, seedHit.seqPos + 1, sequence);
		if (forward.failed() && backward.failed()) return AlignmentItem();
		OnewayTrace merged = std::move(backward);
		if (merged.failed()) {
			merged = std::move(forward);
		} else if (!forward.failed()) {
			ORACLE_ASSERT(merged.trace.size() > 0);
			ORACLE_ASSERT(merged.trace.back().DPposition == forward.trace[0].DPposition);
			merged.trace.pop_back();
			merged.trace.insert(merged.trace.end(), forward.trace.begin(), forward.trace.end());
			merged.score += forward.score;
		}
		AlignmentItem result;
		result.trace = std::make_shared<OnewayTrace>(std::move(merged));
		ORACLE_ASSERT(result.trace->trace.size() > 0);
		size_t seqstart = result.trace->trace[0].DPposition.seqPos;
		size_t seqend = result.trace->trace.back().DPposition.seqPos;
		ORACLE_ASSERT(seqend < sequence.size());
		result.alignmentScore = result.trace->score;
		result.alignmentStart = seqstart;
		result.alignmentEnd = seqend + 1;
		return result;
	}
};

// ------------------------------------------------------------------ co-linear chaining (K4)

// Pair-valued range-max structure keyed by read coordinate. The reference uses a treap
// (src/AlignmentGraph.cpp:1600-1710); RMQ results do not depend on the tree shape, so a sorted map
// scanned over the query range is an exact stand-in (the oracle favours obviousness over speed).
struct RangeMax {
	typedef std::pair<long long, long long> V;
	std::multimap<long long, V> entries;
	V defaultValue;
	explicit RangeMax(V d) : defaultValue(d) {}
	void add(long long key, V value) { entries.emplace(key, value); }
	V RMQ(long long l, long long r) const
	{
		bool any = false;
		V best = defaultValue;
		for (auto it = entries.lower_bound(l); it != entries.end() && it->first <= r; ++it) {
			if (!any || it->second > best) best = it->second;
			any = true;
		}
		return best;
	}
};

// reference: src/AlignmentGraph.cpp:1741-1863
inline std::pair<std::vector<size_t>, size_t> colinearChainingByComponent(const AlignmentGraph& g, size_t cid, const std::vector<gc::Anchor>& A, const std::vector<size_t>& aids)
{
	typedef long long LL;
	typedef std::pair<LL, LL> P;
	const std::vector<size_t>& cids = g.component_ids[cid];
	size_t N = cids.size();
	LL K = (LL)g.mpc[cid].size();
	P defaultValue = { -(LL)N * 2, -1 };
	for (size_t j : aids) defaultValue.first -= (LL)(A[j].y + 1 - A[j].x) * 2;
	std::vector<RangeMax> T(K, RangeMax(defaultValue)), I(K, RangeMax(defaultValue));
	struct Endpoint { LL node, anchor, kind; };   // kind: -1 start, -2 end, k>=0 forwarded along path k
	std::vector<Endpoint> endpoints;
	std::vector<P> C(A.size());
	const auto& cidx = g.component_idx;
	for (size_t j : aids) {
		endpoints.push_back({ (LL)cidx[A[j].path[0]], (LL)j, -1 });
		endpoints.push_back({ (LL)cidx[A[j].path.back()], (LL)j, -2 });
		for (const auto& b : g.backwards[cid][cidx[A[j].path[0]]]) endpoints.push_back({ (LL)b.first, (LL)j, (LL)b.second });
		C[j] = { (LL)(A[j].y - A[j].x + 1), -1 };
	}
	// group order inside one node is immaterial to the result (all updates are max-merges), so a stable
	// sort is used here where the reference uses an unstable one (:1772)
	std::stable_sort(endpoints.begin(), endpoints.end(), [&](const Endpoint& a, const Endpoint& b) { return g.topo_ids[cid][a.node] < g.topo_ids[cid][b.node]; });
	for (size_t vidx = 0, ridx = 0; vidx < endpoints.size(); vidx = ridx) {
		LL v = endpoints[vidx].node;
		ridx = vidx + 1;
		while (ridx < endpoints.size() && endpoints[ridx].node == v) ridx++;
		std::vector<LL> ids;
		for (size_t e = vidx; e < ridx; e++) if (endpoints[e].kind < 0) ids.push_back(endpoints[e].anchor);
		if (!ids.empty()) {
			std::sort(ids.begin(), ids.end(), [&](LL a, LL b) { if (A[a].y != A[b].y) return A[a].y < A[b].y; if (A[a].x != A[b].x) return A[a].x < A[b].x; return a < b; });
			ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
			RangeMax tmpT(defaultValue), tmpI(defaultValue);
			for (LL j : ids) {
				if ((LL)cidx[A[j].path[0]] == v) {
					P q = tmpT.RMQ(0, (LL)A[j].x - 1);
					C[j] = std::max(C[j], P { (LL)(A[j].y - A[j].x + 1) + q.first, q.second });
					q = tmpI.RMQ((LL)A[j].x, (LL)A[j].y - 1);
					C[j] = std::max(C[j], P { (LL)A[j].y + q.first, q.second });
				}
				if ((LL)cidx[A[j].path.back()] == v) {
					tmpT.add((LL)A[j].y, { C[j].first, j });
					tmpI.add((LL)A[j].y, { C[j].first - (LL)A[j].y, j });
				}
			}
		}
		for (size_t e = vidx; e < ridx; e++) {
			if (endpoints[e].kind != -2) continue;
			LL j = endpoints[e].anchor;
			for (size_t k : g.paths[cid][v]) {
				T[k].add((LL)A[j].y, { C[j].first, j });
				I[k].add((LL)A[j].y, { C[j].first - (LL)A[j].y, j });
			}
		}
		for (size_t e = vidx; e < ridx; e++) {
			if (endpoints[e].kind < 0) continue;
			LL j = endpoints[e].anchor, k = endpoints[e].kind;
			P q = T[k].RMQ(0, (LL)A[j].x - 1);
			C[j] = std::max(C[j], P { (LL)(A[j].y - A[j].x + 1) + q.first, q.second });
			q = I[k].RMQ((LL)A[j].x, (LL)A[j].y - 1);
			C[j] = std::max(C[j], P { (LL)A[j].y + q.first, q.second });
		}
	}
	P best = { 0, -1 };
	for (size_t j : aids) best = std::max(best, P { C[j].first, (LL)j });
	std::vector<size_t> ret;
	for (LL i = best.second; i != -1; i = C[i].second) {
		ret.push_back((size_t)i);
		if (i == C[i].second) break;
	}
	std::reverse(ret.begin(), ret.end());
	return { ret, (size_t)best.first };
}

// reference: src/AlignmentGraph.cpp:1712-1739. Returns (chain, score).
inline std::pair<std::vector<size_t>, size_t> colinearChaining(const AlignmentGraph& g, const std::vector<gc::Anchor>& A)
{
	std::vector<std::pair<size_t, size_t>> cs(A.size());
	for (size_t i = 0; i < A.size(); i++) cs[i] = { g.component_map[A[i].path.back()], i };
	std::sort(cs.begin(), cs.end());
	std::pair<std::vector<size_t>, size_t> best { {}, 0 };
	bool first = true;
	for (size_t i = 0, j; i < cs.size(); i = j) {
		std::vector<size_t> aids;
		for (j = i; j < cs.size() && cs[j].first == cs[i].first; j++) aids.push_back(cs[j].second);
		auto tmp = colinearChainingByComponent(g, cs[i].first, A, aids);
		if (first || tmp.second > best.second) { first = false; best = tmp; }
	}
	return best;
}

// ------------------------------------------------------------------ edit distance (for the selection rule)

// Global (NW) edit distance: the value edlibAlign(..., EDLIB_MODE_NW, EDLIB_TASK_DISTANCE) returns at src/Aligner.cpp:645,845,
// computed like edlib does (banded, k doubling): oracle/edlib_path.hpp.
inline size_t editDistanceNW(const std::string& a, const std::string& b) { return (size_t)editDistanceBanded(a, b); }

} // namespace oracle
