#include "gc_glue.hpp"
#include <algorithm>
#include <climits>
#include <map>
#include <stdexcept>

namespace gc {

void expandSeeds(const MinimizerIndex& index, const KmerMatch* matches, size_t nMatches, size_t readLength, double density, std::vector<SeedRec>& out)
{
	struct Match { size_t pos, start, count; };
	std::vector<Match> m(nMatches);
	for (size_t i = 0; i < nMatches; i++) {
		size_t start = index.startPos[matches[i].key];
		m[i] = { matches[i].pos, start, (size_t)(index.startPos[matches[i].key + 1] - start) };
	}
	// "prefer less common minimizers": unstable sort by count alone, on the position-ordered list (:497)
	std::sort(m.begin(), m.end(), [](const Match& l, const Match& r) { return l.count < r.count; });
	size_t maxHits = (size_t)(readLength * density);
	if (density == -1) maxHits = SIZE_MAX;
	size_t seedsHere = 0, allowedCount = 0;
	out.clear();
	for (const Match& x : m) {
		if (seedsHere >= maxHits && x.count > allowedCount) break;
		allowedCount = x.count;
		for (size_t i = x.start; i < x.start + x.count; i++) {
			SeedRec s;
			s.node = (uint32_t)(index.positions[i] >> 6);
			s.offset = (uint32_t)(index.positions[i] & 63);
			s.seqPos = (uint32_t)x.pos;
			s.matchLen = (uint32_t)index.k;
			s.rawGoodness = index.maxCount - x.count;
			s.goodness = 0;
			s.clusterSize = 0;
			out.push_back(s);
		}
		seedsHere += x.count;
	}
}

void orderSeedsByChaining(const AlignmentGraph& graph, std::vector<SeedRec>& seeds)
{
	// seeds of one chain are clustered by diagonal; chains are independent, so the container order is immaterial
	std::map<size_t, std::vector<std::pair<size_t, size_t>>> byChain;
	for (size_t i = 0; i < seeds.size(); i++) {
		size_t diagonalBase = graph.chainApproxPos[seeds[i].node] + seeds[i].offset;
		if (diagonalBase < seeds[i].seqPos) throw std::runtime_error("seed before chain start");   // assert at :259
		byChain[graph.chainNumber[seeds[i].node]].emplace_back(i, diagonalBase - seeds[i].seqPos);
	}
	for (auto& entry : byChain) {
		auto& v = entry.second;
		std::sort(v.begin(), v.end(), [](std::pair<size_t, size_t> l, std::pair<size_t, size_t> r) { return l.second < r.second; });
		size_t clusterStart = 0;
		for (size_t i = 1; i <= v.size(); i++) {
			if (i < v.size() && v[i].second <= v[i - 1].second + 100) continue;
			std::sort(v.begin() + clusterStart, v.begin() + i, [&seeds](std::pair<size_t, size_t> l, std::pair<size_t, size_t> r) { return seeds[l.first].seqPos < seeds[r.first].seqPos; });
			size_t matchingBps = 0;
			int lastEnd = INT_MIN;
			for (size_t j = clusterStart; j < i; j++) {
				const SeedRec& s = seeds[v[j].first];
				int thisStart = (int)s.seqPos - (int)s.matchLen + 1;
				int thisEnd = (int)s.seqPos;
				if (thisEnd < lastEnd || thisEnd <= thisStart) throw std::runtime_error("seed cluster order");   // asserts at :279-280
				matchingBps += (size_t)(thisEnd - std::max(thisStart, lastEnd));
				lastEnd = thisEnd;
			}
			for (size_t j = clusterStart; j < i; j++) {
				seeds[v[j].first].goodness = matchingBps + seeds[v[j].first].rawGoodness;
				seeds[v[j].first].clusterSize = i - clusterStart;
			}
			clusterStart = i;
		}
	}
	std::sort(seeds.begin(), seeds.end(), [](const SeedRec& l, const SeedRec& r) { return l.goodness < r.goodness; });
	std::reverse(seeds.begin(), seeds.end());
}

void fragmentWindows(std::vector<SeedRec>& seeds, size_t readLength, size_t splitLen, size_t splitGap, std::vector<FragmentWindow>& out)
{
	std::sort(seeds.begin(), seeds.end(), [](const SeedRec& l, const SeedRec& r) { return l.seqPos < r.seqPos; });
	out.clear();
	size_t sl = 0, sr = 0;
	for (size_t l = 0; l + splitLen <= readLength; l += splitGap) {
		while (sr < seeds.size() && seeds[sr].seqPos + seeds[sr].matchLen <= l + splitLen) sr++;
		while (sl < sr && seeds[sl].seqPos < l) sl++;
		if (sl >= sr) continue;
		out.push_back({ (uint32_t)l, (uint32_t)sl, (uint32_t)sr });
	}
}

} // namespace gc
