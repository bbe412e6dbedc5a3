#include "gc_glue.hpp"
#include <algorithm>
#include <climits>
#include <cmath>

namespace gc {

void expandSeeds(const MinimizerIndex& index, const KmerMatch* matches, size_t nMatches, size_t readLength, double density, std::vector<SeedRec>& out, GlueScratch& scratch)
{
	auto& m = scratch.matches;
	m.resize(nMatches);
	for (size_t i = 0; i < nMatches; i++) {
		uint64_t start = index.startPos[matches[i].key];
		m[i] = { matches[i].pos, (uint32_t)(index.startPos[matches[i].key + 1] - start), start };
	}
	// "prefer less common minimizers": unstable sort by count alone, on the position-ordered list (:497)
	std::sort(m.begin(), m.end(), [](const GlueScratch::Match& l, const GlueScratch::Match& r) { return l.count < r.count; });
	size_t maxHits = (size_t)(readLength * density);
	if (density == -1) maxHits = SIZE_MAX;
	size_t seedsHere = 0, allowedCount = 0;
	out.clear();
	for (const auto& x : m) {
		if (seedsHere >= maxHits && x.count > allowedCount) break;
		allowedCount = x.count;
		for (uint64_t i = x.start; i < x.start + x.count; i++) {
			SeedRec s;
			s.node = (uint32_t)(index.positions[i] >> 6);
			s.offset = (uint8_t)(index.positions[i] & 63);
			s.seqPos = x.pos;
			s.matchLen = (uint8_t)index.k;
			s.rawGoodness = (uint32_t)(index.maxCount - x.count);
			s.goodness = 0;
			s.clusterSize = 0;
			out.push_back(s);
		}
		seedsHere += x.count;
	}
}

bool orderSeedsByChaining(const AlignmentGraph& graph, std::vector<SeedRec>& seeds, GlueScratch& scratch)
{
	// Seeds of one chain are clustered by diagonal (gap <= 100). Chains are independent and a cluster's
	// goodness depends only on the multiset of its seeds' read positions, so one sort by (chain, diagonal)
	// replaces the reference's per-chain hash map + two unstable sorts without changing any value.
	auto& d = scratch.diags;
	d.resize(seeds.size());
	for (size_t i = 0; i < seeds.size(); i++) {
		uint64_t base = graph.chainApproxPos[seeds[i].node] + seeds[i].offset;
		if (base < seeds[i].seqPos) return false;   // assert at :259
		d[i] = { graph.chainNumber[seeds[i].node], base - seeds[i].seqPos, (uint32_t)i };
	}
	std::sort(d.begin(), d.end(), [](const GlueScratch::Diag& l, const GlueScratch::Diag& r) { return l.chain != r.chain ? l.chain < r.chain : l.diagonal < r.diagonal; });
	auto& pos = scratch.clusterPos;
	size_t clusterStart = 0;
	for (size_t i = 1; i <= d.size(); i++) {
		if (i < d.size() && d[i].chain == d[i - 1].chain && d[i].diagonal <= d[i - 1].diagonal + 100) continue;
		pos.clear();
		for (size_t j = clusterStart; j < i; j++) pos.push_back(seeds[d[j].index].seqPos);
		std::sort(pos.begin(), pos.end());
		size_t matchingBps = 0;
		int lastEnd = INT_MIN;
		int matchLen = seeds[d[clusterStart].index].matchLen;
		for (uint32_t p : pos) {
			int thisStart = (int)p - matchLen + 1, thisEnd = (int)p;
			if (thisEnd <= thisStart) return false;   // assert at :280
			matchingBps += (size_t)(thisEnd - std::max(thisStart, lastEnd));
			lastEnd = thisEnd;
		}
		size_t size = i - clusterStart;
		for (size_t j = clusterStart; j < i; j++) {
			SeedRec& s = seeds[d[j].index];
			s.goodness = (uint32_t)(matchingBps + s.rawGoodness);
			s.clusterSize = (uint16_t)std::min<size_t>(size, 65535);
		}
		clusterStart = i;
	}
	std::sort(seeds.begin(), seeds.end(), [](const SeedRec& l, const SeedRec& r) { return l.goodness < r.goodness; });   // :293, order-critical
	std::reverse(seeds.begin(), seeds.end());
	return true;
}

void fragmentWindows(std::vector<SeedRec>& seeds, size_t readLength, size_t splitLen, size_t splitGap, std::vector<FragmentWindow>& out)
{
	std::sort(seeds.begin(), seeds.end(), [](const SeedRec& l, const SeedRec& r) { return l.seqPos < r.seqPos; });   // src/Aligner.cpp:667, order-critical
	out.clear();
	size_t sl = 0, sr = 0;
	for (size_t l = 0; l + splitLen <= readLength; l += splitGap) {
		while (sr < seeds.size() && (size_t)seeds[sr].seqPos + seeds[sr].matchLen <= l + splitLen) sr++;
		while (sl < sr && seeds[sl].seqPos < l) sl++;
		if (sl >= sr) continue;
		out.push_back({ (uint32_t)l, (uint32_t)sl, (uint32_t)sr });
	}
}

static const double kEuler = 2.71828182845904523536028747135266249775724709369995;

EValueModel::EValueModel(double minIdentity) : match(1), mismatch(-minIdentity / (1.0 - minIdentity)), lambda(-1), K(-1)
{
	// lambda solves E[e^(lambda * score)] = 1 for a fair match / mismatch coin; bisection on (0, 0.7) (src/EValue.cpp:50-76)
	double below = 0, above = 0.7;
	for (int step = 0; step < 100 && below != above; step++) {
		const double mid = (below + above) * 0.5;
		const double f = pow(kEuler, mid * match) * .5 + pow(kEuler, mid * mismatch) * 0.5 - 1;
		if (f < 0) below = mid;
		else if (f > 0) above = mid;
		else { below = above = mid; }
	}
	lambda = (below + above) / 2;
	// K from nine terms of the series over binomially distributed scores (src/EValue.cpp:78-105)
	double series = 0;
	std::vector<size_t> binom { 1 };
	for (int k = 1; k < 10; k++) {
		binom.push_back(0);
		for (size_t j = binom.size() - 1; j > 0; j--) binom[j] += binom[j - 1];   // next row of Pascal's triangle, in place
		size_t rowSum = 0;
		for (size_t n : binom) rowSum += n;
		double negativePart = 0, nonNegativeMass = 0;
		for (size_t j = 0; j < binom.size(); j++) {
			const double score = (double)j * match + (double)(binom.size() - 1 - j) * mismatch;
			const double p = (double)binom[j] / (double)rowSum;
			if (score < 0) negativePart += pow(kEuler, lambda * score) * p;
			if (score >= 0) nonNegativeMass += p;
		}
		series += (negativePart + nonNegativeMass) / (double)k;
	}
	const double meanWeighted = .5 * match * pow(kEuler, lambda * match) + .5 * mismatch * pow(kEuler, lambda * mismatch);
	const double cStar = pow(kEuler, -2 * series) / (lambda * meanWeighted);
	K = cStar * lambda / (1.0 - pow(kEuler, -lambda));
}

double EValueModel::alignmentScore(size_t alignmentLength, size_t numEdits) const { return alignmentLength * match - numEdits * (mismatch - match); }

double EValueModel::evalue(size_t databaseSize, size_t querySize, size_t alignmentLength, size_t numEdits) const
{
	return K * databaseSize * querySize * pow(kEuler, -lambda * alignmentScore(alignmentLength, numEdits));
}

} // namespace gc
