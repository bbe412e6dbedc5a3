// Minimizer index construction (start-up, CPU). reference: src/MinimizerSeeder.cpp:104-189 (window
// minimizers), :299-492 (initMinimizers), :557-575 (initMaxCount).
//
// The reference stores the index behind a BBHash MPHF + sdsl packed vectors (both un-vendored, see
// SURVEY.md §8c). Only three observable properties of that storage matter for seeds and they are kept:
//   1. which (k-mer -> positions) lists exist: every window minimizer of every bigraph node, all hash
//      ties included;
//   2. the order inside a k-mer's position list: filled back to front in arrival order (:473-482),
//      i.e. REVERSE arrival order, arrival = nodeLookup iteration x position order at -t 1;
//   3. maxCount = counts[floor(0.999*n)] + 1 over all keys except "the last MPHF index" (:564).
//      Which key that is depends on the MPHF and is parity-unpinned; defined here as the largest k-mer.
#include "gc_graph.hpp"
#include <exception>
#include <thread>
#include <algorithm>
#include <deque>
#include <tuple>

namespace gc {

uint64_t minimizerHash(uint64_t key)   // reference: src/MinimizerSeeder.cpp:45-54
{
	key = (~key) + (key << 21);
	key = key ^ (key >> 24);
	key = (key + (key << 3)) + (key << 8);
	key = key ^ (key >> 14);
	key = (key + (key << 2)) + (key << 4);
	key = key ^ (key >> 28);
	key = key + (key << 31);
	return key;
}

static inline int baseCode(char c)
{
	switch (c) {
		case 'a': case 'A': return 0;
		case 'c': case 'C': return 1;
		case 'g': case 'G': return 2;
		case 't': case 'T': return 3;
	}
	return -1;
}

// Sliding-window minimizers with a monotone deque, every tie reported once.
// reference: src/MinimizerSeeder.cpp:104-189 (iterateMinimizersReal). Emission rules kept:
//  - a run of valid (ACGT) characters shorter than the window yields nothing;
//  - the first full window reports all entries equal to the window minimum, in position order;
//  - afterwards a new report happens when the minimum changes (all ties), or when the newly
//    entered k-mer ties the current minimum (that one k-mer only).
template <typename F>
static void forEachWindowMinimizer(const std::string& str, size_t k, size_t w, F&& emit)
{
	if (str.size() < k) return;
	const size_t kmersPerWindow = w - k + 1;
	const uint64_t mask = ~(~(uint64_t)0 << (k * 2));
	struct Entry { size_t pos; uint64_t kmer, hash; };
	std::deque<Entry> window;
	size_t offset = 0;
	while (true) {
	restart:
		while (offset < str.size() && baseCode(str[offset]) < 0) offset++;
		if (offset + w > str.size()) return;
		uint64_t kmer = 0;
		for (size_t i = 0; i < k; i++) {
			int c = baseCode(str[offset + i]);
			if (c < 0) { offset += i; goto restart; }
			kmer = (kmer << 2) | (uint64_t)c;
		}
		window.clear();
		window.push_back({ offset + k - 1, kmer, minimizerHash(kmer) });
		for (size_t i = k; i < k + kmersPerWindow; i++) {
			int c = baseCode(str[offset + i]);
			if (c < 0) { offset += i; goto restart; }
			kmer = ((kmer << 2) & mask) | (uint64_t)c;
			uint64_t h = minimizerHash(kmer);
			while (!window.empty() && window.back().hash > h) window.pop_back();
			window.push_back({ offset + i, kmer, h });
		}
		for (auto it = window.begin(); it != window.end() && it->hash == window.front().hash; ++it) emit(it->pos, it->kmer);
		for (size_t i = k + kmersPerWindow; offset + i < str.size(); i++) {
			int c = baseCode(str[offset + i]);
			if (c < 0) { offset += i; goto restart; }
			kmer = ((kmer << 2) & mask) | (uint64_t)c;
			uint64_t h = minimizerHash(kmer);
			uint64_t oldMinimum = window.front().hash;
			bool frontPopped = false;
			while (!window.empty() && window.front().pos <= offset + i - kmersPerWindow) { frontPopped = true; window.pop_front(); }
			if (frontPopped)
				while (window.size() >= 2 && window[0].hash == window[1].hash) window.pop_front();
			while (!window.empty() && window.back().hash > h) window.pop_back();
			window.push_back({ offset + i, kmer, h });
			if (window.front().hash != oldMinimum) {
				for (auto it = window.begin(); it != window.end() && it->hash == window.front().hash; ++it) emit(it->pos, it->kmer);
			} else if (window.back().hash == window.front().hash) {
				emit(window.back().pos, window.back().kmer);
			}
		}
		return;
	}
}

MinimizerIndex MinimizerIndex::Build(const AlignmentGraph& g, size_t k, size_t w, double keepLeastFrequentFraction)
{
	MinimizerIndex idx;
	idx.k = k;
	idx.w = w;
	// minimizers that end inside an overlap prefix are skipped (:323-340,369); zero for 0M graphs
	// (by node id, flat: the reference's unordered_map holds one entry per bigraph node - 25 GB of nodes and buckets at 3.1 Gbp)
	int maxId = -1;
	for (size_t i = 0; i < g.NodeSize(); i++) maxId = std::max(maxId, g.nodeIDs[i]);
	std::vector<uint32_t> nodeMinimizerStart((size_t)(maxId + 1), 0);
	for (size_t i = 0; i < g.NodeSize(); i++) {
		uint32_t& start = nodeMinimizerStart[(size_t)g.nodeIDs[i]];
		for (size_t nb : g.inNeighbors[i])
			if (g.nodeIDs[nb] != g.nodeIDs[i]) { start = std::max(start, (uint32_t)g.nodeOffset[i]); break; }
	}
	const std::vector<int>& idOrder = g.nodeLookupOrder;   // arrival order at -t 1: nodeLookup iteration order (:354-357)
	if (idOrder.size() != g.nodeLookup.size()) throw std::runtime_error("MinimizerIndex::Build: the graph's node lookup order is missing");
	// The nodes are scanned in contiguous chunks of that order by several threads; joining the chunks' lists in chunk order
	// gives the single-threaded arrival order.
	auto scan = [&](size_t from, size_t to, std::vector<std::pair<uint64_t, uint64_t>>& out) {   // (kmer, packed position)
		std::string sequence;
		for (size_t at = from; at < to; at++) {
			int nodeId = idOrder[at];
			const NodeLookup::Span splitNodes = g.nodeLookup.at(nodeId);
			sequence.resize(g.originalNodeSize.at(nodeId));
			size_t filled = 0;
			for (size_t split : splitNodes)
				for (size_t j = 0; j < g.nodeLength[split]; j++) sequence[filled++] = g.NodeSequences(split, j);
			size_t minStart = nodeMinimizerStart.at((size_t)nodeId);
			forEachWindowMinimizer(sequence, k, w, [&](size_t pos, uint64_t kmer) {
				if (pos < minStart) return;
				size_t split = g.GetUnitigNode(nodeId, pos);
				out.emplace_back(kmer, ((uint64_t)split << 6) + (pos - g.nodeOffset[split]));
			});
		}
	};
	std::vector<std::pair<uint64_t, uint64_t>> arrivals;
	size_t workers = std::min<size_t>(buildThreads(), std::max<size_t>(1, idOrder.size() / 4096));
	if (workers <= 1) {
		scan(0, idOrder.size(), arrivals);
	} else {
		std::vector<std::vector<std::pair<uint64_t, uint64_t>>> parts(workers);
		std::vector<std::exception_ptr> errors(workers);
		std::vector<std::thread> threads;
		for (size_t t = 0; t < workers; t++)
			threads.emplace_back([&, t]() {
				try {
					scan(idOrder.size() * t / workers, idOrder.size() * (t + 1) / workers, parts[t]);
				} catch (...) {
					errors[t] = std::current_exception();
				}
			});
		for (auto& th : threads) th.join();
		for (auto& e : errors) if (e) std::rethrow_exception(e);
		size_t total = 0;
		for (const auto& part : parts) total += part.size();
		arrivals.reserve(total);
		for (auto& part : parts) { arrivals.insert(arrivals.end(), part.begin(), part.end()); std::vector<std::pair<uint64_t, uint64_t>>().swap(part); }
	}
	// group by k-mer; inside a group the reference's list is the arrivals reversed (:473-482)
	std::vector<size_t> order(arrivals.size());
	for (size_t i = 0; i < order.size(); i++) order[i] = i;
	std::sort(order.begin(), order.end(), [&](size_t a, size_t b) {
		if (arrivals[a].first != arrivals[b].first) return arrivals[a].first < arrivals[b].first;
		return a > b;
	});
	idx.positions.reserve(order.size());
	for (size_t i = 0; i < order.size(); i++) {
		uint64_t kmer = arrivals[order[i]].first;
		if (idx.kmers.empty() || idx.kmers.back() != kmer) {
			idx.kmers.push_back(kmer);
			idx.startPos.push_back(i);
		}
		idx.positions.push_back(arrivals[order[i]].second);
	}
	idx.startPos.push_back(order.size());
	idx.maxCount = minimizerMaxCount(idx.startPos, keepLeastFrequentFraction);
	return idx;
}

// reference: src/MinimizerSeeder.cpp:557-575 (the last k-mer's list is left out of the quantile there, and so here)
size_t minimizerMaxCount(const std::vector<uint64_t>& startPos, double keepLeastFrequentFraction)
{
	size_t nKmers = startPos.empty() ? 0 : startPos.size() - 1;
	if (nKmers < 2) return 0;
	std::vector<size_t> counts;
	counts.reserve(nKmers - 1);
	for (size_t i = 0; i + 1 < nKmers; i++) counts.push_back(startPos[i + 1] - startPos[i]);
	std::sort(counts.begin(), counts.end());
	size_t at = (size_t)(counts.size() * keepLeastFrequentFraction);
	if (at == counts.size()) at = counts.size() - 1;
	return counts[at] + 1;
}

} // namespace gc
