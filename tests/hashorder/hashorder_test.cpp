// gc::HashOrder (graphchainer_amd/csrc/host/gc_hashorder.hpp) against the real libstdc++ containers whose iteration order it replays: the three key types the graph build
// depends on, sizes that cross many rehashes, ascending / shuffled / strided / clustered keys, keys erased afterwards. Prints "ok <cases>" or the first difference.
#include "gc_graph.hpp"
#include "gc_hashorder.hpp"
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <unordered_map>
#include <vector>

static int cases = 0;

template <class Map, class Key, class HashFn>
static bool check(const std::vector<Key>& keys, HashFn hashOf, const char* what)
{
	Map real;
	gc::HashOrder emu;
	for (const Key& k : keys) { real[k]; emu.insert(hashOf(k)); }
	std::vector<uint32_t> order = emu.order();
	if (order.size() != real.size()) { printf("FAIL %s: sizes %zu / %zu\n", what, order.size(), real.size()); return false; }
	size_t at = 0;
	for (const auto& kv : real) {
		if (!(keys[order[at]] == kv.first)) { printf("FAIL %s: position %zu of %zu\n", what, at, keys.size()); return false; }
		at++;
	}
	cases++;
	return true;
}

int main()
{
	std::mt19937_64 rng(12345);
	bool ok = true;
	auto intHash = [](int k) { return std::hash<int>()(k); };
	auto posHash = [](const gc::NodePos& p) { return gc::NodePosHash()(p); };
	for (size_t n : { (size_t)0, (size_t)1, (size_t)2, (size_t)11, (size_t)12, (size_t)13, (size_t)14, (size_t)29, (size_t)30, (size_t)100, (size_t)1000, (size_t)54321, (size_t)300000, (size_t)2000000 }) {
		std::vector<int> ascending(n), shuffled(n), strided(n), doubled, clustered(n), negative(n);
		for (size_t i = 0; i < n; i++) { ascending[i] = (int)i; shuffled[i] = (int)i; strided[i] = (int)(i * 7919 % 1000003 + (i / 1000003) * 1000003); clustered[i] = (int)((i % 97) * 100000 + i / 97); negative[i] = (int)i - (int)(n / 2); }
		for (size_t i = n; i > 1; i--) std::swap(shuffled[i - 1], shuffled[rng() % i]);
		for (size_t i = 0; i < n; i++) { doubled.push_back((int)(2 * i)); doubled.push_back((int)(2 * i + 1)); }   // AddNode's keys: 2 id, 2 id + 1
		ok = ok && check<std::unordered_map<int, std::string>>(ascending, intHash, "int ascending");
		ok = ok && check<std::unordered_map<int, std::string>>(shuffled, intHash, "int shuffled");
		ok = ok && check<std::unordered_map<int, std::vector<size_t>>>(doubled, intHash, "int doubled");
		ok = ok && check<std::unordered_map<int, size_t>>(strided, intHash, "int strided");
		ok = ok && check<std::unordered_map<int, size_t>>(clustered, intHash, "int clustered");
		ok = ok && check<std::unordered_map<int, size_t>>(negative, intHash, "int negative");
		// GFA link sources: (id, end) pairs in a file-like order - forward links of a chain, then some reverse ones, ids partly shuffled
		std::vector<gc::NodePos> sources;
		for (size_t i = 0; i < n; i++) sources.push_back(gc::NodePos { shuffled[i] / 2 * 2 + (int)(i & 1), (i % 3) != 0 });
		std::vector<gc::NodePos> unique;
		{
			std::unordered_map<gc::NodePos, int, gc::NodePosHash> seen;
			for (const gc::NodePos& p : sources) if (seen.emplace(p, 1).second) unique.push_back(p);
		}
		ok = ok && check<std::unordered_map<gc::NodePos, std::vector<gc::NodePos>, gc::NodePosHash>>(unique, posHash, "NodePos");
		if (!ok) return 1;
		// erased keys keep the others' order (orphan link sources are erased after parsing, src/GfaGraph.cpp:329-368)
		{
			std::unordered_map<gc::NodePos, std::vector<gc::NodePos>, gc::NodePosHash> real;
			gc::HashOrder emu;
			for (const gc::NodePos& p : unique) { real[p]; emu.insert(posHash(p)); }
			std::vector<char> gone(unique.size(), 0);
			for (size_t i = 0; i < unique.size(); i += 5) { real.erase(unique[i]); gone[i] = 1; }
			std::vector<uint32_t> order = emu.order();
			size_t at = 0;
			for (const auto& kv : real) {
				while (at < order.size() && gone[order[at]]) at++;
				if (at >= order.size() || !(unique[order[at]] == kv.first)) { printf("FAIL erase: n %zu\n", n); return 1; }
				at++;
			}
			cases++;
		}
	}
	printf("ok %d\n", cases);
	return 0;
}
