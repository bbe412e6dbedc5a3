// Maps keyed by bigraph node id (2 x GFA segment number, + 1 for the reverse strand: dense in a graph built from a GFA file) without a heap node per key.
//
// The reference keeps nodeLookup / originalNodeSize / originalNodeName in std::unordered_map<int, ...> (src/AlignmentGraph.h:145-149). Nothing downstream depends
// on THEIR iteration order except MinimizerIndex::Build, which follows AlignmentGraph::nodeLookupOrder (recorded when the graph is built, gc_hashorder.hpp), so the
// product keeps the values in vectors indexed by id: three node-based inserts per bigraph node were a third of a first build, and 200 bytes per id at human-genome scale.
// Ids a caller of gc_graph_create makes up may be sparse or negative: those go to a fallback hash map.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <unordered_map>
#include <vector>

namespace gc {

template <class T>
class DenseIdMap {
public:
	T& operator[](int id)
	{
		if (!sparse.empty()) { auto known = sparse.find(id); if (known != sparse.end()) return known->second; }   // (an id that went there before the dense range grew past it)
		if (isDense(id)) {
			if ((size_t)id >= values.size()) { values.resize((size_t)id + 1); present.resize((size_t)id + 1, 0); }
			if (!present[(size_t)id]) { present[(size_t)id] = 1; entries++; }
			return values[(size_t)id];
		}
		auto it = sparse.find(id);
		if (it == sparse.end()) { entries++; it = sparse.emplace(id, T()).first; }
		return it->second;
	}
	const T* find(int id) const
	{
		if (id >= 0 && (size_t)id < values.size() && present[(size_t)id]) return &values[(size_t)id];
		if (sparse.empty()) return nullptr;
		auto it = sparse.find(id);
		return it == sparse.end() ? nullptr : &it->second;
	}
	const T* end() const { return nullptr; }
	const T& at(int id) const { const T* p = find(id); if (!p) throw std::out_of_range("DenseIdMap::at: no such node id"); return *p; }
	size_t count(int id) const { return find(id) ? 1 : 0; }
	size_t size() const { return entries; }
	void reserve(size_t n) { values.reserve(n); present.reserve(n); }
	void denseUpTo(size_t n) { if (n > values.size()) { values.resize(n); present.resize(n, 0); } }   // ids below n will live in the vector, whatever order they come in
	// the ids [0, n) all at once (a graph built from a GFA file): the caller fills values() from several threads
	void assignDense(size_t n) { values.assign(n, T()); present.assign(n, 1); sparse.clear(); entries = n; }
	std::vector<T>& denseValues() { return values; }
	template <class F> void forEach(F f) { for (size_t i = 0; i < values.size(); i++) if (present[i]) f((int)i, values[i]); for (auto& kv : sparse) f(kv.first, kv.second); }
	template <class F> void forEach(F f) const { for (size_t i = 0; i < values.size(); i++) if (present[i]) f((int)i, values[i]); for (const auto& kv : sparse) f(kv.first, kv.second); }

private:
	// dense while the id stays within a few times the number of ids seen (a GFA's ids are 0 .. 2 n - 1); anything else is the caller's own numbering
	bool isDense(int id) const { return id >= 0 && ((size_t)id < values.size() || (size_t)id <= 4 * entries + 1024); }
	std::vector<T> values;
	std::vector<uint8_t> present;
	std::unordered_map<int, T> sparse;
	size_t entries = 0;
};

// nodeLookup: the split nodes of every bigraph node, in offset order (src/AlignmentGraph.h:145). One flat pool and a (begin, count) pair per id
// instead of a std::vector per id.
class NodeLookup {
public:
	struct Span {
		const size_t* first; const size_t* last;
		size_t size() const { return (size_t)(last - first); }
		bool empty() const { return first == last; }
		size_t operator[](size_t i) const { return first[i]; }
		size_t back() const { return last[-1]; }
		const size_t* begin() const { return first; }
		const size_t* end() const { return last; }
	};
	struct Range { uint64_t begin = 0; uint32_t count = 0; };
	bool contains(int id) const { return ranges.find(id) != nullptr; }
	size_t count(int id) const { return contains(id) ? 1 : 0; }
	size_t size() const { return ranges.size(); }
	Span at(int id) const { const Range& r = ranges.at(id); return Span { pool.data() + r.begin, pool.data() + r.begin + r.count }; }
	// a new id with its split nodes (appended to the pool)
	void add(int id, const size_t* nodes, size_t n)
	{
		if (contains(id)) throw std::runtime_error("NodeLookup::add: node id seen twice");
		Range& r = ranges[id];
		r.begin = pool.size(); r.count = (uint32_t)n;
		pool.insert(pool.end(), nodes, nodes + n);
	}
	void reserve(size_t ids, size_t nodes) { ranges.reserve(ids); pool.reserve(nodes); }
	void denseUpTo(size_t n) { ranges.denseUpTo(n); }
	// ids [0, nIds) with known counts, filled by the caller (several threads): returns after sizing; range(id) / poolData() give the storage
	void assignDense(size_t nIds, size_t nNodes) { ranges.assignDense(nIds); pool.assign(nNodes, 0); }
	std::vector<Range>& denseRanges() { return ranges.denseValues(); }
	std::vector<size_t>& nodes() { return pool; }   // (renumbering rewrites the values in place)
	const std::vector<size_t>& nodes() const { return pool; }

private:
	DenseIdMap<Range> ranges;
	std::vector<size_t> pool;
};

} // namespace gc
