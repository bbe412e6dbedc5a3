// Index cache; see gc_index_cache.hpp for the format.
#include "gc_index_cache.hpp"
#include "gc_stageclock.hpp"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <unordered_map>
#include <fcntl.h>
#include <malloc.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace gc {
namespace {

const char MAGIC[8] = { 'G', 'C', 'A', 'M', 'D', 'I', 'D', 'X' };

inline uint64_t fnv(uint64_t h, const uint8_t* p, size_t n)
{
	for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
	return h;
}
constexpr uint64_t FNV_SEED = 0xcbf29ce484222325ull;

// where serialised bytes go: a file, or a comparison with bytes already in memory (CheckIndexCache)
struct Sink {
	virtual ~Sink() {}
	virtual void write(const uint8_t* p, size_t n) = 0;
};

struct FileSink : Sink {
	FILE* f;
	explicit FileSink(const std::string& path) : f(fopen(path.c_str(), "wb")) { if (!f) throw std::runtime_error("cannot create " + path); }
	~FileSink() override { if (f) fclose(f); }
	void write(const uint8_t* p, size_t n) override { if (fwrite(p, 1, n, f) != n) throw std::runtime_error("short write to the index cache"); }
	void close() { int rc = fclose(f); f = nullptr; if (rc != 0) throw std::runtime_error("cannot finish writing the index cache"); }
};

struct CompareSink : Sink {
	const uint8_t* expect;
	size_t size, at = 0;
	CompareSink(const uint8_t* e, size_t n) : expect(e), size(n) {}
	void write(const uint8_t* p, size_t n) override
	{
		if (at + n > size || memcmp(expect + at, p, n) != 0) throw std::runtime_error("index cache does not re-serialise to itself near byte " + std::to_string(at));
		at += n;
	}
};

class Out {
public:
	explicit Out(Sink& s) : sink(s) { buf.reserve(CAP + 16); }
	void raw(const void* p, size_t n)
	{
		const uint8_t* b = (const uint8_t*)p;
		while (n > 0) {
			size_t take = std::min(n, CAP - buf.size());
			buf.insert(buf.end(), b, b + take);
			b += take; n -= take;
			if (buf.size() >= CAP) flush();
		}
	}
	void num(uint64_t v)
	{
		uint8_t tmp[10];
		size_t n = 0;
		while (v >= 0x80) { tmp[n++] = (uint8_t)(v | 0x80); v >>= 7; }
		tmp[n++] = (uint8_t)v;
		raw(tmp, n);
	}
	void finish()
	{
		flush();
		uint64_t h = hash;
		uint8_t tail[8];
		for (int i = 0; i < 8; i++) tail[i] = (uint8_t)(h >> (8 * i));
		sink.write(tail, 8);
	}
	// ---- typed fields ----
	void operator()(const uint64_t& v) { num(v); }
	void operator()(const int& v) { num((uint64_t)(uint32_t)v); }
	void operator()(const bool& v) { num(v ? 1 : 0); }
	void operator()(const std::string& s) { num(s.size()); raw(s.data(), s.size()); }
	void operator()(const std::vector<bool>& v)
	{
		num(v.size());
		uint8_t byte = 0;
		for (size_t i = 0; i < v.size(); i++) {
			if (v[i]) byte |= (uint8_t)(1u << (i & 7));
			if ((i & 7) == 7 || i + 1 == v.size()) { raw(&byte, 1); byte = 0; }
		}
	}
	void operator()(const AmbiguousSeq& s) { num(s.A); num(s.T); num(s.C); num(s.G); }
	template <typename A, typename B> void operator()(const std::pair<A, B>& p) { (*this)(p.first); (*this)(p.second); }
	template <typename T, size_t N> void operator()(const std::array<T, N>& a) { for (const T& x : a) (*this)(x); }
	template <typename T> void operator()(const std::vector<T>& v) { num(v.size()); for (const T& x : v) (*this)(x); }
private:
	static constexpr size_t CAP = 1 << 20;
	void flush() { if (buf.empty()) return; hash = fnv(hash, buf.data(), buf.size()); sink.write(buf.data(), buf.size()); buf.clear(); }
	Sink& sink;
	std::vector<uint8_t> buf;
	uint64_t hash = FNV_SEED;
};

class In {
public:
	In(const uint8_t* p, size_t n) : base(p), end(n) {}
	size_t at = 0;
	// the mapped pages behind the cursor go back to the system every 256 MB: a 59 GB cache (3.1 Gbp) would otherwise sit in the process's resident memory beside the
	// graph it is parsed into (clean file pages: a later reader - CheckIndexCache's comparison - faults them in again)
	void dropConsumed()
	{
		const size_t page = 4096, upTo = at & ~(page - 1);
		if (upTo < dropped + ((size_t)256 << 20)) return;
		const uintptr_t from = ((uintptr_t)base + dropped + page - 1) & ~(uintptr_t)(page - 1), to = ((uintptr_t)base + upTo) & ~(uintptr_t)(page - 1);
		if (to > from) (void)madvise((void*)from, (size_t)(to - from), MADV_DONTNEED);
		dropped = upTo;
	}
	void raw(void* p, size_t n) { need(n); memcpy(p, base + at, n); at += n; }
	uint64_t num()
	{
		uint64_t v = 0;
		for (int shift = 0; shift < 70; shift += 7) {
			need(1);
			uint8_t b = base[at++];
			v |= (uint64_t)(b & 0x7f) << shift;
			if (!(b & 0x80)) return v;
		}
		throw std::runtime_error("index cache: malformed number");
	}
	// an element count: every element takes at least minBits bits of the remaining payload
	size_t count(size_t minBits = 8)
	{
		uint64_t n = num();
		if (n > (uint64_t)(end - at) * 8 / minBits + 8) throw std::runtime_error("index cache: element count exceeds the file size");
		dropConsumed();
		return (size_t)n;
	}
	void operator()(uint64_t& v) { v = num(); }
	void operator()(int& v) { v = (int)(uint32_t)num(); }
	void operator()(bool& v) { v = num() != 0; }
	void operator()(std::string& s) { size_t n = count(); s.resize(n); raw(&s[0], n); }
	void operator()(std::vector<bool>& v)
	{
		size_t n = count(1);
		v.assign(n, false);
		need((n + 7) / 8);
		for (size_t i = 0; i < n; i++) if (base[at + i / 8] >> (i & 7) & 1) v[i] = true;
		at += (n + 7) / 8;
	}
	void operator()(AmbiguousSeq& s) { s.A = num(); s.T = num(); s.C = num(); s.G = num(); }
	template <typename A, typename B> void operator()(std::pair<A, B>& p) { (*this)(p.first); (*this)(p.second); }
	template <typename T, size_t N> void operator()(std::array<T, N>& a) { for (T& x : a) (*this)(x); }
	template <typename T> void operator()(std::vector<T>& v) { size_t n = count(); v.clear(); v.resize(n); for (T& x : v) (*this)(x); }
private:
	void need(size_t n) const { if (n > end - at) throw std::runtime_error("index cache: truncated"); }
	const uint8_t* base;
	size_t end;
	size_t dropped = 0;
};

// The one list of fields, shared by both directions. The three id-keyed hash maps travel as parallel vectors in
// nodeLookup's iteration order, which MinimizerIndex::Build depends on (src/MinimizerSeeder.cpp:354-357 iterates the map);
// the loaded graph keeps that order in nodeLookupOrder because re-inserting the keys would not reproduce it.
struct GraphTables {
	std::vector<int> ids;
	std::vector<std::vector<size_t>> splitNodes;
	std::vector<size_t> sizes;
	std::vector<std::string> names;
};

template <typename Archive, typename Graph, typename Tables>
void graphFields(Archive& a, Graph& g, Tables& t)
{
	a(g.nodeLength); a(g.nodeOffset); a(g.nodeIDs);
	a(g.inNeighbors); a(g.outNeighbors);
	a(g.reverse); a(g.linearizable); a(g.ambiguousNodes);
	a(g.nodeSequences); a(g.ambiguousNodeSequences);
	a(g.componentNumber); a(g.chainNumber); a(g.chainApproxPos);
	a(g.bpSize); a(g.firstAmbiguous); a(g.finalized);
	a(g.component_map); a(g.component_idx); a(g.component_ids);
	a(g.topo); a(g.topo_ids);
	a(g.mpc); a(g.paths); a(g.backwards);
	a(t.ids); a(t.splitNodes); a(t.sizes); a(t.names);
}

template <typename Archive, typename Index>
void seederFields(Archive& a, Index& idx)
{
	a(idx.k); a(idx.w); a(idx.maxCount);
	a(idx.kmers); a(idx.startPos); a(idx.positions);
}

void serialise(Sink& sink, const AlignmentGraph& g, const MinimizerIndex* idx)
{
	if (!g.finalized) throw std::runtime_error("only a finalized graph can be cached");
	Out out(sink);
	out.raw(MAGIC, 8);
	out.num(INDEX_CACHE_VERSION);
	GraphTables t;
	t.ids = g.nodeLookupOrder;
	if (t.ids.size() != g.nodeLookup.size()) throw std::runtime_error("node lookup order is out of date");
	for (int id : t.ids) {
		const NodeLookup::Span nodes = g.nodeLookup.at(id);
		t.splitNodes.emplace_back(nodes.begin(), nodes.end());
		t.sizes.push_back(g.originalNodeSize.at(id));
		const std::string* name = g.originalNodeName.find(id);
		t.names.push_back(name ? *name : std::string());
	}
	graphFields(out, g, t);
	out.num(idx ? 1 : 0);
	if (idx) seederFields(out, *idx);
	out.finish();
}

struct Mapping {
	const uint8_t* p = nullptr;
	size_t n = 0;
	explicit Mapping(const std::string& path)
	{
		int fd = open(path.c_str(), O_RDONLY);
		if (fd < 0) throw std::runtime_error("cannot open index cache " + path);
		struct stat st;
		if (fstat(fd, &st) != 0 || st.st_size < 8 + 1 + 8) { close(fd); throw std::runtime_error("index cache " + path + " is too short"); }
		n = (size_t)st.st_size;
		void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
		close(fd);
		if (m == MAP_FAILED) throw std::runtime_error("cannot map index cache " + path);
		p = (const uint8_t*)m;
	}
	~Mapping() { if (p) munmap((void*)p, n); }
};

// Every index stored in the file is checked against the array it points into, so that a file that passes is safe to use
// without further bounds checks (uploadGraph and the stitching code index these arrays directly).
void validateGraph(const AlignmentGraph& g, const GraphTables& t)
{
	auto bad = [](const char* what) { throw std::runtime_error(std::string("index cache: ") + what); };
	const size_t n = g.nodeLength.size();
	if (g.linearizable.size() != n || g.chainNumber.size() != n || g.chainApproxPos.size() != n) bad("inconsistent per-node arrays");
	const size_t plain = std::min(g.firstAmbiguous, n);
	if (g.nodeSequences.size() != plain || g.ambiguousNodeSequences.size() != n - plain) bad("sequence arrays do not match the node count");
	for (size_t i = 0; i < n; i++) {
		if (g.nodeLength[i] < 1 || g.nodeLength[i] > (size_t)AlignmentGraph::SPLIT_NODE_SIZE) bad("node length outside 1..64");
		for (size_t v : g.inNeighbors[i]) if (v >= n) bad("in-neighbour outside the graph");
		for (size_t v : g.outNeighbors[i]) if (v >= n) bad("out-neighbour outside the graph");
	}
	// the kernels walk the graph in componentNumber order and terminate because it is a topological rank (src/AlignmentGraph.cpp:1100-1105 on
	// a DAG): an edge that does not go strictly upwards (a cycle, or a damaged rank) would let a device loop run forever - refuse it here.
	// Adjacency must also be symmetric: the backtrace reads in-neighbours, the forward pass out-neighbours.
	for (size_t i = 0; i < n; i++) {
		for (size_t v : g.outNeighbors[i]) {
			if (!(g.componentNumber[v] > g.componentNumber[i])) bad("an edge does not follow the topological rank (cyclic graph or damaged rank)");
			if (std::find(g.inNeighbors[v].begin(), g.inNeighbors[v].end(), i) == g.inNeighbors[v].end()) bad("an edge is missing from its target's in-neighbours");
		}
		for (size_t u : g.inNeighbors[i]) if (std::find(g.outNeighbors[u].begin(), g.outNeighbors[u].end(), i) == g.outNeighbors[u].end()) bad("an edge is missing from its source's out-neighbours");
	}
	const size_t nComp = g.component_ids.size();
	if (g.topo.size() != nComp || g.topo_ids.size() != nComp || g.mpc.size() != nComp || g.paths.size() != nComp || g.backwards.size() != nComp) bad("inconsistent component tables");
	size_t members = 0;
	for (size_t c = 0; c < nComp; c++) {
		const size_t m = g.component_ids[c].size(), width = g.mpc[c].size();
		members += m;
		if (g.topo[c].size() != m || g.topo_ids[c].size() != m || g.paths[c].size() != m || g.backwards[c].size() != m) bad("inconsistent tables inside a component");
		for (size_t v : g.component_ids[c]) if (v >= n) bad("component member outside the graph");
		for (size_t v : g.topo[c]) if (v >= m) bad("topological order entry outside its component");
		for (size_t v : g.topo_ids[c]) if (v >= m) bad("topological position outside its component");
		for (const auto& path : g.mpc[c]) for (size_t v : path) if (v >= n) bad("path cover node outside the graph");
		for (const auto& through : g.paths[c]) for (size_t k : through) if (k >= width) bad("path id outside the cover");
		for (const auto& back : g.backwards[c]) for (const auto& b : back) if (b.first >= m || b.second >= width) bad("backward link outside its component");
	}
	if (members != n) bad("components do not partition the graph");
	for (size_t i = 0; i < n; i++) {
		if (g.component_map[i] >= nComp) bad("component id outside the table");
		if (g.component_idx[i] >= g.component_ids[g.component_map[i]].size() || g.component_ids[g.component_map[i]][g.component_idx[i]] != i) bad("component index does not point back at its node");
	}
	size_t listed = 0;
	for (size_t k = 0; k < t.ids.size(); k++) {
		if (t.ids[k] < 0) bad("negative node id");
		size_t expect = 0, bp = 0;
		for (size_t v : t.splitNodes[k]) {
			if (v >= n || g.nodeIDs[v] != t.ids[k] || g.nodeOffset[v] != expect) bad("split nodes do not tile their original node");
			expect += (size_t)AlignmentGraph::SPLIT_NODE_SIZE;
			bp += g.nodeLength[v];
		}
		if (bp != t.sizes[k]) bad("original node size does not match its split nodes");
		listed += t.splitNodes[k].size();
	}
	if (listed != n) bad("node lookup does not cover the graph");
	// the reverse-strand twin of every position is looked up as lookup[id ^ 1][(size - 1 - offset) / 64] (twinOf in the host pipeline and on the
	// device): every bigraph node needs its twin, listed once, with the same size
	if (g.firstAmbiguous > n) bad("first ambiguous node outside the graph");
	std::unordered_map<int, size_t> sizeOf;
	sizeOf.reserve(t.ids.size());
	for (size_t k = 0; k < t.ids.size(); k++) if (!sizeOf.emplace(t.ids[k], t.sizes[k]).second) bad("a node id is listed twice");
	for (size_t k = 0; k < t.ids.size(); k++) {
		auto twin = sizeOf.find(t.ids[k] ^ 1);
		if (twin == sizeOf.end() || twin->second != t.sizes[k]) bad("a node has no reverse-strand twin of the same size");
	}
}

void validateSeeder(const MinimizerIndex& idx, const AlignmentGraph& g)
{
	const size_t nodes = g.nodeLength.size();
	auto bad = [](const char* what) { throw std::runtime_error(std::string("index cache: ") + what); };
	if (idx.startPos.size() != idx.kmers.size() + 1 || idx.startPos.front() != 0 || idx.startPos.back() != idx.positions.size()) bad("inconsistent minimizer index");
	for (size_t i = 0; i + 1 < idx.startPos.size(); i++) if (idx.startPos[i] > idx.startPos[i + 1]) bad("minimizer offsets are not monotone");
	for (size_t i = 0; i + 1 < idx.kmers.size(); i++) if (idx.kmers[i] >= idx.kmers[i + 1]) bad("minimizer k-mers are not sorted");
	for (uint64_t p : idx.positions) if ((p >> 6) >= nodes || (p & 63) >= g.nodeLength[p >> 6]) bad("minimizer position outside the graph");
	if (idx.k < 1 || idx.k > 31) bad("minimizer length outside 1..31");
	if (!idx.kmers.empty() && (idx.kmers.back() >> (2 * idx.k)) != 0) bad("k-mer wider than 2k bits");
}

IndexCacheInfo parse(const Mapping& file, AlignmentGraph& g, MinimizerIndex& idx)
{
	StageClock clock("gc cache");
	if (memcmp(file.p, MAGIC, 8) != 0) throw std::runtime_error("not an index cache (bad magic)");
	size_t payload = file.n - 8;
	uint64_t stored = 0;
	for (int i = 0; i < 8; i++) stored |= (uint64_t)file.p[payload + i] << (8 * i);
	if (fnv(FNV_SEED, file.p, payload) != stored) throw std::runtime_error("index cache checksum mismatch (truncated or damaged file)");
	(void)madvise((void*)file.p, file.n, MADV_DONTNEED);   // (mmap returns page-aligned memory: the pages the checksum touched go back before the graph is parsed beside them)
	clock.lap("checksum");
	In in(file.p, payload);
	in.at = 8;
	uint64_t version = in.num();
	if (version != INDEX_CACHE_VERSION) throw std::runtime_error("index cache version " + std::to_string(version) + " is not the supported version " + std::to_string(INDEX_CACHE_VERSION));
	g = AlignmentGraph();
	GraphTables t;
	graphFields(in, g, t);
	clock.lap("graph fields");
	size_t n = g.nodeLength.size();
	if (t.splitNodes.size() != t.ids.size() || t.sizes.size() != t.ids.size() || t.names.size() != t.ids.size()) throw std::runtime_error("index cache: inconsistent node tables");
	if (g.nodeOffset.size() != n || g.nodeIDs.size() != n || g.inNeighbors.size() != n || g.outNeighbors.size() != n || g.reverse.size() != n
		|| g.componentNumber.size() != n || g.component_map.size() != n || g.component_idx.size() != n || !g.finalized)
		throw std::runtime_error("index cache: inconsistent graph arrays");
	validateGraph(g, t);
	clock.lap("graph validation");
	g.nodeLookup.reserve(t.ids.size(), n); g.originalNodeSize.reserve(t.ids.size()); g.originalNodeName.reserve(t.ids.size());
	for (size_t i = 0; i < t.ids.size(); i++) {
		g.nodeLookup.add(t.ids[i], t.splitNodes[i].data(), t.splitNodes[i].size());
		g.originalNodeSize[t.ids[i]] = t.sizes[i];
		if (!t.names[i].empty()) g.originalNodeName[t.ids[i]] = std::move(t.names[i]);
	}
	if (g.nodeLookup.size() != t.ids.size()) throw std::runtime_error("index cache: duplicate node id");
	g.nodeLookupOrder = std::move(t.ids);
	t = GraphTables();   // (the per-id staging lists - 20 GiB at 3.1 Gbp - go before the minimizer index is read beside the graph)
	malloc_trim(0);
	clock.lap("node tables");
	IndexCacheInfo info;
	info.hasSeeder = in.num() != 0;
	idx = MinimizerIndex();
	if (info.hasSeeder) {
		seederFields(in, idx);
		clock.lap("minimizer index fields");
		validateSeeder(idx, g);
		clock.lap("minimizer index validation");
	}
	if (in.at != payload) throw std::runtime_error("index cache: trailing bytes");
	info.nodes = n; info.bp = g.bpSize; info.kmers = idx.kmers.size(); info.positions = idx.positions.size();
	info.k = idx.k; info.w = idx.w; info.fileBytes = file.n;
	return info;
}

} // namespace

void SaveIndexCache(const std::string& path, const AlignmentGraph& g, const MinimizerIndex* idx)
{
	// written next to the target and renamed, so that a reader never sees a half-written cache
	std::string tmp = path + ".tmp" + std::to_string((long)getpid());
	try {
		FileSink sink(tmp);
		serialise(sink, g, idx);
		sink.close();
	} catch (...) {
		remove(tmp.c_str());
		throw;
	}
	if (rename(tmp.c_str(), path.c_str()) != 0) { remove(tmp.c_str()); throw std::runtime_error("cannot move the index cache to " + path); }
}

IndexCacheInfo LoadIndexCache(const std::string& path, AlignmentGraph& g, MinimizerIndex& idx)
{
	Mapping file(path);
	return parse(file, g, idx);
}

IndexCacheInfo CheckIndexCache(const std::string& path)
{
	Mapping file(path);
	AlignmentGraph g;
	MinimizerIndex idx;
	IndexCacheInfo info = parse(file, g, idx);
	CompareSink cmp(file.p, file.n);
	serialise(cmp, g, info.hasSeeder ? &idx : nullptr);
	if (cmp.at != file.n) throw std::runtime_error("index cache re-serialises to a different length");
	return info;
}

} // namespace gc
