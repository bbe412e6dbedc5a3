// Host-side graph model for the MI355X GraphChainer hot path.
//
// This is the start-up ("A0") data the per-read kernels consume:
// the split-node DAG with the reference's exact node numbering, adjacency order,
// topological componentNumber, chain labels, the minimum-path-cover (MPC) index
// and the minimizer index. It is built once on the CPU and uploaded to HBM.
//
// Numbering parity with the reference depends on libstdc++ unordered_map iteration
// order (reference: src/BigraphToDigraph.cpp:229,251, src/AlignmentGraph.cpp:583).
// Two builders: LoadFromFile + BuildFromGFA use the same container types and insertion
// sequences as the reference (the literal restatement; tests compare the other one with it),
// BuildFromGFAFile (r4) replays the containers' order (gc_hashorder.hpp) over flat arrays
// filled by several threads - what gc_graph_create_from_gfa and gc_index_build run.
#pragma once
#include "gc_idmap.hpp"
#include <cstdint>
#include <cstddef>
#include <array>
#include <istream>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace gc {

struct NodePos {            // reference: src/GfaGraph.h:10-21
	int id;
	bool end;
	bool operator==(const NodePos& o) const { return id == o.id && end == o.end; }
};
struct NodePosHash {        // reference: src/GfaGraph.h:25-33
	size_t operator()(const NodePos& x) const { return std::hash<int>()(x.id) ^ std::hash<bool>()(x.end); }
};

// Parsed GFA (S and L lines only). reference: src/GfaGraph.cpp:212-370
struct GfaGraph {
	std::unordered_map<int, std::string> nodes;
	std::unordered_map<NodePos, std::vector<NodePos>, NodePosHash> edges;
	std::vector<std::pair<std::pair<NodePos, NodePos>, size_t>> overlaps; // (edge, overlap) in file order
	std::unordered_map<int, std::string> originalNodeName;
	static GfaGraph LoadFromStream(std::istream& in);
	static GfaGraph LoadFromFile(const std::string& path);
};

struct AmbiguousSeq { uint64_t A, T, C, G; };   // one-hot per base, reference: src/AlignmentGraph.h:36-68

struct Anchor {             // reference: src/AlignmentGraph.h:112-115
	std::vector<size_t> path;
	size_t x, y;
};

class AlignmentGraph {
public:
	static constexpr int SPLIT_NODE_SIZE = 64;   // reference: src/AlignmentGraph.h:20
	static constexpr size_t BP_IN_CHUNK = 32;
	static constexpr size_t CHUNKS_IN_NODE = 2;

	// ---- construction (reference: src/AlignmentGraph.cpp:51-307) ----
	static AlignmentGraph BuildFromGFA(const GfaGraph& gfa);   // reference: src/BigraphToDigraph.cpp:215-267
	static AlignmentGraph BuildFromGFAFile(const std::string& path);   // GfaGraph::LoadFromFile + BuildFromGFA in one, flat and threaded (gc_graph_fast.cpp); GC_BUILD_REFERENCE_CONTAINERS=1: the two above
	void AddNode(int nodeId, const std::string& sequence, const std::string& name, bool reverseNode, const std::vector<size_t>& breakpoints);
	void AddEdgeNodeId(int from, int to, size_t startOffset);
	void Finalize();
	void buildMPC(bool shrinkToMinimum = true);                // reference: src/AlignmentGraph.cpp:1465-1489

	// ---- accessors ----
	size_t NodeSize() const { return nodeLength.size(); }
	size_t NodeLength(size_t i) const { return nodeLength[i]; }
	size_t NodeOffset(size_t i) const { return nodeOffset[i]; }
	int NodeID(size_t i) const { return nodeIDs[i]; }
	char NodeSequences(size_t node, size_t pos) const;         // reference: src/AlignmentGraph.cpp:751-796
	size_t GetUnitigNode(int nodeId, size_t offset) const;     // reference: src/AlignmentGraph.cpp:832-848
	std::pair<int, size_t> GetReversePosition(int nodeId, size_t offset) const; // :850-868
	std::string OriginalNodeName(int nodeId) const;
	size_t SizeInBP() const { return bpSize; }

	// chaining over the MPC index, CPU (reference: src/AlignmentGraph.cpp:1712-1863). Used by the
	// oracle only; the product runs the HIP chaining kernel.
	std::vector<size_t> getChainPath(size_t S, size_t T, long long sepLimit) const; // :1866-1916

	// ---- data (reference: src/AlignmentGraph.h:145-172) ----
	std::vector<size_t> nodeLength;
	NodeLookup nodeLookup;                         // (gc_idmap.hpp: flat storage; the reference's is an unordered_map<int, vector<size_t>>)
	DenseIdMap<size_t> originalNodeSize;
	DenseIdMap<std::string> originalNodeName;
	// the iteration order of the reference's nodeLookup (an unordered_map) as it is when the graph has been built: what MinimizerIndex::Build follows
	// (src/MinimizerSeeder.cpp:354-357). Always set: by the builders, by the index cache, by gc_graph_create.
	std::vector<int> nodeLookupOrder;
	std::vector<size_t> nodeOffset;
	std::vector<int> nodeIDs;
	std::vector<std::vector<size_t>> inNeighbors;
	std::vector<std::vector<size_t>> outNeighbors;
	std::vector<bool> reverse;
	std::vector<bool> linearizable;
	std::vector<std::array<uint64_t, 2>> nodeSequences;        // 2 bits / bp, nodes < firstAmbiguous
	std::vector<AmbiguousSeq> ambiguousNodeSequences;          // nodes >= firstAmbiguous
	std::vector<bool> ambiguousNodes;
	std::vector<size_t> componentNumber;
	std::vector<size_t> chainNumber;
	std::vector<size_t> chainApproxPos;
	size_t bpSize = 0;
	size_t firstAmbiguous = SIZE_MAX;
	bool finalized = false;

	// MPC index (reference: src/AlignmentGraph.h:166-172)
	std::vector<size_t> component_map, component_idx;
	std::vector<std::vector<size_t>> component_ids;
	std::vector<std::vector<size_t>> topo, topo_ids;
	std::vector<std::vector<std::vector<size_t>>> mpc, paths;
	std::vector<std::vector<std::vector<std::pair<size_t, size_t>>>> backwards;

private:
	// the literal builder's nodeLookup, the reference's container (AddNode / AddEdgeNodeId work on it); BuildFromGFA turns it into nodeLookup + nodeLookupOrder
	std::unordered_map<int, std::vector<size_t>> buildLookup;
	void AddSplitNode(int nodeId, int offset, const std::string& sequence, bool reverseNode);
	void RenumberAmbiguousToEnd();
	void findLinearizable();
	void doComponentOrder();
	void findChains();
	void fixChainApproxPos(size_t start);
	void buildComponentsMap();
	std::vector<std::vector<size_t>> greedyCover(size_t cid) const;
	std::vector<std::vector<size_t>> shrink(size_t cid, const std::vector<std::vector<size_t>>& pc);
	void computeMPCIndex(size_t cid, const std::vector<std::vector<size_t>>& pc);
};

// ---- Minimizer index (reference: src/MinimizerSeeder.cpp:299-492, 557-575) ----
// Flat replacement of the reference's BBHash MPHF + sdsl packed vectors: sorted distinct k-mers,
// a prefix-sum array, and the per-k-mer position lists in the reference's order (reverse arrival).
struct MinimizerIndex {
	size_t k = 15, w = 20;
	std::vector<uint64_t> kmers;       // sorted distinct minimizer k-mers
	std::vector<uint64_t> startPos;    // size kmers.size()+1
	std::vector<uint64_t> positions;   // (splitNode << 6) | offsetInSplitNode of the k-mer's LAST base
	size_t maxCount = 0;               // k-mers with count >= maxCount give no seeds
	static MinimizerIndex Build(const AlignmentGraph& g, size_t k, size_t w, double keepLeastFrequentFraction);
};

size_t minimizerMaxCount(const std::vector<uint64_t>& startPos, double keepLeastFrequentFraction);   // src/MinimizerSeeder.cpp:557-575

// Threads the start-up builders may use (components of the MPC index, node chunks of the minimizer scan): GC_BUILD_THREADS,
// default = hardware threads. The results do not depend on it.
size_t buildThreads();
double cpuQuota();   // CPUs' worth of time the cgroup grants (cpu.max), 0 = unlimited / unknown

// 2-bit hash used to pick window minimizers. reference: src/MinimizerSeeder.cpp:45-54
uint64_t minimizerHash(uint64_t key);
std::string ReverseComplement(const std::string& s);   // reference: src/CommonUtils.cpp:67-134
char Complement(char c);

} // namespace gc
