// On-disk cache of the start-up data: the finalized split-node graph with its MPC index, and
// optionally the minimizer index, so that a later run skips GFA parsing, node splitting, the topological order, the
// greedy path cover + max-flow shrink and the minimizer scan (reference: src/AlignmentGraph.cpp:1267-1391,1465-1495,
// src/MinimizerSeeder.cpp:299-492). The reference declares saveMPC/loadMPC (src/AlignmentGraph.h:96-97) but leaves
// both bodies empty, so there is no reference format to follow; this one is private to this library.
//
// Format: "GCAMDIDX" magic, then the version and every field as LEB128 varints (vector<bool> bit-packed, strings as
// length + bytes), then the FNV-1a-64 of all preceding bytes. A load checks magic, version, bounds and checksum and
// throws on any mismatch: a stale or damaged cache must never be used silently.
#pragma once
#include <cstdint>
#include <string>
#include "gc_graph.hpp"

namespace gc {

struct IndexCacheInfo {
	uint64_t nodes = 0, bp = 0, kmers = 0, positions = 0, k = 0, w = 0, fileBytes = 0;
	bool hasSeeder = false;
};

constexpr uint32_t INDEX_CACHE_VERSION = 1;

// idx may be null (graph only)
void SaveIndexCache(const std::string& path, const AlignmentGraph& g, const MinimizerIndex* idx);
// fills g (and idx when the file holds one); throws std::runtime_error on any format error
IndexCacheInfo LoadIndexCache(const std::string& path, AlignmentGraph& g, MinimizerIndex& idx);
// load, then serialise what was loaded again and compare it byte for byte with the file
IndexCacheInfo CheckIndexCache(const std::string& path);

} // namespace gc
