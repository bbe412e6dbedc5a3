// Host-side per-read glue of the batched pipeline: the order-critical small sorts that sit between the
// seed-lookup kernel and the extension kernels. They stay on the host with the same libstdc++ std::sort
// and the same inputs as the reference, because their (unstable) tie order defines anchor indices and
// thereby chaining tie-breaks (SURVEY.md §7 "Hard parts").
#pragma once
#include "gc_graph.hpp"
#include <cstdint>
#include <vector>

namespace gc {

struct SeedRec {   // reference: SeedHit, src/GraphAlignerWrapper.h:11-37 (only the fields the hot path reads)
	uint32_t node, offset;        // split-node coordinates of the seed base (alignmentGraphNodeId/Offset)
	uint32_t seqPos;              // read position of the LAST base of the k-mer
	uint32_t matchLen;
	uint64_t rawGoodness, goodness, clusterSize;
};

struct KmerMatch { uint32_t pos, key; };   // output of the seed-lookup kernel: read position, index into MinimizerIndex::kmers

// reference: MinimizerSeeder::addMinimizers + matchToSeedHit, src/MinimizerSeeder.cpp:494-520,546-555
void expandSeeds(const MinimizerIndex& index, const KmerMatch* matches, size_t nMatches, size_t readLength, double density, std::vector<SeedRec>& out);

// reference: GraphAligner::orderSeedsByChaining, src/GraphAligner.h:233-295. Throws std::runtime_error where the
// reference's asserts would (the caller flags the read).
void orderSeedsByChaining(const AlignmentGraph& graph, std::vector<SeedRec>& seeds);

struct FragmentWindow { uint32_t l, sl, sr; };
// reference: the fragment loop of src/Aligner.cpp:667-679 (sort by seqPos, then the two-pointer window)
void fragmentWindows(std::vector<SeedRec>& seeds, size_t readLength, size_t splitLen, size_t splitGap, std::vector<FragmentWindow>& out);

} // namespace gc
