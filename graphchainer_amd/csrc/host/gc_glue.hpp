// Host-side per-read glue of the batched pipeline: the order-critical small sorts that sit between the
// seed-lookup kernel and the extension kernels. They stay on the host with the same libstdc++ std::sort
// and the same inputs as the reference, because their (unstable) tie order defines anchor indices and
// thereby chaining tie-breaks (SURVEY.md §7 "Hard parts"). An unstable std::sort's permutation depends
// only on the comparison outcomes, so sorting these compact records instead of the reference's 80-byte
// SeedHit gives the identical order.
#pragma once
#include "gc_graph.hpp"
#include <cstdint>
#include <vector>

namespace gc {

struct SeedRec {   // reference: SeedHit, src/GraphAlignerWrapper.h:11-37 (only the fields the hot path reads)
	uint32_t node;                // split node of the seed base (alignmentGraphNodeId)
	uint32_t seqPos;              // read position of the LAST base of the k-mer
	uint32_t goodness;            // seedGoodness
	uint16_t clusterSize;         // seedClusterSize, saturated
	uint8_t offset;               // offset in the split node (alignmentGraphNodeOffset)
	uint8_t matchLen;
	uint32_t rawGoodness;
};

struct KmerMatch { uint32_t pos, key; };   // output of the seed-lookup kernel: read position, index into MinimizerIndex::kmers

struct FragmentWindow { uint32_t l, sl, sr; };

struct GlueScratch {   // per-thread reusable buffers
	struct Match { uint32_t pos, count; uint64_t start; };
	std::vector<Match> matches;
	struct Diag { uint64_t chain, diagonal; uint32_t index; };
	std::vector<Diag> diags;
	std::vector<uint32_t> clusterPos;
};

// reference: MinimizerSeeder::addMinimizers + matchToSeedHit, src/MinimizerSeeder.cpp:494-520,546-555
void expandSeeds(const MinimizerIndex& index, const KmerMatch* matches, size_t nMatches, size_t readLength, double density, std::vector<SeedRec>& out, GlueScratch& scratch);

// reference: GraphAligner::orderSeedsByChaining, src/GraphAligner.h:233-295. Returns false where the
// reference's asserts would throw (the caller flags the read).
bool orderSeedsByChaining(const AlignmentGraph& graph, std::vector<SeedRec>& seeds, GlueScratch& scratch);

// reference: the fragment loop of src/Aligner.cpp:667-679 (sort by seqPos, then the two-pointer window)
void fragmentWindows(std::vector<SeedRec>& seeds, size_t readLength, size_t splitLen, size_t splitGap, std::vector<FragmentWindow>& out);

// E-value of an alignment for --E-cutoff (reference: EValueCalculator, src/EValue.cpp:16-105; used by SelectECutoff,
// src/AlignmentSelection.cpp:91-99, with the 70 % identity model of src/Aligner.cpp:478-482). Doubles, the reference's operation
// order and the same libm, so the keep / drop decision is the reference's.
class EValueModel {
public:
	explicit EValueModel(double minIdentity = 0.7);
	double alignmentScore(size_t alignmentLength, size_t numEdits) const;
	double evalue(size_t databaseSize, size_t querySize, size_t alignmentLength, size_t numEdits) const;
	bool keeps(double cutoff, size_t databaseSize, size_t querySize, size_t alignmentLength, size_t numEdits) const { return cutoff == -1 || evalue(databaseSize, querySize, alignmentLength, numEdits) <= cutoff; }
private:
	double match, mismatch, lambda, K;
};

} // namespace gc
