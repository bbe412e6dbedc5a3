// Work-item / result records shared between the host pipeline and the HIP kernels, and the launch API.
#pragma once
#include "gc_device.hpp"

// Issue priority of the latency-bound helper kernels' waves (s_setprio, 0-3): the round loop's small kernels, the seed glue, chaining, stitching. They issue a few per cent of a batch's
// instructions and waited for their turn behind the whole-read kernel's waves when five batches are in flight. r4, five in flight, average per launch: k_stitch 109 -> 41 ms, k_chain<2> 31 -> 17,
// k_seed_glue 50 -> 45; the step 152.4 -> 151.2 ms (three interleaved pairs, `gpurun_out/r4_prio`): the whole-read kernel does not notice what they take. -DGC_HELPER_PRIO=0 turns it off.
#ifndef GC_HELPER_PRIO
#define GC_HELPER_PRIO 3
#endif
#if GC_HELPER_PRIO
#define GC_RAISE_PRIO() __builtin_amdgcn_s_setprio(GC_HELPER_PRIO)
#else
#define GC_RAISE_PRIO() ((void)0)
#endif
// the same for the edit-distance and path-letter kernels (vector work beside the scalar-bound whole-read kernel): -DGC_ED_PRIO=3, measured and off - 148.9 / 147.1 / 152.9 ms per batch against 149.2 / 151.1 / 150.8 (`gpurun_out/r4_edprio`)
#ifndef GC_ED_PRIO
#define GC_ED_PRIO 0
#endif
#if GC_ED_PRIO
#define GC_RAISE_ED_PRIO() __builtin_amdgcn_s_setprio(GC_ED_PRIO)
#else
#define GC_RAISE_ED_PRIO() ((void)0)
#endif

namespace gcdev {

struct SeedIndex {   // minimizer index in HBM (reference: MinimizerSeeder buckets, src/MinimizerSeeder.h:16-30)
	const uint64_t* table;      // open addressing: (low 32 bits of the k-mer << 32) | keyIndex, empty = all ones
	const uint64_t* wideKmers;  // k > 15 (the reference takes minimizer lengths up to 31, src/MinimizerSeeder.cpp:63): the k-mer of every key, checked when a slot's
	                            // 32-bit tag matches; null for k <= 15, where the tag is the whole k-mer
	uint32_t tableMask;
	const uint32_t* filter;     // 2^filterBits-bit membership filter over the keys: bit filterBit(kmer) is set for every key
	uint32_t filterShift;       // 64 - filterBits
	const uint64_t* startPos;   // [nKeys+1]
	const uint64_t* positions;  // [startPos[nKeys]] occurrences: (split node << 6) | offset of the k-mer's last base, every k-mer's list in the reference's order
	uint32_t nKeys;
	uint32_t maxCount;
	int32_t k, w;
};

struct ExtItem {     // one seed extension: align bases[seqOff .. seqOff+seqLen) starting just after (node, offset)
	uint64_t seqOff;
	uint32_t seqLen;
	uint32_t node;
	uint32_t offset;
	uint32_t pad;       // (r6) the read the sequence belongs to
};

struct ExtResult {
	uint64_t traceOff;
	uint32_t traceLen;
	uint32_t status;   // EXT_*
	int32_t score;
	uint32_t pad;      // bit 0 (r5): the last-slice minimum was attained in more than one node (ExtCounters::flattenTie)
};

struct Fragment;
// which work items a k_extend launch runs: all (mode 0), the two extensions of every fragment's first seed (1), a list with its count on the device (2)
struct ExtSelection { uint32_t mode = 0; const Fragment* frags = nullptr; uint32_t nFrags = 0; const uint32_t* list = nullptr; const unsigned long long* listCount = nullptr; };
// lazy extension rounds of k_build_anchors (see the kernel): this round's pending fragments in, next round's work list and pending fragments out
struct AnchorRounds {
	uint32_t lazy = 0, round = 0, parkAll = 0;
	const uint32_t* pending = nullptr; const unsigned long long* pendingCount = nullptr;
	uint32_t* nextList = nullptr; unsigned long long* nextListCount = nullptr;
	uint32_t* nextPending = nullptr; unsigned long long* nextPendingCount = nullptr;
	uint32_t* fragNext = nullptr;
};
struct Fragment {    // one read fragment with its seed window [seedBegin, seedEnd) in the batch-wide seed array
	uint32_t read;
	uint32_t l;
	uint32_t seedBegin, seedEnd;
};

struct FragSeed {    // a seed in fragment-pass order; node/offset = forward-strand split-node coordinates
	uint32_t node, offset;
	uint32_t seqPos;   // read position of the seed base
	uint32_t pad;
};

struct AnchorRec {   // one slot per (fragment, seed); valid anchors are compacted per read by the chaining kernel
	uint32_t valid;
	uint32_t x, y;
	uint32_t firstNode, firstOffset, firstSeqPos;   // Apos[0] (split node coords, seqPos inside the fragment)
	uint32_t lastNode, lastOffset, lastSeqPos;      // Apos[1]
	int32_t score;
	uint32_t pathLen;
	uint32_t pad;
	uint64_t pathOff;
};

struct ReadChainJob {
	uint32_t slotBegin, nSlots;   // this read's anchor slots
	uint32_t chainBegin;          // where its chain goes in chainOut (capacity nSlots)
	uint32_t nKeys;               // number of fragment positions of the read
	uint32_t fragBegin, nFrags;   // this read's fragments (for the "no anchors after a failed fragment" rule)
};

struct ChainCaps { uint32_t capAnchors, capEndpoints, capTable, capBack; };   // most anchors of a read; most entries (anchor x paths through its end node); widest path cover

// ---- seed glue on the device (gc_seedglue.hip) ----
struct GlueRead { uint32_t nSeeds, seedOff, nFrags, fragBegin, nSlots, slotBegin, failed, pad; };   // per read: where its seeds / fragments / anchor slots are
struct GlueElem;
struct GlueStaging {   // per-seed staging arrays (capacity: the sum over reads of their hits' occurrence counts), per-read window staging
	uint32_t *mPos, *mStartLo, *mStartHi;                       // per hit (a read's hits use the first slots of its seed range)
	uint32_t *sSeqPos, *sNode, *sOffset, *sGood, *sCluster;     // per seed, expansion order
	GlueElem* sortBuf; uint32_t* sortScratch;                   // reads beyond the LDS capacity: their sort array, and 3 words per seed occurrence + 64 per read for the wave sort's lists (gc_stdsort_wave.hpp) / the cluster sums
	uint32_t* winBuf;                                           // (l, sl, sr, first slot) per window, at 4 * winCapOff[read]
};

// ---- whole-read pass (K3-long) ----
struct LongSeed {    // a seed in goodness order (after OrderSeeds); 16 bytes: 3 M of them go up per 10 k-read batch
	uint32_t node;                  // forward start: split node of the seed base
	uint32_t seqPos;
	uint32_t goodness;
	uint16_t clusterSize;           // saturated (only compared with --seeds-clustersize)
	uint8_t offset;                 // offset of the seed base in the split node
	uint8_t pad;
};

struct LongJob {     // one read
	uint64_t maskOff;               // word offset of the read's match masks: [strand][base][maskWords]
	uint32_t maskWords, pad;
	uint64_t readOff;               // forward bases at bases[readOff..], reverse complement at bases[rcBase + readOff..]
	uint32_t readLen;
	uint32_t seedBegin, seedEnd;    // into the LongSeed array
	uint32_t alnBegin;              // this read's slots in the alignment output (capacity maxAlignments)
};

struct LongCell {    // merged trace cell in the reference's output coordinates
	int32_t node;                   // bigraph node id
	uint32_t offset;                // offset in the original node
	uint32_t seqPos;
	uint32_t nodeSwitch;
};

struct LongAln {     // one accepted alignment, in acceptance order
	uint32_t start, end;            // alignmentStart, alignmentEnd
	uint32_t score;
	uint32_t goodness;
	uint64_t traceOff;
	uint32_t traceLen;
	uint32_t pad;
};

struct LongReadResult { uint32_t nAlignments, seedsExtended, status, pad; };   // pad (r5): the read's flatten ties (extensions whose last-slice minimum was tied between nodes)

// round-based whole-read pass: per-read state carried between rounds, and the per-round work items
struct LongState { uint32_t si, nAln, extended, status, e2eScore, candBegin, candCount, pad1; };   // candBegin/candCount: this round's candidate seeds (pairs of work items)
struct LongWork {    // one direction of one seed extension
	uint64_t maskOff;             // word offset of this read+strand's four match-mask bit vectors
	uint32_t maskWords, startBit; // words per bit vector; read position (on that strand) of row 0
	uint32_t seqLen, node, offset;
	uint32_t read;
};
struct LongWorkResult { uint64_t traceOff; uint32_t traceLen, status; int32_t score; uint32_t pad; };   // pad bit 0: as ExtResult::pad

// ---- launchers (all asynchronous on `stream`) ------------------------------------------------------
void launchSeedLookup(hipStream_t stream, const SeedIndex& idx, const char* bases, const uint64_t* readOff, uint32_t nReads,
	uint64_t* matchCursor, uint32_t* readMatchOff, uint32_t* readMatchCount, uint2* matches, uint64_t matchCapacity, uint32_t* tmp, uint64_t totalBases, const uint32_t* chunkRead, const uint64_t* packed, const uint64_t* invalid);

// per read the capacity bound of its seed list and the exclusive scan of those bounds (readSeedOff[nReads + 1]); *total = their sum
void launchSeedCaps(hipStream_t stream, const SeedIndex& idx, uint32_t nReads, const uint8_t* invalidRead, const uint2* matches, const uint32_t* readMatchOff, const uint32_t* readMatchCount,
	uint32_t* readSeedCap, uint32_t* readSeedOff, unsigned long long* total);
// addMinimizers + orderSeedsByChaining + the sort by position + the fragment windows, one wave per read (cursors: [0] fragments, [1] anchor slots,
// [2] trace cells the fragment extensions may need, [3] most slots of a read, [4] most seeds of a window)
void launchSeedGlue(hipStream_t stream, const SeedIndex& idx, const DGraph& g, const uint64_t* readOff, uint32_t nReads, const uint8_t* invalidRead, const uint2* matches, const uint32_t* readMatchOff,
	const uint32_t* readMatchCount, const uint32_t* readSeedOff, const uint32_t* winCapOff, double density, uint32_t splitLen, uint32_t splitGap, bool longPass, const GlueStaging& st,
	uint32_t* perRead /* 6 x (nReads + 1) words of scratch */, LongSeed* longSeeds, FragSeed* readSeeds, Fragment* frags, uint32_t* fragFirstSeed, ReadChainJob* jobs, GlueRead* out, unsigned long long* cursors);
uint64_t glueElemBytes();
// test entry: arrays of (key << 32 | index) elements sorted by key with the wave-cooperative replay of std::sort (gc_stdsort_wave.hpp); scratch: 3 words per element + 64 per array
void launchTestStdSort(hipStream_t stream, unsigned long long* elems, const uint64_t* off, uint32_t nArrays, uint32_t* scratch, long depthLimit);

// the read batch as the fragment extension kernel sees it: per read and strand four match-mask bit vectors (gc_reads.hip)
struct FragReads { const uint64_t* masks; const uint64_t* maskOff; const uint32_t* maskWords; const uint64_t* readOff; uint64_t totalBases; };
// fragment extensions, one per lane in lockstep phases (gc_extend_frag.hip); what it declines carries EXT_OVERFLOW and is rerun by launchExtend (k_extend_slab)
uint32_t extendFragWaves();                        // resident waves of the kernel on this device: the grid, and the size of its item scratch
uint64_t extendFragScratchBytes(uint32_t waves);
void launchExtendFrag(hipStream_t stream, const DGraph& g, const CorrectnessTables* ct, int32_t bandwidth, const ExtItem* work, uint32_t nWork, const FragReads& reads, ExtResult* results,
	uint4* itemScratch, uint32_t scratchWaves, PoolCell* tracePool, unsigned long long* traceCursor, uint64_t traceCapacity, unsigned long long* counters, ExtSelection sel, unsigned long long* claim,
	uint32_t* retryList, unsigned long long* retryCount, unsigned long long* stamps = nullptr);   // (stamps: eight words of the profiling build, -DGC_FRAG_STAMPS) claim, retryCount: zeroed words of this launch's own; retryList [nWork]: the declined items, for launchExtend with a list selection
void launchBuildNodeRecs(hipStream_t stream, const DGraph& g, NodeRec* out);   // DGraph::nodeRec from the arrays already uploaded
uint64_t extendSlabBytes(const ExtendConfig& cfg);
uint32_t extendGridLanes(uint32_t nWork);
void launchExtend(hipStream_t stream, const DGraph& g, const CorrectnessTables* ct, const uint8_t* iupac, const ExtendConfig& cfg,
	const ExtItem* work, uint32_t nWork, const char* bases, ExtResult* results, uint8_t* scratch, uint64_t slabBytes,
	PoolCell* tracePool, unsigned long long* traceCursor, uint64_t traceCapacity, unsigned long long* counters,
	uint32_t retryStatus = 0, uint32_t retryLanes = 4096, ExtSelection sel = ExtSelection());   // retryStatus != 0: a small grid of `retryLanes` lanes reruns only the items whose result has that status (with the larger slabs of `cfg`)

void launchBuildAnchors(hipStream_t stream, const DGraph& g, const Fragment* frags, uint32_t nFrags, const FragSeed* seeds, const ExtResult* ext,
	const PoolCell* tracePool, int32_t splitLen, AnchorRec* anchors, uint32_t* fragStatus, uint32_t* fragExtended,
	uint32_t* pathPool, unsigned long long* pathCursor, uint64_t pathCapacity, AnchorRounds rounds = AnchorRounds(), uint32_t* readTies = nullptr);   // readTies [reads], zeroed by the caller: += the ExtResult::pad flags of the extensions the reference would have run

// gc_results.hip (r5): the result's dense anchor arrays made on the device. perRead[r] = (anchors kept, path words, seeds extended, bit 0 a fragment failed | bit 1 capacity),
// readOff[2 r] / [2 r + 1] = anchors / path words before read r (totals behind the last read and in hostTotals[0..1], pinned)
struct AnchorArrays { uint32_t *x, *y, *firstNode, *firstOffset, *firstSeqPos, *lastNode, *lastOffset, *lastSeqPos; int32_t* score; unsigned long long* pathOff; uint32_t* path; };
void launchAnchorCounts(hipStream_t stream, const ReadChainJob* jobs, uint32_t nReads, const Fragment* frags, const uint32_t* fragStatus, const uint32_t* fragExtended, const AnchorRec* anchors,
	uint4* perRead, uint32_t* readSlotEnd, unsigned long long* readOff, unsigned long long* hostTotals);
void launchAnchorCompact(hipStream_t stream, const ReadChainJob* jobs, uint32_t nReads, const AnchorRec* anchors, const uint32_t* pathPool, const uint32_t* readSlotEnd, const unsigned long long* readOff, const AnchorArrays& out);

uint64_t chainScratchBytes(const ChainCaps& caps);
uint32_t chainGridBlocks(uint32_t nReads);
uint32_t chainScratchBlocks(uint32_t nReads);
bool chainLdsLaunch(uint32_t fewestSlots, bool forceScratch);   // does launchChain run its LDS launch for a batch whose smallest read has this many anchor slots?
void launchChain(hipStream_t stream, const DGraph& g, const ReadChainJob* jobs, uint32_t nReads, const AnchorRec* anchors, const Fragment* frags, const uint32_t* fragStatus,
	int32_t splitLen, int32_t splitGap, ChainCaps caps, uint8_t* scratch, uint32_t* chainOut, uint32_t* chainLen, unsigned long long* chainScore, uint32_t* chainStatus, bool forceScratch = false,
	uint32_t fewestSlots = 0);   // fewestSlots: the batch's smallest read in anchor slots (0: unknown) - when no read can fit an LDS class that launch is skipped

void launchLongPass(hipStream_t stream, const DGraph& g, const CorrectnessTables* ct, const uint8_t* iupac, const ExtendConfig& cfg, const LongJob* jobs, uint32_t nReads,
	const LongSeed* seeds, const char* bases, uint64_t rcBase, uint32_t minClusterSize, uint32_t maxAlignments, uint8_t* scratch, uint64_t slabBytes,
	LongCell* cellPool, unsigned long long* cellCursor, uint64_t cellCapacity, LongAln* alns, LongReadResult* results, unsigned long long* counters);
uint64_t longSlabBytes(const ExtendConfig& cfg);
uint64_t longWaveWordsPerLane(const ExtendConfig& cfg);
void launchLongInit(hipStream_t stream, const LongJob* jobs, uint32_t nReads, LongState* state);
void launchLongSelect(hipStream_t stream, const DGraph& g, const LongJob* jobs, uint32_t nReads, const LongSeed* seeds, uint64_t rcBase, uint32_t minClusterSize, uint32_t maxCandidates,
	LongState* state, const LongAln* alns, const LongCell* cellPool, LongWork* work, uint32_t* workLen, uint32_t* candSeed, unsigned long long* workCount, uint64_t workCapacity);
uint32_t longExtendTeamSize(uint32_t nWork);
void launchLongExtend(hipStream_t stream, const DGraph& g, const CorrectnessTables* ct, const uint64_t* masks, const ExtendConfig& cfg, const LongWork* work, const uint32_t* order, uint32_t nWork,
	unsigned long long* scratch, uint32_t lanes, uint32_t blocks, unsigned long long* tracePool, unsigned long long* traceCursor, uint64_t traceCapacity, LongWorkResult* results, unsigned long long* counters,
	unsigned long long* nextSlot, uint32_t retryStatus = 0, const unsigned long long* nWorkOnDevice = nullptr, uint32_t* capListOut = nullptr, unsigned long long* capCountOut = nullptr,
	bool gridCoversCount = false);   // nWorkOnDevice: `order` is a list whose length only the device knows (then nWork is its upper bound; gridCoversCount: blocks x lanes >= that bound, no fetch loop needed)
#ifdef GC_EXPERIMENTS   // measured and rejected alternatives of the whole-read pass: only in `make experiments`
// the whole inter-round step in one launch: merge of the previous round + select + execution order + work count to device and host (k_long_round)
void launchLongRound(hipStream_t stream, const DGraph& g, const LongJob* jobs, uint32_t nReads, const LongSeed* seeds, uint32_t minClusterSize, uint32_t round, uint32_t forceCand, uint32_t gridLimit,
	LongState* state, LongAln* alns, LongCell* cellPool, unsigned long long* cellCursor, uint64_t cellCapacity, uint32_t maxAlignments, LongWork* work, uint32_t* workLen, uint32_t* candSeed,
	const LongWorkResult* results, const unsigned long long* tracePool, unsigned long long* cursorSets, unsigned long long* roundInfo, unsigned long long* ticket, uint32_t* order, uint32_t maxLen, uint32_t orderMode,
	unsigned long long* hostInfo, uint64_t workCapacity);
// the same extensions one per LANE with the plain-layout core and a per-lane HBM slab (k_long_extend_lane: the layout measurement of DESIGN.md §11, GC_LONG_LANE=1)
void launchLongExtendLane(hipStream_t stream, const DGraph& g, const CorrectnessTables* ct, const uint64_t* masks, const ExtendConfig& cfg, const LongWork* work, const uint32_t* order, uint32_t nWork,
	uint8_t* scratch, uint64_t scratchBytes, unsigned long long* tracePool, unsigned long long* traceCursor, uint64_t traceCapacity, LongWorkResult* results, unsigned long long* counters);
// the same extensions one per LANE as per-lane state machines (gc_sm.hip); what outgrows its tables answers EXT_SM_DECLINED (6) and is rerun by launchLongExtend
uint64_t longSmSlabBytes(const ExtendConfig& cfg);
void launchLongExtendSm(hipStream_t stream, const DGraph& g, const CorrectnessTables* ct, const uint64_t* masks, const ExtendConfig& cfg, const LongWork* work, const uint32_t* order, uint32_t nWork,
	uint8_t* scratch, uint64_t scratchBytes, unsigned long long* tracePool, unsigned long long* traceCursor, uint64_t traceCapacity, LongWorkResult* results, unsigned long long* counters, unsigned long long* nextSlot);
void launchLongRetryList(hipStream_t stream, const LongWorkResult* results, uint32_t nWork, uint32_t status, uint32_t* list, unsigned long long* listCount);   // retryStatus != 0: only work items whose result has that status (e.g. EXT_LDS_CAP) are run
#endif
void launchLongMerge(hipStream_t stream, const DGraph& g, const LongJob* jobs, uint32_t nReads, const LongSeed* seeds, const uint32_t* candSeed, const LongWorkResult* results,
	const unsigned long long* tracePool, uint32_t maxAlignments, LongState* state, LongAln* alns, LongCell* cellPool, unsigned long long* cellCursor, uint64_t cellCapacity);
// ---- path sequences + NW edit distances (gc_editdist.hip)
struct PathSeqJob {   // one path to spell out as letters
	uint64_t srcOff;              // first LongCell (whole-read alignment) or first path node (stitched chain)
	uint64_t outOff;              // where its letters go
	uint32_t count;               // cells / nodes
	uint32_t outCap;              // letters reserved
	uint32_t firstOffset, lastOffset;   // stitched chain only: offsets of the first and last base
};
struct EdRead { uint64_t readOff, eqOff; uint32_t len, words; };   // read bases + its exact-match masks [A,C,G,T][words]
struct EdPair {
	uint64_t lettersOff;          // path letters (matrix columns)
	uint32_t m;                   // their number, unless a length array is given (then lengths[lenIndex])
	uint32_t lenIndex;
	uint32_t read;                // index into EdRead (matrix rows)
	uint32_t k;                   // first band half-width to try
};
void launchLongPathSeq(hipStream_t stream, const DGraph& g, const PathSeqJob* jobs, uint32_t nJobs, const LongCell* cellPool, char* letters, uint32_t* outLen);
// ---- chain stitching on the device (gc_stitch.hip) ----
struct StitchInfo {   // the longest stitched piece of a read's chain
	uint64_t start;               // its first node in the dense output
	uint64_t cells;               // bases on it (size of pathToTrace's result); 0: no chain
	uint32_t len;                 // nodes
	uint32_t firstOffset, lastOffset;
	uint32_t status;              // != 0: did not fit the kernel's tables (1 piece or region full, 2 bridge search too wide, 3 unusable anchor
	                              // record, 4 output array full), the host stitches this read
	uint32_t scoreSum;            // (r5) the chain's anchors' alignment scores added up: the edits the fragments' extensions found, from which the chain's NW edit distance gets its first band
	uint32_t pad;
};
// slotOf: [total anchor slots] scratch; regions: stitchRegionWords(total slots, reads) words of scratch; dense: the results,
// stitchDenseWords(...) words, filled from *denseCursor (zeroed by the caller) upwards; StitchInfo.start indexes dense
void launchStitch(hipStream_t stream, const DGraph& g, const ReadChainJob* jobs, uint32_t nReads, const AnchorRec* anchors, const Fragment* frags, const uint32_t* fragStatus,
	const uint32_t* chainOut, const uint32_t* chainLen, const uint32_t* chainStatus, const uint32_t* pathPool, uint64_t pathCapacity, long long colinearGap, uint32_t* slotOf,
	uint32_t* regions, uint32_t* dense, uint64_t denseCap, unsigned long long* denseCursor, StitchInfo* info,
	uint32_t setMax = 0, uint32_t bfsCap = 0, int sizeClass = 0, unsigned long long* spill = nullptr);   // smaller table limits (tests: forces reads onto the host path); 0 = the kernel's own. sizeClass: 0 default, 3 (r5) long reads: node set and wide searches in `spill`
uint64_t stitchSpillWordsPerBlock();       // 64-bit words of HBM scratch per block of a class-3 launch ...
uint32_t stitchSpillBlocks(uint32_t nReads);   // ... and its blocks
uint64_t stitchRegionWords(uint64_t totalSlots, uint64_t nReads);
uint64_t stitchDenseWords(uint64_t totalSlots, uint64_t nReads);
// srcOff with bit 63 set reads its nodes from altNodes (host-stitched reads) instead of pathNodes
void launchChainPathSeq(hipStream_t stream, const DGraph& g, const PathSeqJob* jobs, uint32_t nJobs, const uint32_t* pathNodes, const uint32_t* altNodes, char* letters, uint32_t* outLen);
uint32_t editDistanceMaxK(uint32_t unitBlocks);
// the whole matrix by one workgroup of `threads` (>= the read's 64-row blocks, <= editDistanceBlockMaxRows() / 64): pairs whose band would cover most of it
void launchEditDistanceBlock(hipStream_t stream, uint32_t threads, const EdPair* pairs, uint32_t nPairs, const EdRead* reads, const char* bases, const uint64_t* eqMasks,
	const char* letters, const uint32_t* lettersLen, int64_t* outDistance);
uint32_t editDistanceBlockMaxRows();
// several pairs per wave (2: teams of 32 lanes, 3: teams of 21), unit of one block; editDistanceTeamMaxK: the first band that no longer fits the team
uint32_t editDistanceTeamMaxK(uint32_t pairsPerWave);
void launchEditDistanceTeam(hipStream_t stream, uint32_t pairsPerWave, const EdPair* pairs, uint32_t nPairs, const EdRead* reads, const char* bases, const uint64_t* eqMasks,
	const char* letters, const uint32_t* lettersLen, int64_t* outDistance);
void launchEditDistance(hipStream_t stream, uint32_t unitBlocks, const EdPair* pairs, uint32_t nPairs, const EdRead* reads, const char* bases, const uint64_t* eqMasks,
	const char* letters, const uint32_t* lettersLen, int64_t* outDistance);
// ---- output encoding on the device (gc_output.hip): GAF path + CIGAR text and the vg::Path wire bytes of final alignments
struct OutNames { const uint32_t* nameOff; const char* nameBytes; };   // [bigraph node ids + 1]: the GFA name of every node's segment (empty: the encoders print id / 2)
struct OutJob {      // one final alignment
	uint64_t cellOff;             // its merged trace in the LongCell pool
	uint64_t readOff;             // the read's forward bases
	uint32_t cellLen, readLen;
	uint32_t flags;               // 1: CIGAR with M instead of = / X, 2: GAF pieces wanted, 4: vg::Path bytes wanted
	uint32_t pad;
};
struct OutRec {      // what the counting pass leaves per job (the writing pass sets steps = ~0 when it did not land on these sizes)
	uint64_t nodePathLen, nodePathStart, nodePathEnd;
	uint32_t pathTextLen, cigarLen, vgLen, steps;
	uint32_t matches, mismatches, insertions, deletions;
	uint32_t readStart, readEnd;
};
// counting pass + placement: recs, offsets[3][nJobs + 1] (path text, CIGAR, vg bytes; the last entries and totals[0..2] are the sums); mapSizeAtCell: one word per pool cell
void launchOutCount(hipStream_t stream, const DGraph& g, const OutNames& names, const uint8_t* iupac, const OutJob* jobs, uint32_t nJobs, const LongCell* cellPool, const char* bases,
	OutRec* recs, uint64_t* offsets, uint32_t* mapSizeAtCell, unsigned long long* totals);
void launchOutWrite(hipStream_t stream, const DGraph& g, const OutNames& names, const uint8_t* iupac, const OutJob* jobs, uint32_t nJobs, const LongCell* cellPool, const char* bases,
	OutRec* recs, const uint64_t* offsets, uint32_t* mapSizeAtCell, char* pathText, char* cigarText, uint8_t* vgBytes);
// ---- DEFLATE of independent byte streams (gc_deflate.hip): one dynamic-Huffman block of literals per stream (stored blocks where that is smaller or the code
// would be deeper than 15 bits). plan[s] = { deflate bytes, mode }, lens[260 * s + symbol]; the caller places stream s at a 4-byte aligned outOff[s] with room for plan[s].x rounded up to 4
void launchDeflatePlan(hipStream_t stream, const uint8_t* raw, const uint64_t* rawOff, uint32_t nStreams, uint8_t* lens, uint2* plan);
void launchDeflateWrite(hipStream_t stream, const uint8_t* raw, const uint64_t* rawOff, uint32_t nStreams, const uint8_t* lens, const uint2* plan, uint8_t* out, const uint64_t* outOff);
// r6: the same with LZ77 matches in front of the Huffman stage (one probe of a hash of 4-byte prefixes per position, greedy); lens: deflateLzLensStride() bytes per stream
uint32_t deflateLzLensStride();
void launchDeflateLzPlan(hipStream_t stream, const uint8_t* raw, const uint64_t* rawOff, uint32_t nStreams, uint8_t* lens, uint2* plan);
void launchDeflateLzWrite(hipStream_t stream, const uint8_t* raw, const uint64_t* rawOff, uint32_t nStreams, const uint8_t* lens, const uint2* plan, uint8_t* out, const uint64_t* outOff);
// ---- minimizer index construction on the device (gc_minimizer.hip): the graph's window minimizers as (k-mer << 34 | reversed
// arrival index, packed position) pairs, sorted; returns their number (~0 on failure: the caller builds on the host), arrays are hipMalloc'd
uint64_t buildMinimizerPairsDevice(const DGraph& g, const int32_t* idOrderDev, uint32_t nIds, uint32_t k, uint32_t w, uint64_t** outKeys, uint64_t** outValues);
// ---- fragment pass work items from the host's sorted seeds and windows (gc_kernels.hip)
void launchBuildFragmentWork(hipStream_t stream, const DGraph& g, const Fragment* frags, const uint32_t* fragFirstSeed, uint32_t nFrags, const FragSeed* readSeeds, const uint64_t* readOffsets,
	uint64_t totalBases, uint32_t splitLen, FragSeed* fragSeeds, ExtItem* work, ExtResult* results = nullptr);
// ---- read batch preparation (gc_reads.hip): reverse-complement strand, match-mask / exact-match bit vectors, 2-bit packing, all from the raw bases
void launchPackReads(hipStream_t stream, const uint64_t* readOff, uint32_t nReads, uint64_t totalBases, char* bases, const uint64_t* maskOff, const uint32_t* maskWords, uint64_t* masks,
	const uint64_t* eqOff, uint64_t* eqMasks, uint8_t* readInvalid, uint64_t* packed, uint64_t* invalidBits, uint32_t* chunkRead);
// ---- alignment path of (stitched chain path, read): edlib's EDLIB_TASK_PATH (gc_edpath.hip) ----
struct EdPathJob {
	uint64_t queryOff;            // path letters (rows), in `letters`
	uint64_t targetOff;           // read bases (columns), in `bases`
	uint64_t opsOff;              // where the op string goes (capacity queryLen + targetLen)
	uint32_t queryLen, targetLen;
	int32_t best;                 // their NW edit distance (from k_edit_distance)
	uint32_t pad;
};
#define ED_PATH_LEAF_CELLS 52432ull   // blocks x columns of a directly traced sub-problem: 20 * cells + 8 * columns < 1 MB (edlib/src/edlib.cpp:1204-1207)
uint64_t editPathScratchBytes(uint32_t maxQ, uint32_t maxT);   // per resident wave
uint32_t editPathGridBlocks(uint32_t nJobs);
// ops: 0 match, 1 path letter alone, 2 read base alone, 3 mismatch; opsLen[i] = 0 when edlib would return no alignment
void launchEditPath(hipStream_t stream, const EdPathJob* jobs, uint32_t nJobs, const char* letters, const char* bases, uint8_t* scratch, uint32_t maxQ, uint32_t maxT,
	uint8_t* opsOut, uint32_t* opsLen);
void launchLongOrder(hipStream_t stream, const uint32_t* workLen, const unsigned long long* workCount, uint32_t* order, uint32_t maxLen, uint32_t mode);
void launchPublish(hipStream_t stream, const unsigned long long* src, unsigned long long* dst, uint32_t nWords);
void launchZeroWords(hipStream_t stream, unsigned long long* dst, uint32_t nWords);   // nWords <= 64
void launchLongFinish(hipStream_t stream, uint32_t nReads, const LongState* state, LongReadResult* results);


} // namespace gcdev
