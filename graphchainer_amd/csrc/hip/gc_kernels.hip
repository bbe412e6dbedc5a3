// HIP kernels of the GraphChainer hot path for gfx950 (MI355X). See gc_device.hpp for the device data model
// and DESIGN.md for the mapping rationale and the per-kernel roofline accounting.
#include "gc_kernels.hpp"
#include "gc_device_wave.hpp"
#include <cstdlib>

namespace gcdev {

// =====================================================================================================
// K1 - minimizer seed lookup. reference: MinimizerSeeder::getSeeds / iterateKmers,
// src/MinimizerSeeder.cpp:59-102,522-544. One wave per read; lanes sweep 64 read positions at a time.
// For every read position whose 15-mer is valid and indexed with count < maxCount, and that passes the
// reference's "same k-mer as the previous position" thinning rule (:91), emits (pos, keyIndex).
// Output order inside a read is by position (ballot-rank compaction), which the host's std::sort by count
// then consumes exactly like the reference does (:497).
// =====================================================================================================

__device__ __forceinline__ int baseCode(uint8_t c)
{
	switch (c) {
		case 'a': case 'A': return 0;
		case 'c': case 'C': return 1;
		case 'g': case 'G': return 2;
		case 't': case 'T': return 3;
	}
	return -1;
}

__device__ __forceinline__ uint32_t hashKmer(uint64_t kmer)
{
	kmer *= 0x9E3779B97F4A7C15ull;
	return (uint32_t)(kmer >> 32);
}

// 9 of 10 read k-mers are not graph minimizers; a probe of the 0.1-0.5 GB table is a random HBM access each, a bit of the
// 32 MB filter is a cache hit. A clear bit proves the k-mer is not a key (a set bit proves nothing: ~2 % false positives).
__device__ __forceinline__ uint32_t filterBit(uint64_t kmer, uint32_t shift) { return (uint32_t)((kmer * 0xD6E8FEB86659FD93ull) >> shift); }

// returns key index or 0xffffffff. Open addressing, linear probing; slot = {kmer:32, index:32}.
__device__ __forceinline__ uint32_t lookupKmer(const SeedIndex& idx, uint64_t kmer)
{
	const uint32_t fb = filterBit(kmer, idx.filterShift);
	if (!((idx.filter[fb >> 5] >> (fb & 31)) & 1u)) return 0xffffffffu;
	uint32_t h = hashKmer(kmer) & idx.tableMask;
	while (true) {
		uint64_t slot = idx.table[h];
		if (slot == ~0ull) return 0xffffffffu;
		if ((uint32_t)(slot >> 32) == (uint32_t)kmer && (!idx.wideKmers || idx.wideKmers[(uint32_t)slot] == kmer)) return (uint32_t)slot;
		h = (h + 1) & idx.tableMask;
	}
}

// K1a: one thread per read base. For the k-mer that ends at the base: the reference's thinning rule, the index probe and the
// frequency cut; writes key index + 1 (0 = nothing to emit) into tmp[global position].
__global__ void __launch_bounds__(256) k_seed_probe(SeedIndex idx, const char* __restrict__ bases, const uint64_t* __restrict__ readOff, const uint32_t* __restrict__ chunkRead, const uint64_t* __restrict__ packed, const uint64_t* __restrict__ invalid, uint32_t nReads, uint64_t totalBases, uint32_t* __restrict__ tmp)
{
	const int k = idx.k;
	const int realWindow = idx.w - idx.k + 1;
	const uint64_t mask = ~(~0ull << (2 * k));
	for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < totalBases; p += (uint64_t)gridDim.x * blockDim.x) {
		// the read this base belongs to: last r with readOff[r] <= p (chunkRead: the read of the first base of every 64-base chunk)
		uint32_t lo = chunkRead[p >> 6];
		while (lo + 1 < nReads && readOff[lo + 1] <= p) lo++;
		const char* seq = bases + readOff[lo];
		const int pos = (int)(p - readOff[lo]);
		uint32_t found = 0;
		if (pos >= k - 1) {
			// k-mer ending at pos, from the 2-bit big-endian packing of the concatenated reads (base j in bits 63-2(j%32), 62-2(j%32)
			// of word j/32): the k bases are one bit field of a two-word window; a second bit vector marks the non-ACGT bases
			const uint64_t w = p >> 5;
			const uint64_t lo64 = packed[w], hi64 = w ? packed[w - 1] : 0ull;
			const uint32_t sh = 2u * (31u - (uint32_t)(p & 31));
			const uint64_t kmer = (sh ? ((lo64 >> sh) | (hi64 << (64 - sh))) : lo64) & mask;
			const uint64_t iw = p >> 6;
			const uint64_t ilo = invalid[iw], ihi = iw ? invalid[iw - 1] : 0ull;
			const uint32_t ish = 63u - (uint32_t)(p & 63);   // same big-endian convention: base j in bit 63-(j%64)
			const uint64_t window = ish ? ((ilo >> ish) | (ihi << (64 - ish))) : ilo;
			const bool valid = (window & ((1ull << k) - 1)) == 0;
			if (valid) {
				// thinning (:91): inside a streak of identical consecutive k-mers only every realWindow-th is emitted.
				// Consecutive k-mers are identical only in a homopolymer; walk back to the streak start.
				int streakStart = pos;
				int c0 = baseCode((uint8_t)seq[pos]);
				bool homopolymer = (kmer == (uint64_t)c0 * (mask / 3));
				if (homopolymer) {
					while (streakStart - k >= 0 && baseCode((uint8_t)seq[streakStart - k]) == c0) streakStart--;
				}
				bool emit = ((pos - streakStart) % realWindow) == 0;
				if (emit) {
					uint32_t key = lookupKmer(idx, kmer);
					if (key != 0xffffffffu) {
						uint32_t count = (uint32_t)(idx.startPos[key + 1] - idx.startPos[key]);
						if (count < idx.maxCount) found = key + 1;
					}
				}
			}
		}
		tmp[p] = found;
	}
}

// K1b: one wave per read: counts the read's hits, reserves a contiguous output range and compacts them in position order.
__global__ void __launch_bounds__(64) k_seed_compact(const uint64_t* __restrict__ readOff, uint32_t nReads, const uint32_t* __restrict__ tmp,
	uint64_t* __restrict__ matchCursor, uint32_t* __restrict__ readMatchOff, uint32_t* __restrict__ readMatchCount, uint2* __restrict__ matches, uint64_t matchCapacity)
{
	const int lane = threadIdx.x;
	for (uint32_t r = blockIdx.x; r < nReads; r += gridDim.x) {
		const int len = (int)(readOff[r + 1] - readOff[r]);
		const uint32_t* mine = tmp + readOff[r];
		uint32_t total = 0;
		for (int base = 0; base < len; base += 64) {
			int pos = base + lane;
			total += (uint32_t)__popcll(__ballot(pos < len && mine[pos] != 0));
		}
		// reserve a contiguous output range for this read
		uint64_t outBase = 0;
		if (lane == 0) outBase = atomicAdd((unsigned long long*)matchCursor, (unsigned long long)total);
		outBase = __shfl(outBase, 0);
		if (lane == 0) { readMatchOff[r] = (uint32_t)outBase; readMatchCount[r] = total; }
		if (outBase + total > matchCapacity) continue;   // host checks the cursor and retries with a larger buffer
		uint32_t written = 0;
		for (int base = 0; base < len; base += 64) {
			int pos = base + lane;
			uint32_t found = pos < len ? mine[pos] : 0;
			unsigned long long ballot = __ballot(found != 0);
			if (found) {
				uint32_t rank = (uint32_t)__popcll(ballot & ((1ull << lane) - 1));
				matches[outBase + written + rank] = make_uint2((uint32_t)pos, found - 1);
			}
			written += (uint32_t)__popcll(ballot);
		}
	}
}

// =====================================================================================================
// K3 - seed extension. One lane per extension (see gc_device.hpp). A persistent grid strides over the work
// items; every lane owns a scratch slab in HBM; finished traces are appended to a shared pool.
// reference: GraphAlignerBitvectorBanded::getReverseTraceFromSeed, src/GraphAlignerBitvectorBanded.h:46-71.
// =====================================================================================================

__device__ __forceinline__ LaneScratch laneScratch(uint8_t* slab, const ExtendConfig& cfg)
{
	LaneScratch sc;
	uint8_t* p = slab;
	sc.slices = (SliceInfo*)p;   p += sizeof(SliceInfo) * cfg.maxSlices;
	sc.items = (NodeItem*)p;     p += sizeof(NodeItem) * cfg.maxItems;
	sc.pending = (Pending*)p;    p += sizeof(Pending) * cfg.maxPending;
	sc.columns = (WCol*)p;       p += sizeof(WCol) * 64;
	sc.colMask = 63; sc.colStride = 1;
	sc.trace = (TraceCell*)p;    p += sizeof(TraceCell) * cfg.maxTrace;
	sc.itemNodes = (uint32_t*)p;
	return sc;
}

#ifndef GC_EXTEND_RING
#define GC_EXTEND_RING 8   // columns per lane of the backtrace ring in LDS (power of two; 0: the whole tile in the lane's HBM slab, as before r4)
#endif
// 4 waves per SIMD (<= 128 VGPRs; the kernel wanted 131 and ran 3): it waits on memory 56 % of the time, so the extra wave
// pays for the 4 spilled registers: 34.2 -> 29.0 ms alone on cfg2 (5 or 6 waves spill 57 / 196 registers and lose).
#define GC_EXTEND_SLAB_PARAMS DGraph g, const CorrectnessTables* __restrict__ ct, const uint8_t* __restrict__ iupac, ExtendConfig cfg, 	const ExtItem* __restrict__ work, uint32_t nWork, const char* __restrict__ bases, ExtResult* __restrict__ results, 	uint8_t* __restrict__ scratch, uint64_t slabBytes, PoolCell* __restrict__ tracePool, unsigned long long* __restrict__ traceCursor, uint64_t traceCapacity, 	unsigned long long* __restrict__ counters, uint32_t retryStatus, ExtSelection sel
__device__ __forceinline__ void extendSlabBody(GC_EXTEND_SLAB_PARAMS)
{
#if defined(GC_EXTEND_PRIO) && GC_EXTEND_PRIO
	__builtin_amdgcn_s_setprio(GC_EXTEND_PRIO);   // (experiment, -DGC_EXTEND_PRIO=3: the fragment extension's waves ahead of the whole-read kernel's - 149.9 / 149.2 / 152.3 ms per batch against 154.0 / 147.2 / 151.6, `gpurun_out/r4_extprio`: no effect, off)
#endif
	const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t stride = gridDim.x * blockDim.x;
	LaneScratch sc = laneScratch(scratch + (uint64_t)tid * slabBytes, cfg);
#if GC_EXTEND_RING
	// r4: the backtrace's recomputed columns in LDS - a ring of the last GC_EXTEND_RING columns per lane, lane-interleaved - instead of 64 columns in the lane's HBM slab
	__shared__ WCol columnRing[GC_EXTEND_RING * 64];
	sc.columns = columnRing + threadIdx.x;
	sc.colMask = GC_EXTEND_RING - 1; sc.colStride = 64;
#endif
	ExtCounters cnt {};
	// which work items: all of them, the two extensions of every fragment's first seed, or a list written by k_build_anchors (count on the device)
	const uint32_t nSelected = sel.mode == 0 ? nWork : sel.mode == 1 ? 2 * sel.nFrags : (uint32_t)*sel.listCount;
	for (uint32_t at = tid; at < nSelected; at += stride) {
		const uint32_t w = sel.mode == 0 ? at : sel.mode == 1 ? 2 * sel.frags[at >> 1].seedBegin + (at & 1u) : sel.list[at];
		if (retryStatus != 0 && results[w].status != retryStatus) continue;   // retry launch (larger slabs): only the items the first launch gave up on
		ExtItem it = work[w];
		uint32_t nTrace = 0;
		int32_t score = 0;
		uint32_t status = extendSeed(g, *ct, iupac, cfg, sc, bases + it.seqOff, (int)it.seqLen, it.node, it.offset, nTrace, score, cnt);
		ExtResult res;
		res.status = status;
		res.score = score;
		res.traceLen = 0;
		res.traceOff = 0;
		res.pad = cnt.flattenTie;   // r5: the extension's last-slice minimum was attained in more than one node (k_build_anchors sums what the reference would have run, per read)
		if (status == EXT_OK) {
			unsigned long long base = atomicAdd(traceCursor, (unsigned long long)nTrace);
			if (base + nTrace <= traceCapacity) {
				for (uint32_t i = 0; i < nTrace; i++) { const TraceCell c = sc.trace[i]; tracePool[base + i] = PoolCell { c.node, (uint16_t)c.offsetAndSwitch, (int16_t)c.seqPos }; }
				res.traceOff = base;
				res.traceLen = nTrace;
			} else {
				res.status = EXT_OVERFLOW;
			}
		}
		results[w] = res;
	}
	// one set of atomics per lane that did work
	if (cnt.extensions) {
		atomicAdd(&counters[0], cnt.dpTiles);
		atomicAdd(&counters[1], cnt.recomputeTiles);
		atomicAdd(&counters[2], cnt.columnSteps);
		atomicAdd(&counters[3], cnt.traceItems);
		atomicAdd(&counters[4], cnt.extensions);
		atomicAdd(&counters[5], cnt.backtraceTiles);
	}
}

// 4 waves per SIMD (<= 128 VGPRs): everything, when the lockstep kernel is switched off (GC_EXTEND_SLAB=1)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) k_extend_slab(GC_EXTEND_SLAB_PARAMS)
{
	extendSlabBody(g, ct, iupac, cfg, work, nWork, bases, results, scratch, slabBytes, tracePool, traceCursor, traceCapacity, counters, retryStatus, sel);
}
// =====================================================================================================
// K3b - fragment post-pass: merges the two one-way traces of every seed, replays the reference's serial
// "seed already lies on an earlier alignment" filter inside each fragment, and emits anchors.
// reference: AlignOneWay src/GraphAligner.h:114-203 (non-sloppy), exactAlignmentPart :407-461,
// getAlignmentFromSeed :567-626, anchor construction src/Aligner.cpp:706-729. One lane per fragment.
// =====================================================================================================

__device__ __forceinline__ void twinOf(const DGraph& g, uint32_t node, uint32_t offset, uint32_t& twinNode, uint32_t& twinOffset);

struct MergedView {   // virtual view of a seed's merged trace (backward cells, then forward cells) in forward-strand split coords
	const PoolCell* bw; uint32_t nBw;   // backward device trace without its final row -1 cell (nBw cells used)
	const PoolCell* fw; uint32_t nFw;   // forward device trace (used in reverse order)
	int32_t p;                            // seed position inside the fragment
	__device__ uint32_t size() const { return nBw + nFw; }
};

// cell i of the merged trace: (split node, offset in split node, seqPos in fragment)
__device__ inline void mergedCell(const DGraph& g, const MergedView& v, uint32_t i, uint32_t& node, uint32_t& offset, int32_t& seqPos)
{
	if (i < v.nBw) {
		const PoolCell& c = v.bw[i];
		// backward rows count away from the seed: row b -> fragment position p-1-b (fixReverseTraceSeqPosAndOrder, :543-565)
		seqPos = v.p - 1 - c.seqPos;
		// reverse-strand twin of the cell (GetReversePosition + GetUnitigNode)
		uint32_t off = c.offsetAndSwitch & 255u;
		int32_t id = g.nodeIDs[c.node];
		uint32_t orig = g.nodeOffset[c.node] + off;
		uint32_t rev = g.origSize[id] - 1 - orig;
		uint32_t twin = g.lookup[g.lookupOff[id ^ 1] + rev / 64];
		node = twin;
		offset = rev - g.nodeOffset[twin];
	} else {
		const PoolCell& c = v.fw[v.nFw - 1 - (i - v.nBw)];
		seqPos = v.p + 1 + c.seqPos;   // row -1 -> p
		node = c.node;
		offset = c.offsetAndSwitch & 255u;
	}
}

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 8))) k_build_anchors(DGraph g, const Fragment* __restrict__ frags, uint32_t nFrags, const FragSeed* __restrict__ seeds,
	const ExtResult* __restrict__ ext, const PoolCell* __restrict__ tracePool, int32_t splitLen,
	AnchorRec* __restrict__ anchors, uint32_t* __restrict__ fragStatus, uint32_t* __restrict__ fragExtended,
	uint32_t* __restrict__ pathPool, unsigned long long* __restrict__ pathCursor, uint64_t pathCapacity, AnchorRounds rounds, uint32_t* __restrict__ readTies)
{
	GC_RAISE_PRIO();
	// Lazy extension (rounds.lazy): a seed's two extensions run only when the reference would run them, i.e. when the seed does not lie on an
	// earlier alignment of its fragment (on cfg2 more than half of the seeds do). Round 0 has the first seed of every fragment extended and
	// walks all fragments; a fragment that reaches a seed it must extend and whose extensions have not run yet parks itself - the seed's
	// two work items go to the next round's list, the fragment to the next round's pending list - and resumes there in the next round.
	uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
	if (rounds.lazy && rounds.round > 0) {
		if (f >= (uint32_t)*rounds.pendingCount) return;
		f = rounds.pending[f];
	} else if (f >= nFrags) return;
	Fragment fr = frags[f];
	uint32_t status = 0;       // 0 ok, 1 the reference would throw in this fragment, 2 capacity overflow
	uint32_t extended = 0;
	// (extensions consumed in this launch whose flattenLastSliceEnd minimum was tied between nodes - ExtResult::pad, one in thirty - are added to the read's count where they are met:
	// a counter kept across the seed loop cost six registers and a wave per SIMD)
	uint32_t nSeeds = fr.seedEnd - fr.seedBegin;
	uint32_t firstSeed = 0;
	if (rounds.lazy && rounds.round > 0) { firstSeed = rounds.fragNext[f]; extended = fragExtended[f]; }
	else for (uint32_t k = 0; k < nSeeds; k++) anchors[fr.seedBegin + k].valid = 0;
	for (uint32_t k = firstSeed; k < nSeeds && status == 0; k++) {
		uint32_t sIdx = fr.seedBegin + k;
		FragSeed sd = seeds[sIdx];
		int32_t p = (int32_t)sd.seqPos - (int32_t)fr.l;
		// --- filter: does the seed cell lie on an earlier accepted alignment of this fragment? (:163-173)
		bool skip = false, haveTwin = false;
		uint32_t twinNode = 0, twinOffset = 0;
		for (uint32_t a = 0; a < k && !skip && status == 0; a++) {
			const AnchorRec& prev = anchors[fr.seedBegin + a];
			if (!prev.valid) continue;
			if (!(prev.lastSeqPos > prev.firstSeqPos)) { status = 1; break; }   // assert at :412
			if ((int32_t)prev.lastSeqPos < p || (int32_t)prev.firstSeqPos > p) continue;
			FragSeed ps = seeds[fr.seedBegin + a];
			const ExtResult& pb = ext[2 * (size_t)(fr.seedBegin + a)];
			const ExtResult& pf = ext[2 * (size_t)(fr.seedBegin + a) + 1];
			MergedView v;
			v.p = (int32_t)ps.seqPos - (int32_t)fr.l;
			bool hasB = pb.status == EXT_OK && v.p > 0, hasF = pf.status == EXT_OK && v.p < splitLen - 1;
			v.bw = tracePool + pb.traceOff; v.nBw = hasB ? (hasF ? pb.traceLen - 1 : pb.traceLen) : 0;
			v.fw = tracePool + pf.traceOff; v.nFw = hasF ? pf.traceLen : 0;
			// Is (seed node, seed offset, p) a cell of this alignment? Only the cells of read position p can be, and a one-way trace's rows never
			// increase along it: a binary search finds them (the reference searches by seqPos too, :415-437; r2 walked every cell of every earlier
			// alignment - 23 GB fetched per 10 k reads, each backward cell through the five gathers of the strand flip). Backward cells are
			// compared on their own strand, against the seed's reverse-strand twin (the flip is a bijection), computed once per seed.
			if (v.nBw > 0) {
				const int32_t row = v.p - 1 - p;   // backward row of read position p
				if (!haveTwin) { twinOf(g, sd.node, sd.offset, twinNode, twinOffset); haveTwin = true; }
				uint32_t lo = 0, hi = v.nBw;      // first cell with seqPos <= row
				while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (v.bw[mid].seqPos > row) lo = mid + 1; else hi = mid; }
				for (uint32_t i = lo; i < v.nBw && !skip; i++) {
					const PoolCell c = v.bw[i];
					if (c.seqPos != row) break;
					if (c.node == twinNode && (c.offsetAndSwitch & 255u) == twinOffset) skip = true;
				}
			}
			if (v.nFw > 0 && !skip) {
				const int32_t row = p - v.p - 1;   // forward row of read position p (row -1 is the seed's own position)
				uint32_t lo = 0, hi = v.nFw;
				while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (v.fw[mid].seqPos > row) lo = mid + 1; else hi = mid; }
				for (uint32_t i = lo; i < v.nFw && !skip; i++) {
					const PoolCell c = v.fw[i];
					if (c.seqPos != row) break;
					if (c.node == sd.node && (c.offsetAndSwitch & 255u) == sd.offset) skip = true;
				}
			}
		}
		if (skip || status != 0) continue;
		// --- this seed is extended: getAlignmentFromSeed (:567-626)
		const ExtResult& eb = ext[2 * (size_t)sIdx];
		const ExtResult& ef = ext[2 * (size_t)sIdx + 1];
		if (rounds.lazy && eb.status == EXT_NOT_RUN) {   // (both directions are queued together, so one test covers both)
			// (the last parking round queues all the seeds the fragment has left: the few fragments that get this far then finish in one more
			// round, at the price of some extensions the reference would not have run)
			const uint32_t nQueued = rounds.parkAll ? nSeeds - k : 1u;
			const unsigned long long at = atomicAdd(rounds.nextListCount, 2ull * nQueued);
			for (uint32_t q = 0; q < nQueued; q++) {
				rounds.nextList[at + 2 * q] = 2 * (sIdx + q);
				rounds.nextList[at + 2 * q + 1] = 2 * (sIdx + q) + 1;
			}
			rounds.nextPending[atomicAdd(rounds.nextPendingCount, 1ull)] = f;
			rounds.fragNext[f] = k;
			fragExtended[f] = extended;
			return;
		}
		extended++;
		bool runB = p > 0, runF = p < splitLen - 1;
		// (the reference runs the backward extension first and the forward one only if that did not throw, src/GraphAligner.h:499-511)
		if (readTies && runB && (eb.pad & 1u)) atomicAdd(&readTies[fr.read], 1u);
		if (readTies && runF && !(runB && eb.status == EXT_ASSERT) && (ef.pad & 1u)) atomicAdd(&readTies[fr.read], 1u);
		if ((runB && eb.status == EXT_ASSERT) || (runF && ef.status == EXT_ASSERT)) { status = 1; break; }
		if ((runB && eb.status == EXT_OVERFLOW) || (runF && ef.status == EXT_OVERFLOW)) { status = 2; break; }
		bool hasB = runB && eb.status == EXT_OK, hasF = runF && ef.status == EXT_OK;
		if (!hasB && !hasF) continue;   // emptyAlignment: alignmentFailed()
		MergedView v;
		v.p = p;
		v.bw = tracePool + eb.traceOff; v.nBw = hasB ? (hasF ? eb.traceLen - 1 : eb.traceLen) : 0;
		v.fw = tracePool + ef.traceOff; v.nFw = hasF ? ef.traceLen : 0;
		uint32_t n = v.size();
		// anchor path = distinct consecutive split nodes along the trace (src/Aligner.cpp:714-720). ONE pass over the two traces (r3): a 35-row fragment crosses one to three
		// split nodes, so the path collects in six registers and only a longer one walks the traces a second time; the cells are fetched eight at a time (twelve-byte cells at a
		// per-lane stride: the eight loads of a group go out together and use a cache line while it is there - issued one per iteration, every load found its line evicted by the
		// other waves' lines and the kernel fetched 20 GB per 10 k reads for 2.5 GB of cells); the strand flip of a backward cell keeps its node's four table values while the
		// node stays the same.
		constexpr uint32_t PATH_REGS = 6;
		uint32_t pn0 = 0, pn1 = 0, pn2 = 0, pn3 = 0, pn4 = 0, pn5 = 0;
		uint32_t pathLen = 0, last = 0xffffffffu;
		AnchorRec rec;
		rec.firstNode = rec.firstOffset = rec.firstSeqPos = 0;
		rec.lastNode = rec.lastOffset = rec.lastSeqPos = 0;
		bool haveFirst = false;
		auto visit = [&](uint32_t node, uint32_t off, int32_t sp) __attribute__((always_inline)) {
			if (node != last) {
				switch (pathLen) { case 0: pn0 = node; break; case 1: pn1 = node; break; case 2: pn2 = node; break; case 3: pn3 = node; break; case 4: pn4 = node; break; case 5: pn5 = node; break; default: break; }
				pathLen++; last = node;
			}
			if (!haveFirst) { rec.firstNode = node; rec.firstOffset = off; rec.firstSeqPos = (uint32_t)sp; haveFirst = true; }
			rec.lastNode = node; rec.lastOffset = off; rec.lastSeqPos = (uint32_t)sp;
		};
		{
			uint32_t cachedNode = 0xffffffffu, nodeOff = 0, origSz = 0, lookupBase = 0, cachedBlock = 0xffffffffu, twin = 0, twinBase = 0;
			for (uint32_t i0 = 0; i0 < v.nBw; i0 += 8) {
				PoolCell c[8];
#pragma unroll
				for (uint32_t u = 0; u < 8; u++) c[u] = v.bw[min(i0 + u, v.nBw - 1)];
#pragma unroll
				for (uint32_t u = 0; u < 8; u++) {
					if (i0 + u >= v.nBw) break;
					if (c[u].node != cachedNode) {
						cachedNode = c[u].node;
						const int32_t id = g.nodeIDs[cachedNode];
						nodeOff = g.nodeOffset[cachedNode]; origSz = g.origSize[id]; lookupBase = g.lookupOff[id ^ 1];
						cachedBlock = 0xffffffffu;
					}
					const uint32_t rev = origSz - 1 - (nodeOff + (c[u].offsetAndSwitch & 255u));   // GetReversePosition + GetUnitigNode, as in mergedCell
					if (rev / 64 != cachedBlock) { cachedBlock = rev / 64; twin = g.lookup[lookupBase + cachedBlock]; twinBase = g.nodeOffset[twin]; }
					visit(twin, rev - twinBase, v.p - 1 - c[u].seqPos);
				}
			}
			for (uint32_t i0 = 0; i0 < v.nFw; i0 += 8) {   // the forward trace runs towards the seed: used back to front
				PoolCell c[8];
#pragma unroll
				for (uint32_t u = 0; u < 8; u++) c[u] = v.fw[v.nFw - 1 - min(i0 + u, v.nFw - 1)];
#pragma unroll
				for (uint32_t u = 0; u < 8; u++) {
					if (i0 + u >= v.nFw) break;
					visit(c[u].node, c[u].offsetAndSwitch & 255u, v.p + 1 + c[u].seqPos);
				}
			}
		}
		unsigned long long base = atomicAdd(pathCursor, (unsigned long long)pathLen);
		if (base + pathLen > pathCapacity) { status = 2; break; }
		rec.valid = 1;
		rec.x = fr.l;
		rec.y = fr.l + (uint32_t)splitLen - 1;
		rec.pad = 0;
		rec.pathOff = base;
		rec.pathLen = pathLen;
		rec.score = (hasB ? eb.score : 0) + (hasF ? ef.score : 0);
		if (pathLen <= PATH_REGS) {
			if (pathLen > 0) pathPool[base] = pn0;
			if (pathLen > 1) pathPool[base + 1] = pn1;
			if (pathLen > 2) pathPool[base + 2] = pn2;
			if (pathLen > 3) pathPool[base + 3] = pn3;
			if (pathLen > 4) pathPool[base + 4] = pn4;
			if (pathLen > 5) pathPool[base + 5] = pn5;
		} else {
			uint32_t w = 0;
			last = 0xffffffffu;
			for (uint32_t i = 0; i < n; i++) {
				uint32_t node, off; int32_t sp;
				mergedCell(g, v, i, node, off, sp);
				if (node != last) { pathPool[base + w++] = node; last = node; }
			}
		}
		anchors[sIdx] = rec;
	}
	if (status != 0) for (uint32_t k = 0; k < nSeeds; k++) anchors[fr.seedBegin + k].valid = 0;   // a throwing AlignOneWay returns nothing
	fragStatus[f] = status;
	fragExtended[f] = extended;
}

// =====================================================================================================
// K4 - co-linear chaining. reference: AlignmentGraph::colinearChaining / colinearChainingByComponent,
// src/AlignmentGraph.cpp:1712-1863. One wave per read.
//
// The reference sweeps anchor endpoints in topological order and keeps, per path k of the minimum path cover,
// two treaps keyed by read coordinate. What that sweep computes for anchor j is
//     C[j] = max( (len_j, -1),
//                 max_i (len_j + C[i].first, i)         over i with  y_i <  x_j      and  end(i) => start(j)
//                 max_i (y_j - y_i + C[i].first, i)     over i with  x_j <= y_i < y_j and  end(i) => start(j) )
// with lexicographic pair max (ties to the larger anchor index, :1812,1849) and  u => v  meaning "u strictly
// reaches v" (looked up through backwards[v], :1766,1834-1845) or "u == v and i sorts before j by (y,x)"
// (the node-local pass, :1785-1822). Both read conditions imply x_i < x_j, i.e. i belongs to an earlier
// fragment, and C[i] is final before any j of a later fragment reads it. So the same maxima can be taken in
// FRAGMENT order with a direct reachability test per pair, which needs no sort and no tables and is parallel
// over i: lanes scan the earlier anchors, a wave max-reduction yields C[j]. Anchor slots are already in
// fragment order. u strictly reaches v  <=>  some path k through u has pos_k(u) <= pos_k(last node of path k
// that strictly reaches v) - exactly the (node, k) pairs of backwards[v] (computeMPCIndex, :1373-1384).
// Cost O(n^2 K / 64) per read; n is a few hundred anchors.
// =====================================================================================================

#define CHAIN_NEEDS_SCRATCH 9u   // chainStatus between the two launches

__device__ __forceinline__ unsigned long long packScore(long long score, long long anchor)
{
	return ((unsigned long long)(score + (1ll << 30)) << 32) | (unsigned long long)(uint32_t)(anchor + 1);
}
__device__ __forceinline__ long long unpackScore(unsigned long long v) { return (long long)(v >> 32) - (1ll << 30); }
__device__ __forceinline__ long long unpackAnchor(unsigned long long v) { return (long long)(uint32_t)v - 1; }

// end(i) == u, start(j) == v, same weakly connected component already checked
__device__ __forceinline__ bool reachesStrictly(const DGraph& g, uint32_t u, uint32_t v)
{
	uint32_t pu = g.pathsOff[u], puEnd = g.pathsOff[u + 1];
	uint32_t bv = g.backOff[v], bvEnd = g.backOff[v + 1];
	while (pu < puEnd && bv < bvEnd) {   // both lists are ascending in path id
		uint32_t ku = g.paths[pu], kv = g.backPath[bv];
		if (ku == kv) {
			if (g.pathsPos[pu] <= g.backPos[bv]) return true;
			pu++; bv++;
		} else if (ku < kv) pu++;
		else bv++;
	}
	return false;
}

#define CHAIN_LDS_ANCHORS 1536
#define CHAIN_LDS_ENTRIES 2304
#define CHAIN_LDS_WIDTH 512

// Same maxima, without a reachability test per pair (r2). "end(i) strictly reaches start(j)" means: some path k of the cover runs
// through end(i) at a position <= the position of the last node of k that reaches start(j) - the (k, position) pairs of
// backwards[start(j)] (computeMPCIndex, :1373-1384). So every anchor i leaves one ENTRY (k, position of end(i) on k) per path
// through its end node, kept in LDS in anchor (= fragment) order, and anchor j is answered by one scan of the entries of earlier
// fragments against a small table thr[k] = position from backwards[start(j)] (-1 elsewhere): the reference's per-path treaps keyed
// by read coordinate, with the sweep order turned from topological to fragment order so that "earlier in the read" is a prefix.
// No global loads inside the DP loop (the pair-wise version fetched ~20x its input: two sorted lists merged per pair): entries and
// threshold lists of all anchors are laid out in LDS by the set-up pass, lanes in parallel.
struct ChainEntry { uint32_t pos; uint16_t k, anchor; };   // an anchor's end node on path k at position pos

struct ChainArrays {   // one read's working set: LDS (template LDS) or this block's HBM scratch
	uint32_t* aStart; unsigned long long* C;
	uint16_t* aFrag; uint16_t* aComp; uint16_t* entBegin; uint32_t* backBegin;
	ChainEntry* ent; uint2* back; int32_t* thr;   // back: the anchors' threshold lists (path, position), always in this block's HBM scratch (a wide cover makes them long)
	ChainEntry* entByComp; uint16_t* aBucket; uint16_t* entByCompBegin;   // scratch launch only (r4): the entries grouped by weakly connected component, see CHAIN_BUCKETS
	uint16_t* anchorByBucket;                                              // ... and (r5) the anchors grouped the same way, in anchor order inside a bucket
};
// r4, the scratch launch (reads with more anchors than the LDS classes hold - every 50 kb read): an anchor is only ever chained to anchors of its own weakly connected component, but the scan
// above visits the entries of ALL earlier anchors and drops the others one by one. On a 1 Gbp graph a 15-mer has a chance hit somewhere in the genome as often as not: a 50 kb read brings ~10 000
// anchors, most of them strays spread over the other chromosomes' components, and the scan was 3.1 s per 2 000 reads (`gpurun_out/r4_cfg5z`). The entries are therefore regrouped by component -
// component -> bucket through a 512-slot table, a counting sort that keeps the anchor order inside a bucket - and anchor j scans the part of its own bucket that earlier fragments filled.
// A read that touches more than 256 components keeps the plain scan.
#define CHAIN_BUCKETS 512u

// LDS == 0: the working set in this block's HBM scratch; 1: in LDS, the large class (53 KB: three blocks per CU); 2 (r4): in LDS, half the size (26.5 KB, six blocks per CU) - what a
// 10 kb read needs (~300 anchors, ~600 entries, cover width of a few) and the class launchChain picks when the batch's largest read fits it; a read that outgrows its class goes to the scratch launch
template <int LDS>
__global__ void __launch_bounds__(64) k_chain(DGraph g, const ReadChainJob* __restrict__ jobs, uint32_t nReads, const AnchorRec* __restrict__ anchors,
	const Fragment* __restrict__ frags, const uint32_t* __restrict__ fragStatus, int32_t splitLen, int32_t splitGap, ChainCaps caps, uint8_t* __restrict__ scratch, uint64_t scratchStride,
	uint32_t* __restrict__ chainOut, uint32_t* __restrict__ chainLen, unsigned long long* __restrict__ chainScore, uint32_t* __restrict__ chainStatus, uint32_t forceScratch)
{
	GC_RAISE_PRIO();
	constexpr uint32_t LDS_ANCHORS = LDS == 2 ? CHAIN_LDS_ANCHORS / 2 : CHAIN_LDS_ANCHORS, LDS_ENTRIES = LDS == 2 ? CHAIN_LDS_ENTRIES / 2 : CHAIN_LDS_ENTRIES, LDS_WIDTH = LDS == 2 ? CHAIN_LDS_WIDTH / 2 : CHAIN_LDS_WIDTH;
	__shared__ uint32_t sStart[LDS ? LDS_ANCHORS : 1];
	__shared__ unsigned long long sC[LDS ? LDS_ANCHORS : 1];
	__shared__ uint16_t sFrag[LDS ? LDS_ANCHORS : 1], sComp[LDS ? LDS_ANCHORS : 1], sEntBegin[LDS ? LDS_ANCHORS + 1 : 1];
	__shared__ uint32_t sBackBegin[LDS ? LDS_ANCHORS + 1 : 1];
	__shared__ ChainEntry sEnt[LDS ? LDS_ENTRIES : 1];
	__shared__ int32_t sThr[LDS ? LDS_WIDTH : 1];
	__shared__ uint32_t bKey[LDS ? 1 : CHAIN_BUCKETS], bStart[LDS ? 1 : CHAIN_BUCKETS + 1], bFilled[LDS ? 1 : CHAIN_BUCKETS], bUsed;   // bucket -> component, first entry, entries of earlier fragments
	__shared__ uint32_t bAnchors[LDS ? 1 : CHAIN_BUCKETS], bAnchorStart[LDS ? 1 : CHAIN_BUCKETS + 1], bAnchorsPlaced[LDS ? 1 : CHAIN_BUCKETS];   // (r5) anchors per bucket, and where the bucket's anchors begin in anchorByBucket
	constexpr uint32_t SCRATCH_THR = 2048;
	__shared__ int32_t sThrScratch[LDS ? 1 : SCRATCH_THR];   // (r5) the scratch launch's threshold table stays in LDS whenever the graph's widest path cover fits: it is scattered into, read per scanned entry and cleared for every anchor
	const int lane = threadIdx.x;
	// (16-bit indices: at most 65535 anchors, entries and threshold-list items per read, cover width and components below 65536; beyond that the read is flagged)
	const uint32_t capA = LDS ? LDS_ANCHORS : (caps.capAnchors < 65535u ? caps.capAnchors : 65535u);
	const uint32_t capE = LDS ? LDS_ENTRIES : (caps.capEndpoints < 65535u ? caps.capEndpoints : 65535u);
	const uint32_t capB = caps.capBack;
	const uint32_t capW = LDS ? LDS_WIDTH : (caps.capTable < 65535u ? caps.capTable : 65535u);
	ChainArrays A;
	uint8_t* base = scratch + (uint64_t)blockIdx.x * scratchStride;
	A.back = (uint2*)base; base += 8ull * capB;
	if (LDS) { A.aStart = sStart; A.C = sC; A.aFrag = sFrag; A.aComp = sComp; A.entBegin = sEntBegin; A.backBegin = sBackBegin; A.ent = sEnt; A.thr = sThr; }
	else {
		A.C = (unsigned long long*)base; base += 8ull * capA;
		A.ent = (ChainEntry*)base; base += 8ull * capE;
		A.aStart = (uint32_t*)base; base += 4ull * capA;
		A.backBegin = (uint32_t*)base; base += 4ull * (capA + 1);
		A.thr = (int32_t*)base; base += 4ull * capW;
		if (capW <= SCRATCH_THR) A.thr = sThrScratch;
		A.aFrag = (uint16_t*)base; base += 2ull * capA;
		A.aComp = (uint16_t*)base; base += 2ull * capA;
		A.entBegin = (uint16_t*)base; base += 2ull * (capA + 1);
		A.aBucket = (uint16_t*)base; base += 2ull * (capA + 1);
		A.entByCompBegin = (uint16_t*)base; base += 2ull * (capA + 1);
		A.anchorByBucket = (uint16_t*)base; base += 2ull * (capA + 1);
		base = (uint8_t*)(((uintptr_t)base + 7) & ~(uintptr_t)7);
		A.entByComp = (ChainEntry*)base;
	}
	for (uint32_t w = lane; w < capW; w += 64) A.thr[w] = -1;
	__syncthreads();
	for (uint32_t r = blockIdx.x; r < nReads; r += gridDim.x) {
		ReadChainJob job = jobs[r];
		if (!LDS && !(forceScratch & 4u) && chainStatus[r] != CHAIN_NEEDS_SCRATCH) continue;   // second launch: only the reads the LDS launch passed on (bit 2: there was no LDS launch - every read)
		// (r5) a read with more than twice the class's anchors in SLOTS is sent on without a look at its anchors: routing only - the scratch launch takes any read
		if (LDS && job.nSlots > 2 * LDS_ANCHORS) { if (lane == 0) { chainStatus[r] = CHAIN_NEEDS_SCRATCH; chainLen[r] = 0; chainScore[r] = 0; } continue; }
		// the reference never resets its `cont` flag after a fragment whose extension threw, so fragments after the
		// first failed one contribute no anchors (src/Aligner.cpp:695-703): slots from that fragment on are cut off
		uint32_t cut = job.nSlots;
		for (uint32_t f = lane; f < job.nFrags; f += 64)
			if (fragStatus[job.fragBegin + f] == 1) { uint32_t c = frags[job.fragBegin + f].seedBegin - job.slotBegin; cut = c < cut ? c : cut; }
		for (int d = 32; d > 0; d >>= 1) { uint32_t o = __shfl_xor(cut, d); cut = o < cut ? o : cut; }
		// Compact the read's valid anchors in slot order (== the order the reference pushes them, src/Aligner.cpp:706-721). Every anchor gets
		// its entries (one per path through its END node) and its threshold list: for its START node v, per path k the last position on k
		// that may precede it - backwards[v] (strict reachability, :1373-1384) plus (k, position of v) for the paths through v itself,
		// which adds "same node" (:1785-1822: an anchor of an earlier fragment ending in v sorts before j by (y, x)).
		uint32_t nA = 0, nE = 0, nB = 0;
		bool fits = !(LDS && forceScratch);   // test hook: every read through the scratch launch
		for (uint32_t s0 = 0; s0 < cut && fits; s0 += 64) {
			uint32_t s = s0 + lane;
			bool valid = false;
			AnchorRec rec;
			uint32_t pBegin = 0, pCount = 0, bBegin = 0, bCount = 0, vBegin = 0, vCount = 0, comp = 0;
			if (s < cut) { rec = anchors[job.slotBegin + s]; valid = rec.valid != 0; }
			if (valid) {
				pBegin = g.pathsOff[rec.lastNode]; pCount = g.pathsOff[rec.lastNode + 1] - pBegin;
				bBegin = g.backOff[rec.firstNode]; bCount = g.backOff[rec.firstNode + 1] - bBegin;
				vBegin = g.pathsOff[rec.firstNode]; vCount = g.pathsOff[rec.firstNode + 1] - vBegin;
				comp = g.componentMap[rec.lastNode];
			}
			unsigned long long ballot = __ballot(valid);
			uint32_t inclE = pCount, inclB = bCount + vCount;
			for (int d = 1; d < 64; d <<= 1) {
				uint32_t oe = __shfl_up(inclE, d), ob = __shfl_up(inclB, d);
				if (lane >= d) { inclE += oe; inclB += ob; }
			}
			const uint32_t chunkE = __shfl(inclE, 63), chunkB = __shfl(inclB, 63), chunkA = (uint32_t)__popcll(ballot);
			const bool wide = valid && (comp > 65535u || g.mpcWidth[comp] > capW || rec.x / (uint32_t)splitGap > 65535u);
			if (nA + chunkA > capA || nE + chunkE > capE || nB + chunkB > capB || __any(wide)) { fits = false; break; }
			if (valid) {
				const uint32_t a = nA + (uint32_t)__popcll(ballot & ((1ull << lane) - 1));
				const uint32_t e0 = nE + inclE - pCount, t0 = nB + inclB - (bCount + vCount);
				A.aStart[a] = rec.firstNode;
				A.aFrag[a] = (uint16_t)(rec.x / (uint32_t)splitGap);   // fragment starts are multiples of the step: x and the index order alike
				A.aComp[a] = (uint16_t)comp;
				A.C[a] = packScore((long long)rec.y - (long long)rec.x + 1, -1);   // :1769
				A.entBegin[a] = (uint16_t)e0;
				A.backBegin[a] = t0;
				for (uint32_t i = 0; i < pCount; i++) A.ent[e0 + i] = ChainEntry { g.pathsPos[pBegin + i], (uint16_t)g.paths[pBegin + i], (uint16_t)a };
				for (uint32_t i = 0; i < bCount; i++) A.back[t0 + i] = make_uint2(g.backPath[bBegin + i], g.backPos[bBegin + i]);
				for (uint32_t i = 0; i < vCount; i++) A.back[t0 + bCount + i] = make_uint2(g.paths[vBegin + i], g.pathsPos[vBegin + i]);
			}
			nA += chunkA; nE += chunkE; nB += chunkB;
		}
		if (!fits) {
			if (lane == 0) { chainStatus[r] = LDS ? CHAIN_NEEDS_SCRATCH : 1u; chainLen[r] = 0; chainScore[r] = 0; }
			__syncthreads();
			continue;
		}
		if (lane == 0) { A.entBegin[nA] = (uint16_t)nE; A.backBegin[nA] = nB; }
		__syncthreads();
		bool bucketed = false;
		if (LDS == 0) {
			for (uint32_t b = lane; b < CHAIN_BUCKETS; b += 64) { bKey[b] = 0xffffffffu; bStart[b] = 0; bFilled[b] = 0; bAnchors[b] = 0; bAnchorsPlaced[b] = 0; }
			if (lane == 0) { bUsed = 0; bStart[CHAIN_BUCKETS] = 0; }
			__syncthreads();
			// component -> bucket (the slot its id hashes to), entries per bucket
			for (uint32_t a = lane; a < nA; a += 64) {
				const uint32_t comp = A.aComp[a];
				uint32_t slot = (comp * 2654435761u >> 16) & (CHAIN_BUCKETS - 1);
				for (uint32_t probes = 0; probes < CHAIN_BUCKETS; probes++, slot = (slot + 1) & (CHAIN_BUCKETS - 1)) {
					const uint32_t seen = atomicCAS(&bKey[slot], 0xffffffffu, comp);
					if (seen == 0xffffffffu) atomicAdd(&bUsed, 1u);
					if (seen == 0xffffffffu || seen == comp) break;
				}
				A.aBucket[a] = (uint16_t)slot;
				atomicAdd(&bStart[slot], (uint32_t)A.entBegin[a + 1] - (uint32_t)A.entBegin[a]);
				atomicAdd(&bAnchors[slot], 1u);
			}
			__syncthreads();
			bucketed = bUsed <= CHAIN_BUCKETS / 2 && !(forceScratch & 2u);   // (beyond that the probing may have wrapped: plain scan; bit 1 of forceScratch: test hook, plain scan)
			if (bucketed) {
				// exclusive scan of the counts (one wave, eight buckets per lane), then every anchor's place in its bucket in anchor order - lane 0, one pass
				uint32_t mine[CHAIN_BUCKETS / 64], sum = 0, mineA[CHAIN_BUCKETS / 64], sumA = 0;
				for (uint32_t k = 0; k < CHAIN_BUCKETS / 64; k++) { mine[k] = bStart[lane * (CHAIN_BUCKETS / 64) + k]; sum += mine[k]; mineA[k] = bAnchors[lane * (CHAIN_BUCKETS / 64) + k]; sumA += mineA[k]; }
				uint32_t incl = sum, inclA = sumA;
				for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d), oa = __shfl_up(inclA, d); if (lane >= d) { incl += o; inclA += oa; } }
				uint32_t at = incl - sum, atA = inclA - sumA;
				__syncthreads();
				for (uint32_t k = 0; k < CHAIN_BUCKETS / 64; k++) { bStart[lane * (CHAIN_BUCKETS / 64) + k] = at; at += mine[k]; bAnchorStart[lane * (CHAIN_BUCKETS / 64) + k] = atA; atA += mineA[k]; }
				__syncthreads();
				// every anchor's place in its bucket, in anchor order (r5: 64 anchors at a time - a lane adds up the entries of the chunk's earlier anchors that share its bucket from
				// the other lanes' registers, the buckets' fill counts advance by LDS atomics between chunks; r4 walked the read's ~10 000 anchors on lane 0 through its HBM scratch)
				for (uint32_t a0 = 0; a0 < nA; a0 += 64) {
					const uint32_t a = a0 + lane;
					const bool have = a < nA;
					const uint32_t slot = have ? (uint32_t)A.aBucket[a] : 0xffffffffu;
					const uint32_t cnt = have ? (uint32_t)A.entBegin[a + 1] - (uint32_t)A.entBegin[a] : 0u;
					uint32_t before = 0, anchorsBefore = 0;
					for (int l = 0; l < 63; l++) { const uint32_t s = __shfl(slot, l), c = __shfl(cnt, l); if (l < lane && s == slot) { before += c; anchorsBefore++; } }
					if (have) {
						A.entByCompBegin[a] = (uint16_t)(bStart[slot] + bFilled[slot] + before);
						A.anchorByBucket[bAnchorStart[slot] + bAnchorsPlaced[slot] + anchorsBefore] = (uint16_t)a;
					}
					__syncthreads();
					if (have) { if (cnt) atomicAdd(&bFilled[slot], cnt); atomicAdd(&bAnchorsPlaced[slot], 1u); }
					__syncthreads();
				}
				for (uint32_t b = lane; b < CHAIN_BUCKETS; b += 64) bFilled[b] = 0;
				__syncthreads();
				for (uint32_t a = lane; a < nA; a += 64)
					for (uint32_t e = A.entBegin[a], to = A.entByCompBegin[a]; e < A.entBegin[a + 1]; e++, to++) A.entByComp[to] = A.ent[e];
				__syncthreads();
			}
		}
		// anchors are ordered by fragment; g0 = first anchor of j's fragment, the entries of anchors < g0 are a prefix. The threshold lists
		// of consecutive anchors are consecutive in `back`: a 64-item window of them is kept in registers (item w0 + lane), refilled with one
		// coalesced load when an anchor's list runs past it - the only global access of the DP loop, a few times per hundred anchors on a narrow cover.
		uint32_t w0 = 0;
		uint2 win = (uint32_t)lane < nB ? A.back[lane] : make_uint2(0, 0);
		// one anchor's step: thr[k] = the last position on path k that may precede start(j) (its threshold list scattered into the table), the scan of the entries
		// [scanBegin, scanEnd) - earlier fragments' anchors - the wave maximum into C[j], the table cleared again. sameComponent: the entries are j's component's already
		auto relax = [&](uint32_t j, uint32_t fj, const ChainEntry* entries, uint32_t scanBegin, uint32_t scanEnd, bool sameComponent) {
			const uint32_t b0 = A.backBegin[j], b1 = A.backBegin[j + 1];
			if ((b0 >= w0 && b1 <= w0 + 64) || (b1 - b0 <= 64 && (w0 = b0, win = b0 + lane < nB ? A.back[b0 + lane] : make_uint2(0, 0), true))) {
				const uint32_t t = w0 + lane;
				if (t >= b0 && t < b1) atomicMax(&A.thr[win.x], (int32_t)win.y);
			} else {
				for (uint32_t t = b0 + lane; t < b1; t += 64) { const uint2 it = A.back[t]; atomicMax(&A.thr[it.x], (int32_t)it.y); }   // a list longer than the window (wide cover)
			}
			__syncthreads();
			const uint32_t compV = A.aComp[j];
			const long long xj = (long long)fj * splitGap, yj = xj + splitLen - 1;
			unsigned long long best = 0;
			for (uint32_t e = scanBegin + lane; e < scanEnd; e += 64) {
				const ChainEntry en = entries[e];
				const uint32_t i = en.anchor;
				if ((!sameComponent && A.aComp[i] != compV) || (int32_t)en.pos > A.thr[en.k]) continue;   // (path ids are per component)
				const long long yi = (long long)A.aFrag[i] * splitGap + splitLen - 1;
				const long long ci = unpackScore(A.C[i]);
				unsigned long long cand;
				if (yi <= xj - 1) cand = packScore((long long)splitLen + ci, i);
				else cand = packScore(yj - yi + ci, i);   // x_j <= y_i <= y_j - 1
				best = cand > best ? cand : best;
			}
			for (int d = 32; d > 0; d >>= 1) {
				unsigned long long o = __shfl_xor(best, d);
				best = o > best ? o : best;
			}
			if (lane == 0 && best > A.C[j]) A.C[j] = best;
			__syncthreads();
			if (b1 <= w0 + 64 && b0 >= w0) {
				const uint32_t t = w0 + lane;
				if (t >= b0 && t < b1) A.thr[win.x] = -1;
			} else {
				for (uint32_t t = b0 + lane; t < b1; t += 64) A.thr[A.back[t].x] = -1;
			}
		};
		// The read's chain (:1713-1733, :1847-1862): per component the lexicographic maximum of (coverage, anchor index); over the components, visited in ascending id, the first
		// strictly greater coverage - i.e. the largest coverage, among equals the smallest component id, inside it the largest anchor index: a wave maximum over the anchors of
		// coverage << 32 | (65535 - component) << 16 | anchor (r5; r4 let lane 0 walk all anchors twice per component - a million loads from the scratch for a 50 kb read that
		// touches fifty components)
		unsigned long long top = 0;
		auto topOver = [&](uint32_t count, auto anchorAt) {
			unsigned long long mine = 0;
			for (uint32_t q = lane; q < count; q += 64) {
				const uint32_t a = anchorAt(q);
				const unsigned long long key = ((unsigned long long)unpackScore(A.C[a]) << 32) | ((unsigned long long)(65535u - (uint32_t)A.aComp[a]) << 16) | a;
				mine = key > mine ? key : mine;
			}
			for (int d = 32; d > 0; d >>= 1) { const unsigned long long o = __shfl_xor(mine, d); mine = o > mine ? o : mine; }
			top = mine > top ? mine : top;
		};
		if (LDS == 0 && bucketed) {
			// r5: component by component, the one with the most anchors first - and only while a component can still win: an anchor adds at most splitLen to a chain's coverage, so a
			// bucket of c anchors cannot reach more than c x splitLen. On a genome-sized graph a 50 kb read brings ~10 000 anchors of which ~1 400 lie where it comes from; the strays
			// on the other chromosomes' components (a few hundred each) never come near that component's coverage and used to cost six sevenths of the kernel. What is skipped has no
			// part in the result: the chain and its coverage are the winner's, C of the other components is never output.
			while (true) {
				uint32_t pick = 0;   // anchors << 10 | (1023 - slot): the fullest bucket not yet done, the lowest slot among equals
				for (uint32_t k = 0; k < CHAIN_BUCKETS / 64; k++) { const uint32_t slot = lane * (CHAIN_BUCKETS / 64) + k, c = bAnchors[slot]; const uint32_t key = c ? (c << 10) | (1023u - slot) : 0u; pick = key > pick ? key : pick; }
				for (int d = 32; d > 0; d >>= 1) { const uint32_t o = __shfl_xor(pick, d); pick = o > pick ? o : pick; }
				const uint32_t count = pick >> 10, slot = 1023u - (pick & 1023u);
				if (count == 0) break;
				if ((unsigned long long)count * (unsigned long long)splitLen < (top >> 32)) break;   // (an equal bound could still tie, and the smaller component id wins a tie: only strictly less is safe)
				__syncthreads();
				if (lane == 0) bAnchors[slot] = 0;
				const uint16_t* list = A.anchorByBucket + bAnchorStart[slot];
				const uint32_t entFirst = bStart[slot];
				uint32_t filled = 0, groupFrom = 0, fragPrev = 0;
				for (uint32_t q = 0; q < count; q++) {
					const uint32_t j = list[q];
					const uint32_t fj = A.aFrag[j];
					if (q > 0 && fj != fragPrev) {   // the fragment that just ended joins the prefix: C of its anchors is final
						uint32_t add = 0;
						for (uint32_t x = groupFrom + lane; x < q; x += 64) { const uint32_t a = list[x]; add += (uint32_t)A.entBegin[a + 1] - (uint32_t)A.entBegin[a]; }
						for (int d = 32; d > 0; d >>= 1) add += __shfl_xor(add, d);
						filled += add;
						groupFrom = q;
						__syncthreads();
					}
					fragPrev = fj;
					if (filled == 0) continue;
					relax(j, fj, A.entByComp, entFirst, entFirst + filled, true);
				}
				__syncthreads();
				topOver(count, [&](uint32_t q) { return (uint32_t)list[q]; });
			}
		} else {
			// anchors are ordered by fragment; g0 = first anchor of j's fragment, the entries of anchors < g0 are a prefix. The threshold lists
			// of consecutive anchors are consecutive in `back`: a 64-item window of them is kept in registers (item w0 + lane), refilled with one
			// coalesced load when an anchor's list runs past it - the only global access of the DP loop, a few times per hundred anchors on a narrow cover.
			uint32_t g0 = 0, entPrefix = 0;
			for (uint32_t j = 0; j < nA; j++) {
				const uint32_t fj = A.aFrag[j];
				if (j > 0 && fj != A.aFrag[j - 1]) { g0 = j; entPrefix = A.entBegin[j]; __syncthreads(); }   // C of the previous fragment's anchors is now final and visible
				if (g0 == 0) continue;
				relax(j, fj, A.ent, 0, entPrefix, false);
			}
			__syncthreads();
			topOver(nA, [&](uint32_t q) { return q; });
		}
		if (lane == 0) {
			uint32_t status = 0;
			const long long best = nA ? (long long)(top >> 32) : 0;
			uint32_t bestLen = 0;
			uint32_t* out = chainOut + job.chainBegin;
			if (nA) {
				uint32_t n = 0;
				for (long long i = (long long)(top & 0xffffu); i != -1; i = unpackAnchor(A.C[i])) {   // :1851-1862
					if (n >= job.nSlots) { status = 1; break; }
					out[n++] = (uint32_t)i;
				}
				for (uint32_t i = 0; i < n / 2; i++) { uint32_t t = out[i]; out[i] = out[n - 1 - i]; out[n - 1 - i] = t; }
				bestLen = n;
			}
			chainLen[r] = bestLen;
			chainScore[r] = (unsigned long long)best;
			chainStatus[r] = status;
		}
		__syncthreads();
	}
}

// =====================================================================================================
// K3-long - the whole-read GraphAligner pass. reference: AlignOneWay with sloppyOptimizations,
// src/GraphAligner.h:114-203 (called from src/Aligner.cpp:565), getAlignmentFromSeed :567-626,
// exactAlignmentPart :407-461. The reference walks a read's seeds in goodness order and every decision
// (skip because an earlier alignment covers the seed, stop because the read is aligned end to end)
// depends on the alignments produced so far, so a read is one sequential unit: one lane per read, the
// multi-slice extension core of gc_device.hpp does the work. All reads of a batch have similar length,
// which keeps the lanes of a wave in step at slice granularity.
// =====================================================================================================

struct LongSlab { LaneScratch sc; TraceCell* traceB; };   // backward trace is parked while the forward extension reuses sc.trace

__device__ __forceinline__ LongSlab longSlab(uint8_t* slab, const ExtendConfig& cfg)
{
	LongSlab ls;
	ls.sc = laneScratch(slab, cfg);
	ls.traceB = (TraceCell*)(ls.sc.itemNodes + cfg.maxItems);   // (behind the lane's extension slab)
	return ls;
}

// reverse-strand twin of (split node, offset): GetReversePosition + GetUnitigNode (src/AlignmentGraph.cpp:832-868)
__device__ __forceinline__ void twinOf(const DGraph& g, uint32_t node, uint32_t offset, uint32_t& twinNode, uint32_t& twinOffset)
{
	const int32_t id = g.nodeIDs[node];
	const uint32_t rev = g.origSize[id] - 1 - (g.nodeOffset[node] + offset);
	twinNode = g.lookup[g.lookupOff[id ^ 1] + rev / 64];
	twinOffset = rev - g.nodeOffset[twinNode];
}

// Is the seed's cell on this alignment's trace? (:407-461). Returns 1 yes, 0 no, 2 the reference asserts.
__device__ inline int onTrace(const LongCell* trace, uint32_t n, uint32_t seqPos, int32_t compareNode, uint32_t nodeOffset)
{
	if (!(trace[n - 1].seqPos > trace[0].seqPos)) return 2;
	if (trace[n - 1].seqPos < seqPos || trace[0].seqPos > seqPos) return 0;
	uint32_t lo = 0, hi = n;   // first cell with seqPos >= target (seqPos is non-decreasing along the trace)
	while (lo < hi) {
		uint32_t mid = (lo + hi) / 2;
		if (trace[mid].seqPos < seqPos) lo = mid + 1; else hi = mid;
	}
	for (uint32_t i = lo; i < n && trace[i].seqPos == seqPos; i++)
		if (trace[i].node == compareNode && trace[i].offset == nodeOffset) return 1;
	return 0;
}

// seedScoreForEndToEndAln update after an alignment was added. reference: src/GraphAligner.h:181-198 - the list is
// sorted by alignmentStart; if the first starts at 0, later alignments whose start is <= the running end extend it;
// if the union reaches the read's end the minimum seedGoodness of the contributing alignments becomes the cut-off.
// Visiting in ascending start makes the outcome independent of the order among equal starts, so no sort is needed.
__device__ inline uint32_t endToEndScore(LongAln* mine, uint32_t nAln, uint32_t readLen, uint32_t current)
{
	bool anyAtZero = false;
	for (uint32_t a = 0; a < nAln; a++) if (mine[a].start == 0) anyAtZero = true;
	if (!anyAtZero) return current;
	uint32_t contiguousEnd = 0, minGoodness = 0xffffffffu;
	bool first = true;
	for (uint32_t visited = 0; visited < nAln; visited++) {
		uint32_t best = 0xffffffffu;
		for (uint32_t a = 0; a < nAln; a++) {
			if (mine[a].pad) continue;
			if (best == 0xffffffffu || mine[a].start < mine[best].start) best = a;
		}
		mine[best].pad = 1;
		if (first) { contiguousEnd = mine[best].end; minGoodness = mine[best].goodness; first = false; continue; }
		if (mine[best].start <= contiguousEnd) {
			minGoodness = mine[best].goodness < minGoodness ? mine[best].goodness : minGoodness;
			contiguousEnd = mine[best].end > contiguousEnd ? mine[best].end : contiguousEnd;
		}
	}
	for (uint32_t a = 0; a < nAln; a++) mine[a].pad = 0;
	return contiguousEnd == readLen ? minGoodness : current;
}

__global__ void __launch_bounds__(64) k_long_pass(DGraph g, const CorrectnessTables* __restrict__ ct, const uint8_t* __restrict__ iupac, ExtendConfig cfg,
	const LongJob* __restrict__ jobs, uint32_t nReads, const LongSeed* __restrict__ seeds, const char* __restrict__ bases, uint64_t rcBase,
	uint32_t minClusterSize, uint32_t maxAlignments, uint8_t* __restrict__ scratch, uint64_t slabBytes,
	LongCell* __restrict__ cellPool, unsigned long long* __restrict__ cellCursor, uint64_t cellCapacity,
	LongAln* __restrict__ alns, LongReadResult* __restrict__ results, unsigned long long* __restrict__ counters)
{
	const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t stride = gridDim.x * blockDim.x;
	LongSlab ls = longSlab(scratch + (uint64_t)tid * slabBytes, cfg);
	ExtCounters cnt {};
	for (uint32_t r = tid; r < nReads; r += stride) {
		LongJob job = jobs[r];
		LongAln* mine = alns + job.alnBegin;
		uint32_t nAln = 0, extended = 0, status = 0, ties = 0;
		uint32_t e2eScore = 0;   // seedScoreForEndToEndAln
		const int L = (int)job.readLen;
		// Lanes of a wave must reach the expensive part (the extension) together: a plain loop over seeds would let
		// every lane extend at a different iteration and serialise the wave's extensions. So each lane first advances
		// to its next seed that needs extending (cheap, divergent), then all lanes that found one extend in step.
		uint32_t si = job.seedBegin;
		while (true) {
			bool have = false;
			LongSeed sd {};
			for (; status == 0 && si < job.seedEnd && !have; si++) {
				sd = seeds[si];
				if (sd.goodness < e2eScore) { si = job.seedEnd; break; }   // aligned end to end, skip the rest (:127-131)
				if (sd.clusterSize < minClusterSize) continue;              // :141-146
				bool skip = false;
				for (uint32_t a = 0; a < nAln; a++)                          // sloppy overlap rule (:147-161)
					if (mine[a].start <= sd.seqPos && mine[a].end >= sd.seqPos && mine[a].goodness > sd.goodness) { skip = true; break; }
				if (skip) continue;
				int32_t compareNode = g.nodeIDs[sd.node];
				uint32_t compareOffset = g.nodeOffset[sd.node] + sd.offset;
				for (uint32_t a = 0; a < nAln && !skip; a++) {              // exactAlignmentPart (:163-173)
					int on = onTrace(cellPool + mine[a].traceOff, mine[a].traceLen, sd.seqPos, compareNode, compareOffset);
					if (on == 2) { status = 1; break; }
					if (on == 1) skip = true;
				}
				if (skip || status) continue;
				have = true;
			}
			if (!__any(have)) break;
			if (!have) continue;
			extended++;
			// getAlignmentFromSeed (:567-626): backward on revcomp(read[0..p)), forward on read(p..]
			const int p = (int)sd.seqPos;
			uint32_t nB = 0, nF = 0;
			int32_t scoreB = 0, scoreF = 0;
			uint32_t stB = EXT_FAILED, stF = EXT_FAILED;
			if (p > 0) {
				uint32_t twinNode, twinOffset;
				twinOf(g, sd.node, sd.offset, twinNode, twinOffset);
				stB = extendSeed(g, *ct, iupac, cfg, ls.sc, bases + rcBase + job.readOff + (uint64_t)(L - p), p, twinNode, twinOffset, nB, scoreB, cnt);
				ties += cnt.flattenTie;
				if (stB == EXT_OK) for (uint32_t i = 0; i < nB; i++) ls.traceB[i] = ls.sc.trace[i];
			}
			if (p < L - 1 && stB != EXT_ASSERT) { stF = extendSeed(g, *ct, iupac, cfg, ls.sc, bases + job.readOff + (uint64_t)(p + 1), L - 1 - p, sd.node, sd.offset, nF, scoreF, cnt); ties += cnt.flattenTie; }   // (a throwing backward extension ends getAlignmentFromSeed before the forward one runs)
			if (stB == EXT_ASSERT || stF == EXT_ASSERT) { status = 1; break; }
			if (stB == EXT_OVERFLOW || stF == EXT_OVERFLOW) { status = 2; break; }
			bool hasB = stB == EXT_OK, hasF = stF == EXT_OK;
			if (!hasB && !hasF) continue;   // alignmentFailed()
			if (nAln >= maxAlignments) { status = 3; break; }
			uint32_t useB = hasB ? (hasF ? nB - 1 : nB) : 0;
			uint32_t total = useB + (hasF ? nF : 0);
			unsigned long long base = atomicAdd(cellCursor, (unsigned long long)total);
			if (base + total > cellCapacity) { status = 4; break; }
			LongCell* outCells = cellPool + base;
			for (uint32_t i = 0; i < useB; i++) {   // fixReverseTraceSeqPosAndOrder (:543-565)
				const TraceCell& c = ls.traceB[i];
				uint32_t off = c.offsetAndSwitch & 255u;
				int32_t id = g.nodeIDs[c.node];
				uint32_t orig = g.nodeOffset[c.node] + off;
				LongCell oc;
				oc.node = id ^ 1;
				oc.offset = g.origSize[id] - 1 - orig;
				oc.seqPos = (uint32_t)(p - 1 - c.seqPos);
				oc.nodeSwitch = (i + 1 < nB) ? ((ls.traceB[i + 1].offsetAndSwitch >> 8) & 1u) : 0u;
				outCells[i] = oc;
			}
			if (hasF) for (uint32_t i = 0; i < nF; i++) {   // fixForwardTraceSeqPos (:527-540), device order reversed
				const TraceCell& c = ls.sc.trace[nF - 1 - i];
				LongCell oc;
				oc.node = g.nodeIDs[c.node];
				oc.offset = g.nodeOffset[c.node] + (c.offsetAndSwitch & 255u);
				oc.seqPos = (uint32_t)(p + 1 + c.seqPos);
				oc.nodeSwitch = (c.offsetAndSwitch >> 8) & 1u;
				outCells[useB + i] = oc;
			}
			LongAln al;
			al.start = outCells[0].seqPos;
			al.end = outCells[total - 1].seqPos + 1;
			al.score = (uint32_t)((hasB ? scoreB : 0) + (hasF ? scoreF : 0));
			al.goodness = sd.goodness;
			al.traceOff = base;
			al.traceLen = total;
			al.pad = 0;
			mine[nAln++] = al;
			e2eScore = endToEndScore(mine, nAln, (uint32_t)L, e2eScore);
		}
		LongReadResult rr;
		rr.nAlignments = status == 1 ? 0 : nAln;   // a throwing AlignOneWay returns nothing (src/Aligner.cpp:585-592)
		rr.seedsExtended = extended;
		rr.status = status;
		rr.pad = ties;
		results[r] = rr;
	}
	if (cnt.extensions) {
		atomicAdd(&counters[0], cnt.dpTiles);
		atomicAdd(&counters[1], cnt.recomputeTiles);
		atomicAdd(&counters[2], cnt.columnSteps);
		atomicAdd(&counters[3], cnt.traceItems);
		atomicAdd(&counters[4], cnt.extensions);
		atomicAdd(&counters[5], cnt.backtraceTiles);
	}
}

// =====================================================================================================
// K3-long in rounds. The monolithic kernel above keeps a lane busy for as many rounds as its read needs while the
// other 63 lanes of the wave wait (3.6 seeds are extended per read on average, up to 10). Here every round is
//   select  - one wave per read: the reference's skip rules for 64 seeds at once, then its in-order scan on the ballot
//             masks; emits two work items (backward, forward) per chosen seed;
//   extend  - one wave per work item (k_long_extend<1>), longest first; the trace goes to a pool;
//   merge   - one wave per read: joins the two traces 64 cells at a time, appends the alignment, updates the cut-off;
// and the host launches rounds until no read emits work. Results are identical to the monolithic kernels.
// =====================================================================================================

__global__ void __launch_bounds__(256) k_long_init(const LongJob* __restrict__ jobs, uint32_t nReads, LongState* __restrict__ state)
{
	GC_RAISE_PRIO();
	uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= nReads) return;
	LongState st { jobs[r].seedBegin, 0, 0, 0, 0, 0, 0, 0 };
	state[r] = st;
}

// Skip rules of AlignOneWay for one seed against alignments [aFrom, aTo) of the read (:147-173).
// Returns 0 extend, 1 skip, 2 the reference asserts.
__device__ inline int seedSkipped(const DGraph& g, const LongSeed& sd, const LongAln* mine, uint32_t aFrom, uint32_t aTo, const LongCell* cellPool)
{
	for (uint32_t a = aFrom; a < aTo; a++)                          // sloppy overlap rule (:147-161)
		if (mine[a].start <= sd.seqPos && mine[a].end >= sd.seqPos && mine[a].goodness > sd.goodness) return 1;
	int32_t compareNode = g.nodeIDs[sd.node];
	uint32_t compareOffset = g.nodeOffset[sd.node] + sd.offset;
	for (uint32_t a = aFrom; a < aTo; a++) {                        // exactAlignmentPart (:163-173)
		int on = onTrace(cellPool + mine[a].traceOff, mine[a].traceLen, sd.seqPos, compareNode, compareOffset);
		if (on == 2) return 2;
		if (on == 1) return 1;
	}
	return 0;
}

// One wave per read: advance to the next seed(s) that need extending and emit their work items. The skip rules of 64
// seeds are evaluated at once (each is a chain of dependent loads: a binary search in every alignment's trace), then
// the reference's in-order scan is replayed on the ballot masks. With maxCandidates > 1 (used for the tail rounds,
// when few reads are still active and the chip is idle) further seeds are emitted speculatively: they pass the skip
// rules against the alignments known now; k_long_merge re-checks each of them against the alignments added before it
// in the same round, so the outcome equals one-seed-per-round.
#define LONG_MAX_CANDIDATES 8
__device__ __forceinline__ void longSelectRead(const DGraph& g, const uint32_t r, const uint32_t lane, const LongJob* __restrict__ jobs, const LongSeed* __restrict__ seeds, uint32_t minClusterSize,
	uint32_t maxCandidates, LongState* __restrict__ state, const LongAln* __restrict__ alns, const LongCell* __restrict__ cellPool, LongWork* __restrict__ work, uint32_t* __restrict__ workLen, uint32_t* __restrict__ candSeed,
	unsigned long long* __restrict__ workCount, uint64_t workCapacity)
{
	LongState st = state[r];
	st.candCount = 0;
	if (st.status != 0) { if (lane == 0) state[r] = st; return; }
	LongJob job = jobs[r];
	const LongAln* mine = alns + job.alnBegin;
	uint32_t cand[LONG_MAX_CANDIDATES];
	uint32_t nCand = 0;
	uint32_t si = st.si;
	bool stop = false;
	while (!stop && si < job.seedEnd) {
		const uint32_t idx = si + lane;
		const bool valid = idx < job.seedEnd;
		LongSeed sd = seeds[valid ? idx : si];
		const bool cut = valid && sd.goodness < st.e2eScore;                     // aligned end to end (:127-131)
		const bool small = sd.clusterSize < minClusterSize;                       // :141-146
		int sk = 1;
		if (valid && !cut && !small) sk = seedSkipped(g, sd, mine, 0, st.nAln, cellPool);
		const uint64_t cutMask = __ballot(cut), assertMask = __ballot(sk == 2), candMask = __ballot(sk == 0);
		const uint32_t nValid = job.seedEnd - si < 64 ? job.seedEnd - si : 64;
		const uint32_t firstCut = cutMask ? (uint32_t)__ffsll((unsigned long long)cutMask) - 1 : 64;
		const uint32_t limit = firstCut < nValid ? firstCut : nValid;
		uint32_t b = 0;
		for (; b < limit; b++) {   // the reference's scan, in seed order (uniform across the wave)
			if ((assertMask >> b) & 1) { st.status = 1; stop = true; break; }
			if ((candMask >> b) & 1) {
				for (uint32_t k = 0; k < LONG_MAX_CANDIDATES; k++) if (k == nCand) cand[k] = si + b;
				nCand++;
				if (nCand == maxCandidates) { b++; stop = true; break; }
			}
		}
		si += b;
		if (!stop && firstCut < nValid) { si = job.seedEnd; stop = true; }
	}
	st.si = si;
	if (nCand > 0) {
		unsigned long long at = 0;
		if (lane == 0) at = atomicAdd(workCount, 2ull * nCand);
		at = __shfl(at, 0);
		if (at + 2ull * nCand > workCapacity) { st.status = 2; nCand = 0; }
		st.candBegin = (uint32_t)(at / 2);
		st.candCount = nCand;
		uint32_t myCand = 0;
		for (uint32_t k = 0; k < LONG_MAX_CANDIDATES; k++) if (k == lane) myCand = cand[k];
		if (lane < nCand) {
			const uint32_t c = lane;
			LongSeed sd = seeds[myCand];
			const uint32_t L = job.readLen, p = sd.seqPos;
			// backward: rows are revcomp(read[0..p)) = reverse-complement strand from position L-p; forward: read(p..] from p+1
			uint32_t twinNode, twinOffset;
			twinOf(g, sd.node, sd.offset, twinNode, twinOffset);
			work[at + 2 * c] = LongWork { job.maskOff + 4ull * job.maskWords, job.maskWords, L - p, p, twinNode, twinOffset, r };
			work[at + 2 * c + 1] = LongWork { job.maskOff, job.maskWords, p + 1, L - 1 - p, sd.node, sd.offset, r };
			workLen[at + 2 * c] = p;
			workLen[at + 2 * c + 1] = L - 1 - p;
			candSeed[at / 2 + c] = myCand;
		}
	}
	if (lane == 0) state[r] = st;
}
__global__ void __launch_bounds__(64) k_long_select(DGraph g, const LongJob* __restrict__ jobs, uint32_t nReads, const LongSeed* __restrict__ seeds, uint64_t rcBase, uint32_t minClusterSize,
	uint32_t maxCandidates, LongState* __restrict__ state, const LongAln* __restrict__ alns, const LongCell* __restrict__ cellPool, LongWork* __restrict__ work, uint32_t* __restrict__ workLen, uint32_t* __restrict__ candSeed,
	unsigned long long* __restrict__ workCount, uint64_t workCapacity)
{
	GC_RAISE_PRIO();
	const uint32_t r = blockIdx.x, lane = threadIdx.x;
	if (r >= nReads) return;
	longSelectRead(g, r, lane, jobs, seeds, minClusterSize, maxCandidates, state, alns, cellPool, work, workLen, candSeed, workCount, workCapacity);
}

// LANES = active lanes per wave ("team"). The pass is latency-bound and leaves most of the chip idle, so when there
// are fewer work items than the chip has SIMDs x 64 lanes, running fewer lanes per wave shortens every wave: a wave's
// instruction stream is the union of its lanes' divergent paths, and LDS per wave shrinks so more waves fit per CU.
#ifndef GC_LONG_MIN_WAVES
#define GC_LONG_MIN_WAVES 1
#endif
template <int LANES, bool PERSISTENT>
#ifndef GC_LONG_WAVES_ONE
#define GC_LONG_WAVES_ONE 5   // waves per SIMD the one-extension-per-wave instantiation is compiled for. 8: 64 VGPRs, 35 of them spilled to 112 B of scratch per lane; 7: 72 / 80 B; 6: 80 / 48 B;
                              // 5 (and 4): 87 VGPRs, no scratch. The kernel is bound by the CU's scalar unit, not by latency: all five measure the same, so the build without scratch is kept
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(LANES == 1 ? GC_LONG_WAVES_ONE : GC_LONG_MIN_WAVES, 8))) k_long_extend(DGraph g, const CorrectnessTables* __restrict__ ct, const uint64_t* __restrict__ masks, ExtendConfig cfg,
	const LongWork* __restrict__ work, const uint32_t* __restrict__ order, uint32_t nWork, unsigned long long* __restrict__ scratch, uint64_t wordsPerLane,
	unsigned long long* __restrict__ tracePool, unsigned long long* __restrict__ traceCursor, uint64_t traceCapacity, LongWorkResult* __restrict__ results, unsigned long long* __restrict__ counters,
	unsigned long long* __restrict__ nextSlot, uint32_t retryStatus, const unsigned long long* __restrict__ nWorkOnDevice, uint32_t* __restrict__ capListOut, unsigned long long* __restrict__ capCountOut)
{
	__shared__ WaveLdsT<LANES> lds;
	if (nWorkOnDevice) nWork = (uint32_t)*nWorkOnDevice;   // (the retry launch: its items are a list another kernel has just written)
	// One extension per wave (LANES == 1): all 64 lanes stay alive and run the same code on the same values - nothing depends
	// on the lane id, so the compiler keeps the extension's state in scalar registers - and the lanes' VGPRs hold the 64
	// backtrace columns. Stores hit one address with one value; atomics and the result record go through lane 0 only.
	if (LANES > 1 && threadIdx.x >= LANES) return;
	const uint32_t lane = LANES == 1 ? 0u : threadIdx.x;
	const bool leader = LANES > 1 || threadIdx.x == 0;
	WaveScratch wsx;
	wsx.base = scratch + (uint64_t)blockIdx.x * wordsPerLane * LANES;
	wsx.lane = lane;
	wsx.lanes = LANES;
	wsx.maxSlices = cfg.maxSlices; wsx.maxItems = cfg.maxItems; wsx.maxTrace = cfg.maxTrace;
	wsx.maxCols = LANES == 1 ? cfg.maxCols : 0;   // (the column store is laid out for one extension per wave; the region is reserved for every team size)
	wsx.allLanes = LANES == 1;
	wsx.regCap = cfg.regCap;
	ExtCounters cnt {};
	// One work item per wave while the launch fits the scratch (the host sizes it for up to 65536 lanes in flight under a memory budget); larger rounds run
	// persistent waves that fetch work items in execution order (longest first). The two are separate instantiations:
	// the fetch loop costs the common case 6 % (265 vs 283 ms on cfg2) in register pressure.
	bool done = false;
	while (true) {
		unsigned long long first = 0;
		if (!PERSISTENT) { if (done) break; done = true; first = (unsigned long long)blockIdx.x * LANES; }
		else {
			if (threadIdx.x == 0) first = atomicAdd(nextSlot, (unsigned long long)LANES);
			if (LANES > 1) first = __shfl(first, 0);
			else first = (unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)first) | ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(first >> 32)) << 32);   // an atomic's result is per-lane to the compiler: say it is uniform so the extension state stays in scalar registers
		}
		if (first >= nWork) break;
		const uint32_t slot = (uint32_t)first + lane;
		if (slot >= nWork) break;
		const uint32_t w = order[slot];   // execution order (longest first / length-balanced waves), results stay indexed by work item
		if (retryStatus != 0 && results[w].status != retryStatus) { if (PERSISTENT) continue; else break; }   // retry launch: only the items the first launch gave up on
		LongWork it = work[w];
		LongWorkResult res { 0, 0, EXT_FAILED, 0, 0 };
		if (it.seqLen > 0) {
			uint32_t nTrace = 0;
			int32_t score = 0;
			EqSource eqSrc { masks + it.maskOff, it.maskWords, it.startBit };
			res.status = extendSeedWave<LANES == 1>(g, *ct, eqSrc, cfg.bandwidth, (lds_u32*)&lds.w[0][0], wsx, (int)it.seqLen, it.node, it.offset, 0, nTrace, score, cnt);
			res.score = score;
			res.pad = cnt.flattenTie;   // (k_long_merge adds the flags of the extensions the reference would have run to the read's count)
			if (res.status == EXT_OK) {
				unsigned long long base = 0;
				if (leader) base = atomicAdd(traceCursor, (unsigned long long)nTrace);
				if (LANES == 1) base = (unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)base) | ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32);
				if (base + nTrace <= traceCapacity) {
					if (LANES == 1) { for (uint32_t i = threadIdx.x; i < nTrace; i += 64) tracePool[base + i] = wsx.word(wsx.traceBase(i, 0)); }
					else for (uint32_t i = 0; i < nTrace; i++) tracePool[base + i] = wsx.word(wsx.traceBase(i, 0));
					res.traceOff = base;
					res.traceLen = nTrace;
				} else res.status = EXT_OVERFLOW;
			}
		}
		if (leader) results[w] = res;
		// (the work items whose band outgrew this layout's tables go on the list of the retry launch: almost always none - the list saves a kernel
		// that looked at every result, and the retry is a handful of waves)
		if (leader && capListOut && res.status == EXT_LDS_CAP) capListOut[atomicAdd(capCountOut, 1ull)] = w;
	}
	if (cnt.extensions && leader) {
		atomicAdd(&counters[0], cnt.dpTiles);
		atomicAdd(&counters[1], cnt.recomputeTiles);
		atomicAdd(&counters[2], cnt.columnSteps);
		atomicAdd(&counters[3], cnt.traceItems);
		atomicAdd(&counters[4], cnt.extensions);
		atomicAdd(&counters[5], cnt.backtraceTiles);
#ifdef GC_STAMPS
		for (int i = 0; i < 16; i++) atomicAdd(&counters[8 + i], cnt.cyc[i]);
#endif
	}
}

#ifdef GC_EXPERIMENTS   // (`make -C graphchainer_amd/csrc experiments`: measured and rejected alternatives are not part of the product library)
// The measurement VERDICT r3 asked for (GC_LONG_LANE=1, experiments build): the same rounds, but every LANE takes one work item and runs the
// plain-layout core (extendSeedT, gc_device.hpp: the core of k_extend and k_long_pass) with its band state in a per-lane HBM slab - no LDS tables, no
// state machine, <= 128 VGPRs (4 waves per SIMD). Work items arrive longest first (k_long_order), so a wave's 64 extensions have about the same number of
// slices. What outgrows the slab answers EXT_OVERFLOW and its read goes to the plain-layout fallback like any other overflow.
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) k_long_extend_lane(DGraph g, const CorrectnessTables* __restrict__ ct, const uint64_t* __restrict__ masks, ExtendConfig cfg,
	const LongWork* __restrict__ work, const uint32_t* __restrict__ order, uint32_t nWork, uint8_t* __restrict__ scratch, uint64_t slabBytes,
	unsigned long long* __restrict__ tracePool, unsigned long long* __restrict__ traceCursor, uint64_t traceCapacity, LongWorkResult* __restrict__ results, unsigned long long* __restrict__ counters)
{
	const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t stride = gridDim.x * blockDim.x;
	LaneScratch sc = laneScratch(scratch + (uint64_t)tid * slabBytes, cfg);
	ExtCounters cnt {};
	for (uint32_t slot = tid; slot < nWork; slot += stride) {
		const uint32_t w = order[slot];
		LongWork it = work[w];
		LongWorkResult res { 0, 0, EXT_FAILED, 0, 0 };
		if (it.seqLen > 0) {
			uint32_t nTrace = 0;
			int32_t score = 0;
			EqSource eqSrc { masks + it.maskOff, it.maskWords, it.startBit };
			res.status = extendSeedT(g, *ct, eqSrc, cfg, sc, (int)it.seqLen, it.node, it.offset, nTrace, score, cnt);
			res.score = score;
			res.pad = cnt.flattenTie;
			if (res.status == EXT_OK) {
				unsigned long long base = atomicAdd(traceCursor, (unsigned long long)nTrace);
				if (base + nTrace <= traceCapacity) {
					for (uint32_t i = 0; i < nTrace; i++) {
						const TraceCell c = sc.trace[i];
						tracePool[base + i] = packCell(Cell { c.node, c.offsetAndSwitch & 255u, c.seqPos }, (c.offsetAndSwitch >> 8) & 1u);
					}
					res.traceOff = base;
					res.traceLen = nTrace;
				} else res.status = EXT_OVERFLOW;
			}
		}
		results[w] = res;
	}
	if (cnt.extensions) {
		atomicAdd(&counters[0], cnt.dpTiles);
		atomicAdd(&counters[1], cnt.recomputeTiles);
		atomicAdd(&counters[2], cnt.columnSteps);
		atomicAdd(&counters[3], cnt.traceItems);
		atomicAdd(&counters[4], cnt.extensions);
		atomicAdd(&counters[5], cnt.backtraceTiles);
	}
}

#endif

// One wave per read: the decisions are taken redundantly by all lanes (uniform control flow, lane 0 does the single
// writes), the trace -> cell conversion - the bulk of the work, ~1.5 cells per read base - runs 64 cells at a time.
__device__ __forceinline__ void longMergeRead(const DGraph& g, const uint32_t r, const uint32_t lane, const LongJob* __restrict__ jobs, const LongSeed* __restrict__ seeds, const uint32_t* __restrict__ candSeed,
	const LongWorkResult* __restrict__ results, const unsigned long long* __restrict__ tracePool, uint32_t maxAlignments,
	LongState* state, LongAln* alns, LongCell* cellPool, unsigned long long* __restrict__ cellCursor, uint64_t cellCapacity)
{
	LongState st = state[r];
	if (st.candCount == 0 || st.status != 0) return;
	LongJob job = jobs[r];
	LongAln* mine = alns + job.alnBegin;
	const int L = (int)job.readLen;
	const uint32_t nAlnAtSelect = st.nAln;
	for (uint32_t c = 0; c < st.candCount && st.status == 0; c++) {
		const uint32_t pair = st.candBegin + c;
		const uint32_t seedIdx = candSeed[pair];
		LongSeed sd = seeds[seedIdx];
		if (c > 0) {
			// re-check against what this round added before this candidate (the select kernel checked the older alignments)
			if (sd.goodness < st.e2eScore) { st.si = job.seedEnd; break; }
			int sk = seedSkipped(g, sd, mine, nAlnAtSelect, st.nAln, cellPool);
			if (sk == 2) { st.status = 1; break; }
			if (sk == 1) continue;
		}
		st.extended++;
		const LongWorkResult rb = results[2 * pair], rf = results[2 * pair + 1];
		const int p = (int)sd.seqPos;
		bool runB = p > 0, runF = p < L - 1;
		uint32_t stB = runB ? rb.status : EXT_FAILED, stF = runF ? rf.status : EXT_FAILED;
		if (runB) st.pad1 += rb.pad & 1u;                                              // LongState::pad1 = the read's flatten ties (gc_result::flatten_ties_long)
		if (runF && stB != EXT_ASSERT) st.pad1 += rf.pad & 1u;                         // (the forward extension only runs when the backward one did not throw)
		if (stB == EXT_ASSERT || stF == EXT_ASSERT) { st.status = 1; break; }          // getAlignmentFromSeed threw
		if (stB == EXT_OVERFLOW || stF == EXT_OVERFLOW) { st.status = 2; break; }
		if (stB == EXT_LDS_CAP || stF == EXT_LDS_CAP) { st.status = 5; break; }
		bool hasB = stB == EXT_OK, hasF = stF == EXT_OK;
		if (!hasB && !hasF) continue;   // alignmentFailed()
		uint32_t nB = rb.traceLen, nF = rf.traceLen;
		uint32_t useB = hasB ? (hasF ? nB - 1 : nB) : 0;
		uint32_t total = useB + (hasF ? nF : 0);
		if (st.nAln >= maxAlignments) { st.status = 3; break; }
		unsigned long long base = 0;
		if (lane == 0) base = atomicAdd(cellCursor, (unsigned long long)total);
		base = __shfl(base, 0);
		if (base + total > cellCapacity) { st.status = 4; break; }
		LongCell* outCells = cellPool + base;
		for (uint32_t i = lane; i < useB; i += 64) {   // fixReverseTraceSeqPosAndOrder (:543-565)
			TraceCell tc = unpackCell(tracePool[rb.traceOff + i]);
			uint32_t off = tc.offsetAndSwitch & 255u;
			int32_t id = g.nodeIDs[tc.node];
			uint32_t orig = g.nodeOffset[tc.node] + off;
			LongCell oc;
			oc.node = id ^ 1;
			oc.offset = g.origSize[id] - 1 - orig;
			oc.seqPos = (uint32_t)(p - 1 - tc.seqPos);
			oc.nodeSwitch = (i + 1 < nB) ? ((unpackCell(tracePool[rb.traceOff + i + 1]).offsetAndSwitch >> 8) & 1u) : 0u;
			outCells[i] = oc;
		}
		if (hasF) for (uint32_t i = lane; i < nF; i += 64) {   // fixForwardTraceSeqPos (:527-540), device order reversed
			TraceCell tc = unpackCell(tracePool[rf.traceOff + (nF - 1 - i)]);
			LongCell oc;
			oc.node = g.nodeIDs[tc.node];
			oc.offset = g.nodeOffset[tc.node] + (tc.offsetAndSwitch & 255u);
			oc.seqPos = (uint32_t)(p + 1 + tc.seqPos);
			oc.nodeSwitch = (tc.offsetAndSwitch >> 8) & 1u;
			outCells[useB + i] = oc;
		}
		__threadfence_block();   // the cells are read back below and by the next candidate's skip rules (one wave per read: the fence orders its own stores and loads)
		LongAln al;
		al.start = outCells[0].seqPos;
		al.end = outCells[total - 1].seqPos + 1;
		al.score = (uint32_t)((hasB ? rb.score : 0) + (hasF ? rf.score : 0));
		al.goodness = sd.goodness;
		al.traceOff = base;
		al.traceLen = total;
		al.pad = 0;
		uint32_t e2e = 0;
		if (lane == 0) {
			mine[st.nAln] = al;
			e2e = endToEndScore(mine, st.nAln + 1, (uint32_t)L, st.e2eScore);
		}
		st.nAln++;
		st.e2eScore = __shfl(e2e, 0);
		__threadfence_block();
	}
	st.candCount = 0;
	if (lane == 0) state[r] = st;
}

__global__ void __launch_bounds__(64) k_long_merge(DGraph g, const LongJob* __restrict__ jobs, uint32_t nReads, const LongSeed* __restrict__ seeds, const uint32_t* __restrict__ candSeed,
	const LongWorkResult* __restrict__ results, const unsigned long long* __restrict__ tracePool, uint32_t maxAlignments,
	LongState* state, LongAln* alns, LongCell* cellPool, unsigned long long* __restrict__ cellCursor, uint64_t cellCapacity)
{
	GC_RAISE_PRIO();
	const uint32_t r = blockIdx.x;
	if (r >= nReads) return;
	longMergeRead(g, r, threadIdx.x, jobs, seeds, candSeed, results, tracePool, maxAlignments, state, alns, cellPool, cellCursor, cellCapacity);
}

// Execution order of a round's work items: longest extensions first (a counting sort over 1024 length classes, one block;
// the order inside a class is whatever the atomics give - results do not depend on it). mode 0: identity.
#ifndef GC_ORDER_THREADS
#define GC_ORDER_THREADS 256    // threads of the ordering block (r5: 256 - under config 5 the 1 024-thread block waited up to 150 ms per round for sixteen free wave slots on one CU, with the whole-read token held; r4 note: (-DGC_ORDER_THREADS=256: a wave scan of the histogram and a block that finds its wave slots sooner among five batches' kernels - measured, 147.8 / 151.7 / 150.5 ms per batch against 150.8 / 150.2 / 149.8, `gpurun_out/r4_ord`: no effect)
#endif
__global__ void __launch_bounds__(GC_ORDER_THREADS) k_long_order(const uint32_t* __restrict__ workLen, const unsigned long long* __restrict__ workCount, uint32_t* __restrict__ order, uint32_t shift, uint32_t mode)
{
	GC_RAISE_PRIO();
	__shared__ uint32_t hist[1024];
	__shared__ uint32_t start[1024];
	const uint32_t n = (uint32_t)*workCount, tid = threadIdx.x, T = GC_ORDER_THREADS;
	if (mode == 0) { for (uint32_t i = tid; i < n; i += T) order[i] = i; return; }
	for (uint32_t b = tid; b < 1024; b += T) hist[b] = 0;
	__syncthreads();
	for (uint32_t i = tid; i < n; i += T) { uint32_t b = workLen[i] >> shift; atomicAdd(&hist[1023 - (b < 1023 ? b : 1023)], 1u); }
	__syncthreads();
#if GC_ORDER_THREADS == 1024
	if (tid == 0) { uint32_t at = 0; for (uint32_t b = 0; b < 1024; b++) { start[b] = at; at += hist[b]; } }
#else
	if (tid < 64) {   // the first wave: sixteen classes per lane, a wave scan of the lanes' sums
		uint32_t mine[16], sum = 0;
		for (uint32_t k = 0; k < 16; k++) { mine[k] = hist[tid * 16 + k]; sum += mine[k]; }
		uint32_t incl = sum;
		for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if ((int)tid >= d) incl += o; }
		uint32_t at = incl - sum;
		for (uint32_t k = 0; k < 16; k++) { start[tid * 16 + k] = at; at += mine[k]; }
	}
#endif
	__syncthreads();
	for (uint32_t i = tid; i < n; i += T) { uint32_t b = workLen[i] >> shift; order[atomicAdd(&start[1023 - (b < 1023 ? b : 1023)], 1u)] = i; }
}

#ifdef GC_EXPERIMENTS
// One launch per round instead of five (r4): the previous round's merge, this round's select, the execution order and the hand-over of the round's work count.
// One wave per read runs the read's merge and then its select (both only touch that read's state); every wave then takes a ticket, and the wave that takes the
// last one - all work items of the round are in place by then - sorts them into execution order (longest first, 64 lanes over an LDS histogram of 1024 length
// classes) and publishes the count: to the device (roundInfo[round]: the next round's speculation rule reads it, and the extension kernel takes its item count from the
// round's cursor set) and to pinned host memory. The cursors come in two sets that alternate with the round's parity; a round zeroes the NEXT round's set (its last
// users - the previous round's select and extension kernels - are complete in stream order). With that the host can queue several rounds without waiting for any
// count: between two extension kernels of a pass sit this kernel and the (almost always empty) retry, not zero / select / order / publish / host round trip / merge,
// each of which waited for a wave slot among the other batches' kernels (r3: 28 ms of a pass's 149 with five batches in flight).
__global__ void __launch_bounds__(64) k_long_round(DGraph g, const LongJob* __restrict__ jobs, uint32_t nReads, const LongSeed* __restrict__ seeds, uint32_t minClusterSize, uint32_t round, uint32_t forceCand,
	uint32_t gridLimit, LongState* state, LongAln* alns, LongCell* cellPool, unsigned long long* cellCursor, uint64_t cellCapacity, uint32_t maxAlignments,
	LongWork* work, uint32_t* workLen, uint32_t* candSeed, const LongWorkResult* results, const unsigned long long* tracePool,
	unsigned long long* cursorSets, unsigned long long* roundInfo, unsigned long long* ticket, uint32_t* order, uint32_t orderShift, uint32_t orderMode,
	volatile unsigned long long* hostInfo, uint64_t workCapacity)
{
	__shared__ uint32_t hist[1024];
	const uint32_t r = blockIdx.x, lane = threadIdx.x;
	unsigned long long* cur = cursorSets + 4 * (round & 1u);
	unsigned long long* next = cursorSets + 4 * ((round + 1u) & 1u);
	if (r == 0 && lane < 4) next[lane] = 0;
	// candidates per read (the host loop's rule, on the device): one; several once few reads are still active (exact: the merge re-checks them in order), within
	// what the work arrays, the trace budget (four seeds' worth per read) and the extension launch's grid hold
	uint32_t maxCand = 1;
	if (round > 0) {
		const uint64_t lastWork = roundInfo[round - 1];
		if (lastWork > 0) {
			maxCand = (uint32_t)min((uint64_t)8, max((uint64_t)1, (2ull * nReads) / lastWork));
			if (lastWork < 8192) maxCand = 8;
			maxCand = (uint32_t)min((uint64_t)maxCand, max((uint64_t)1, (8ull * nReads) / lastWork));
			maxCand = (uint32_t)min((uint64_t)maxCand, max((uint64_t)1, (uint64_t)gridLimit / lastWork));   // at most lastWork / 2 reads are active: 2 x maxCand items each
		}
	}
	if (forceCand) maxCand = forceCand;
	if (r < nReads) {
		// the candidates' seed indices are double-buffered by the round's parity: this read's select writes the new round's list while other reads' merges still read the previous round's
		// (everything else a select writes - work items, lengths - no merge reads; the extension results and traces a merge reads are written by the extension kernels, between the rounds)
		if (round > 0) { longMergeRead(g, r, lane, jobs, seeds, candSeed + (uint64_t)((round - 1u) & 1u) * workCapacity, results, tracePool, maxAlignments, state, alns, cellPool, cellCursor, cellCapacity); __threadfence_block(); }
		longSelectRead(g, r, lane, jobs, seeds, minClusterSize, maxCand, state, alns, cellPool, work, workLen, candSeed + (uint64_t)(round & 1u) * workCapacity, cur, workCapacity);
	}
	__threadfence();   // this wave's work items before its ticket
	uint32_t last = 0;
	if (lane == 0) last = atomicAdd(ticket, 1ull) == (unsigned long long)gridDim.x - 1 ? 1u : 0u;
	last = __shfl(last, 0);
	if (!last) return;
	__threadfence();
	const uint32_t n = (uint32_t)atomicAdd(cur, 0ull);
	const volatile uint32_t* lens = workLen;
	if (orderMode == 0) { for (uint32_t i = lane; i < n; i += 64) order[i] = i; }
	else {
		for (uint32_t b = lane; b < 1024; b += 64) hist[b] = 0;
		__syncthreads();
		for (uint32_t i = lane; i < n; i += 64) { const uint32_t b = lens[i] >> orderShift; atomicAdd(&hist[1023 - (b < 1023 ? b : 1023)], 1u); }
		__syncthreads();
		// exclusive scan of the 1024 classes: 16 per lane, a wave scan of the lanes' sums
		uint32_t mine[16], sum = 0;
		for (int k = 0; k < 16; k++) { mine[k] = hist[lane * 16 + k]; sum += mine[k]; }
		uint32_t incl = sum;
		for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if ((int)lane >= d) incl += o; }
		uint32_t at = incl - sum;
		for (int k = 0; k < 16; k++) { hist[lane * 16 + k] = at; at += mine[k]; }
		__syncthreads();
		for (uint32_t i = lane; i < n; i += 64) { const uint32_t b = lens[i] >> orderShift; order[atomicAdd(&hist[1023 - (b < 1023 ? b : 1023)], 1u)] = i; }
	}
	if (lane == 0) {
		roundInfo[round] = n;
		*ticket = 0;
		hostInfo[2 + round] = n;
		__threadfence_system();
		hostInfo[0] = round + 1;   // rounds published so far
		__threadfence_system();
	}
}

#endif

// copies a few cursor words into pinned host memory through the compute queue (a copy-engine transfer would queue behind bulk uploads)
__global__ void k_publish(const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst, uint32_t nWords)
{
	GC_RAISE_PRIO();
	if (threadIdx.x < nWords) { dst[threadIdx.x] = src[threadIdx.x]; __threadfence_system(); }
}

// a few words set to zero by a kernel: a hipMemsetAsync inside the round loop may go through a copy engine and queue behind another
// batch's bulk uploads
__global__ void k_zero_words(unsigned long long* __restrict__ dst, uint32_t nWords)
{
	GC_RAISE_PRIO();
	if (threadIdx.x < nWords) dst[threadIdx.x] = 0;
}

__global__ void __launch_bounds__(256) k_long_finish(uint32_t nReads, const LongState* __restrict__ state, LongReadResult* __restrict__ results)
{
	GC_RAISE_PRIO();
	uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= nReads) return;
	LongState st = state[r];
	LongReadResult rr { st.status == 1 ? 0u : st.nAln, st.extended, st.status, st.pad1 };   // pad = the read's flatten ties   // a throwing AlignOneWay returns nothing (src/Aligner.cpp:585-592)
	results[r] = rr;
}

// =====================================================================================================
// launchers
// =====================================================================================================

void launchSeedLookup(hipStream_t stream, const SeedIndex& idx, const char* bases, const uint64_t* readOff, uint32_t nReads,
	uint64_t* matchCursor, uint32_t* readMatchOff, uint32_t* readMatchCount, uint2* matches, uint64_t matchCapacity, uint32_t* tmp, uint64_t totalBases, const uint32_t* chunkRead, const uint64_t* packed, const uint64_t* invalid)
{
	if (nReads == 0) return;
	uint64_t probeBlocks = (totalBases + 255) / 256;
	if (probeBlocks > 65536) probeBlocks = 65536;
	if (probeBlocks) hipLaunchKernelGGL(k_seed_probe, dim3((uint32_t)probeBlocks), dim3(256), 0, stream, idx, bases, readOff, chunkRead, packed, invalid, nReads, totalBases, tmp);
	uint32_t blocks = nReads < 16384 ? nReads : 16384;
	hipLaunchKernelGGL(k_seed_compact, dim3(blocks), dim3(64), 0, stream, readOff, nReads, tmp, matchCursor, readMatchOff, readMatchCount, matches, matchCapacity);
}

uint64_t extendSlabBytes(const ExtendConfig& cfg)
{
	uint64_t b = sizeof(SliceInfo) * (uint64_t)cfg.maxSlices + sizeof(NodeItem) * (uint64_t)cfg.maxItems + sizeof(Pending) * (uint64_t)cfg.maxPending + sizeof(WCol) * 64 + sizeof(TraceCell) * (uint64_t)cfg.maxTrace + sizeof(uint32_t) * (uint64_t)cfg.maxItems;
	return (b + 63) & ~63ull;
}

uint32_t extendGridLanes(uint32_t nWork)
{
	// 256 CUs x 16 resident waves is plenty to hide latency for this register-heavy kernel; never more lanes than work
	uint32_t lanes = 256u * 16u * 64u;
	uint32_t need = (nWork + 63) / 64 * 64;
	return need < lanes ? need : lanes;
}

void launchExtend(hipStream_t stream, const DGraph& g, const CorrectnessTables* ct, const uint8_t* iupac, const ExtendConfig& cfg,
	const ExtItem* work, uint32_t nWork, const char* bases, ExtResult* results, uint8_t* scratch, uint64_t slabBytes,
	PoolCell* tracePool, unsigned long long* traceCursor, uint64_t traceCapacity, unsigned long long* counters, uint32_t retryStatus, uint32_t retryLanes, ExtSelection sel)
{
	if (nWork == 0) return;
	const uint32_t upper = sel.mode == 1 ? 2 * sel.nFrags : nWork;   // (a device-side list holds at most nWork items; waves beyond its count leave at once)
	uint32_t lanes = retryStatus ? retryLanes : extendGridLanes(upper);
	hipLaunchKernelGGL(k_extend_slab, dim3(lanes / 64), dim3(64), 0, stream, g, ct, iupac, cfg, work, nWork, bases, results, scratch, slabBytes, tracePool, traceCursor, traceCapacity, counters, retryStatus, sel);
}

void launchBuildAnchors(hipStream_t stream, const DGraph& g, const Fragment* frags, uint32_t nFrags, const FragSeed* seeds, const ExtResult* ext,
	const PoolCell* tracePool, int32_t splitLen, AnchorRec* anchors, uint32_t* fragStatus, uint32_t* fragExtended,
	uint32_t* pathPool, unsigned long long* pathCursor, uint64_t pathCapacity, AnchorRounds rounds, uint32_t* readTies)
{
	if (nFrags == 0) return;
	hipLaunchKernelGGL(k_build_anchors, dim3((nFrags + 63) / 64), dim3(64), 0, stream, g, frags, nFrags, seeds, ext, tracePool, splitLen, anchors, fragStatus, fragExtended, pathPool, pathCursor, pathCapacity, rounds, readTies);
}

uint64_t chainScratchBytes(const ChainCaps& caps)
{
	// the scratch launch's arrays for one read: anchors (capAnchors), entries (capEndpoints), threshold table (capTable)
	uint64_t b = 8ull * caps.capBack + 8ull * caps.capAnchors + 8ull * caps.capEndpoints + 4ull * caps.capAnchors + 4ull * (caps.capAnchors + 1) + 4ull * caps.capTable + 6ull * (caps.capAnchors + 1) + 256;
	b += 6ull * (caps.capAnchors + 1) + 8ull * caps.capEndpoints + 64;   // r4: bucket of every anchor, its entries' place by component, the entries in that order; r5: the anchors by bucket
	return (b + 63) & ~63ull;
}

uint32_t chainGridBlocks(uint32_t nReads) { return nReads < 2048 ? nReads : 2048; }   // three blocks fit a CU (LDS); every block owns a threshold-list region in HBM
uint32_t chainScratchBlocks(uint32_t nReads) { return nReads < 1024 ? nReads : 1024; }   // (r5: 1 024 - every block owns ~1.8 MB of scratch at config 5's read sizes, and the scratch launch is short since its DP runs per component)  r4:   // reads that do not fit the LDS tables: few on 10 kb reads (waves whose read is done leave at once), ALL of them on 50 kb reads (config 5: 256 blocks took 727 ms per 2 000 reads)

bool chainLdsLaunch(uint32_t fewestSlots, bool forceScratch) { return forceScratch || fewestSlots <= 2 * CHAIN_LDS_ANCHORS; }

void launchChain(hipStream_t stream, const DGraph& g, const ReadChainJob* jobs, uint32_t nReads, const AnchorRec* anchors, const Fragment* frags, const uint32_t* fragStatus,
	int32_t splitLen, int32_t splitGap, ChainCaps caps, uint8_t* scratch, uint32_t* chainOut, uint32_t* chainLen, unsigned long long* chainScore, uint32_t* chainStatus, bool forceScratch, uint32_t fewestSlots)
{
	if (nReads == 0) return;
	// (r5) fewestSlots: the batch's smallest read in anchor slots. When even that one is beyond twice the large LDS class, the LDS launch would send every read on - 2 048 blocks of
	// 53 KB that waited up to 570 ms for LDS room among config 5's kernels to do nothing (`gpurun_out/r5_cfg5_c`): skipped, the scratch launch takes every read
	const bool ldsLaunch = chainLdsLaunch(fewestSlots, forceScratch);
	const uint32_t scratchFlags = ((getenv("GC_CHAIN_PLAIN_SCAN") && atoi(getenv("GC_CHAIN_PLAIN_SCAN")) == 1) ? 2u : 0u) | (ldsLaunch ? 0u : 4u);
	// the half-size LDS class when the batch's largest read fits it (its entries are checked per read: a read with more goes to the scratch launch below)
	static const bool largeOnly = getenv("GC_CHAIN_LARGE") && atoi(getenv("GC_CHAIN_LARGE"));   // (the r3 launch, for A/B)
	const bool small = !largeOnly && caps.capAnchors <= CHAIN_LDS_ANCHORS / 2 && caps.capTable <= CHAIN_LDS_WIDTH / 2;
	if (!ldsLaunch) {}
	else if (small) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chain<2>), dim3(chainGridBlocks(nReads)), dim3(64), 0, stream, g, jobs, nReads, anchors, frags, fragStatus, splitLen, splitGap, caps, scratch, chainScratchBytes(caps), chainOut, chainLen, chainScore, chainStatus, forceScratch ? 1u : 0u);
	else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chain<1>), dim3(chainGridBlocks(nReads)), dim3(64), 0, stream, g, jobs, nReads, anchors, frags, fragStatus, splitLen, splitGap, caps, scratch, chainScratchBytes(caps), chainOut, chainLen, chainScore, chainStatus, forceScratch ? 1u : 0u);
	// reads with more anchors / entries than the LDS tables hold, or on a cover wider than the LDS threshold table (waves whose read is done leave at once).
	// The batch's bounds tell when no read can need it (cfg2: 400 slots per read at most, cover width 2): 2 048 waves that look and leave cost 7 ms of queueing per batch.
	// (it is then a safety net of 64 waves for what the bounds do not show - a graph with more than 65 535 components, a read beyond 65 535 fragment positions; r5: eight waves
	// took 9-12 ms to look at 10 000 reads' status words, two dependent loads per read, in the fragment pipeline's critical path)
	const bool cannotBeNeeded = !forceScratch && caps.capAnchors <= CHAIN_LDS_ANCHORS / (small ? 2 : 1) && caps.capEndpoints <= CHAIN_LDS_ENTRIES / (small ? 2 : 1) && caps.capTable <= CHAIN_LDS_WIDTH / (small ? 2 : 1);
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chain<0>), dim3(cannotBeNeeded ? (nReads < 64 ? nReads : 64) : chainScratchBlocks(nReads)), dim3(64), 0, stream, g, jobs, nReads, anchors, frags, fragStatus, splitLen, splitGap, caps, scratch, chainScratchBytes(caps), chainOut, chainLen, chainScore, chainStatus, scratchFlags);
}

uint64_t longWaveWordsPerLane(const ExtendConfig& cfg) { return waveScratchWords(cfg.maxSlices, cfg.maxItems, cfg.maxTrace, cfg.maxCols); }

void launchLongInit(hipStream_t stream, const LongJob* jobs, uint32_t nReads, LongState* state)
{
	if (nReads) hipLaunchKernelGGL(k_long_init, dim3((nReads + 255) / 256), dim3(256), 0, stream, jobs, nReads, state);
}
void launchLongSelect(hipStream_t stream, const DGraph& g, const LongJob* jobs, uint32_t nReads, const LongSeed* seeds, uint64_t rcBase, uint32_t minClusterSize, uint32_t maxCandidates,
	LongState* state, const LongAln* alns, const LongCell* cellPool, LongWork* work, uint32_t* workLen, uint32_t* candSeed, unsigned long long* workCount, uint64_t workCapacity)
{
	if (maxCandidates < 1) maxCandidates = 1;
	if (maxCandidates > LONG_MAX_CANDIDATES) maxCandidates = LONG_MAX_CANDIDATES;
	if (nReads) hipLaunchKernelGGL(k_long_select, dim3(nReads), dim3(64), 0, stream, g, jobs, nReads, seeds, rcBase, minClusterSize, maxCandidates, state, alns, cellPool, work, workLen, candSeed, workCount, workCapacity);
}
uint32_t longExtendTeamSize(uint32_t nWork)
{
	// Measured on MI355X (cfg2, 20k extensions in the first round): 1 lane per wave 260 ms, 2 lanes 352 ms, 4 lanes 397 ms,
	// 8 lanes 555 ms for the whole pass. The extension core is branchy serial code; lanes sharing a wave pay for the
	// union of their paths, and the chip has far more wave slots than a round has extensions. GC_TEST_LONG_TEAM overrides.
	if (const char* env = getenv("GC_TEST_LONG_TEAM")) { int v = atoi(env); if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64) return (uint32_t)v; }
	(void)nWork;
	return 1;
}

void launchLongExtend(hipStream_t stream, const DGraph& g, const CorrectnessTables* ct, const uint64_t* masks, const ExtendConfig& cfg, const LongWork* work, const uint32_t* order, uint32_t nWork,
	unsigned long long* scratch, uint32_t lanes, uint32_t blocks, unsigned long long* tracePool, unsigned long long* traceCursor, uint64_t traceCapacity, LongWorkResult* results, unsigned long long* counters,
	unsigned long long* nextSlot, uint32_t retryStatus, const unsigned long long* nWorkOnDevice, uint32_t* capListOut, unsigned long long* capCountOut, bool gridCoversCount)
{
	if (!nWork) return;
	uint64_t words = longWaveWordsPerLane(cfg);
	// fewer lanes than work items (or a count only the device knows): waves loop and fetch - unless the caller vouches that the grid covers whatever the count turns out to be
	// (gridCoversCount: the sync-free round loop, whose launches are sized by a bound; waves beyond the count leave at once)
	const bool persistent = (nWorkOnDevice && !gridCoversCount) || (uint64_t)blocks * lanes < nWork;
	// GC_LONG_WAVES_PER_SIMD=w (experiment): an unused dynamic LDS allocation per wave caps the kernel at w waves per SIMD, leaving wave slots
	// and registers to the fragment pipeline's kernels that share the device with it
#ifdef GC_EXPERIMENTS
	static const uint32_t ldsPad = []() { const char* e = getenv("GC_LONG_WAVES_PER_SIMD"); int w = e ? atoi(e) : 0; return (w >= 1 && w <= 7) ? (uint32_t)((160u * 1024u / (4u * (uint32_t)w)) & ~255u) : 0u; }();
	const uint32_t pad = lanes == 1 ? ldsPad : 0;
#else
	const uint32_t pad = 0;
#endif
#define GC_LAUNCH_TEAM(N) do { if (persistent) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_long_extend<N, true>), dim3(blocks), dim3(64), pad, stream, g, ct, masks, cfg, work, order, nWork, scratch, words, tracePool, traceCursor, traceCapacity, results, counters, nextSlot, retryStatus, nWorkOnDevice, capListOut, capCountOut); \
	else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_long_extend<N, false>), dim3(blocks), dim3(64), pad, stream, g, ct, masks, cfg, work, order, nWork, scratch, words, tracePool, traceCursor, traceCapacity, results, counters, nextSlot, retryStatus, nWorkOnDevice, capListOut, capCountOut); } while (0)
	switch (lanes) {
		case 1: GC_LAUNCH_TEAM(1); break;
		case 2: GC_LAUNCH_TEAM(2); break;
		case 4: GC_LAUNCH_TEAM(4); break;
		case 8: GC_LAUNCH_TEAM(8); break;
		case 16: GC_LAUNCH_TEAM(16); break;
		case 32: GC_LAUNCH_TEAM(32); break;
		default: GC_LAUNCH_TEAM(64); break;
	}
#undef GC_LAUNCH_TEAM
}
#ifdef GC_EXPERIMENTS
void launchLongExtendLane(hipStream_t stream, const DGraph& g, const CorrectnessTables* ct, const uint64_t* masks, const ExtendConfig& cfg, const LongWork* work, const uint32_t* order, uint32_t nWork,
	uint8_t* scratch, uint64_t scratchBytes, unsigned long long* tracePool, unsigned long long* traceCursor, uint64_t traceCapacity, LongWorkResult* results, unsigned long long* counters)
{
	if (!nWork) return;
	const uint64_t slab = extendSlabBytes(cfg);
	uint64_t lanes = ((uint64_t)nWork + 63) / 64 * 64;
	const uint64_t fit = scratchBytes / slab / 64 * 64;   // lanes whose slabs fit the scratch: a larger round strides
	if (lanes > fit) lanes = fit;
	if (lanes > 256ull * 16 * 64) lanes = 256ull * 16 * 64;
	if (lanes == 0) return;
	hipLaunchKernelGGL(k_long_extend_lane, dim3((uint32_t)(lanes / 64)), dim3(64), 0, stream, g, ct, masks, cfg, work, order, nWork, scratch, slab, tracePool, traceCursor, traceCapacity, results, counters);
}
#endif
// Work items of the fragment pass, built where they are used (src/GraphAligner.h:499-511 per seed of a fragment window): the host sorts each
// read's seeds and cuts the windows (order-critical, host/gc_glue.cpp) and hands over 16 B per fragment and per seed; one thread per fragment
// expands its seed window into the per-slot records - the seed in fragment order and the two extensions (backward: reverse complement of the
// fragment's prefix from the seed's reverse-strand twin; forward: the suffix from the seed). On the host this was 70 ms per 10 k reads
// (345 MB of records through pinned memory, a twin lookup per slot) on the fragment pipeline's critical path, plus the upload.
__global__ void __launch_bounds__(256) k_build_fragment_work(DGraph g, const Fragment* __restrict__ frags, const uint32_t* __restrict__ fragFirstSeed, uint32_t nFrags,
	const FragSeed* __restrict__ readSeeds, const uint64_t* __restrict__ readOffsets, uint64_t totalBases, uint32_t splitLen, FragSeed* __restrict__ fragSeeds, ExtItem* __restrict__ work,
	ExtResult* __restrict__ results)
{
	const uint32_t F = blockIdx.x * 256 + threadIdx.x;
	if (F >= nFrags) return;
	const Fragment fr = frags[F];
	const uint64_t readOff = readOffsets[fr.read], len = readOffsets[fr.read + 1] - readOff;
	const FragSeed* seeds = readSeeds + fragFirstSeed[F];
	for (uint32_t k = 0; k < fr.seedEnd - fr.seedBegin; k++) {
		const FragSeed s = seeds[k];
		const uint32_t slot = fr.seedBegin + k, p = s.seqPos - fr.l;
		fragSeeds[slot] = s;
		ExtItem b;
		b.seqOff = totalBases + readOff + (len - fr.l - p);
		b.seqLen = p;
		twinOf(g, s.node, s.offset, b.node, b.offset);
		b.pad = fr.read;
		ExtItem f;
		f.seqOff = readOff + fr.l + p + 1;
		f.seqLen = splitLen - 1 - p;
		f.node = s.node;
		f.offset = s.offset;
		f.pad = fr.read;
		work[2 * (size_t)slot] = b;
		work[2 * (size_t)slot + 1] = f;
		if (results) {   // lazy extension: nothing has run yet
			results[2 * (size_t)slot] = ExtResult { 0, 0, EXT_NOT_RUN, 0, 0 };
			results[2 * (size_t)slot + 1] = ExtResult { 0, 0, EXT_NOT_RUN, 0, 0 };
		}
	}
}
void launchBuildFragmentWork(hipStream_t stream, const DGraph& g, const Fragment* frags, const uint32_t* fragFirstSeed, uint32_t nFrags, const FragSeed* readSeeds, const uint64_t* readOffsets,
	uint64_t totalBases, uint32_t splitLen, FragSeed* fragSeeds, ExtItem* work, ExtResult* results)
{
	if (nFrags) hipLaunchKernelGGL(k_build_fragment_work, dim3((nFrags + 255) / 256), dim3(256), 0, stream, g, frags, fragFirstSeed, nFrags, readSeeds, readOffsets, totalBases, splitLen, fragSeeds, work, results);
}
#ifdef GC_EXPERIMENTS   // (only the state-machine experiment lists its declined items this way; the product's extension kernel writes its own retry list)
// work items of a round whose extension ended with `status` (EXT_LDS_CAP: the band outgrew the register tables), as a list for the retry launch:
// almost always empty, so the retry costs one small kernel and a handful of waves instead of one wave per pair of work items
__global__ void __launch_bounds__(256) k_long_retry_list(const LongWorkResult* __restrict__ results, uint32_t nWork, uint32_t status, uint32_t* __restrict__ list, unsigned long long* __restrict__ listCount)
{
	const uint32_t w = blockIdx.x * 256 + threadIdx.x;
	if (w < nWork && results[w].status == status) list[atomicAdd(listCount, 1ull)] = w;
}
void launchLongRetryList(hipStream_t stream, const LongWorkResult* results, uint32_t nWork, uint32_t status, uint32_t* list, unsigned long long* listCount)
{
	if (nWork) hipLaunchKernelGGL(k_long_retry_list, dim3((nWork + 255) / 256), dim3(256), 0, stream, results, nWork, status, list, listCount);
}
#endif
void launchLongMerge(hipStream_t stream, const DGraph& g, const LongJob* jobs, uint32_t nReads, const LongSeed* seeds, const uint32_t* candSeed, const LongWorkResult* results,
	const unsigned long long* tracePool, uint32_t maxAlignments, LongState* state, LongAln* alns, LongCell* cellPool, unsigned long long* cellCursor, uint64_t cellCapacity)
{
	if (nReads) hipLaunchKernelGGL(k_long_merge, dim3(nReads), dim3(64), 0, stream, g, jobs, nReads, seeds, candSeed, results, tracePool, maxAlignments, state, alns, cellPool, cellCursor, cellCapacity);
}
#ifdef GC_EXPERIMENTS
void launchLongRound(hipStream_t stream, const DGraph& g, const LongJob* jobs, uint32_t nReads, const LongSeed* seeds, uint32_t minClusterSize, uint32_t round, uint32_t forceCand, uint32_t gridLimit,
	LongState* state, LongAln* alns, LongCell* cellPool, unsigned long long* cellCursor, uint64_t cellCapacity, uint32_t maxAlignments, LongWork* work, uint32_t* workLen, uint32_t* candSeed,
	const LongWorkResult* results, const unsigned long long* tracePool, unsigned long long* cursorSets, unsigned long long* roundInfo, unsigned long long* ticket, uint32_t* order, uint32_t maxLen, uint32_t orderMode,
	unsigned long long* hostInfo, uint64_t workCapacity)
{
	if (!nReads) return;
	uint32_t shift = 0;
	while ((maxLen >> shift) > 1023) shift++;
	hipLaunchKernelGGL(k_long_round, dim3(nReads), dim3(64), 0, stream, g, jobs, nReads, seeds, minClusterSize, round, forceCand, gridLimit, state, alns, cellPool, cellCursor, cellCapacity, maxAlignments,
		work, workLen, candSeed, results, tracePool, cursorSets, roundInfo, ticket, order, shift, orderMode, (volatile unsigned long long*)hostInfo, workCapacity);
}
#endif
void launchLongOrder(hipStream_t stream, const uint32_t* workLen, const unsigned long long* workCount, uint32_t* order, uint32_t maxLen, uint32_t mode)
{
	uint32_t shift = 0;
	while ((maxLen >> shift) > 1023) shift++;
	hipLaunchKernelGGL(k_long_order, dim3(1), dim3(GC_ORDER_THREADS), 0, stream, workLen, workCount, order, shift, mode);
}
void launchPublish(hipStream_t stream, const unsigned long long* src, unsigned long long* dst, uint32_t nWords)
{
	hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, stream, src, dst, nWords);
}
void launchZeroWords(hipStream_t stream, unsigned long long* dst, uint32_t nWords)
{
	hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, stream, dst, nWords);
}
void launchLongFinish(hipStream_t stream, uint32_t nReads, const LongState* state, LongReadResult* results)
{
	if (nReads) hipLaunchKernelGGL(k_long_finish, dim3((nReads + 255) / 256), dim3(256), 0, stream, nReads, state, results);
}

uint64_t longSlabBytes(const ExtendConfig& cfg) { return (extendSlabBytes(cfg) + sizeof(TraceCell) * (uint64_t)cfg.maxTrace + 63) & ~63ull; }

void launchLongPass(hipStream_t stream, const DGraph& g, const CorrectnessTables* ct, const uint8_t* iupac, const ExtendConfig& cfg, const LongJob* jobs, uint32_t nReads,
	const LongSeed* seeds, const char* bases, uint64_t rcBase, uint32_t minClusterSize, uint32_t maxAlignments, uint8_t* scratch, uint64_t slabBytes,
	LongCell* cellPool, unsigned long long* cellCursor, uint64_t cellCapacity, LongAln* alns, LongReadResult* results, unsigned long long* counters)
{
	if (nReads == 0) return;
	uint32_t blocks = (nReads + 63) / 64;
	hipLaunchKernelGGL(k_long_pass, dim3(blocks), dim3(64), 0, stream, g, ct, iupac, cfg, jobs, nReads, seeds, bases, rcBase, minClusterSize, maxAlignments, scratch, slabBytes,
		cellPool, cellCursor, cellCapacity, alns, results, counters);
}

} // namespace gcdev
