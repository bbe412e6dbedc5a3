// k_seed_glue: everything between the seed-lookup kernel and the extension kernels, on the device, one wave per read:
//   addMinimizers (src/MinimizerSeeder.cpp:494-520: the read's index hits sorted by occurrence count, expanded into seeds until the density
//   cut-off), orderSeedsByChaining (src/GraphAligner.h:233-295: seeds clustered by chain and diagonal, a cluster's matching base pairs,
//   seeds by goodness, best first), the sort by read position and the fragment windows (src/Aligner.cpp:667-679).
// The reference feeds three UNSTABLE std::sort calls into order-sensitive logic there (matches by count, seeds by goodness, seeds by
// seqPos): their tie order defines the seed order of the whole-read pass, the anchor indices and the chaining tie-breaks. r1 / r2 therefore
// kept this part on the host with the same libstdc++ (1.0 CPU-second per 10 k reads, 50-80 ms on a batch's critical path, 36 MB of matches
// down and 40 MB of seeds up per batch). Here lane 0 of the read's wave runs libstdc++'s own algorithm (gc_stdsort.hpp: introsort with its
// exact pivot, partition, depth-limit and insertion-sort rules, checked permutation for permutation against the local libstdc++), so the
// orders are the host's; the data-parallel parts (count lookup, hit expansion, diagonals, record emission) use all 64 lanes.
// The sort's working array (16 B elements) lives in LDS for reads of up to GLUE_LDS_ELEMS seeds, in HBM beyond.
#include "gc_kernels.hpp"
#include "gc_stdsort.hpp"
#include "gc_stdsort_wave.hpp"
#include <climits>

namespace gcdev {


struct GlueElem { uint32_t k0, k1lo, k1hi, id; };

struct ByK0 { __host__ __device__ __forceinline__ bool operator()(const GlueElem& l, const GlueElem& r) const { return l.k0 < r.k0; } };
struct ByChainDiagonal {
	__host__ __device__ __forceinline__ bool operator()(const GlueElem& l, const GlueElem& r) const
	{
		if (l.k0 != r.k0) return l.k0 < r.k0;
		const uint64_t dl = (uint64_t)l.k1lo | ((uint64_t)l.k1hi << 32), dr = (uint64_t)r.k1lo | ((uint64_t)r.k1hi << 32);
		return dl < dr;
	}
};
struct U32Less { __host__ __device__ __forceinline__ bool operator()(uint32_t l, uint32_t r) const { return l < r; } };
struct HiLess { __host__ __device__ __forceinline__ bool operator()(uint64_t l, uint64_t r) const { return (uint32_t)(l >> 32) < (uint32_t)(r >> 32); } };   // (key << 32 | id) elements, compared by key alone

typedef __attribute__((address_space(3))) uint32_t glue_lds_u32;
typedef __attribute__((address_space(3))) uint64_t glue_lds_u64;

// what a read's pass leaves for the emission kernel (k_glue_emit) and for the batch's totals
struct GlueCounts { uint32_t *nSeeds, *nFrags, *nSlots, *failed; };

// Any-order sorts (the clusters' (chain, diagonal) order and the positions inside a cluster: only the sorted VALUES matter) run on all 64 lanes:
// bitonic sort of the structure-of-arrays (k0, k2:k1, id) in LDS, m = the power of two >= n (the tail is padded with maximal keys by the caller).
__device__ __forceinline__ void glueBitonic(glue_lds_u32* k0, glue_lds_u32* k1, glue_lds_u32* k2, glue_lds_u32* id, uint32_t m, uint32_t lane)
{
	for (uint32_t k = 2; k <= m; k <<= 1) {
		for (uint32_t j = k >> 1; j > 0; j >>= 1) {
			for (uint32_t i = lane; i < m; i += 64) {
				const uint32_t x = i ^ j;
				if (x > i) {
					const uint32_t a0 = k0[i], a1 = k1[i], a2 = k2[i], b0 = k0[x], b1 = k1[x], b2 = k2[x];
					const bool aGreater = a0 != b0 ? a0 > b0 : (a2 != b2 ? a2 > b2 : a1 > b1);
					const bool ascending = (i & k) == 0;
					if (aGreater == ascending && !(a0 == b0 && a1 == b1 && a2 == b2)) {
						const uint32_t ai = id[i], bi = id[x];
						k0[i] = b0; k1[i] = b1; k2[i] = b2; id[i] = bi;
						k0[x] = a0; k1[x] = a1; k2[x] = a2; id[x] = ai;
					}
				}
			}
			__syncthreads();
		}
	}
}

// fragment windows (src/Aligner.cpp:672-679): two pointers over the position-sorted seeds; key(i) = read position of seed i in that order.
// Leaves (l, sl, sr, first slot) per window in `win`, and the counts the batch needs.
template <class Key>
__device__ __forceinline__ void glueWindows(Key key, uint32_t nS, uint32_t len, uint32_t splitLen, uint32_t splitGap, uint32_t matchLen, uint32_t* win, uint32_t& nFout, uint32_t& slotsOut, unsigned long long& budgetOut, uint32_t& widestOut)
{
	uint32_t sl = 0, sr = 0, nF = 0, slots = 0, widest = 0;
	unsigned long long budget = 0;
	for (uint64_t l = 0; l + splitLen <= len; l += splitGap) {
		while (sr < nS && (uint64_t)key(sr) + matchLen <= l + splitLen) sr++;
		while (sl < sr && key(sl) < l) sl++;
		if (sl >= sr) continue;
		win[4 * nF] = (uint32_t)l; win[4 * nF + 1] = sl; win[4 * nF + 2] = sr; win[4 * nF + 3] = slots;
		for (uint32_t k = sl; k < sr; k++) {
			// trace cells the two extensions of this seed may need: backward p rows, forward split_len - 1 - p (src/GraphAligner.h:499-511)
			const uint32_t p = key(k) - (uint32_t)l, q = splitLen - 1 - p;
			budget += (p ? p + 24 : 0) + (q ? q + 24 : 0);
		}
		widest = widest > sr - sl ? widest : sr - sl;
		slots += sr - sl;
		nF++;
	}
	nFout = nF; slotsOut = slots; budgetOut = budget; widestOut = widest;
}

// The same windows by all lanes (r5). The two pointers of src/Aligner.cpp:672-679 do not depend on the window before: sr(l) = the seeds with key + matchLen <= l + splitLen - an upper
// bound in the sorted keys - and sl(l) = min(sr(l), first seed with key >= l): the pointer sl only ever waits at sr, and what it waits for is monotone in l. So a lane takes a
// fragment position, two binary searches give its window, and ballots / wave scans number the non-empty windows and their anchor slots in position order.
template <class Key>
__device__ __forceinline__ void glueWindowsWave(Key key, uint32_t nS, uint32_t len, uint32_t splitLen, uint32_t splitGap, uint32_t matchLen, uint32_t* win, uint32_t lane,
	uint32_t& nFout, uint32_t& slotsOut, unsigned long long& budgetOut, uint32_t& widestOut)
{
	uint32_t nF = 0, slots = 0, widest = 0;
	unsigned long long budget = 0;
	const uint64_t nPos = len >= splitLen ? (uint64_t)(len - splitLen) / splitGap + 1 : 0;
	for (uint64_t f0 = 0; f0 < nPos; f0 += 64) {
		const uint64_t f = f0 + lane;
		const uint64_t l = f * splitGap;
		uint32_t sl = 0, sr = 0;
		if (f < nPos) {
			uint32_t lo = 0, hi = nS;   // first seed with key + matchLen > l + splitLen
			while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((uint64_t)key(mid) + matchLen <= l + splitLen) lo = mid + 1; else hi = mid; }
			sr = lo;
			lo = 0; hi = sr;            // first seed with key >= l (not beyond sr)
			while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((uint64_t)key(mid) < l) lo = mid + 1; else hi = mid; }
			sl = lo;
		}
		const bool has = sl < sr;
		const unsigned long long which = __ballot(has);
		const uint32_t mine = sr - sl;
		uint32_t incl = has ? mine : 0u;
		for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if ((int)lane >= d) incl += o; }
		unsigned long long b = 0;
		if (has) {
			const uint32_t at = nF + (uint32_t)__popcll(which & ((1ull << lane) - 1ull));
			win[4 * at] = (uint32_t)l; win[4 * at + 1] = sl; win[4 * at + 2] = sr; win[4 * at + 3] = slots + incl - mine;
			for (uint32_t k = sl; k < sr; k++) {   // trace cells the two extensions of this seed may need: backward p rows, forward split_len - 1 - p (src/GraphAligner.h:499-511)
				const uint32_t p = key(k) - (uint32_t)l, q = splitLen - 1 - p;
				b += (p ? p + 24 : 0) + (q ? q + 24 : 0);
			}
		}
		uint32_t w = has ? mine : 0u;
		for (int d = 32; d > 0; d >>= 1) { b += __shfl_xor(b, d); const uint32_t o = __shfl_xor(w, d); w = o > w ? o : w; }
		budget += b;
		widest = w > widest ? w : widest;
		slots += __shfl(incl, 63);
		nF += (uint32_t)__popcll(which);
	}
	nFout = nF; slotsOut = slots; budgetOut = budget; widestOut = widest;
}

// addMinimizers' cut-off over the hits sorted by occurrence count (src/MinimizerSeeder.cpp:501-519): hit k is expanded unless seedsHere >= maxHits and its count is larger than
// the previous hit's (`allowed`) - i.e. the list ends at the first k whose exclusive prefix sum of counts has reached maxHits and whose count exceeds its predecessor's (hit 0
// exceeds the initial 0). All lanes: count(k) reads hit k's count, place(k, sum) notes where its seeds start; returns the kept hits and the seeds they expand to.
template <class Count, class Place>
__device__ __forceinline__ void glueDensityCut(uint32_t nM, uint64_t maxHits, uint32_t lane, Count count, Place place, uint32_t& keptOut, uint64_t& seedsOut)
{
	uint64_t running = 0;
	uint32_t kept = nM, prevTail = 0;
	for (uint32_t base = 0; base < nM; base += 64) {
		const uint32_t k = base + lane;
		const uint32_t cnt = k < nM ? count(k) : 0u;
		uint64_t incl = cnt;
		for (int d = 1; d < 64; d <<= 1) { const uint64_t o = __shfl_up(incl, d); if ((int)lane >= d) incl += o; }
		const uint64_t before = running + incl - cnt;
		uint32_t prev = __shfl_up(cnt, 1);
		if (lane == 0) prev = prevTail;
		const bool stop = k < nM && before >= maxHits && cnt > prev;
		const unsigned long long stops = __ballot(stop);
		const uint32_t firstStop = stops ? (uint32_t)__ffsll((unsigned long long)stops) - 1u : 64u;
		if (k < nM && lane < firstStop) place(k, before);
		if (stops) { kept = base + firstStop; running = __shfl(before, (int)firstStop); break; }
		running += __shfl(incl, 63);
		prevTail = __shfl(cnt, 63);
	}
	keptOut = kept; seedsOut = running;
}

// Two instantiations: GLUE_LDS_ELEMS = 1024 (24 KB of LDS, six blocks per CU) takes the reads whose seed bound fits it - every 10 kb read; the other one
// (no LDS image, so as many blocks per CU as wave slots) takes the rest on HBM arrays (50 kb reads carry ~2 200 seeds). The serial sorts are bound by
// the latency of a lane's dependent accesses, so what counts for the long reads is how many run at once: a 4096-element LDS image (96 KB, one block
// per CU) measured 820 ms per 2 000 x 50 kb reads against 231 ms for the HBM arrays at six blocks per CU.
template <uint32_t GLUE_LDS_ELEMS, uint32_t SKIP_UP_TO>
__global__ void __launch_bounds__(64) k_seed_glue(SeedIndex idx, DGraph g, const uint64_t* __restrict__ readOff, uint32_t nReads, const uint8_t* __restrict__ invalidRead,
	const uint2* __restrict__ matches, const uint32_t* __restrict__ readMatchOff, const uint32_t* __restrict__ readMatchCount, const uint32_t* __restrict__ readSeedOff, const uint32_t* __restrict__ winCapOff,
	double density, uint32_t splitLen, uint32_t splitGap, uint32_t longPass, GlueStaging st,
	LongSeed* __restrict__ longSeeds, FragSeed* __restrict__ readSeeds, GlueCounts counts, unsigned long long* __restrict__ cursors)
{
	GC_RAISE_PRIO();
	// one LDS image, reused along the read's pass: [0] k0 (offsets of the kept hits / chain / cluster id), [1] k1, [2] k2 (diagonal; the 8-byte
	// elements of the three order-critical sorts alias these two), [3] id, [4] cluster sums (first: cluster ids in sorted order), [5] cluster sizes
	__shared__ uint32_t lds[6 * GLUE_LDS_ELEMS];
	__shared__ uint32_t shared[8];
	glue_lds_u32* const eK0 = (glue_lds_u32*)&lds[0];
	glue_lds_u32* const eK1 = (glue_lds_u32*)&lds[GLUE_LDS_ELEMS];
	glue_lds_u32* const eK2 = (glue_lds_u32*)&lds[2 * GLUE_LDS_ELEMS];
	glue_lds_u32* const eId = (glue_lds_u32*)&lds[3 * GLUE_LDS_ELEMS];
	glue_lds_u32* const sSum = (glue_lds_u32*)&lds[4 * GLUE_LDS_ELEMS];
	glue_lds_u32* const sCnt = (glue_lds_u32*)&lds[5 * GLUE_LDS_ELEMS];
	glue_lds_u64* const s64 = (glue_lds_u64*)&lds[GLUE_LDS_ELEMS];
	// scratch of the wave sort for the reads of the LDS image: [3] id, [4], [5] are dead whenever one of the three order-critical sorts runs (3 x GLUE_LDS_ELEMS words >= waveSortScratchWords)
	uint32_t* const sortScratch = &lds[GLUE_LDS_ELEMS > 1 ? 3 * GLUE_LDS_ELEMS : 0];
	static_assert(GLUE_LDS_ELEMS == 1 || 3 * GLUE_LDS_ELEMS >= 2 * GLUE_LDS_ELEMS + 12 * (GLUE_LDS_ELEMS / 17 + 4) + 8, "the wave sort's scratch must fit the three idle arrays");
	const uint32_t lane = threadIdx.x;
	const uint32_t matchLen = (uint32_t)idx.k;
	for (uint32_t r = blockIdx.x; r < nReads; r += gridDim.x) {
		__syncthreads();
		const uint32_t len = (uint32_t)(readOff[r + 1] - readOff[r]);
		const uint32_t nM = invalidRead[r] ? 0u : readMatchCount[r];
		const uint32_t mOff = readMatchOff[r], sOff = readSeedOff[r];
		const uint32_t cap = readSeedOff[r + 1] - sOff;   // bound of the read's seed list: the sum of its hits' occurrence counts
		if (SKIP_UP_TO ? cap <= SKIP_UP_TO : cap > GLUE_LDS_ELEMS) continue;   // (the other instantiation's read)
		uint32_t* const win = st.winBuf + 4ull * winCapOff[r];
		uint32_t nS = 0, nF = 0, slots = 0, widest = 0;
		unsigned long long budget = 0;
		bool failed = invalidRead[r] != 0;
		uint64_t maxHits = (uint64_t)((double)len * density);
		if (density == -1) maxHits = ~0ull;
		if (nM > 0 && cap <= GLUE_LDS_ELEMS) {
			// ================= the usual case: everything the serial parts touch is in LDS
			// ---- addMinimizers: count and first occurrence of every hit (matches come in read order)
			for (uint32_t i = lane; i < nM; i += 64) {
				const uint2 m = matches[mOff + i];
				const uint64_t start = idx.startPos[m.y];
				const uint32_t cnt = (uint32_t)(idx.startPos[m.y + 1] - start);
				s64[i] = ((uint64_t)cnt << 32) | i;
				st.mPos[sOff + i] = m.x; st.mStartLo[sOff + i] = (uint32_t)start; st.mStartHi[sOff + i] = (uint32_t)(start >> 32);
			}
			__threadfence();
			__syncthreads();
			// "prefer less common minimizers": the reference's unstable sort by count (:497), replayed by the whole wave (gc_stdsort_wave.hpp; r3 / r4: lane 0 alone)
			gcsort::gcStdSortWave<uint64_t>(s64, nM, HiLess(), sortScratch, lane);
			uint32_t kept = 0;
			{
				uint64_t seedsHere = 0;
				glueDensityCut(nM, maxHits, lane, [&](uint32_t k) { return (uint32_t)(s64[k] >> 32); }, [&](uint32_t k, uint64_t sum) { eK0[k] = (uint32_t)sum; }, kept, seedsHere);
				nS = (uint32_t)seedsHere;
			}
			__syncthreads();
			// ---- hit expansion (:509-516, matchToSeedHit :546-555)
			for (uint32_t k = lane; k < kept; k += 64) {
				const uint64_t e = s64[k];
				const uint32_t mi = (uint32_t)e, cnt = (uint32_t)(e >> 32), first = eK0[k];
				const uint64_t start = (uint64_t)st.mStartLo[sOff + mi] | ((uint64_t)st.mStartHi[sOff + mi] << 32);
				const uint32_t pos = st.mPos[sOff + mi];
				for (uint32_t j = 0; j < cnt; j++) {
					const uint64_t p = idx.positions[start + j];
					const uint32_t s = sOff + first + j;
					st.sSeqPos[s] = pos; st.sNode[s] = (uint32_t)(p >> 6); st.sOffset[s] = (uint32_t)(p & 63); st.sGood[s] = idx.maxCount - cnt;   // rawSeedGoodness
				}
			}
			__threadfence();
			__syncthreads();
			if (nS > 0) {
				// ---- orderSeedsByChaining: diagonal of every seed on its chain (:245-262)
				uint32_t m = 2;
				while (m < nS) m <<= 1;
				uint32_t bad = 0;
				for (uint32_t s = lane; s < m; s += 64) {
					if (s < nS) {
						const uint32_t node = st.sNode[sOff + s], pos = st.sSeqPos[sOff + s];
						const uint64_t base = g.chainApproxPos[node] + st.sOffset[sOff + s];
						if (base < pos) bad = 1;   // the reference asserts (:259)
						const uint64_t diagonal = base - pos;
						eK0[s] = g.chainNumber[node]; eK1[s] = (uint32_t)diagonal; eK2[s] = (uint32_t)(diagonal >> 32); eId[s] = s;
					} else { eK0[s] = 0xffffffffu; eK1[s] = 0xffffffffu; eK2[s] = 0xffffffffu; eId[s] = 0xffffffffu; }
				}
				failed = __any(bad != 0);
				__syncthreads();
				if (!failed) {
					// clusters: same chain, neighbouring diagonals at most 100 apart (:263-292). The reference builds them with a hash map and two unstable
					// sorts; a cluster's goodness depends only on the multiset of its seeds' read positions, so any sort by (chain, diagonal) gives the same values
					glueBitonic(eK0, eK1, eK2, eId, m, lane);
					if (lane == 0) {
						uint32_t cid = 0, prevChain = eK0[0];
						uint64_t prevDiag = (uint64_t)eK1[0] | ((uint64_t)eK2[0] << 32);
						sSum[0] = 0;
						for (uint32_t i = 1; i < nS; i++) {
							const uint32_t chain = eK0[i];
							const uint64_t diag = (uint64_t)eK1[i] | ((uint64_t)eK2[i] << 32);
							if (!(chain == prevChain && diag <= prevDiag + 100)) cid++;
							sSum[i] = cid;
							prevChain = chain; prevDiag = diag;
						}
						shared[2] = cid + 1;
					}
					__syncthreads();
					const uint32_t nClusters = shared[2];
					for (uint32_t i = lane; i < nS; i += 64) { eK0[i] = sSum[i]; eK1[i] = st.sSeqPos[sOff + eId[i]]; eK2[i] = 0; }
					__syncthreads();
					// positions inside every cluster in ascending order, then a cluster's matching base pairs: a seed adds the part of its k-mer that
					// the previous one (by position) does not cover (:270-284)
					glueBitonic(eK0, eK1, eK2, eId, m, lane);
					for (uint32_t c = lane; c < nClusters; c += 64) { sSum[c] = 0; sCnt[c] = 0; }
					__syncthreads();
					for (uint32_t i = lane; i < nS; i += 64) {
						const uint32_t c = eK0[i];
						uint32_t add = matchLen - 1;
						if (i > 0 && eK0[i - 1] == c) { const uint32_t d = eK1[i] - eK1[i - 1]; add = d < add ? d : add; }
						atomicAdd((uint32_t*)&lds[4 * GLUE_LDS_ELEMS + c], add);
						atomicAdd((uint32_t*)&lds[5 * GLUE_LDS_ELEMS + c], 1u);
					}
					__syncthreads();
					// goodness and cluster size of every seed (staging, read back below: the order-critical sort runs on the seeds in expansion order)
					for (uint32_t i = lane; i < nS; i += 64) {
						const uint32_t c = eK0[i], s = sOff + eId[i];
						const uint32_t size = sCnt[c];
						st.sGood[s] = sSum[c] + st.sGood[s];
						st.sCluster[s] = size < 65535 ? size : 65535;
					}
					__threadfence();
					__syncthreads();
					for (uint32_t s = lane; s < nS; s += 64) s64[s] = ((uint64_t)st.sGood[sOff + s] << 32) | s;
					__syncthreads();
					// ---- seeds by goodness, best first (:293-294): the unstable sort runs on the seeds in expansion order, then the list is reversed
					gcsort::gcStdSortWave<uint64_t>(s64, nS, HiLess(), sortScratch, lane);
					for (uint32_t i = lane; i < nS / 2; i += 64) { const uint32_t j = nS - 1 - i; const uint64_t t = s64[i]; s64[i] = s64[j]; s64[j] = t; }
					__syncthreads();
					// the whole-read pass's seed list, and the next sort's keys (it runs on this order, src/Aligner.cpp:667)
					for (uint32_t i = lane; i < nS; i += 64) {
						const uint32_t id = (uint32_t)s64[i], s = sOff + id;
						const uint32_t pos = st.sSeqPos[s];
						if (longPass) {
							LongSeed ls;
							ls.node = st.sNode[s]; ls.seqPos = pos; ls.goodness = st.sGood[s]; ls.clusterSize = (uint16_t)st.sCluster[s]; ls.offset = (uint8_t)st.sOffset[s]; ls.pad = 0;
							longSeeds[sOff + i] = ls;
						}
						s64[i] = ((uint64_t)pos << 32) | id;
					}
					__syncthreads();
					gcsort::gcStdSortWave<uint64_t>(s64, nS, HiLess(), sortScratch, lane);   // seeds by read position, the reference's unstable sort (src/Aligner.cpp:667)
					for (uint32_t i = lane; i < nS; i += 64) {
						const uint32_t s = sOff + (uint32_t)s64[i];
						readSeeds[sOff + i] = FragSeed { st.sNode[s], st.sOffset[s], st.sSeqPos[s], st.sGood[s] };   // (pad carries the goodness: the seed_* result arrays)
					}
					glueWindowsWave([&](uint32_t i) { return (uint32_t)(s64[i] >> 32); }, nS, len, splitLen, splitGap, matchLen, win, lane, nF, slots, budget, widest);
				}
			}
		} else if (nM > 0) {
			// ================= a read with more seed occurrences than the LDS image holds: the same steps on HBM arrays. r5: the sorts run on the whole wave
			// (gc_stdsort_wave.hpp), the density cut-off and the clusters are wave scans; r3 / r4 ran all of it on lane 0 (650 ms for the 18 000 seed occurrences of a 50 kb
			// read on a 960 Mbp graph)
			GlueElem* a = st.sortBuf + sOff;
			uint32_t* const scratch = st.sortScratch + 3ull * sOff + 64ull * r;   // waveSortScratchWords(cap) <= 3 cap + 64; between the sorts: cluster sums and sizes
			for (uint32_t i = lane; i < nM; i += 64) {
				const uint2 m = matches[mOff + i];
				const uint64_t start = idx.startPos[m.y];
				const uint32_t cnt = (uint32_t)(idx.startPos[m.y + 1] - start);
				a[i] = GlueElem { cnt, 0, 0, i };
				st.mPos[sOff + i] = m.x; st.mStartLo[sOff + i] = (uint32_t)start; st.mStartHi[sOff + i] = (uint32_t)(start >> 32);
			}
			__threadfence_block();
			__syncthreads();
			gcsort::gcStdSortWave<GlueElem>(a, nM, ByK0(), scratch, lane);
			uint32_t kept = 0;
			{
				uint64_t seedsHere = 0;
				glueDensityCut(nM, maxHits, lane, [&](uint32_t k) { return a[k].k0; }, [&](uint32_t k, uint64_t sum) { a[k].k1lo = (uint32_t)sum; }, kept, seedsHere);
				nS = (uint32_t)seedsHere;
			}
			__threadfence_block();
			__syncthreads();
			for (uint32_t k = lane; k < kept; k += 64) {
				const GlueElem e = a[k];
				const uint32_t mi = e.id, cnt = e.k0;
				const uint64_t start = (uint64_t)st.mStartLo[sOff + mi] | ((uint64_t)st.mStartHi[sOff + mi] << 32);
				const uint32_t pos = st.mPos[sOff + mi];
				for (uint32_t j = 0; j < cnt; j++) {
					const uint64_t p = idx.positions[start + j];
					const uint32_t s = sOff + e.k1lo + j;
					st.sSeqPos[s] = pos; st.sNode[s] = (uint32_t)(p >> 6); st.sOffset[s] = (uint32_t)(p & 63); st.sGood[s] = idx.maxCount - cnt;
				}
			}
			__threadfence_block();
			__syncthreads();
			if (nS > 0) {
				uint32_t bad = 0;
				for (uint32_t s = lane; s < nS; s += 64) {
					const uint32_t node = st.sNode[sOff + s], pos = st.sSeqPos[sOff + s];
					const uint64_t base = g.chainApproxPos[node] + st.sOffset[sOff + s];
					if (base < pos) bad = 1;
					const uint64_t diagonal = base - pos;
					a[s] = GlueElem { g.chainNumber[node], (uint32_t)diagonal, (uint32_t)(diagonal >> 32), s };
				}
				failed = __any(bad != 0);
				__threadfence_block();
				__syncthreads();
				if (!failed) {
					// clusters (:263-292): same chain, neighbouring diagonals at most 100 apart. A cluster's goodness depends only on the multiset of its seeds' read positions, so
					// any sort by (chain, diagonal) and any sort by (cluster, position) give the reference's values (the LDS path uses bitonic sorts for the same reason)
					gcsort::gcStdSortWave<GlueElem>(a, nS, ByChainDiagonal(), scratch, lane);
					uint32_t cidBefore = 0, tailChain = 0;
					uint64_t tailDiag = 0;
					for (uint32_t base = 0; base < nS; base += 64) {
						const uint32_t i = base + lane;
						GlueElem x = GlueElem { 0, 0, 0, 0 };
						if (i < nS) x = a[i];
						const uint64_t dx = (uint64_t)x.k1lo | ((uint64_t)x.k1hi << 32);
						uint32_t pChain = __shfl_up(x.k0, 1);
						uint64_t pDiag = __shfl_up(dx, 1);
						if (lane == 0) { pChain = tailChain; pDiag = tailDiag; }
						const bool opens = i < nS && i > 0 && !(x.k0 == pChain && dx <= pDiag + 100);
						const unsigned long long opened = __ballot(opens);
						const uint32_t cid = cidBefore + (uint32_t)__popcll(opened & ((2ull << lane) - 1ull));
						tailChain = __shfl(x.k0, 63); tailDiag = __shfl(dx, 63);   // (before the elements are overwritten: the next 64 compare with this one's last)
						if (i < nS) a[i] = GlueElem { cid, st.sSeqPos[sOff + x.id], 0, x.id };
						cidBefore += (uint32_t)__popcll(opened);
					}
					const uint32_t nClusters = cidBefore + 1;
					__threadfence_block();
					__syncthreads();
					gcsort::gcStdSortWave<GlueElem>(a, nS, ByChainDiagonal(), scratch, lane);   // by (cluster, read position)
					uint32_t* const cSum = scratch;
					uint32_t* const cCnt = scratch + nS;
					for (uint32_t c = lane; c < nClusters; c += 64) { cSum[c] = 0; cCnt[c] = 0; }
					__threadfence_block();
					__syncthreads();
					// a cluster's matching base pairs: a seed adds the part of its k-mer that the previous one (by position) does not cover (:270-284)
					for (uint32_t i = lane; i < nS; i += 64) {
						const GlueElem x = a[i];
						uint32_t add = matchLen - 1;
						if (i > 0) { const GlueElem y = a[i - 1]; if (y.k0 == x.k0) { const uint32_t d = x.k1lo - y.k1lo; add = d < add ? d : add; } }
						atomicAdd(&cSum[x.k0], add);
						atomicAdd(&cCnt[x.k0], 1u);
					}
					__threadfence_block();
					__syncthreads();
					for (uint32_t i = lane; i < nS; i += 64) {
						const GlueElem x = a[i];
						const uint32_t s = sOff + x.id, size = cCnt[x.k0];
						st.sGood[s] = cSum[x.k0] + st.sGood[s];
						st.sCluster[s] = size < 65535 ? size : 65535;
					}
					__threadfence_block();
					__syncthreads();
					for (uint32_t s = lane; s < nS; s += 64) a[s] = GlueElem { st.sGood[sOff + s], 0, 0, s };
					__threadfence_block();
					__syncthreads();
					// ---- seeds by goodness, best first (:293-294): the unstable sort on the seeds in expansion order, then the list reversed
					gcsort::gcStdSortWave<GlueElem>(a, nS, ByK0(), scratch, lane);
					for (uint32_t i = lane; i < nS / 2; i += 64) { const uint32_t j = nS - 1 - i; const GlueElem t = a[i]; a[i] = a[j]; a[j] = t; }
					__threadfence_block();
					__syncthreads();
					for (uint32_t i = lane; i < nS; i += 64) {
						const uint32_t s = sOff + a[i].id;
						if (longPass) {
							LongSeed ls;
							ls.node = st.sNode[s]; ls.seqPos = st.sSeqPos[s]; ls.goodness = st.sGood[s]; ls.clusterSize = (uint16_t)st.sCluster[s]; ls.offset = (uint8_t)st.sOffset[s]; ls.pad = 0;
							longSeeds[sOff + i] = ls;
						}
						a[i].k0 = st.sSeqPos[s];
					}
					__threadfence_block();
					__syncthreads();
					gcsort::gcStdSortWave<GlueElem>(a, nS, ByK0(), scratch, lane);   // seeds by read position (src/Aligner.cpp:667)
					for (uint32_t i = lane; i < nS; i += 64) {
						const uint32_t s = sOff + a[i].id;
						readSeeds[sOff + i] = FragSeed { st.sNode[s], st.sOffset[s], st.sSeqPos[s], st.sGood[s] };
					}
					__threadfence_block();
					__syncthreads();
					glueWindowsWave([&](uint32_t i) { return a[i].k0; }, nS, len, splitLen, splitGap, matchLen, win, lane, nF, slots, budget, widest);
				}
			}
		}
		if (lane == 0) {
			if (failed) { nS = 0; nF = 0; slots = 0; budget = 0; widest = 0; }   // an invalid read, or orderSeedsByChaining's assertion: the read is dropped (seeds cleared)
			counts.nSeeds[r] = nS; counts.nFrags[r] = nF; counts.nSlots[r] = slots; counts.failed[r] = failed ? 1u : 0u;
			if (budget) atomicAdd(&cursors[2], budget);
			atomicMax(&cursors[3], (unsigned long long)slots);
			atomicMax(&cursors[4], (unsigned long long)widest);
		}
	}
}

// the batch's fragment and job records, once the exclusive scans over the reads' fragment and slot counts are known: one wave per read
__global__ void __launch_bounds__(64) k_glue_emit(const uint64_t* __restrict__ readOff, uint32_t nReads, const uint32_t* __restrict__ readSeedOff, const uint32_t* __restrict__ winCapOff, const uint32_t* __restrict__ winBuf,
	GlueCounts counts, const uint32_t* __restrict__ fragOff, const uint32_t* __restrict__ slotOff, uint32_t splitLen, uint32_t splitGap,
	Fragment* __restrict__ frags, uint32_t* __restrict__ fragFirstSeed, ReadChainJob* __restrict__ jobs, GlueRead* __restrict__ out)
{
	GC_RAISE_PRIO();
	const uint32_t lane = threadIdx.x;
	for (uint32_t r = blockIdx.x; r < nReads; r += gridDim.x) {
		const uint32_t nF = counts.nFrags[r], fragBegin = fragOff[r], slotBegin = slotOff[r], sOff = readSeedOff[r];
		const uint32_t* win = winBuf + 4ull * winCapOff[r];
		for (uint32_t f = lane; f < nF; f += 64) {
			const uint32_t l = win[4 * f], wsl = win[4 * f + 1], wsr = win[4 * f + 2], first = win[4 * f + 3];
			frags[fragBegin + f] = Fragment { r, l, slotBegin + first, slotBegin + first + (wsr - wsl) };
			fragFirstSeed[fragBegin + f] = sOff + wsl;
		}
		if (lane == 0) {
			const uint32_t len = (uint32_t)(readOff[r + 1] - readOff[r]);
			const uint32_t nKeys = len >= splitLen ? (len - splitLen) / splitGap + 1 : 1;
			out[r] = GlueRead { counts.nSeeds[r], sOff, nF, fragBegin, counts.nSlots[r], slotBegin, counts.failed[r], 0 };
			jobs[r] = ReadChainJob { slotBegin, counts.nSlots[r], slotBegin, nKeys, fragBegin, nF };
		}
	}
}

// per read: the capacity bound of its seed list (sum of its hits' occurrence counts) and, by an exclusive scan, where its staging begins
__global__ void __launch_bounds__(64) k_seed_caps(SeedIndex idx, uint32_t nReads, const uint8_t* __restrict__ invalidRead, const uint2* __restrict__ matches, const uint32_t* __restrict__ readMatchOff,
	const uint32_t* __restrict__ readMatchCount, uint32_t* __restrict__ readSeedCap)
{
	GC_RAISE_PRIO();
	const uint32_t lane = threadIdx.x;
	for (uint32_t r = blockIdx.x; r < nReads; r += gridDim.x) {
		const uint32_t nM = invalidRead[r] ? 0u : readMatchCount[r], mOff = readMatchOff[r];
		unsigned long long sum = 0;
		for (uint32_t i = lane; i < nM; i += 64) { const uint32_t key = matches[mOff + i].y; sum += idx.startPos[key + 1] - idx.startPos[key]; }
		for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
		// never less than the hit count (every hit has at least one occurrence) - the sort array of the hits lives in the same staging
		if (lane == 0) readSeedCap[r] = (uint32_t)(sum > 0xfffffffeull ? 0xfffffffeull : sum);
	}
}
__global__ void __launch_bounds__(1024) k_exclusive_scan_u32(const uint32_t* __restrict__ in0, uint32_t n, uint32_t* __restrict__ out0, unsigned long long* __restrict__ total0,
	const uint32_t* __restrict__ in1, uint32_t* __restrict__ out1, unsigned long long* __restrict__ total1)
{
	GC_RAISE_PRIO();
	const uint32_t* in = blockIdx.x ? in1 : in0;   // (two arrays in one launch: block 0 / block 1)
	uint32_t* out = blockIdx.x ? out1 : out0;
	unsigned long long* total = blockIdx.x ? total1 : total0;
	// (256 threads, r4: a 1 024-thread block needs sixteen free wave slots on ONE CU, and waited 16 ms for them among config 5's long-running whole-read waves - for a scan of 2 000 numbers)
	__shared__ unsigned long long part[1024];
	const uint32_t t = threadIdx.x, T = blockDim.x;
	const uint32_t per = (n + T - 1) / T;
	const uint32_t b = t * per, e = b + per < n ? b + per : n;
	unsigned long long s = 0;
	for (uint32_t i = b; i < e; i++) s += in[i];
	part[t] = s;
	__syncthreads();
	if (t == 0) {
		unsigned long long run = 0;
		for (uint32_t i = 0; i < T; i++) { const unsigned long long v = part[i]; part[i] = run; run += v; }
		*total = run;
		out[n] = (uint32_t)(run > 0xffffffffull ? 0xffffffffull : run);   // (the host refuses a batch whose total does not fit 32 bits)
	}
	__syncthreads();
	unsigned long long run = part[t];
	for (uint32_t i = b; i < e; i++) { out[i] = (uint32_t)(run > 0xffffffffull ? 0xffffffffull : run); run += in[i]; }
}
uint64_t glueElemBytes() { return sizeof(GlueElem); }

// test entry (gc_std_sort_permutations): array s = elems[off[s] .. off[s + 1]) of (key << 32 | index) elements sorted by key alone with the wave sort, in place; one wave per array.
// Arrays of up to 1024 elements are sorted in LDS (the seed glue's usual case), longer ones where they lie, in HBM.
__global__ void __launch_bounds__(64) k_test_std_sort(unsigned long long* __restrict__ elems, const uint64_t* __restrict__ off, uint32_t nArrays, uint32_t* __restrict__ scratch, long depthLimit)
{
	__shared__ uint64_t image[1024];
	__shared__ uint32_t ldsScratch[3 * 1024];
	const uint32_t lane = threadIdx.x;
	for (uint32_t s = blockIdx.x; s < nArrays; s += gridDim.x) {
		const uint64_t b = off[s];
		const uint32_t n = (uint32_t)(off[s + 1] - b);
		__syncthreads();
		if (n <= 1024) {
			for (uint32_t i = lane; i < n; i += 64) image[i] = elems[b + i];
			__syncthreads();
			gcsort::gcStdSortWave<uint64_t>((glue_lds_u64*)&image[0], n, HiLess(), ldsScratch, lane, depthLimit);
			for (uint32_t i = lane; i < n; i += 64) elems[b + i] = image[i];
		} else {
			gcsort::gcStdSortWave<uint64_t>((uint64_t*)(elems + b), n, HiLess(), scratch + 3ull * b + 64ull * s, lane, depthLimit);
		}
	}
}
void launchTestStdSort(hipStream_t stream, unsigned long long* elems, const uint64_t* off, uint32_t nArrays, uint32_t* scratch, long depthLimit)
{
	if (nArrays) hipLaunchKernelGGL(k_test_std_sort, dim3(nArrays < 4096 ? nArrays : 4096), dim3(64), 0, stream, elems, off, nArrays, scratch, depthLimit);
}
void launchSeedCaps(hipStream_t stream, const SeedIndex& idx, uint32_t nReads, const uint8_t* invalidRead, const uint2* matches, const uint32_t* readMatchOff, const uint32_t* readMatchCount,
	uint32_t* readSeedCap, uint32_t* readSeedOff, unsigned long long* total)
{
	if (!nReads) return;
	hipLaunchKernelGGL(k_seed_caps, dim3(nReads < 16384 ? nReads : 16384), dim3(64), 0, stream, idx, nReads, invalidRead, matches, readMatchOff, readMatchCount, readSeedCap);
	hipLaunchKernelGGL(k_exclusive_scan_u32, dim3(1), dim3(256), 0, stream, (const uint32_t*)readSeedCap, nReads, readSeedOff, total, (const uint32_t*)nullptr, (uint32_t*)nullptr, (unsigned long long*)nullptr);
}

void launchSeedGlue(hipStream_t stream, const SeedIndex& idx, const DGraph& g, const uint64_t* readOff, uint32_t nReads, const uint8_t* invalidRead, const uint2* matches, const uint32_t* readMatchOff,
	const uint32_t* readMatchCount, const uint32_t* readSeedOff, const uint32_t* winCapOff, double density, uint32_t splitLen, uint32_t splitGap, bool longPass, const GlueStaging& st,
	uint32_t* perRead /* 6 x (nReads + 1) words of scratch */, LongSeed* longSeeds, FragSeed* readSeeds, Fragment* frags, uint32_t* fragFirstSeed, ReadChainJob* jobs, GlueRead* out, unsigned long long* cursors)
{
	if (!nReads) return;
	const uint32_t blocks = nReads < 16384 ? nReads : 16384;
	const uint64_t stride = (uint64_t)nReads + 1;
	GlueCounts counts { perRead, perRead + stride, perRead + 2 * stride, perRead + 3 * stride };
	uint32_t* fragOff = perRead + 4 * stride;
	uint32_t* slotOff = perRead + 5 * stride;
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_seed_glue<1024, 0>), dim3(blocks), dim3(64), 0, stream, idx, g, readOff, nReads, invalidRead, matches, readMatchOff, readMatchCount, readSeedOff, winCapOff, density, splitLen, splitGap, longPass ? 1u : 0u, st,
		longSeeds, readSeeds, counts, cursors);
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_seed_glue<1, 1024>), dim3(blocks), dim3(64), 0, stream, idx, g, readOff, nReads, invalidRead, matches, readMatchOff, readMatchCount, readSeedOff, winCapOff, density, splitLen, splitGap, longPass ? 1u : 0u, st,
		longSeeds, readSeeds, counts, cursors);
	// where every read's fragments and anchor slots begin: exclusive scans in read order (cursors[0] = fragments, [1] = slots of the batch)
	hipLaunchKernelGGL(k_exclusive_scan_u32, dim3(2), dim3(256), 0, stream, (const uint32_t*)counts.nFrags, nReads, fragOff, cursors, (const uint32_t*)counts.nSlots, slotOff, cursors + 1);
	hipLaunchKernelGGL(k_glue_emit, dim3(blocks), dim3(64), 0, stream, readOff, nReads, readSeedOff, winCapOff, (const uint32_t*)st.winBuf, counts, (const uint32_t*)fragOff, (const uint32_t*)slotOff, splitLen, splitGap, frags, fragFirstSeed, jobs, out);
}

} // namespace gcdev
