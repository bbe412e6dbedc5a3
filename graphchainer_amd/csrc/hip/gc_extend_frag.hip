// K3 (r6) - fragment seed extension: one extension per lane, the lanes of a wave in lockstep phases (the core and the reasons: gc_frag_core.hpp).
// reference: GraphAlignerBitvectorBanded::getReverseTraceFromSeed, src/GraphAlignerBitvectorBanded.h:46-71, called per seed and direction from
// src/GraphAligner.h:499-511. What this kernel declines (EXT_OVERFLOW: more than 64 rows, an ambiguous graph node, more pending nodes / tiles than its tables hold)
// is rerun by the plain-layout kernel k_extend_slab (gc_kernels.hip) right behind it.
#include "gc_kernels.hpp"
#include "gc_frag_core.hpp"

namespace gcdev {
using namespace gcfrag;

// The schedule: a run of a handler costs its instructions whatever the number of lanes in it, a column step costs one step whatever the number of lanes still inside
// a tile. The wave stays in the column loop until enough lanes wait for one of the two handler groups: the DP's (fetch, tile end, pop, finish: GC_FRAG_DP_AT lanes) or
// the walk's (GC_FRAG_WALK_AT lanes - an extension spends a third of its handler visits there, so walkers are gathered longer). Priced on the CPU by
// tests/frag_host/frag_wave_sim.cpp (one trigger of 24 lanes for everything: 89 k vector instructions per 64 extensions; 16 / 32: 59 k).
#ifndef GC_FRAG_DP_AT
#define GC_FRAG_DP_AT 16
#endif
#ifndef GC_FRAG_WALK_AT
#define GC_FRAG_WALK_AT 32
#endif
#ifndef GC_FRAG_BURST
#define GC_FRAG_BURST 4      // column steps between two looks at the triggers
#endif
#define GC_FRAG_CLAIM 256u
#define GC_FRAG_TRACE_PIECE 1024u   // trace cells a wave takes from the pool at a time (a sweep's walkers ask for ~40 each)
// profiling build (make variant NAME=fragstamps FLAGS=-DGC_FRAG_STAMPS): wave-cycles per section of the loop below, added up into stamps[0..7] (columns, fetch, tile end, pop, finish, walk)
#ifdef GC_FRAG_STAMPS
#define GC_FRAG_MARK(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); stampCycles[i] += now_ - stampAt; stampAt = now_; } while (0)
#else
#define GC_FRAG_MARK(i) ((void)0)
#endif   // work items a wave claims at a time (neighbouring items are neighbouring fragments of a read: neighbouring graph nodes)

struct FragDevStore {
	uint32_t* lds;       // this lane's column of the wave's LDS words: word w at lds[64 w]
	uint4* items;        // this lane's column of the wave's item planes: item k = [3 k] start column, [3 k + 1] end column, [3 k + 2] { start score, node, end score, - }, each plane 64 lanes wide
	PoolCell* pool;
	__device__ __forceinline__ uint32_t ld(uint32_t w) const { return lds[64 * w]; }
	__device__ __forceinline__ void st(uint32_t w, uint32_t v) const { lds[64 * w] = v; }
	__device__ __forceinline__ void itemSetStart(uint32_t k, uint64_t VP, uint64_t VN, int32_t score, uint32_t node) const
	{
		items[64 * (3 * k)] = make_uint4((uint32_t)VP, (uint32_t)(VP >> 32), (uint32_t)VN, (uint32_t)(VN >> 32));
		*(uint2*)&items[64 * (3 * k + 2)] = make_uint2((uint32_t)score, node);
	}
	__device__ __forceinline__ void itemSetEnd(uint32_t k, uint64_t VP, uint64_t VN, int32_t score) const
	{
		items[64 * (3 * k + 1)] = make_uint4((uint32_t)VP, (uint32_t)(VP >> 32), (uint32_t)VN, (uint32_t)(VN >> 32));
		((uint32_t*)&items[64 * (3 * k + 2)])[2] = (uint32_t)score;
	}
	__device__ __forceinline__ WS itemStart(uint32_t k) const
	{
		const uint4 c = items[64 * (3 * k)];
		return WS { (uint64_t)c.x | ((uint64_t)c.y << 32), (uint64_t)c.z | ((uint64_t)c.w << 32), (int32_t)((const uint32_t*)&items[64 * (3 * k + 2)])[0] };
	}
	__device__ __forceinline__ WS itemEnd(uint32_t k) const
	{
		const uint4 c = items[64 * (3 * k + 1)];
		return WS { (uint64_t)c.x | ((uint64_t)c.y << 32), (uint64_t)c.z | ((uint64_t)c.w << 32), (int32_t)((const uint32_t*)&items[64 * (3 * k + 2)])[2] };
	}
	__device__ __forceinline__ uint32_t itemNode(uint32_t k) const { return ((const uint32_t*)&items[64 * (3 * k + 2)])[1]; }
	__device__ __forceinline__ void traceSet(uint64_t at, uint32_t node, int32_t seqPos, uint32_t offsetAndSwitch) const { pool[at] = PoolCell { node, (uint16_t)offsetAndSwitch, (int16_t)seqPos }; }
};

uint64_t extendFragScratchBytes(uint32_t waves) { return (uint64_t)waves * FRAG_I * 3 * 64 * sizeof(uint4); }

// 127 VGPRs or fewer: four waves per SIMD, which is also what the LDS words of four waves per SIMD leave room for
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) k_extend(DGraph g, const CorrectnessTables* __restrict__ ct, int32_t bandwidth,
	const ExtItem* __restrict__ work, uint32_t nWork, const FragReads reads, ExtResult* __restrict__ results, uint4* __restrict__ itemScratch,
	PoolCell* __restrict__ tracePool, unsigned long long* __restrict__ traceCursor, uint64_t traceCapacity, unsigned long long* __restrict__ counters, ExtSelection sel,
	unsigned long long* __restrict__ claim, uint32_t* __restrict__ retryList, unsigned long long* __restrict__ retryCount, unsigned long long* __restrict__ stamps)
{
	__shared__ uint32_t ldsWords[FRAG_WORDS * 64];
	__shared__ uint32_t waveCounters[8];
	const uint32_t lane = threadIdx.x;
	if (lane < 8) waveCounters[lane] = 0;
	FragParams P;
	P.bandwidth = bandwidth;
	P.keepMask = __ballot(fragSliceKept(*ct, (int)lane));
	FragMem<FragDevStore> m;
	m.lds = ldsWords + lane;
	m.items = itemScratch + (uint64_t)blockIdx.x * (FRAG_I * 3 * 64) + lane;
	m.pool = tracePool;
	Lane L;
	L.phase = PH_FETCH;
	L.work = 0xffffffffu;
	const uint32_t nSelected = sel.mode == 0 ? nWork : sel.mode == 1 ? 2 * sel.nFrags : (uint32_t)*sel.listCount;
	unsigned long long blockNext = 0, blockEnd = 0;   // the wave's claimed range of the selection (uniform)
	unsigned long long pieceNext = 0, pieceEnd = 0;   // the wave's piece of the trace pool (uniform)
	const uint64_t below = (1ull << lane) - 1;
#ifdef GC_FRAG_STAMPS
	unsigned long long stampCycles[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, stampAt = __builtin_readcyclecounter();
#endif
	__syncthreads();
	for (;;) {
		const uint32_t nIdle = (uint32_t)__popcll(__ballot(L.phase == PH_IDLE));
		if (nIdle == 64) break;
		const uint32_t nActive = 64 - nIdle;
		// ---- the column loop: until one of the handler groups has its lanes together (or nobody is inside a tile)
		bool runDp, runWalk;
		for (;;) {
			const uint32_t nCols = (uint32_t)__popcll(__ballot(L.phase == PH_COLS)), nWalk = (uint32_t)__popcll(__ballot(L.phase == PH_WALK));
			const uint32_t nDp = nActive - nCols - nWalk;
			const uint32_t dpAt = nActive - nWalk < GC_FRAG_DP_AT ? nActive - nWalk : GC_FRAG_DP_AT, walkAt = nActive < GC_FRAG_WALK_AT ? nActive : GC_FRAG_WALK_AT;
			runDp = nDp > 0 && nDp >= dpAt;
			runWalk = nWalk >= walkAt;
			if (runDp || runWalk) break;
			if (nCols == 0) { runDp = nDp >= nWalk; runWalk = !runDp; break; }
			if (L.phase == PH_COLS) fragColumns<GC_FRAG_BURST>(L, m);
		}
		GC_FRAG_MARK(0);
		// ---- one sweep of the handlers, each with every lane that is in its phase
		const uint64_t fetching = runDp ? __ballot(L.phase == PH_FETCH) : 0ull;
		if (fetching) {
			if (L.phase == PH_FETCH) {
				if (L.work != 0xffffffffu) {
					ExtResult res;
					// (a full trace pool answers EXT_OVERFLOW like anything else that did not fit, but is not handed to the plain-layout kernel: that would find the same pool full - a stream's
					// first batch on a 960 Mbp graph sent 7.5 M of 42 M extensions through its 2 048 lanes for 2.4 s before the host sized the pool again)
					const bool poolFull = L.status == EXT_POOL_FULL;
					res.status = poolFull ? (uint32_t)EXT_OVERFLOW : L.status;
					res.score = L.resultScore;
					res.traceOff = L.status == EXT_OK ? L.traceBase : 0;
					res.traceLen = L.status == EXT_OK ? L.nTrace : 0;
					res.pad = L.status == EXT_OK ? L.tie : 0;
					results[L.work] = res;
					if (poolFull) {}
					else if (L.status == EXT_OVERFLOW) { retryList[atomicAdd(retryCount, 1ull)] = L.work; atomicAdd(&waveCounters[6], 1u); }   // declined: the plain-layout kernel runs it (and counts its work)
					else {
						const bool flat = L.len < 64;
						const uint32_t dpTiles = L.cntTiles & 0xffffu, btTiles = L.cntTiles >> 16, dpCols = L.cntCols & 0xffffu, btCols = L.cntCols >> 16;
						atomicAdd(&waveCounters[0], dpTiles);
						atomicAdd(&waveCounters[1], (flat ? dpTiles : 0u) + btTiles);
						atomicAdd(&waveCounters[2], (flat ? 2u : 1u) * dpCols + btCols);
						atomicAdd(&waveCounters[3], L.status == EXT_OK ? L.nTrace : 0u);
						atomicAdd(&waveCounters[4], 1u);
						atomicAdd(&waveCounters[5], btTiles);
					}
					L.work = 0xffffffffu;
				}
			}
			// the fetching lanes take consecutive items of the wave's claimed range, and of a new one when it runs out
			const uint32_t need = (uint32_t)__popcll(fetching);
			const unsigned long long avail = blockEnd - blockNext;
			unsigned long long fresh = 0;
			if (need > avail) {
				const int first = __ffsll((long long)fetching) - 1;
				if ((int)lane == first) fresh = atomicAdd(claim, (unsigned long long)GC_FRAG_CLAIM);
				fresh = __shfl(fresh, first);
			}
			if (L.phase == PH_FETCH) {
				const uint32_t rank = (uint32_t)__popcll(fetching & below);
				const unsigned long long at = rank < avail ? blockNext + rank : fresh + (rank - avail);
				if (at >= nSelected) L.phase = PH_IDLE;
				else {
					const uint32_t w = sel.mode == 0 ? (uint32_t)at : sel.mode == 1 ? 2 * sel.frags[at >> 1].seedBegin + ((uint32_t)at & 1u) : sel.list[at];
					const ExtItem it = work[w];
					// the rows' match masks come from the read's bit vectors (built at upload): strand 1 = the reverse complement, whose bases lie behind all forward bases
					const uint32_t strand = it.seqOff >= reads.totalBases ? 1u : 0u;
					const uint32_t words = reads.maskWords[it.pad];
					EqSource src;
					src.masks = reads.masks + reads.maskOff[it.pad] + (uint64_t)strand * 4 * words;
					src.words = words;
					src.startBit = (uint32_t)(it.seqOff - (strand ? reads.totalBases : 0) - reads.readOff[it.pad]);
					fragBegin(g, P, L, m, w, it.seqLen, it.node, it.offset, src);
				}
			}
			if (need > avail) { blockNext = fresh + (need - avail); blockEnd = fresh + GC_FRAG_CLAIM; }
			else blockNext += need;
		}
		GC_FRAG_MARK(1);
		if (runDp && L.phase == PH_TILE_END) fragTileEnd(g, P, L, m);
		GC_FRAG_MARK(2);
		if (runDp && L.phase == PH_POP) fragPop(g, P, L, m);
		GC_FRAG_MARK(3);
		if (runDp) {
			bool walk = false;
			if (L.phase == PH_FINISH) walk = fragFinish(P, L);
			const uint64_t walking = __ballot(walk);
			if (walking) {
				// trace cells for the lanes that start their walk in this sweep, out of the wave's own piece of the pool (a new piece - one atomic on the pool's cursor,
				// a round trip to memory with every lane waiting - only when the piece runs out; what is left of the old one stays unused)
				uint32_t incl = walk ? L.traceCap : 0u;
				for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t up = __shfl_up(incl, d); if (lane >= d) incl += up; }
				const uint32_t total = __shfl(incl, 63);
				if (pieceEnd - pieceNext < total) {
					const unsigned long long take = total > GC_FRAG_TRACE_PIECE ? total : GC_FRAG_TRACE_PIECE;
					unsigned long long got = 0;
					if (lane == 0) got = atomicAdd(traceCursor, take);
					pieceNext = __shfl(got, 0);
					pieceEnd = pieceNext + take;
				}
				const unsigned long long base = pieceNext;
				pieceNext += total;
				if (walk) {
					L.traceBase = base + incl - L.traceCap;
					if (L.traceBase + L.traceCap > traceCapacity) fragRetire(L, EXT_POOL_FULL);   // the pool is full: the host sizes it again and runs the stage again (fragmentPoolsOverflowed)
					else fragWalkBegin(L, m);
				}
			}
		}
		GC_FRAG_MARK(4);
		if (runWalk) while (L.phase == PH_WALK) fragWalkStep(g, P, L, m);
		GC_FRAG_MARK(5);
	}
	__syncthreads();
#ifdef GC_FRAG_STAMPS
	if (lane == 0) for (int i = 0; i < 8; i++) atomicAdd(&stamps[i], stampCycles[i]);
#endif
	if (lane < 7 && waveCounters[lane]) atomicAdd(&counters[lane], (unsigned long long)waveCounters[lane]);   // [6]: extensions handed to the plain-layout kernel
}

// the per-node records, made on the device from the arrays already there (at upload)
__global__ void __launch_bounds__(256) k_build_node_rec(DGraph g, NodeRec* __restrict__ out)
{
	const uint32_t v = blockIdx.x * 256 + threadIdx.x;
	if (v >= g.nNodes) return;
	NodeRec r;
	const uint32_t outBegin = g.outOff[v], inBegin = g.inOff[v];
	const uint32_t outDeg = g.outOff[v + 1] - outBegin, inDeg = g.inOff[v + 1] - inBegin;
	const bool plain = v < g.firstAmbiguous;
	r.comp = g.componentNumber[v]; r.outOff = outBegin; r.inOff = inBegin;
	r.meta = (uint32_t)g.nodeLength[v] | (plain ? 0u : NODEREC_SLOW) | ((outDeg < 255u ? outDeg : 255u) << 8) | ((inDeg < 255u ? inDeg : 255u) << 16);
	r.w0 = plain ? g.nodeSeq[2 * (size_t)v] : 0ull;
	r.w1 = plain ? g.nodeSeq[2 * (size_t)v + 1] : 0ull;
	out[v] = r;
}
void launchBuildNodeRecs(hipStream_t stream, const DGraph& g, NodeRec* out)
{
	if (g.nNodes) hipLaunchKernelGGL(k_build_node_rec, dim3((g.nNodes + 255) / 256), dim3(256), 0, stream, g, out);
}

uint32_t extendFragWaves()
{
	static const uint32_t waves = []() {
		int perCu = 0, dev = 0, cus = 256;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_extend, 64, 0) != hipSuccess || perCu <= 0) perCu = 12;
		hipDeviceProp_t prop;
		if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
		return (uint32_t)(perCu * cus);
	}();
	return waves;
}

void launchExtendFrag(hipStream_t stream, const DGraph& g, const CorrectnessTables* ct, int32_t bandwidth, const ExtItem* work, uint32_t nWork, const FragReads& reads, ExtResult* results,
	uint4* itemScratch, uint32_t scratchWaves, PoolCell* tracePool, unsigned long long* traceCursor, uint64_t traceCapacity, unsigned long long* counters, ExtSelection sel, unsigned long long* claim,
	uint32_t* retryList, unsigned long long* retryCount, unsigned long long* stamps)
{
	if (nWork == 0) return;
	const uint32_t upper = sel.mode == 1 ? 2 * sel.nFrags : nWork;
	uint32_t waves = (upper + 63) / 64;
	if (waves > scratchWaves) waves = scratchWaves;
	if (waves == 0) return;
	hipLaunchKernelGGL(k_extend, dim3(waves), dim3(64), 0, stream, g, ct, bandwidth, work, nWork, reads, results, itemScratch, tracePool, traceCursor, traceCapacity, counters, sel, claim, retryList, retryCount, stamps);
}

} // namespace gcdev
