// k_long_extend_sm: the whole-read pass's seed extensions, one per LANE, as per-lane state machines (gc_sm_core.hpp).
//
// Why: with one extension per wave (k_long_extend<1>, gc_device_wave.hpp) the extension's state lives in scalar registers and the CU's
// single scalar unit - shared by all 32 resident waves - is what the kernel saturates (7.6e10 scalar instructions per 10 k-read batch, 84 %
// of the scalar issue peak, 2.4 % of the HBM yardstick); the vector pipes idle. Several extensions per wave in the natural loop nest pay
// max-over-lanes at every level (node length, nodes per slice, trace length) and measured slower. Here a wave holds LANES extensions that
// are each in one of four phases; the wave loop votes, runs one phase for the lanes that are in it, and votes again.
//
// Layout: per lane an HBM slab (NodeItems 64 B, slice records 32 B, the 64 walk-mask columns of the backtrace's current tile), contiguous
// per lane (the lanes are at different items, so interleaving buys no coalescing; a 64 B item is one full sector); per lane 304 words of LDS,
// lane-interleaved ([word][lane]): the pending queue (16 entries: node, componentNumber, column) and the node tables of the current and the
// previous slice (32 entries: node, start score, minimum). The trace goes straight into the round's trace pool, into a slot reserved when the
// extension starts (its length is bounded by 1.5 x rows + 512, the bound the host sizes the pool by).
// An extension that outgrows a table answers EXT_SM_DECLINED and is rerun by k_long_extend<1> (gc_batch.hip: runLongGroup).
#include "gc_kernels.hpp"
#include "gc_sm_core.hpp"

namespace gcdev {

using namespace gcsm;

typedef __attribute__((address_space(3))) uint32_t sm_lds_u32;   // keeps the table accesses ds_read / ds_write (a generic pointer compiles to flat_*)
template <int LANES> struct SmLdsView {
	sm_lds_u32* base;
	uint32_t lane;
	__device__ __forceinline__ uint32_t ld(uint32_t w) const { return base[w * LANES + lane]; }
	__device__ __forceinline__ void st(uint32_t w, uint32_t v) const { base[w * LANES + lane] = v; }
};

struct SmPolicy { uint32_t colBurst, walkBurst, weightCol, weightB, weightBt, weightWalk; };

template <int LANES>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) k_long_extend_sm(DGraph g, const CorrectnessTables* __restrict__ ctp, const uint64_t* __restrict__ masks, SmParams P, SmPolicy pol,
	const LongWork* __restrict__ work, const uint32_t* __restrict__ order, uint32_t nWork, uint8_t* __restrict__ scratch, uint64_t slabBytes,
	unsigned long long* __restrict__ tracePool, unsigned long long* __restrict__ traceCursor, uint64_t traceCapacity, LongWorkResult* __restrict__ results, unsigned long long* __restrict__ counters,
	unsigned long long* __restrict__ nextSlot)
{
	__shared__ uint32_t lds[SM_LANE_WORDS * LANES];
	if (LANES < 64 && threadIdx.x >= LANES) return;
	const uint32_t lane = threadIdx.x;
	const SmLdsView<LANES> view { (sm_lds_u32*)&lds[0], lane };
	const CorrectnessTables& ct = *ctp;
	SmLane L {};
	{
		uint8_t* slab = scratch + ((uint64_t)blockIdx.x * LANES + lane) * slabBytes;
		L.items = (NodeItem*)slab;
		L.slices = (SmSlice*)(slab + (uint64_t)P.maxItems * sizeof(NodeItem));
		L.cols = (SmWalkCol*)(slab + (uint64_t)P.maxItems * sizeof(NodeItem) + (uint64_t)P.maxSlices * sizeof(SmSlice));
	}
	L.state = SM_FETCH;
#ifdef GC_SM_STAMPS
	// profiling build (make variant NAME=smstamps FLAGS=-DGC_SM_STAMPS): wave-cycles and executions per phase, lanes served per execution
	unsigned long long stCyc[5] = { 0, 0, 0, 0, 0 }, stCnt[5] = { 0, 0, 0, 0, 0 }, stLanes[5] = { 0, 0, 0, 0, 0 }, stMark = __builtin_amdgcn_s_memtime();
#define SM_STAMP(i, lanes) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stCyc[i] += now_ - stMark; stMark = now_; stCnt[i]++; stLanes[i] += (lanes); } while (0)
#else
#define SM_STAMP(i, lanes) ((void)0)
#endif
	ExtCounters total {};   // work of the extensions this lane finished (a declined one is rerun, and counted, by k_long_extend<1>)
	while (true) {
		// ---- housekeeping: finished lanes publish their result and fetch the next work item (execution order: longest first)
		if (__any(L.state == SM_RETIRE || L.state == SM_FETCH)) {
			if (L.state == SM_RETIRE) {
				LongWorkResult res { 0, 0, L.status, L.score, 0 };
				if (L.status == EXT_OK) { res.traceOff = (uint64_t)(L.trace - tracePool); res.traceLen = L.nTrace; }
				results[L.work] = res;
				if (L.status != EXT_SM_DECLINED) {
					total.dpTiles += L.cnt.dpTiles; total.recomputeTiles += L.cnt.recomputeTiles; total.columnSteps += L.cnt.columnSteps;
					total.traceItems += L.cnt.traceItems; total.extensions += L.cnt.extensions; total.backtraceTiles += L.cnt.backtraceTiles;
				}
				L.state = SM_FETCH;
			}
			if (L.state == SM_FETCH) {
				const unsigned long long slot = atomicAdd(nextSlot, 1ull);
				if (slot >= nWork) L.state = SM_IDLE;
				else {
					const uint32_t w = order[slot];
					const LongWork it = work[w];
					L.work = w;
					if (it.seqLen == 0) results[w] = LongWorkResult { 0, 0, EXT_FAILED, 0, 0 };   // (stays in SM_FETCH: next item on the next turn)
					else {
						const unsigned long long cap = (unsigned long long)it.seqLen + it.seqLen / 2 + 512;
						const unsigned long long base = atomicAdd(traceCursor, cap);
						if (base + cap > traceCapacity) {
							atomicAdd(traceCursor, 0ull - cap);
							results[w] = LongWorkResult { 0, 0, EXT_SM_DECLINED, 0, 0 };   // the exact-size path of k_long_extend<1> decides
						} else {
							L.len = (int32_t)it.seqLen;
							L.startNode = it.node; L.startOffset = it.offset;
							L.masks = masks + it.maskOff; L.maskWords = it.maskWords; L.startBit = it.startBit;
							L.trace = tracePool + base; L.traceCap = (uint32_t)cap;
							L.cnt = ExtCounters {};
							smBegin(g, ct, P, L, view);
						}
					}
				}
			}
		}
		// ---- vote: which phase runs next
		const uint32_t nB = (uint32_t)__popcll(__ballot(L.state == SM_B)), nCol = (uint32_t)__popcll(__ballot(L.state == SM_COL));
		const uint32_t nBt = (uint32_t)__popcll(__ballot(L.state == SM_BT)), nWalk = (uint32_t)__popcll(__ballot(L.state == SM_WALK));
		if (nB + nCol + nBt + nWalk == 0) {
			if (__any(L.state == SM_RETIRE || L.state == SM_FETCH)) continue;
			break;   // every lane is idle: the work list is exhausted
		}
		const uint32_t sB = nB * pol.weightB, sCol = nCol * pol.weightCol, sBt = nBt * pol.weightBt, sWalk = nWalk * pol.weightWalk;
		uint32_t pick = SM_COL, top = sCol;
		if (sWalk > top) { pick = SM_WALK; top = sWalk; }
		if (sB > top) { pick = SM_B; top = sB; }
		if (sBt > top) { pick = SM_BT; top = sBt; }
		SM_STAMP(4, 0);   // housekeeping + vote
		if (pick == SM_COL) {
			for (uint32_t k = 0; k < pol.colBurst; k++) {
				if (L.state == SM_COL) smPhaseCol(g, L);
				SM_STAMP(1, nCol);
				if (!__any(L.state == SM_COL)) break;
			}
		} else if (pick == SM_WALK) {
			for (uint32_t k = 0; k < pol.walkBurst; k++) {
				if (L.state == SM_WALK) smPhaseWalk(L);
				SM_STAMP(3, nWalk);
				if (!__any(L.state == SM_WALK)) break;
			}
		} else if (pick == SM_B) {
			if (L.state == SM_B) smPhaseB(g, ct, P, L, view);
			SM_STAMP(0, nB);
		} else {
			if (L.state == SM_BT) smPhaseBt(g, P, L, view);
			SM_STAMP(2, nBt);
		}
	}
#ifdef GC_SM_STAMPS
	if (threadIdx.x == 0) for (int i = 0; i < 5; i++) { atomicAdd(&counters[8 + i], stCyc[i]); atomicAdd(&counters[13 + i], stCnt[i]); atomicAdd(&counters[18 + i], stLanes[i]); }
#endif
	if (total.extensions) {
		atomicAdd(&counters[0], (unsigned long long)total.dpTiles);
		atomicAdd(&counters[1], (unsigned long long)total.recomputeTiles);
		atomicAdd(&counters[2], (unsigned long long)total.columnSteps);
		atomicAdd(&counters[3], (unsigned long long)total.traceItems);
		atomicAdd(&counters[4], (unsigned long long)total.extensions);
		atomicAdd(&counters[5], (unsigned long long)total.backtraceTiles);
	}
}

uint64_t longSmSlabBytes(const ExtendConfig& cfg)
{
	SmParams P { cfg.bandwidth, cfg.maxItems, cfg.maxSlices };
	return (smSlabBytes(P) + 255) & ~255ull;
}

static uint32_t envU32(const char* name, uint32_t dflt) { const char* e = getenv(name); return e ? (uint32_t)atoi(e) : dflt; }

// lanes per wave for a round of nWork extensions (GC_LONG_SM_LANES overrides): a wave's instruction count grows with its lanes (they are
// in different phases), so small rounds - which cost one wave's latency - run few lanes per wave and spread over the chip's 1024 SIMDs
uint32_t longSmLanes(uint32_t nWork)
{
	const uint32_t forced = envU32("GC_LONG_SM_LANES", 0);
	if (forced == 8 || forced == 16 || forced == 32 || forced == 64) return forced;
	if (nWork <= 8 * 1024) return 8;
	if (nWork <= 16 * 1024) return 16;
	if (nWork <= 32 * 1024) return 32;
	return 64;
}

void launchLongExtendSm(hipStream_t stream, const DGraph& g, const CorrectnessTables* ct, const uint64_t* masks, const ExtendConfig& cfg, const LongWork* work, const uint32_t* order, uint32_t nWork,
	uint8_t* scratch, uint64_t scratchBytes, unsigned long long* tracePool, unsigned long long* traceCursor, uint64_t traceCapacity, LongWorkResult* results, unsigned long long* counters, unsigned long long* nextSlot)
{
	if (!nWork) return;
	const SmParams P { cfg.bandwidth, cfg.maxItems, cfg.maxSlices };
	const uint64_t slab = longSmSlabBytes(cfg);
	const uint32_t lanes = longSmLanes(nWork);
	uint64_t blocks = ((uint64_t)nWork + lanes - 1) / lanes;
	blocks = std::min<uint64_t>(blocks, std::max<uint64_t>(1, scratchBytes / (slab * lanes)));
	blocks = std::min<uint64_t>(blocks, envU32("GC_LONG_SM_MAX_BLOCKS", 4096));
	const SmPolicy pol { envU32("GC_LONG_SM_COL_BURST", 8), envU32("GC_LONG_SM_WALK_BURST", 8), envU32("GC_LONG_SM_W_COL", 4), envU32("GC_LONG_SM_W_B", 4), envU32("GC_LONG_SM_W_BT", 4), envU32("GC_LONG_SM_W_WALK", 4) };
#define GC_LAUNCH_SM(N) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_long_extend_sm<N>), dim3((uint32_t)blocks), dim3(64), 0, stream, g, ct, masks, P, pol, work, order, nWork, scratch, slab, tracePool, traceCursor, traceCapacity, results, counters, nextSlot)
	switch (lanes) {
		case 8: GC_LAUNCH_SM(8); break;
		case 16: GC_LAUNCH_SM(16); break;
		case 32: GC_LAUNCH_SM(32); break;
		default: GC_LAUNCH_SM(64); break;
	}
#undef GC_LAUNCH_SM
}

} // namespace gcdev
