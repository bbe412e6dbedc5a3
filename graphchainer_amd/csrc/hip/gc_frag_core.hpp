// Fragment seed extension, one extension per LANE, as a per-lane state machine whose phases a wave runs TOGETHER (k_extend, r6).
//
// What it computes: GraphAlignerBitvectorBanded::getReverseTraceFromSeed (src/GraphAlignerBitvectorBanded.h:46-71) for a sequence of at most 64 bases - the
// backward / forward part of a --colinear-split-len fragment (35 by default) beside its seed (src/GraphAligner.h:499-511). Such an extension has ONE slice behind
// the initial one, which is what this core is specialised for (everything else - longer sequences, ambiguous graph nodes, a band that outgrows the tables below -
// answers EXT_OVERFLOW and is rerun by the plain-layout core extendSeedT, gc_device.hpp, which this file restates case by case):
//   - the previous slice of every lookup is the initial slice, i.e. the seed's split node alone with the ramp |column - startOffset| (...Common.h:1243-1279): "is
//     this node in the previous slice" is `node == startNode`, and its horizontal deltas are a function of the column and startOffset;
//   - on a DAG the seed's node is popped first and never reached again, so its tile starts from the source column unmerged and every other tile has no previous-slice
//     twin (hinP = 1, hinN = 0, first row forced: ...Common.h:1052-1117);
//   - the correctness HMM (src/AlignmentCorrectnessEstimation.cpp:105-129) makes ONE step from its initial state: whether the slice is kept is a function of
//     min(minScore, 63) alone - a 64-bit mask made once per launch;
//   - no item of the slice is read by a later slice: an item is its start column, its end column and its node.
//
// Why a state machine: the natural loop nest (pop a node / run its <= 63 columns / push its out-edges / ... / walk back) makes a wave pay for the UNION of its 64
// lanes' paths - r5 measured 6.1 vector wave-instructions per lane column step where 0.9 would do. Here every lane is in one PHASE; the wave runs column steps
// while enough lanes are inside a tile and sweeps the tile-boundary handlers when enough lanes wait at one, each handler with all the lanes that need it.
// State: the pending queue (DP) and the backtrace's column ring (walk) share one region of LDS, word-interleaved across the lanes; the items (52 B) go to a
// lane-interleaved region of HBM (coalesced 16 B per lane); trace cells are written straight into the shared trace pool (the room an extension needs is bounded
// by rows + 1 + score before its walk starts). The graph is read through one 32-byte record per node (NodeRec).
//
// The file is plain C++ over a memory policy M: the kernel instantiates it with LDS / HBM accessors (gc_kernels.hip), tests/frag_host/frag_host_test.cpp with
// arrays and drives single lanes on the CPU against the oracle.
#pragma once
#include "gc_device.hpp"

namespace gcfrag {
using namespace gcdev;

enum : uint32_t { PH_IDLE = 0, PH_FETCH = 1, PH_POP = 2, PH_COLS = 3, PH_TILE_END = 4, PH_FINISH = 5, PH_WALK = 6 };
enum : uint32_t { EXT_POOL_FULL = 8u };   // (inside the kernel only: the shared trace pool had no room for the walk's cells)
enum : uint32_t { TF_START = 1u, TF_WALK = 2u };   // Lane::tileFlags: the tile is the seed's node (the only one with a previous-slice twin); the column loop refills the walk's ring

#ifndef GC_FRAG_QUEUE
#define GC_FRAG_QUEUE 6      // pending nodes of a slice per lane (LDS)
#endif
#ifndef GC_FRAG_ITEMS
#define GC_FRAG_ITEMS 24     // tiles of a slice per lane (HBM, lane-interleaved; 0.4 % of cfg2's extensions hold 17-22, tests/frag_host)
#endif
// the walk keeps the last columns of its current tile in a ring of 32 words per lane: only the rows of the sequence matter to it, so an extension of up to 32 rows
// keeps 16 columns of 2 x 32 bits and a longer one 8 columns of 2 x 64 bits (a refill recomputes the tile from its first column: 133 -> 80 recomputed columns per
// extension between 8 and 16, tests/frag_host)
#define GC_FRAG_RING_WORDS 32

struct FragParams {
	int32_t bandwidth;
	uint64_t keepMask;       // bit i: a slice whose minimum is min(i, 63) above the initial slice's is kept and currently correct (the HMM's one step)
};

// The HMM's single step laid open: the slice is kept when CorrectFromCorrect holds for the initial state (src/GraphAlignerBitvectorBanded.h:589-607) and survives
// removeWronglyAlignedEnd (...Common.h:1231-1241) when it is currently correct; both use only +, max and >= on doubles, as extendSeedT does
__host__ __device__ inline bool fragSliceKept(const CorrectnessTables& ct, int idx)
{
	const bool cfc = ct.initCorrect + ct.c2c >= ct.initFalse + ct.f2c;
	const double a = ct.initCorrect + ct.c2c, b = ct.initFalse + ct.f2c;
	const double c = ct.initCorrect + ct.c2f, d = ct.initFalse + ct.f2f;
	const double correct = (a > b ? a : b) + ct.correctOdds[idx];
	const double wrong = (c > d ? c : d) + ct.wrongOdds[idx];
	return cfc && correct > wrong;
}

// Per-lane working memory. S supplies the storage: ld / st of 32-bit words (the kernel: LDS, word w of lane l at [w * 64 + l]), the items and the trace pool;
// FragMem lays the queue (DP) and the ring (walk) over the SAME words - a lane is in one of the two parts of an extension at a time.
constexpr uint32_t FRAG_Q = GC_FRAG_QUEUE, FRAG_I = GC_FRAG_ITEMS;
constexpr uint32_t FRAG_WORDS = 7 * FRAG_Q > GC_FRAG_RING_WORDS ? 7 * FRAG_Q : GC_FRAG_RING_WORDS;
template <class S>
struct FragMem : S {
	__device__ uint64_t ld64(uint32_t w) { return (uint64_t)this->ld(w) | ((uint64_t)this->ld(w + 1) << 32); }
	__device__ void st64(uint32_t w, uint64_t v) { this->st(w, (uint32_t)v); this->st(w + 1, (uint32_t)(v >> 32)); }
	// pending queue: VP [0, 2Q), VN [2Q, 4Q), score [4Q, 5Q), node [5Q, 6Q), componentNumber [6Q, 7Q)
	__device__ uint64_t qVP(uint32_t i) { return ld64(2 * i); }
	__device__ uint64_t qVN(uint32_t i) { return ld64(2 * FRAG_Q + 2 * i); }
	__device__ int32_t qScore(uint32_t i) { return (int32_t)this->ld(4 * FRAG_Q + i); }
	__device__ uint32_t qNode(uint32_t i) { return this->ld(5 * FRAG_Q + i); }
	__device__ uint32_t qComp(uint32_t i) { return this->ld(6 * FRAG_Q + i); }
	__device__ void qSetColumn(uint32_t i, uint64_t VP, uint64_t VN, int32_t score) { st64(2 * i, VP); st64(2 * FRAG_Q + 2 * i, VN); this->st(4 * FRAG_Q + i, (uint32_t)score); }
	__device__ void qSet(uint32_t i, uint64_t VP, uint64_t VN, int32_t score, uint32_t node, uint32_t comp) { qSetColumn(i, VP, VN, score); this->st(5 * FRAG_Q + i, node); this->st(6 * FRAG_Q + i, comp); }
	// the walk's ring, column c (already cut to the sequence's rows): narrow (<= 32 rows) VP at [c & 15], VN at [16 + (c & 15)]; wide VP at [2 (c & 7)], VN at [16 + 2 (c & 7)]
	__device__ uint64_t ringVP(uint32_t c, bool wide) { return wide ? ld64(2 * (c & 7u)) : (uint64_t)this->ld(c & 15u); }
	__device__ uint64_t ringVN(uint32_t c, bool wide) { return wide ? ld64(16 + 2 * (c & 7u)) : (uint64_t)this->ld(16 + (c & 15u)); }
	__device__ void ringSet(uint32_t c, uint64_t VP, uint64_t VN, bool wide)
	{
		if (wide) { st64(2 * (c & 7u), VP); st64(16 + 2 * (c & 7u), VN); }
		else { this->st(c & 15u, (uint32_t)VP); this->st(16 + (c & 15u), (uint32_t)VN); }
	}
	__device__ uint32_t idGet(uint32_t k) { return this->ld(GC_FRAG_RING_WORDS + k); }
	__device__ void idSet(uint32_t k, uint32_t node) { this->st(GC_FRAG_RING_WORDS + k, node); }
};
__device__ inline uint32_t fragRingColumns(uint32_t len) { return len > 32 ? 8u : 16u; }
// the words the ring leaves of the queue's region hold the first items' node ids during the walk ("which item is node v": asked at every tile the walk enters and for every
// in-neighbour at a crossing - from the item planes that is one dependent load per item looked at)
constexpr uint32_t FRAG_IDS = FRAG_WORDS - GC_FRAG_RING_WORDS;

struct Lane {
	uint32_t phase;
	uint32_t work;            // index of the work item (0xffffffff: none); its result is written when the lane comes back to PH_FETCH
	uint32_t status;
	uint32_t len;             // rows, 1..64
	uint32_t startNode, startOffset, startLen;
	Eq4 eq;
	// the slice
	uint32_t nItems, nPending;
	int32_t curMin;           // minimum of row 63 over the slice's tiles so far (cur.minScore before the flattening = currentMin of calculateSlice)
	int32_t flatMin; uint32_t flatNode, flatOffset;   // flattenLastSliceEnd fused into the column loop (bit 31 of flatOffset: the minimum is attained in a second node)
	// the tile in the column loop (DP: a popped node; walk: the recompute of the tile the walk stands in)
	uint32_t node, pos, tileLen, tileFlags, nodeLen;
	uint64_t VP, VN; int32_t score;
	int32_t flat;             // DP: the column's value at the sequence's last row (what flattenLastSliceEnd reads, ...Common.h:1210-1218), kept by its horizontal deltas
	uint64_t w0, w1;
	int32_t tileMin;
	uint32_t outBegin, outDeg, t0, t1;   // the tile's out-edges; the first two targets are fetched when the tile is popped and used when it ends
	// the walk
	uint64_t HP, HN;                     // the recomputed columns' scores at the sequence's LAST row: the first column's and the deltas from column to column (the walk never looks above that row)
	uint32_t n0, n1;          // the first two in-neighbours of the walk's tile (fetched when it is entered)
	// the rest of the walk's state lives in the registers the DP has left (a lane is in one of the two parts of an extension at a time; the kernel sits at the register
	// count of four waves per SIMD): where the walk stands, the item of its tile, the ring's lowest column, a pending refill, the tile's in-edges, the first column's score
	__device__ uint32_t& hereNode() { return flatNode; }
	__device__ uint32_t& hereOffset() { return flatOffset; }
	__device__ int32_t& hereSeqPos() { return flatMin; }
	__device__ uint32_t& curItem() { return nPending; }
	__device__ uint32_t& ringLo() { return outBegin; }
	__device__ uint32_t& refillTo() { return outDeg; }
	__device__ uint32_t& inBegin() { return t0; }
	__device__ uint32_t& inDeg() { return t1; }
	__device__ int32_t& colStart() { return tileMin; }
	__device__ uint32_t curItem() const { return nPending; }
	__device__ uint32_t inBegin() const { return t0; }
	__device__ uint32_t inDeg() const { return t1; }
	uint32_t nTrace, traceCap; uint64_t traceBase;
	int32_t resultScore; uint32_t tie;
	// counters of this extension, in the reference's units (ExtCounters): tiles and columns, the DP's in the low halves, the walk's in the high halves
	uint32_t cntTiles, cntCols;
};

__device__ inline uint64_t eqOfColumn2(const Eq4& eq, uint64_t w0, uint64_t w1, uint32_t pos)
{
	const uint64_t w = pos < 32 ? w0 : w1;
	const uint32_t code = (uint32_t)(w >> ((pos & 31) * 2));
	const uint64_t lo = (code & 1) ? eq.c : eq.a;
	const uint64_t hi = (code & 1) ? eq.t : eq.g;
	return (code & 2) ? hi : lo;
}

__device__ inline void fragRetire(Lane& L, uint32_t status) { L.status = status; L.phase = PH_FETCH; }

__device__ inline uint64_t fragFlatMask(uint32_t len) { return len >= 64 ? ~0ull : ((1ull << len) - 1); }

// the fused flattenLastSliceEnd (...Common.h:1210-1218), slice-wide: strict '<' in pop order and column order; a column AT the minimum in another node is the tie
__device__ inline void fragFlatUpdate(Lane& L, uint32_t pos)
{
	const int32_t f = L.flat;
	if (f < L.flatMin) { L.flatMin = f; L.flatNode = L.node; L.flatOffset = pos; }
	else if (f == L.flatMin && L.node != L.flatNode) L.flatOffset |= 0x80000000u;
}

// ---- PH_FETCH -> PH_POP: a new work item. `rows` yields the match masks of the item's rows (EqFromBases / EqSource of gc_device.hpp)
template <class M, class EQS>
__device__ inline void fragBegin(const DGraph& g, const FragParams& P, Lane& L, M& m, uint32_t work, uint32_t len, uint32_t node, uint32_t offset, const EQS& rows)
{
	L.work = work; L.status = EXT_OK; L.len = len; L.startNode = node; L.startOffset = offset;
	L.nTrace = 0; L.resultScore = 0; L.tie = 0;
	L.cntTiles = L.cntCols = 0;
	if (len == 0) { fragRetire(L, EXT_FAILED); return; }      // no slice: table.slices.size() <= 1
	if (len > 64) { fragRetire(L, EXT_OVERFLOW); return; }    // more than one slice: the plain-layout core
	const NodeRec r = g.nodeRec[node];
	if (r.meta & NODEREC_SLOW) { fragRetire(L, EXT_OVERFLOW); return; }
	L.startLen = r.meta & 127u;
	rows.rows((int)len, 0, L.eq);
	// the queue holds the seed's node with the source column of the initial slice (...Banded.h:235-277 with j == 0: every node of the previous slice)
	m.qSet(0, ~0ull, 0ull, (int32_t)offset + 64, node, r.comp);
	L.nPending = 1; L.nItems = 0;
	L.curMin = INT32_MAX - P.bandwidth - 1;
	L.flatMin = INT32_MAX; L.flatNode = 0xffffffffu; L.flatOffset = 0x7fffffffu;
	L.node = 0xffffffffu;
	L.phase = PH_POP;
}

// ---- PH_POP -> PH_COLS | PH_TILE_END | PH_FINISH: the pending node of lowest topological rank becomes the next tile (ComponentPriorityQueue order on a DAG)
template <class M>
__device__ inline void fragPop(const DGraph& g, const FragParams& P, Lane& L, M& m)
{
	if (L.nPending == 0) { L.phase = PH_FINISH; return; }
	uint32_t best = 0, bestComp = m.qComp(0);
	for (uint32_t i = 1; i < L.nPending; i++) { GC_LOOP_TICK(2); const uint32_t c = m.qComp(i); if (c < bestComp) { best = i; bestComp = c; } }
	const uint32_t node = m.qNode(best);
	uint64_t VP = m.qVP(best), VN = m.qVN(best); int32_t score = m.qScore(best);
	const uint32_t last = L.nPending - 1;
	if (best != last) m.qSet(best, m.qVP(last), m.qVN(last), m.qScore(last), m.qNode(last), m.qComp(last));
	L.nPending = last;
	if (L.nItems >= GC_FRAG_ITEMS) { fragRetire(L, EXT_OVERFLOW); return; }
	const NodeRec r = g.nodeRec[node];
	const uint32_t nodeLength = r.meta & 127u;
	const bool isStart = node == L.startNode;
	// the seed's node is the slice's first tile and starts from the source column as it is: its "row above" (computeTile's prevExists part) equals the source's
	// score before the start, so neither the merge nor the delta fix-up of ...Common.h:1052-1117 changes anything. Anything else is not a DAG's doing.
	if (isStart && !(VP == ~0ull && VN == 0ull && score == (int32_t)L.startOffset + 64)) { fragRetire(L, EXT_OVERFLOW); return; }
	L.node = node; L.w0 = r.w0; L.w1 = r.w1;
	L.outBegin = r.outOff; L.outDeg = (r.meta >> 8) & 255u;
	L.t0 = L.outDeg > 0 ? g.outAdj[r.outOff] : 0u;
	L.t1 = L.outDeg > 1 ? g.outAdj[r.outOff + 1] : 0u;
	L.VP = VP; L.VN = VN; L.score = score;
	L.tileMin = score;   // (sic) the tile's minimum starts from the first column before any merge, ...Common.h:968
	m.itemSetStart(L.nItems, VP, VN, score, node);
	{
		const uint64_t above = ~fragFlatMask(L.len);
		L.flat = score - popc64(VP & above) + popc64(VN & above);
	}
	fragFlatUpdate(L, 0);
	L.pos = 1; L.tileLen = nodeLength; L.nodeLen = nodeLength; L.tileFlags = isStart ? TF_START : 0u;
	L.cntTiles++; L.cntCols += nodeLength;
	L.phase = nodeLength > 1 ? PH_COLS : PH_TILE_END;
}

// ---- PH_COLS: one column of the tile (getNextSlice, ...Common.h:243-263, with the carries of ...Common.h:1118-1161). Written with selects instead of branches: the
// lanes of a wave are in DP tiles and walk tiles, seed tiles and others at the same time, and every branch that assigns loop-carried state costs register copies
template <class M>
__device__ inline void fragColumnStep(Lane& L, M& m)
{
	const uint32_t pos = L.pos;
	const bool isStart = (L.tileFlags & TF_START) != 0, walk = (L.tileFlags & TF_WALK) != 0;
	uint64_t Eq = eqOfColumn2(L.eq, L.w0, L.w1, pos);
	// the row above: the seed's tile has the initial slice's ramp (-1 up to the seed's column, +1 behind it), every other tile nothing (+1, and the first row forced)
	const uint64_t hinN = (isStart && pos <= L.startOffset) ? 1ull : 0ull;
	const uint64_t hinP = hinN ^ 1ull;
	const uint64_t force = isStart ? 0ull : 1ull;
	Eq &= ~force;
	// myersStep (gc_device.hpp) with the horizontal deltas of every row left in Ph / Mh
	const uint64_t VP0 = L.VP, VN0 = L.VN;
	const uint64_t Xv = Eq | VN0;
	Eq |= hinN;
	const uint64_t Xh = (((Eq & VP0) + VP0) ^ VP0) | Eq;
	const uint64_t Ph = VN0 | ~(Xh | VP0);
	const uint64_t Mh = VP0 & Xh;
	const uint64_t sMh = (Mh << 1) | hinN, sPh = (Ph << 1) | hinP;
	const uint64_t VP = (sMh | ~(Xv | sPh)) & ~force;
	const uint64_t VN = (sPh & Xv) | force;
	L.VP = VP; L.VN = VN;
	const uint32_t top = L.len - 1;
	const int32_t dTop = (int32_t)((Ph >> top) & 1ull) - (int32_t)((Mh >> top) & 1ull);   // the step of the sequence's last row from the previous column to this one
	if (walk) {
		// the walk's columns: cut to the sequence's rows, their score taken at its last row
		const uint64_t rows = fragFlatMask(L.len);
		m.ringSet(pos, VP & rows, VN & rows, L.len > 32);
		L.HP |= ((Ph >> top) & 1ull) << pos; L.HN |= ((Mh >> top) & 1ull) << pos;
	}
	// the DP's bookkeeping (its registers hold the walk's state during a walk: selects keep that in place)
	const int32_t score = L.score + (int32_t)(Ph >> 63) - (int32_t)(Mh >> 63);
	const int32_t flat = L.flat + dTop;   // (forcing the first row's delta does not move any row's value, only the implied row above)
	L.score = score;
	L.flat = walk ? L.flat : flat;
	L.tileMin = (!walk && score < L.tileMin) ? score : L.tileMin;
	{
		// the fused flattenLastSliceEnd, as fragFlatUpdate
		const bool lower = !walk && flat < L.flatMin, tied = !walk && flat == L.flatMin && L.node != L.flatNode;
		L.flatMin = lower ? flat : L.flatMin;
		L.flatNode = lower ? L.node : L.flatNode;
		L.flatOffset = lower ? pos : (tied ? (L.flatOffset | 0x80000000u) : L.flatOffset);
	}
	L.pos = pos + 1;
}
template <class M>
__device__ inline void fragColumn(Lane& L, M& m)
{
	fragColumnStep(L, m);
	if (L.pos >= L.tileLen) L.phase = (L.tileFlags & TF_WALK) ? PH_WALK : PH_TILE_END;
}
// up to BURST columns of the tile before the lane looks up: the kernel's column loop checks its triggers once per burst (a lane whose tile ends inside a burst idles
// for the rest of it - it would wait for the sweep anyway)
template <int BURST, class M>
__device__ inline void fragColumns(Lane& L, M& m)
{
#pragma unroll
	for (int u = 0; u < BURST; u++) if (L.pos < L.tileLen) fragColumnStep(L, m);
	if (L.pos >= L.tileLen) L.phase = (L.tileFlags & TF_WALK) ? PH_WALK : PH_TILE_END;
}

// one out-edge of a finished tile folded into the queue (the per-edge part of calculateNodeInner, ...Common.h:903-964, for a target without a previous-slice twin)
template <class M>
__device__ inline void fragPush(Lane& L, M& m, uint32_t target, const NodeRec& r, const WS& end)
{
	if (target == L.startNode) { fragRetire(L, EXT_OVERFLOW); return; }   // a cycle through the seed's node: not this core's case
	if (r.meta & NODEREC_SLOW) { fragRetire(L, EXT_OVERFLOW); return; }
	uint64_t hp, hn;
	WS add = myersStep(eqOfColumn2(L.eq, r.w0, r.w1, 0), end, 1, 0, hp, hn);
	add.VP &= ~1ull; add.VN |= 1ull;
	uint32_t slot = L.nPending;
	for (uint32_t i = 0; i < L.nPending; i++) { GC_LOOP_TICK(3); if (m.qNode(i) == target) { slot = i; break; } }
	if (slot == L.nPending) {
		if (L.nPending >= GC_FRAG_QUEUE) { fragRetire(L, EXT_OVERFLOW); return; }
		m.qSet(slot, add.VP, add.VN, add.score, target, r.comp);
		L.nPending++;
	} else {
		const WS merged = wsMerge(WS { m.qVP(slot), m.qVN(slot), m.qScore(slot) }, add);
		m.qSetColumn(slot, merged.VP, merged.VN, merged.score);
	}
}

// ---- PH_TILE_END -> PH_POP: the tile's item is complete; band test and out-edges (calculateSlice, ...Banded.h:340-400)
template <class M>
__device__ inline void fragTileEnd(const DGraph& g, const FragParams& P, Lane& L, M& m)
{
	m.itemSetEnd(L.nItems, L.VP, L.VN, L.score);
	L.nItems++;
	if (L.tileMin > 1 + P.bandwidth + 128) { fragRetire(L, EXT_ASSERT); return; }   // ...Banded.h:352 (previousQuitScore = 0 + 1)
	if (L.tileMin < L.curMin) L.curMin = L.tileMin;
	const WS end { L.VP, L.VN, L.score };
	const int32_t endMin = wsColumnMin(end);
	if (endMin < 0) { fragRetire(L, EXT_ASSERT); return; }   // ...Banded.h:368 (previousMinScore = 0)
	L.phase = PH_POP;
	if (endMin <= L.curMin + P.bandwidth) {
		if (L.outDeg == 255u) { fragRetire(L, EXT_OVERFLOW); return; }
		// (the records of the first two targets are asked for together: one round trip for the common node instead of one per edge)
		const uint32_t deg = L.outDeg;
		NodeRec r0 {}, r1 {};
		if (deg > 0) r0 = g.nodeRec[L.t0];
		if (deg > 1) r1 = g.nodeRec[L.t1];
		if (deg > 0) fragPush(L, m, L.t0, r0, end);
		if (deg > 1 && L.phase == PH_POP) fragPush(L, m, L.t1, r1, end);
		for (uint32_t e = 2; e < deg && L.phase == PH_POP; e++) { const uint32_t t = g.outAdj[L.outBegin + e]; fragPush(L, m, t, g.nodeRec[t], end); }
	}
}

// ---- PH_FINISH: the slice is complete. Returns true when the walk is to start (the caller reserves L.traceCap cells of the pool at L.traceBase and calls fragWalkBegin)
__device__ inline bool fragFinish(const FragParams& P, Lane& L)
{
	const int32_t minScore = L.flatMin;
	if (L.nItems == 0 || minScore < 0) { fragRetire(L, EXT_ASSERT); return false; }   // ...Banded.h:463 (the initial slice's minimum is 0)
	if (!((P.keepMask >> (minScore < 64 ? minScore : 63)) & 1ull)) { fragRetire(L, EXT_FAILED); return false; }
	if (minScore > (int32_t)L.len + 128) { fragRetire(L, EXT_ASSERT); return false; }
	L.resultScore = minScore;
	L.tie = L.len < 64 ? (L.flatOffset >> 31) : 0u;
	// cells of the walk: the start cell, one per row down to row -1, one per horizontal step (each costs one, the ramp of row -1 included): rows + 1 + score (met with
	// equality, never exceeded in tests/frag_host); the steps outside the band are not priced: what does not fit answers EXT_OVERFLOW
	L.traceCap = L.len + (uint32_t)minScore + 2u;
	return true;
}

template <class M>
__device__ inline bool fragTracePush(Lane& L, M& m, uint32_t node, uint32_t offset, int32_t seqPos, bool nodeSwitch)
{
	if (L.nTrace >= L.traceCap) { fragRetire(L, EXT_OVERFLOW); return false; }
	m.traceSet(L.traceBase + L.nTrace, node, seqPos, offset | (nodeSwitch ? 256u : 0u));
	L.nTrace++;
	return true;
}

template <class M>
__device__ inline void fragWalkBegin(Lane& L, M& m)
{
	{
		// the first items' node ids into the words the queue has left (all loads first, then the stores)
		uint32_t id[FRAG_IDS];
		for (uint32_t k = 0; k < FRAG_IDS; k++) id[k] = m.itemNode(k < L.nItems ? k : 0u);
		for (uint32_t k = 0; k < FRAG_IDS; k++) m.idSet(k, id[k]);
	}
	L.hereNode() = L.flatNode; L.hereOffset() = L.flatOffset & 0x7fffffffu; L.hereSeqPos() = (int32_t)L.len - 1;   // (the first two are where they already are)
	L.node = 0xffffffffu; L.refillTo() = 0xffffffffu; L.ringLo() = 0; L.curItem() = 0;
	L.phase = PH_WALK;
	fragTracePush(L, m, L.hereNode(), L.hereOffset(), L.hereSeqPos(), false);
}

template <class M>
__device__ inline int fragFindItem(const Lane& L, M& m, uint32_t node)
{
	const uint32_t inLds = L.nItems < FRAG_IDS ? L.nItems : FRAG_IDS;
	for (uint32_t k = 0; k < inLds; k++) { GC_LOOP_TICK(4); if (m.idGet(k) == node) return (int)k; }
	for (uint32_t k = FRAG_IDS; k < L.nItems; k++) { GC_LOOP_TICK(4); if (m.itemNode(k) == node) return (int)k; }
	return -1;
}

struct FragCell { uint32_t node, offset; int32_t seqPos; };

// pickBacktraceCorner (...Common.h:710-804) at the first row of the only slice: the row above is the initial slice
template <class M>
__device__ inline bool fragCorner(const DGraph& g, const FragParams& P, const Lane& L, M& m, FragCell& out, bool& nodeSwitch)
{
	const uint32_t node = L.node;
	const int32_t quitScore = L.resultScore + P.bandwidth;
	const int32_t previousQuitScore = 1;
	const int32_t scoreHere = wsValue(m.itemStart(L.curItem()), 0);
	const bool prevSelf = node == L.startNode;
	const int32_t initStart = (int32_t)L.startOffset, initEnd = (int32_t)L.startLen - 1 - (int32_t)L.startOffset;   // the initial item's first and last score
	if (scoreHere > quitScore) {
		int32_t smallest = scoreHere + 1;
		out = FragCell { 0, 0, 0 };
		nodeSwitch = false;
		if (prevSelf) { smallest = initStart; out = FragCell { node, 0, -1 }; }
		for (uint32_t e = 0; e < L.inDeg(); e++) {
			const uint32_t nb = e == 0 ? L.n0 : e == 1 ? L.n1 : g.inAdj[L.inBegin() + e];
			if (nb == L.startNode && initEnd <= smallest) { smallest = initEnd; out = FragCell { nb, L.startLen - 1, -1 }; nodeSwitch = true; }
			const int c = fragFindItem(L, m, nb);
			if (c >= 0 && nb != node) {
				const int32_t v = wsValue(m.itemEnd((uint32_t)c), 0);
				if (v < smallest) { smallest = v; out = FragCell { nb, (uint32_t)g.nodeLength[nb] - 1, 0 }; nodeSwitch = true; }
			}
		}
		return true;
	}
	const int eqBit = (int)(eqOfColumn2(L.eq, L.w0, L.w1, 0) & 1);
	if (prevSelf && initStart == scoreHere - 1) { out = FragCell { node, 0, -1 }; nodeSwitch = false; return true; }
	FragCell bestInvalid { 0xffffffffu, 0xffffffffu, -1 };
	int32_t bestInvalidScore = scoreHere + 1;
	for (uint32_t e = 0; e < L.inDeg(); e++) {
		const uint32_t nb = e == 0 ? L.n0 : e == 1 ? L.n1 : g.inAdj[L.inBegin() + e];
		const int c = fragFindItem(L, m, nb);
		if (c >= 0 && wsValue(m.itemEnd((uint32_t)c), 0) == scoreHere - 1) { out = FragCell { nb, (uint32_t)g.nodeLength[nb] - 1, 0 }; nodeSwitch = true; return true; }
		if (nb == L.startNode) {
			const int32_t corner = initEnd;
			if (corner > previousQuitScore) {
				if (corner < bestInvalidScore) { bestInvalidScore = corner; bestInvalid = FragCell { nb, L.startLen - 1, -1 }; }
			} else if (corner == scoreHere - (eqBit ? 0 : 1)) {
				out = FragCell { nb, L.startLen - 1, -1 }; nodeSwitch = true; return true;
			}
		}
	}
	if (bestInvalidScore < scoreHere + 1) { out = bestInvalid; nodeSwitch = true; return true; }
	return false;   // the reference asserts here
}

// ---- PH_WALK: getReverseTraceFromTable's loop (...Common.h:392-544) from where the lane stands up to the next tile whose columns have to be (re)computed (PH_COLS) or to
// the end of the trace (PH_FETCH). The reference's loop does ONE thing per turn, chosen by where it stands: recompute the tile it enters / a corner / a vertical crossing /
// a horizontal crossing / steps inside the tile. Here the cases are laid out in the order a walk meets them - inside the tile, then the crossing at its edge, then the ramp
// of row -1 or the next tile - and each is guarded by the loop's own condition, re-read where it stands: one call is several turns of that loop, and the lanes of a wave,
// which come back from the column loop together, run each case's code once per call instead of once per turn (r6: 70 turns per 64 extensions -> 21, 33 % of the kernel's
// wave-cycles in this handler before)
template <class M>
__device__ inline void fragWalkStep(const DGraph& g, const FragParams& P, Lane& L, M& m)
{
	const uint32_t NO_REFILL = 0xffffffffu;
	auto column = [&](uint32_t c) -> WS {
		const uint64_t mask = (c >= 63 ? ~0ull : ((2ull << c) - 1)) & ~1ull;
		return WS { m.ringVP(c, L.len > 32), m.ringVN(c, L.len > 32), L.colStart() + popc64(L.HP & mask) - popc64(L.HN & mask) };   // (a column of rows 0..len-1: the rows above hold zeros)
	};
	// the tile the walk stands in has its columns in the ring: the cases below may run
	auto inTile = [&]() { return L.phase == PH_WALK && L.hereSeqPos() != -1 && L.hereNode() == L.node && L.refillTo() == NO_REFILL; };
	auto moveTo = [&](const FragCell& c) { L.hereNode() = c.node; L.hereOffset() = c.offset; L.hereSeqPos() = c.seqPos; };
	auto corner = [&]() {
		FragCell nxt; bool sw;
		if (!fragCorner(g, P, L, m, nxt, sw)) { fragRetire(L, EXT_ASSERT); return; }
		if (!fragTracePush(L, m, nxt.node, nxt.offset, nxt.seqPos, sw)) return;
		moveTo(nxt);
	};
	const int32_t quitScore = L.resultScore + P.bandwidth, previousQuitScore = 1;

	// ---- inside the tile (pickBacktraceInside, ...Common.h:556-597): vertical, then diagonal, then horizontal
	if (inTile() && L.hereSeqPos() > 0 && L.hereOffset() > 0) {
		const uint32_t curNode = L.node;
		uint32_t hori = L.hereOffset();
		int vert = L.hereSeqPos();   // (the only slice starts at row 0)
		while (hori > 0 && vert > 0) {
			GC_LOOP_TICK(5);
			if (hori - 1 < L.ringLo()) { L.refillTo() = hori; break; }
			const WS colHere = column(hori), colLeft = column(hori - 1);
			const int32_t scoreHere = wsValue(colHere, vert);
			const int32_t vertical = wsValue(colHere, vert - 1);
			const int32_t diagonal = wsValue(colLeft, vert - 1);
			const int eqBit = (int)((eqOfColumn2(L.eq, L.w0, L.w1, hori) >> vert) & 1);
			if (vertical == scoreHere - 1) { vert--; }
			else if (diagonal == scoreHere - (eqBit ? 0 : 1)) { hori--; vert--; }
			else {
				if (wsValue(colLeft, vert) != scoreHere - 1) { fragRetire(L, EXT_ASSERT); break; }
				hori--;
			}
			if (!fragTracePush(L, m, curNode, hori, vert, false)) break;
		}
		if (L.phase == PH_WALK) { L.hereOffset() = hori; L.hereSeqPos() = vert; }
	}
	// ---- first row, not the first column: vertical crossing into the initial slice (...Common.h:451-477, pickBacktraceVerticalCrossing :665-708)
	if (inTile() && L.hereSeqPos() == 0 && L.hereOffset() > 0) {
		const uint32_t curNode = L.node;
		if (curNode != L.startNode) {
			// the node is not in the initial slice: along the first row to the tile's first column (the corner below)
			L.hereOffset() = 0;
			fragTracePush(L, m, curNode, 0, 0, false);
		} else {
			uint32_t off = L.hereOffset();
			while (off > 0) {
				if (off - 1 < L.ringLo()) { L.refillTo() = off; break; }
				if (wsValue(column(off - 1), 0) != wsValue(column(off), 0) - 1) break;
				off--;
				if (!fragTracePush(L, m, curNode, off, 0, false)) break;
			}
			if (L.phase == PH_WALK) L.hereOffset() = off;
			if (L.phase == PH_WALK && L.refillTo() == NO_REFILL && off > 0) {
				// the initial item: scores |column - startOffset|, i.e. deltas -1 up to the seed's column and +1 behind it
				const uint64_t upToOffset = L.startOffset >= 63 ? ~0ull : ((1ull << (L.startOffset + 1)) - 1);
				const uint64_t nodeMask = L.startLen >= 64 ? ~0ull : ((1ull << L.startLen) - 1);
				const uint64_t pnHN = upToOffset & ~1ull, pnHP = nodeMask & ~upToOffset;
				const int32_t scoreHere = wsValue(column(off), 0);
				int32_t scoreDiagonal = (int32_t)L.startOffset;
				const uint64_t lowMask = off >= 1 ? (((1ull << off) - 1) & ~1ull) : 0ull;
				scoreDiagonal += popc64(pnHP & lowMask) - popc64(pnHN & lowMask);
				const int32_t scoreUp = scoreDiagonal + (int32_t)((pnHP >> off) & 1) - (int32_t)((pnHN >> off) & 1);
				FragCell nxt { curNode, off, -1 };
				bool ok = true;
				if (scoreHere > quitScore || scoreDiagonal > previousQuitScore || scoreUp > previousQuitScore) {
					nxt = scoreDiagonal < scoreUp ? FragCell { curNode, off - 1, -1 } : FragCell { curNode, off, -1 };
				} else {
					const int eqBit = (int)(eqOfColumn2(L.eq, L.w0, L.w1, off) & 1);
					if (scoreUp == scoreHere - 1) nxt = FragCell { curNode, off, -1 };
					else if (scoreDiagonal == scoreHere - (eqBit ? 0 : 1)) nxt = FragCell { curNode, off - 1, -1 };
					else { fragRetire(L, EXT_ASSERT); ok = false; }
				}
				if (ok && fragTracePush(L, m, nxt.node, nxt.offset, nxt.seqPos, false)) moveTo(nxt);
			}
		}
	}
	// ---- first column, not the first row: horizontal crossing into an in-neighbour (...Common.h:478-499, pickBacktraceHorizontalCrossing :599-663)
	if (inTile() && L.hereSeqPos() > 0 && L.hereOffset() == 0) {
		const uint32_t curNode = L.node;
		const WS start = m.itemStart(L.curItem());
		int32_t sp = L.hereSeqPos();
		while ((sp & 63) != 0 && (start.VP & (1ull << (sp & 63)))) {
			sp--;
			if (!fragTracePush(L, m, curNode, 0, sp, false)) break;
		}
		if (L.phase == PH_WALK) {
			L.hereSeqPos() = sp;
			const int offset = sp & 63;
			if (offset != 0) {   // (row 0: the corner below)
				const int eqBit = (int)((eqOfColumn2(L.eq, L.w0, L.w1, 0) >> offset) & 1);
				const int32_t scoreHere = wsValue(start, offset);
				FragCell nxt { 0, 0, 0 };
				bool sw = false, found = false;
				if (scoreHere > quitScore) {
					int32_t smallest = wsValue(start, offset - 1);
					nxt = FragCell { curNode, 0, sp - 1 };
					for (uint32_t e = 0; e < L.inDeg(); e++) {
						const uint32_t nb = e == 0 ? L.n0 : e == 1 ? L.n1 : g.inAdj[L.inBegin() + e];
						const int c = fragFindItem(L, m, nb);
						if (c < 0) continue;
						const WS ne = m.itemEnd((uint32_t)c);
						const uint32_t nbLast = (uint32_t)g.nodeLength[nb] - 1;
						if (wsValue(ne, offset - 1) <= smallest) { smallest = wsValue(ne, offset - 1); nxt = FragCell { nb, nbLast, sp - 1 }; sw = true; }
						if (wsValue(ne, offset) < smallest && nb != curNode) { smallest = wsValue(ne, offset); nxt = FragCell { nb, nbLast, sp }; sw = true; }
					}
					found = true;
				} else {
					for (uint32_t e = 0; e < L.inDeg() && !found; e++) {
						const uint32_t nb = e == 0 ? L.n0 : e == 1 ? L.n1 : g.inAdj[L.inBegin() + e];
						const int c = fragFindItem(L, m, nb);
						if (c < 0) continue;
						const WS ne = m.itemEnd((uint32_t)c);
						const uint32_t nbLast = (uint32_t)g.nodeLength[nb] - 1;
						if (wsValue(ne, offset) == scoreHere - 1) { nxt = FragCell { nb, nbLast, sp }; sw = true; found = true; }
						else if (wsValue(ne, offset - 1) == scoreHere - (eqBit ? 0 : 1)) { nxt = FragCell { nb, nbLast, sp - 1 }; sw = true; found = true; }
					}
				}
				if (!found) fragRetire(L, EXT_ASSERT);
				else if (fragTracePush(L, m, nxt.node, nxt.offset, nxt.seqPos, sw)) moveTo(nxt);
			}
		}
	}
	// ---- first row, first column: the corner (pickBacktraceCorner)
	if (inTile() && L.hereSeqPos() == 0 && L.hereOffset() == 0) corner();
	// ---- row -1: left along the initial ramp towards the seed's column (...Common.h:508-542; the in-neighbour hop needs a second node in the initial slice)
	if (L.phase == PH_WALK && L.hereSeqPos() == -1) {
		if (L.hereNode() != L.startNode) { fragRetire(L, EXT_ASSERT); return; }
		uint32_t off = L.hereOffset();
		while (true) {
			int32_t b = (int32_t)off - (int32_t)L.startOffset; if (b < 0) b = -b;
			int32_t bl = (int32_t)off - 1 - (int32_t)L.startOffset; if (bl < 0) bl = -bl;
			if (!(b != 0 && off > 0 && bl == b - 1)) break;
			off--;
			if (!fragTracePush(L, m, L.hereNode(), off, -1, false)) return;
		}
		fragRetire(L, EXT_OK);
		return;
	}
	// ---- the next tile, or more columns of this one: recalcNodeWordslice (...Common.h:828-852) into the ring, up to the column the walk stands on (it only moves left)
	if (L.phase != PH_WALK) return;
	const bool entering = L.hereNode() != L.node;
	if (entering) {
		const int item = fragFindItem(L, m, L.hereNode());
		if (item < 0) { fragRetire(L, EXT_ASSERT); return; }
		const NodeRec r = g.nodeRec[L.hereNode()];
		L.curItem() = (uint32_t)item; L.node = L.hereNode();
		L.w0 = r.w0; L.w1 = r.w1;
		L.inBegin() = r.inOff; L.inDeg() = (r.meta >> 16) & 255u;
		L.tileFlags = TF_WALK | (L.hereNode() == L.startNode ? TF_START : 0u);
		L.nodeLen = r.meta & 127u;
		L.cntTiles += 1u << 16; L.cntCols += L.nodeLen << 16;
		if (L.inDeg() == 255u) { fragRetire(L, EXT_OVERFLOW); return; }
		L.n0 = L.inDeg() > 0 ? g.inAdj[r.inOff] : 0u;
		L.n1 = L.inDeg() > 1 ? g.inAdj[r.inOff + 1] : 0u;
	}
	if (entering || L.refillTo() != NO_REFILL) {
		const uint32_t upTo = entering ? L.hereOffset() : L.refillTo();
		const WS start = m.itemStart(L.curItem());
		const uint64_t rows = fragFlatMask(L.len);
		L.VP = start.VP; L.VN = start.VN;
		L.colStart() = wsValue(start, (int)L.len - 1); L.HP = 0; L.HN = 0;
		m.ringSet(0, start.VP & rows, start.VN & rows, L.len > 32);
		const uint32_t ring = fragRingColumns(L.len);
		L.ringLo() = upTo > ring - 1 ? upTo - (ring - 1) : 0;
		L.refillTo() = NO_REFILL;
		L.pos = 1;
		L.tileLen = upTo + 1 < L.nodeLen ? upTo + 1 : L.nodeLen;
		if (L.tileLen > 1) L.phase = PH_COLS;
	}
}

} // namespace gcfrag
