// Whole-read seed extension as a per-lane STATE MACHINE (k_long_extend_sm): one extension per LANE, many per wave.
//
// Same algorithm and results as extendSeed (gc_device.hpp) / extendSeedWave (gc_device_wave.hpp) - the reference's
// getReverseTraceFromSeed (src/GraphAlignerBitvectorBanded.h:46-71): slices of 64 read rows, nodes popped in componentNumber order,
// the band rule on a running minimum, WordSlice merges, the correctness HMM, removeWronglyAlignedEnd, the backtrace with its
// crossing rules - but cut into PHASES that each advance one lane by one event:
//
//   SM_B     tile boundary: finish the tile just computed (item record, band test, out-edge pushes), close / open a slice when the
//            pending queue is empty (HMM, slice record, source pushes), pop the next node and set its tile up
//   SM_COL   one Myers column of the current tile (DP, or the backtrace's recompute which also leaves the walk masks of the column)
//   SM_BT    backtrace at a tile boundary: (slice, node) change -> lookups + recompute set-up; corner / vertical / horizontal crossing
//   SM_WALK  one cell of the walk inside a tile (three bit tests on the column's walk masks)
//
// A wave holds LANES independent extensions in different phases. The kernel's wave loop votes (ballots over the lanes' states),
// runs ONE phase for the lanes that are in it, and votes again: a lane never waits for another lane's loop bounds (node length, nodes
// per slice, slices, trace length) - the cost of the natural nesting, where a wave pays max-over-lanes at every level - only for its
// phase's turn. With the extension's state in VGPRs the vector pipe does the column steps that the one-extension-per-wave layout runs
// on the CU's single scalar unit.
//
// This header is plain C++ (no HIP builtins): the kernel wrapper (gc_sm.hip) supplies the LDS table view and the wave loop, and
// tests/sm_host compiles the same phases with g++ and drives ONE lane on the CPU against the oracle's extension
// (test infrastructure only: the library has no CPU path and never calls these functions on the host).
#pragma once
#include "gc_device.hpp"

namespace gcsm {
using namespace gcdev;

#if defined(__HIPCC__)
#define SM_FN __device__ __forceinline__
#else
#define SM_FN inline
#endif

enum : uint32_t { SM_FETCH = 0, SM_B = 1, SM_COL = 2, SM_BT = 3, SM_WALK = 4, SM_RETIRE = 5, SM_IDLE = 6 };
enum : uint32_t { EXT_SM_DECLINED = 6 };   // a table of this layout is too small for the extension: the one-extension-per-wave kernel takes it

// per-lane tables (LDS on the device): pending queue + the node tables of the current and the previous slice
constexpr uint32_t SM_PENDING_CAP = 16, SM_TABLE_CAP = 32;
constexpr uint32_t SM_PENDING_WORDS = 7, SM_TABLE_WORDS = 3;
constexpr uint32_t SM_TABLE_BASE = SM_PENDING_CAP * SM_PENDING_WORDS;
constexpr uint32_t SM_LANE_WORDS = SM_TABLE_BASE + 2 * SM_TABLE_CAP * SM_TABLE_WORDS;   // 304 words = 1216 B per lane

struct SmSlice { int32_t minScore; uint32_t minNode, minOffset, first, count; int32_t bandwidth; int32_t j; uint32_t flags; };   // 32 B (as WSlice)
struct SmWalkCol { uint64_t up, diag, left; int32_t row0; uint32_t pad; };   // one recomputed column of the backtrace's current tile, 32 B

struct SmParams {
	int32_t bandwidth;
	uint32_t maxItems, maxSlices;
};
inline uint64_t smSlabBytes(const SmParams& p) { return (uint64_t)p.maxItems * sizeof(NodeItem) + (uint64_t)p.maxSlices * sizeof(SmSlice) + 64 * sizeof(SmWalkCol); }

// one trace cell per 8-byte word, the format k_long_merge reads (unpackCell, gc_device_wave.hpp): node | (seqPos+1) << 32 (24 bits) | offset << 56 (6 bits) | nodeSwitch << 62
SM_FN unsigned long long smPackCell(uint32_t node, uint32_t offset, int32_t seqPos, bool sw) { return (unsigned long long)node | ((unsigned long long)(uint32_t)(seqPos + 1) << 32) | ((unsigned long long)offset << 56) | ((unsigned long long)(sw ? 1 : 0) << 62); }

SM_FN int smPopc(uint64_t x) { return __builtin_popcountll(x); }
SM_FN int32_t smBefore(const WS& w) { return w.score - smPopc(w.VP) + smPopc(w.VN); }
SM_FN int32_t smValue(const WS& w, int row) { uint64_t above = row < 63 ? (~0ull << (row + 1)) : 0ull; return w.score + smPopc(w.VN & above) - smPopc(w.VP & above); }

// pointwise minimum of two columns (src/WordSlice.h:491-530), as wsMerge
SM_FN WS smMerge(const WS& a, const WS& b)
{
	int32_t d = smBefore(a) - smBefore(b);
	uint64_t takeB = 0, fixP = 0, fixN = 0, fixMask = 0;
	uint64_t diff = (a.VP ^ b.VP) | (a.VN ^ b.VN);
	int pos = 0;
	while (diff) {
		int r = __builtin_ctzll(diff);
		diff &= diff - 1;
		uint64_t bit = 1ull << r;
		if (d > 0 && r > pos) takeB |= (bit - 1) & ~((1ull << pos) - 1);
		int da = (int)((a.VP >> r) & 1) - (int)((a.VN >> r) & 1);
		int db = (int)((b.VP >> r) & 1) - (int)((b.VN >> r) & 1);
		int nd = d + da - db;
		bool before = d > 0, after = nd > 0;
		if (before != after) {
			int delta = after ? (db - d) : (da + d);
			fixMask |= bit;
			if (delta > 0) fixP |= bit;
			if (delta < 0) fixN |= bit;
		}
		if (after) takeB |= bit;
		d = nd;
		pos = r + 1;
	}
	if (d > 0 && pos < 64) takeB |= ~((1ull << pos) - 1);
	WS res;
	res.VP = (((a.VP & ~takeB) | (b.VP & takeB)) & ~fixMask) | fixP;
	res.VN = (((a.VN & ~takeB) | (b.VN & takeB)) & ~fixMask) | fixN;
	res.score = a.score < b.score ? a.score : b.score;
	return res;
}

// min over rows -1..63 of a column (changedMinScore against an absent old column, src/WordSlice.h:252), as wsColumnMin
SM_FN int32_t smColumnMin(const WS& w)
{
	int32_t before = smBefore(w);
	int32_t best = before;
	uint64_t vn = w.VN;
	while (vn) {
		int r = __builtin_ctzll(vn);
		uint64_t run = vn & ~(vn + (1ull << r));
		int last = 63 - __builtin_clzll(run);
		uint64_t upto = last < 63 ? ((1ull << (last + 1)) - 1) : ~0ull;
		int32_t v = before + smPopc(w.VP & upto) - smPopc(w.VN & upto);
		best = v < best ? v : best;
		vn &= ~upto;
	}
	return best;
}

// ---- per-lane state ------------------------------------------------------------------------------------
struct SmLane {
	uint32_t state, status;
	// the work item
	uint32_t work;                       // index of the work item (results stay indexed by it)
	int32_t len, numSlices;
	uint32_t startNode, startOffset;
	const uint64_t* masks; uint32_t maskWords, startBit;
	unsigned long long* trace;           // this extension's reserved slot of the trace pool
	uint32_t traceCap, nTrace;
	int32_t score;
	// the lane's HBM slab
	NodeItem* items; SmSlice* slices; SmWalkCol* cols;
	// DP: slice level
	uint32_t fresh;
	int32_t slice, prevJ, j;
	uint32_t nSlices, nItems;
	int32_t prevMinScore, prevBandwidth, previousQuitScore;
	double prevCorrect, prevFalse;
	uint32_t nPrev, prevFirst, pb;       // previous slice: entries, first item, table buffer (the current slice uses buffer pb ^ 1)
	uint32_t nPending;
	uint32_t curFirst, curCount; int32_t curMinScore; uint32_t curMinNode, curMinOffset;
	int32_t flatMin; uint32_t flatNode, flatOffset;
	int32_t currentMin, flatRows;
	Eq4 eq;
	// tile level (DP and backtrace recompute)
	uint32_t tileActive, btMode, amb;
	uint32_t node; int32_t nodeLength, pos;
	uint64_t VP, VN; int32_t colScore;
	uint64_t HP, HN, pHP, pHN, codes, w1;
	int32_t forceUntil; uint64_t forceEq;
	int32_t tMinScore; uint32_t tMinOffset; int32_t tFlatMin; uint32_t tFlatOffset;
	// backtrace
	uint32_t hereNode, hereOffset; int32_t hereSeqPos;
	uint32_t curSliceIdx, curNode, btVerify;
	SmSlice cs, ps;
	uint64_t cSVP, cSVN, cEVP, cEVN; int32_t cSScore, cEScore;   // current item: first and last column
	uint32_t pExists; int32_t pSScore; uint64_t pHPall, pHNall;  // the same node's item in the previous slice
	uint32_t hori; int32_t vert; uint64_t up, diag, left; uint32_t unfit;
	ExtCounters cnt;
};

// ---- table views ------------------------------------------------------------------------------------------
// T provides ld(word) / st(word, value) on the lane's SM_LANE_WORDS words.
template <class T> struct SmTab {
	const T& t;
	SM_FN uint32_t qNode(uint32_t e) const { return t.ld(e * SM_PENDING_WORDS); }
	SM_FN uint32_t qComp(uint32_t e) const { return t.ld(e * SM_PENDING_WORDS + 1); }
	SM_FN WS qWs(uint32_t e) const
	{
		const uint32_t b = e * SM_PENDING_WORDS;
		WS w;
		w.score = (int32_t)t.ld(b + 2);
		w.VP = (uint64_t)t.ld(b + 3) | ((uint64_t)t.ld(b + 4) << 32);
		w.VN = (uint64_t)t.ld(b + 5) | ((uint64_t)t.ld(b + 6) << 32);
		return w;
	}
	SM_FN void qSetWs(uint32_t e, const WS& w) const
	{
		const uint32_t b = e * SM_PENDING_WORDS;
		t.st(b + 2, (uint32_t)w.score); t.st(b + 3, (uint32_t)w.VP); t.st(b + 4, (uint32_t)(w.VP >> 32)); t.st(b + 5, (uint32_t)w.VN); t.st(b + 6, (uint32_t)(w.VN >> 32));
	}
	SM_FN void qSet(uint32_t e, uint32_t node, uint32_t comp, const WS& w) const { t.st(e * SM_PENDING_WORDS, node); t.st(e * SM_PENDING_WORDS + 1, comp); qSetWs(e, w); }
	SM_FN void qMove(uint32_t dst, uint32_t src) const { for (uint32_t k = 0; k < SM_PENDING_WORDS; k++) t.st(dst * SM_PENDING_WORDS + k, t.ld(src * SM_PENDING_WORDS + k)); }
	SM_FN uint32_t tNode(uint32_t b, uint32_t e) const { return t.ld(SM_TABLE_BASE + (b * SM_TABLE_CAP + e) * SM_TABLE_WORDS); }
	SM_FN int32_t tStart(uint32_t b, uint32_t e) const { return (int32_t)t.ld(SM_TABLE_BASE + (b * SM_TABLE_CAP + e) * SM_TABLE_WORDS + 1); }
	SM_FN int32_t tMin(uint32_t b, uint32_t e) const { return (int32_t)t.ld(SM_TABLE_BASE + (b * SM_TABLE_CAP + e) * SM_TABLE_WORDS + 2); }
	SM_FN void tSet(uint32_t b, uint32_t e, uint32_t node, int32_t start, int32_t mn) const
	{
		const uint32_t w = SM_TABLE_BASE + (b * SM_TABLE_CAP + e) * SM_TABLE_WORDS;
		t.st(w, node); t.st(w + 1, (uint32_t)start); t.st(w + 2, (uint32_t)mn);
	}
	SM_FN int find(uint32_t b, uint32_t n, uint32_t node) const { for (uint32_t i = 0; i < n; i++) if (tNode(b, i) == node) return (int)i; return -1; }
	SM_FN int qFind(uint32_t n, uint32_t node) const { for (uint32_t i = 0; i < n; i++) if (qNode(i) == node) return (int)i; return -1; }
};

SM_FN void smRetire(SmLane& L, uint32_t status) { L.status = status; L.state = SM_RETIRE; }

SM_FN bool smPushTrace(SmLane& L, uint32_t node, uint32_t offset, int32_t seqPos, bool sw)
{
	if (L.nTrace >= L.traceCap) { smRetire(L, EXT_OVERFLOW); return false; }
	L.trace[L.nTrace++] = smPackCell(node, offset, seqPos, sw);
	return true;
}

// ---- start of an extension: the initial slice (src/GraphAlignerBitvectorCommon.h:1243-1279) -----------------------------
template <class T>
SM_FN void smBegin(const DGraph& g, const CorrectnessTables& ct, const SmParams& P, SmLane& L, const T& t)
{
	const SmTab<T> tab { t };
	L.status = EXT_OK;
	L.nTrace = 0;
	L.score = 0;
	L.cnt.extensions++;
	L.numSlices = (L.len + 63) / 64;
	if ((uint32_t)L.numSlices + 1 > P.maxSlices) { smRetire(L, EXT_OVERFLOW); return; }
	const int nl = g.nodeLength[L.startNode];
	NodeItem it;
	it.node = L.startNode;
	it.sVP = it.sVN = it.eVP = it.eVN = 0;
	it.sScore = (int32_t)L.startOffset;
	it.eScore = nl - 1 - (int32_t)L.startOffset;
	it.minScore = 0;
	const uint64_t upToOffset = L.startOffset >= 63 ? ~0ull : ((1ull << (L.startOffset + 1)) - 1);
	const uint64_t nodeMask = nl >= 64 ? ~0ull : ((1ull << nl) - 1);
	it.HN = upToOffset & ~1ull;
	it.HP = nodeMask & ~upToOffset;
	L.items[0] = it;
	SmSlice s0;
	s0.minScore = 0; s0.minNode = L.startNode; s0.minOffset = L.startOffset; s0.first = 0; s0.count = 1; s0.bandwidth = 1; s0.j = -64; s0.flags = 1;
	L.slices[0] = s0;
	L.pb = 0;
	tab.tSet(0, 0, L.startNode, it.sScore, 0);
	L.nPrev = 1; L.prevFirst = 0;
	L.nItems = 1; L.nSlices = 1;
	L.prevMinScore = 0; L.prevBandwidth = 1; L.prevJ = -64;
	L.prevCorrect = ct.initCorrect; L.prevFalse = ct.initFalse;
	L.slice = 0;
	L.fresh = 1;
	L.nPending = 0;
	L.tileActive = 0;
	L.btMode = 0;
	L.state = SM_B;
}

// Folds one incoming edge into a pending node (src/GraphAlignerBitvectorCommon.h:903-964), as pushEdge in gc_device.hpp
template <class T>
SM_FN void smPushEdge(const DGraph& g, SmLane& L, const SmTab<T>& tab, uint32_t target, const WS& incoming, bool skipFirst)
{
	const int found = tab.qFind(L.nPending, target);
	WS add = incoming;
	if (!skipFirst) {
		const int pi = tab.find(L.pb, L.nPrev, target);
		const bool prevExists = pi >= 0;
		const int32_t prevStart = prevExists ? tab.tStart(L.pb, (uint32_t)pi) : 0;
		uint64_t hinP, hinN;
		if (prevExists) {
			const int32_t before = smBefore(incoming);
			if (prevStart < before) { hinP = 0; hinN = 1; }
			else if (prevStart > before) { hinP = 1; hinN = 0; }
			else { hinP = 0; hinN = 0; }
		} else { hinP = 1; hinN = 0; }
		const NodeSeq nseq = loadNodeSeq(g, target);
		uint64_t hp, hn;
		add = myersStep(eqOfColumn(L.eq, nseq, 0), incoming, hinP, hinN, hp, hn);
		if (!prevExists || smBefore(add) < prevStart) { add.VP &= ~1ull; add.VN |= 1ull; }
	}
	if (found < 0) {
		if (L.nPending >= SM_PENDING_CAP) { smRetire(L, EXT_SM_DECLINED); return; }
		tab.qSet(L.nPending, target, g.componentNumber[target], add);
		L.nPending++;
	} else {
		tab.qSetWs((uint32_t)found, smMerge(tab.qWs((uint32_t)found), add));
	}
}

// The part of computeTile (gc_device.hpp) before its column loop: merge with the row above, first-row repair, first column.
// prevHP / prevHN come in whole and leave shifted so that bit 0 is the carry of column 1.
SM_FN void smTileSetup(const DGraph& g, SmLane& L, uint32_t node, WS ws, bool prevExists, int32_t prevStartScore, uint64_t prevHP, uint64_t prevHN, int flatRows)
{
	L.node = node;
	L.nodeLength = g.nodeLength[node];
	const NodeSeq seq = loadNodeSeq(g, node);
	L.amb = seq.ambiguous ? 1u : 0u;
	L.codes = seq.w0 >> 2;
	L.w1 = seq.w1;
	L.tMinScore = ws.score;   // (sic) before the merge with the row above, ...Common.h:968 vs :1052-1058
	L.tMinOffset = 0;
	if (prevExists && smBefore(ws) > prevStartScore) ws = smMerge(ws, wsSource(prevStartScore));
	int forceUntil = 0;
	if (prevExists) {
		int32_t scoreBefore = smBefore(ws);
		int32_t scoreComparison = prevStartScore;
		if (scoreBefore > scoreComparison) L.status = EXT_ASSERT;
		if (scoreBefore < scoreComparison) {
			for (int fix = 1; fix < 64; fix++) {
				const int32_t next = scoreComparison + (int32_t)((prevHP >> fix) & 1) - (int32_t)((prevHN >> fix) & 1);
				const uint64_t mask = 1ull << fix;
				if (scoreBefore > next) L.status = EXT_ASSERT;
				if (scoreBefore < next) { prevHP |= mask; prevHN &= ~mask; forceUntil = fix; }
				if (scoreBefore == next) { prevHP &= ~mask; prevHN &= ~mask; }
				scoreBefore++;
				scoreComparison = next;
				if (scoreBefore >= scoreComparison) break;
			}
		}
	} else {
		forceUntil = L.nodeLength;
	}
	L.forceUntil = forceUntil;
#ifdef SM_HOST_STATS
	{ extern unsigned long long g_smStat[8]; const unsigned long long c = (unsigned long long)(L.nodeLength - 1); g_smStat[L.btMode ? 4 : 0] += c; if (!prevExists) g_smStat[(L.btMode ? 4 : 0) + 1] += c; else if (forceUntil == 0) g_smStat[(L.btMode ? 4 : 0) + 2] += c; else g_smStat[(L.btMode ? 4 : 0) + 3] += c; }
#endif
	L.forceEq = prevExists ? ~0ull : ~1ull;
	L.VP = ws.VP; L.VN = ws.VN; L.colScore = ws.score;
	L.pHP = prevHP >> 1; L.pHN = prevHN >> 1;
	L.HP = 0; L.HN = 0;
	L.tFlatMin = INT32_MAX;
	L.tFlatOffset = 0;
	if (flatRows > 0) {
		const uint64_t flatMask = ~(~0ull << flatRows);
		L.tFlatMin = ws.score - smPopc(ws.VP & ~flatMask) + smPopc(ws.VN & ~flatMask);
	}
	L.pos = 1;
}

// ---- SM_COL: one Myers column (src/GraphAlignerBitvectorCommon.h:243-263,1118-1161) ------------------------------------
SM_FN void smPhaseCol(const DGraph& g, SmLane& L)
{
	const int pos = L.pos;
	uint64_t rawEq;
	if (!L.amb) {
		const uint32_t code = (uint32_t)L.codes & 3u;
		const uint64_t lo = (code & 1) ? L.eq.c : L.eq.a, hi = (code & 1) ? L.eq.t : L.eq.g;
		rawEq = (code & 2) ? hi : lo;
		L.codes >>= 2;
		if (pos == 31) L.codes = L.w1;
	} else {
		rawEq = eqOfColumn(L.eq, loadNodeSeq(g, L.node), pos);
	}
	uint64_t Eq = rawEq & L.forceEq;
	const uint64_t hinP = L.pHP & 1ull, hinN = L.pHN & 1ull;
	L.pHP >>= 1; L.pHN >>= 1;
	const uint64_t pVP = L.VP, pVN = L.VN;
	const uint64_t Xv = Eq | pVN;
	Eq |= hinN;
	const uint64_t Xh = (((Eq & pVP) + pVP) ^ pVP) | Eq;
	const uint64_t Ph = pVN | ~(Xh | pVP);
	const uint64_t Mh = pVP & Xh;
	const uint64_t sMh = (Mh << 1) | hinN, sPh = (Ph << 1) | hinP;
	const uint64_t hp = Ph >> 63, hn = Mh >> 63;
	uint64_t VP = sMh | ~(Xv | sPh);
	uint64_t VN = sPh & Xv;
	if (L.forceUntil >= pos) { VP &= ~1ull; VN |= 1ull; }
	const int32_t score = L.colScore - (int32_t)hn + (int32_t)hp;
	L.VP = VP; L.VN = VN; L.colScore = score;
	if (score < L.tMinScore) { L.tMinScore = score; L.tMinOffset = (uint32_t)pos; }
	if (!L.btMode && L.flatRows > 0) {
		const uint64_t flatMask = ~(~0ull << L.flatRows);
		const int32_t f = score - smPopc(VP & ~flatMask) + smPopc(VN & ~flatMask);
		if (f < L.tFlatMin) { L.tFlatMin = f; L.tFlatOffset = (uint32_t)pos; }
	}
	L.HP |= hp << pos;
	L.HN |= hn << pos;
	if (L.btMode) {
		// walk masks of this column (see setWalkMasks, gc_device_wave.hpp): value(r,c) - value(r,c-1) = Ph_r - Mh_r is an identity of the step, so
		// "the diagonal predecessor fits" and "the left predecessor fits" are bit operations on the step's own intermediates
		const uint64_t flat = ~(Ph | Mh), level = ~(pVP | pVN);
		const uint64_t same = (flat & level) | (Ph & pVN) | (Mh & pVP);
		const uint64_t more = (Ph & level) | (flat & pVP);
		SmWalkCol c;
		c.up = VP;
		c.diag = (same & rawEq) | (more & ~rawEq);
		c.left = Ph;
		c.row0 = score + smPopc(VN & ~1ull) - smPopc(VP & ~1ull);
		c.pad = 0;
		L.cols[pos] = c;
	}
	L.pos = pos + 1;
	if (L.pos >= L.nodeLength) L.state = L.btMode ? SM_BT : SM_B;
}

// ---- SM_B ---------------------------------------------------------------------------------------------------
// start of the backtrace (after the last slice): removeWronglyAlignedEnd (...Common.h:1231-1241) and the start cell
SM_FN void smBacktraceBegin(SmLane& L)
{
	bool currentlyCorrect = (L.slices[L.nSlices - 1].flags & 1u) != 0;
	while (!currentlyCorrect) {
		currentlyCorrect = (L.slices[L.nSlices - 1].flags & 4u) != 0;
		L.nSlices--;
		if (L.nSlices == 0) break;
	}
	if (L.nSlices <= 1) { smRetire(L, EXT_FAILED); return; }
	const SmSlice last = L.slices[L.nSlices - 1];
	if (last.minScore < 0 || last.minScore > L.len + 128) { smRetire(L, EXT_ASSERT); return; }
	L.score = last.minScore;
	L.hereNode = last.minNode; L.hereOffset = last.minOffset; L.hereSeqPos = (last.j + 63 < L.len - 1) ? last.j + 63 : L.len - 1;
	if (!smPushTrace(L, L.hereNode, L.hereOffset, L.hereSeqPos, false)) return;
	L.curSliceIdx = 0xffffffffu; L.curNode = 0xffffffffu;
	L.btVerify = 0;
	L.btMode = 1;
	L.state = SM_BT;
}

template <class T>
SM_FN void smPhaseB(const DGraph& g, const CorrectnessTables& ct, const SmParams& P, SmLane& L, const T& t)
{
	const SmTab<T> tab { t };
	const uint32_t cb = L.pb ^ 1u;
	const int bandwidth = P.bandwidth;
	if (L.tileActive) {
		// ---- the tile just computed (calculateSlice's loop body after calculateNodeInner, ...Banded.h:336-395)
		L.tileActive = 0;
		if (L.status != EXT_OK) { smRetire(L, L.status); return; }
		NodeItem& out = L.items[L.nItems];
		out.eVP = L.VP; out.eVN = L.VN; out.HP = L.HP; out.HN = L.HN;
		out.eScore = L.colScore; out.minScore = L.tMinScore;
		const int32_t sScore = out.sScore;
		tab.tSet(cb, L.curCount, L.node, sScore, L.tMinScore);
		L.nItems++;
		L.curCount++;
		L.cnt.dpTiles++;
		L.cnt.columnSteps += (uint32_t)L.nodeLength;
		if (L.flatRows > 0) { L.cnt.recomputeTiles++; L.cnt.columnSteps += (uint32_t)L.nodeLength; }
		if (L.tMinScore > L.previousQuitScore + bandwidth + 128) { smRetire(L, EXT_ASSERT); return; }
		L.currentMin = L.tMinScore < L.currentMin ? L.tMinScore : L.currentMin;
		if (L.tMinScore < L.curMinScore) { L.curMinScore = L.tMinScore; L.curMinNode = L.node; L.curMinOffset = L.tMinOffset; }
		if (L.flatRows > 0 && L.tFlatMin < L.flatMin) { L.flatMin = L.tFlatMin; L.flatNode = L.node; L.flatOffset = L.tFlatOffset; }
		const WS newEnd { L.VP, L.VN, L.colScore };
		const int32_t newEndMin = smColumnMin(newEnd);
		if (newEndMin < L.prevMinScore) { smRetire(L, EXT_ASSERT); return; }
		if (newEndMin <= L.currentMin + bandwidth) {
			const uint32_t e0 = g.outOff[L.node], e1 = g.outOff[L.node + 1];
			for (uint32_t e = e0; e < e1; e++) {
				smPushEdge(g, L, tab, g.outAdj[e], newEnd, false);
				if (L.state == SM_RETIRE) return;
			}
		}
	}
	if (L.fresh || L.nPending == 0) {
		if (!L.fresh) {
			// ---- the slice is complete (...Banded.h:396-426, 560-607; the HMM: src/AlignmentCorrectnessEstimation.cpp:105-129)
			if (L.curCount == 0) { smRetire(L, EXT_ASSERT); return; }
			SmSlice cur;
			cur.first = L.curFirst; cur.count = L.curCount; cur.bandwidth = bandwidth; cur.j = L.j;
			cur.minScore = L.curMinScore; cur.minNode = L.curMinNode; cur.minOffset = L.curMinOffset;
			if (L.flatRows > 0) { cur.minScore = L.flatMin; cur.minNode = L.flatNode; cur.minOffset = L.flatOffset; }
			if (cur.minScore < L.prevMinScore) { smRetire(L, EXT_ASSERT); return; }
			const int mm = cur.minScore - L.prevMinScore;
			const int idx = mm < 64 ? mm : 63;
			const bool cfc = L.prevCorrect + ct.c2c >= L.prevFalse + ct.f2c;
			const bool ffc = L.prevCorrect + ct.c2f >= L.prevFalse + ct.f2f;
			const double a = L.prevCorrect + ct.c2c, b = L.prevFalse + ct.f2c;
			const double c = L.prevCorrect + ct.c2f, d = L.prevFalse + ct.f2f;
			const double curCorrect = (a > b ? a : b) + ct.correctOdds[idx];
			const double curFalse = (c > d ? c : d) + ct.wrongOdds[idx];
			cur.flags = (curCorrect > curFalse ? 1u : 0u) | (cfc ? 2u : 0u) | (ffc ? 4u : 0u);
			bool more = (cur.flags & 2u) != 0;   // !CorrectFromCorrect: stop, the slice is not kept (...Banded.h:589-607)
			if (more) {
				L.slices[L.nSlices++] = cur;
				L.prevMinScore = cur.minScore; L.prevBandwidth = cur.bandwidth; L.prevJ = cur.j; L.prevCorrect = curCorrect; L.prevFalse = curFalse;
				L.nPrev = cur.count; L.prevFirst = cur.first;
				L.pb = cb;
				L.slice++;
				if (L.slice >= L.numSlices) more = false;
			}
			if (!more) { smBacktraceBegin(L); return; }
		}
		L.fresh = 0;
		// ---- next slice: match masks, the previous slice's in-band nodes as sources (...Banded.h:235-277; linearizable is all-false)
		const int j = L.prevJ + 64;
		L.j = j;
		const EqSource src { L.masks, L.maskWords, L.startBit };
		eqVectorBits(src, L.len, j, L.eq);
		L.previousQuitScore = L.prevMinScore + L.prevBandwidth;
		L.flatRows = (j + 64 > L.len) ? (L.len - j) : 0;
		L.nPending = 0;
		for (uint32_t i = 0; i < L.nPrev; i++) {
			if (j != 0 && tab.tMin(L.pb, i) > L.previousQuitScore) continue;
			smPushEdge(g, L, tab, tab.tNode(L.pb, i), wsSource(tab.tStart(L.pb, i)), true);
			if (L.state == SM_RETIRE) return;
		}
		L.curFirst = L.nItems; L.curCount = 0;
		L.curMinScore = INT32_MAX - bandwidth - 1; L.curMinNode = 0xffffffffu; L.curMinOffset = 0xffffffffu;
		L.flatMin = INT32_MAX; L.flatNode = 0xffffffffu; L.flatOffset = 0xffffffffu;
		L.currentMin = L.curMinScore;
		if (L.nPending == 0) { smRetire(L, EXT_ASSERT); return; }
	}
	// ---- pop the pending node with the lowest topological rank (ComponentPriorityQueue order on a DAG) and set its tile up
	const uint32_t pbuf = L.pb, cbuf = L.pb ^ 1u;
	uint32_t best = 0;
	uint32_t bestComp = tab.qComp(0);
	for (uint32_t i = 1; i < L.nPending; i++) { const uint32_t c = tab.qComp(i); if (c < bestComp) { bestComp = c; best = i; } }
	const uint32_t pnode = tab.qNode(best);
	const WS pws = tab.qWs(best);
	if (best != L.nPending - 1) tab.qMove(best, L.nPending - 1);
	L.nPending--;
	if (L.nItems >= P.maxItems) { smRetire(L, EXT_OVERFLOW); return; }
	if (L.curCount >= SM_TABLE_CAP) { smRetire(L, EXT_SM_DECLINED); return; }
	(void)cbuf;
	const int pi = tab.find(pbuf, L.nPrev, pnode);
	const bool prevExists = pi >= 0;
	int32_t prevStart = 0; uint64_t prevHP = ~0ull, prevHN = 0ull;
	if (prevExists) { const NodeItem& p = L.items[L.prevFirst + (uint32_t)pi]; prevStart = p.sScore; prevHP = p.HP; prevHN = p.HN; }
	smTileSetup(g, L, pnode, pws, prevExists, prevStart, prevHP, prevHN, L.flatRows);
	NodeItem& out = L.items[L.nItems];
	out.node = pnode;
	out.sVP = L.VP; out.sVN = L.VN; out.sScore = L.colScore;
	L.tileActive = 1;
	L.btMode = 0;
	L.state = L.nodeLength > 1 ? SM_COL : SM_B;
}

// ---- SM_BT: the backtrace at a tile boundary (getReverseTraceFromTable, ...Common.h:392-544) -------------------------------
// node ids of the backtrace's current / previous slice live in the node words of table buffers 0 / 1
template <class T>
SM_FN void smFillIds(const SmLane& L, const SmTab<T>& tab, const SmSlice& sl, uint32_t b)
{
	for (uint32_t i = 0; i < sl.count; i++) tab.t.st(SM_TABLE_BASE + (b * SM_TABLE_CAP + i) * SM_TABLE_WORDS, L.items[sl.first + i].node);
}
template <class T>
SM_FN int smFindIn(const SmTab<T>& tab, const SmSlice& sl, uint32_t b, uint32_t node)
{
	const int i = tab.find(b, sl.count, node);
	return i < 0 ? -1 : (int)(sl.first + (uint32_t)i);
}

// corner rule (pickBacktraceCorner, ...Common.h:710-804)
template <class T>
SM_FN bool smCorner(const DGraph& g, const SmLane& L, const SmTab<T>& tab, uint32_t& outNode, uint32_t& outOffset, int32_t& outSeqPos, bool& nodeSwitch)
{
	const int32_t j = L.cs.j;
	const int32_t quitScore = L.cs.minScore + L.cs.bandwidth;
	const int32_t previousQuitScore = L.ps.minScore + L.ps.bandwidth;
	const WS start { L.cSVP, L.cSVN, L.cSScore };
	const int32_t scoreHere = smValue(start, 0);
	const uint32_t inBegin = g.inOff[L.curNode], inEnd = g.inOff[L.curNode + 1];
	if (scoreHere > quitScore) {
		int32_t smallest = scoreHere + 1;
		outNode = 0; outOffset = 0; outSeqPos = 0;
		nodeSwitch = false;
		if (L.pExists) { smallest = L.pSScore; outNode = L.curNode; outOffset = 0; outSeqPos = j - 1; }
		for (uint32_t e = inBegin; e < inEnd; e++) {
			const uint32_t nb = g.inAdj[e];
			const int p = smFindIn(tab, L.ps, 1, nb);
			if (p >= 0) { const int32_t es = L.items[p].eScore; if (es <= smallest) { smallest = es; outNode = nb; outOffset = (uint32_t)g.nodeLength[nb] - 1; outSeqPos = j - 1; nodeSwitch = true; } }
			const int c = smFindIn(tab, L.cs, 0, nb);
			if (c >= 0 && nb != L.curNode) {
				const NodeItem& ci = L.items[c];
				const int32_t v = smValue(WS { ci.eVP, ci.eVN, ci.eScore }, 0);
				if (v < smallest) { smallest = v; outNode = nb; outOffset = (uint32_t)g.nodeLength[nb] - 1; outSeqPos = j; nodeSwitch = true; }
			}
		}
		return true;
	}
	const NodeSeq nseq = loadNodeSeq(g, L.curNode);
	const int eqBit = (int)(eqOfColumn(L.eq, nseq, 0) & 1);
	if (L.pExists && L.pSScore == scoreHere - 1) { outNode = L.curNode; outOffset = 0; outSeqPos = j - 1; nodeSwitch = false; return true; }
	uint32_t biNode = 0xffffffffu, biOffset = 0xffffffffu; int32_t biSeqPos = -1;
	int32_t bestInvalidScore = scoreHere + 1;
	for (uint32_t e = inBegin; e < inEnd; e++) {
		const uint32_t nb = g.inAdj[e];
		const int c = smFindIn(tab, L.cs, 0, nb);
		if (c >= 0) {
			const NodeItem& ci = L.items[c];
			if (smValue(WS { ci.eVP, ci.eVN, ci.eScore }, 0) == scoreHere - 1) { outNode = nb; outOffset = (uint32_t)g.nodeLength[nb] - 1; outSeqPos = j; nodeSwitch = true; return true; }
		}
		const int p = smFindIn(tab, L.ps, 1, nb);
		if (p >= 0) {
			const int32_t cornerScore = L.items[p].eScore;
			if (cornerScore > previousQuitScore) {
				if (cornerScore < bestInvalidScore) { bestInvalidScore = cornerScore; biNode = nb; biOffset = (uint32_t)g.nodeLength[nb] - 1; biSeqPos = j - 1; }
			} else if (cornerScore == scoreHere - (eqBit ? 0 : 1)) {
				outNode = nb; outOffset = (uint32_t)g.nodeLength[nb] - 1; outSeqPos = j - 1; nodeSwitch = true; return true;
			}
		}
	}
	if (bestInvalidScore < scoreHere + 1) { outNode = biNode; outOffset = biOffset; outSeqPos = biSeqPos; nodeSwitch = true; return true; }
	return false;
}

template <class T>
SM_FN void smPhaseBt(const DGraph& g, const SmParams& P, SmLane& L, const T& t)
{
	const SmTab<T> tab { t };
	(void)P;
	if (L.hereSeqPos == -1) {
		// row -1: walk left along the initial ramp (...Common.h:508-542; the initial slice holds only the seed node)
		if (L.hereNode != L.startNode) { smRetire(L, EXT_ASSERT); return; }
		uint32_t off = L.hereOffset;
		while (true) {
			int32_t b = (int32_t)off - (int32_t)L.startOffset; if (b < 0) b = -b;
			int32_t bl = (int32_t)off - 1 - (int32_t)L.startOffset; if (bl < 0) bl = -bl;
			if (!(b != 0 && off > 0 && bl == b - 1)) break;
			off--;
			if (!smPushTrace(L, L.hereNode, off, -1, false)) return;
		}
		L.cnt.traceItems += L.nTrace;
		smRetire(L, EXT_OK);
		return;
	}
	const uint32_t s = (uint32_t)(L.hereSeqPos / 64) + 1;
	if (s >= L.nSlices) { smRetire(L, EXT_ASSERT); return; }
	if (s != L.curSliceIdx || L.hereNode != L.curNode) {
		if (s != L.curSliceIdx) {
			L.cs = L.slices[s]; L.ps = L.slices[s - 1];
			const EqSource src { L.masks, L.maskWords, L.startBit };
			eqVectorBits(src, L.len, L.cs.j, L.eq);
			smFillIds(L, tab, L.cs, 0);
			smFillIds(L, tab, L.ps, 1);
		}
		L.curSliceIdx = s;
		L.curNode = L.hereNode;
		const int ci = smFindIn(tab, L.cs, 0, L.curNode);
		if (ci < 0) { smRetire(L, EXT_ASSERT); return; }
		{ const NodeItem& it = L.items[ci]; L.cSVP = it.sVP; L.cSVN = it.sVN; L.cSScore = it.sScore; L.cEVP = it.eVP; L.cEVN = it.eVN; L.cEScore = it.eScore; }
		const int pi = smFindIn(tab, L.ps, 1, L.curNode);
		L.pExists = pi >= 0 ? 1u : 0u;
		L.pSScore = 0; L.pHPall = ~0ull; L.pHNall = 0ull;
		if (pi >= 0) { const NodeItem& it = L.items[pi]; L.pSScore = it.sScore; L.pHPall = it.HP; L.pHNall = it.HN; }
		// recompute the tile's columns (recalcNodeWordslice, ...Common.h:828-852)
		L.btMode = 1;
		smTileSetup(g, L, L.curNode, WS { L.cSVP, L.cSVN, L.cSScore }, L.pExists != 0, L.pSScore, L.pHPall, L.pHNall, 0);
		{ SmWalkCol c; c.up = L.VP; c.diag = 0; c.left = 0; c.row0 = L.colScore + smPopc(L.VN & ~1ull) - smPopc(L.VP & ~1ull); c.pad = 0; L.cols[0] = c; }
		L.cnt.recomputeTiles++; L.cnt.backtraceTiles++; L.cnt.columnSteps += (uint32_t)L.nodeLength;
		L.btVerify = 1;
		if (L.nodeLength > 1) { L.state = SM_COL; return; }
	}
	if (L.btVerify) {
		L.btVerify = 0;
		if (L.VP != L.cEVP || L.VN != L.cEVN || L.colScore != L.cEScore) L.status = EXT_ASSERT;   // sliceConsistency, ...Common.h:848-850
		if (L.status != EXT_OK) { smRetire(L, L.status); return; }
	}
	const int row = L.hereSeqPos & 63;
	const int32_t quitScore = L.cs.minScore + L.cs.bandwidth, previousQuitScore = L.ps.minScore + L.ps.bandwidth;
	if (row == 0 && L.hereOffset == 0) {
		uint32_t nn, no; int32_t nsp; bool sw;
		if (!smCorner(g, L, tab, nn, no, nsp, sw)) { smRetire(L, EXT_ASSERT); return; }
		if (!smPushTrace(L, nn, no, nsp, sw)) return;
		L.hereNode = nn; L.hereOffset = no; L.hereSeqPos = nsp;
		return;
	}
	if (row == 0) {
		// vertical crossing into the previous slice (...Common.h:451-477, pickBacktraceVerticalCrossing :665-708)
		if (!L.pExists) {
			L.hereOffset = 0;
			smPushTrace(L, L.curNode, 0, L.hereSeqPos, false);
			return;
		}
		uint32_t off = L.hereOffset;
		while (off > 0 && L.cols[off - 1].row0 == L.cols[off].row0 - 1) {
			off--;
			if (!smPushTrace(L, L.curNode, off, L.hereSeqPos, false)) return;
		}
		L.hereOffset = off;
		if (off == 0) {
			uint32_t nn, no; int32_t nsp; bool sw;
			if (!smCorner(g, L, tab, nn, no, nsp, sw)) { smRetire(L, EXT_ASSERT); return; }
			if (!smPushTrace(L, nn, no, nsp, sw)) return;
			L.hereNode = nn; L.hereOffset = no; L.hereSeqPos = nsp;
			return;
		}
		const int32_t scoreHere = L.cols[off].row0;
		int32_t scoreDiagonal = L.pSScore;
		const uint64_t lowMask = ((1ull << off) - 1) & ~1ull;
		scoreDiagonal += smPopc(L.pHPall & lowMask) - smPopc(L.pHNall & lowMask);
		const int32_t scoreUp = scoreDiagonal + (int32_t)((L.pHPall >> off) & 1) - (int32_t)((L.pHNall >> off) & 1);
		uint32_t no;
		if (scoreHere > quitScore || scoreDiagonal > previousQuitScore || scoreUp > previousQuitScore) {
			no = scoreDiagonal < scoreUp ? off - 1 : off;
		} else {
			const NodeSeq nseq = loadNodeSeq(g, L.curNode);
			const int eqBit = (int)(eqOfColumn(L.eq, nseq, (int)off) & 1);
			if (scoreUp == scoreHere - 1) no = off;
			else if (scoreDiagonal == scoreHere - (eqBit ? 0 : 1)) no = off - 1;
			else { smRetire(L, EXT_ASSERT); return; }
		}
		if (!smPushTrace(L, L.curNode, no, L.hereSeqPos - 1, false)) return;
		L.hereOffset = no; L.hereSeqPos = L.hereSeqPos - 1;
		return;
	}
	if (L.hereOffset == 0) {
		// horizontal crossing into an in-neighbour (...Common.h:478-499, pickBacktraceHorizontalCrossing :599-663)
		const WS start { L.cSVP, L.cSVN, L.cSScore };
		int32_t sp = L.hereSeqPos;
		while ((sp & 63) != 0 && (start.VP & (1ull << (sp & 63)))) {
			sp--;
			if (!smPushTrace(L, L.curNode, 0, sp, false)) return;
		}
		L.hereSeqPos = sp;
		const int offset = sp & 63;
		if (offset == 0) {
			uint32_t nn, no; int32_t nsp; bool sw;
			if (!smCorner(g, L, tab, nn, no, nsp, sw)) { smRetire(L, EXT_ASSERT); return; }
			if (!smPushTrace(L, nn, no, nsp, sw)) return;
			L.hereNode = nn; L.hereOffset = no; L.hereSeqPos = nsp;
			return;
		}
		const NodeSeq nseq = loadNodeSeq(g, L.curNode);
		const int eqBit = (int)((eqOfColumn(L.eq, nseq, 0) >> offset) & 1);
		const int32_t scoreHere = smValue(start, offset);
		uint32_t nn = 0, no = 0; int32_t nsp = 0;
		bool sw = false, found = false;
		const uint32_t inBegin = g.inOff[L.curNode], inEnd = g.inOff[L.curNode + 1];
		if (scoreHere > quitScore) {
			int32_t smallest = smValue(start, offset - 1);
			nn = L.curNode; no = 0; nsp = sp - 1;
			for (uint32_t e = inBegin; e < inEnd; e++) {
				const uint32_t nb = g.inAdj[e];
				const int c = smFindIn(tab, L.cs, 0, nb);
				if (c < 0) continue;
				const NodeItem& ci = L.items[c];
				const WS ne { ci.eVP, ci.eVN, ci.eScore };
				if (smValue(ne, offset - 1) <= smallest) { smallest = smValue(ne, offset - 1); nn = nb; no = (uint32_t)g.nodeLength[nb] - 1; nsp = sp - 1; sw = true; }
				if (smValue(ne, offset) < smallest && nb != L.curNode) { smallest = smValue(ne, offset); nn = nb; no = (uint32_t)g.nodeLength[nb] - 1; nsp = sp; sw = true; }
			}
			found = true;
		} else {
			for (uint32_t e = inBegin; e < inEnd && !found; e++) {
				const uint32_t nb = g.inAdj[e];
				const int c = smFindIn(tab, L.cs, 0, nb);
				if (c < 0) continue;
				const NodeItem& ci = L.items[c];
				const WS ne { ci.eVP, ci.eVN, ci.eScore };
				if (smValue(ne, offset) == scoreHere - 1) { nn = nb; no = (uint32_t)g.nodeLength[nb] - 1; nsp = sp; sw = true; found = true; }
				else if (smValue(ne, offset - 1) == scoreHere - (eqBit ? 0 : 1)) { nn = nb; no = (uint32_t)g.nodeLength[nb] - 1; nsp = sp - 1; sw = true; found = true; }
			}
		}
		if (!found) { smRetire(L, EXT_ASSERT); return; }
		if (!smPushTrace(L, nn, no, nsp, sw)) return;
		L.hereNode = nn; L.hereOffset = no; L.hereSeqPos = nsp;
		return;
	}
	// inside the tile (pickBacktraceInside, ...Common.h:556-597): the walk runs in SM_WALK on the column's masks
	L.hori = L.hereOffset;
	L.vert = row;
	{ const SmWalkCol c = L.cols[L.hori]; L.up = c.up; L.diag = c.diag; L.left = c.left; }
	L.unfit = 0;
	L.state = SM_WALK;
}

// ---- SM_WALK: one cell inside the tile: vertical, then diagonal, then horizontal --------------------------------------------
SM_FN void smPhaseWalk(SmLane& L)
{
	const int vert = L.vert;
	const uint32_t u = (uint32_t)(L.up >> vert) & 1u, d = (uint32_t)(L.diag >> vert) & 1u, l = (uint32_t)(L.left >> vert) & 1u;
	L.unfit |= (u | d | l) ^ 1u;
	L.vert = vert - (int)(u | d);
	if (!u) {
		L.hori--;
		const SmWalkCol c = L.cols[L.hori];
		L.up = c.up; L.diag = c.diag; L.left = c.left;
	}
	if (!smPushTrace(L, L.curNode, L.hori, L.cs.j + L.vert, false)) return;
	if (!(L.hori > 0 && L.vert > 0)) {
		if (L.unfit) { smRetire(L, EXT_ASSERT); return; }
		L.hereNode = L.curNode; L.hereOffset = L.hori; L.hereSeqPos = L.cs.j + L.vert;
		L.state = SM_BT;
	}
}

} // namespace gcsm
