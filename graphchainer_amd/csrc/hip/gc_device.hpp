// Device-side data layout and the banded bit-vector seed-extension core for gfx950 (MI355X).
//
// Mapping: one LANE per seed extension. The inner loop of the reference is a chain of
// up to 63 dependent 64-bit Myers column steps per (node, 64-row slice) tile
// (reference: src/GraphAlignerBitvectorCommon.h:1118-1161): rows are already packed in the 64-bit word, columns
// are serial, slices are serial and the band of the next slice depends on this slice's minimum, so the
// parallel axis is the pool of independent extensions (fragments x seeds x 2 directions x reads).
// Integer/bitwise only: no MFMA.
//
// All graph arrays live in HBM (uploaded once); per-lane scratch is a slab of HBM indexed by the lane's
// global id so a persistent grid can stride over any number of work items.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// GC_LOOP_TICK(i): a hook of the host-side schedule simulator (tests/frag_host/frag_wave_sim.cpp), which counts the turns of the data-dependent loops; nothing in the product
#ifndef GC_LOOP_TICK
#define GC_LOOP_TICK(i) ((void)0)
#endif

namespace gcdev {

// Everything the fragment extension kernel asks about a split node, in one aligned 32-byte load (r6): the node's arrays lie in five places, i.e. five cache lines per
// node a seed's chance hit lands on; the records of neighbouring nodes (ids a few apart) share a line.
struct NodeRec {
	uint32_t comp;     // componentNumber
	uint32_t outOff;   // first out-edge in outAdj
	uint32_t inOff;    // first in-edge in inAdj
	uint32_t meta;     // bits 0-6 length (1..64), bit 7 NODEREC_SLOW (an ambiguous node: its letters are not in w0 / w1), bits 8-15 out-degree, bits 16-23 in-degree (255: or more)
	uint64_t w0, w1;   // nodeSeq[2 * node], [2 * node + 1]
};
enum : uint32_t { NODEREC_SLOW = 128u };

struct DGraph {
	uint32_t nNodes;
	uint32_t firstAmbiguous;
	const uint8_t*  nodeLength;       // [n]
	const uint32_t* nodeOffset;       // [n] offset inside the original (bigraph) node
	const int32_t*  nodeIDs;          // [n] bigraph node id
	const uint64_t* nodeSeq;          // [2*firstAmbiguous]  2 bits / bp
	const uint64_t* ambSeq;           // [4*(n-firstAmbiguous)]  A,C,G,T one-hot words
	const uint32_t* inOff;  const uint32_t* inAdj;     // CSR, reference neighbour order
	const uint32_t* outOff; const uint32_t* outAdj;
	const uint32_t* componentNumber;  // topological rank
	// reverse-strand twin lookup (GetReversePosition + GetUnitigNode, src/AlignmentGraph.cpp:832-868)
	const uint32_t* origSize;         // [nBigraph] length of the original node
	const uint32_t* lookupOff;        // [nBigraph+1] into lookup
	const uint32_t* lookup;           // split nodes of each bigraph node in offset order
	// MPC index, flattened over components
	const uint32_t* componentMap;     // [n] weakly connected component id
	const uint32_t* topoId;           // [n] topological position inside its component
	const uint32_t* pathsOff; const uint32_t* paths; const uint32_t* pathsPos;   // [n+1]; path ids through node (ascending) and the node's position on each
	const uint32_t* backOff;  const uint32_t* backNode; const uint32_t* backPath; const uint32_t* backPos;   // [n+1]; (last node of path k strictly reaching v, k, its position on k)
	const uint32_t* mpcWidth;         // [nComponents]
	// seed clustering (orderSeedsByChaining, src/GraphAligner.h:245-262): chain of every split node and its approximate position on it
	const uint32_t* chainNumber;      // [n]
	const uint64_t* chainApproxPos;   // [n]
	const NodeRec* nodeRec;           // [n] (r6) the per-node record of the fragment extension kernel
};

struct WS {   // one DP column over 64 read rows (reference: src/WordSlice.h:150-166)
	uint64_t VP, VN;
	int32_t score;   // score of row 63
};

__device__ __forceinline__ int popc64(uint64_t x) { return __popcll(x); }
__device__ __forceinline__ int32_t wsBefore(const WS& w) { return w.score - popc64(w.VP) + popc64(w.VN); }
__device__ __forceinline__ int32_t wsValue(const WS& w, int row)   // reference: src/WordSlice.h:177-186
{
	uint64_t above = row < 63 ? (~0ull << (row + 1)) : 0ull;
	return w.score + popc64(w.VN & above) - popc64(w.VP & above);
}

// Pointwise minimum of two columns (reference: src/WordSlice.h:491-530). Walks only rows whose deltas differ.
__device__ inline WS wsMerge(const WS& a, const WS& b)
{
	int32_t d = wsBefore(a) - wsBefore(b);
	uint64_t takeB = 0, fixP = 0, fixN = 0, fixMask = 0;
	uint64_t diff = (a.VP ^ b.VP) | (a.VN ^ b.VN);
	int pos = 0;
	while (diff) {
		GC_LOOP_TICK(0);
		int r = __ffsll((long long)diff) - 1;
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

// min over rows -1..63 of a column: what changedMinScore (src/WordSlice.h:252) returns against an absent
// old column ({0,0,INT_MAX}), the only case that occurs when every node is computed once per slice (DAG).
__device__ inline int32_t wsColumnMin(const WS& w)
{
	int32_t before = wsBefore(w);
	int32_t best = before;
	uint64_t vn = w.VN;
	while (vn) {
		GC_LOOP_TICK(1);
		int r = __ffsll((long long)vn) - 1;
		// skip to the end of this run of VN bits: the value is lowest there
		uint64_t run = vn & ~(vn + (1ull << r));   // the contiguous run starting at r
		int last = 63 - __clzll((long long)run);
		uint64_t upto = last < 63 ? ((1ull << (last + 1)) - 1) : ~0ull;
		int32_t v = before + popc64(w.VP & upto) - popc64(w.VN & upto);
		best = v < best ? v : best;
		vn &= ~upto;
	}
	return best;
}

// Myers column step with carries (reference: src/GraphAlignerBitvectorCommon.h:243-263).
__device__ __forceinline__ WS myersStep(uint64_t Eq, WS s, uint64_t hinP, uint64_t hinN, uint64_t& houtP, uint64_t& houtN)
{
	uint64_t Xv = Eq | s.VN;
	Eq |= hinN;
	uint64_t Xh = (((Eq & s.VP) + s.VP) ^ s.VP) | Eq;
	uint64_t Ph = s.VN | ~(Xh | s.VP);
	uint64_t Mh = s.VP & Xh;
	uint64_t sMh = (Mh << 1) | hinN;
	uint64_t sPh = (Ph << 1) | hinP;
	houtN = Mh >> 63;
	houtP = Ph >> 63;
	WS r;
	r.VP = sMh | ~(Xv | sPh);
	r.VN = sPh & Xv;
	r.score = s.score - (int32_t)houtN + (int32_t)houtP;
	return r;
}

__device__ __forceinline__ WS wsSource(int32_t previousScore) { return WS { ~0ull, 0ull, previousScore + 64 }; }   // ...Common.h:806-810

// ---- per-extension state kept in the lane's HBM slab ------------------------------------------------

struct NodeItem {   // reference: src/NodeSlice.h:15-47 (80 B there, 64 B here)
	uint64_t sVP, sVN, eVP, eVN, HP, HN;
	int32_t sScore, eScore, minScore;
	uint32_t node;
};
__device__ __forceinline__ WS itemStart(const NodeItem& it) { return WS { it.sVP, it.sVN, it.sScore }; }
__device__ __forceinline__ WS itemEnd(const NodeItem& it) { return WS { it.eVP, it.eVN, it.eScore }; }

struct SliceInfo {  // reference: DPSlice, src/GraphAlignerBitvectorCommon.h:138-214
	int32_t minScore;
	uint32_t minNode, minOffset;
	uint32_t first, count;      // this slice's NodeItems are items[first .. first+count)
	int32_t bandwidth;
	int32_t j;                  // first read row of the slice (-64 for the initial slice)
	uint32_t flags;             // bit0 currentlyCorrect, bit1 correctFromCorrect, bit2 falseFromCorrect
	double correctLogOdds, falseLogOdds;
};

struct Pending {    // a node scheduled in the current slice with its incoming columns already folded
	uint64_t VP, VN;
	int32_t score;
	uint32_t node;
	uint32_t comp;
	uint32_t hasIncoming;
};

struct TraceCell {  // one cell of a backtrace, split-node coordinates
	uint32_t node;
	int32_t seqPos;             // -1 for the row above the first slice
	uint32_t offsetAndSwitch;   // offset in split node | nodeSwitch << 8
};

// A cell of the fragment pass's shared trace pool (r6): 8 bytes instead of the cores' 12-byte working format - the pool is the largest buffer of a batch in flight (18 GB of
// 42 for 2 000 x 50 kb reads on a 960 Mbp graph), every cell is written once by an extension kernel and read by k_build_anchors. seqPos is a row of a fragment's extension:
// -1 .. --colinear-split-len - 1 (gc_align_batch takes split lengths of 16 .. 64: one slice per fragment extension).
struct PoolCell {
	uint32_t node;
	uint16_t offsetAndSwitch;   // offset in split node | nodeSwitch << 8
	int16_t seqPos;
};

struct CorrectnessTables {   // reference: src/AlignmentCorrectnessEstimation.cpp:15-70, built on the host with libm
	double correctOdds[64], wrongOdds[64];
	double f2c, f2f, c2f, c2c;
	double initCorrect, initFalse;
};

struct ExtendConfig {
	int32_t bandwidth;
	uint32_t maxItems;      // NodeItem capacity of a lane's slab
	uint32_t maxSlices;     // SliceInfo capacity (>= numSlices+1)
	uint32_t maxPending;
	uint32_t maxTrace;
	uint32_t maxCols = 0;   // whole-read pass, one extension per wave: columns of the DP kept for the backtrace (0: the backtrace recomputes its tiles)
	uint32_t regCap = 64;   // whole-read pass, register tables: nodes per slice before the extension is retried with the LDS/HBM tables (test hook, <= 64)
};

// status codes of one extension
enum : uint32_t { EXT_OK = 0, EXT_FAILED = 1, EXT_ASSERT = 2, EXT_OVERFLOW = 3 };
enum : uint32_t { EXT_NOT_RUN = 7 };   // fragment pass with lazy extension: the work item has not been run (yet)
enum : uint32_t { EXT_LDS_CAP = 5 };   // whole-read pass: a slice has more nodes than the wave tables hold (retried with larger tables, then the plain layout)

// One recomputed column of the backtrace's current tile: the vertical deltas only (16 B; r4: was the whole WS, 24 B). Its score - the value of row 63 - follows from the tile's
// start score and the bottom row's horizontal deltas, which the recompute leaves in registers (ColumnScores): 4.4 M fragment extensions x 2-3 tiles x up to 64 columns go through
// the per-lane slab twice (written by the recompute, read by the walk), the largest part of what k_extend moves beyond its algorithmic bytes.
struct WCol { uint64_t VP, VN; };
struct ColumnScores {
	int32_t start; uint64_t HP, HN;   // score of column 0; bit c: the bottom row steps +1 / -1 from column c - 1 to c
	__device__ __forceinline__ int32_t at(uint32_t c) const { const uint64_t m = (c >= 63 ? ~0ull : ((2ull << c) - 1)) & ~1ull; return start + popc64(HP & m) - popc64(HN & m); }
};
struct LaneScratch {
	SliceInfo* slices;
	NodeItem* items;
	Pending* pending;
	WCol* columns;      // one node's recomputed columns for the backtrace: column c at columns[(c & colMask) * colStride]
	uint32_t colMask;   // 63: the whole tile (per-lane slab in HBM). Smaller (r4, k_extend): a ring of the last colMask + 1 columns in LDS - the walk only moves left, and when it leaves
	uint32_t colStride; // the ring the tile is recomputed up to the column it stands on
	TraceCell* trace;
	uint32_t* itemNodes; // the items' node ids again, packed (r4): "is this node in that slice" scans 4 B per item instead of pulling a 64 B item record per probe
};

struct ExtCounters {
	// 32-bit: they are per lane (per wave in the whole-read kernel) and flushed with one 64-bit atomic each at the kernel's end; in the
	// whole-read kernel they live in scalar registers for the kernel's whole life, where every pair of them is one more spill
	uint32_t dpTiles, recomputeTiles, columnSteps, traceItems, extensions, backtraceTiles;
	// r5, per extension (set by the core, read by the kernel right after it): 1 when the extension produced a trace whose backtrace started in the last partial slice and that slice's
	// minimum (the fused flattenLastSliceEnd) was attained in more than one node - the one place where the reference's parallel-hashmap iteration order, which this build replaces by
	// band-entry order, can pick another cell (gc_result::flatten_ties)
	uint32_t flattenTie;
#ifdef GC_STAMPS
	unsigned long long cyc[16], tMark;   // profiling build only (make stamps): lane-cycles per section of extendSeedWave
#endif
};
// GC_MARK(i): everything since the previous mark is charged to bucket i. Compiles to nothing in the product build.
#ifdef GC_STAMPS
#define GC_MARK_START() (cnt.tMark = clock64())
#define GC_MARK(i) do { unsigned long long now_ = clock64(); cnt.cyc[i] += now_ - cnt.tMark; cnt.tMark = now_; } while (0)
#else
#define GC_MARK_START() ((void)0)
#define GC_MARK(i) ((void)0)
#endif

// 4 match masks (A,C,G,T) of read rows j..j+63. reference: ...Common.h:280-319. iupac[c] = set of bases c matches.
// Four named members, never an array: an array indexed by the node's 2-bit base code ends up in scratch (or LDS),
// and the Myers column chain would then wait on a memory load at every step.
struct Eq4 { uint64_t a, c, g, t; };
__device__ inline void eqVector(const char* seq, int len, int j, const uint8_t* iupac, Eq4& eq)
{
	uint64_t a = 0, c = 0, g = 0, t = 0;
	int n = len - j;
	if (n > 64) n = 64;
	for (int i = 0; i < n; i++) {
		uint32_t m = iupac[(uint8_t)seq[j + i]];
		uint64_t bit = 1ull << i;
		a |= (m & 1) ? bit : 0ull;
		c |= (m & 2) ? bit : 0ull;
		g |= (m & 4) ? bit : 0ull;
		t |= (m & 8) ? bit : 0ull;
	}
	eq.a = a; eq.c = c; eq.g = g; eq.t = t;
}

// Match masks from per-read precomputed bit vectors: for each strand of each read four bit vectors (A,C,G,T) say which
// bases a read position matches; the 64 rows of a slice are a 64-bit window of them (two words + a funnel shift)
// instead of 64 dependent byte loads per slice.
struct EqSource {
	const uint64_t* masks;   // [4][words] for this read and strand
	uint32_t words;          // words per bit vector
	uint32_t startBit;       // read position of the extension's row 0
	__device__ __forceinline__ void rows(int len, int j, Eq4& eq) const;
};
__device__ inline void eqVectorBits(const EqSource& src, int len, int j, Eq4& eq)
{
	int n = len - j;
	if (n > 64) n = 64;
	uint32_t bit = src.startBit + (uint32_t)j;
	uint32_t w = bit >> 6, sh = bit & 63;
	uint64_t keep = n >= 64 ? ~0ull : ((1ull << n) - 1);
	uint64_t out[4];
	for (int b = 0; b < 4; b++) {
		const uint64_t* v = src.masks + (size_t)b * src.words;
		uint64_t lo = v[w] >> sh;
		uint64_t hi = (sh != 0 && w + 1 < src.words) ? (v[w + 1] << (64 - sh)) : 0ull;
		out[b] = (lo | hi) & keep;
	}
	eq.a = out[0]; eq.c = out[1]; eq.g = out[2]; eq.t = out[3];
}
__device__ __forceinline__ void EqSource::rows(int len, int j, Eq4& eq) const { eqVectorBits(*this, len, j, eq); }

struct NodeSeq { uint64_t w0, w1, w2, w3; bool ambiguous; };
__device__ __forceinline__ NodeSeq loadNodeSeq(const DGraph& g, uint32_t node)
{
	NodeSeq s;
	if (node < g.firstAmbiguous) {
		s.w0 = g.nodeSeq[2 * (size_t)node];
		s.w1 = g.nodeSeq[2 * (size_t)node + 1];
		s.w2 = s.w3 = 0;
		s.ambiguous = false;
	} else {
		const uint64_t* p = g.ambSeq + 4 * (size_t)(node - g.firstAmbiguous);
		s.w0 = p[0]; s.w1 = p[1]; s.w2 = p[2]; s.w3 = p[3];
		s.ambiguous = true;
	}
	return s;
}
__device__ __forceinline__ uint64_t eqOfColumn(const Eq4& eq, const NodeSeq& s, int pos)
{
	if (!s.ambiguous) {
		uint64_t w = pos < 32 ? s.w0 : s.w1;
		uint32_t code = (uint32_t)(w >> ((pos & 31) * 2));
		uint64_t lo = (code & 1) ? eq.c : eq.a;   // codes 0/1
		uint64_t hi = (code & 1) ? eq.t : eq.g;   // codes 2/3
		return (code & 2) ? hi : lo;
	}
	uint64_t r = 0;
	r |= ((s.w0 >> pos) & 1) ? eq.a : 0ull;
	r |= ((s.w1 >> pos) & 1) ? eq.c : 0ull;
	r |= ((s.w2 >> pos) & 1) ? eq.g : 0ull;
	r |= ((s.w3 >> pos) & 1) ? eq.t : 0ull;
	return r;
}

__device__ inline int findItem(const uint32_t* itemNodes, const SliceInfo& sl, uint32_t node)
{
	for (uint32_t i = 0; i < sl.count; i++)
		if (itemNodes[sl.first + i] == node) return (int)(sl.first + i);
	return -1;
}

// One (node, slice) tile: first column = `ws` (already merged over all incoming edges), then up to 63 Myers
// steps. reference: src/GraphAlignerBitvectorCommon.h:1052-1167 for a node that is new in this slice.
// If `columns` != nullptr every column is stored (backtrace recompute). If flatRows > 0 the minimum over
// columns of the score at row flatRows-1 is tracked (fused flattenLastSliceEnd, ...Common.h:1210-1218).
struct TileResult { int32_t minScore; uint32_t minOffset; int32_t flatMin; uint32_t flatOffset; };
__device__ inline TileResult computeTile(const DGraph& g, uint32_t node, WS ws, bool prevExists, int32_t prevStartScore, uint64_t prevHP, uint64_t prevHN,
	const Eq4& eq, NodeItem& out, WCol* columns, int flatRows, uint32_t& status, uint32_t colMask = 63, uint32_t colStride = 1, int lastColumn = 63)
{
	int nodeLength = g.nodeLength[node];
	if (lastColumn + 1 < nodeLength) nodeLength = lastColumn + 1;   // (backtrace ring refill: the columns up to the one the walk stands on; `out`'s end column is then that column)
	NodeSeq seq = loadNodeSeq(g, node);
	TileResult r;
	r.minScore = ws.score;   // (sic) taken before the merge with the row above, ...Common.h:968 vs :1052-1058
	r.minOffset = 0;
	if (prevExists && wsBefore(ws) > prevStartScore) ws = wsMerge(ws, wsSource(prevStartScore));
	int forceUntil = 0;
	if (prevExists) {
		int32_t scoreBefore = wsBefore(ws);
		int32_t scoreComparison = prevStartScore;
		if (scoreBefore > scoreComparison) status = EXT_ASSERT;
		if (scoreBefore < scoreComparison) {
			for (int fix = 1; fix < 64; fix++) {
				int32_t next = scoreComparison + (int32_t)((prevHP >> fix) & 1) - (int32_t)((prevHN >> fix) & 1);
				uint64_t mask = 1ull << fix;
				if (scoreBefore > next) status = EXT_ASSERT;
				if (scoreBefore < next) { prevHP |= mask; prevHN &= ~mask; forceUntil = fix; }
				if (scoreBefore == next) { prevHP &= ~mask; prevHN &= ~mask; }
				scoreBefore++;
				scoreComparison = next;
				if (scoreBefore >= scoreComparison) break;
			}
		}
	} else {
		forceUntil = nodeLength;
	}
	out.node = node;
	out.sVP = ws.VP; out.sVN = ws.VN; out.sScore = ws.score;
	uint64_t flatMask = flatRows > 0 ? ~(~0ull << flatRows) : 0;
	r.flatMin = INT32_MAX;
	r.flatOffset = 0;
	if (flatRows > 0) r.flatMin = ws.score - popc64(ws.VP & ~flatMask) + popc64(ws.VN & ~flatMask);
	if (columns) columns[0] = WCol { ws.VP, ws.VN };   // ((0 & colMask) * colStride)
	uint64_t forceEq = prevExists ? ~0ull : ~1ull;
	uint64_t HP = 0, HN = 0;
	for (int pos = 1; pos < nodeLength; pos++) {
		uint64_t Eq = eqOfColumn(eq, seq, pos) & forceEq;
		uint64_t hp, hn;
		ws = myersStep(Eq, ws, (prevHP >> pos) & 1, (prevHN >> pos) & 1, hp, hn);
		if (forceUntil >= pos) { ws.VP &= ~1ull; ws.VN |= 1ull; }
		if (ws.score < r.minScore) { r.minScore = ws.score; r.minOffset = (uint32_t)pos; }
		if (flatRows > 0) {
			int32_t f = ws.score - popc64(ws.VP & ~flatMask) + popc64(ws.VN & ~flatMask);
			if (f < r.flatMin) { r.flatMin = f; r.flatOffset = (uint32_t)pos; }
		}
		if (columns) columns[((uint32_t)pos & colMask) * colStride] = WCol { ws.VP, ws.VN };
		HP |= hp << pos;
		HN |= hn << pos;
	}
	out.HP = HP; out.HN = HN;
	out.eVP = ws.VP; out.eVN = ws.VN; out.eScore = ws.score;
	return r;
}

// Folds one incoming edge into a pending node (creating it if needed).
// reference: the per-edge part of calculateNodeInner, ...Common.h:903-964; edges coming from the previous slice
// ("skipFirst") are merged as they are, edges from an in-neighbour are first stepped into the node's column 0.
__device__ inline void pushEdge(const DGraph& g, Pending* pending, uint32_t& nPending, const ExtendConfig& cfg, uint32_t target, WS incoming, bool skipFirst,
	const NodeItem* items, const uint32_t* itemNodes, const SliceInfo& prevSlice, const Eq4& eq, uint32_t& status)
{
	uint32_t slot = nPending;
	for (uint32_t i = 0; i < nPending; i++)
		if (pending[i].node == target) { slot = i; break; }
	WS add = incoming;
	if (!skipFirst) {
		int prevIdx = findItem(itemNodes, prevSlice, target);
		uint64_t hinP, hinN;
		bool prevExists = prevIdx >= 0;
		int32_t prevStart = prevExists ? items[prevIdx].sScore : 0;
		if (prevExists) {
			int32_t before = wsBefore(incoming);
			if (prevStart < before) { hinP = 0; hinN = 1; }
			else if (prevStart > before) { hinP = 1; hinN = 0; }
			else { hinP = 0; hinN = 0; }
		} else { hinP = 1; hinN = 0; }
		NodeSeq seq = loadNodeSeq(g, target);
		uint64_t hp, hn;
		add = myersStep(eqOfColumn(eq, seq, 0), incoming, hinP, hinN, hp, hn);
		if (!prevExists || wsBefore(add) < prevStart) { add.VP &= ~1ull; add.VN |= 1ull; }
	}
	if (slot == nPending) {
		if (nPending >= cfg.maxPending) { status = EXT_OVERFLOW; return; }
		pending[slot].node = target;
		pending[slot].comp = g.componentNumber[target];
		pending[slot].VP = add.VP; pending[slot].VN = add.VN; pending[slot].score = add.score;
		pending[slot].hasIncoming = 1;
		nPending++;
	} else {
		WS cur { pending[slot].VP, pending[slot].VN, pending[slot].score };
		WS m = wsMerge(cur, add);
		pending[slot].VP = m.VP; pending[slot].VN = m.VN; pending[slot].score = m.score;
	}
}

// Recomputes all columns of (slice s, node) into sc.columns. reference: recalcNodeWordslice, ...Common.h:828-852
// `upTo`: the last column wanted (a ring keeps the colMask + 1 columns that end there); `entering`: the walk's first visit of the tile - counted, in the reference's units (it recomputes
// the whole tile once) - as opposed to a refill of the ring
__device__ inline ColumnScores recomputeColumns(const DGraph& g, const LaneScratch& sc, uint32_t s, int itemIdx, const Eq4& eq, uint32_t& status, ExtCounters& cnt, uint32_t upTo = 63, bool entering = true)
{
	const NodeItem& it = sc.items[itemIdx];
	int prevIdx = findItem(sc.itemNodes, sc.slices[s - 1], it.node);
	bool prevExists = prevIdx >= 0;
	NodeItem scratch;
	const int nodeLength = g.nodeLength[it.node];
	computeTile(g, it.node, itemStart(it), prevExists, prevExists ? sc.items[prevIdx].sScore : 0,
		prevExists ? sc.items[prevIdx].HP : ~0ull, prevExists ? sc.items[prevIdx].HN : 0ull, eq, scratch, sc.columns, 0, status, sc.colMask, sc.colStride, (int)upTo);
	if ((int)upTo + 1 >= nodeLength && (scratch.eVP != it.eVP || scratch.eVN != it.eVN || scratch.eScore != it.eScore)) status = EXT_ASSERT;   // sliceConsistency, :848-850 (whole tiles only)
	if (entering) {
		cnt.recomputeTiles++;
		cnt.backtraceTiles++;
		cnt.columnSteps += nodeLength;
	}
	return ColumnScores { scratch.sScore, scratch.HP, scratch.HN };
}

struct Cell { uint32_t node; uint32_t offset; int32_t seqPos; };

__device__ inline bool pushTrace(const LaneScratch& sc, const ExtendConfig& cfg, uint32_t& nTrace, Cell c, bool nodeSwitch, uint32_t& status)
{
	if (nTrace >= cfg.maxTrace) { status = EXT_OVERFLOW; return false; }
	sc.trace[nTrace].node = c.node;
	sc.trace[nTrace].seqPos = c.seqPos;
	sc.trace[nTrace].offsetAndSwitch = c.offset | (nodeSwitch ? 256u : 0u);
	nTrace++;
	return true;
}

// reference: pickBacktraceCorner, ...Common.h:710-804 (scoresNotValid is never set: unlimited cells per slice)
__device__ inline bool backtraceCorner(const DGraph& g, const LaneScratch& sc, uint32_t s, uint32_t node, int itemIdx, const Eq4& eq, Cell& out, bool& nodeSwitch)
{
	const SliceInfo& cur = sc.slices[s];
	const SliceInfo& prev = sc.slices[s - 1];
	int32_t j = cur.j;
	int32_t quitScore = cur.minScore + cur.bandwidth;
	int32_t previousQuitScore = prev.minScore + prev.bandwidth;
	int32_t scoreHere = wsValue(itemStart(sc.items[itemIdx]), 0);
	int prevSelf = findItem(sc.itemNodes, prev, node);
	uint32_t inBegin = g.inOff[node], inEnd = g.inOff[node + 1];
	if (scoreHere > quitScore) {
		int32_t smallest = scoreHere + 1;
		out = Cell { 0, 0, 0 };
		nodeSwitch = false;
		if (prevSelf >= 0) { smallest = sc.items[prevSelf].sScore; out = Cell { node, 0, j - 1 }; }
		for (uint32_t e = inBegin; e < inEnd; e++) {
			uint32_t nb = g.inAdj[e];
			int p = findItem(sc.itemNodes, prev, nb);
			if (p >= 0 && sc.items[p].eScore <= smallest) { smallest = sc.items[p].eScore; out = Cell { nb, (uint32_t)g.nodeLength[nb] - 1, j - 1 }; nodeSwitch = true; }
			int c = findItem(sc.itemNodes, cur, nb);
			if (c >= 0 && nb != node) {
				int32_t v = wsValue(itemEnd(sc.items[c]), 0);
				if (v < smallest) { smallest = v; out = Cell { nb, (uint32_t)g.nodeLength[nb] - 1, j }; nodeSwitch = true; }
			}
		}
		return true;
	}
	NodeSeq seq = loadNodeSeq(g, node);
	int eqBit = (int)(eqOfColumn(eq, seq, 0) & 1);   // sequence[j] vs first base of the node
	if (prevSelf >= 0 && sc.items[prevSelf].sScore == scoreHere - 1) { out = Cell { node, 0, j - 1 }; nodeSwitch = false; return true; }
	Cell bestInvalid { 0xffffffffu, 0xffffffffu, -1 };
	int32_t bestInvalidScore = scoreHere + 1;
	for (uint32_t e = inBegin; e < inEnd; e++) {
		uint32_t nb = g.inAdj[e];
		int c = findItem(sc.itemNodes, cur, nb);
		if (c >= 0 && wsValue(itemEnd(sc.items[c]), 0) == scoreHere - 1) { out = Cell { nb, (uint32_t)g.nodeLength[nb] - 1, j }; nodeSwitch = true; return true; }
		int p = findItem(sc.itemNodes, prev, nb);
		if (p >= 0) {
			int32_t corner = sc.items[p].eScore;
			if (corner > previousQuitScore) {
				if (corner < bestInvalidScore) { bestInvalidScore = corner; bestInvalid = Cell { nb, (uint32_t)g.nodeLength[nb] - 1, j - 1 }; }
			} else if (corner == scoreHere - (eqBit ? 0 : 1)) {
				out = Cell { nb, (uint32_t)g.nodeLength[nb] - 1, j - 1 }; nodeSwitch = true; return true;
			}
		}
	}
	if (bestInvalidScore < scoreHere + 1) { out = bestInvalid; nodeSwitch = true; return true; }
	return false;   // the reference asserts here
}

// Full seed extension: slices, correctness HMM, trimming, backtrace.
// reference: getReverseTraceFromSeed, src/GraphAlignerBitvectorBanded.h:46-71. Returns status; on EXT_OK the
// trace (start cell first, row -1 last) is in sc.trace[0..nTrace) and `score` is the alignment score.
// The match masks of a slice come from `eqs` (EqFromBases: the read's letters; EqSource: the per-read bit vectors built at upload).
struct EqFromBases {
	const char* seq; const uint8_t* iupac;
	__device__ __forceinline__ void rows(int len, int j, Eq4& eq) const { eqVector(seq, len, j, iupac, eq); }
};
template <class EQS>
__device__ inline uint32_t extendSeedT(const DGraph& g, const CorrectnessTables& ct, const EQS& eqs, const ExtendConfig& cfg, const LaneScratch& sc,
	int len, uint32_t startNode, uint32_t startOffset, uint32_t& nTrace, int32_t& score, ExtCounters& cnt)
{
	uint32_t status = EXT_OK;
	nTrace = 0;
	score = 0;
	cnt.extensions++;
	cnt.flattenTie = 0;
	int numSlices = (len + 63) / 64;
	if ((uint32_t)numSlices + 1 > cfg.maxSlices) return EXT_OVERFLOW;
	// ---- initial slice: row -1 scores are |column - startOffset| on the seed's split node (...Common.h:1243-1279)
	{
		int nl = g.nodeLength[startNode];
		NodeItem& it = sc.items[0];
		it.node = startNode;
		sc.itemNodes[0] = startNode;
		it.sVP = it.sVN = it.eVP = it.eVN = 0;
		it.sScore = (int32_t)startOffset;
		it.eScore = nl - 1 - (int32_t)startOffset;
		it.minScore = 0;
		uint64_t upToOffset = startOffset >= 63 ? ~0ull : ((1ull << (startOffset + 1)) - 1);
		uint64_t nodeMask = nl >= 64 ? ~0ull : ((1ull << nl) - 1);
		it.HN = upToOffset & ~1ull;
		it.HP = nodeMask & ~upToOffset;
		SliceInfo& s0 = sc.slices[0];
		s0.minScore = 0; s0.minNode = startNode; s0.minOffset = startOffset;
		s0.first = 0; s0.count = 1; s0.bandwidth = 1; s0.j = -64;
		s0.correctLogOdds = ct.initCorrect; s0.falseLogOdds = ct.initFalse;
		s0.flags = 1;   // log(0.8) > log(0.2)
	}
	uint32_t nItems = 1;
	uint32_t nSlices = 1;
	Eq4 eq;
	for (int slice = 0; slice < numSlices; slice++) {
		const SliceInfo prev = sc.slices[nSlices - 1];
		int j = prev.j + 64;
		eqs.rows(len, j, eq);
		int32_t previousQuitScore = prev.minScore + prev.bandwidth;
		int32_t previousMinScore = prev.minScore;
		int bandwidth = cfg.bandwidth;
		int flatRows = (j + 64 > len) ? (len - j) : 0;   // last partial slice (...Banded.h:414)
		// seed the queue from the previous slice's in-band nodes (...Banded.h:235-277; linearizable is all-false)
		uint32_t nPending = 0;
		for (uint32_t i = 0; i < prev.count; i++) {
			const NodeItem& it = sc.items[prev.first + i];
			if (j != 0 && it.minScore > previousQuitScore) continue;
			pushEdge(g, sc.pending, nPending, cfg, it.node, wsSource(it.sScore), true, sc.items, sc.itemNodes, prev, eq, status);
		}
		if (status != EXT_OK) return status;
		SliceInfo cur;
		cur.first = nItems; cur.count = 0; cur.bandwidth = bandwidth; cur.j = j;
		cur.minScore = INT32_MAX - bandwidth - 1; cur.minNode = 0xffffffffu; cur.minOffset = 0xffffffffu;
		int32_t flatMin = INT32_MAX; uint32_t flatNode = 0xffffffffu, flatOffset = 0xffffffffu;
		int32_t currentMin = cur.minScore;
		while (nPending > 0) {
			// pop the pending node with the lowest topological rank (ComponentPriorityQueue order on a DAG)
			uint32_t best = 0;
			for (uint32_t i = 1; i < nPending; i++) if (sc.pending[i].comp < sc.pending[best].comp) best = i;
			Pending p = sc.pending[best];
			sc.pending[best] = sc.pending[nPending - 1];
			nPending--;
			if (nItems >= cfg.maxItems) return EXT_OVERFLOW;
			int prevIdx = findItem(sc.itemNodes, prev, p.node);
			bool prevExists = prevIdx >= 0;
			NodeItem& out = sc.items[nItems];
			TileResult tr = computeTile(g, p.node, WS { p.VP, p.VN, p.score }, prevExists, prevExists ? sc.items[prevIdx].sScore : 0,
				prevExists ? sc.items[prevIdx].HP : ~0ull, prevExists ? sc.items[prevIdx].HN : 0ull, eq, out, nullptr, flatRows, status);
			if (status != EXT_OK) return status;
			out.minScore = tr.minScore;
			sc.itemNodes[nItems] = p.node;
			nItems++;
			cur.count++;
			cnt.dpTiles++;
			cnt.columnSteps += g.nodeLength[p.node];
			if (flatRows > 0) { cnt.recomputeTiles++; cnt.columnSteps += g.nodeLength[p.node]; }   // the reference recomputes the tile in flattenLastSliceEnd
			if (tr.minScore > previousQuitScore + bandwidth + 128) return EXT_ASSERT;   // ...Banded.h:352
			currentMin = tr.minScore < currentMin ? tr.minScore : currentMin;
			if (tr.minScore < cur.minScore) { cur.minScore = tr.minScore; cur.minNode = p.node; cur.minOffset = tr.minOffset; }
			if (flatRows > 0) {   // strict '<' in pop order = band-entry order (the defined tie order); a second node AT the minimum is the tie gc_result::flatten_ties counts (bit 31 of flatOffset: no register of its own)
				if (tr.flatMin < flatMin) { flatMin = tr.flatMin; flatNode = p.node; flatOffset = tr.flatOffset; }
				else if (tr.flatMin == flatMin) flatOffset |= 0x80000000u;
			}
			WS newEnd = itemEnd(out);
			int32_t newEndMin = wsColumnMin(newEnd);
			if (newEndMin < previousMinScore) return EXT_ASSERT;   // ...Banded.h:368
			if (newEndMin <= currentMin + bandwidth) {
				for (uint32_t e = g.outOff[p.node]; e < g.outOff[p.node + 1]; e++) {
					pushEdge(g, sc.pending, nPending, cfg, g.outAdj[e], newEnd, false, sc.items, sc.itemNodes, prev, eq, status);
					if (status != EXT_OK) return status;
				}
			}
		}
		if (cur.count == 0) return EXT_ASSERT;
		const uint32_t flatTie = flatRows > 0 ? (flatOffset >> 31) << 3 : 0u;   // kept in the slice's flags (bit 3)
		if (flatRows > 0) { cur.minScore = flatMin; cur.minNode = flatNode; cur.minOffset = flatOffset & 0x7fffffffu; }
		if (cur.minScore < prev.minScore) return EXT_ASSERT;   // ...Banded.h:463
		// correctness HMM (src/AlignmentCorrectnessEstimation.cpp:105-129): +, max, >= only
		{
			int mm = cur.minScore - prev.minScore;
			int idx = mm < 64 ? mm : 63;
			bool cfc = prev.correctLogOdds + ct.c2c >= prev.falseLogOdds + ct.f2c;
			bool ffc = prev.correctLogOdds + ct.c2f >= prev.falseLogOdds + ct.f2f;
			double a = prev.correctLogOdds + ct.c2c, b = prev.falseLogOdds + ct.f2c;
			double c = prev.correctLogOdds + ct.c2f, d = prev.falseLogOdds + ct.f2f;
			cur.correctLogOdds = (a > b ? a : b) + ct.correctOdds[idx];
			cur.falseLogOdds = (c > d ? c : d) + ct.wrongOdds[idx];
			cur.flags = (cur.correctLogOdds > cur.falseLogOdds ? 1u : 0u) | (cfc ? 2u : 0u) | (ffc ? 4u : 0u) | flatTie;
		}
		if (!(cur.flags & 2u)) break;   // !CorrectFromCorrect: stop, slice not kept (...Banded.h:589-607)
		sc.slices[nSlices++] = cur;
	}
	// removeWronglyAlignedEnd, ...Common.h:1231-1241
	{
		bool currentlyCorrect = (sc.slices[nSlices - 1].flags & 1u) != 0;
		while (!currentlyCorrect) {
			currentlyCorrect = (sc.slices[nSlices - 1].flags & 4u) != 0;
			nSlices--;
			if (nSlices == 0) break;
		}
	}
	if (nSlices <= 1) return EXT_FAILED;
	const SliceInfo& last = sc.slices[nSlices - 1];
	if (last.minScore < 0 || last.minScore > len + 128) return EXT_ASSERT;
	score = last.minScore;

	// ---- backtrace (getReverseTraceFromTable, ...Common.h:392-544)
	Cell here { last.minNode, last.minOffset, (last.j + 63 < len - 1) ? last.j + 63 : len - 1 };
	if (!pushTrace(sc, cfg, nTrace, here, false, status)) return status;
	uint32_t curSlice = 0xffffffffu, curNode = 0xffffffffu;
	int curItem = -1;
	ColumnScores colScores { 0, 0, 0 };
	uint32_t ringLo = 0;          // the lowest column the store holds (0 with a whole-tile store)
	auto column = [&](uint32_t c) -> WS { const WCol w = sc.columns[(c & sc.colMask) * sc.colStride]; return WS { w.VP, w.VN, colScores.at(c) }; };
	// a walk that is about to read columns c and c - 1 of a ring that no longer holds c - 1 sets refillTo = c and goes back to the top of the loop, where the tile is recomputed up to c
	// (ONE call site of the recompute: a second inlined copy of the tile loop costs the kernel 90 spilled registers)
	const uint32_t NO_REFILL = 0xffffffffu;
	uint32_t refillTo = NO_REFILL;
	while (here.seqPos != -1) {
		uint32_t s = (uint32_t)(here.seqPos / 64) + 1;
		if (s >= nSlices) return EXT_ASSERT;
		const bool entering = s != curSlice || here.node != curNode;
		if (entering) {
			if (s != curSlice) eqs.rows(len, sc.slices[s].j, eq);
			curSlice = s;
			curNode = here.node;
			curItem = findItem(sc.itemNodes, sc.slices[s], curNode);
			if (curItem < 0) return EXT_ASSERT;
		}
		if (entering || refillTo != NO_REFILL) {
			// a whole-tile store takes the whole tile (and checks it against the DP's end column, as the reference does); a ring starts at the column the walk enters at - it only moves left
			const uint32_t upTo = entering ? (sc.colMask >= 63 ? 63u : here.offset) : refillTo;
			colScores = recomputeColumns(g, sc, s, curItem, eq, status, cnt, upTo, entering);
			ringLo = upTo > sc.colMask ? upTo - sc.colMask : 0;
			refillTo = NO_REFILL;
			if (status != EXT_OK) return status;
		}
		const SliceInfo& cs = sc.slices[s];
		const SliceInfo& ps = sc.slices[s - 1];
		int row = here.seqPos & 63;
		if (row == 0 && here.offset == 0) {
			Cell nxt; bool sw;
			if (!backtraceCorner(g, sc, s, curNode, curItem, eq, nxt, sw)) return EXT_ASSERT;
			if (!pushTrace(sc, cfg, nTrace, nxt, sw, status)) return status;
			here = nxt;
			continue;
		}
		if (row == 0) {
			// vertical crossing into the previous slice (...Common.h:451-477, pickBacktraceVerticalCrossing :665-708)
			int prevIdx = findItem(sc.itemNodes, ps, curNode);
			if (prevIdx < 0) {
				here = Cell { curNode, 0, here.seqPos };
				if (!pushTrace(sc, cfg, nTrace, here, false, status)) return status;
				continue;
			}
			uint32_t off = here.offset;
			while (off > 0) {
				if (off - 1 < ringLo) { refillTo = off; break; }
				if (wsValue(column(off - 1), 0) != wsValue(column(off), 0) - 1) break;
				off--;
				if (!pushTrace(sc, cfg, nTrace, Cell { curNode, off, here.seqPos }, false, status)) return status;
			}
			here.offset = off;
			if (refillTo != NO_REFILL) continue;
			if (off == 0) {
				Cell nxt; bool sw;
				if (!backtraceCorner(g, sc, s, curNode, curItem, eq, nxt, sw)) return EXT_ASSERT;
				if (!pushTrace(sc, cfg, nTrace, nxt, sw, status)) return status;
				here = nxt;
				continue;
			}
			const NodeItem& pn = sc.items[prevIdx];
			int32_t scoreHere = wsValue(column(off), 0);
			int32_t scoreDiagonal = pn.sScore;
			uint64_t lowMask = off >= 1 ? (((1ull << off) - 1) & ~1ull) : 0ull;   // bits 1..off-1
			scoreDiagonal += popc64(pn.HP & lowMask) - popc64(pn.HN & lowMask);
			int32_t scoreUp = scoreDiagonal + (int32_t)((pn.HP >> off) & 1) - (int32_t)((pn.HN >> off) & 1);
			int32_t quitScore = cs.minScore + cs.bandwidth, previousQuitScore = ps.minScore + ps.bandwidth;
			Cell nxt;
			if (scoreHere > quitScore || scoreDiagonal > previousQuitScore || scoreUp > previousQuitScore) {
				nxt = scoreDiagonal < scoreUp ? Cell { curNode, off - 1, here.seqPos - 1 } : Cell { curNode, off, here.seqPos - 1 };
			} else {
				NodeSeq nseq = loadNodeSeq(g, curNode);
				int eqBit = (int)(eqOfColumn(eq, nseq, (int)off) & 1);
				if (scoreUp == scoreHere - 1) nxt = Cell { curNode, off, here.seqPos - 1 };
				else if (scoreDiagonal == scoreHere - (eqBit ? 0 : 1)) nxt = Cell { curNode, off - 1, here.seqPos - 1 };
				else return EXT_ASSERT;
			}
			if (!pushTrace(sc, cfg, nTrace, nxt, false, status)) return status;
			here = nxt;
			continue;
		}
		if (here.offset == 0) {
			// horizontal crossing into an in-neighbour (...Common.h:478-499, pickBacktraceHorizontalCrossing :599-663)
			WS start = itemStart(sc.items[curItem]);
			int32_t sp = here.seqPos;
			while ((sp & 63) != 0 && (start.VP & (1ull << (sp & 63)))) {
				sp--;
				if (!pushTrace(sc, cfg, nTrace, Cell { curNode, 0, sp }, false, status)) return status;
			}
			here.seqPos = sp;
			int offset = sp & 63;
			if (offset == 0) {
				Cell nxt; bool sw;
				if (!backtraceCorner(g, sc, s, curNode, curItem, eq, nxt, sw)) return EXT_ASSERT;
				if (!pushTrace(sc, cfg, nTrace, nxt, sw, status)) return status;
				here = nxt;
				continue;
			}
			NodeSeq nseq = loadNodeSeq(g, curNode);
			int eqBit = (int)((eqOfColumn(eq, nseq, 0) >> offset) & 1);
			int32_t scoreHere = wsValue(start, offset);
			int32_t quitScore = cs.minScore + cs.bandwidth;
			Cell nxt { 0, 0, 0 };
			bool sw = false, found = false;
			if (scoreHere > quitScore) {
				int32_t smallest = wsValue(start, offset - 1);
				nxt = Cell { curNode, 0, sp - 1 };
				for (uint32_t e = g.inOff[curNode]; e < g.inOff[curNode + 1]; e++) {
					uint32_t nb = g.inAdj[e];
					int c = findItem(sc.itemNodes, cs, nb);
					if (c < 0) continue;
					WS ne = itemEnd(sc.items[c]);
					if (wsValue(ne, offset - 1) <= smallest) { smallest = wsValue(ne, offset - 1); nxt = Cell { nb, (uint32_t)g.nodeLength[nb] - 1, sp - 1 }; sw = true; }
					if (wsValue(ne, offset) < smallest && nb != curNode) { smallest = wsValue(ne, offset); nxt = Cell { nb, (uint32_t)g.nodeLength[nb] - 1, sp }; sw = true; }
				}
				found = true;
			} else {
				for (uint32_t e = g.inOff[curNode]; e < g.inOff[curNode + 1] && !found; e++) {
					uint32_t nb = g.inAdj[e];
					int c = findItem(sc.itemNodes, cs, nb);
					if (c < 0) continue;
					WS ne = itemEnd(sc.items[c]);
					if (wsValue(ne, offset) == scoreHere - 1) { nxt = Cell { nb, (uint32_t)g.nodeLength[nb] - 1, sp }; sw = true; found = true; }
					else if (wsValue(ne, offset - 1) == scoreHere - (eqBit ? 0 : 1)) { nxt = Cell { nb, (uint32_t)g.nodeLength[nb] - 1, sp - 1 }; sw = true; found = true; }
				}
			}
			if (!found) return EXT_ASSERT;
			if (!pushTrace(sc, cfg, nTrace, nxt, sw, status)) return status;
			here = nxt;
			continue;
		}
		// inside the tile (pickBacktraceInside, ...Common.h:556-597): vertical, then diagonal, then horizontal
		{
			uint32_t hori = here.offset;
			int vert = row;
			NodeSeq nseq = loadNodeSeq(g, curNode);
			while (hori > 0 && vert > 0) {
				if (hori - 1 < ringLo) { refillTo = hori; break; }
				const WS colHere = column(hori), colLeft = column(hori - 1);
				int32_t scoreHere = wsValue(colHere, vert);
				int32_t vertical = wsValue(colHere, vert - 1);
				int32_t diagonal = wsValue(colLeft, vert - 1);
				int eqBit = (int)((eqOfColumn(eq, nseq, (int)hori) >> vert) & 1);
				if (vertical == scoreHere - 1) { vert--; }
				else if (diagonal == scoreHere - (eqBit ? 0 : 1)) { hori--; vert--; }
				else {
					if (wsValue(colLeft, vert) != scoreHere - 1) return EXT_ASSERT;
					hori--;
				}
				if (!pushTrace(sc, cfg, nTrace, Cell { curNode, hori, cs.j + vert }, false, status)) return status;
			}
			here = Cell { curNode, hori, cs.j + vert };
		}
	}
	// row -1: walk left along the initial ramp, then maybe into an in-neighbour that is in the initial slice
	// (...Common.h:508-542; the initial slice holds only the seed node)
	{
		const NodeItem& n0 = sc.items[0];
		if (here.node != n0.node) return EXT_ASSERT;
		uint32_t off = here.offset;
		// before[i] = |i - startOffset|: decreasing towards startOffset from the right
		while (true) {
			int32_t b = (int32_t)off - (int32_t)startOffset; if (b < 0) b = -b;
			int32_t bl = (int32_t)off - 1 - (int32_t)startOffset; if (bl < 0) bl = -bl;
			if (!(b != 0 && off > 0 && bl == b - 1)) break;
			off--;
			if (!pushTrace(sc, cfg, nTrace, Cell { here.node, off, -1 }, false, status)) return status;
		}
		// the in-neighbour hop (:528-541) needs the neighbour inside slices[0], which only contains the seed node
		// (a self-loop would be needed), so it cannot trigger on a DAG.
	}
	cnt.traceItems += nTrace;
	cnt.flattenTie = (sc.slices[nSlices - 1].flags >> 3) & 1u;   // the backtrace started in a flattened slice whose minimum was tied between nodes (read back here: no register held across the walk)
	return status;
}

__device__ inline uint32_t extendSeed(const DGraph& g, const CorrectnessTables& ct, const uint8_t* iupac, const ExtendConfig& cfg, const LaneScratch& sc,
	const char* seq, int len, uint32_t startNode, uint32_t startOffset, uint32_t& nTrace, int32_t& score, ExtCounters& cnt)
{
	return extendSeedT(g, ct, EqFromBases { seq, iupac }, cfg, sc, len, startNode, startOffset, nTrace, score, cnt);
}

} // namespace gcdev
