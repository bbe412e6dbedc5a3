// Whole-read variant of the seed-extension core (same algorithm and results as extendSeed in gc_device.hpp), in two
// instantiations of one template:
//
//  * REGCOLS = true - one extension per wave (k_long_extend<1>, the production path). All 64 lanes run the same uniform
//    code, so the extension's state sits in scalar registers, and the lanes' VGPRs carry the per-entry data: the band
//    tables (previous slice, current slice, pending queue: entry e in lane e, 64 entries), the 64 recomputed columns of
//    the backtrace (column c in lane c), the node ids of the backtrace's current/previous slice and the trace cells on
//    their way out (64 at a time). No LDS.
//  * REGCOLS = false - LANES extensions per wave, one per lane (experiments, and the retry for slices with more than 64
//    nodes): the tables live in LDS, lane-interleaved ([word][lane]), 28 entries each with a spill area of 228 more in
//    HBM; the backtrace columns and id lists alias the table words (the two phases do not overlap in time).
//
// In both, the DP table kept for the backtrace (8 words per (slice,node) item), the per-slice records and the trace live
// in HBM, interleaved across the lanes of the team at 8-byte granularity (word w of lane l at base[w*lanes + l]).
// A slice that needs more nodes than the tables hold ends the extension with EXT_LDS_CAP (see k_long_extend's retry and
// the plain-layout fallback k_long_pass).
#pragma once
#include "gc_device.hpp"
#include <type_traits>

namespace gcdev {


// The scalar-lean forms of the one-extension-per-wave instantiation's hot loops (r2 / r3): each GC_LEAN_* names one rewrite whose plain twin stays in the source as
// the readable statement of the same step, as the code of the multi-lane instantiations (the register-table retry runs LANES = 2) and of the host compile. The product build
// always takes the lean forms; only the experiments build (-DGC_EXPERIMENTS) may switch one off for an A/B (`make variant FLAGS="-DGC_EXPERIMENTS -DGC_LEAN_WALK=0"`).
#ifndef GC_EXPERIMENTS
#if defined(GC_LEAN_COLUMNS) || defined(GC_LEAN_WALK) || defined(GC_LEAN_PUSH) || defined(GC_LEAN_TABLES) || defined(GC_LEAN_COLMIN) || defined(GC_LEAN_MERGE) || defined(GC_LEAN_POP) || defined(GC_LEAN_KINDS) || defined(GC_LEAN_DIAGRUN)
#error "GC_LEAN_* can only be set in the experiments build (-DGC_EXPERIMENTS)"
#endif
#endif
#ifndef GC_LEAN_COLUMNS
#define GC_LEAN_COLUMNS 1   // column loop: match mask by s_cselect_b64, carries by s_bfe_u64, the column's consumers in one vector epilogue per tile
#endif
#ifndef GC_LEAN_WALK
#define GC_LEAN_WALK 1      // backtrace walk on bit masks made by one vector pass per tile
#endif
#ifndef GC_LEAN_PUSH
#define GC_LEAN_PUSH 1      // trace cells through v_writelane
#endif
#ifndef GC_LEAN_TABLES
#define GC_LEAN_TABLES 1    // band-table entries through v_writelane
#endif
#ifndef GC_LEAN_COLMIN
#define GC_LEAN_COLMIN 1    // minimum of a tile's end column with lane = row
#endif
#ifndef GC_LEAN_MERGE
#define GC_LEAN_MERGE 1     // mergeTwoSlices with lane = row
#endif
#ifndef GC_LEAN_POP
#define GC_LEAN_POP 1       // the pop of the pending queue as a wave minimum over the lanes' component numbers (r3)
#endif
#ifndef GC_LEAN_KINDS
#define GC_LEAN_KINDS 1     // three copies of the column loop: general / node in the previous slice with nothing forced / node new in this slice (constant carries)
#endif
#ifndef GC_LEAN_UNROLL
#define GC_LEAN_UNROLL 1
#endif
#ifndef GC_LEAN_DIAGRUN
#define GC_LEAN_DIAGRUN 1   // the backtrace's diagonal runs inside a tile resolved by one ballot and emitted by the vector pipe
#endif
#ifndef WAVE_CAP
#define WAVE_CAP 28
#endif

// LDS image of one wave: 4-byte words, lane-interleaved ([word][lane]: a wave-level access to the same logical word
// hits 64 consecutive banks). Every lane owns the words [*][lane]; both phases of an extension use only the lane's own
// words, so lanes of one wave may be in different phases. Per entry 7 words for each of: previous-slice table,
// current-slice table (the two swap every slice) and pending queue -> 21 words; 28 entries = 588 words = 150.5 KB per wave.
#define WAVE_ENTRY_WORDS 7
#define WAVE_WORDS (3 * WAVE_CAP * WAVE_ENTRY_WORDS)
template <int LANES> struct WaveLdsT { uint32_t w[WAVE_WORDS][LANES]; };
typedef WaveLdsT<64> WaveLds;
static_assert(sizeof(WaveLds) <= 160 * 1024, "WaveLds must fit the CU's LDS");
static_assert(64 * 5 <= WAVE_WORDS, "the 64 backtrace columns (5 words each) alias the lane's table words");

// Entries beyond WAVE_CAP (rare: very dense variant clusters) spill to lane-interleaved HBM words, so a slice may hold
// up to WAVE_CAP + WAVE_SPILL nodes before the extension gives up with EXT_LDS_CAP.
#define WAVE_SPILL 228
#define WAVE_MAX_ENTRIES (WAVE_CAP + WAVE_SPILL)
#define WAVE_SPILL_WORDS (3 * WAVE_SPILL * 4)

typedef __attribute__((address_space(3))) uint32_t lds_u32;   // keeps LDS accesses as ds_read/ds_write (a generic pointer compiles to flat_*)

// REGCOLS (one extension per wave, all 64 lanes alive and running the same uniform code): the 64 recomputed columns of
// the backtrace live in five VGPRs, column c in lane c (v_writelane / v_readlane with a uniform index) instead of LDS.
// (this clang has no writelane builtin; "mine if my lane id equals c" is one compare and five selects, as cheap)
#if defined(__HIP_DEVICE_COMPILE__)
#define GC_READLANE(reg, lane) __builtin_amdgcn_readlane((int)(reg), (int)(lane))
#else
#define GC_READLANE(reg, lane) ((int)(reg))
#endif
// A uniform 64-bit value as an SGPR pair (inline asm with "s" operands must not be handed a VGPR; free when the value already is scalar)
__device__ __forceinline__ uint64_t gcUniform64(uint64_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) << 32);
#else
	return x;
#endif
}
template <bool REGCOLS>
struct LaneLdsT {   // one lane's view
	lds_u32* l;                  // LDS words of the team, [word][lanes]
	uint32_t lane, lanes;        // this lane's index in its team and the team size (active lanes per wave)
	unsigned long long* spill;   // team-interleaved HBM words for entries >= WAVE_CAP: word w of this lane at spill[w*lanes + lane]
	__device__ __forceinline__ uint32_t ldL(uint32_t word) const { return l[word * lanes + lane]; }
	__device__ __forceinline__ void stL(uint32_t word, uint32_t v) const { l[word * lanes + lane] = v; }
	__device__ __forceinline__ uint64_t ldL64(uint32_t word) const { return (uint64_t)ldL(word) | ((uint64_t)ldL(word + 1) << 32); }
	__device__ __forceinline__ void stL64(uint32_t word, uint64_t v) const { stL(word, (uint32_t)v); stL(word + 1, (uint32_t)(v >> 32)); }
	__device__ __forceinline__ unsigned long long& S(uint32_t word) const { return spill[(uint64_t)word * lanes + lane]; }
	// an entry of table t (0/1 = slice tables, 2 = pending queue): 7 words in LDS, or (spilled) 5 HBM words {w0|w1<<32, w2, 64-bit a, 64-bit b}
	struct Entry { uint32_t w0, w1, w2; uint64_t a, b; };
	__device__ __forceinline__ uint32_t ldsBase(uint32_t t, uint32_t e) const { return (t * WAVE_CAP + e) * WAVE_ENTRY_WORDS; }
	__device__ __forceinline__ uint32_t spillBase(uint32_t t, uint32_t e) const { return (t * WAVE_SPILL + (e - WAVE_CAP)) * 4; }
	// REGCOLS: the three tables live in registers across the lanes as well - entry e of table t in lane e (64 entries, no spill):
	// reads are v_readlane with a uniform index, writes one compare + selects, "find node" one compare + ballot.
	mutable uint32_t tw[3][7];
	uint32_t regCap;
	__device__ __forceinline__ uint32_t maxEntries() const { return REGCOLS ? regCap : (uint32_t)WAVE_MAX_ENTRIES; }
	__device__ __forceinline__ Entry get(uint32_t t, uint32_t e) const
	{
		Entry x;
		if (REGCOLS) {
			x.w0 = (uint32_t)GC_READLANE(tw[t][0], e); x.w1 = (uint32_t)GC_READLANE(tw[t][1], e); x.w2 = (uint32_t)GC_READLANE(tw[t][2], e);
			x.a = (uint64_t)(uint32_t)GC_READLANE(tw[t][3], e) | ((uint64_t)(uint32_t)GC_READLANE(tw[t][4], e) << 32);
			x.b = (uint64_t)(uint32_t)GC_READLANE(tw[t][5], e) | ((uint64_t)(uint32_t)GC_READLANE(tw[t][6], e) << 32);
			return x;
		}
		if (e < WAVE_CAP) {
			uint32_t base = ldsBase(t, e);
			x.w0 = ldL(base); x.w1 = ldL(base + 1); x.w2 = ldL(base + 2); x.a = ldL64(base + 3); x.b = ldL64(base + 5);
		} else {
			uint32_t base = spillBase(t, e);
			unsigned long long p = S(base);
			x.w0 = (uint32_t)p; x.w1 = (uint32_t)(p >> 32); x.w2 = (uint32_t)S(base + 1); x.a = S(base + 2); x.b = S(base + 3);
		}
		return x;
	}
	__device__ __forceinline__ void set(uint32_t t, uint32_t e, const Entry& x) const
	{
#if GC_LEAN_TABLES && defined(__HIP_DEVICE_COMPILE__)
		if (REGCOLS) {   // seven v_writelane with the entry index in M0 (the values are uniform: straight from SGPRs, no compare, no moves)
			const uint32_t es = (uint32_t)__builtin_amdgcn_readfirstlane((int)e);
			const uint32_t v0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)x.w0), v1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)x.w1), v2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)x.w2);
			const uint64_t a = gcUniform64(x.a), b = gcUniform64(x.b);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
			asm("s_mov_b32 m0, %7\n\tv_writelane_b32 %0, %8, m0\n\tv_writelane_b32 %1, %9, m0\n\tv_writelane_b32 %2, %10, m0\n\tv_writelane_b32 %3, %11, m0\n\tv_writelane_b32 %4, %12, m0\n\tv_writelane_b32 %5, %13, m0\n\tv_writelane_b32 %6, %14, m0"
				: "+v"(tw[t][0]), "+v"(tw[t][1]), "+v"(tw[t][2]), "+v"(tw[t][3]), "+v"(tw[t][4]), "+v"(tw[t][5]), "+v"(tw[t][6])
				: "s"(es), "s"(v0), "s"(v1), "s"(v2), "s"((uint32_t)a), "s"((uint32_t)(a >> 32)), "s"((uint32_t)b), "s"((uint32_t)(b >> 32)) : "m0");
#pragma clang diagnostic pop
			return;
		}
#endif
		if (REGCOLS) {
			const bool mine = threadIdx.x == e;
			tw[t][0] = mine ? x.w0 : tw[t][0]; tw[t][1] = mine ? x.w1 : tw[t][1]; tw[t][2] = mine ? x.w2 : tw[t][2];
			tw[t][3] = mine ? (uint32_t)x.a : tw[t][3]; tw[t][4] = mine ? (uint32_t)(x.a >> 32) : tw[t][4];
			tw[t][5] = mine ? (uint32_t)x.b : tw[t][5]; tw[t][6] = mine ? (uint32_t)(x.b >> 32) : tw[t][6];
			return;
		}
		if (e < WAVE_CAP) {
			uint32_t base = ldsBase(t, e);
			stL(base, x.w0); stL(base + 1, x.w1); stL(base + 2, x.w2); stL64(base + 3, x.a); stL64(base + 5, x.b);
		} else {
			uint32_t base = spillBase(t, e);
			S(base) = (unsigned long long)x.w0 | ((unsigned long long)x.w1 << 32); S(base + 1) = x.w2; S(base + 2) = x.a; S(base + 3) = x.b;
		}
	}
	// REGCOLS: table 1 (the slice just finished) becomes table 0 (the previous slice)
	__device__ __forceinline__ void rotate() const { for (int k = 0; k < 7; k++) tw[0][k] = tw[1][k]; }
	__device__ __forceinline__ uint32_t word0(uint32_t t, uint32_t e) const { if (REGCOLS) return (uint32_t)GC_READLANE(tw[t][0], e); return e < WAVE_CAP ? ldL(ldsBase(t, e)) : (uint32_t)S(spillBase(t, e)); }
	__device__ __forceinline__ uint32_t word1(uint32_t t, uint32_t e) const { if (REGCOLS) return (uint32_t)GC_READLANE(tw[t][1], e); return e < WAVE_CAP ? ldL(ldsBase(t, e) + 1) : (uint32_t)(S(spillBase(t, e)) >> 32); }
	__device__ __forceinline__ uint32_t word2(uint32_t t, uint32_t e) const { if (REGCOLS) return (uint32_t)GC_READLANE(tw[t][2], e); return e < WAVE_CAP ? ldL(ldsBase(t, e) + 2) : (uint32_t)S(spillBase(t, e) + 1); }
	// index of the entry of table t (first n entries) whose word 0 (the node) equals `node`, or -1
	__device__ __forceinline__ int find(uint32_t t, uint32_t n, uint32_t node) const
	{
		if (REGCOLS) {
			unsigned long long m = __ballot(tw[t][0] == node);
			if (n < 64) m &= (1ull << n) - 1;
			return m ? __ffsll((long long)m) - 1 : -1;
		}
		for (uint32_t i = 0; i < n; i++) if (word0(t, i) == node) return (int)i;
		return -1;
	}
	// slice tables: buffer b (0/1), entry e: node, startScore, minScore, HP, HN
	__device__ __forceinline__ uint32_t pNode(int b, uint32_t e) const { return word0((uint32_t)b, e); }
	__device__ __forceinline__ int32_t pStart(int b, uint32_t e) const { return (int32_t)word1((uint32_t)b, e); }
	__device__ __forceinline__ int32_t pMin(int b, uint32_t e) const { return (int32_t)word2((uint32_t)b, e); }
	__device__ __forceinline__ void pSet(int b, uint32_t e, uint32_t node, int32_t start, int32_t mn, uint64_t hp, uint64_t hn) const { set((uint32_t)b, e, Entry { node, (uint32_t)start, (uint32_t)mn, hp, hn }); }
	// pending queue entry e: node, comp, score, VP, VN
	__device__ __forceinline__ uint32_t qNode(uint32_t e) const { return word0(2u, e); }
	__device__ __forceinline__ uint32_t qComp(uint32_t e) const { return word1(2u, e); }
	__device__ __forceinline__ WS qWs(uint32_t e) const { Entry x = get(2u, e); return WS { x.a, x.b, (int32_t)x.w2 }; }
	__device__ __forceinline__ void qSet(uint32_t e, uint32_t node, uint32_t comp, const WS& x) const { set(2u, e, Entry { node, comp, (uint32_t)x.score, x.VP, x.VN }); }
	__device__ __forceinline__ void qSetWs(uint32_t e, const WS& x) const { Entry old = get(2u, e); set(2u, e, Entry { old.w0, old.w1, (uint32_t)x.score, x.VP, x.VN }); }
	__device__ __forceinline__ void qMove(uint32_t dst, uint32_t src) const { set(2u, dst, get(2u, src)); }
	// the pending entry with the smallest componentNumber (the reference's priority queue pops by it, src/ComponentPriorityQueue.h; the first of equals, as the scalar scan picks)
	__device__ __forceinline__ uint32_t qArgMin(uint32_t n) const
	{
#if GC_LEAN_POP && defined(__HIP_DEVICE_COMPILE__)
		if (REGCOLS) {
			// entry e sits in lane e: a wave minimum by DPP and one ballot instead of a scalar loop of readlane / compare / select per entry (~25 scalar instructions per tile on cfg2)
			const uint32_t mine = threadIdx.x < n ? tw[2][1] : 0xffffffffu;
			uint32_t v = mine;
			auto lower = [](uint32_t a, uint32_t b) { return a < b ? a : b; };
			v = lower(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x111, 0xf, 0xf, false));
			v = lower(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x112, 0xf, 0xf, false));
			v = lower(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x114, 0xf, 0xf, false));
			v = lower(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x118, 0xf, 0xf, false));
			v = lower(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x142, 0xa, 0xf, false));
			v = lower(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, 0x143, 0xc, 0xf, false));
			const uint32_t smallest = (uint32_t)GC_READLANE(v, 63);
			const unsigned long long at = __ballot(mine == smallest && threadIdx.x < n);
			return (uint32_t)__ffsll((long long)at) - 1u;
		}
#endif
		uint32_t best = 0;
		uint32_t bestComp = qComp(0);
		for (uint32_t i = 1; i < n; i++) { uint32_t c = qComp(i); if (c < bestComp) { bestComp = c; best = i; } }
		return best;
	}
	// backtrace columns (alias the LDS table words, or registers across the lanes): column c: VP, VN, score
	mutable uint32_t cr[5];
	mutable uint32_t idCur, idPrev;
	static constexpr bool eqInLanes = REGCOLS;   // (the name is historical: "this is the one-extension-per-wave instantiation", whose lean column loop and walk masks apply)
	// REGCOLS: node ids of the items of the backtrace's current / previous slice, item i in lane i (idCur / idPrev above)
	// REGCOLS backtrace: per column c >= 1 of the recomputed tile (lane c), for all 64 rows at once: "the diagonal predecessor fits" and
	// "the left predecessor fits" (see setWalkMasks); valid only after a recompute that went through the lean column loop
	mutable uint32_t wm[4] = { 0, 0, 0, 0 };
	mutable bool walkMasks = false;
	// The walk inside a tile asks three things per cell (src/GraphAlignerBitvectorCommon.h:555-708): is the cell above one less (the column's VP
	// bit), is the diagonal cell equal / one less depending on the match bit, is the left cell one less. Cell values of neighbouring columns
	// differ by the horizontal deltas Ph / Mh of the Myers step that produced the column from its left neighbour - an identity of the step
	// itself, for any input column: value(r,c) - value(r,c-1) = Ph_r - Mh_r, hence value(r,c) - value(r-1,c-1) = (Ph_r - Mh_r) + (VP_r - VN_r of
	// column c-1) for r >= 1. Both columns are in the lanes (column c in lane c), so one vector pass redoes the horizontal part of every
	// column's step at once (lane c reads column c-1 from its neighbour) and leaves two row masks per column; the scalar walk then tests bits.
#if defined(__HIP_DEVICE_COMPILE__)
	__device__ __forceinline__ void setWalkMasks(uint64_t codes0, uint64_t codes1, const Eq4& eq, uint64_t forceEq, uint64_t prevHN) const
	{
		const uint32_t c = threadIdx.x;
		const uint32_t p0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cr[0], 0x138, 0xf, 0xf, false), p1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cr[1], 0x138, 0xf, 0xf, false);
		const uint32_t n0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cr[2], 0x138, 0xf, 0xf, false), n1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cr[3], 0x138, 0xf, 0xf, false);
		const uint64_t pVP = (uint64_t)p0 | ((uint64_t)p1 << 32), pVN = (uint64_t)n0 | ((uint64_t)n1 << 32);   // column c - 1 (wave_shr:1)
		const uint32_t code = (uint32_t)((c < 32 ? codes0 : codes1) >> ((c & 31u) * 2)) & 3u;
		const uint64_t rawEq = code == 0 ? eq.a : code == 1 ? eq.c : code == 2 ? eq.g : eq.t;
		const uint64_t Eq = (rawEq & forceEq) | ((prevHN >> c) & 1ull);
		const uint64_t Xh = (((Eq & pVP) + pVP) ^ pVP) | Eq;
		const uint64_t Ph = pVN | ~(Xh | pVP), Mh = pVP & Xh;
		const uint64_t flat = ~(Ph | Mh), level = ~(pVP | pVN);
		const uint64_t same = (flat & level) | (Ph & pVN) | (Mh & pVP);   // value(r,c) == value(r-1,c-1)
		const uint64_t more = (Ph & level) | (flat & pVP);                // value(r,c) == value(r-1,c-1) + 1
		const uint64_t diag = (same & rawEq) | (more & ~rawEq);
		wm[0] = (uint32_t)diag; wm[1] = (uint32_t)(diag >> 32); wm[2] = (uint32_t)Ph; wm[3] = (uint32_t)(Ph >> 32);
		walkMasks = true;
	}
	__device__ __forceinline__ void loadWalkMasks(uint32_t c, uint64_t& up, uint64_t& diag, uint64_t& left) const
	{
		up = (uint64_t)(uint32_t)GC_READLANE(cr[0], c) | ((uint64_t)(uint32_t)GC_READLANE(cr[1], c) << 32);
		diag = (uint64_t)(uint32_t)GC_READLANE(wm[0], c) | ((uint64_t)(uint32_t)GC_READLANE(wm[1], c) << 32);
		left = (uint64_t)(uint32_t)GC_READLANE(wm[2], c) | ((uint64_t)(uint32_t)GC_READLANE(wm[3], c) << 32);
	}
#endif
	__device__ __forceinline__ void colSet(uint32_t c, const WS& x) const
	{
		if (REGCOLS) {
			const bool mine = threadIdx.x == c;
			cr[0] = mine ? (uint32_t)x.VP : cr[0];
			cr[1] = mine ? (uint32_t)(x.VP >> 32) : cr[1];
			cr[2] = mine ? (uint32_t)x.VN : cr[2];
			cr[3] = mine ? (uint32_t)(x.VN >> 32) : cr[3];
			cr[4] = mine ? (uint32_t)x.score : cr[4];
		} else { uint32_t base = c * 5; stL64(base, x.VP); stL64(base + 2, x.VN); stL(base + 4, (uint32_t)x.score); }
	}
	__device__ __forceinline__ WS col(uint32_t c) const
	{
		if (REGCOLS) {
			uint32_t a0 = (uint32_t)GC_READLANE((int)cr[0], (int)c), a1 = (uint32_t)GC_READLANE((int)cr[1], (int)c);
			uint32_t b0 = (uint32_t)GC_READLANE((int)cr[2], (int)c), b1 = (uint32_t)GC_READLANE((int)cr[3], (int)c);
			return WS { (uint64_t)a0 | ((uint64_t)a1 << 32), (uint64_t)b0 | ((uint64_t)b1 << 32), GC_READLANE((int)cr[4], (int)c) };
		}
		uint32_t base = c * 5;
		return WS { ldL64(base), ldL64(base + 2), (int32_t)ldL(base + 4) };
	}
};
typedef LaneLdsT<false> LaneLds;

// HBM scratch of one wave, lane-interleaved 8-byte words
struct WaveScratch {
	unsigned long long* base;   // team base
	uint32_t lane, lanes;
	uint32_t maxSlices, maxItems, maxTrace;
	uint32_t maxCols;           // one extension per wave: room for the DP's columns (VP, VN: two words each), kept so that the backtrace need not recompute its tiles; 0 = recompute
	uint32_t regCap;            // register tables: entries per table (<= 64)
	bool allLanes;              // all 64 lanes run one extension (identical values): single-record stores go through lane 0 only
	__device__ __forceinline__ bool storer() const { return !allLanes || threadIdx.x == 0; }
	// word offsets (per lane) of the regions
	__device__ __forceinline__ unsigned long long& word(uint64_t w) const { return base[w * lanes + lane]; }
	__device__ __forceinline__ uint64_t sliceBase(uint32_t s) const { return (uint64_t)s * 4; }
	__device__ __forceinline__ uint64_t itemBase(uint32_t i) const { return (uint64_t)maxSlices * 4 + (uint64_t)i * 8; }
	__device__ __forceinline__ uint64_t traceBase(uint32_t t, uint32_t which) const { return (uint64_t)maxSlices * 4 + (uint64_t)maxItems * 8 + (uint64_t)which * maxTrace + t; }
	__device__ __forceinline__ uint64_t colBase() const { return (uint64_t)maxSlices * 4 + (uint64_t)maxItems * 8 + 2ull * maxTrace; }   // (lanes == 1 whenever maxCols > 0)
	__device__ __forceinline__ unsigned long long* spillBase() const { return base + ((uint64_t)maxSlices * 4 + (uint64_t)maxItems * 8 + 2ull * maxTrace + 2ull * maxCols) * lanes; }
};
__host__ __device__ inline uint64_t waveScratchWords(uint32_t maxSlices, uint32_t maxItems, uint32_t maxTrace, uint32_t maxCols = 0) { return (uint64_t)maxSlices * 4 + (uint64_t)maxItems * 8 + 2ull * maxTrace + 2ull * maxCols + WAVE_SPILL_WORDS; }

struct WSlice { int32_t minScore; uint32_t minNode, minOffset, first, count; int32_t bandwidth; int32_t j; uint32_t flags; };

__device__ __forceinline__ void storeSlice(const WaveScratch& ws, uint32_t s, const WSlice& x)
{
	uint64_t b = ws.sliceBase(s);
	if (!ws.storer()) return;
	ws.word(b) = ((unsigned long long)(uint32_t)x.minScore << 32) | x.minNode;
	ws.word(b + 1) = ((unsigned long long)x.minOffset << 32) | x.first;
	ws.word(b + 2) = ((unsigned long long)x.count << 32) | (uint32_t)x.bandwidth;
	ws.word(b + 3) = ((unsigned long long)(uint32_t)x.j << 32) | x.flags;
}
__device__ __forceinline__ WSlice loadSlice(const WaveScratch& ws, uint32_t s)
{
	uint64_t b = ws.sliceBase(s);
	unsigned long long a = ws.word(b), c = ws.word(b + 1), d = ws.word(b + 2), e = ws.word(b + 3);
	WSlice x;
	x.minScore = (int32_t)(a >> 32); x.minNode = (uint32_t)a;
	x.minOffset = (uint32_t)(c >> 32); x.first = (uint32_t)c;
	x.count = (uint32_t)(d >> 32); x.bandwidth = (int32_t)(uint32_t)d;
	x.j = (int32_t)(e >> 32); x.flags = (uint32_t)e;
	return x;
}
__device__ __forceinline__ void storeItem(const WaveScratch& ws, uint32_t i, const NodeItem& it)
{
	uint64_t b = ws.itemBase(i);
	if (!ws.storer()) return;
	ws.word(b) = it.sVP; ws.word(b + 1) = it.sVN; ws.word(b + 2) = it.eVP; ws.word(b + 3) = it.eVN; ws.word(b + 4) = it.HP; ws.word(b + 5) = it.HN;
	ws.word(b + 6) = ((unsigned long long)(uint32_t)it.sScore << 32) | (uint32_t)it.eScore;
	ws.word(b + 7) = ((unsigned long long)(uint32_t)it.minScore << 32) | it.node;
}
__device__ __forceinline__ NodeItem loadItem(const WaveScratch& ws, uint32_t i)
{
	uint64_t b = ws.itemBase(i);
	NodeItem it;
	it.sVP = ws.word(b); it.sVN = ws.word(b + 1); it.eVP = ws.word(b + 2); it.eVN = ws.word(b + 3); it.HP = ws.word(b + 4); it.HN = ws.word(b + 5);
	unsigned long long s = ws.word(b + 6), m = ws.word(b + 7);
	it.sScore = (int32_t)(s >> 32); it.eScore = (int32_t)(uint32_t)s;
	it.minScore = (int32_t)(m >> 32); it.node = (uint32_t)m;
	return it;
}
__device__ __forceinline__ uint32_t itemNode(const WaveScratch& ws, uint32_t i) { return (uint32_t)ws.word(ws.itemBase(i) + 7); }

// table lookup for the backtrace: index of `node` among the items of a slice, or -1
__device__ inline int findItemW(const WaveScratch& ws, const WSlice& sl, uint32_t node)
{
	for (uint32_t i = 0; i < sl.count; i++)
		if (itemNode(ws, sl.first + i) == node) return (int)(sl.first + i);
	return -1;
}

// one trace cell per 8-byte word: node | (seqPos+1) << 32 (24 bits) | offset << 56 (6 bits) | nodeSwitch << 62
__device__ __forceinline__ unsigned long long packCell(Cell c, bool sw) { return (unsigned long long)c.node | ((unsigned long long)(uint32_t)(c.seqPos + 1) << 32) | ((unsigned long long)c.offset << 56) | ((unsigned long long)(sw ? 1 : 0) << 62); }
__device__ __forceinline__ TraceCell unpackCell(unsigned long long w)
{
	TraceCell t;
	t.node = (uint32_t)w;
	t.seqPos = (int32_t)((w >> 32) & 0xffffffu) - 1;
	t.offsetAndSwitch = (uint32_t)((w >> 56) & 63u) | (((w >> 62) & 1u) ? 256u : 0u);
	return t;
}

// Pointwise minimum of two columns (src/WordSlice.h:491-530) for the one-extension-per-wave layout: lane r takes the smaller of the two
// values of row r (two masked popcounts each), fetches row r-1's minimum from its neighbour lane, and two ballots over the differences are
// the merged column's deltas. Same column as wsMerge (the pointwise minimum is unique), without its scalar loop over the differing rows.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ WS wsMergeWave(const WS& a, const WS& b)
{
	const uint64_t upTo = ~(~1ull << threadIdx.x);   // bits 0..r
	const int32_t beforeA = wsBefore(a), beforeB = wsBefore(b);
	const int32_t va = beforeA + popc64(a.VP & upTo) - popc64(a.VN & upTo);
	const int32_t vb = beforeB + popc64(b.VP & upTo) - popc64(b.VN & upTo);
	const int32_t m = va < vb ? va : vb;
	const int32_t beforeMin = beforeA < beforeB ? beforeA : beforeB;
	int32_t above = __builtin_amdgcn_update_dpp(0, m, 0x138, 0xf, 0xf, false);   // wave_shr:1: row r - 1
	if (threadIdx.x == 0) above = beforeMin;
	WS res;
	res.VP = __ballot(m == above + 1);
	res.VN = __ballot(m == above - 1);
	res.score = a.score < b.score ? a.score : b.score;
	return res;
}
#endif

// (node, slice) tile on a node that is new in this slice; same as computeTile but the previous-slice summary comes
// in by value and columns (backtrace recompute) go to the LDS column view.
// MODE 0: DP tile. MODE 1: the backtrace's recompute (columns, their scores and the walk masks stay in the lanes). MODE 2: DP tile whose
// columns are also left in the lanes' column registers (column c in lane c) for the caller to store: the backtrace then loads them back
// instead of running the column loop a second time (one extension per wave with a column store, WaveScratch::maxCols).
// What the preamble leaves for the caller of MODE 2 / the stored-column backtrace: the repaired carries of the row above.
struct TilePreamble { uint64_t prevHP, prevHN, forceEq; int forceUntil; };
template <int MODE, typename LANE_TABLES>
__device__ __forceinline__ TileResult computeTileW(const DGraph& g, uint32_t node, WS ws, bool prevExists, int32_t prevStartScore, uint64_t prevHP, uint64_t prevHN,
	const Eq4& eq, NodeItem& out, const LANE_TABLES& tables, int flatRows, uint32_t& status, TilePreamble* preambleOnly = nullptr)
{
	constexpr bool COLUMNS = MODE == 1;
	int nodeLength = g.nodeLength[node];
	NodeSeq seq = loadNodeSeq(g, node);
	TileResult r;
	r.minScore = ws.score;   // (sic) before the merge with the row above, ...Common.h:968 vs :1052-1058
	r.minOffset = 0;
#if GC_LEAN_MERGE && defined(__HIP_DEVICE_COMPILE__)
	if (prevExists && wsBefore(ws) > prevStartScore) ws = LANE_TABLES::eqInLanes ? wsMergeWave(ws, wsSource(prevStartScore)) : wsMerge(ws, wsSource(prevStartScore));
#else
	if (prevExists && wsBefore(ws) > prevStartScore) ws = wsMerge(ws, wsSource(prevStartScore));
#endif
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
	if (COLUMNS) tables.colSet(0, ws);
	uint64_t forceEq = prevExists ? ~0ull : ~1ull;
	if (preambleOnly) { preambleOnly->prevHP = prevHP; preambleOnly->prevHN = prevHN; preambleOnly->forceEq = forceEq; preambleOnly->forceUntil = forceUntil; return r; }   // (stored-column backtrace: no column loop)
#if GC_LEAN_COLUMNS && defined(__HIP_DEVICE_COMPILE__)
	// One extension per wave: every value below is uniform and the loop is the kernel's scalar-issue bottleneck (80 % of its SALU
	// instructions, 60 per column as the compiler writes the generic loop further down). Same arithmetic with the per-column overhead cut:
	// the match mask picked by three s_cselect_b64 on the running 2-bit code (instead of eight 32-bit selects), the carries of the row above
	// and the forced-first-row flag fetched with one s_bfe_u64 each on a running field descriptor, loop-invariant masks folded into the four
	// match masks, and the bottom row's horizontal deltas parked in lane `pos` of a VGPR (one v_writelane per column, two ballots per tile)
	// instead of two 64-bit shift-or pairs. Not taken for IUPAC nodes and for the read's last slice (row-limited minimum).
	if (LANE_TABLES::eqInLanes && !seq.ambiguous && flatRows <= 0) {
		const uint64_t eA = gcUniform64(eq.a & forceEq), eC = gcUniform64(eq.c & forceEq), eG = gcUniform64(eq.g & forceEq), eT = gcUniform64(eq.t & forceEq);
		const uint64_t forced = gcUniform64(forceUntil >= 63 ? ~0ull : ((2ull << forceUntil) - 1));   // columns 1..forceUntil: first row forced
		prevHP = gcUniform64(prevHP); prevHN = gcUniform64(prevHN);
		uint64_t VP = ws.VP, VN = ws.VN;
		// The bottom row's horizontal deltas (bit 63 of Ph / Mh) go to lane pos of two VGPRs, words as they are; everything that follows from
		// them - HP / HN, the column scores, their minimum and where, the end score - is made once per tile by the vector pipe (ballots, a
		// wave prefix sum, a wave minimum): eight scalar instructions per column less, about thirty vector instructions per tile more.
		uint32_t plusWord = 0, minusWord = 0;
		// two copies of the loop: most tiles sit on a node that was in the previous slice with nothing to repair (forceUntil == 0), and their
		// columns carry no forced first row - three scalar instructions per column less than the general copy
		// KIND 0: the general copy. KIND 1: the node was in the previous slice and nothing had to be repaired (forceUntil == 0): no forced first row.
		// KIND 2: the node is new in this slice (47 % of the DP's columns on cfg2): the row above contributes the constant carries (+1, 0) and
		// every column's first row is forced - three s_bfe_u64 and two ORs less per column.
		auto columnLoop = [&, &plusWord = plusWord, &minusWord = minusWord, &tables = tables](auto kindTag) __attribute__((always_inline)) {   // (explicit captures: asm operands alone do not make a generic lambda capture)
		constexpr int KIND = decltype(kindTag)::value;
		// the column counter carries the field width of s_bfe_u64's descriptor in bit 16 (offset = bits 5:0, width = bits 22:16), and goes into M0
		// as it is: v_writelane takes the lane from M0's low six bits
		int pos = 1 | (1 << 16);
#pragma unroll 1
		for (int half = 0; half < 2; half++) {
			uint64_t codes = half ? seq.w1 : (seq.w0 >> 2);
			codes = gcUniform64(codes);
			const int end = (half ? nodeLength : (nodeLength < 32 ? nodeLength : 32)) | (1 << 16);
#pragma unroll GC_LEAN_UNROLL
			for (; pos < end; pos++) {
				uint64_t lo, hi, Eq, hinP = 1, hinN = 0, f = 0;
				const uint32_t posS = (uint32_t)__builtin_amdgcn_readfirstlane(pos);
				const uint32_t desc = posS;
				asm("s_bitcmp1_b32 %3, 0\n\ts_cselect_b64 %0, %5, %4\n\ts_cselect_b64 %1, %7, %6\n\ts_bitcmp1_b32 %3, 1\n\ts_cselect_b64 %2, %1, %0"
					: "=&s"(lo), "=&s"(hi), "=&s"(Eq) : "s"((uint32_t)codes), "s"(eA), "s"(eC), "s"(eG), "s"(eT) : "scc");
				if (KIND != 2) {
					asm("s_bfe_u64 %0, %1, %2" : "=s"(hinP) : "s"(prevHP), "s"(desc) : "scc");
					asm("s_bfe_u64 %0, %1, %2" : "=s"(hinN) : "s"(prevHN), "s"(desc) : "scc");
				}
				if (KIND == 0) asm("s_bfe_u64 %0, %1, %2" : "=s"(f) : "s"(forced), "s"(desc) : "scc");
				codes >>= 2;
				const uint64_t Xv = Eq | VN;
				if (KIND != 2) Eq |= hinN;
				const uint64_t Xh = (((Eq & VP) + VP) ^ VP) | Eq;
				const uint64_t Ph = VN | ~(Xh | VP);
				const uint64_t Mh = VP & Xh;
				const uint64_t sPh = (Ph << 1) | hinP, sMh = KIND != 2 ? ((Mh << 1) | hinN) : (Mh << 1);
				if (KIND == 0) { VP = (sMh | ~(Xv | sPh)) & ~f; VN = (sPh & Xv) | f; }
				else if (KIND == 1) { VP = sMh | ~(Xv | sPh); VN = sPh & Xv; }
				else {
					// (r6) a node that is new in this slice enters with its first row forced (VP bit 0 clear, VN bit 0 set: every incoming column was forced when it was pushed, and the
					// pointwise minimum of forced columns is forced), its match masks have bit 0 cleared (forceEq) and its carries are (+1, 0): then Xv and sPh both have bit 0 set,
					// sMh has it clear, and the recurrence itself leaves VP bit 0 clear and VN bit 0 set - the two forcing operations of the general copy are no-ops here
					VP = sMh | ~(Xv | sPh); VN = sPh & Xv;
				}
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
				// (two SGPR operands exceed gfx9's constant bus; M0 as lane select does not count)
				if (MODE == 0) {
					asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0" : "+v"(plusWord), "+v"(minusWord) : "s"(posS), "s"((uint32_t)(Ph >> 32)), "s"((uint32_t)(Mh >> 32)) : "m0");
				} else {
					// backtrace recompute / column store: the column itself goes to lane pos of the column registers (its score follows after the loop)
					asm("s_mov_b32 m0, %6\n\tv_writelane_b32 %0, %7, m0\n\tv_writelane_b32 %1, %8, m0\n\tv_writelane_b32 %2, %9, m0\n\tv_writelane_b32 %3, %10, m0\n\tv_writelane_b32 %4, %11, m0\n\tv_writelane_b32 %5, %12, m0"
						: "+v"(tables.cr[0]), "+v"(tables.cr[1]), "+v"(tables.cr[2]), "+v"(tables.cr[3]), "+v"(plusWord), "+v"(minusWord)
						: "s"(posS), "s"((uint32_t)VP), "s"((uint32_t)(VP >> 32)), "s"((uint32_t)VN), "s"((uint32_t)(VN >> 32)), "s"((uint32_t)(Ph >> 32)), "s"((uint32_t)(Mh >> 32)) : "m0");
				}
#pragma clang diagnostic pop
			}
		}
		};
#if GC_LEAN_KINDS
		if (!prevExists) columnLoop(std::integral_constant<int, 2>());
		else if (forceUntil == 0) columnLoop(std::integral_constant<int, 1>());
		else columnLoop(std::integral_constant<int, 0>());
#else
		columnLoop(std::integral_constant<int, 0>());
#endif
		const uint64_t HP = __ballot((int32_t)plusWord < 0), HN = __ballot((int32_t)minusWord < 0);   // lanes 0 and >= nodeLength still hold 0
		out.HP = HP; out.HN = HN;
		out.eVP = VP; out.eVN = VN; out.eScore = ws.score + popc64(HP) - popc64(HN);
		if (nodeLength > 1) {
			// column scores: prefix sum of the deltas across the lanes (row_shr 1, 2, 4, 8 inside a row of 16, then the rows' totals)
			int32_t run = (int32_t)(plusWord >> 31) - (int32_t)(minusWord >> 31);
			run += __builtin_amdgcn_update_dpp(0, run, 0x111, 0xf, 0xf, false);
			run += __builtin_amdgcn_update_dpp(0, run, 0x112, 0xf, 0xf, false);
			run += __builtin_amdgcn_update_dpp(0, run, 0x114, 0xf, 0xf, false);
			run += __builtin_amdgcn_update_dpp(0, run, 0x118, 0xf, 0xf, false);
			run += __builtin_amdgcn_update_dpp(0, run, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
			run += __builtin_amdgcn_update_dpp(0, run, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
			const int32_t columnScore = ws.score + run;
			if (COLUMNS) {
				if (threadIdx.x >= 1 && threadIdx.x < (uint32_t)nodeLength) tables.cr[4] = (uint32_t)columnScore;
			} else {
				// (score << 6) | column: the smallest key is the first column with the smallest score; column 0 enters with the tile's starting minimum
				uint32_t key = threadIdx.x == 0 ? ((uint32_t)r.minScore << 6) : threadIdx.x < (uint32_t)nodeLength ? (((uint32_t)columnScore << 6) | threadIdx.x) : 0xffffffffu;
				auto lower = [](uint32_t a, uint32_t b) { return a < b ? a : b; };
				key = lower(key, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)key, 0x111, 0xf, 0xf, false));
				key = lower(key, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)key, 0x112, 0xf, 0xf, false));
				key = lower(key, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)key, 0x114, 0xf, 0xf, false));
				key = lower(key, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)key, 0x118, 0xf, 0xf, false));
				key = lower(key, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)key, 0x142, 0xa, 0xf, false));
				key = lower(key, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)key, 0x143, 0xc, 0xf, false));
				const uint32_t minKey = (uint32_t)GC_READLANE(key, 63);
				r.minScore = (int32_t)(minKey >> 6);
				r.minOffset = minKey & 63u;
			}
		}
#if GC_LEAN_WALK
		if (COLUMNS) tables.setWalkMasks(seq.w0, seq.w1, eq, forceEq, prevHN);
#endif
		return r;
	}
	if (COLUMNS && LANE_TABLES::eqInLanes) tables.walkMasks = false;
#endif
	uint64_t HP = 0, HN = 0;
	for (int pos = 1; pos < nodeLength; pos++) {
		uint64_t Eq;
		Eq = eqOfColumn(eq, seq, pos);
		Eq &= forceEq;
		uint64_t hp, hn;
		ws = myersStep(Eq, ws, (prevHP >> pos) & 1, (prevHN >> pos) & 1, hp, hn);
		{ const uint64_t f = (uint64_t)((uint32_t)(pos - forceUntil - 1) >> 31); ws.VP &= ~f; ws.VN |= f; }   // forceUntil >= pos, as the sign bit of a difference (a compare would go through the vector pipe here)
		if (ws.score < r.minScore) { r.minScore = ws.score; r.minOffset = (uint32_t)pos; }
		if (flatRows > 0) {
			int32_t f = ws.score - popc64(ws.VP & ~flatMask) + popc64(ws.VN & ~flatMask);
			if (f < r.flatMin) { r.flatMin = f; r.flatOffset = (uint32_t)pos; }
		}
		if (MODE != 0) tables.colSet((uint32_t)pos, ws);
		HP |= hp << pos;
		HN |= hn << pos;
	}
	out.HP = HP; out.HN = HN;
	out.eVP = ws.VP; out.eVN = ws.VN; out.eScore = ws.score;
	return r;
}


// Full seed extension, wave layout. Trace goes to trace region `which` of the wave scratch (start cell first).
template <bool REGCOLS>
__device__ __forceinline__ uint32_t extendSeedWave(const DGraph& g, const CorrectnessTables& ct, const EqSource& eqSrc, int bandwidthCfg, lds_u32* lds, const WaveScratch& wsx,
	int len, uint32_t startNode, uint32_t startOffset, uint32_t which, uint32_t& nTrace, int32_t& score, ExtCounters& cnt)
{
	const LaneLdsT<REGCOLS> L { lds, wsx.lane, wsx.lanes, wsx.spillBase(), {}, wsx.regCap < 64 ? wsx.regCap : 64u, { 0, 0, 0, 0, 0 }, 0, 0 };
	uint32_t status = EXT_OK;
	nTrace = 0;
	score = 0;
	cnt.extensions++;
	cnt.flattenTie = 0;
	int numSlices = (len + 63) / 64;
	if ((uint32_t)numSlices + 1 > wsx.maxSlices) return EXT_OVERFLOW;
	// ---- initial slice (...Common.h:1243-1279)
	int buf = 0;   // table buffer `buf` = previous slice (LDS tables swap roles every slice; register tables rotate instead and stay 0/1)
	uint32_t nPrev = 1;
	{
		int nl = g.nodeLength[startNode];
		NodeItem it;
		it.node = startNode;
		it.sVP = it.sVN = it.eVP = it.eVN = 0;
		it.sScore = (int32_t)startOffset;
		it.eScore = nl - 1 - (int32_t)startOffset;
		it.minScore = 0;
		uint64_t upToOffset = startOffset >= 63 ? ~0ull : ((1ull << (startOffset + 1)) - 1);
		uint64_t nodeMask = nl >= 64 ? ~0ull : ((1ull << nl) - 1);
		it.HN = upToOffset & ~1ull;
		it.HP = nodeMask & ~upToOffset;
		storeItem(wsx, 0, it);
		L.pSet(0, 0, startNode, it.sScore, 0, it.HP, it.HN);
		WSlice s0;
		s0.minScore = 0; s0.minNode = startNode; s0.minOffset = startOffset; s0.first = 0; s0.count = 1; s0.bandwidth = 1; s0.j = -64; s0.flags = 1;
		storeSlice(wsx, 0, s0);
	}
	uint32_t nItems = 1, nSlices = 1;
	const bool storeCols = REGCOLS && wsx.maxCols > 0;
	uint32_t nCols = 0;   // columns in the column store so far
	GC_MARK_START();
	int32_t prevMinScore = 0, prevBandwidth = 1, prevJ = -64;
	double prevCorrect = ct.initCorrect, prevFalse = ct.initFalse;
	Eq4 eq;
	for (int slice = 0; slice < numSlices; slice++) {
		int j = prevJ + 64;
		eqVectorBits(eqSrc, len, j, eq);
		int32_t previousQuitScore = prevMinScore + prevBandwidth;
		int bandwidth = bandwidthCfg;
		int flatRows = (j + 64 > len) ? (len - j) : 0;
		const int cb = REGCOLS ? 1 : (buf ^ 1);   // table buffer `cb` = current slice
		auto prevFind = [&](uint32_t node) __attribute__((always_inline)) -> int { return L.find((uint32_t)(REGCOLS ? 0 : buf), nPrev, node); };
		uint32_t nPending = 0;
		auto pushEdge = [&](uint32_t target, WS incoming, bool skipFirst) __attribute__((always_inline)) {
			int found = L.find(2u, nPending, target);
			uint32_t slot = found >= 0 ? (uint32_t)found : nPending;
			WS add = incoming;
			if (!skipFirst) {
				int pi = prevFind(target);
				bool prevExists = pi >= 0;
				int32_t prevStart = prevExists ? L.pStart(buf, (uint32_t)pi) : 0;
				uint64_t hinP, hinN;
				if (prevExists) {
					int32_t before = wsBefore(incoming);
					if (prevStart < before) { hinP = 0; hinN = 1; }
					else if (prevStart > before) { hinP = 1; hinN = 0; }
					else { hinP = 0; hinN = 0; }
				} else { hinP = 1; hinN = 0; }
				NodeSeq nseq = loadNodeSeq(g, target);
				uint64_t hp, hn;
#if GC_LEAN_COLUMNS && defined(__HIP_DEVICE_COMPILE__)
				uint64_t eqFirst;
				if (REGCOLS && !nseq.ambiguous) {   // the first column's match mask by the same three scalar selects as in the column loop
					uint64_t lo, hi;
					const uint32_t code = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)nseq.w0);
					const uint64_t mA = gcUniform64(eq.a), mC = gcUniform64(eq.c), mG = gcUniform64(eq.g), mT = gcUniform64(eq.t);
					asm("s_bitcmp1_b32 %3, 0\n\ts_cselect_b64 %0, %5, %4\n\ts_cselect_b64 %1, %7, %6\n\ts_bitcmp1_b32 %3, 1\n\ts_cselect_b64 %2, %1, %0"
						: "=&s"(lo), "=&s"(hi), "=&s"(eqFirst) : "s"(code), "s"(mA), "s"(mC), "s"(mG), "s"(mT) : "scc");
				} else eqFirst = eqOfColumn(eq, nseq, 0);
				add = myersStep(eqFirst, incoming, hinP, hinN, hp, hn);
#else
				add = myersStep(eqOfColumn(eq, nseq, 0), incoming, hinP, hinN, hp, hn);
#endif
				if (!prevExists || wsBefore(add) < prevStart) { add.VP &= ~1ull; add.VN |= 1ull; }
			}
			if (slot == nPending) {
				if (nPending >= L.maxEntries()) { status = EXT_LDS_CAP; return; }
				L.qSet(slot, target, g.componentNumber[target], add);
				nPending++;
			} else {
#if GC_LEAN_MERGE && defined(__HIP_DEVICE_COMPILE__)
				L.qSetWs(slot, REGCOLS ? wsMergeWave(L.qWs(slot), add) : wsMerge(L.qWs(slot), add));
#else
				L.qSetWs(slot, wsMerge(L.qWs(slot), add));
#endif
			}
		};
		for (uint32_t i = 0; i < nPrev; i++) {
			if (j != 0 && L.pMin(buf, i) > previousQuitScore) continue;
			pushEdge(L.pNode(buf, i), wsSource(L.pStart(buf, i)), true);
		}
		if (status != EXT_OK) return status;
		GC_MARK(0);   // slice prologue: match masks + source pushes
		WSlice cur;
		cur.first = nItems; cur.count = 0; cur.bandwidth = bandwidth; cur.j = j;
		cur.minScore = INT32_MAX - bandwidth - 1; cur.minNode = 0xffffffffu; cur.minOffset = 0xffffffffu;
		int32_t flatMin = INT32_MAX; uint32_t flatNode = 0xffffffffu, flatOffset = 0xffffffffu;
		int32_t currentMin = cur.minScore;
		while (nPending > 0) {
			const uint32_t best = L.qArgMin(nPending);
			uint32_t pnode = L.qNode(best);
			WS pws = L.qWs(best);
			if (best != nPending - 1) L.qMove(best, nPending - 1);
			nPending--;
			if (nItems >= wsx.maxItems) return EXT_OVERFLOW;
			if (cur.count >= L.maxEntries()) return EXT_LDS_CAP;
			int pi = prevFind(pnode);
			bool prevExists = pi >= 0;
			NodeItem out;
			typename LaneLdsT<REGCOLS>::Entry pe { 0, 0, 0, ~0ull, 0ull };
			if (prevExists) pe = L.get((uint32_t)buf, (uint32_t)pi);
			GC_MARK(1);   // pop + previous-slice lookup
			const uint32_t tileLength = g.nodeLength[pnode];
			if (storeCols && nCols + tileLength - 1 > wsx.maxCols) return EXT_OVERFLOW;
			TileResult tr = storeCols ? computeTileW<2>(g, pnode, pws, prevExists, (int32_t)pe.w1, pe.a, pe.b, eq, out, L, flatRows, status)
				: computeTileW<0>(g, pnode, pws, prevExists, (int32_t)pe.w1, pe.a, pe.b, eq, out, L, flatRows, status);
			GC_MARK(2);   // tile columns
			if (status != EXT_OK) return status;
			out.minScore = tr.minScore;
			if (storeCols) {
				// columns 1 .. length-1 (VP, VN; column c sits in lane c) go to the column store in one coalesced 16 B-per-lane store; the item record
				// carries their offset in the slot of its minimum (the band tables hold that, the slab's copy is never read in this layout)
				if (threadIdx.x >= 1 && threadIdx.x < tileLength) {
					ulonglong2 v;
					v.x = (unsigned long long)L.cr[0] | ((unsigned long long)L.cr[1] << 32);
					v.y = (unsigned long long)L.cr[2] | ((unsigned long long)L.cr[3] << 32);
					*(ulonglong2*)(wsx.base + wsx.colBase() + 2ull * (nCols + threadIdx.x - 1)) = v;
				}
				NodeItem stored = out;
				stored.minScore = (int32_t)nCols;
				storeItem(wsx, nItems, stored);
				nCols += tileLength - 1;
			} else storeItem(wsx, nItems, out);
			L.pSet(cb, cur.count, pnode, out.sScore, out.minScore, out.HP, out.HN);
			nItems++;
			cur.count++;
			cnt.dpTiles++;
			cnt.columnSteps += g.nodeLength[pnode];
			if (flatRows > 0) { cnt.recomputeTiles++; cnt.columnSteps += g.nodeLength[pnode]; }
			if (tr.minScore > previousQuitScore + bandwidth + 128) return EXT_ASSERT;
			currentMin = tr.minScore < currentMin ? tr.minScore : currentMin;
			if (tr.minScore < cur.minScore) { cur.minScore = tr.minScore; cur.minNode = pnode; cur.minOffset = tr.minOffset; }
			if (flatRows > 0) {   // (as in extendSeedT: a second node AT the minimum is the tie gc_result::flatten_ties_long counts; bit 31 of flatOffset carries it to the slice's flags)
				if (tr.flatMin < flatMin) { flatMin = tr.flatMin; flatNode = pnode; flatOffset = tr.flatOffset; }
				else if (tr.flatMin == flatMin) flatOffset |= 0x80000000u;
			}
			WS newEnd = itemEnd(out);
#if GC_LEAN_COLUMNS && GC_LEAN_COLMIN && defined(__HIP_DEVICE_COMPILE__)
			int32_t newEndMin;
			if (REGCOLS) {
				// minimum over rows -1..63 of the tile's end column, all rows at once (lane r: the value of row r from two masked popcounts; wave
				// minimum by DPP) instead of a scalar loop over the column's runs of -1 deltas
				const uint64_t upTo = ~(~1ull << threadIdx.x);   // bits 0..r
				const int32_t before = wsBefore(newEnd);
				uint32_t v = (uint32_t)(before + popc64(newEnd.VP & upTo) - popc64(newEnd.VN & upTo));
				auto lower = [](uint32_t a, uint32_t b) { return (int32_t)a < (int32_t)b ? a : b; };
				v = lower(v, (uint32_t)__builtin_amdgcn_update_dpp(0x7fffffff, (int)v, 0x111, 0xf, 0xf, false));
				v = lower(v, (uint32_t)__builtin_amdgcn_update_dpp(0x7fffffff, (int)v, 0x112, 0xf, 0xf, false));
				v = lower(v, (uint32_t)__builtin_amdgcn_update_dpp(0x7fffffff, (int)v, 0x114, 0xf, 0xf, false));
				v = lower(v, (uint32_t)__builtin_amdgcn_update_dpp(0x7fffffff, (int)v, 0x118, 0xf, 0xf, false));
				v = lower(v, (uint32_t)__builtin_amdgcn_update_dpp(0x7fffffff, (int)v, 0x142, 0xa, 0xf, false));
				v = lower(v, (uint32_t)__builtin_amdgcn_update_dpp(0x7fffffff, (int)v, 0x143, 0xc, 0xf, false));
				const int32_t rowsMin = (int32_t)GC_READLANE(v, 63);
				newEndMin = rowsMin < before ? rowsMin : before;
			} else newEndMin = wsColumnMin(newEnd);
#else
			int32_t newEndMin = wsColumnMin(newEnd);
#endif
			if (newEndMin < prevMinScore) return EXT_ASSERT;
			if (newEndMin <= currentMin + bandwidth) {
				GC_MARK(3);   // item store + bookkeeping
				for (uint32_t e = g.outOff[pnode]; e < g.outOff[pnode + 1]; e++) {
					pushEdge(g.outAdj[e], newEnd, false);
					if (status != EXT_OK) return status;
				}
				GC_MARK(4);   // out-edge pushes
			} else GC_MARK(3);
		}
		GC_MARK(1);
		if (cur.count == 0) return EXT_ASSERT;
		const uint32_t flatTie = flatRows > 0 ? (flatOffset >> 31) << 3 : 0u;
		if (flatRows > 0) { cur.minScore = flatMin; cur.minNode = flatNode; cur.minOffset = flatOffset & 0x7fffffffu; }
		if (cur.minScore < prevMinScore) return EXT_ASSERT;
		double curCorrect, curFalse;
		{
			int mm = cur.minScore - prevMinScore;
			int idx = mm < 64 ? mm : 63;
			bool cfc = prevCorrect + ct.c2c >= prevFalse + ct.f2c;
			bool ffc = prevCorrect + ct.c2f >= prevFalse + ct.f2f;
			double a = prevCorrect + ct.c2c, b = prevFalse + ct.f2c;
			double c = prevCorrect + ct.c2f, d = prevFalse + ct.f2f;
			curCorrect = (a > b ? a : b) + ct.correctOdds[idx];
			curFalse = (c > d ? c : d) + ct.wrongOdds[idx];
			cur.flags = (curCorrect > curFalse ? 1u : 0u) | (cfc ? 2u : 0u) | (ffc ? 4u : 0u) | flatTie;
		}
		if (!(cur.flags & 2u)) break;
		storeSlice(wsx, nSlices++, cur);
		prevMinScore = cur.minScore; prevBandwidth = cur.bandwidth; prevJ = cur.j; prevCorrect = curCorrect; prevFalse = curFalse;
		nPrev = cur.count;
		if (REGCOLS) L.rotate(); else buf = cb;
		GC_MARK(5);
	}
	GC_MARK(5);   // slice epilogue (HMM, slice record)
	// removeWronglyAlignedEnd
	{
		bool currentlyCorrect = (loadSlice(wsx, nSlices - 1).flags & 1u) != 0;
		while (!currentlyCorrect) {
			currentlyCorrect = (loadSlice(wsx, nSlices - 1).flags & 4u) != 0;
			nSlices--;
			if (nSlices == 0) break;
		}
	}
	if (nSlices <= 1) return EXT_FAILED;
	WSlice last = loadSlice(wsx, nSlices - 1);
	if (last.minScore < 0 || last.minScore > len + 128) return EXT_ASSERT;
	score = last.minScore;

	// ---- backtrace. The LDS is reused for the recomputed columns of the current (slice, node).
	// REGCOLS: trace cells are collected 64 at a time in a register pair across the lanes (cell i of the current group in
	// lane i) and flushed with one coalesced 512 B store, instead of one 8 B store per cell from 64 identical lanes.
	uint32_t tbLo = 0, tbHi = 0;
	auto flushTrace = [&](uint32_t count) __attribute__((always_inline)) {   // the last `count` (1..64) cells pushed
		if (REGCOLS && count && threadIdx.x < count) wsx.base[(wsx.traceBase(nTrace - count, which)) * wsx.lanes + wsx.lane + threadIdx.x] = (unsigned long long)tbLo | ((unsigned long long)tbHi << 32);
	};
	auto pushTraceW = [&](Cell c, bool sw) __attribute__((always_inline)) -> bool {
#if GC_LEAN_WALK && GC_LEAN_PUSH && defined(__HIP_DEVICE_COMPILE__)
		if (REGCOLS) {
			// cell into lane (nTrace mod 64) of the register pair; the room check happens once per 64 cells, at the flush (a trace that
			// outgrows its buffer is noticed at the block's end or at the final flush instead of at the cell - same verdict, EXT_OVERFLOW)
			const unsigned long long cell = packCell(c, sw);
			const uint32_t slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)(nTrace & 63u));
			const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)cell), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(cell >> 32));
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
			asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0" : "+v"(tbLo), "+v"(tbHi) : "s"(slot), "s"(lo), "s"(hi) : "m0");
#pragma clang diagnostic pop
			nTrace++;
			if ((nTrace & 63u) == 0) {
				if (nTrace > wsx.maxTrace) { status = EXT_OVERFLOW; return false; }
				wsx.base[(wsx.traceBase(nTrace - 64, which)) * wsx.lanes + wsx.lane + threadIdx.x] = (unsigned long long)tbLo | ((unsigned long long)tbHi << 32);
			}
			return true;
		}
#endif
		if (nTrace >= wsx.maxTrace) { status = EXT_OVERFLOW; return false; }
		const unsigned long long cell = packCell(c, sw);
		if (REGCOLS) {
			const bool mine = threadIdx.x == (nTrace & 63u);
			tbLo = mine ? (uint32_t)cell : tbLo;
			tbHi = mine ? (uint32_t)(cell >> 32) : tbHi;
			nTrace++;
			if ((nTrace & 63u) == 0) flushTrace(64);
		} else {
			wsx.word(wsx.traceBase(nTrace, which)) = cell;
			nTrace++;
		}
		return true;
	};
	Cell here { last.minNode, last.minOffset, (last.j + 63 < len - 1) ? last.j + 63 : len - 1 };
	if (!pushTraceW(here, false)) return status;
	uint32_t curSliceIdx = 0xffffffffu, curNode = 0xffffffffu;
	WSlice cs = last, ps = last;
	// Node ids of the current and the previous slice, cached in the lane's LDS words behind the 64 columns: item lookups
	// ("is this in-neighbour in the band of slice s / s-1?") are the backtrace's most frequent question, and answering
	// them with a chain of dependent HBM loads per probe dominated the kernel. Slices with more than 64 nodes fall back
	// to the HBM scan.
	const uint32_t ID_CACHE = 64, idsCurBase = 64 * 5, idsPrevBase = 64 * 5 + ID_CACHE;
	static_assert(64 * 5 + 2 * 64 <= WAVE_WORDS, "id caches must fit behind the backtrace columns");
	auto fillIds = [&](const WSlice& sl, uint32_t base) __attribute__((always_inline)) {
		if (REGCOLS) {   // one strided load: lane i fetches the node of item i (a register-table slice has at most 64)
			uint32_t id = 0xffffffffu;
			if (threadIdx.x < sl.count) id = (uint32_t)wsx.base[(wsx.itemBase(sl.first + threadIdx.x) + 7) * wsx.lanes + wsx.lane];
			if (base == idsCurBase) L.idCur = id; else L.idPrev = id;
			return;
		}
		uint32_t n = sl.count < ID_CACHE ? sl.count : ID_CACHE;
		for (uint32_t i = 0; i < n; i++) L.stL(base + i, itemNode(wsx, sl.first + i));
	};
	auto findIn = [&](const WSlice& sl, uint32_t base, uint32_t node) __attribute__((always_inline)) -> int {
		if (REGCOLS) {
			unsigned long long m = base == idsCurBase ? __ballot(L.idCur == node) : __ballot(L.idPrev == node);
			if (sl.count < 64) m &= (1ull << sl.count) - 1;
			return m ? (int)(sl.first + (uint32_t)__ffsll((long long)m) - 1) : -1;
		}
		if (sl.count > ID_CACHE) return findItemW(wsx, sl, node);
		for (uint32_t i = 0; i < sl.count; i++) if (L.ldL(base + i) == node) return (int)(sl.first + i);
		return -1;
	};
	auto findCur = [&](uint32_t node) __attribute__((always_inline)) -> int { return findIn(cs, idsCurBase, node); };
	auto findPrev = [&](uint32_t node) __attribute__((always_inline)) -> int { return findIn(ps, idsPrevBase, node); };
	NodeItem curIt {};
	NodeItem prevIt {};
	bool prevItExists = false;
	// corner rule (pickBacktraceCorner, ...Common.h:710-804)
	auto corner = [&](Cell& out, bool& nodeSwitch) __attribute__((always_inline)) -> bool {
		GC_MARK(10);
		struct MarkOnExit { ExtCounters& cnt; __device__ ~MarkOnExit() { GC_MARK(9); } } markOnExit { cnt };   // bucket 9: corner rule
		int32_t j = cs.j;
		int32_t quitScore = cs.minScore + cs.bandwidth;
		int32_t previousQuitScore = ps.minScore + ps.bandwidth;
		int32_t scoreHere = wsValue(itemStart(curIt), 0);
		uint32_t inBegin = g.inOff[curNode], inEnd = g.inOff[curNode + 1];
		if (scoreHere > quitScore) {
			int32_t smallest = scoreHere + 1;
			out = Cell { 0, 0, 0 };
			nodeSwitch = false;
			if (prevItExists) { smallest = prevIt.sScore; out = Cell { curNode, 0, j - 1 }; }
			for (uint32_t e = inBegin; e < inEnd; e++) {
				uint32_t nb = g.inAdj[e];
				int p = findPrev(nb);
				if (p >= 0) { NodeItem pn = loadItem(wsx, (uint32_t)p); if (pn.eScore <= smallest) { smallest = pn.eScore; out = Cell { nb, (uint32_t)g.nodeLength[nb] - 1, j - 1 }; nodeSwitch = true; } }
				int c = findCur(nb);
				if (c >= 0 && nb != curNode) {
					int32_t v = wsValue(itemEnd(loadItem(wsx, (uint32_t)c)), 0);
					if (v < smallest) { smallest = v; out = Cell { nb, (uint32_t)g.nodeLength[nb] - 1, j }; nodeSwitch = true; }
				}
			}
			return true;
		}
		NodeSeq nseq = loadNodeSeq(g, curNode);
		int eqBit = (int)(eqOfColumn(eq, nseq, 0) & 1);
		if (prevItExists && prevIt.sScore == scoreHere - 1) { out = Cell { curNode, 0, j - 1 }; nodeSwitch = false; return true; }
		Cell bestInvalid { 0xffffffffu, 0xffffffffu, -1 };
		int32_t bestInvalidScore = scoreHere + 1;
		for (uint32_t e = inBegin; e < inEnd; e++) {
			uint32_t nb = g.inAdj[e];
			int c = findCur(nb);
			if (c >= 0 && wsValue(itemEnd(loadItem(wsx, (uint32_t)c)), 0) == scoreHere - 1) { out = Cell { nb, (uint32_t)g.nodeLength[nb] - 1, j }; nodeSwitch = true; return true; }
			int p = findPrev(nb);
			if (p >= 0) {
				int32_t cornerScore = loadItem(wsx, (uint32_t)p).eScore;
				if (cornerScore > previousQuitScore) {
					if (cornerScore < bestInvalidScore) { bestInvalidScore = cornerScore; bestInvalid = Cell { nb, (uint32_t)g.nodeLength[nb] - 1, j - 1 }; }
				} else if (cornerScore == scoreHere - (eqBit ? 0 : 1)) {
					out = Cell { nb, (uint32_t)g.nodeLength[nb] - 1, j - 1 }; nodeSwitch = true; return true;
				}
			}
		}
		if (bestInvalidScore < scoreHere + 1) { out = bestInvalid; nodeSwitch = true; return true; }
		return false;
	};
	while (here.seqPos != -1) {
		uint32_t s = (uint32_t)(here.seqPos / 64) + 1;
		if (s >= nSlices) return EXT_ASSERT;
		if (s != curSliceIdx || here.node != curNode) {
			GC_MARK(10);  // walking inside a tile / corner rules
			if (s != curSliceIdx) { cs = loadSlice(wsx, s); ps = loadSlice(wsx, s - 1); eqVectorBits(eqSrc, len, cs.j, eq); fillIds(cs, idsCurBase); fillIds(ps, idsPrevBase); }
			GC_MARK(6);   // backtrace: slice change
			curSliceIdx = s;
			curNode = here.node;
			int ci = findCur(curNode);
			if (ci < 0) return EXT_ASSERT;
			curIt = loadItem(wsx, (uint32_t)ci);
			int pi = findPrev(curNode);
			prevItExists = pi >= 0;
			if (prevItExists) prevIt = loadItem(wsx, (uint32_t)pi);
			GC_MARK(7);   // backtrace: item lookups + loads
			NodeItem scratchItem;
#if defined(__HIP_DEVICE_COMPILE__)
			if (storeCols) {
				// The DP kept this tile's columns: load them back (column c into lane c; column 0 is the item's start column), rebuild the column
				// scores from the item's bottom-row deltas (score of column c = start score + HP bits 1..c - HN bits 1..c) and make the walk masks -
				// the tile's column loop is not run a second time (45 % of the kernel's column steps were these recomputes). The preamble (merge with
				// the row above, first-row repair) is redone because the masks need its carries.
				TilePreamble pre;
				computeTileW<1>(g, curNode, itemStart(curIt), prevItExists, prevItExists ? prevIt.sScore : 0, prevItExists ? prevIt.HP : ~0ull, prevItExists ? prevIt.HN : 0ull,
					eq, scratchItem, L, 0, status, &pre);
				const uint32_t tileLength = g.nodeLength[curNode];
				const uint32_t c = threadIdx.x;
				unsigned long long vp = curIt.sVP, vn = curIt.sVN;
				if (c >= 1 && c < tileLength) {
					const ulonglong2 v = *(const ulonglong2*)(wsx.base + wsx.colBase() + 2ull * ((uint32_t)curIt.minScore + c - 1));
					vp = v.x; vn = v.y;
				}
				L.cr[0] = (uint32_t)vp; L.cr[1] = (uint32_t)(vp >> 32); L.cr[2] = (uint32_t)vn; L.cr[3] = (uint32_t)(vn >> 32);
				const uint64_t upToC = (c >= 63 ? ~0ull : ((2ull << c) - 1)) & ~1ull;   // bits 1..c
				L.cr[4] = (uint32_t)(curIt.sScore + popc64(curIt.HP & upToC) - popc64(curIt.HN & upToC));
				const NodeSeq seq = loadNodeSeq(g, curNode);
				L.walkMasks = false;
#if GC_LEAN_WALK
				if (!seq.ambiguous) L.setWalkMasks(seq.w0, seq.w1, eq, pre.forceEq, pre.prevHN);
#endif
			} else
#endif
			{
				computeTileW<1>(g, curNode, itemStart(curIt), prevItExists, prevItExists ? prevIt.sScore : 0, prevItExists ? prevIt.HP : ~0ull, prevItExists ? prevIt.HN : 0ull,
					eq, scratchItem, L, 0, status);
				if (scratchItem.eVP != curIt.eVP || scratchItem.eVN != curIt.eVN || scratchItem.eScore != curIt.eScore) status = EXT_ASSERT;
			}
			cnt.recomputeTiles++; cnt.backtraceTiles++; cnt.columnSteps += g.nodeLength[curNode];   // (the reference's units: it does recompute)
			if (status != EXT_OK) return status;
			GC_MARK(8);   // backtrace: column recompute
		}
		int row = here.seqPos & 63;
		if (row == 0 && here.offset == 0) {
			Cell nxt; bool sw;
			if (!corner(nxt, sw)) return EXT_ASSERT;
			if (!pushTraceW(nxt, sw)) return status;
			here = nxt;
			continue;
		}
		if (row == 0) {
			if (!prevItExists) {
				here = Cell { curNode, 0, here.seqPos };
				if (!pushTraceW(here, false)) return status;
				continue;
			}
			uint32_t off = here.offset;
			while (off > 0 && wsValue(L.col(off - 1), 0) == wsValue(L.col(off), 0) - 1) {
				off--;
				if (!pushTraceW(Cell { curNode, off, here.seqPos }, false)) return status;
			}
			here.offset = off;
			if (off == 0) {
				Cell nxt; bool sw;
				if (!corner(nxt, sw)) return EXT_ASSERT;
				if (!pushTraceW(nxt, sw)) return status;
				here = nxt;
				continue;
			}
			int32_t scoreHere = wsValue(L.col(off), 0);
			int32_t scoreDiagonal = prevIt.sScore;
			uint64_t lowMask = ((1ull << off) - 1) & ~1ull;
			scoreDiagonal += popc64(prevIt.HP & lowMask) - popc64(prevIt.HN & lowMask);
			int32_t scoreUp = scoreDiagonal + (int32_t)((prevIt.HP >> off) & 1) - (int32_t)((prevIt.HN >> off) & 1);
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
			if (!pushTraceW(nxt, false)) return status;
			here = nxt;
			continue;
		}
		if (here.offset == 0) {
			WS start = itemStart(curIt);
			int32_t sp = here.seqPos;
			while ((sp & 63) != 0 && (start.VP & (1ull << (sp & 63)))) {
				sp--;
				if (!pushTraceW(Cell { curNode, 0, sp }, false)) return status;
			}
			here.seqPos = sp;
			int offset = sp & 63;
			if (offset == 0) {
				Cell nxt; bool sw;
				if (!corner(nxt, sw)) return EXT_ASSERT;
				if (!pushTraceW(nxt, sw)) return status;
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
					int c = findCur(nb);
					if (c < 0) continue;
					WS ne = itemEnd(loadItem(wsx, (uint32_t)c));
					if (wsValue(ne, offset - 1) <= smallest) { smallest = wsValue(ne, offset - 1); nxt = Cell { nb, (uint32_t)g.nodeLength[nb] - 1, sp - 1 }; sw = true; }
					if (wsValue(ne, offset) < smallest && nb != curNode) { smallest = wsValue(ne, offset); nxt = Cell { nb, (uint32_t)g.nodeLength[nb] - 1, sp }; sw = true; }
				}
				found = true;
			} else {
				for (uint32_t e = g.inOff[curNode]; e < g.inOff[curNode + 1] && !found; e++) {
					uint32_t nb = g.inAdj[e];
					int c = findCur(nb);
					if (c < 0) continue;
					WS ne = itemEnd(loadItem(wsx, (uint32_t)c));
					if (wsValue(ne, offset) == scoreHere - 1) { nxt = Cell { nb, (uint32_t)g.nodeLength[nb] - 1, sp }; sw = true; found = true; }
					else if (wsValue(ne, offset - 1) == scoreHere - (eqBit ? 0 : 1)) { nxt = Cell { nb, (uint32_t)g.nodeLength[nb] - 1, sp - 1 }; sw = true; found = true; }
				}
			}
			if (!found) return EXT_ASSERT;
			if (!pushTraceW(nxt, sw)) return status;
			here = nxt;
			continue;
		}
#if GC_LEAN_WALK && defined(__HIP_DEVICE_COMPILE__)
		if (REGCOLS && L.walkMasks) {
			// (row masks of the tile's columns are in the lanes, see setWalkMasks: the walk only tests bits)
			uint32_t hori = here.offset;
			int vert = row;
			uint64_t up, diag, left;
			L.loadWalkMasks(hori, up, diag, left);
			uint32_t unfit = 0;   // a cell none of whose three predecessors fits: the reference's assertion; reported after the tile's walk
			while (hori > 0 && vert > 0) {
#if GC_LEAN_DIAGRUN
				// Most steps are diagonal, in runs: from (hori, vert) the walk goes diagonally as long as, in column hori - i, the cell of row vert - i
				// has no step up and a fitting diagonal predecessor. Column c keeps its masks in lane c, so lane c tests its own bit c - (hori - vert)
				// and one ballot shows the whole run; its cells (node, hori - 1 - k, row vert - 1 - k) are written by the lanes of the trace staging
				// registers that they fall into - one vector pass per run instead of ~25 scalar instructions per cell.
				{
					const int dgn = (int)hori - vert;
					const int myRow = (int)threadIdx.x - dgn;
					const uint64_t myUp = (uint64_t)L.cr[0] | ((uint64_t)L.cr[1] << 32), myDiag = (uint64_t)L.wm[0] | ((uint64_t)L.wm[1] << 32);
					const bool diagonalStep = threadIdx.x >= 1 && myRow >= 1 && myRow < 64 && ((myDiag >> (myRow & 63)) & 1ull) && !((myUp >> (myRow & 63)) & 1ull);
					const uint64_t runMask = __ballot(diagonalStep);
					const uint64_t stops = ~runMask & (hori >= 63 ? ~0ull : ((2ull << hori) - 1));   // columns <= hori where the run cannot continue (bit 0 always)
					const uint32_t firstStop = 63u - (uint32_t)__builtin_clzll(stops);
					const uint32_t run = hori - firstStop;
					if (run > 0) {
						uint32_t remaining = run, k0 = 0;
						while (remaining) {
							const uint32_t slot0 = nTrace & 63u;
							const uint32_t take = remaining < 64u - slot0 ? remaining : 64u - slot0;
							if (threadIdx.x >= slot0 && threadIdx.x < slot0 + take) {
								const uint32_t k = k0 + (threadIdx.x - slot0);
								const unsigned long long cell = packCell(Cell { curNode, hori - 1 - k, cs.j + vert - 1 - (int)k }, false);
								tbLo = (uint32_t)cell; tbHi = (uint32_t)(cell >> 32);
							}
							nTrace += take; remaining -= take; k0 += take;
							if ((nTrace & 63u) == 0) {
								if (nTrace > wsx.maxTrace) return EXT_OVERFLOW;
								wsx.base[(wsx.traceBase(nTrace - 64, which)) * wsx.lanes + wsx.lane + threadIdx.x] = (unsigned long long)tbLo | ((unsigned long long)tbHi << 32);
							}
						}
						hori -= run; vert -= (int)run;
						if (hori > 0 && vert > 0) L.loadWalkMasks(hori, up, diag, left);
						continue;
					}
				}
#endif
				const uint32_t u = (uint32_t)(up >> vert) & 1u, d = (uint32_t)(diag >> vert) & 1u, l = (uint32_t)(left >> vert) & 1u;
				unfit |= (u | d | l) ^ 1u;
				vert -= (int)(u | d);                                        // up: vertical == scoreHere - 1; else diagonal == scoreHere - (match ? 0 : 1)
				if (!u) { hori--; L.loadWalkMasks(hori, up, diag, left); }   // diagonal, or left (the left cell must then be one less)
				if (!pushTraceW(Cell { curNode, hori, cs.j + vert }, false)) return status;
			}
			if (unfit) return EXT_ASSERT;
			here = Cell { curNode, hori, cs.j + vert };
		} else if (REGCOLS) {
			// Tiles recomputed by the generic column loop (IUPAC nodes): the same three tests on cell values, all rows of a column pair at once.
			// One extension per wave: the walk inside a tile costs scalar instructions per cell (the kernel's bound), and its three tests compare
			// cell values of two neighbouring columns. The 64 lanes hold one row each, so for a column pair all rows are compared at once: lane r
			// computes value(r, hori) and value(r, hori - 1) from the stored columns (two masked popcounts), fetches value(r - 1, hori - 1) from its
			// neighbour lane (wave_shr), and four ballots give, for every row, "a step up is taken" (the column's VP itself), "the diagonal
			// predecessor fits with / without a match", "the left predecessor fits". The scalar side then only tests bits: 90 -> about 25
			// scalar instructions per cell. Same comparisons on the same values as the loop below (src/GraphAlignerBitvectorCommon.h:555-708).
			uint32_t hori = here.offset;
			int vert = row;
			const NodeSeq nseq = loadNodeSeq(g, curNode);
			const uint32_t laneRow = threadIdx.x;
			const uint64_t above = ~1ull << laneRow;   // rows below this lane's row in the word (bits > r)
			auto rowValues = [&](const WS& c) __attribute__((always_inline)) -> int32_t { return c.score - popc64(c.VP & above) + popc64(c.VN & above); };
			WS ch = L.col(hori);
			int32_t av = rowValues(ch);
			while (hori > 0 && vert > 0) {
				const WS cl = L.col(hori - 1);
				const int32_t bv = rowValues(cl);
				const int32_t bUp = __builtin_amdgcn_update_dpp(0, bv, 0x138, 0xf, 0xf, false);   // wave_shr:1: value(r - 1, hori - 1)
				const uint64_t diagSame = __ballot(bUp == av), diagLess = __ballot(bUp == av - 1), leftFits = __ballot(bv == av - 1);
				uint64_t eqColumn = eqOfColumn(eq, nseq, (int)hori);
				eqColumn = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)eqColumn) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(eqColumn >> 32)) << 32);   // (uniform; keeps the tests below on the scalar unit)
				const uint64_t diagFits = (diagSame & eqColumn) | (diagLess & ~eqColumn);
				bool moved = false;
				do {
					const uint64_t bit = 1ull << vert;
					if (ch.VP & bit) vert--;                                 // vertical == scoreHere - 1
					else if (diagFits & bit) { hori--; vert--; moved = true; }   // diagonal == scoreHere - (match ? 0 : 1)
					else {
						if (!(leftFits & bit)) return EXT_ASSERT;
						hori--;
						moved = true;
					}
					if (!pushTraceW(Cell { curNode, hori, cs.j + vert }, false)) return status;
				} while (!moved && vert > 0);
				if (moved) { ch = cl; av = bv; }
			}
			here = Cell { curNode, hori, cs.j + vert };
		} else
#endif
		{
			uint32_t hori = here.offset;
			int vert = row;
			NodeSeq nseq = loadNodeSeq(g, curNode);
			// Values are tracked incrementally: a step up changes a column's value by its vertical delta bit, so the two
			// columns are re-read (and one value recomputed from popcounts) only when the walk moves one column left.
			WS ch = L.col(hori), cl = L.col(hori - 1);
			int32_t scoreHere = wsValue(ch, vert), leftHere = wsValue(cl, vert);
			while (hori > 0 && vert > 0) {
				int32_t vertical = scoreHere - (int32_t)((ch.VP >> vert) & 1) + (int32_t)((ch.VN >> vert) & 1);
				int32_t diagonal = leftHere - (int32_t)((cl.VP >> vert) & 1) + (int32_t)((cl.VN >> vert) & 1);
				int eqBit = (int)((eqOfColumn(eq, nseq, (int)hori) >> vert) & 1);
				bool left = false;
				if (vertical == scoreHere - 1) { vert--; scoreHere = vertical; leftHere = diagonal; }
				else if (diagonal == scoreHere - (eqBit ? 0 : 1)) { hori--; vert--; scoreHere = diagonal; left = true; }
				else {
					if (leftHere != scoreHere - 1) return EXT_ASSERT;
					hori--;
					scoreHere = leftHere;
					left = true;
				}
				if (left) {
					ch = cl;
					if (hori > 0) { cl = L.col(hori - 1); leftHere = wsValue(cl, vert); }
				}
				if (!pushTraceW(Cell { curNode, hori, cs.j + vert }, false)) return status;
			}
			here = Cell { curNode, hori, cs.j + vert };
		}
	}
	{
		if (here.node != startNode) return EXT_ASSERT;
		uint32_t off = here.offset;
		while (true) {
			int32_t b = (int32_t)off - (int32_t)startOffset; if (b < 0) b = -b;
			int32_t bl = (int32_t)off - 1 - (int32_t)startOffset; if (bl < 0) bl = -bl;
			if (!(b != 0 && off > 0 && bl == b - 1)) break;
			off--;
			if (!pushTraceW(Cell { here.node, off, -1 }, false)) return status;
		}
	}
	if (REGCOLS && nTrace > wsx.maxTrace) return EXT_OVERFLOW;
	flushTrace(nTrace & 63u);
	cnt.traceItems += nTrace;
	cnt.flattenTie = (loadSlice(wsx, nSlices - 1).flags >> 3) & 1u;   // (see extendSeedT)
	GC_MARK(10);
	return status;
}

} // namespace gcdev
