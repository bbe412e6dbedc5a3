// ORACLE (test infrastructure, not product code): CPU restatement of the reference's WordSlice.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// reference: src/WordSlice.h:150-753. A WordSlice is one DP column restricted to 64 read rows:
// VP/VN bit r = +1/-1 vertical delta between rows r-1 and r, scoreEnd = score of row 63.
// Parity: pinned. tests/test_oracle_units.py checks every function below against the reference's own
// WordSlice.h compiled unmodified into oracle/_ref/libref_units.so (random + edge-case columns).
#pragma once
#include <cstdint>
#include <algorithm>
#include <climits>

namespace oracle {

struct WordSlice {
	uint64_t VP = 0, VN = 0;
	int32_t scoreEnd = 0;
	WordSlice() {}
	WordSlice(uint64_t vp, uint64_t vn, int32_t s) : VP(vp), VN(vn), scoreEnd(s) {}
	bool operator==(const WordSlice& o) const { return VP == o.VP && VN == o.VN && scoreEnd == o.scoreEnd; }

	// reference: src/WordSlice.h:244-247
	int32_t getScoreBeforeStart() const { return scoreEnd - __builtin_popcountll(VP) + __builtin_popcountll(VN); }

	// reference: src/WordSlice.h:177-186. Score at row `row` (0..63).
	int32_t getValue(int row) const
	{
		uint64_t above = row < 63 ? (~(uint64_t)0 << (row + 1)) : 0;
		return scoreEnd + __builtin_popcountll(VN & above) - __builtin_popcountll(VP & above);
	}
};

// Pointwise minimum of two columns (including the row before the first one).
// reference: src/WordSlice.h:491-530 (mergeTwoSlices) + :555-653 (differenceMasksBitTwiddle).
// Own formulation: walk only the rows where the two columns' deltas differ, tracking the running
// difference a[r]-b[r]; rows in between keep whichever side is currently smaller (a on ties).
inline WordSlice mergeTwoSlices(const WordSlice& a, const WordSlice& b)
{
	int32_t d = a.getScoreBeforeStart() - b.getScoreBeforeStart();   // a - b at row -1
	uint64_t takeB = 0;            // rows where b is strictly smaller
	uint64_t fixP = 0, fixN = 0, fixMask = 0;   // explicit deltas at rows where the smaller side switches
	uint64_t diff = (a.VP ^ b.VP) | (a.VN ^ b.VN);
	int pos = 0;
	while (diff) {
		int r = __builtin_ctzll(diff);
		diff &= diff - 1;
		if (d > 0 && r > pos) takeB |= ((r >= 64 ? 0 : ((uint64_t)1 << r)) - 1) & ~(((uint64_t)1 << pos) - 1);
		int da = (int)((a.VP >> r) & 1) - (int)((a.VN >> r) & 1);
		int db = (int)((b.VP >> r) & 1) - (int)((b.VN >> r) & 1);
		int nd = d + da - db;
		bool before = d > 0, after = nd > 0;
		if (before != after) {
			int delta = after ? (db - d) : (da + d);   // res[r]-res[r-1] across the switch
			fixMask |= (uint64_t)1 << r;
			if (delta > 0) fixP |= (uint64_t)1 << r;
			if (delta < 0) fixN |= (uint64_t)1 << r;
		}
		if (after) takeB |= (uint64_t)1 << r;
		d = nd;
		pos = r + 1;
	}
	if (d > 0 && pos < 64) takeB |= ~(((uint64_t)1 << pos) - 1);
	WordSlice res;
	res.VP = (((a.VP & ~takeB) | (b.VP & takeB)) & ~fixMask) | fixP;
	res.VN = (((a.VN & ~takeB) | (b.VN & takeB)) & ~fixMask) | fixN;
	res.scoreEnd = std::min(a.scoreEnd, b.scoreEnd);
	return res;
}

// min over rows -1..63 of cur[r] where cur[r] < old[r]; INT_MAX if there is no such row.
// reference: src/WordSlice.h:252-259 (changedMinScore), cell-by-cell twin :292-301.
inline int32_t changedMinScore(const WordSlice& cur, const WordSlice& old)
{
	int32_t c = cur.getScoreBeforeStart(), o = old.getScoreBeforeStart();
	int32_t best = c < o ? c : INT_MAX;
	for (int r = 0; r < 64; r++) {
		c += (int)((cur.VP >> r) & 1) - (int)((cur.VN >> r) & 1);
		o += (int)((old.VP >> r) & 1) - (int)((old.VN >> r) & 1);
		if (c < o && c < best) best = c;
	}
	return best;
}

// One Myers bit-vector column step with horizontal carry-in (hinP/hinN = delta of the row above the
// word between the previous and the new column). reference: src/GraphAlignerBitvectorCommon.h:243-263
// (Myers 1999, p.405/408, the two-carry variant). Returns the new column and the carry-out of row 63.
struct StepResult { WordSlice ws; uint64_t houtP, houtN; };
inline StepResult getNextSlice(uint64_t Eq, WordSlice s, uint64_t hinP, uint64_t hinN)
{
	uint64_t Xv = Eq | s.VN;
	Eq |= hinN;
	uint64_t Xh = (((Eq & s.VP) + s.VP) ^ s.VP) | Eq;
	uint64_t Ph = s.VN | ~(Xh | s.VP);
	uint64_t Mh = s.VP & Xh;
	uint64_t shiftedMh = (Mh << 1) | hinN;
	uint64_t shiftedPh = (Ph << 1) | hinP;
	StepResult r;
	r.houtN = Mh >> 63;
	r.houtP = Ph >> 63;
	r.ws.VP = shiftedMh | ~(Xv | shiftedPh);
	r.ws.VN = shiftedPh & Xv;
	r.ws.scoreEnd = s.scoreEnd - (int32_t)r.houtN + (int32_t)r.houtP;
	return r;
}

// reference: src/GraphAlignerBitvectorCommon.h:806-810. Column entered from the slice above at score s.
inline WordSlice getSourceSliceFromScore(int32_t previousScore) { return WordSlice(~(uint64_t)0, 0, previousScore + 64); }

// reference: src/GraphAlignerBitvectorCommon.h:265-273. Keep only the first `row` rows.
inline WordSlice flattenWordSlice(WordSlice s, size_t row)
{
	uint64_t mask = ~(~(uint64_t)0 << row);
	s.scoreEnd -= __builtin_popcountll(s.VP & ~mask);
	s.scoreEnd += __builtin_popcountll(s.VN & ~mask);
	s.VP &= mask;
	s.VN &= mask;
	return s;
}

} // namespace oracle
