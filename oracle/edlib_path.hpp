// ORACLE (test infrastructure, not product code): restatement of the alignment PATH edlib returns for
// edlibAlign(query, target, EDLIB_MODE_NW, EDLIB_TASK_PATH), the call at src/Aligner.cpp:845 that turns the stitched chain
// into the read's final trace. edlib is vendored in the reference (edlib/src/edlib.cpp); the real thing is compiled
// unmodified into oracle/_ref and tests/test_oracle_units.py pins this restatement to it op for op.
//
// What decides the op string (all of it is in edlib's obtainAlignment family, none of it in the band bookkeeping):
//   * obtainAlignment, edlib/src/edlib.cpp:1175-1220: an empty side gives all-insert / all-delete; a problem whose stored
//     matrix would take less than 1 MB ((2*8 + 4) bytes per block and column + 8 per column) is traced back directly,
//     anything larger is split by Hirschberg.
//   * obtainAlignmentHirschberg, :1237-1419: the target is cut in the middle (left half = targetLength / 2 columns); the
//     left half's last column and the reversed right half's last column are computed; the split row is the FIRST query
//     row r (ascending, r = 0 .. queryLength - 2) whose left score plus the right score below it equals the optimum, then
//     the boundary row -1 (no query consumed left of the cut), then queryLength - 1 (:1339-1372); the two halves recurse
//     with their own scores.
//   * obtainAlignmentTraceback, :917-1170: from the bottom-right cell, a move UP (EDLIB_EDOP_INSERT, a query letter
//     alone) is taken whenever it is tight, else LEFT (EDLIB_EDOP_DELETE, a target letter alone), else the diagonal
//     (match when the scores are equal, mismatch otherwise).
// edlib computes its columns inside an Ukkonen band of half-width `bestScore`; the cells a tight move or an optimal split
// can touch lie on optimal paths, where band values are exact, so the result does not depend on the band's bookkeeping: the
// restatement uses a plain |row - column| <= bestScore band for the Hirschberg columns (as edlib's CPU cost is what the CPU
// baseline times) and whole columns for the small traced-back leaves. edlibAlign itself returns no alignment when either string
// is empty (:133-152).
#pragma once
#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

namespace oracle {

enum : unsigned char { EDOP_MATCH = 0, EDOP_INSERT = 1, EDOP_DELETE = 2, EDOP_MISMATCH = 3 };   // edlib/include/edlib.h:47-50

namespace edpath {

const int32_t INF_SCORE = 1 << 28;

// Scores of the last column of the NW matrix of query (rows) against target (columns): out[r + 1] = D(r, T - 1) for
// r = -1 .. Q - 1 (out[0] = boundary row). Multi-word Myers (the recurrence of calculateBlock, :395-432) inside Ukkonen's band
// |row - column| <= k (block granular): band values are upper bounds, exact on every path of cost <= k, which is all the
// callers look at; rows outside the band at the last column read INF_SCORE. k < 0: whole columns.
inline std::vector<int32_t> lastColumn(const unsigned char* query, size_t Q, const unsigned char* target, size_t T, long long k = -1)
{
	const size_t words = (Q + 63) / 64;
	std::vector<int32_t> out(Q + 1, INF_SCORE);
	out[0] = (int32_t)T;
	if (Q == 0 || T == 0) { for (size_t r = 0; r < Q && T == 0; r++) out[r + 1] = (int32_t)r + 1; return out; }
	// match masks of the letters that occur (A, C, G, T and whatever else the strings hold)
	std::vector<int> slot(256, -1);
	std::vector<uint64_t> peq;
	auto masksOf = [&](unsigned char c) -> const uint64_t* {
		if (slot[c] < 0) {
			slot[c] = (int)(peq.size() / words);
			peq.resize(peq.size() + words, 0);
			uint64_t* m = peq.data() + (size_t)slot[c] * words;
			for (size_t i = 0; i < Q; i++) if (query[i] == c) m[i / 64] |= (uint64_t)1 << (i % 64);
		}
		return peq.data() + (size_t)slot[c] * words;
	};
	peq.reserve(8 * words);
	for (size_t j = 0; j < T; j++) if (slot[target[j]] < 0) { peq.reserve(peq.size() + words); masksOf(target[j]); }
	const long long band = k < 0 ? (long long)(Q + T) : k;
	std::vector<uint64_t> VP(words, ~(uint64_t)0), VN(words, 0);
	std::vector<int32_t> score(words);   // value of the block's last row
	size_t first = 0, last = (size_t)std::min<long long>((long long)words - 1, band / 64);
	for (size_t b = 0; b <= last; b++) score[b] = (int32_t)(64 * (b + 1));
	auto step = [&](size_t b, const uint64_t* eqRow, int hin) -> int {
		uint64_t Eq = eqRow[b];
		const uint64_t vp = VP[b], vn = VN[b];
		const uint64_t hinP = hin > 0, hinN = hin < 0;
		const uint64_t Xv = Eq | vn;
		Eq |= hinN;
		const uint64_t Xh = (((Eq & vp) + vp) ^ vp) | Eq;
		uint64_t Ph = vn | ~(Xh | vp);
		uint64_t Mh = vp & Xh;
		const int hout = (int)(Ph >> 63) - (int)(Mh >> 63);
		Ph = (Ph << 1) | hinP;
		Mh = (Mh << 1) | hinN;
		VP[b] = Mh | ~(Xv | Ph);
		VN[b] = Ph & Xv;
		return hout;
	};
	for (size_t j = 0; j < T; j++) {
		const uint64_t* eqRow = masksOf(target[j]);
		int hout = 1;   // row -1 (or the row above the band) grows by one per column
		for (size_t b = first; b <= last; b++) { hout = step(b, eqRow, hout); score[b] += hout; }
		const size_t wantLast = (size_t)std::min<long long>((long long)words - 1, ((long long)j + band) / 64);
		while (last < wantLast) {   // a block entering the band: +1 per row below its upper neighbour, then this column's step (cf. :806-813)
			last++;
			VP[last] = ~(uint64_t)0; VN[last] = 0;
			const int newHout = step(last, eqRow, hout);
			score[last] = score[last - 1] - hout + 64 + newHout;
			hout = newHout;
		}
		const long long lowRow = (long long)j + 1 - band;
		if (lowRow > 0) first = std::max(first, (size_t)std::min<long long>((long long)last, lowRow / 64));
	}
	for (size_t b = first; b <= last; b++) {
		int32_t v = score[b];
		for (int i = 63; i >= 0; i--) {
			const size_t r = 64 * b + (size_t)i;
			if (r < Q) out[r + 1] = v;
			v -= (int32_t)((VP[b] >> i) & 1) - (int32_t)((VN[b] >> i) & 1);
		}
	}
	return out;
}

// obtainAlignmentTraceback, edlib/src/edlib.cpp:917-1170: whole columns are kept (a leaf is at most 1 MB by construction) as
// vertical delta words plus the value above every block, so a cell is two popcounts away.
inline void traceback(const unsigned char* query, size_t Q, const unsigned char* target, size_t T, std::vector<unsigned char>& ops)
{
	const size_t words = (Q + 63) / 64;
	std::vector<uint64_t> peq(256 * words, 0);
	for (size_t i = 0; i < Q; i++) peq[(size_t)query[i] * words + i / 64] |= (uint64_t)1 << (i % 64);
	std::vector<uint64_t> P(T * words), M(T * words);
	std::vector<int32_t> top(T * words);   // value of the row above the block
	{
		std::vector<uint64_t> VP(words, ~(uint64_t)0), VN(words, 0);
		std::vector<int32_t> above(words);
		for (size_t b = 0; b < words; b++) above[b] = (int32_t)(64 * b);
		for (size_t j = 0; j < T; j++) {
			uint64_t hinP = 1, hinN = 0;
			const uint64_t* eqRow = peq.data() + (size_t)target[j] * words;
			for (size_t w = 0; w < words; w++) {
				above[w] += (int32_t)hinP - (int32_t)hinN;
				uint64_t Eq = eqRow[w];
				const uint64_t vp = VP[w], vn = VN[w];
				const uint64_t Xv = Eq | vn;
				Eq |= hinN;
				const uint64_t Xh = (((Eq & vp) + vp) ^ vp) | Eq;
				uint64_t Ph = vn | ~(Xh | vp);
				uint64_t Mh = vp & Xh;
				const uint64_t outP = Ph >> 63, outN = Mh >> 63;
				Ph = (Ph << 1) | hinP;
				Mh = (Mh << 1) | hinN;
				VP[w] = Mh | ~(Xv | Ph);
				VN[w] = Ph & Xv;
				hinP = outP; hinN = outN;
				P[j * words + w] = VP[w]; M[j * words + w] = VN[w]; top[j * words + w] = above[w];
			}
		}
	}
	auto cell = [&](long long r, long long c) -> int32_t {   // r = -1 .. Q-1, c = -1 .. T-1
		if (c < 0) return (int32_t)(r + 1);
		if (r < 0) return (int32_t)(c + 1);
		const size_t at = (size_t)c * words + (size_t)r / 64;
		const unsigned i = (unsigned)r % 64;
		const uint64_t low = i == 63 ? ~(uint64_t)0 : (((uint64_t)1 << (i + 1)) - 1);
		return top[at] + __builtin_popcountll(P[at] & low) - __builtin_popcountll(M[at] & low);
	};
	long long c = (long long)T - 1, r = (long long)Q - 1;
	std::vector<unsigned char> rev;
	while (r >= 0 || c >= 0) {
		if (c < 0) { rev.push_back(EDOP_INSERT); r--; continue; }          // left border: only query letters remain
		if (r < 0) { rev.push_back(EDOP_DELETE); c--; continue; }          // top border: only target letters remain
		const int32_t here = cell(r, c);
		if (cell(r - 1, c) + 1 == here) { rev.push_back(EDOP_INSERT); r--; }
		else if (cell(r, c - 1) + 1 == here) { rev.push_back(EDOP_DELETE); c--; }
		else { rev.push_back(cell(r - 1, c - 1) == here ? EDOP_MATCH : EDOP_MISMATCH); r--; c--; }
	}
	ops.insert(ops.end(), rev.rbegin(), rev.rend());
}

// obtainAlignment, edlib/src/edlib.cpp:1175-1220 + obtainAlignmentHirschberg :1237-1419. Returns false where edlib returns
// EDLIB_STATUS_ERROR (no split found).
inline bool obtainAlignment(const unsigned char* query, size_t Q, const unsigned char* target, size_t T, int32_t bestScore, std::vector<unsigned char>& ops)
{
	if (Q == 0 || T == 0) {
		ops.insert(ops.end(), Q + T, Q == 0 ? (unsigned char)EDOP_DELETE : (unsigned char)EDOP_INSERT);
		return true;
	}
	const long long maxNumBlocks = (long long)((Q + 63) / 64);
	const long long alignmentDataSize = (2ll * 8 + 4) * maxNumBlocks * (long long)T + 2ll * 4 * (long long)T;   // :1204-1205
	if (alignmentDataSize < 1024 * 1024) {
		traceback(query, Q, target, T, ops);
		return true;
	}
	const size_t leftHalfWidth = T / 2, rightHalfWidth = T - leftHalfWidth;
	std::vector<int32_t> left = lastColumn(query, Q, target, leftHalfWidth, bestScore);
	std::vector<unsigned char> rq(query, query + Q), rt(target, target + T);
	std::reverse(rq.begin(), rq.end());
	std::reverse(rt.begin(), rt.end());
	std::vector<int32_t> right = lastColumn(rq.data(), Q, rt.data(), rightHalfWidth, bestScore);
	// right[x] = cost of aligning the last x... rows: reversed row index x - 1 = original row Q - x; the cell below-right of (r, cut) is
	// original row r + 1, i.e. the suffix query[r+1 ..] of length Q - r - 1 -> right[Q - r - 1]
	long long split = -2;
	int32_t leftScore = -1, rightScore = -1;
	for (long long r = 0; r + 1 < (long long)Q; r++) {
		if (left[r + 1] + right[Q - r - 1] == bestScore) { split = r; leftScore = left[r + 1]; rightScore = right[Q - r - 1]; break; }
	}
	if (split == -2 && (int32_t)leftHalfWidth + right[Q] == bestScore) { split = -1; leftScore = (int32_t)leftHalfWidth; rightScore = right[Q]; }
	if (split == -2 && left[Q] + (int32_t)rightHalfWidth == bestScore) { split = (long long)Q - 1; leftScore = left[Q]; rightScore = (int32_t)rightHalfWidth; }
	if (split == -2) return false;
	const size_t ulHeight = (size_t)(split + 1);
	if (!obtainAlignment(query, ulHeight, target, leftHalfWidth, leftScore, ops)) return false;
	return obtainAlignment(query + ulHeight, Q - ulHeight, target + leftHalfWidth, rightHalfWidth, rightScore, ops);
}

} // namespace edpath

// The NW edit distance the way edlib finds it (edlib/src/edlib.cpp:176-194): band half-width k = 64, doubled until the result fits.
inline int32_t editDistanceBanded(const std::string& a, const std::string& b)
{
	if (a.empty() || b.empty()) return (int32_t)std::max(a.size(), b.size());
	const long long limit = (long long)std::max(a.size(), b.size());
	for (long long k = 64;; k *= 2) {
		if (k >= (long long)(a.size() > b.size() ? a.size() - b.size() : b.size() - a.size())) {
			const int32_t d = edpath::lastColumn((const unsigned char*)a.data(), a.size(), (const unsigned char*)b.data(), b.size(), k)[a.size()];
			if (d <= k) return d;
		}
		if (k >= limit) return edpath::lastColumn((const unsigned char*)a.data(), a.size(), (const unsigned char*)b.data(), b.size(), -1)[a.size()];
	}
}

// edlibAlign(query, target, NW, PATH): edit distance + op string (empty when a side is empty, edlib/src/edlib.cpp:133-152, or
// when the alignment could not be built - obtainAlignment's status is dropped at :270).
inline int32_t edlibPathNW(const std::string& query, const std::string& target, std::vector<unsigned char>& ops)
{
	ops.clear();
	if (query.empty() || target.empty()) return (int32_t)std::max(query.size(), target.size());
	const unsigned char* q = (const unsigned char*)query.data();
	const unsigned char* t = (const unsigned char*)target.data();
	const int32_t best = editDistanceBanded(query, target);
	if (!edpath::obtainAlignment(q, query.size(), t, target.size(), best, ops)) ops.clear();
	return best;
}

} // namespace oracle
