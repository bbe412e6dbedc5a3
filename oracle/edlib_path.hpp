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
// can touch lie on optimal paths, where band values are exact, so the result does not depend on the band and the
// restatement computes whole columns. edlibAlign itself returns no alignment when either string is empty (:133-152).
#pragma once
#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

namespace oracle {

enum : unsigned char { EDOP_MATCH = 0, EDOP_INSERT = 1, EDOP_DELETE = 2, EDOP_MISMATCH = 3 };   // edlib/include/edlib.h:47-50

namespace edpath {

// Scores of the last column of the NW matrix of query (rows) against target (columns): out[r + 1] = D(r, T - 1) for
// r = -1 .. Q - 1 (out[0] = boundary row). Multi-word Myers (the recurrence of calculateBlock, :395-432), whole columns.
// If `columns` is given (T x words each) the vertical delta vectors of every column are kept for the traceback.
struct Columns { std::vector<uint64_t> P, M; size_t words = 0; };
inline std::vector<int32_t> lastColumn(const unsigned char* query, size_t Q, const unsigned char* target, size_t T, Columns* columns = nullptr)
{
	const size_t words = (Q + 63) / 64;
	std::vector<uint64_t> peq(256 * words, 0);
	for (size_t i = 0; i < Q; i++) peq[(size_t)query[i] * words + i / 64] |= (uint64_t)1 << (i % 64);
	std::vector<uint64_t> VP(words, ~(uint64_t)0), VN(words, 0);
	if (columns) { columns->words = words; columns->P.assign(T * words, 0); columns->M.assign(T * words, 0); }
	for (size_t j = 0; j < T; j++) {
		uint64_t hinP = 1, hinN = 0;   // row -1 grows by one per column
		const uint64_t* eqRow = peq.data() + (size_t)target[j] * words;
		for (size_t w = 0; w < words; w++) {
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
		}
		if (columns) for (size_t w = 0; w < words; w++) { columns->P[j * words + w] = VP[w]; columns->M[j * words + w] = VN[w]; }
	}
	std::vector<int32_t> out(Q + 1);
	out[0] = (int32_t)T;
	for (size_t r = 0; r < Q; r++) out[r + 1] = out[r] + (int32_t)((VP[r / 64] >> (r % 64)) & 1) - (int32_t)((VN[r / 64] >> (r % 64)) & 1);
	return out;
}

// obtainAlignmentTraceback, edlib/src/edlib.cpp:917-1170, on whole stored columns.
inline void traceback(const unsigned char* query, size_t Q, const unsigned char* target, size_t T, std::vector<unsigned char>& ops)
{
	Columns cols;
	lastColumn(query, Q, target, T, &cols);
	// value of cell (r, c), r = -1 .. Q-1, c = -1 .. T-1, from the stored vertical deltas of column c
	std::vector<int32_t> colScore;   // scores of the current column and the one to its left, rebuilt when the walk changes column
	auto columnValues = [&](long long c, std::vector<int32_t>& v) {
		v.resize(Q + 1);
		if (c < 0) { for (size_t r = 0; r <= Q; r++) v[r] = (int32_t)r; return; }
		v[0] = (int32_t)c + 1;
		for (size_t r = 0; r < Q; r++) v[r + 1] = v[r] + (int32_t)((cols.P[(size_t)c * cols.words + r / 64] >> (r % 64)) & 1) - (int32_t)((cols.M[(size_t)c * cols.words + r / 64] >> (r % 64)) & 1);
	};
	std::vector<int32_t> cur, left;
	long long c = (long long)T - 1, r = (long long)Q - 1;
	columnValues(c, cur);
	columnValues(c - 1, left);
	std::vector<unsigned char> rev;
	while (r >= 0 || c >= 0) {
		if (c < 0) { rev.push_back(EDOP_INSERT); r--; continue; }          // left boundary: only query letters remain
		if (r < 0) { rev.push_back(EDOP_DELETE); c--; continue; }          // top boundary: only target letters remain
		const int32_t here = cur[r + 1], up = cur[r], lft = left[r + 1], diag = left[r];
		if (up + 1 == here) { rev.push_back(EDOP_INSERT); r--; }
		else if (lft + 1 == here) { rev.push_back(EDOP_DELETE); c--; cur.swap(left); columnValues(c - 1, left); }
		else { rev.push_back(diag == here ? EDOP_MATCH : EDOP_MISMATCH); r--; c--; cur.swap(left); columnValues(c - 1, left); }
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
	std::vector<int32_t> left = lastColumn(query, Q, target, leftHalfWidth);
	std::vector<unsigned char> rq(query, query + Q), rt(target, target + T);
	std::reverse(rq.begin(), rq.end());
	std::reverse(rt.begin(), rt.end());
	std::vector<int32_t> right = lastColumn(rq.data(), Q, rt.data(), rightHalfWidth);
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

// edlibAlign(query, target, NW, PATH): edit distance + op string (empty when a side is empty, edlib/src/edlib.cpp:133-152, or
// when the alignment could not be built - obtainAlignment's status is dropped at :270).
inline int32_t edlibPathNW(const std::string& query, const std::string& target, std::vector<unsigned char>& ops)
{
	ops.clear();
	if (query.empty() || target.empty()) return (int32_t)std::max(query.size(), target.size());
	const unsigned char* q = (const unsigned char*)query.data();
	const unsigned char* t = (const unsigned char*)target.data();
	const int32_t best = edpath::lastColumn(q, query.size(), t, target.size())[query.size()];
	if (!edpath::obtainAlignment(q, query.size(), t, target.size(), best, ops)) ops.clear();
	return best;
}

} // namespace oracle
