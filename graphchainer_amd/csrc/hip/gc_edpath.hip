// The alignment PATH of (stitched chain path, read) - the second half of SURVEY.md §8 row f1.
//
// A read whose chained alignment wins (src/Aligner.cpp:901-905) gets its final trace from
// edlibAlign(pathseq, read, EDLIB_MODE_NW, EDLIB_TASK_PATH) (:845): the op string walked over the stitched path and the
// read (:848-876). The op string is not unique - which optimal alignment comes out is decided by edlib's
// obtainAlignment family (edlib/src/edlib.cpp:917-1419), and only by these three rules:
//   * a sub-problem whose stored matrix would be below 1 MB (20 bytes per 64-row block and column + 8 per column) is
//     traced back directly, from the bottom-right cell preferring UP (a path letter alone, op 1), then LEFT (a read base
//     alone, op 2), then the diagonal (0 match / 3 mismatch) (:917-1170, :1204-1212);
//   * a larger one is cut at the middle column of the target (the read): the split row is the first query row r =
//     0 .. Q-2 whose score in the left half's last column plus the score below-right of it in the reversed right half
//     equals the optimum, else the row -1, else Q-1; the halves recurse with those two scores (:1237-1419);
//   * an empty side is all inserts / all deletes (:1181-1189).
// Cells on optimal paths are exact inside edlib's Ukkonen band, so none of this depends on the band; the kernel computes
// whole columns. (The parity tests compare the ops with those of the real edlib compiled from the reference tree.)
//
// k_edit_path: one wave per (path, read) pair, depth-first over the sub-problems (left before right, so ops are appended
// in order). Rows = path letters, 64 per block; the column sweep is a skewed wavefront - lane l owns block 64s + l of
// strip s and works on column t - l at step t, the horizontal delta leaving block b is what block b + 1 needs one step
// later (one cross-lane shuffle per step); a strip's bottom deltas go through a byte array to the next strip. A leaf
// keeps every column's (VP, VN, score above the block) in HBM scratch and is walked back by the wave (uniform control
// flow, lane 0 stores the ops).
#include "gc_kernels.hpp"
#include <hip/hip_runtime.h>

namespace gcdev {

namespace {

struct EdPathScratch {   // one wave's HBM scratch
	uint64_t* peq;        // [blocks][4] exact-match masks of the current query segment
	uint8_t* carry;       // [columns] horizontal delta (+1 biased) under the strip just computed
	int32_t* colLeft;     // [Q + 1] last-column scores of the left half (index r + 1, [0] = row -1)
	int32_t* colRight;    // [Q + 1] the same for the reversed right half
	uint64_t* P;          // leaf: [column][block] vertical +1 bits
	uint64_t* M;          //       vertical -1 bits
	int32_t* top;         //       score of the row above the block
	uint8_t* tmpOps;      // a leaf's ops, last first
};

struct Seg { const char* p; uint32_t n; bool rev; __device__ __forceinline__ char at(uint32_t i) const { return rev ? p[n - 1 - i] : p[i]; } };

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// Column sweep of query (rows) against the first `cols` letters of target. colScores != nullptr: scores of the last
// column, colScores[r + 1] = D(r, cols - 1), [0] = cols. store: every column's block state goes to S.P / S.M / S.top.
#define ED_PATH_RING 4096u
__device__ void sweepColumns(const Seg& query, const Seg& target, uint32_t cols, const EdPathScratch& S, int32_t* colScores, bool store, uint8_t* ringChar, uint8_t* ringCarry, uint8_t* ringOut)
{
	const uint32_t lane = threadIdx.x;
	const uint32_t Q = query.n;
	const uint32_t nBlocks = (Q + 63) / 64;
	// exact-match masks of the query rows (letters other than A, C, G, T are compared one by one when a column carries one)
	for (uint32_t b = lane; b < nBlocks; b += 64) {
		uint64_t a = 0, c = 0, g = 0, t = 0;
		const uint32_t rows = Q - 64 * b < 64 ? Q - 64 * b : 64;
		for (uint32_t i = 0; i < rows; i++) {
			const char ch = query.at(64 * b + i);
			const uint64_t bit = 1ull << i;
			a |= ch == 'A' ? bit : 0; c |= ch == 'C' ? bit : 0; g |= ch == 'G' ? bit : 0; t |= ch == 'T' ? bit : 0;
		}
		S.peq[4 * b] = a; S.peq[4 * b + 1] = c; S.peq[4 * b + 2] = g; S.peq[4 * b + 3] = t;
	}
	__syncthreads();
	const uint32_t nStrips = (nBlocks + 63) / 64;
	for (uint32_t s = 0; s < nStrips; s++) {
		const uint32_t b = 64 * s + lane;
		const bool active = b < nBlocks;
		const uint32_t lanesHere = nBlocks - 64 * s < 64 ? nBlocks - 64 * s : 64;
		uint64_t eqA = 0, eqC = 0, eqG = 0, eqT = 0;
		if (active) { eqA = S.peq[4 * b]; eqC = S.peq[4 * b + 1]; eqG = S.peq[4 * b + 2]; eqT = S.peq[4 * b + 3]; }
		uint64_t VP = ~0ull, VN = 0;
		int32_t top = (int32_t)(64 * b);          // D(64b - 1, -1)
		int32_t houtPrev = 0;
		const bool carryOut = s + 1 < nStrips;
		const uint32_t steps = cols + lanesHere - 1;
		// Column letters (and, below the first strip, the deltas coming down from the strip above) go through LDS rings refilled 2048 ahead
		// every 1024 steps by all lanes: a global load per step sat on the step's dependent chain (~0.85 us per step measured).
		// The deltas leaving the strip's last block take the same route in the other direction: an LDS ring written every step, flushed to
		// the byte array at the refill points - a global store per step made every following step wait for it (the compiler guards the
		// store's registers with s_waitcnt vmcnt(0): ~1 us per step, 66 ms per 10 kb pair).
		uint32_t loadedEnd = 0, flushedEnd = 0;
		for (uint32_t t = 0; t < steps; t++) {
			if ((t & 1023u) == 0) {
				if (carryOut && t >= 64) {   // columns below t - 63 have left lane 63
					__syncthreads();
					for (uint32_t c = flushedEnd + lane; c < t - 63; c += 64) S.carry[c] = ringOut[c & (ED_PATH_RING - 1)];
					flushedEnd = t - 63;
				}
				const uint32_t end = t + 2048 < cols ? t + 2048 : cols;
				for (uint32_t c = loadedEnd + lane; c < end; c += 64) {
					ringChar[c & (ED_PATH_RING - 1)] = (uint8_t)target.at(c);
					if (s > 0) ringCarry[c & (ED_PATH_RING - 1)] = S.carry[c];
				}
				loadedEnd = end > loadedEnd ? end : loadedEnd;
				__syncthreads();
			}
			const int32_t fromAbove = __shfl_up(houtPrev, 1);
			const uint32_t j = t - lane;
			const char ch = (char)ringChar[j & (ED_PATH_RING - 1)];
			const uint8_t carryIn = ringCarry[j & (ED_PATH_RING - 1)];
			if (!active || t < lane || j >= cols) continue;
			int32_t hin = fromAbove;
			if (lane == 0) hin = s == 0 ? 1 : (int32_t)carryIn - 1;
			uint64_t Eq;
			if (ch == 'A') Eq = eqA; else if (ch == 'C') Eq = eqC; else if (ch == 'G') Eq = eqG; else if (ch == 'T') Eq = eqT;
			else {
				Eq = 0;
				const uint32_t rows = Q - 64 * b < 64 ? Q - 64 * b : 64;
				for (uint32_t i = 0; i < rows; i++) if (query.at(64 * b + i) == ch) Eq |= 1ull << i;
			}
			const uint64_t hinP = hin > 0 ? 1ull : 0ull, hinN = hin < 0 ? 1ull : 0ull;
			const uint64_t Xv = Eq | VN;
			Eq |= hinN;
			const uint64_t Xh = (((Eq & VP) + VP) ^ VP) | Eq;
			uint64_t Ph = VN | ~(Xh | VP);
			uint64_t Mh = VP & Xh;
			houtPrev = (int32_t)(Ph >> 63) - (int32_t)(Mh >> 63);
			Ph = (Ph << 1) | hinP;
			Mh = (Mh << 1) | hinN;
			VP = Mh | ~(Xv | Ph);
			VN = Ph & Xv;
			top += hin;
			if (store) {
				const uint64_t at = (uint64_t)b * cols + j;   // [block][column]: the walk back reads runs of columns of one block
				S.P[at] = VP; S.M[at] = VN; S.top[at] = top;
			}
			if (carryOut && lane == 63) ringOut[j & (ED_PATH_RING - 1)] = (uint8_t)(houtPrev + 1);
		}
		if (carryOut) {
			__syncthreads();
			for (uint32_t c = flushedEnd + lane; c < cols; c += 64) S.carry[c] = ringOut[c & (ED_PATH_RING - 1)];
		}
		if (colScores && active) {
			int32_t v = top;
			const uint32_t rows = Q - 64 * b < 64 ? Q - 64 * b : 64;
			for (uint32_t i = 0; i < rows; i++) {
				v += (int32_t)((VP >> i) & 1) - (int32_t)((VN >> i) & 1);
				colScores[64 * b + i + 1] = v;
			}
		}
		__syncthreads();   // the strip's carries (and stored columns) are complete before anything reads them
	}
	if (colScores && lane == 0) colScores[0] = (int32_t)cols;
	__syncthreads();
}

struct BlockCol { uint64_t P, M; int32_t top; };
__device__ __forceinline__ int32_t cellValue(const BlockCol& x, uint32_t i)   // row i (0..63) of the block
{
	const uint64_t low = i >= 63 ? ~0ull : ((1ull << (i + 1)) - 1);
	return x.top + __popcll(x.P & low) - __popcll(x.M & low);
}

// obtainAlignmentTraceback (edlib/src/edlib.cpp:917-1170) over the stored columns; returns the number of ops written to
// S.tmpOps, last op first. The walk is one dependent chain; what it reads is kept in registers across the wave: a window of
// 64 columns of the current block (column w0 - l in lane l, three coalesced loads per refill), read with v_readlane.
__device__ uint32_t walkBack(uint32_t Q, uint32_t T, const EdPathScratch& S)
{
	const uint32_t lane = threadIdx.x;
	int32_t r = (int32_t)Q - 1, c = (int32_t)T - 1;
	uint32_t n = 0;
	const bool writer = lane == 0;
	uint32_t wb = (uint32_t)r / 64;
	int32_t w0 = c;
	uint32_t wPlo = 0, wPhi = 0, wMlo = 0, wMhi = 0, wTop = 0;
	auto refill = [&]() __attribute__((always_inline)) {
		const int32_t col = w0 - (int32_t)lane;
		if (col >= 0) {
			const uint64_t at = (uint64_t)wb * T + (uint32_t)col;
			const uint64_t p = S.P[at], m = S.M[at];
			wPlo = (uint32_t)p; wPhi = (uint32_t)(p >> 32); wMlo = (uint32_t)m; wMhi = (uint32_t)(m >> 32); wTop = (uint32_t)S.top[at];
		}
	};
	auto column = [&](uint32_t idx) __attribute__((always_inline)) -> BlockCol {   // idx: uniform lane index
		BlockCol x;
		x.P = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)wPlo, (int)idx) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)wPhi, (int)idx) << 32);
		x.M = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)wMlo, (int)idx) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)wMhi, (int)idx) << 32);
		x.top = __builtin_amdgcn_readlane((int)wTop, (int)idx);
		return x;
	};
	refill();
	while (r >= 0 && c >= 0) {
		if (w0 - c > 62) { w0 = c; refill(); }   // column c - 1 must be in the window too
		const uint32_t idx = uni((uint32_t)(w0 - c));
		const BlockCol cur = column(idx);
		BlockCol lft { 0, 0, 0 };
		if (c > 0) lft = column(idx + 1);
		const uint32_t i = (uint32_t)r & 63u;
		const int32_t here = cellValue(cur, i);
		const int32_t up = i > 0 ? cellValue(cur, i - 1) : cur.top;
		const int32_t left = c > 0 ? cellValue(lft, i) : r + 1;
		const int32_t diag = c > 0 ? (i > 0 ? cellValue(lft, i - 1) : lft.top) : r;
		uint8_t op;
		bool goUp = false, goLeft = false;
		if (up + 1 == here) { op = 1; goUp = true; }
		else if (left + 1 == here) { op = 2; goLeft = true; }
		else { op = diag == here ? 0 : 3; goUp = goLeft = true; }
		if (writer) S.tmpOps[n] = op;
		n++;
		if (goUp) r--;
		if (goLeft) c--;
		if (r < 0 || c < 0) break;
		const uint32_t nb = (uint32_t)r / 64;
		if (nb != wb) { wb = nb; w0 = c; refill(); }
	}
	// on a border only one kind of move is left (:1043-1048,1067-1073)
	while (r >= 0) { if (writer) S.tmpOps[n] = 1; n++; r--; }
	while (c >= 0) { if (writer) S.tmpOps[n] = 2; n++; c--; }
	return n;
}

} // namespace

__global__ void __launch_bounds__(64) k_edit_path(const EdPathJob* __restrict__ jobs, uint32_t nJobs, const char* __restrict__ letters, const char* __restrict__ bases,
	uint8_t* __restrict__ scratch, uint64_t scratchBytes, uint32_t maxQ, uint32_t maxT, uint8_t* __restrict__ opsOut, uint32_t* __restrict__ opsLen)
{
	GC_RAISE_PRIO();
	__shared__ uint32_t stack[64][5];
	__shared__ uint8_t ringChar[ED_PATH_RING], ringCarry[ED_PATH_RING], ringOut[ED_PATH_RING];
	const uint32_t lane = threadIdx.x;
	EdPathScratch S;
	{
		uint8_t* p = scratch + (uint64_t)blockIdx.x * scratchBytes;
		const uint64_t maxBlocks = (maxQ + 63) / 64;
		S.peq = (uint64_t*)p;       p += maxBlocks * 4 * 8;
		S.colLeft = (int32_t*)p;    p += ((uint64_t)maxQ + 2) * 4;
		S.colRight = (int32_t*)p;   p += ((uint64_t)maxQ + 2) * 4;
		p = (uint8_t*)(((uintptr_t)p + 15) & ~(uintptr_t)15);
		S.P = (uint64_t*)p;         p += ED_PATH_LEAF_CELLS * 8;
		S.M = (uint64_t*)p;         p += ED_PATH_LEAF_CELLS * 8;
		S.top = (int32_t*)p;        p += ED_PATH_LEAF_CELLS * 4;
		S.carry = p;                p += ((uint64_t)maxT + 15) & ~15ull;
		S.tmpOps = p;
	}
	for (uint32_t ji = blockIdx.x; ji < nJobs; ji += gridDim.x) {
		const EdPathJob job = jobs[ji];
		const char* query = letters + job.queryOff;
		const char* target = bases + job.targetOff;
		uint8_t* out = opsOut + job.opsOff;
		uint32_t nOps = 0;
		bool failed = false;
		uint32_t sp = 0;
		if (job.queryLen > 0 && job.targetLen > 0) {   // edlibAlign returns no alignment for an empty side (edlib/src/edlib.cpp:133-152)
			if (lane == 0) { stack[0][0] = 0; stack[0][1] = job.queryLen; stack[0][2] = 0; stack[0][3] = job.targetLen; stack[0][4] = (uint32_t)job.best; }
			sp = 1;
		}
		__syncthreads();
		while (sp > 0 && !failed) {
			sp--;
			const uint32_t q0 = stack[sp][0], Q = stack[sp][1], t0 = stack[sp][2], T = stack[sp][3];
			const int32_t best = (int32_t)stack[sp][4];
			__syncthreads();
			if (Q == 0 || T == 0) {
				const uint32_t n = Q + T;
				const uint8_t op = Q == 0 ? 2 : 1;
				for (uint32_t i = lane; i < n; i += 64) out[nOps + i] = op;
				nOps += n;
				continue;
			}
			const uint64_t nBlocks = (Q + 63) / 64;
			const uint64_t dataSize = 20ull * nBlocks * T + 8ull * T;   // edlib/src/edlib.cpp:1204-1205
			const Seg qf { query + q0, Q, false }, tf { target + t0, T, false };
			if (dataSize < 1024ull * 1024ull) {
				sweepColumns(qf, tf, T, S, nullptr, true, ringChar, ringCarry, ringOut);
				const uint32_t n = walkBack(Q, T, S);
				__syncthreads();
				for (uint32_t i = lane; i < n; i += 64) out[nOps + i] = S.tmpOps[n - 1 - i];
				nOps += n;
				__syncthreads();
				continue;
			}
			const uint32_t leftW = T / 2, rightW = T - leftW;
			sweepColumns(qf, tf, leftW, S, S.colLeft, false, ringChar, ringCarry, ringOut);
			const Seg qr { query + q0, Q, true }, tr { target + t0, T, true };
			sweepColumns(qr, tr, rightW, S, S.colRight, false, ringChar, ringCarry, ringOut);
			// first row whose left score plus the score below-right of it is the optimum (:1339-1351), then the borders (:1353-1372)
			int32_t split = -2, leftScore = 0, rightScore = 0;
			for (uint32_t r0 = 0; r0 + 1 < Q; r0 += 64) {
				const uint32_t r = r0 + lane;
				const bool ok = r + 1 < Q && S.colLeft[r + 1] + S.colRight[Q - r - 1] == best;
				const unsigned long long m = __ballot(ok);
				if (m) { split = (int32_t)(r0 + (uint32_t)__ffsll((long long)m) - 1); break; }
			}
			if (split >= 0) { leftScore = S.colLeft[split + 1]; rightScore = S.colRight[Q - (uint32_t)split - 1]; }
			else if ((int32_t)leftW + S.colRight[Q] == best) { split = -1; leftScore = (int32_t)leftW; rightScore = S.colRight[Q]; }
			else if (S.colLeft[Q] + (int32_t)rightW == best) { split = (int32_t)Q - 1; leftScore = S.colLeft[Q]; rightScore = (int32_t)rightW; }
			else { failed = true; break; }
			leftScore = (int32_t)uni((uint32_t)leftScore); rightScore = (int32_t)uni((uint32_t)rightScore);
			const uint32_t ulHeight = (uint32_t)(split + 1);
			if (sp + 2 > 64) { failed = true; break; }
			__syncthreads();
			if (lane == 0) {
				stack[sp][0] = q0 + ulHeight; stack[sp][1] = Q - ulHeight; stack[sp][2] = t0 + leftW; stack[sp][3] = rightW; stack[sp][4] = (uint32_t)rightScore;
				stack[sp + 1][0] = q0; stack[sp + 1][1] = ulHeight; stack[sp + 1][2] = t0; stack[sp + 1][3] = leftW; stack[sp + 1][4] = (uint32_t)leftScore;
			}
			sp += 2;
			__syncthreads();
		}
		if (lane == 0) opsLen[ji] = failed ? 0u : nOps;   // a failed split leaves the reference with no alignment (status dropped at edlib/src/edlib.cpp:270)
		__syncthreads();
	}
}

uint64_t editPathScratchBytes(uint32_t maxQ, uint32_t maxT)
{
	const uint64_t maxBlocks = (maxQ + 63) / 64;
	uint64_t b = maxBlocks * 4 * 8 + 2 * ((uint64_t)maxQ + 2) * 4 + 16;
	b += (uint64_t)ED_PATH_LEAF_CELLS * 20;
	b += ((uint64_t)maxT + 15) & ~15ull;
	b += (uint64_t)maxQ + maxT + 64;
	return (b + 255) & ~255ull;
}

uint32_t editPathGridBlocks(uint32_t nJobs) { return nJobs < 6144 ? nJobs : 6144; }   // one pair is one long dependent chain: six waves per SIMD hide each other's latency (1.2 MB of scratch per wave)

void launchEditPath(hipStream_t stream, const EdPathJob* jobs, uint32_t nJobs, const char* letters, const char* bases, uint8_t* scratch, uint32_t maxQ, uint32_t maxT,
	uint8_t* opsOut, uint32_t* opsLen)
{
	if (!nJobs) return;
	hipLaunchKernelGGL(k_edit_path, dim3(editPathGridBlocks(nJobs)), dim3(64), 0, stream, jobs, nJobs, letters, bases, scratch, editPathScratchBytes(maxQ, maxT), maxQ, maxT, opsOut, opsLen);
}

} // namespace gcdev
