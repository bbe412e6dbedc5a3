// Path sequences and global (NW) edit distances on the GPU - SURVEY.md §8 row f1.
//
// The reference decides between the whole-read alignment and the chained alignment by two edit distances per read:
// edlibAlign(pathseq, read, EDLIB_MODE_NW) at src/Aligner.cpp:645 (path of the best whole-read alignment,
// traceToSequence :376-408,425-428) and at :845 (path of the stitched chain). edlib itself (edlib/src/edlib.cpp,
// Myers' bit-vector algorithm with Ukkonen's band) is a third-party component; only the VALUE it returns matters here,
// and the global edit distance of two strings is unique, so any exact algorithm is a drop-in for it.
//
// k_edit_distance: one wave per (path, read) pair. Rows = read bases, 64 per 64-bit word ("block"), columns = path
// letters. Lane l owns the units l, l+64, l+128, ... (unit = G consecutive blocks = 64G rows) and the wave sweeps the
// matrix as a skewed wavefront: at step t unit B works on column t - B, so the horizontal delta leaving unit B-1
// at column j is exactly what unit B needs one step later (one cross-lane shuffle per step). Only the cells with
// Ukkonen's corridor are computed (the diagonals d = row - column with |d| + |(n - m) - d| <= k, about k of them): unit B is active
// for columns [64G*B - reachRight, 64G*B + 64G - 1 + reachLeft], enters
// with all-+1 vertical deltas below its upper neighbour and gets +1 as incoming horizontal delta when it is the top
// of the band. Band values are upper bounds and exact along every path of cost <= k, so a result <= k is the edit
// distance; otherwise k doubles and the pass repeats. k < 4000*G keeps a lane's consecutive units disjoint in time
// (no limit when the read has at most 64 units, i.e. up to 4096*G rows); the host escalates G = 1, 2, 4, 8, 16.
#include "gc_kernels.hpp"
#include <hip/hip_runtime.h>

namespace gcdev {

__device__ __forceinline__ char nodeLetter(const DGraph& g, uint32_t node, uint32_t pos)   // AlignmentGraph::NodeSequences, src/AlignmentGraph.cpp:751-796
{
	if (node < g.firstAmbiguous) {
		uint64_t w = g.nodeSeq[2 * (size_t)node + (pos >> 5)];
		const char acgt[4] = { 'A', 'C', 'G', 'T' };
		return acgt[(w >> ((pos & 31) * 2)) & 3];
	}
	const uint64_t* p = g.ambSeq + 4 * (size_t)(node - g.firstAmbiguous);
	uint32_t mask = (uint32_t)((p[0] >> pos) & 1) | (uint32_t)(((p[1] >> pos) & 1) << 1) | (uint32_t)(((p[2] >> pos) & 1) << 2) | (uint32_t)(((p[3] >> pos) & 1) << 3);
	const char iupac[16] = { 'N', 'A', 'C', 'M', 'G', 'R', 'S', 'V', 'T', 'W', 'Y', 'H', 'K', 'D', 'B', 'N' };
	return iupac[mask];
}

__device__ __forceinline__ uint32_t waveInclusiveScan(uint32_t v, uint32_t lane)
{
	for (int d = 1; d < 64; d <<= 1) {
		uint32_t o = __shfl_up(v, d);
		if ((int)lane >= d) v += o;
	}
	return v;
}

// traceToSequence (src/Aligner.cpp:376-408,425-428) of one alignment per read: the letters of the path between the first
// and the last trace cell. Cell j contributes the graph bases between the previous cell's position and its own, so the
// counts are independent, a wave scan places them, and every lane copies its own run.
__global__ void __launch_bounds__(64) k_long_pathseq(DGraph g, const PathSeqJob* __restrict__ jobs, uint32_t nJobs, const LongCell* __restrict__ cellPool,
	char* __restrict__ letters, uint32_t* __restrict__ outLen)
{
	GC_RAISE_ED_PRIO();
	const uint32_t r = blockIdx.x, lane = threadIdx.x;
	if (r >= nJobs) return;
	const PathSeqJob job = jobs[r];
	if (job.count == 0) { if (lane == 0) outLen[r] = 0; return; }
	const LongCell* cells = cellPool + job.srcOff;
	char* out = letters + job.outOff;
	uint32_t total = 0;
	bool overflow = false;
	for (uint32_t base = 0; base < job.count; base += 64) {
		const uint32_t j = base + lane;
		uint32_t cnt = 0, node = 0, off = 0, prevNode = 0, prevOff = 0;
		if (j < job.count) {
			LongCell c = cells[j];
			node = g.lookup[g.lookupOff[c.node] + c.offset / 64];
			off = c.offset & 63u;
			if (j == 0) cnt = 1;
			else {
				LongCell p = cells[j - 1];
				prevNode = g.lookup[g.lookupOff[p.node] + p.offset / 64];
				prevOff = p.offset & 63u;
				if (node == prevNode) cnt = off > prevOff ? off - prevOff : 0;
				else cnt = ((uint32_t)g.nodeLength[prevNode] - (prevOff + 1)) + (off + 1);
			}
		}
		uint32_t incl = waveInclusiveScan(cnt, lane);
		uint32_t at = total + incl - cnt;
		if (at + cnt > job.outCap) overflow = true;
		else if (cnt) {
			uint32_t w = at;
			if (j == 0) out[w++] = nodeLetter(g, node, off);
			else if (node == prevNode) { for (uint32_t o = prevOff + 1; o <= off; o++) out[w++] = nodeLetter(g, node, o); }
			else {
				for (uint32_t o = prevOff + 1; o < g.nodeLength[prevNode]; o++) out[w++] = nodeLetter(g, prevNode, o);
				for (uint32_t o = 0; o <= off; o++) out[w++] = nodeLetter(g, node, o);
			}
		}
		total += __shfl(incl, 63);
	}
	if (__any(overflow)) total = 0xffffffffu;
	if (lane == 0) outLen[r] = total;
}

// pathToTrace (src/Aligner.cpp:409-424) of the stitched chain, as letters: first node from its start offset, last node
// (when it is not also the first) up to its last offset, whole nodes in between.
__global__ void __launch_bounds__(64) k_chain_pathseq(DGraph g, const PathSeqJob* __restrict__ jobs, uint32_t nJobs, const uint32_t* __restrict__ pathNodes,
	const uint32_t* __restrict__ altNodes, char* __restrict__ letters, uint32_t* __restrict__ outLen)
{
	GC_RAISE_ED_PRIO();
	const uint32_t r = blockIdx.x, lane = threadIdx.x;
	if (r >= nJobs) return;
	const PathSeqJob job = jobs[r];
	if (job.count == 0) { if (lane == 0) outLen[r] = 0; return; }
	const uint32_t* nodes = (job.srcOff >> 63) ? altNodes + (job.srcOff & ~(1ull << 63)) : pathNodes + job.srcOff;   // bit 63: stitched on the host
	char* out = letters + job.outOff;
	uint32_t total = 0;
	bool overflow = false;
	for (uint32_t base = 0; base < job.count; base += 64) {
		const uint32_t i = base + lane;
		uint32_t S = 0, L = 0, node = 0;
		if (i < job.count) {
			node = nodes[i];
			L = g.nodeLength[node];
			if (i == 0) S = job.firstOffset;
			else if (i == job.count - 1) L = job.lastOffset + 1;
		}
		uint32_t cnt = L > S ? L - S : 0;
		uint32_t incl = waveInclusiveScan(cnt, lane);
		uint32_t at = total + incl - cnt;
		if (at + cnt > job.outCap) overflow = true;
		else for (uint32_t o = S; o < L; o++) out[at + (o - S)] = nodeLetter(g, node, o);
		total += __shfl(incl, 63);
	}
	if (__any(overflow)) total = 0xffffffffu;
	if (lane == 0) outLen[r] = total;
}

#define ED_RING 8192u

template <int G>
__global__ void __launch_bounds__(64) k_edit_distance(const EdPair* __restrict__ pairs, uint32_t nPairs, const EdRead* __restrict__ reads, const char* __restrict__ bases,
	const uint64_t* __restrict__ eqMasks, const char* __restrict__ letters, const uint32_t* __restrict__ lettersLen, int64_t* __restrict__ outDistance)
{
	GC_RAISE_ED_PRIO();
	__shared__ uint8_t ring[ED_RING];
	__shared__ int32_t resultSlot;
	const uint32_t lane = threadIdx.x;
	constexpr uint32_t RB = 64u * G;                 // rows per unit
	constexpr uint32_t K_MAX = 4000u * G;             // the corridor is about k diagonals wide: a lane's consecutive units (64 units = 4096 G rows apart) stay disjoint in time
	for (uint32_t pi = blockIdx.x; pi < nPairs; pi += gridDim.x) {
		const EdPair pair = pairs[pi];
		const EdRead rd = reads[pair.read];
		const uint32_t n = rd.len;
		const uint32_t m = lettersLen ? lettersLen[pair.lenIndex] : pair.m;
		if (m == 0xffffffffu) { if (lane == 0) outDistance[pi] = -3; continue; }   // path letters overflowed their slot
		if (n == 0 || m == 0) { if (lane == 0) outDistance[pi] = (int64_t)(n + m); continue; }
		const char* path = letters + pair.lettersOff;
		const uint64_t* masks = eqMasks + rd.eqOff;
		const uint32_t nU = (n + RB - 1) / RB;
		const uint32_t diff = n > m ? n - m : m - n;
		const uint32_t cap = n > m ? n : m;
		uint32_t k = pair.k > diff ? pair.k : diff;
		if (k < 1) k = 1;
		if (k > cap) k = cap;
		int64_t answer = -2;                         // -2: this G cannot hold the band, the host retries with a larger one
		// the letter ring is refilled every 1024 steps up to column t + 2048, so it holds the columns from t + 2048 - ED_RING on; unit B reads
		// column t' - B for t' < t + 1024: every unit's column is still there iff nU - 1 + 2048 <= ED_RING - 1 (larger pairs: the next unit size)
		if (nU + 2048 <= ED_RING) while (true) {
			if (k >= K_MAX && nU > 64) break;   // (with at most one unit per lane there is nothing to keep disjoint: any band width works)
			// ---- one banded pass
			uint64_t VP[G], VN[G], eqA[G], eqC[G], eqG[G], eqT[G];
			uint32_t B = lane;                       // current unit of this lane
			int32_t score = 0;
			bool fresh = true;                       // unit state not initialised yet
			uint32_t pack = 0;                       // published (score << 2) | (hout + 1)
			uint32_t loadedEnd = 0;
			// per-unit step window, refreshed when the lane moves to its next unit
			uint32_t tBegin = 0xffffffffu, tEnd = 0xffffffffu, tHinEnd = 0, finalStep = 0xffffffffu;
			const uint32_t lastUnit = (n - 1) / RB;
			// Ukkonen's corridor: a path of cost <= k from (0,0) to (n-1,m-1) spends at least |row - column| to get where it is and at least
			// |(n - m) - (row - column)| to get home, so it only visits diagonals with |d| + |(n - m) - d| <= k: from min(0, n-m) - a to
			// max(0, n-m) + a with a = (k - |n-m|) / 2 - about k diagonals, half of the symmetric |row - column| <= k
			const uint32_t slack = (k - diff) / 2;
			const uint64_t reachRight = (uint64_t)(n > m ? n - m : 0) + slack;   // row - column at most this: columns from rowBase - reachRight
			const uint64_t reachLeft = (uint64_t)(m > n ? m - n : 0) + slack;    // column - row at most this: columns to row + reachLeft
			auto enterUnit = [&](uint32_t b) {
				B = b;
				fresh = true;
				if (b >= nU) { tBegin = 0xffffffffu; tEnd = 0xffffffffu; return; }
				const uint64_t rowBase = (uint64_t)RB * b;
				const uint64_t c0 = rowBase > reachRight ? rowBase - reachRight : 0;
				const uint64_t c1 = rowBase + RB - 1 + reachLeft;
				const uint64_t lastCol = c1 < m - 1 ? c1 : m - 1;
				tBegin = c0 > lastCol ? 0xffffffffu : (uint32_t)(c0 + b);          // (a unit below the band at every column never runs)
				tEnd = (uint32_t)(lastCol + b);
				tHinEnd = b > 0 ? (uint32_t)(rowBase - 1 + reachLeft + b) : 0;      // last step whose column the upper neighbour also computes
				finalStep = b == lastUnit ? m - 1 + b : 0xffffffffu;
			};
			enterUnit(lane);
			if (lane == 0) resultSlot = -1;
			const uint32_t steps = m + nU - 1;
			for (uint32_t t = 0; t < steps; t++) {
				if ((t & 1023u) == 0) {
					uint32_t end = t + 2048 < m ? t + 2048 : m;
					for (uint32_t c = loadedEnd + lane; c < end; c += 64) {
						const uint8_t ch = (uint8_t)path[c];
						ring[c & (ED_RING - 1)] = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4;
					}
					loadedEnd = end > loadedEnd ? end : loadedEnd;
					__syncthreads();
				}
				const uint32_t nbPack = __shfl(pack, (lane + 63) & 63);
				if (B < nU && t > tEnd) { do enterUnit(B + 64); while (B < nU && t > tEnd); }
				if (t < tBegin) continue;
				const uint32_t j = t - B;
				if (fresh) {
					fresh = false;
					const uint32_t wordBase = (RB / 64) * B;
					for (int q = 0; q < G; q++) {
						VP[q] = ~0ull; VN[q] = 0;
						const uint32_t w = wordBase + q;
						const bool in = w < rd.words;
						eqA[q] = in ? masks[w] : 0ull;
						eqC[q] = in ? masks[rd.words + w] : 0ull;
						eqG[q] = in ? masks[2ull * rd.words + w] : 0ull;
						eqT[q] = in ? masks[3ull * rd.words + w] : 0ull;
					}
					if (j == 0) score = (int32_t)(RB * (B + 1));
					else score = (int32_t)(nbPack >> 2) - ((int32_t)(nbPack & 3u) - 1) + (int32_t)RB;   // below the upper neighbour's previous column, all +1
				}
				const uint32_t code = ring[j & (ED_RING - 1)];
				int hin = 1;                         // top of the band (and row -1 of the matrix): +1 per column
				if (B > 0 && t <= tHinEnd) hin = (int)(nbPack & 3u) - 1;
				uint64_t hinP = hin > 0 ? 1 : 0, hinN = hin < 0 ? 1 : 0;
				for (int q = 0; q < G; q++) {
					uint64_t Eq;
					if (code < 4) {
						const uint64_t lo = (code & 1) ? eqC[q] : eqA[q], hi = (code & 1) ? eqT[q] : eqG[q];
						Eq = (code & 2) ? hi : lo;
					} else {                         // any other letter: compare the 64 read bases of this block directly
						Eq = 0;
						const uint8_t letter = (uint8_t)path[j];
						const uint64_t row0 = (uint64_t)RB * B + 64ull * q;
						for (uint32_t i = 0; i < 64 && row0 + i < n; i++) if ((uint8_t)bases[rd.readOff + row0 + i] == letter) Eq |= 1ull << i;
					}
					const uint64_t vp = VP[q], vn = VN[q];
					const uint64_t Xv = Eq | vn;
					Eq |= hinN;
					const uint64_t Xh = (((Eq & vp) + vp) ^ vp) | Eq;
					uint64_t Ph = vn | ~(Xh | vp);
					uint64_t Mh = vp & Xh;
					const uint64_t outP = Ph >> 63, outN = Mh >> 63;
					Ph = (Ph << 1) | hinP;
					Mh = (Mh << 1) | hinN;
					VP[q] = Mh | ~(Xv | Ph);
					VN[q] = Ph & Xv;
					hinP = outP; hinN = outN;
				}
				const int hout = (int)hinP - (int)hinN;
				score += hout;
				pack = ((uint32_t)score << 2) | (uint32_t)(hout + 1);
				if (t == finalStep) {                // the cell (n-1, m-1): subtract the deltas of the padding rows below it
					int32_t d = score;
					for (int q = 0; q < G; q++) {
						const uint64_t row0 = (uint64_t)RB * B + 64ull * q;
						uint64_t pad = 0;
						if (row0 >= n) pad = ~0ull;
						else if (row0 + 64 > n) pad = ~0ull << (n - row0);
						d -= __popcll(VP[q] & pad);
						d += __popcll(VN[q] & pad);
					}
					resultSlot = d;
				}
			}
			__syncthreads();
			const int32_t d = resultSlot;
			__syncthreads();
			if (d >= 0 && (uint32_t)d <= k) { answer = d; break; }
			if (k >= cap) { answer = d; break; }    // the band already covers the whole matrix
			k = k * 2 < cap ? k * 2 : cap;
		}
		if (lane == 0) outDistance[pi] = answer;
	}
}

// Several pairs per wave (unit of one block): the corridor of a 10 kb pair is 16-23 units wide, so a wave with one pair keeps a third of its
// lanes busy and its time is set by the number of steps (columns + units), not by the band. The wave is cut into teams of TEAM lanes - 32: two
// pairs, bands up to k = 1900; 21: three pairs (one lane idles), bands up to k = 1290, which a 10 kb read's whole-read pair and most chain pairs
// fit (r4) - and every team sweeps its own pair: lane l of a team owns the units l, l + TEAM, ... (a lane's consecutive units are disjoint in
// time while k < 65 TEAM - 63), the neighbour shuffle stays inside the team, each team has its own letter ring and result slot, and the teams
// step together (a team that is done, or has a shorter pair, idles). Same arithmetic as k_edit_distance<1>; a pair whose band would outgrow the
// team answers -2 and is rerun by the host with the wider kernels.
#define ED_HALF_KMAX 1900u
#define ED_THIRD_KMAX 1290u
template <uint32_t TEAM, uint32_t RING>
__global__ void __launch_bounds__(64) k_edit_distance_team(const EdPair* __restrict__ pairs, uint32_t nPairs, const EdRead* __restrict__ reads, const char* __restrict__ bases,
	const uint64_t* __restrict__ eqMasks, const char* __restrict__ letters, const uint32_t* __restrict__ lettersLen, int64_t* __restrict__ outDistance)
{
	GC_RAISE_ED_PRIO();
	constexpr uint32_t PAIRS = 64u / TEAM, KMAX = TEAM >= 32 ? ED_HALF_KMAX : ED_THIRD_KMAX, AHEAD = RING / 2, PERIOD = RING / 4;
	static_assert(KMAX < 65 * TEAM - 63, "a lane's consecutive units must stay disjoint in time");
	__shared__ uint8_t ringAll[(PAIRS + 1) * RING];   // (+ 1: the lanes beyond the last whole team address a ring of their own and never touch it)
	__shared__ int32_t resultSlot[PAIRS + 1];
	const uint32_t lane = threadIdx.x, half = lane / TEAM, l = lane - half * TEAM;
	uint8_t* ring = ringAll + half * RING;
	constexpr uint32_t RB = 64u;
	for (uint32_t base = blockIdx.x * PAIRS; base < nPairs; base += gridDim.x * PAIRS) {
		const uint32_t pi = base + half;
		const bool valid = half < PAIRS && pi < nPairs;
		EdPair pair {};
		EdRead rd {};
		uint32_t n = 0, m = 0;
		if (valid) {
			pair = pairs[pi];
			rd = reads[pair.read];
			n = rd.len;
			m = lettersLen ? lettersLen[pair.lenIndex] : pair.m;
		}
		int64_t answer = -2;
		bool more = valid;
		if (valid && m == 0xffffffffu) { answer = -3; more = false; }            // path letters overflowed their slot
		else if (valid && (n == 0 || m == 0)) { answer = (int64_t)(n + m); more = false; }
		const char* path = letters + pair.lettersOff;
		const uint64_t* masks = eqMasks + rd.eqOff;
		const uint32_t nU = (n + RB - 1) / RB;
		const uint32_t diff = n > m ? n - m : m - n;
		const uint32_t cap = n > m ? n : m;
		uint32_t k = pair.k > diff ? pair.k : diff;
		if (k < 1) k = 1;
		if (k > cap) k = cap;
		if (more && !(nU + AHEAD <= RING)) more = false;                       // the ring must still hold column t - (nU - 1) when it is refilled up to t + AHEAD (-2: the wider kernels take it)
		while (__any(more)) {
			if (more && k >= KMAX && nU > TEAM) more = false;                    // -2
			// ---- one banded pass of every half that still has one to do
			uint64_t VP = ~0ull, VN = 0, eqA = 0, eqC = 0, eqG = 0, eqT = 0;
			uint32_t B = l;
			int32_t score = 0;
			bool fresh = true;
			uint32_t pack = 0, loadedEnd = 0;
			uint32_t tBegin = 0xffffffffu, tEnd = 0xffffffffu, tHinEnd = 0, finalStep = 0xffffffffu;
			const uint32_t lastUnit = n ? (n - 1) / RB : 0;
			const uint32_t slack = more ? (k - diff) / 2 : 0;
			const uint64_t reachRight = (uint64_t)(n > m ? n - m : 0) + slack;
			const uint64_t reachLeft = (uint64_t)(m > n ? m - n : 0) + slack;
			auto enterUnit = [&](uint32_t b) {
				B = b;
				fresh = true;
				if (!more || b >= nU) { tBegin = 0xffffffffu; tEnd = 0xffffffffu; return; }
				const uint64_t rowBase = (uint64_t)RB * b;
				const uint64_t c0 = rowBase > reachRight ? rowBase - reachRight : 0;
				const uint64_t c1 = rowBase + RB - 1 + reachLeft;
				const uint64_t lastCol = c1 < m - 1 ? c1 : m - 1;
				tBegin = c0 > lastCol ? 0xffffffffu : (uint32_t)(c0 + b);
				tEnd = (uint32_t)(lastCol + b);
				tHinEnd = b > 0 ? (uint32_t)(rowBase - 1 + reachLeft + b) : 0;
				finalStep = b == lastUnit ? m - 1 + b : 0xffffffffu;
			};
			enterUnit(l);
			if (l == 0 && half < PAIRS) resultSlot[half] = -1;
			const uint32_t steps = more ? m + nU - 1 : 0;
			uint32_t allSteps = steps;
			for (int d = 32; d >= 1; d >>= 1) { const uint32_t other = (uint32_t)__shfl_xor((int)allSteps, d); allSteps = other > allSteps ? other : allSteps; }
			const int neighbour = (int)(half * TEAM + (l + TEAM - 1) % TEAM);
			for (uint32_t t = 0; t < allSteps; t++) {
				if ((t & (PERIOD - 1)) == 0) {
					if (more) {
						const uint32_t end = t + AHEAD < m ? t + AHEAD : m;
						for (uint32_t c = loadedEnd + l; c < end; c += TEAM) {
							const uint8_t ch = (uint8_t)path[c];
							ring[c & (RING - 1)] = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4;
						}
						loadedEnd = end > loadedEnd ? end : loadedEnd;
					}
					__syncthreads();
				}
				const uint32_t nbPack = (uint32_t)__shfl((int)pack, neighbour);
				if (B < nU && t > tEnd) { do enterUnit(B + TEAM); while (B < nU && t > tEnd); }
				if (t < tBegin || t >= steps) continue;
				const uint32_t j = t - B;
				if (fresh) {
					fresh = false;
					const uint32_t w = B;
					const bool in = w < rd.words;
					VP = ~0ull; VN = 0;
					eqA = in ? masks[w] : 0ull;
					eqC = in ? masks[rd.words + w] : 0ull;
					eqG = in ? masks[2ull * rd.words + w] : 0ull;
					eqT = in ? masks[3ull * rd.words + w] : 0ull;
					if (j == 0) score = (int32_t)(RB * (B + 1));
					else score = (int32_t)(nbPack >> 2) - ((int32_t)(nbPack & 3u) - 1) + (int32_t)RB;
				}
				const uint32_t code = ring[j & (RING - 1)];
				int hin = 1;
				if (B > 0 && t <= tHinEnd) hin = (int)(nbPack & 3u) - 1;
				uint64_t hinP = hin > 0 ? 1 : 0, hinN = hin < 0 ? 1 : 0;
				uint64_t Eq;
				if (code < 4) {
					const uint64_t lo = (code & 1) ? eqC : eqA, hi = (code & 1) ? eqT : eqG;
					Eq = (code & 2) ? hi : lo;
				} else {
					Eq = 0;
					const uint8_t letter = (uint8_t)path[j];
					const uint64_t row0 = (uint64_t)RB * B;
					for (uint32_t i = 0; i < 64 && row0 + i < n; i++) if ((uint8_t)bases[rd.readOff + row0 + i] == letter) Eq |= 1ull << i;
				}
				const uint64_t Xv = Eq | VN;
				Eq |= hinN;
				const uint64_t Xh = (((Eq & VP) + VP) ^ VP) | Eq;
				uint64_t Ph = VN | ~(Xh | VP);
				uint64_t Mh = VP & Xh;
				const uint64_t outP = Ph >> 63, outN = Mh >> 63;
				Ph = (Ph << 1) | hinP;
				Mh = (Mh << 1) | hinN;
				VP = Mh | ~(Xv | Ph);
				VN = Ph & Xv;
				const int hout = (int)outP - (int)outN;
				score += hout;
				pack = ((uint32_t)score << 2) | (uint32_t)(hout + 1);
				if (t == finalStep) {
					int32_t d = score;
					const uint64_t row0 = (uint64_t)RB * B;
					uint64_t pad = 0;
					if (row0 >= n) pad = ~0ull;
					else if (row0 + 64 > n) pad = ~0ull << (n - row0);
					d -= __popcll(VP & pad);
					d += __popcll(VN & pad);
					resultSlot[half] = d;
				}
			}
			__syncthreads();
			const int32_t d = resultSlot[half];
			__syncthreads();
			if (more) {
				if (d >= 0 && (uint32_t)d <= k) { answer = d; more = false; }
				else if (k >= cap) { answer = d; more = false; }
				else k = k * 2 < cap ? k * 2 : cap;
			}
		}
		if (l == 0 && valid) outDistance[pi] = answer;
	}
}

// The whole matrix by one WORKGROUP (r4): thread B owns the 64 rows of block B for every column, the threads sweep the matrix as one skewed wavefront (block B is on column t - B at
// step t) and hand the horizontal delta of their bottom row to the next thread through LDS, one barrier per step. For the pairs whose band would cover most of the matrix anyway - a
// chain whose path spells a fraction of its read (distance >= the length difference), a whole-read alignment of a sliver of its read: the banded kernels give such a pair ONE wave
// with up to sixteen blocks per lane and step (config 5: a 50 000 x 1 000 pair held its stream for 385 ms; cfg2: ~800 pairs per batch in k_edit_distance<4>, 11.8 ms alone). Here a
// pair takes columns + blocks steps of one block each: ~1 ms for the former, and every pair of the latter kind gets its own three waves. No band, no retry: the result is exact.
#define ED_BLOCK_THREADS 1024u
__global__ void __launch_bounds__(ED_BLOCK_THREADS) k_edit_distance_block(const EdPair* __restrict__ pairs, uint32_t nPairs, const EdRead* __restrict__ reads, const char* __restrict__ bases,
	const uint64_t* __restrict__ eqMasks, const char* __restrict__ letters, const uint32_t* __restrict__ lettersLen, int64_t* __restrict__ outDistance)
{
	GC_RAISE_ED_PRIO();
	__shared__ int8_t hand[2][ED_BLOCK_THREADS];   // the horizontal delta leaving block B's bottom row at the column it has just computed, double-buffered by step parity
	__shared__ int32_t resultSlot;
	const uint32_t B = threadIdx.x;
	for (uint32_t pi = blockIdx.x; pi < nPairs; pi += gridDim.x) {
		const EdPair pair = pairs[pi];
		const EdRead rd = reads[pair.read];
		const uint32_t n = rd.len;
		const uint32_t m = lettersLen ? lettersLen[pair.lenIndex] : pair.m;
		if (m == 0xffffffffu) { if (B == 0) outDistance[pi] = -3; continue; }   // path letters overflowed their slot
		if (n == 0 || m == 0) { if (B == 0) outDistance[pi] = (int64_t)(n + m); continue; }
		const uint32_t nU = (n + 63u) / 64u;
		if (nU > blockDim.x) { if (B == 0) outDistance[pi] = -2; continue; }       // (the host sends only pairs that fit: -2 = the banded kernels take it)
		const char* path = letters + pair.lettersOff;
		const uint64_t* masks = eqMasks + rd.eqOff;
		const bool mine = B < nU;
		const bool in = mine && B < rd.words;
		const uint64_t eqA = in ? masks[B] : 0ull, eqC = in ? masks[rd.words + B] : 0ull, eqG = in ? masks[2ull * rd.words + B] : 0ull, eqT = in ? masks[3ull * rd.words + B] : 0ull;
		uint64_t VP = ~0ull, VN = 0;
		int32_t score = (int32_t)(64u * (B + 1));                                  // the bottom row of the block in the column before the first: D[i][-1] = i + 1
		const uint32_t lastUnit = (n - 1) / 64u;
		const uint32_t steps = m + nU - 1;
		if (B == 0) resultSlot = -1;
		__syncthreads();
		uint8_t chNext = mine && B == 0 ? (uint8_t)path[0] : 0;                      // the letter of the NEXT step is fetched a step ahead: its latency hides behind this step's work and barrier
		for (uint32_t t = 0; t < steps; t++) {
			const uint32_t j = t - B;
			const uint8_t ch = chNext;
			if (mine && t + 1 >= B && j + 1 < m) chNext = (uint8_t)path[j + 1];
			if (mine && t >= B && j < m) {
				const int hin = B > 0 ? (int)hand[(t + 1) & 1][B - 1] : 1;              // (the row above the matrix climbs by one per column)
				uint64_t Eq;
				if (ch == 'A') Eq = eqA; else if (ch == 'C') Eq = eqC; else if (ch == 'G') Eq = eqG; else if (ch == 'T') Eq = eqT;
				else {
					Eq = 0;
					const uint64_t row0 = 64ull * B;
					for (uint32_t i = 0; i < 64 && row0 + i < n; i++) if ((uint8_t)bases[rd.readOff + row0 + i] == ch) Eq |= 1ull << i;
				}
				const uint64_t hinP = hin > 0 ? 1 : 0, hinN = hin < 0 ? 1 : 0;
				const uint64_t Xv = Eq | VN;
				Eq |= hinN;
				const uint64_t Xh = (((Eq & VP) + VP) ^ VP) | Eq;
				uint64_t Ph = VN | ~(Xh | VP);
				uint64_t Mh = VP & Xh;
				const int hout = (int)(Ph >> 63) - (int)(Mh >> 63);
				Ph = (Ph << 1) | hinP;
				Mh = (Mh << 1) | hinN;
				VP = Mh | ~(Xv | Ph);
				VN = Ph & Xv;
				score += hout;
				hand[t & 1][B] = (int8_t)hout;
				if (B == lastUnit && j == m - 1) {
					int32_t d = score;
					const uint64_t row0 = 64ull * B;
					uint64_t pad = 0;
					if (row0 + 64 > n) pad = ~0ull << (n - row0);                        // rows of the block beyond the read: step back up to row n - 1
					d -= __popcll(VP & pad);
					d += __popcll(VN & pad);
					resultSlot = d;
				}
			}
			__syncthreads();
		}
		if (B == 0) outDistance[pi] = (int64_t)resultSlot;
		__syncthreads();
	}
}

void launchEditDistanceBlock(hipStream_t stream, uint32_t threads, const EdPair* pairs, uint32_t nPairs, const EdRead* reads, const char* bases, const uint64_t* eqMasks,
	const char* letters, const uint32_t* lettersLen, int64_t* outDistance)
{
	if (!nPairs) return;
	threads = ((threads < 64u ? 64u : threads) + 63u) & ~63u;
	if (threads > ED_BLOCK_THREADS) threads = ED_BLOCK_THREADS;
	hipLaunchKernelGGL(k_edit_distance_block, dim3(nPairs < 65536u ? nPairs : 65536u), dim3(threads), 0, stream, pairs, nPairs, reads, bases, eqMasks, letters, lettersLen, outDistance);
}
uint32_t editDistanceBlockMaxRows() { return 64u * ED_BLOCK_THREADS; }

void launchLongPathSeq(hipStream_t stream, const DGraph& g, const PathSeqJob* jobs, uint32_t nJobs, const LongCell* cellPool, char* letters, uint32_t* outLen)
{
	if (nJobs) hipLaunchKernelGGL(k_long_pathseq, dim3(nJobs), dim3(64), 0, stream, g, jobs, nJobs, cellPool, letters, outLen);
}
void launchChainPathSeq(hipStream_t stream, const DGraph& g, const PathSeqJob* jobs, uint32_t nJobs, const uint32_t* pathNodes, const uint32_t* altNodes, char* letters, uint32_t* outLen)
{
	if (nJobs) hipLaunchKernelGGL(k_chain_pathseq, dim3(nJobs), dim3(64), 0, stream, g, jobs, nJobs, pathNodes, altNodes, letters, outLen);
}
uint32_t editDistanceMaxK(uint32_t unitBlocks) { return unitBlocks == 0 ? ED_HALF_KMAX : 4000u * unitBlocks; }   // unit 0: the two-pairs-per-wave kernel
uint32_t editDistanceTeamMaxK(uint32_t pairsPerWave) { return pairsPerWave == 3 ? ED_THIRD_KMAX : ED_HALF_KMAX; }
void launchEditDistanceTeam(hipStream_t stream, uint32_t pairsPerWave, const EdPair* pairs, uint32_t nPairs, const EdRead* reads, const char* bases, const uint64_t* eqMasks,
	const char* letters, const uint32_t* lettersLen, int64_t* outDistance)
{
	if (!nPairs) return;
	const uint32_t waves = (nPairs + pairsPerWave - 1) / pairsPerWave, blocks = waves < 65536 ? waves : 65536;
	if (pairsPerWave == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_edit_distance_team<21, 2048>), dim3(blocks), dim3(64), 0, stream, pairs, nPairs, reads, bases, eqMasks, letters, lettersLen, outDistance);
	else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_edit_distance_team<32, 4096>), dim3(blocks), dim3(64), 0, stream, pairs, nPairs, reads, bases, eqMasks, letters, lettersLen, outDistance);
}
void launchEditDistance(hipStream_t stream, uint32_t unitBlocks, const EdPair* pairs, uint32_t nPairs, const EdRead* reads, const char* bases, const uint64_t* eqMasks,
	const char* letters, const uint32_t* lettersLen, int64_t* outDistance)
{
	if (!nPairs) return;
	uint32_t blocks = nPairs < 65536 ? nPairs : 65536;
	if (unitBlocks == 0) { launchEditDistanceTeam(stream, 2, pairs, nPairs, reads, bases, eqMasks, letters, lettersLen, outDistance); return; }   // two pairs per wave (launchEditDistances sends the pairs with a small first band here)
	switch (unitBlocks) {
		case 1: hipLaunchKernelGGL(k_edit_distance<1>, dim3(blocks), dim3(64), 0, stream, pairs, nPairs, reads, bases, eqMasks, letters, lettersLen, outDistance); break;
		case 2: hipLaunchKernelGGL(k_edit_distance<2>, dim3(blocks), dim3(64), 0, stream, pairs, nPairs, reads, bases, eqMasks, letters, lettersLen, outDistance); break;
		case 4: hipLaunchKernelGGL(k_edit_distance<4>, dim3(blocks), dim3(64), 0, stream, pairs, nPairs, reads, bases, eqMasks, letters, lettersLen, outDistance); break;
		case 8: hipLaunchKernelGGL(k_edit_distance<8>, dim3(blocks), dim3(64), 0, stream, pairs, nPairs, reads, bases, eqMasks, letters, lettersLen, outDistance); break;
		default: hipLaunchKernelGGL(k_edit_distance<16>, dim3(blocks), dim3(64), 0, stream, pairs, nPairs, reads, bases, eqMasks, letters, lettersLen, outDistance); break;
	}
}

} // namespace gcdev
