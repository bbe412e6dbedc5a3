// Chain stitching on the GPU - SURVEY.md §8 row f3.
//
// Reference: the stitching loop of runComponentMappings (src/Aligner.cpp:754-822), AlignmentGraph::getChainPath
// (src/AlignmentGraph.cpp:1866-1916) and the size of pathToTrace's result (src/Aligner.cpp:409-424).
//   - the node paths of the chain's anchors are appended in chain order, every node at most once per piece;
//   - when the next anchor starts in a node that is neither on the piece nor its last node, the two are bridged by the
//     fewest-hops path: a BFS over outNeighbors in adjacency order from the piece's last node, which stops expanding
//     nodes further than the remaining --colinear-gap budget in bp and stops when the target has been reached. The
//     predecessor of a node is the first node that reached it, so the path depends on the visiting order, which is
//     kept: queue order = insertion order, neighbours in CSR order;
//   - no bridge (or a jump of more than the gap inside the same node) ends the piece; the piece with the most bases
//     (pathToTrace length, strictly more to replace an earlier one) is the result.
// The reference's budget test compares an unsigned distance with (size_t)sepLimit, so a negative remaining budget
// (other than the -1 "no limit") also means "no limit"; kept as is.
//
// k_stitch: one wave per read. The lanes together find the cut-off after a failed fragment, number the read's valid
// anchors (the chain holds indices into that numbering, as in k_chain) and clear the LDS tables; lane 0 then runs the
// inherently sequential loop. Its inputs are staged 64 chain anchors at a time: every lane loads one anchor record, the
// first nodes of its path with their lengths, and tests whether the anchor starts in an out-neighbour of the previous
// anchor's last node (then the BFS is known to return that one hop), so lane 0 works from LDS. A longer bridge search is
// run by the whole wave (adjacency loads in parallel, insertions replayed in the sequential order). The piece's node set and the BFS's visited map / queue live in LDS; pieces are
// written one after the other into the read's scratch region, and the best one is copied by all lanes into a dense
// output array (position from an atomic cursor; StitchInfo.start says where).
// Anything that does not fit - more nodes on a piece than half the set's slots, a BFS that visits more than STITCH_BFS_CAP
// nodes, a full region or output array - sets status 1 and the host stitches that read with the same algorithm
// (gc_runtime.hpp: stitchChain).
#include "gc_kernels.hpp"
#include <hip/hip_runtime.h>

namespace gcdev {

// A second, smaller size class of the search tables was measured in r4 and is NOT the default (GC_STITCH_SMALL=1 selects it for reads up to 16 kb): with 512 instead of 1 024
// visited nodes per bridge search (20 KB of LDS per wave instead of 29.5, seven waves per CU instead of five) 39 % of cfg2's reads overflowed the search and went to the
// host's stitching (3 915 of 10 000: a failed search for an upstream anchor walks several hundred nodes before the rank pruning ends it), and with a 1 024-slot node set as well
// a piece's ~450 split nodes no longer fitted: host CPU per batch 0.36 -> 0.8-0.9 s for 2-3 % of batch time (`gpurun_out/r4_rounds`, `r4_rounds2`, `r4_e2etimes`).
#define STITCH_SET_SIZE_LARGE 2048u   // open-addressing slots for the nodes of the current piece; at most half of them are used (STITCH_SET_MAX)
#define STITCH_BFS_CAP_LARGE 1024u    // visited nodes per bridge search; its hash table has twice the slots
#define STITCH_EMPTY 0xffffffffu
#define STITCH_PF_NODES 4u        // path nodes of an anchor staged in LDS (35 bp paths have 1-3)

__device__ __forceinline__ uint32_t stitchHash(uint32_t node, uint32_t mask) { return (node * 2654435761u >> 12) & mask; }

// One chain anchor as lane 0 needs it, staged in LDS by the lane that loaded it.
struct StitchAnchor {
	uint32_t firstNode, firstOffset, lastOffset, pathLen;
	uint32_t pathOffLo, pathOffHi;
	uint32_t node[STITCH_PF_NODES];
	uint32_t nodeLen;             // lengths of node[0..3], one byte each
	uint32_t prevLast;            // last path node of the previous chain anchor
	uint32_t flags;               // 1: firstNode is an out-neighbour of prevLast, 2: unusable record (the host stitches the read)
};

// SPILL (r5, the class of long reads): a 50 kb read's piece holds ~2 500 split nodes and a failed bridge search inside a 50 kb gap budget visits a few thousand - neither fits
// the LDS tables, and tables four times the size (r4's class 2: 107 KB, one wave per CU) lost to sixteen host threads. Here the LDS stays what the default class uses (five waves
// per CU): the piece's node set moves to a generation-stamped hash table in the block's HBM scratch behind a 32 K-bit membership filter in LDS - most look-ups ask for a node that
// is NOT on the piece and end at the filter without touching memory; a new piece bumps the generation and clears the filter, nothing in HBM - and a bridge search that outgrows the
// LDS queue is run again with queue, distances, predecessors and visited table in the scratch (same code, same visiting order; rare: one search in a few hundred).
__host__ __device__ constexpr uint64_t stitchSpillWordsPerBlockC() { return 16384ull /* node set, 64-bit slots */ + (2ull * 16384ull /* visited table */ + 3ull * 16384ull /* queue, distances, predecessors */) / 2; }
#define STITCH_SPILL_SET_SLOTS 16384u     // node set in HBM: 64-bit slots (generation << 32 | node); at most half are used
#define STITCH_SPILL_BFS_CAP 16384u       // visited nodes of a bridge search that runs in HBM
#define STITCH_FILTER_WORDS 1024u
template <uint32_t STITCH_SET_SIZE, uint32_t STITCH_BFS_CAP, bool SPILL>
__global__ void __launch_bounds__(64) k_stitch(DGraph g, const ReadChainJob* __restrict__ jobs, uint32_t nReads, const AnchorRec* __restrict__ anchors,
	const Fragment* __restrict__ frags, const uint32_t* __restrict__ fragStatus, const uint32_t* __restrict__ chainOut, const uint32_t* __restrict__ chainLen,
	const uint32_t* __restrict__ chainStatus, const uint32_t* __restrict__ pathPool, uint64_t pathCapacity, long long colinearGap, uint32_t setMax, uint32_t bfsCap,
	uint32_t* __restrict__ slotOf, uint32_t* __restrict__ regions, uint32_t* __restrict__ dense, uint64_t denseCap, unsigned long long* __restrict__ denseCursor,
	StitchInfo* __restrict__ info, unsigned long long* __restrict__ spill)
{
	GC_RAISE_PRIO();
	constexpr uint32_t STITCH_BFS_TABLE = 2 * STITCH_BFS_CAP;
	constexpr uint32_t INDEX_BITS = STITCH_BFS_CAP <= 1024 ? 11u : 13u, INDEX_MASK = (1u << INDEX_BITS) - 1u;   // a table entry: (generation << INDEX_BITS) | (queue index + 1)
	static_assert(STITCH_BFS_CAP < (1u << INDEX_BITS), "queue index + 1 must fit its bits");
	__shared__ uint32_t setKey[SPILL ? STITCH_FILTER_WORDS : STITCH_SET_SIZE];   // SPILL: the membership filter (one bit per hashed node) in front of the HBM table
	__shared__ uint32_t bfsTable[STITCH_BFS_TABLE];   // (generation << INDEX_BITS) | (queue index + 1)
	__shared__ uint32_t qNode[STITCH_BFS_CAP], qDis[STITCH_BFS_CAP];
	__shared__ uint16_t qPre[STITCH_BFS_CAP];
	uint32_t* const bridge = qDis;   // the path is written out when the search is over and the distances are no longer needed
	__shared__ StitchAnchor staged[64];
	__shared__ uint32_t sOverflow, sBestStart, sBestLen, sSearchFrom;
	__shared__ long long sSearchLimit;
	__shared__ unsigned long long sDenseAt;
	const uint32_t lane = threadIdx.x;
	// SPILL: this block's HBM scratch (zeroed by the launcher): the node set, then the wide search's visited table, queue, distances, predecessors
	unsigned long long* const spillSet = SPILL ? spill + (uint64_t)blockIdx.x * stitchSpillWordsPerBlockC() : nullptr;
	uint32_t* const spillTable = (uint32_t*)(spillSet + STITCH_SPILL_SET_SLOTS);
	uint32_t* const spillNode = spillTable + 2 * STITCH_SPILL_BFS_CAP;
	uint32_t* const spillDis = spillNode + STITCH_SPILL_BFS_CAP;
	uint32_t* const spillPre = spillDis + STITCH_SPILL_BFS_CAP;
	uint32_t pieceGen = 0, spillSearches = 0;   // generations of the node set (one per piece) and of the wide search's table (one per wide search), counted across the block's reads
	for (uint32_t r = blockIdx.x; r < nReads; r += gridDim.x) {
		const ReadChainJob job = jobs[r];
		const uint32_t len = chainLen[r];
		StitchInfo result = { 0, 0, 0, 0, 0, 0, 0, 0 };
		if (chainStatus[r] != 0 || len == 0) { if (lane == 0) info[r] = result; continue; }
		// ---- all lanes: cut-off after the first failed fragment, numbering of the valid anchors (same as k_chain), clear tables
		uint32_t cut = job.nSlots;
		for (uint32_t f = lane; f < job.nFrags; f += 64)
			if (fragStatus[job.fragBegin + f] == 1) { uint32_t c = frags[job.fragBegin + f].seedBegin - job.slotBegin; cut = c < cut ? c : cut; }
		for (int d = 32; d > 0; d >>= 1) { uint32_t o = __shfl_xor(cut, d); cut = o < cut ? o : cut; }
		uint32_t nA = 0;
		for (uint32_t s0 = 0; s0 < cut; s0 += 64) {
			uint32_t s = s0 + lane;
			bool valid = s < cut && anchors[job.slotBegin + s].valid != 0;
			unsigned long long ballot = __ballot(valid);
			if (valid) slotOf[job.slotBegin + nA + (uint32_t)__popcll(ballot & ((1ull << lane) - 1))] = s;
			nA += (uint32_t)__popcll(ballot);
		}
		for (uint32_t i = lane; i < (SPILL ? STITCH_FILTER_WORDS : STITCH_SET_SIZE); i += 64) setKey[i] = SPILL ? 0u : STITCH_EMPTY;
		for (uint32_t i = lane; i < STITCH_BFS_TABLE; i += 64) bfsTable[i] = 0;
		if (lane == 0) sOverflow = 0;
		pieceGen++;
		__syncthreads();

		// ---- lane 0's state: the current piece and the best one so far
		const uint64_t regionBase = 2ull * job.slotBegin + 64ull * r;
		const uint32_t regionCap = 2u * job.nSlots + 64u;
		uint32_t* region = regions + regionBase;
		uint32_t pieceStart = 0, posLen = 0, setCount = 0, generation = 0;
		uint32_t firstOffset = 0, lastOffset = 0, backNode = 0, backLen = 0, firstLen = 0, bestStart = 0;
		uint64_t sumLen = 0;
		bool overflow = false;
		uint32_t why = 0;   // 1 piece/region full, 2 bridge search too wide, 3 unusable anchor record, 4 output array full
		auto contains = [&](uint32_t node) {
			if (SPILL) {
				const uint32_t bit = (node * 0x9E3779B1u) >> 17;   // 15 bits
				if (!((setKey[bit >> 5] >> (bit & 31u)) & 1u)) return false;
				const unsigned long long key = ((unsigned long long)pieceGen << 32) | node;
				for (uint32_t h = stitchHash(node, STITCH_SPILL_SET_SLOTS - 1);; h = (h + 1) & (STITCH_SPILL_SET_SLOTS - 1)) {
					const unsigned long long k = spillSet[h];
					if (k == key) return true;
					if ((uint32_t)(k >> 32) != pieceGen) return false;
				}
			}
			for (uint32_t h = stitchHash(node, STITCH_SET_SIZE - 1);; h = (h + 1) & (STITCH_SET_SIZE - 1)) {
				uint32_t k = setKey[h];
				if (k == node) return true;
				if (k == STITCH_EMPTY) return false;
			}
		};
		// appends a node that is known not to be on the piece; nodeLen 0 = not known
		auto push = [&](uint32_t node, uint32_t nodeLen) {
			if (setCount >= setMax || pieceStart + posLen >= regionCap) { overflow = true; why = 1; return; }
			if (SPILL) {
				const uint32_t bit = (node * 0x9E3779B1u) >> 17;
				setKey[bit >> 5] |= 1u << (bit & 31u);
				uint32_t h = stitchHash(node, STITCH_SPILL_SET_SLOTS - 1);
				while ((uint32_t)(spillSet[h] >> 32) == pieceGen) h = (h + 1) & (STITCH_SPILL_SET_SLOTS - 1);
				spillSet[h] = ((unsigned long long)pieceGen << 32) | node;
			} else {
				uint32_t h = stitchHash(node, STITCH_SET_SIZE - 1);
				while (setKey[h] != STITCH_EMPTY) h = (h + 1) & (STITCH_SET_SIZE - 1);
				setKey[h] = node;
			}
			setCount++;
			region[pieceStart + posLen] = node;
			if (nodeLen == 0) nodeLen = g.nodeLength[node];
			if (posLen == 0) firstLen = nodeLen;
			posLen++;
			sumLen += nodeLen;
			backNode = node;
			backLen = nodeLen;
		};
		// length of pathToTrace(posPath, firstOffset, lastOffset): first node from firstOffset, last node (when it is not
		// the first) up to lastOffset, whole nodes between them
		auto keepIfLonger = [&]() {
			uint64_t cells = firstLen > firstOffset ? firstLen - firstOffset : 0;
			if (posLen > 1) cells += (sumLen - firstLen - backLen) + lastOffset + 1;
			if (result.cells < cells) {
				bestStart = pieceStart;
				result.len = posLen;
				result.firstOffset = firstOffset;
				result.lastOffset = lastOffset;
				result.cells = cells;
			}
		};
		// getChainPath(S, T, sepLimit) by the whole wave: fills bridge[0..n) with the path S..T, returns n (0: not reached).
		// Queue entries are expanded 64 at a time: every lane loads the adjacency of one entry (the slow part: dependent global
		// loads), then all lanes replay the insertions in the reference's order - entry by entry, neighbours in CSR order -
		// executing the same LDS operations on the same values, so the visiting order, and with it every predecessor, is the
		// sequential one. Entries appended during a batch are expanded in a later batch, which is still queue order.
		// (the tables are parameters: the LDS ones, or - SPILL, for a search that outgrew them - the block's HBM scratch)
		auto searchIn = [&](auto* qPre, uint32_t* qNode, uint32_t* qDis, uint32_t* bfsTable, const uint32_t tableSlots, const uint32_t indexBits, uint32_t& generation, const uint32_t bfsCap,
			uint32_t S, uint32_t T, long long sepLimit, bool& tooWide) -> uint32_t {
			uint32_t* const bridge = qDis;
			generation++;
			const uint32_t tag = generation << indexBits, indexMask = (1u << indexBits) - 1u;
			auto visit = [&](uint32_t node, uint32_t index) -> bool {   // true: seen before; otherwise recorded as queue entry `index`
				for (uint32_t h = stitchHash(node, tableSlots - 1);; h = (h + 1) & (tableSlots - 1)) {
					uint32_t e = bfsTable[h];
					if ((e >> indexBits) != generation) { bfsTable[h] = tag | (index + 1); return false; }
					if (qNode[(e & indexMask) - 1] == node) return true;
				}
			};
			// Pruning that cannot change the result: componentNumber never decreases along an edge (it is the topological rank of
			// the node's strongly connected component, src/AlignmentGraph.cpp:1008), so a node with a larger number than T's has
			// no path to T, and neither has anything reached through it; and no node that can reach T is ever first discovered
			// from such a node. Leaving them out of the queue keeps the order, distances and predecessors of all the others, and
			// ends the common failing search - the next anchor lies upstream of the piece's end - after one expansion.
			const uint32_t rankT = g.componentNumber[T];
			if (g.componentNumber[S] > rankT) return 0;
			uint32_t qLen = 1, found = 0;
			qNode[0] = S; qDis[0] = 0; qPre[0] = 0;
			visit(S, 0);
			for (uint32_t i0 = 0, batch = 0; !found && i0 < qLen; i0 += batch) {
				batch = qLen - i0 < 64u ? qLen - i0 : 64u;   // the entries that exist now; the ones they append come in a later batch
				uint32_t deg = 0, e0 = 0, dis = 0, t0 = 0, t1 = 0, t2 = 0, t3 = 0, lens = 0, keep = 0;
				if (lane < batch) {
					uint32_t s = qNode[i0 + lane];
					dis = qDis[i0 + lane];
					if (!((unsigned long long)dis > (unsigned long long)sepLimit)) {
						e0 = g.outOff[s];
						deg = g.outOff[s + 1] - e0;
						if (deg > 0) { t0 = g.outAdj[e0]; lens |= (uint32_t)g.nodeLength[t0]; keep |= g.componentNumber[t0] <= rankT ? 1u : 0u; }
						if (deg > 1) { t1 = g.outAdj[e0 + 1]; lens |= (uint32_t)g.nodeLength[t1] << 8; keep |= g.componentNumber[t1] <= rankT ? 2u : 0u; }
						if (deg > 2) { t2 = g.outAdj[e0 + 2]; lens |= (uint32_t)g.nodeLength[t2] << 16; keep |= g.componentNumber[t2] <= rankT ? 4u : 0u; }
						if (deg > 3) { t3 = g.outAdj[e0 + 3]; lens |= (uint32_t)g.nodeLength[t3] << 24; keep |= g.componentNumber[t3] <= rankT ? 8u : 0u; }
					}
				}
				for (uint32_t l = 0; l < batch && !found; l++) {
					const uint32_t d = __shfl(deg, l);
					if (d == 0) continue;
					const uint32_t dl = __shfl(dis, l), el = __shfl(e0, l), ll = __shfl(lens, l), kp = __shfl(keep, l);
					const uint32_t n0 = __shfl(t0, l), n1 = __shfl(t1, l), n2 = __shfl(t2, l), n3 = __shfl(t3, l);
					for (uint32_t k = 0; k < d; k++) {
						uint32_t t, tLen;
						if (k < 4) { if (!((kp >> k) & 1u)) continue; t = k == 0 ? n0 : k == 1 ? n1 : k == 2 ? n2 : n3; tLen = (ll >> (8 * k)) & 255u; }
						else { t = g.outAdj[el + k]; if (g.componentNumber[t] > rankT) continue; tLen = g.nodeLength[t]; }
						if (qLen >= bfsCap) { tooWide = true; return 0; }
						if (visit(t, qLen)) continue;
						qNode[qLen] = t; qDis[qLen] = dl + tLen; qPre[qLen] = (std::remove_reference_t<decltype(qPre[0])>)(i0 + l);
						qLen++;
						// the reference finishes s's neighbours before it notices that T was reached; the ones after T cannot
						// change pre[T] or anything before it on the path, so the search can stop here
						if (t == T) { found = qLen; break; }
					}
				}
			}
			if (!found) return 0;
			uint32_t hops = 0;
			for (uint32_t i = found - 1; i != 0; i = qPre[i]) hops++;
			uint32_t at = hops;
			for (uint32_t i = found - 1; i != 0; i = qPre[i]) bridge[at--] = qNode[i];
			bridge[0] = S;
			return hops + 1;
		};
		auto findBridge = [&](uint32_t S, uint32_t T, long long sepLimit, bool& tooWide) -> uint32_t {
			return searchIn(qPre, qNode, qDis, bfsTable, STITCH_BFS_TABLE, INDEX_BITS, generation, bfsCap, S, T, sepLimit, tooWide);
		};

		const uint32_t* chain = chainOut + job.chainBegin;
		uint32_t carryLast = 0;   // last path node of the anchor before this batch
		uint32_t scoreSum = 0;    // (all lanes) the chain's anchors' scores
		for (uint32_t c0 = 0; c0 < len; c0 += 64) {
			// ---- all lanes: stage the next 64 chain anchors (record, first path nodes and their lengths, and whether the
			// anchor starts in an out-neighbour of the previous anchor's last node - the common bridge, one hop)
			const uint32_t c = c0 + lane;
			uint32_t myLast = 0, myScore = 0;
			StitchAnchor sa = {};
			if (c < len) {
				uint32_t index = chain[c];
				if (index >= nA) sa.flags = 2;   // cannot happen; leaves the read to the host rather than reading outside
				else {
					const AnchorRec a = anchors[job.slotBegin + slotOf[job.slotBegin + index]];
					myScore = a.score > 0 ? (uint32_t)a.score : 0u;
					sa.firstNode = a.firstNode; sa.firstOffset = a.firstOffset; sa.lastOffset = a.lastOffset; sa.pathLen = a.pathLen;
					sa.pathOffLo = (uint32_t)a.pathOff; sa.pathOffHi = (uint32_t)(a.pathOff >> 32);
					if (a.pathLen == 0 || a.pathOff + a.pathLen > pathCapacity) sa.flags = 2;   // anchor path pool overflow: the host reports it
					else {
#pragma unroll
						for (uint32_t k = 0; k < STITCH_PF_NODES; k++)
							if (k < a.pathLen) {
								uint32_t node = pathPool[a.pathOff + k];
								sa.node[k] = node;
								sa.nodeLen |= (uint32_t)g.nodeLength[node] << (8 * k);
							}
						myLast = pathPool[a.pathOff + a.pathLen - 1];
					}
				}
			}
			for (int d = 32; d > 0; d >>= 1) myScore += __shfl_xor(myScore, d);
			scoreSum += myScore;
			uint32_t prevLast = __shfl_up(myLast, 1);
			if (lane == 0) prevLast = carryLast;
			carryLast = __shfl(myLast, 63);
			if (c < len && c > 0 && sa.flags == 0) {
				sa.prevLast = prevLast;
				if (prevLast != sa.firstNode)
					for (uint32_t e = g.outOff[prevLast]; e < g.outOff[prevLast + 1]; e++) if (g.outAdj[e] == sa.firstNode) { sa.flags = 1; break; }
			}
			staged[lane] = sa;
			__syncthreads();
			const uint32_t batch = len - c0 < 64u ? len - c0 : 64u;
			for (uint32_t i = 0; i < batch; i++) {
				// lane 0, first half: everything up to the point where a bridge search is needed
				const StitchAnchor& a = staged[i];
				const uint32_t* apath = pathPool + (((uint64_t)a.pathOffHi << 32) | a.pathOffLo);
				auto pathNode = [&](uint32_t k) { return k < STITCH_PF_NODES ? a.node[k] : apath[k]; };
				auto pathNodeLen = [&](uint32_t k) { return k < STITCH_PF_NODES ? (a.nodeLen >> (8 * k)) & 255u : 0u; };
				const uint32_t head = a.node[0];
				bool gap = false, search = false, rest = false;
				uint32_t nBridge = 0;
				if (lane == 0 && !overflow) {
					if (a.flags & 2) { overflow = true; why = 3; }
					else if (posLen == 0) {
						for (uint32_t k = 0; k < a.pathLen && !overflow; k++) push(pathNode(k), pathNodeLen(k));   // anchor paths are simple: assign == push each
						firstOffset = a.firstOffset;
						lastOffset = a.lastOffset;
					} else {
						rest = true;
						gap = head == backNode && colinearGap != -1 && (long long)a.firstOffset - (long long)lastOffset > colinearGap + 1;
						if (!contains(head) && backNode != a.firstNode) {
							if ((a.flags & 1) && a.prevLast == backNode) {
								// the target is an out-neighbour of the start: the search expands the start whatever the budget (its
								// distance is 0) and reaches the target at once
								bridge[0] = backNode; bridge[1] = a.firstNode;
								nBridge = 2;
							} else {
								long long gapLimit = colinearGap;
								if (gapLimit != -1) gapLimit -= (long long)a.firstOffset + ((long long)backLen - (long long)lastOffset - 1);
								sSearchFrom = backNode; sSearchLimit = gapLimit;
								search = true;
							}
						}
					}
				}
				// all lanes: the bridge search, when lane 0 asked for one
				const uint32_t* bridgeNow = bridge;   // where this anchor's bridge is: the LDS search's array, or the scratch's after a wide search
				if (__shfl((uint32_t)search, 0)) {
					__syncthreads();
					bool tooWide = false;
					uint32_t n = findBridge(sSearchFrom, a.firstNode, sSearchLimit, tooWide);
					if (SPILL && tooWide) {
						// the search outgrew the LDS queue: once more in the scratch (a search whose target is unreachable walks everything downstream within the gap budget)
						if (spillSearches >= (1u << 17) - 2u) { for (uint32_t k = lane; k < 2 * STITCH_SPILL_BFS_CAP; k += 64) spillTable[k] = 0; spillSearches = 0; __syncthreads(); }
						tooWide = false;
						n = searchIn(spillPre, spillNode, spillDis, spillTable, 2 * STITCH_SPILL_BFS_CAP, 15u, spillSearches, STITCH_SPILL_BFS_CAP, sSearchFrom, a.firstNode, sSearchLimit, tooWide);
						bridgeNow = spillDis;
						__syncthreads();
					}
					if (lane == 0) {
						nBridge = n;
						if (tooWide) { overflow = true; why = 2; }
						else if (n == 0) gap = true;
					}
				}
				// a gap ends the piece: lane 0 scores it, all lanes empty the node set (pieces end often - every failed bridge -
				// and one lane clearing 2048 slots each time was most of the kernel's instructions)
				const bool newPiece = lane == 0 && rest && !overflow && gap;
				if (newPiece) {
					keepIfLonger();
					setCount = 0;
					pieceStart += posLen;
					posLen = 0;
					sumLen = 0;
					firstOffset = a.firstOffset;
				}
				if (__shfl((uint32_t)newPiece, 0)) {
					if (SPILL) { pieceGen++; for (uint32_t k = lane; k < STITCH_FILTER_WORDS; k += 64) setKey[k] = 0; }   // (the HBM table is stamped: the new generation finds it empty)
					else for (uint32_t k = lane; k < STITCH_SET_SIZE; k += 64) setKey[k] = STITCH_EMPTY;
					__syncthreads();
				}
				// lane 0, second half: the bridge (when the piece goes on) and the anchor's own path
				if (lane == 0 && rest && !overflow) {
					if (!gap) for (uint32_t k = 0; k < nBridge && !overflow; k++) if (!contains(bridgeNow[k])) push(bridgeNow[k], bridgeNow[k] == head ? pathNodeLen(0) : 0u);
					for (uint32_t k = 0; k < a.pathLen && !overflow; k++) { uint32_t node = pathNode(k); if (!contains(node)) push(node, pathNodeLen(k)); }
					lastOffset = a.lastOffset;
				}
				if (__shfl((uint32_t)overflow, 0)) break;
			}
			if (lane == 0 && overflow) sOverflow = 1;
			__syncthreads();
			if (sOverflow) break;
		}
		// ---- lane 0 closes the last piece and reserves room in the dense output; all lanes copy the best piece there
		if (lane == 0) {
			if (!overflow && posLen > 0) keepIfLonger();
			unsigned long long at = 0;
			if (!overflow) {
				at = atomicAdd(denseCursor, (unsigned long long)result.len);
				if (at + result.len > denseCap) { overflow = true; why = 4; }
			}
			if (overflow) { result = StitchInfo { 0, 0, 0, 0, 0, why, scoreSum, 0 }; sBestLen = 0; }
			else { result.start = at; sBestLen = result.len; result.scoreSum = scoreSum; }
			sBestStart = bestStart;
			sDenseAt = at;
			info[r] = result;
		}
		__syncthreads();   // also orders lane 0's region writes before the copy, and the next read's table clears after lane 0
		for (uint32_t i = lane; i < sBestLen; i += 64) dense[sDenseAt + i] = region[sBestStart + i];
		__syncthreads();
	}
}

// sizeClass: 0 the default tables (2 048-slot node set, 1 024 visited nodes per bridge search: reads up to ~16 kb), 1 the half-size search (measured in r4, not kept: experiments build),
// 3 (r5) the class of long reads: the default class's LDS plus a node set and a wide-search area in HBM scratch (`spill`: stitchSpillWordsPerBlock() words per block of the launch,
// stitchSpillBlocks(nReads) blocks; zeroed here before every launch - the tables' generation stamps start again with each launch)
void launchStitch(hipStream_t stream, const DGraph& g, const ReadChainJob* jobs, uint32_t nReads, const AnchorRec* anchors, const Fragment* frags, const uint32_t* fragStatus,
	const uint32_t* chainOut, const uint32_t* chainLen, const uint32_t* chainStatus, const uint32_t* pathPool, uint64_t pathCapacity, long long colinearGap, uint32_t* slotOf,
	uint32_t* regions, uint32_t* dense, uint64_t denseCap, unsigned long long* denseCursor, StitchInfo* info, uint32_t setMax, uint32_t bfsCap, int sizeClass, unsigned long long* spill)
{
	if (!nReads) return;
	if (sizeClass == 3 && !spill) sizeClass = 0;
	const uint32_t setSize = sizeClass == 3 ? STITCH_SPILL_SET_SLOTS : STITCH_SET_SIZE_LARGE, capBfs = sizeClass == 1 ? STITCH_BFS_CAP_LARGE / 2 : STITCH_BFS_CAP_LARGE;
	setMax = setMax && setMax < setSize / 2 ? setMax : setSize / 2;
	bfsCap = bfsCap && bfsCap < capBfs ? bfsCap : capBfs;   // (of the LDS search; the wide search of class 3 holds STITCH_SPILL_BFS_CAP)
#ifndef GC_STITCH_BLOCKS
#define GC_STITCH_BLOCKS 16384u
#endif
	uint32_t blocks = sizeClass == 3 ? stitchSpillBlocks(nReads) : (nReads < GC_STITCH_BLOCKS ? nReads : GC_STITCH_BLOCKS);
#define GC_LAUNCH_STITCH(SET, CAP, SPILL) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stitch<SET, CAP, SPILL>), dim3(blocks), dim3(64), 0, stream, g, jobs, nReads, anchors, frags, fragStatus, chainOut, chainLen, chainStatus, pathPool, pathCapacity, colinearGap, setMax, bfsCap, \
		slotOf, regions, dense, denseCap, denseCursor, info, spill)
	if (sizeClass == 3) {
		(void)hipMemsetAsync(spill, 0, (size_t)blocks * stitchSpillWordsPerBlock() * sizeof(unsigned long long), stream);
		GC_LAUNCH_STITCH(STITCH_SET_SIZE_LARGE, STITCH_BFS_CAP_LARGE, true);
	}
#ifdef GC_EXPERIMENTS
	else if (sizeClass == 1) GC_LAUNCH_STITCH(STITCH_SET_SIZE_LARGE, STITCH_BFS_CAP_LARGE / 2, false);
#endif
	else GC_LAUNCH_STITCH(STITCH_SET_SIZE_LARGE, STITCH_BFS_CAP_LARGE, false);
#undef GC_LAUNCH_STITCH
}
uint32_t stitchSpillBlocks(uint32_t nReads) { return nReads < 1024u ? nReads : 1024u; }   // (blocks loop over the reads: four waves per CU on the chip's 256 CUs)

uint64_t stitchSpillWordsPerBlock() { return stitchSpillWordsPerBlockC(); }
uint64_t stitchRegionWords(uint64_t totalSlots, uint64_t nReads) { return 2 * totalSlots + 64 * nReads; }
uint64_t stitchDenseWords(uint64_t totalSlots, uint64_t nReads) { return stitchRegionWords(totalSlots, nReads); }   // a piece is at most its read's region; only the used part is downloaded

} // namespace gcdev
