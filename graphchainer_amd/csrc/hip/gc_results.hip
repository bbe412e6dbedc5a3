// The anchors of a batch as the result's dense arrays, made on the device (r5).
//
// k_build_anchors leaves one AnchorRec per (fragment, seed) SLOT - 3.6 M slots for 10 k x 10 kb reads, 1.6 M of them anchors - and the host used to fetch all of them (200 MB), walk
// them twice on all its workers (count, fill) and write the eleven anchor_* arrays of gc_result from them: 125 of the 320 ms of CPU a batch cost. What the reference keeps of a read
// (src/Aligner.cpp:695-729) is: every valid anchor of its fragments up to the first fragment that threw (`cont` is never reset), in fragment and seed order. That is a per-read
// compaction: one wave per read counts (k_anchor_counts), one block scans the reads' counts (k_anchor_scan), one wave per read writes the dense arrays (k_anchor_compact); what comes
// down is what the result holds (70 MB), and the host copies it to its place. The reads' fragment bookkeeping (seeds extended, capacity flag, failed assertion) is summed there too.
#include "gc_kernels.hpp"

namespace gcdev {

// per read: [0] anchors kept, [1] path words, [2] seeds extended (fragments before the first failed one), [3] bit 0: a fragment failed, bit 1: a fragment before it lost its anchors to a capacity
__global__ void __launch_bounds__(64) k_anchor_counts(const ReadChainJob* __restrict__ jobs, uint32_t nReads, const Fragment* __restrict__ frags, const uint32_t* __restrict__ fragStatus,
	const uint32_t* __restrict__ fragExtended, const AnchorRec* __restrict__ anchors, uint4* __restrict__ perRead, uint32_t* __restrict__ readSlotEnd)
{
	GC_RAISE_PRIO();
	const uint32_t lane = threadIdx.x;
	for (uint32_t r = blockIdx.x; r < nReads; r += gridDim.x) {
		const ReadChainJob job = jobs[r];
		uint32_t firstFailed = job.nFrags;
		for (uint32_t f = lane; f < job.nFrags; f += 64) if (fragStatus[job.fragBegin + f] == 1u) { firstFailed = f; break; }   // (a lane's fragments ascend: its first hit is its smallest)
		for (int d = 32; d > 0; d >>= 1) { const uint32_t o = __shfl_xor(firstFailed, d); firstFailed = o < firstFailed ? o : firstFailed; }
		uint32_t extended = 0, capacity = 0;
		for (uint32_t f = lane; f < firstFailed; f += 64) { extended += fragExtended[job.fragBegin + f]; capacity |= fragStatus[job.fragBegin + f] == 2u ? 1u : 0u; }
		const uint32_t slotEnd = firstFailed < job.nFrags ? frags[job.fragBegin + firstFailed].seedBegin : job.slotBegin + job.nSlots;   // (a fragment's seedBegin is its first slot)
		uint32_t count = 0, words = 0;
		for (uint32_t s = job.slotBegin + lane; s < slotEnd; s += 64) { const AnchorRec& a = anchors[s]; if (a.valid) { count++; words += a.pathLen; } }
		for (int d = 32; d > 0; d >>= 1) { count += __shfl_xor(count, d); words += __shfl_xor(words, d); extended += __shfl_xor(extended, d); capacity |= __shfl_xor(capacity, d); }
		if (lane == 0) { perRead[r] = make_uint4(count, words, extended, (firstFailed < job.nFrags ? 1u : 0u) | (capacity ? 2u : 0u)); readSlotEnd[r] = slotEnd; }
	}
}

// exclusive scan of the reads' counts: readOff[2 r] anchors before read r, readOff[2 r + 1] path words before it; the totals behind the last read and in pinned host memory
__global__ void __launch_bounds__(1024) k_anchor_scan(const uint4* __restrict__ perRead, uint32_t nReads, unsigned long long* __restrict__ readOff, unsigned long long* hostTotals)
{
	__shared__ unsigned long long partA[1024], partW[1024];
	const uint32_t tid = threadIdx.x, per = (nReads + 1023) / 1024;
	const uint32_t from = tid * per < nReads ? tid * per : nReads, to = from + per < nReads ? from + per : nReads;
	unsigned long long a = 0, w = 0;
	for (uint32_t r = from; r < to; r++) { a += perRead[r].x; w += perRead[r].y; }
	partA[tid] = a; partW[tid] = w;
	__syncthreads();
	for (uint32_t d = 1; d < 1024; d <<= 1) {
		unsigned long long oa = 0, ow = 0;
		if (tid >= d) { oa = partA[tid - d]; ow = partW[tid - d]; }
		__syncthreads();
		partA[tid] += oa; partW[tid] += ow;
		__syncthreads();
	}
	unsigned long long atA = partA[tid] - a, atW = partW[tid] - w;
	for (uint32_t r = from; r < to; r++) { readOff[2 * r] = atA; readOff[2 * r + 1] = atW; atA += perRead[r].x; atW += perRead[r].y; }
	if (tid == 1023) { readOff[2 * nReads] = partA[1023]; readOff[2 * nReads + 1] = partW[1023]; hostTotals[0] = partA[1023]; hostTotals[1] = partW[1023]; }
}

__global__ void __launch_bounds__(64) k_anchor_compact(const ReadChainJob* __restrict__ jobs, uint32_t nReads, const AnchorRec* __restrict__ anchors, const uint32_t* __restrict__ pathPool,
	const uint32_t* __restrict__ readSlotEnd, const unsigned long long* __restrict__ readOff, AnchorArrays out)
{
	GC_RAISE_PRIO();
	const uint32_t lane = threadIdx.x;
	for (uint32_t r = blockIdx.x; r < nReads; r += gridDim.x) {
		const uint32_t slotBegin = jobs[r].slotBegin, slotEnd = readSlotEnd[r];
		unsigned long long atA = readOff[2 * r], atW = readOff[2 * r + 1];
		for (uint32_t s0 = slotBegin; s0 < slotEnd; s0 += 64) {
			const uint32_t s = s0 + lane;
			AnchorRec a;
			a.valid = 0; a.pathLen = 0;
			if (s < slotEnd) a = anchors[s];
			const bool keep = a.valid != 0;
			const unsigned long long ballot = __ballot(keep);
			uint32_t wordsIncl = keep ? a.pathLen : 0u;
			for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(wordsIncl, d); if ((int)lane >= d) wordsIncl += o; }
			const uint32_t chunkWords = __shfl(wordsIncl, 63);
			if (keep) {
				const unsigned long long i = atA + (unsigned long long)__popcll(ballot & ((1ull << lane) - 1ull));
				const unsigned long long w = atW + wordsIncl - a.pathLen;
				out.x[i] = a.x; out.y[i] = a.y;
				out.pathOff[i] = w;
				out.firstNode[i] = a.firstNode; out.firstOffset[i] = a.firstOffset; out.firstSeqPos[i] = a.firstSeqPos + a.x;   // (positions inside the fragment -> in the read: the fragment begins at x)
				out.lastNode[i] = a.lastNode; out.lastOffset[i] = a.lastOffset; out.lastSeqPos[i] = a.lastSeqPos + a.x;
				out.score[i] = a.score;
				for (uint32_t k = 0; k < a.pathLen; k++) out.path[w + k] = pathPool[a.pathOff + k];
			}
			atA += (unsigned long long)__popcll(ballot);
			atW += chunkWords;
		}
	}
}

void launchAnchorCounts(hipStream_t stream, const ReadChainJob* jobs, uint32_t nReads, const Fragment* frags, const uint32_t* fragStatus, const uint32_t* fragExtended, const AnchorRec* anchors,
	uint4* perRead, uint32_t* readSlotEnd, unsigned long long* readOff, unsigned long long* hostTotals)
{
	if (nReads == 0) return;
	hipLaunchKernelGGL(k_anchor_counts, dim3(nReads < 16384 ? nReads : 16384), dim3(64), 0, stream, jobs, nReads, frags, fragStatus, fragExtended, anchors, perRead, readSlotEnd);
	hipLaunchKernelGGL(k_anchor_scan, dim3(1), dim3(1024), 0, stream, perRead, nReads, readOff, hostTotals);
}

void launchAnchorCompact(hipStream_t stream, const ReadChainJob* jobs, uint32_t nReads, const AnchorRec* anchors, const uint32_t* pathPool, const uint32_t* readSlotEnd, const unsigned long long* readOff, const AnchorArrays& out)
{
	if (nReads == 0) return;
	hipLaunchKernelGGL(k_anchor_compact, dim3(nReads < 16384 ? nReads : 16384), dim3(64), 0, stream, jobs, nReads, anchors, pathPool, readSlotEnd, readOff, out);
}

} // namespace gcdev
