// Read batch preparation on the device: gc_reads_upload copies the raw bases once and these kernels derive everything the
// hot path reads - the reverse-complemented strand (GraphAligner aligns the prefix of a seed on it, src/GraphAligner.h:499-505),
// the per-read match-mask bit vectors of both strands (IUPAC aware: characterMatch, src/GraphAlignerCommon.h:190-297; a DP slice's
// EqVector is a 64-bit window of them, ...Common.h:280-319), the exact-match bit vectors of the NW kernels, and the 2-bit packing
// the seed kernel cuts its k-mers from (src/MinimizerSeeder.cpp:59-102). The host used to build all of this in serial loops:
// 2.6 s per 10 000 x 10 kb reads, eight times the alignment itself; here it is a few passes over 100 MB at HBM speed.
#include "gc_kernels.hpp"
#include <hip/hip_runtime.h>

namespace gcdev {

namespace {

// set of bases (A=1, C=2, G=4, T=8) a read character can stand for; 0 = not a nucleotide letter (the reference's Complement() asserts)
__device__ __forceinline__ uint32_t iupacSet(char c)
{
	switch (c) {
		case 'A': case 'a': return 1; case 'C': case 'c': return 2; case 'G': case 'g': return 4; case 'T': case 't': case 'U': case 'u': return 8;
		case 'R': case 'r': return 1 | 4; case 'Y': case 'y': return 2 | 8; case 'K': case 'k': return 4 | 8; case 'M': case 'm': return 1 | 2;
		case 'S': case 's': return 2 | 4; case 'W': case 'w': return 1 | 8;
		case 'B': case 'b': return 2 | 4 | 8; case 'D': case 'd': return 1 | 4 | 8; case 'H': case 'h': return 1 | 2 | 8; case 'V': case 'v': return 1 | 2 | 4;
		case 'N': case 'n': return 15;
	}
	return 0;
}
// CommonUtils::Complement, src/CommonUtils.cpp:78-134; 'N' for a character it would assert on (the read is flagged)
__device__ __forceinline__ char complementOf(char c)
{
	switch (c) {
		case 'A': case 'a': return 'T'; case 'C': case 'c': return 'G'; case 'G': case 'g': return 'C'; case 'T': case 't': case 'U': case 'u': return 'A';
		case 'R': case 'r': return 'Y'; case 'Y': case 'y': return 'R'; case 'K': case 'k': return 'M'; case 'M': case 'm': return 'K';
		case 'S': case 's': return 'S'; case 'W': case 'w': return 'W'; case 'B': case 'b': return 'V'; case 'V': case 'v': return 'B';
		case 'D': case 'd': return 'H'; case 'H': case 'h': return 'D';
	}
	return 'N';
}

} // namespace

// One wave per read. Lane l takes the 64-base words l, l + 64, ... of the read: forward word w = bases [64w, 64w + 64), the same
// word of the reverse-complement strand = forward bases len-1-64w downwards. Writes the strand copy, 8 match-mask words, 4 exact words.
__global__ void __launch_bounds__(64) k_pack_read_masks(const uint64_t* __restrict__ readOff, uint32_t nReads, uint64_t totalBases, char* __restrict__ bases /* [2*total]: forward in, rc out */,
	const uint64_t* __restrict__ maskOff, const uint32_t* __restrict__ maskWords, uint64_t* __restrict__ masks, const uint64_t* __restrict__ eqOff, uint64_t* __restrict__ eqMasks,
	uint8_t* __restrict__ readInvalid)
{
	GC_RAISE_PRIO();
	const uint32_t r = blockIdx.x, lane = threadIdx.x;
	if (r >= nReads) return;
	const uint64_t a = readOff[r], len = readOff[r + 1] - a;
	const uint32_t words = maskWords[r];
	uint64_t* m = masks + maskOff[r];
	uint64_t* eq = eqMasks + eqOff[r];
	const char* fw = bases + a;
	char* rc = bases + totalBases + a;
	bool invalid = false;
	for (uint32_t w = lane; w < words; w += 64) {
		uint64_t f[4] = { 0, 0, 0, 0 }, v[4] = { 0, 0, 0, 0 }, e[4] = { 0, 0, 0, 0 };
		const uint64_t base = 64ull * w;
		for (uint32_t i = 0; i < 64 && base + i < len; i++) {
			const uint64_t bit = 1ull << i;
			const char c = fw[base + i];
			const uint32_t s = iupacSet(c);
			invalid |= s == 0;
			f[0] |= (s & 1) ? bit : 0; f[1] |= (s & 2) ? bit : 0; f[2] |= (s & 4) ? bit : 0; f[3] |= (s & 8) ? bit : 0;
			e[0] |= c == 'A' ? bit : 0; e[1] |= c == 'C' ? bit : 0; e[2] |= c == 'G' ? bit : 0; e[3] |= c == 'T' ? bit : 0;
			const char d = complementOf(fw[len - 1 - (base + i)]);   // base i of this word on the reverse-complement strand
			rc[base + i] = d;
			const uint32_t t = iupacSet(d);
			v[0] |= (t & 1) ? bit : 0; v[1] |= (t & 2) ? bit : 0; v[2] |= (t & 4) ? bit : 0; v[3] |= (t & 8) ? bit : 0;
		}
		for (int b = 0; b < 4; b++) {
			m[(uint64_t)b * words + w] = f[b];
			m[(uint64_t)(4 + b) * words + w] = v[b];
			eq[(uint64_t)b * words + w] = e[b];
		}
	}
	const bool any = __any(invalid);
	if (lane == 0) readInvalid[r] = any ? 1 : 0;
}

// One thread per 64 forward bases: the two 2-bit words (32 bases each, first base in the top bits), the "not A, C, G or T" bits
// (first base in the top bit) and the read that holds the chunk's first base.
__global__ void __launch_bounds__(256) k_pack_read_kmers(const char* __restrict__ bases, uint64_t totalBases, const uint64_t* __restrict__ readOff, uint32_t nReads,
	uint64_t* __restrict__ packed, uint64_t* __restrict__ invalidBits, uint32_t* __restrict__ chunkRead)
{
	GC_RAISE_PRIO();
	const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t nChunks = (totalBases >> 6) + 1;
	if (c >= nChunks) return;
	const uint64_t p0 = c << 6;
	uint64_t w[2] = { 0, 0 }, inv = 0;
	for (uint32_t i = 0; i < 64 && p0 + i < totalBases; i++) {
		int code = -1;
		switch (bases[p0 + i]) { case 'a': case 'A': code = 0; break; case 'c': case 'C': code = 1; break; case 'g': case 'G': code = 2; break; case 't': case 'T': code = 3; break; }
		if (code < 0) inv |= 1ull << (63 - i);
		else w[i >> 5] |= (uint64_t)code << (2 * (31 - (i & 31)));
	}
	invalidBits[c] = inv;
	// packed has (totalBases >> 5) + 1 words: the second word of the last chunk may lie beyond it
	const uint64_t nPacked = (totalBases >> 5) + 1;
	if (2 * c < nPacked) packed[2 * c] = w[0];
	if (2 * c + 1 < nPacked) packed[2 * c + 1] = w[1];
	// read containing base p0 (the last read for the chunk past the end): largest r with readOff[r] <= p0, skipping empty reads like the host did
	uint32_t lo = 0, hi = nReads ? nReads - 1 : 0;
	while (lo < hi) {
		const uint32_t mid = (lo + hi + 1) >> 1;
		if (readOff[mid] <= p0) lo = mid; else hi = mid - 1;
	}
	// the host loop advanced while offsets[r + 1] <= p: for p inside read r that is the r with readOff[r] <= p < readOff[r + 1]
	while (lo + 1 < nReads && readOff[lo + 1] <= p0) lo++;
	chunkRead[c] = lo;
}

void launchPackReads(hipStream_t stream, const uint64_t* readOff, uint32_t nReads, uint64_t totalBases, char* bases, const uint64_t* maskOff, const uint32_t* maskWords, uint64_t* masks,
	const uint64_t* eqOff, uint64_t* eqMasks, uint8_t* readInvalid, uint64_t* packed, uint64_t* invalidBits, uint32_t* chunkRead)
{
	if (nReads) hipLaunchKernelGGL(k_pack_read_masks, dim3(nReads), dim3(64), 0, stream, readOff, nReads, totalBases, bases, maskOff, maskWords, masks, eqOff, eqMasks, readInvalid);
	const uint64_t nChunks = (totalBases >> 6) + 1;
	hipLaunchKernelGGL(k_pack_read_kmers, dim3((uint32_t)((nChunks + 255) / 256)), dim3(256), 0, stream, bases, totalBases, readOff, nReads, packed, invalidBits, chunkRead);
}

} // namespace gcdev
