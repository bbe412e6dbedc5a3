// DEFLATE on the device for the GAM writer - SURVEY.md §8 row f2 (r4).
//
// The reference writes a read's alignments as ONE gzip member (writeGAMToQueue, src/Aligner.cpp:261-281: protobuf's GzipOutputStream at zlib's default level). At the hot path's rate
// that deflate is the host's bound: ~1 ms of CPU per 10 kb read, 10-11 CPU-seconds per 10 k reads, 14.6 k reads/s end to end on the 16 CPUs of the pool's boxes against 60 k for GAF
// (DESIGN.md §3.9, §10). A GAM stream is read bases (four letters) and varint-coded path messages: most of what deflate gains on it comes from the Huffman stage, not from LZ77 matches.
// So the device writes every stream as one dynamic-Huffman block of literals (RFC 1951 §3.2.7, no length / distance codes): any inflate reads it, the inflated bytes are the
// reference's, the file is ~1.3x the size zlib's level 6 gives, and the host only frames the member (gzip header, CRC-32, length).
//
// Two kernels, one wave per stream:
//   k_deflate_plan   byte histogram (LDS atomics) -> Huffman code lengths (the symbols ranked by (count, symbol) with all lanes, then the classic two-queue merge on lane 0) ->
//                    the stream's compressed size. A code deeper than 15 bits (needs Fibonacci-like counts) or an empty stream falls back to stored blocks.
//   k_deflate_write  canonical codes from the lengths (bit-reversed: deflate packs Huffman codes MSB first into an LSB-first stream), then header and data through one bit packer:
//                    64 (code, length) items at a time, a wave prefix sum of the lengths gives every item its bit position, the items are OR-ed into an LDS window and the
//                    window's complete words go out with coalesced stores.
// Streams are placed at 4-byte aligned offsets of one dense output (exclusive scan of the planned sizes on the host side of the first kernel's result).
#include "gc_kernels.hpp"
#include <hip/hip_runtime.h>

namespace gcdev {

namespace {

#define DEFLATE_SYMS 257u            // literals 0..255 and end-of-block
#define DEFLATE_HEADER_ITEMS (5u + 19u + 258u)

__device__ __forceinline__ uint32_t reverseBits(uint32_t v, uint32_t n) { return __brev(v) >> (32u - n); }
__device__ __forceinline__ uint32_t scanInclusiveU32(uint32_t v, uint32_t lane)
{
	for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(v, d); if ((int)lane >= d) v += o; }
	return v;
}

} // namespace

// plan[stream] = { compressed bytes (without gzip framing), mode (0 dynamic Huffman block, 1 stored blocks) }; lens[stream * 260 + symbol] = code length (mode 0)
__global__ void __launch_bounds__(64) k_deflate_plan(const uint8_t* __restrict__ raw, const uint64_t* __restrict__ rawOff, uint32_t nStreams, uint8_t* __restrict__ lens, uint2* __restrict__ plan)
{
	GC_RAISE_PRIO();
	__shared__ uint32_t hist[DEFLATE_SYMS];
	__shared__ uint32_t sortedSym[DEFLATE_SYMS];      // used symbols by ascending (count, symbol)
	__shared__ uint32_t sortedW[DEFLATE_SYMS];
	__shared__ uint32_t nodeW[DEFLATE_SYMS];          // internal nodes in creation order
	__shared__ uint16_t leafParent[DEFLATE_SYMS], nodeParent[DEFLATE_SYMS];
	__shared__ uint16_t nodeDepth[DEFLATE_SYMS];
	__shared__ uint8_t symLen[DEFLATE_SYMS + 3];
	__shared__ uint32_t sMode;
	const uint32_t s = blockIdx.x, lane = threadIdx.x;
	if (s >= nStreams) return;
	const uint64_t begin = rawOff[s], n = rawOff[s + 1] - begin;
	const uint8_t* in = raw + begin;
	for (uint32_t i = lane; i < DEFLATE_SYMS; i += 64) { hist[i] = 0; symLen[i] = 0; }
	__syncthreads();
	for (uint64_t i = lane; i < n; i += 64) atomicAdd(&hist[in[i]], 1u);
	if (lane == 0) { hist[256] = 1; sMode = 0; }
	__syncthreads();
	// rank the used symbols by (count, symbol): the rank of an element is the number of smaller keys (m <= 257: a few thousand comparisons per lane)
	uint32_t used = 0;
	for (uint32_t a = lane; a < DEFLATE_SYMS; a += 64) {
		const uint32_t w = hist[a];
		if (!w) continue;
		uint32_t rank = 0;
		for (uint32_t b = 0; b < DEFLATE_SYMS; b++) { const uint32_t v = hist[b]; if (v && (v < w || (v == w && b < a))) rank++; }
		sortedSym[rank] = a; sortedW[rank] = w;
		used++;
	}
	for (int d = 32; d >= 1; d >>= 1) used += __shfl_xor(used, d);
	__syncthreads();
	const uint32_t m = used;
	if (lane == 0) {
		if (n == 0 || m < 2) sMode = 1;
		else {
			// two-queue Huffman: leaves in ascending weight, internal nodes come out in ascending weight too
			uint32_t li = 0, ni = 0, nn = 0;
			auto take = [&](uint32_t parent) -> uint32_t {
				if (li < m && (ni >= nn || sortedW[li] <= nodeW[ni])) { leafParent[li] = (uint16_t)parent; return sortedW[li++]; }
				nodeParent[ni] = (uint16_t)parent; return nodeW[ni++];
			};
			for (uint32_t k = 0; k + 1 < m; k++) { const uint32_t a = take(nn); const uint32_t b = take(nn); nodeW[nn++] = a + b; }
			nodeDepth[nn - 1] = 0;
			uint32_t deepest = 0;
			for (uint32_t k = nn - 1; k-- > 0;) nodeDepth[k] = (uint16_t)(nodeDepth[nodeParent[k]] + 1);
			for (uint32_t k = 0; k < m; k++) { const uint32_t len = (uint32_t)nodeDepth[leafParent[k]] + 1; symLen[sortedSym[k]] = (uint8_t)(len > 255 ? 255 : len); deepest = len > deepest ? len : deepest; }
			if (deepest > 15) sMode = 1;
		}
	}
	__syncthreads();
	const uint32_t mode = sMode;
	// size: header 3 + 5 + 5 + 4 + 19 x 3 + 258 x 4 bits, then the data and the end-of-block code
	unsigned long long bits = 0;
	if (mode == 0) for (uint32_t a = lane; a < DEFLATE_SYMS; a += 64) bits += (unsigned long long)hist[a] * symLen[a];
	for (int d = 32; d >= 1; d >>= 1) bits += __shfl_xor(bits, d);
	for (uint32_t a = lane; a < DEFLATE_SYMS; a += 64) lens[(uint64_t)s * 260 + a] = mode == 0 ? symLen[a] : 0;
	if (lane == 0) {
		const uint64_t blocks = n == 0 ? 1 : (n + 65534) / 65535, stored = n + 5 * blocks, huffman = (3 + 5 + 5 + 4 + 57 + 258 * 4 + bits + 7) / 8;
		const bool useStored = mode == 1 || stored <= huffman;
		plan[s] = make_uint2((uint32_t)(useStored ? stored : huffman), useStored ? 1u : 0u);
	}
}

// writes stream s at out + outOff[s] (4-byte aligned); the bytes beyond the planned size up to the next multiple of 4 are zero
__global__ void __launch_bounds__(64) k_deflate_write(const uint8_t* __restrict__ raw, const uint64_t* __restrict__ rawOff, uint32_t nStreams, const uint8_t* __restrict__ lens, const uint2* __restrict__ plan,
	uint8_t* __restrict__ out, const uint64_t* __restrict__ outOff)
{
	GC_RAISE_PRIO();
	__shared__ uint32_t code[DEFLATE_SYMS];           // (reversed code << 8) | length
	__shared__ uint32_t window[40];                    // bit window of the packer: up to 31 carried bits + 64 x 15 new ones
	__shared__ uint32_t headerItem[DEFLATE_HEADER_ITEMS];
	__shared__ uint32_t blCount[16], nextCode[16];
	const uint32_t s = blockIdx.x, lane = threadIdx.x;
	if (s >= nStreams) return;
	const uint64_t begin = rawOff[s], n = rawOff[s + 1] - begin;
	const uint8_t* in = raw + begin;
	const uint2 pl = plan[s];
	uint8_t* o = out + outOff[s];
	if (pl.y == 1) {
		// stored blocks: BFINAL / BTYPE 00 in one byte, LEN, NLEN, the bytes
		const uint64_t blocks = n == 0 ? 1 : (n + 65534) / 65535;
		for (uint64_t b = 0; b < blocks; b++) {
			const uint64_t from = b * 65535, len = n - from < 65535 ? n - from : 65535;
			uint8_t* p = o + from + 5 * b;
			if (lane == 0) { p[0] = b + 1 == blocks ? 1 : 0; p[1] = (uint8_t)len; p[2] = (uint8_t)(len >> 8); p[3] = (uint8_t)~len; p[4] = (uint8_t)(~len >> 8); }
			for (uint64_t i = lane; i < len; i += 64) p[5 + i] = in[from + i];
		}
		return;
	}
	// canonical codes (RFC 1951 §3.2.2), stored bit-reversed
	if (lane < 16) blCount[lane] = 0;
	__syncthreads();
	for (uint32_t a = lane; a < DEFLATE_SYMS; a += 64) { const uint32_t l = lens[(uint64_t)s * 260 + a]; if (l) atomicAdd(&blCount[l], 1u); }
	__syncthreads();
	if (lane == 0) { uint32_t c = 0; nextCode[0] = 0; for (uint32_t l = 1; l < 16; l++) { c = (c + blCount[l - 1]) << 1; nextCode[l] = c; } }
	__syncthreads();
	// (codes of one length go to the symbols in ascending order: rank of the symbol among those of its length)
	for (uint32_t a = lane; a < DEFLATE_SYMS; a += 64) {
		const uint32_t l = lens[(uint64_t)s * 260 + a];
		uint32_t v = 0;
		if (l) {
			uint32_t rank = 0;
			for (uint32_t b = 0; b < a; b++) rank += lens[(uint64_t)s * 260 + b] == l;
			v = (reverseBits(nextCode[l] + rank, l) << 8) | l;
		}
		code[a] = v;
	}
	// header items: (value << 8) | bits, values LSB first as deflate packs plain fields; the code-length code gives lengths 0..15 the 4-bit codes 0..15 (MSB first: reversed here)
	for (uint32_t i = lane; i < DEFLATE_HEADER_ITEMS; i += 64) {
		uint32_t v;
		if (i == 0) v = (1u << 8) | 1;                     // BFINAL
		else if (i == 1) v = (2u << 8) | 2;                // BTYPE = dynamic
		else if (i == 2) v = (0u << 8) | 5;                // HLIT: 257 literal / length codes
		else if (i == 3) v = (0u << 8) | 5;                // HDIST: 1 distance code (of zero bits: no distances at all)
		else if (i == 4) v = (15u << 8) | 4;               // HCLEN: all 19 code-length code lengths
		else if (i < 24) {
			const uint32_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
			v = ((order[i - 5] < 16 ? 4u : 0u) << 8) | 3;
		} else {
			const uint32_t a = i - 24;                          // 257 literal / length code lengths, then the distance code's
			const uint32_t l = a < DEFLATE_SYMS ? lens[(uint64_t)s * 260 + a] : 0;
			v = (reverseBits(l, 4) << 8) | 4;
		}
		headerItem[i] = v;
	}
	for (uint32_t i = lane; i < 40; i += 64) window[i] = 0;
	__syncthreads();
	// the bit packer
	uint64_t wordsOut = 0;                                    // complete 32-bit words written so far
	uint32_t carry = 0;                                       // bits in window[0] left over from the previous group
	uint32_t* o32 = (uint32_t*)o;
	auto pack = [&](uint32_t item, bool valid) {
		const uint32_t len = valid ? (item & 255u) : 0, value = item >> 8;
		const uint32_t incl = scanInclusiveU32(len, lane);
		const uint32_t at = carry + incl - len;
		if (len) {
			atomicOr(&window[at >> 5], value << (at & 31u));
			if ((at & 31u) + len > 32) atomicOr(&window[(at >> 5) + 1], value >> (32u - (at & 31u)));
		}
		__syncthreads();
		const uint32_t total = carry + __shfl(incl, 63);
		const uint32_t full = total >> 5;
		if (lane < full) o32[wordsOut + lane] = window[lane];
		const uint32_t rest = lane == 0 ? window[full] : 0;
		__syncthreads();
		if (lane < 40) window[lane] = 0;                     // (64 lanes cover the 40 words)
		__syncthreads();
		if (lane == 0) window[0] = rest;
		__syncthreads();
		wordsOut += full;
		carry = total & 31u;
	};
	for (uint32_t i = 0; i < DEFLATE_HEADER_ITEMS; i += 64) pack(i + lane < DEFLATE_HEADER_ITEMS ? headerItem[i + lane] : 0, i + lane < DEFLATE_HEADER_ITEMS);
	for (uint64_t i = 0; i < n; i += 64) pack(i + lane < n ? code[in[i + lane]] : 0, i + lane < n);
	pack(code[256], lane == 0);
	if (carry && lane == 0) o32[wordsOut] = window[0];
}

void launchDeflatePlan(hipStream_t stream, const uint8_t* raw, const uint64_t* rawOff, uint32_t nStreams, uint8_t* lens, uint2* plan)
{
	if (nStreams) hipLaunchKernelGGL(k_deflate_plan, dim3(nStreams), dim3(64), 0, stream, raw, rawOff, nStreams, lens, plan);
}
void launchDeflateWrite(hipStream_t stream, const uint8_t* raw, const uint64_t* rawOff, uint32_t nStreams, const uint8_t* lens, const uint2* plan, uint8_t* out, const uint64_t* outOff)
{
	if (nStreams) hipLaunchKernelGGL(k_deflate_write, dim3(nStreams), dim3(64), 0, stream, raw, rawOff, nStreams, lens, plan, out, outOff);
}

} // namespace gcdev
