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

// =====================================================================================================================================================================
// r6: LZ77 matches in front of the Huffman stage (GC_GAM_DEVICE_LZ). A GAM group is a read's letters and a few hundred Mapping messages whose tags, lengths and small
// varints repeat: literal-only Huffman blocks are twice zlib's bytes, a one-probe hash of 4-byte prefixes brings that to ~1.15-1.25 x (measured on the oracle's groups
// with the executable model of this code that zlib's inflate checks, tests/deflate_model.py).
// The tokens are a function of the bytes alone, so they are made twice - once to count symbols (plan), once to write them - and never stored:
//   per 64 positions, lane = position: hash of the 4 bytes there -> the last earlier position with that hash (LDS table of position + 1, filled by atomicMax after all
//   lanes have looked: candidates come from earlier chunks) -> match length by comparing bytes (<= 258, distance <= 32 768, >= 4 to count);
//   greedy parse: the token starts of the chunk are the orbit of the entry position under "next = position + (match length or 1)" - six doubling rounds (a reach mask in
//   LDS, the jump function squared by a lane permute) instead of a walk over 64 positions; the exit position enters the next chunk;
//   a start lane's token is a literal or (length symbol + extra bits, distance symbol + extra bits) by the closed forms of RFC 1951's tables.
// lens[stream * 320 + s]: code lengths of the 286 literal / length symbols, then of the 30 distance symbols. Two-queue Huffman per alphabet as above; a code deeper than
// 15 bits or an empty stream falls back to stored blocks.
#define LZ_TBITS 12u
#define LZ_LL 286u
#define LZ_D 30u
#define LZ_STRIDE 320u
#define LZ_HEADER_ITEMS (5u + 19u + LZ_LL + LZ_D)

namespace {

struct LzToken { bool start; uint32_t len, dist, lit; };

__device__ __forceinline__ uint32_t lzLenSymbol(uint32_t L, uint32_t& extraBits, uint32_t& extra)
{
	if (L == 258) { extraBits = 0; extra = 0; return 28; }
	const uint32_t v = L - 3;
	if (v < 8) { extraBits = 0; extra = 0; return v; }
	const uint32_t k = 31u - (uint32_t)__clz((int)v);
	extraBits = k - 2; extra = v & ((1u << (k - 2)) - 1u);
	return 4 * (k - 1) + ((v >> (k - 2)) & 3u);
}
__device__ __forceinline__ uint32_t lzDistSymbol(uint32_t D, uint32_t& extraBits, uint32_t& extra)
{
	const uint32_t v = D - 1;
	if (v < 4) { extraBits = 0; extra = 0; return v; }
	const uint32_t k = 31u - (uint32_t)__clz((int)v);
	extraBits = k - 1; extra = v & ((1u << (k - 1)) - 1u);
	return 2 * k + ((v >> (k - 1)) & 1u);
}

// the tokens that start in [p, p + 64); cursor: the first position not yet covered by a token (uniform, advanced)
__device__ __forceinline__ LzToken lzChunk(const uint8_t* __restrict__ in, uint64_t n, uint64_t p, uint32_t lane, uint32_t* table, uint32_t* reach, uint64_t& cursor)
{
	LzToken t { false, 0, 0, 0 };
	const uint64_t pos = p + lane;
	const bool inside = pos < n, hashable = pos + 4 <= n;
	uint32_t b0 = inside ? in[pos] : 0u, h = 0, cand = 0;
	if (hashable) {
		const uint32_t key = b0 | ((uint32_t)in[pos + 1] << 8) | ((uint32_t)in[pos + 2] << 16) | ((uint32_t)in[pos + 3] << 24);
		h = (key * 2654435761u) >> (32u - LZ_TBITS);
		cand = table[h];
	}
	__syncthreads();
	if (hashable) atomicMax(&table[h], (uint32_t)pos + 1u);
	t.lit = b0;
	if (cand && pos - (cand - 1) <= 32768ull) {
		const uint64_t c = cand - 1;
		const uint32_t most = n - pos < 258 ? (uint32_t)(n - pos) : 258u;
		uint32_t l = 0;
		while (l < most && in[c + l] == in[pos + l]) l++;
		if (l >= 4) { t.len = l; t.dist = (uint32_t)(pos - c); }
	}
	__syncthreads();
	if (cursor >= p + 64) return t;   // a match of an earlier chunk covers this one
	const uint32_t entry = (uint32_t)(cursor - p), step = t.len ? t.len : 1u;
	uint32_t J = lane + step > 64u ? 64u : lane + step;
	bool S = inside && lane == entry;
	for (int k = 0; k < 6; k++) {
		if (lane < 2) reach[lane] = 0;
		__syncthreads();
		if (S && J < 64) atomicOr(&reach[J >> 5], 1u << (J & 31u));
		__syncthreads();
		S = S || (inside && ((reach[lane >> 5] >> (lane & 31u)) & 1u));
		const uint32_t J2 = __shfl(J, J < 64 ? J : 0u);
		J = J < 64 ? J2 : 64u;
		__syncthreads();
	}
	t.start = S;
	uint32_t end = S ? lane + step : 0u;
	for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(end, d); end = o > end ? o : end; }
	if (end) cursor = p + end;
	return t;
}

// Huffman code lengths of one alphabet by the two-queue merge (lane 0) over the symbols ranked by (count, symbol) (all lanes); returns the deepest code (0: no symbol, 1: one symbol)
__device__ __forceinline__ uint32_t lzBuildLengths(const uint32_t* hist, uint32_t nSyms, uint8_t* symLen, uint32_t* sortedSym, uint32_t* sortedW, uint32_t* nodeW, uint16_t* leafParent, uint16_t* nodeParent,
	uint16_t* nodeDepth, uint32_t* sDeepest, uint32_t lane)
{
	for (uint32_t a = lane; a < nSyms; a += 64) symLen[a] = 0;
	uint32_t used = 0;
	for (uint32_t a = lane; a < nSyms; a += 64) {
		const uint32_t w = hist[a];
		if (!w) continue;
		uint32_t rank = 0;
		for (uint32_t b = 0; b < nSyms; b++) { const uint32_t v = hist[b]; if (v && (v < w || (v == w && b < a))) rank++; }
		sortedSym[rank] = a; sortedW[rank] = w;
		used++;
	}
	for (int d = 32; d >= 1; d >>= 1) used += __shfl_xor(used, d);
	__syncthreads();
	const uint32_t m = used;
	if (lane == 0) {
		uint32_t deepest = 0;
		if (m == 1) { symLen[sortedSym[0]] = 1; deepest = 1; }
		else if (m >= 2) {
			uint32_t li = 0, ni = 0, nn = 0;
			auto take = [&](uint32_t parent) -> uint32_t {
				if (li < m && (ni >= nn || sortedW[li] <= nodeW[ni])) { leafParent[li] = (uint16_t)parent; return sortedW[li++]; }
				nodeParent[ni] = (uint16_t)parent; return nodeW[ni++];
			};
			for (uint32_t k = 0; k + 1 < m; k++) { const uint32_t a = take(nn); const uint32_t b = take(nn); nodeW[nn++] = a + b; }
			nodeDepth[nn - 1] = 0;
			for (uint32_t k = nn - 1; k-- > 0;) nodeDepth[k] = (uint16_t)(nodeDepth[nodeParent[k]] + 1);
			for (uint32_t k = 0; k < m; k++) { const uint32_t len = (uint32_t)nodeDepth[leafParent[k]] + 1; symLen[sortedSym[k]] = (uint8_t)(len > 255 ? 255 : len); deepest = len > deepest ? len : deepest; }
		}
		*sDeepest = deepest;
	}
	__syncthreads();
	return *sDeepest;
}

} // namespace

__global__ void __launch_bounds__(64) k_deflate_lz_plan(const uint8_t* __restrict__ raw, const uint64_t* __restrict__ rawOff, uint32_t nStreams, uint8_t* __restrict__ lens, uint2* __restrict__ plan)
{
	GC_RAISE_PRIO();
	__shared__ uint32_t table[1u << LZ_TBITS];
	__shared__ uint32_t reach[2];
	__shared__ uint32_t histLL[LZ_LL], histD[32];
	__shared__ uint32_t sortedSym[LZ_LL], sortedW[LZ_LL], nodeW[LZ_LL];
	__shared__ uint16_t leafParent[LZ_LL], nodeParent[LZ_LL], nodeDepth[LZ_LL];
	__shared__ uint8_t lenLL[LZ_LL + 2], lenD[32];
	__shared__ uint32_t sDeepest;
	const uint32_t s = blockIdx.x, lane = threadIdx.x;
	if (s >= nStreams) return;
	const uint64_t begin = rawOff[s], n = rawOff[s + 1] - begin;
	const uint8_t* in = raw + begin;
	for (uint32_t i = lane; i < (1u << LZ_TBITS); i += 64) table[i] = 0;
	for (uint32_t i = lane; i < LZ_LL; i += 64) histLL[i] = 0;
	if (lane < 32) histD[lane] = 0;
	__syncthreads();
	uint64_t cursor = 0;
	unsigned long long extraBits = 0;
	for (uint64_t p = 0; p < n; p += 64) {
		const LzToken t = lzChunk(in, n, p, lane, table, reach, cursor);
		if (t.start) {
			if (t.len) {
				uint32_t eb, ev, fb, fv;
				atomicAdd(&histLL[257u + lzLenSymbol(t.len, eb, ev)], 1u);
				atomicAdd(&histD[lzDistSymbol(t.dist, fb, fv)], 1u);
				extraBits += eb + fb;
			} else atomicAdd(&histLL[t.lit], 1u);
		}
	}
	if (lane == 0) histLL[256] = 1;
	__syncthreads();
	const uint32_t deepLL = lzBuildLengths(histLL, LZ_LL, lenLL, sortedSym, sortedW, nodeW, leafParent, nodeParent, nodeDepth, &sDeepest, lane);
	const uint32_t deepD = lzBuildLengths(histD, LZ_D, lenD, sortedSym, sortedW, nodeW, leafParent, nodeParent, nodeDepth, &sDeepest, lane);
	const bool codeOk = n > 0 && deepLL >= 2 && deepLL <= 15 && deepD <= 15;
	unsigned long long bits = extraBits;
	if (codeOk) {
		for (uint32_t a = lane; a < LZ_LL; a += 64) bits += (unsigned long long)histLL[a] * lenLL[a];
		if (lane < LZ_D) bits += (unsigned long long)histD[lane] * lenD[lane];
	}
	for (int d = 32; d >= 1; d >>= 1) bits += __shfl_xor(bits, d);
	for (uint32_t a = lane; a < LZ_STRIDE; a += 64) lens[(uint64_t)s * LZ_STRIDE + a] = !codeOk ? 0 : a < LZ_LL ? lenLL[a] : a < LZ_LL + LZ_D ? lenD[a - LZ_LL] : 0;
	if (lane == 0) {
		const uint64_t blocks = n == 0 ? 1 : (n + 65534) / 65535, stored = n + 5 * blocks, huffman = (3 + 5 + 5 + 4 + 57 + (LZ_LL + LZ_D) * 4 + bits + 7) / 8;
		const bool useStored = !codeOk || stored <= huffman;
		plan[s] = make_uint2((uint32_t)(useStored ? stored : huffman), useStored ? 1u : 0u);
	}
}

__global__ void __launch_bounds__(64) k_deflate_lz_write(const uint8_t* __restrict__ raw, const uint64_t* __restrict__ rawOff, uint32_t nStreams, const uint8_t* __restrict__ lens, const uint2* __restrict__ plan,
	uint8_t* __restrict__ out, const uint64_t* __restrict__ outOff)
{
	GC_RAISE_PRIO();
	__shared__ uint32_t table[1u << LZ_TBITS];
	__shared__ uint32_t reach[2];
	__shared__ uint32_t code[LZ_LL + LZ_D];             // (reversed code << 8) | length; the distance symbols behind the literal / length ones
	__shared__ uint32_t window[104];                     // bit window of the packer: up to 31 carried bits + 64 x 48 new ones
	__shared__ uint32_t headerItem[LZ_HEADER_ITEMS];
	__shared__ uint32_t blCount[2][16], nextCode[2][16];
	const uint32_t s = blockIdx.x, lane = threadIdx.x;
	if (s >= nStreams) return;
	const uint64_t begin = rawOff[s], n = rawOff[s + 1] - begin;
	const uint8_t* in = raw + begin;
	const uint2 pl = plan[s];
	uint8_t* o = out + outOff[s];
	if (pl.y == 1) {
		const uint64_t blocks = n == 0 ? 1 : (n + 65534) / 65535;
		for (uint64_t b = 0; b < blocks; b++) {
			const uint64_t from = b * 65535, len = n - from < 65535 ? n - from : 65535;
			uint8_t* q = o + from + 5 * b;
			if (lane == 0) { q[0] = b + 1 == blocks ? 1 : 0; q[1] = (uint8_t)len; q[2] = (uint8_t)(len >> 8); q[3] = (uint8_t)~len; q[4] = (uint8_t)(~len >> 8); }
			for (uint64_t i = lane; i < len; i += 64) q[5 + i] = in[from + i];
		}
		return;
	}
	const uint8_t* L = lens + (uint64_t)s * LZ_STRIDE;
	if (lane < 32) blCount[lane >> 4][lane & 15] = 0;
	for (uint32_t i = lane; i < (1u << LZ_TBITS); i += 64) table[i] = 0;
	__syncthreads();
	for (uint32_t a = lane; a < LZ_LL + LZ_D; a += 64) { const uint32_t l = L[a]; if (l) atomicAdd(&blCount[a < LZ_LL ? 0 : 1][l], 1u); }
	__syncthreads();
	if (lane < 2) { uint32_t c = 0; nextCode[lane][0] = 0; for (uint32_t l = 1; l < 16; l++) { c = (c + blCount[lane][l - 1]) << 1; nextCode[lane][l] = c; } }
	__syncthreads();
	for (uint32_t a = lane; a < LZ_LL + LZ_D; a += 64) {
		const uint32_t l = L[a], which = a < LZ_LL ? 0u : 1u, first = which ? LZ_LL : 0u;
		uint32_t v = 0;
		if (l) {
			uint32_t rank = 0;
			for (uint32_t b = first; b < a; b++) rank += L[b] == l;
			v = (reverseBits(nextCode[which][l] + rank, l) << 8) | l;
		}
		code[a] = v;
	}
	for (uint32_t i = lane; i < LZ_HEADER_ITEMS; i += 64) {
		uint32_t v;
		if (i == 0) v = (1u << 8) | 1;                     // BFINAL
		else if (i == 1) v = (2u << 8) | 2;                // BTYPE = dynamic
		else if (i == 2) v = ((LZ_LL - 257u) << 8) | 5;    // HLIT
		else if (i == 3) v = ((LZ_D - 1u) << 8) | 5;       // HDIST
		else if (i == 4) v = (15u << 8) | 4;               // HCLEN: all 19 code-length code lengths
		else if (i < 24) {
			const uint32_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
			v = ((order[i - 5] < 16 ? 4u : 0u) << 8) | 3;
		} else v = (reverseBits(L[i - 24], 4) << 8) | 4;   // the 316 code lengths, each by the 4-bit code of its value
		headerItem[i] = v;
	}
	for (uint32_t i = lane; i < 104; i += 64) window[i] = 0;
	__syncthreads();
	uint64_t wordsOut = 0;
	uint32_t carry = 0;
	uint32_t* o32 = (uint32_t*)o;
	auto pack = [&](unsigned long long value, uint32_t len) {   // len <= 48 bits of value, LSB first
		const uint32_t incl = scanInclusiveU32(len, lane);
		const uint32_t at = carry + incl - len;
		if (len) {
			const uint32_t sh = at & 31u, w = at >> 5;
			const unsigned long long lo = value << sh;
			const uint32_t hi = sh ? (uint32_t)(value >> (64u - sh)) : 0u;
			if ((uint32_t)lo) atomicOr(&window[w], (uint32_t)lo);
			if ((uint32_t)(lo >> 32)) atomicOr(&window[w + 1], (uint32_t)(lo >> 32));
			if (hi) atomicOr(&window[w + 2], hi);
		}
		__syncthreads();
		const uint32_t total = carry + __shfl(incl, 63);
		const uint32_t full = total >> 5;
		for (uint32_t i = lane; i < full; i += 64) o32[wordsOut + i] = window[i];
		const uint32_t rest = lane == 0 ? window[full] : 0;
		__syncthreads();
		for (uint32_t i = lane; i < 104; i += 64) window[i] = 0;
		__syncthreads();
		if (lane == 0) window[0] = rest;
		__syncthreads();
		wordsOut += full;
		carry = total & 31u;
	};
	for (uint32_t i = 0; i < LZ_HEADER_ITEMS; i += 64) { const uint32_t it = i + lane < LZ_HEADER_ITEMS ? headerItem[i + lane] : 0; pack(it >> 8, it & 255u); }
	uint64_t cursor = 0;
	for (uint64_t p = 0; p < n; p += 64) {
		const LzToken t = lzChunk(in, n, p, lane, table, reach, cursor);
		unsigned long long value = 0;
		uint32_t len = 0;
		if (t.start) {
			if (t.len) {
				uint32_t eb, ev, fb, fv;
				const uint32_t cl = code[257u + lzLenSymbol(t.len, eb, ev)], cd = code[LZ_LL + lzDistSymbol(t.dist, fb, fv)];
				value = cl >> 8; len = cl & 255u;
				value |= (unsigned long long)ev << len; len += eb;
				value |= (unsigned long long)(cd >> 8) << len; len += cd & 255u;
				value |= (unsigned long long)fv << len; len += fb;
			} else { const uint32_t c = code[t.lit]; value = c >> 8; len = c & 255u; }
		}
		pack(value, len);
	}
	{ const uint32_t c = code[256]; pack(lane == 0 ? c >> 8 : 0, lane == 0 ? c & 255u : 0); }
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
uint32_t deflateLzLensStride() { return LZ_STRIDE; }
void launchDeflateLzPlan(hipStream_t stream, const uint8_t* raw, const uint64_t* rawOff, uint32_t nStreams, uint8_t* lens, uint2* plan)
{
	if (nStreams) hipLaunchKernelGGL(k_deflate_lz_plan, dim3(nStreams), dim3(64), 0, stream, raw, rawOff, nStreams, lens, plan);
}
void launchDeflateLzWrite(hipStream_t stream, const uint8_t* raw, const uint64_t* rawOff, uint32_t nStreams, const uint8_t* lens, const uint2* plan, uint8_t* out, const uint64_t* outOff)
{
	if (nStreams) hipLaunchKernelGGL(k_deflate_lz_write, dim3(nStreams), dim3(64), 0, stream, raw, rawOff, nStreams, lens, plan, out, outOff);
}

} // namespace gcdev
