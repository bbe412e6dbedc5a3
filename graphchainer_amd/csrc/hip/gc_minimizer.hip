// Minimizer index construction on the device (start-up accelerated). Same index as host/gc_minimizer.cpp builds
// (reference: src/MinimizerSeeder.cpp:104-189 window minimizers, :299-492 initMinimizers): every window minimizer of every bigraph
// node, all hash ties the reference reports, position lists in reverse arrival order (arrival = nodeLookup iteration x position).
//
// The window scan is a small state machine per node (a monotone deque of at most w - k + 2 k-mers with the reference's tie rules),
// sequential inside a node and independent across nodes: one lane per bigraph node, in arrival order, reading the node's letters
// from the 2-bit / one-hot split-node sequences already in HBM. Pass 1 counts a node's reports, an exclusive scan places them,
// pass 2 writes (k-mer << 34 | reversed arrival index, packed position), one radix sort over the 64-bit keys groups the k-mers and
// leaves every list in the reference's order. On config 2 (101 M graph bases, 28 M reports) the host builder took 6.2 s on 256
// threads (its final sort is serial).
#include "gc_kernels.hpp"
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

namespace gcdev {

namespace {

#define MZ_DEQUE 32   // w - k + 2 entries at most (w - k + 1 <= 30)

__device__ __forceinline__ uint64_t minimizerHashDev(uint64_t key)   // src/MinimizerSeeder.cpp:45-54
{
	key = (~key) + (key << 21);
	key = key ^ (key >> 24);
	key = (key + (key << 3)) + (key << 8);
	key = key ^ (key >> 14);
	key = (key + (key << 2)) + (key << 4);
	key = key ^ (key >> 28);
	key = key + (key << 31);
	return key;
}

// base code 0..3 of position p of bigraph node id, -1 for anything else (an IUPAC letter of an ambiguous split node)
// p == size answers -1 too: the reference's first window reads one character past w (the std::string's terminator when the node ends there),
// which is what makes a run of exactly w valid letters report nothing (:131-135)
__device__ __forceinline__ int baseAt(const DGraph& g, int32_t id, uint32_t size, uint32_t p)
{
	if (p >= size) return -1;
	const uint32_t node = g.lookup[g.lookupOff[id] + p / 64], off = p & 63u;
	if (node < g.firstAmbiguous) return (int)((g.nodeSeq[2 * (size_t)node + (off >> 5)] >> ((off & 31) * 2)) & 3);
	const uint64_t* q = g.ambSeq + 4 * (size_t)(node - g.firstAmbiguous);
	const uint32_t mask = (uint32_t)((q[0] >> off) & 1) | (uint32_t)(((q[1] >> off) & 1) << 1) | (uint32_t)(((q[2] >> off) & 1) << 2) | (uint32_t)(((q[3] >> off) & 1) << 3);
	return mask == 1 ? 0 : mask == 2 ? 1 : mask == 4 ? 2 : mask == 8 ? 3 : -1;
}

struct DequeEntry { uint32_t pos; uint32_t kmer; uint64_t hash; };

// iterateMinimizersReal (src/MinimizerSeeder.cpp:104-189) over one node; emit(pos, kmer) per report. The deque lives in LDS, MZ_DEQUE
// entries per lane, as a ring.
template <typename F>
__device__ __forceinline__ void scanNode(const DGraph& g, int32_t id, uint32_t size, uint32_t k, uint32_t w, DequeEntry* dq, F&& emit)
{
	if (size < k) return;
	const uint32_t kmersPerWindow = w - k + 1;
	const uint64_t mask = ~(~(uint64_t)0 << (k * 2));
	uint32_t head = 0, count = 0;   // ring: entries head .. head + count - 1
	auto at = [&](uint32_t i) -> DequeEntry& { return dq[(head + i) & (MZ_DEQUE - 1)]; };
	uint32_t offset = 0;
	while (true) {
		while (offset < size && baseAt(g, id, size, offset) < 0) offset++;
		if (offset + w > size) return;
		uint64_t kmer = 0;
		bool restart = false;
		for (uint32_t i = 0; i < k; i++) {
			int c = baseAt(g, id, size, offset + i);
			if (c < 0) { offset += i; restart = true; break; }
			kmer = (kmer << 2) | (uint64_t)c;
		}
		if (restart) continue;
		head = 0; count = 0;
		at(count++) = DequeEntry { offset + k - 1, (uint32_t)kmer, minimizerHashDev(kmer) };
		for (uint32_t i = k; i < k + kmersPerWindow; i++) {
			int c = baseAt(g, id, size, offset + i);
			if (c < 0) { offset += i; restart = true; break; }
			kmer = ((kmer << 2) & mask) | (uint64_t)c;
			const uint64_t h = minimizerHashDev(kmer);
			while (count > 0 && at(count - 1).hash > h) count--;
			at(count++) = DequeEntry { offset + i, (uint32_t)kmer, h };
		}
		if (restart) continue;
		for (uint32_t e = 0; e < count && at(e).hash == at(0).hash; e++) emit(at(e).pos, at(e).kmer);
		for (uint32_t i = k + kmersPerWindow; offset + i < size; i++) {
			int c = baseAt(g, id, size, offset + i);
			if (c < 0) { offset += i; restart = true; break; }
			kmer = ((kmer << 2) & mask) | (uint64_t)c;
			const uint64_t h = minimizerHashDev(kmer);
			const uint64_t oldMinimum = at(0).hash;
			bool frontPopped = false;
			while (count > 0 && at(0).pos <= offset + i - kmersPerWindow) { frontPopped = true; head++; count--; }
			if (frontPopped) while (count >= 2 && at(0).hash == at(1).hash) { head++; count--; }
			while (count > 0 && at(count - 1).hash > h) count--;
			at(count++) = DequeEntry { offset + i, (uint32_t)kmer, h };
			if (at(0).hash != oldMinimum) {
				for (uint32_t e = 0; e < count && at(e).hash == at(0).hash; e++) emit(at(e).pos, at(e).kmer);
			} else if (at(count - 1).hash == at(0).hash) {
				emit(at(count - 1).pos, at(count - 1).kmer);
			}
		}
		if (!restart) return;
	}
}

} // namespace

// PASS 1: counts[i] = reports of node idOrder[i]; PASS 2 (keys != nullptr): writes them at offsets[i]
__global__ void __launch_bounds__(64) k_minimizer_scan(DGraph g, const int32_t* __restrict__ idOrder, uint32_t nIds, uint32_t k, uint32_t w, const uint64_t* __restrict__ offsets,
	uint32_t* __restrict__ counts, uint64_t* __restrict__ keys, uint64_t* __restrict__ values)
{
	__shared__ DequeEntry deques[64 * MZ_DEQUE];
	const uint32_t i = blockIdx.x * 64 + threadIdx.x;
	if (i >= nIds) return;
	DequeEntry* dq = deques + (size_t)threadIdx.x * MZ_DEQUE;
	const int32_t id = idOrder[i];
	const uint32_t size = g.origSize[id];
	uint32_t n = 0;
	if (!keys) {
		scanNode(g, id, size, k, w, dq, [&](uint32_t, uint32_t) { n++; });
		counts[i] = n;
	} else {
		const uint64_t base = offsets[i];
		scanNode(g, id, size, k, w, dq, [&](uint32_t pos, uint32_t kmer) {
			const uint32_t split = g.lookup[g.lookupOff[id] + pos / 64];
			const uint64_t arrival = base + n;
			keys[arrival] = ((uint64_t)kmer << 34) | ((1ull << 34) - 1 - arrival);   // ascending key = ascending k-mer, then DEscending arrival (:473-482)
			values[arrival] = ((uint64_t)split << 6) + (pos - g.nodeOffset[split]);
			n++;
		});
	}
}

// Returns the number of reports; *outKeys / *outValues (device, hipMalloc'd, caller frees) hold them sorted. idOrderDev: bigraph node ids in
// arrival order.
uint64_t buildMinimizerPairsDevice(const DGraph& g, const int32_t* idOrderDev, uint32_t nIds, uint32_t k, uint32_t w, uint64_t** outKeys, uint64_t** outValues)
{
	*outKeys = *outValues = nullptr;
	if (nIds == 0) return 0;
	uint32_t* counts = nullptr;
	uint64_t* offsets = nullptr;
	if (hipMalloc((void**)&counts, (size_t)nIds * 4) != hipSuccess || hipMalloc((void**)&offsets, ((size_t)nIds + 1) * 8) != hipSuccess) { if (counts) (void)hipFree(counts); return ~0ull; }
	const uint32_t blocks = (nIds + 63) / 64;
	hipLaunchKernelGGL(k_minimizer_scan, dim3(blocks), dim3(64), 0, nullptr, g, idOrderDev, nIds, k, w, (const uint64_t*)nullptr, counts, (uint64_t*)nullptr, (uint64_t*)nullptr);
	// exclusive scan of the counts into 64-bit offsets
	void* tmp = nullptr; size_t tmpBytes = 0;
	auto scanOp = [&]() { return hipcub::DeviceScan::ExclusiveSum(tmp, tmpBytes, hipcub::TransformInputIterator<uint64_t, hipcub::CastOp<uint64_t>, const uint32_t*>(counts, hipcub::CastOp<uint64_t>()), offsets, (int)nIds); };
	uint64_t total = ~0ull;
	uint64_t *keys = nullptr, *values = nullptr, *keysAlt = nullptr, *valuesAlt = nullptr;
	void* sortTmp = nullptr;
	do {
		if (scanOp() != hipSuccess || hipMalloc(&tmp, tmpBytes ? tmpBytes : 16) != hipSuccess || scanOp() != hipSuccess) break;
		uint64_t lastOff = 0; uint32_t lastCount = 0;
		if (hipMemcpy(&lastOff, offsets + (nIds - 1), 8, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(&lastCount, counts + (nIds - 1), 4, hipMemcpyDeviceToHost) != hipSuccess) break;
		const uint64_t n = lastOff + lastCount;
		if (n >= (1ull << 34) || n > 0x7fffffffull) break;   // arrival indices are packed into 34 key bits; hipCUB counts in int
		if (n == 0) { total = 0; break; }
		if (hipMalloc((void**)&keys, n * 8) != hipSuccess || hipMalloc((void**)&values, n * 8) != hipSuccess || hipMalloc((void**)&keysAlt, n * 8) != hipSuccess || hipMalloc((void**)&valuesAlt, n * 8) != hipSuccess) break;
		hipLaunchKernelGGL(k_minimizer_scan, dim3(blocks), dim3(64), 0, nullptr, g, idOrderDev, nIds, k, w, (const uint64_t*)offsets, counts, keys, values);
		hipcub::DoubleBuffer<uint64_t> dk(keys, keysAlt), dv(values, valuesAlt);
		size_t sortBytes = 0;
		if (hipcub::DeviceRadixSort::SortPairs(nullptr, sortBytes, dk, dv, (int)n, 0, 64) != hipSuccess || hipMalloc(&sortTmp, sortBytes ? sortBytes : 16) != hipSuccess) break;
		if (hipcub::DeviceRadixSort::SortPairs(sortTmp, sortBytes, dk, dv, (int)n, 0, 64) != hipSuccess || hipDeviceSynchronize() != hipSuccess) break;
		*outKeys = dk.Current(); *outValues = dv.Current();
		if (dk.Current() == keys) keys = nullptr; else keysAlt = nullptr;
		if (dv.Current() == values) values = nullptr; else valuesAlt = nullptr;
		total = n;
	} while (false);
	for (void* p : { (void*)counts, (void*)offsets, tmp, sortTmp, (void*)keys, (void*)values, (void*)keysAlt, (void*)valuesAlt }) if (p) (void)hipFree(p);
	return total;
}

} // namespace gcdev
