// libstdc++'s std::sort on the 64 lanes of ONE wave, permutation for permutation (r5).
//
// gc_stdsort.hpp restates the sort for one lane; k_seed_glue ran it on lane 0 for the reference's three order-critical unstable sorts (matches by count, seeds by goodness,
// seeds by read position). A serial introsort is bound by the latency of its dependent accesses: 13 ms per 10 kb read beside nothing, 57 ms under five batches in flight, and
// 650 ms for the 18 000 seed occurrences a 50 kb read collects on a 960 Mbp graph (`gpurun_out/r5_cfg5_base`: a third of that workload's kernel time). The permutation
// std::sort produces is a function of its comparison outcomes, and the work decomposes into steps that do not depend on one another (gc_stdsort.hpp, gcStdSortBySteps:
// checked against the local libstdc++ by tests/stdsort):
//   - one partition is two lists of stops taken from the UNCHANGED range (indices with a[i] >= pivot from the left, a[i] <= pivot from the right), a count K of
//     pairs to swap, and a cut - a ballot per 64 elements, a binary search and K independent swaps instead of two pointers walking towards each other;
//   - the two sides of a cut never touch each other again: ranges are worked off in any order, i.e. by different lanes at once;
//   - the final insertion pass never moves an element out of its leaf range (<= 16 elements): a leaf is insertion-sorted by the lane that produced it.
// So: ranges above GC_SORT_COOP_MIN elements are partitioned by all lanes together, one after the other; the others go to a list that the lanes work off one range each,
// level by level (one partition step per range and level, the children go to the next level's list). The critical path of an n-element sort falls from ~14 n dependent
// element visits to ~n / 64 per cooperative level plus a few hundred for the last levels.
//
// gcStdSortWave<T>(a, n, less, scratch, lane): all 64 lanes of a one-wave block call it together. `a`: anything indexable (HBM pointer, LDS view);
// scratch: waveSortScratchWords(n) 32-bit words (HBM or LDS, as a generic pointer), contents undefined on entry and exit.
#pragma once
#include "gc_stdsort.hpp"
#include <hip/hip_runtime.h>

#ifndef GC_SORT_COOP_MIN
#define GC_SORT_COOP_MIN 256   // ranges with more elements are partitioned by the whole wave
#endif

namespace gcsort {

__host__ __device__ inline uint32_t waveSortRangeCap(uint32_t n) { return n / 17 + 4; }                                   // pending ranges hold more than 16 elements each and are disjoint
__host__ __device__ inline uint32_t waveSortScratchWords(uint32_t n) { return 2 * n + 12 * waveSortRangeCap(n) + 8; }       // two stop lists, four range lists of three words, four counters

template <class T, class A, class Less>
__device__ __forceinline__ long medianToFirstPick(A& a, long first, long last, Less less)   // __move_median_to_first's choice among first + 1, middle, last - 1
{
	const long ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
	if (less(a[ia], a[ib])) {
		if (less(a[ib], a[ic])) return ib;
		if (less(a[ia], a[ic])) return ic;
		return ia;
	}
	if (less(a[ia], a[ic])) return ia;
	if (less(a[ib], a[ic])) return ic;
	return ib;
}

template <class T, class A, class Less>
__device__ inline void gcStdSortWave(A a, uint32_t n, Less less, uint32_t* scratch, uint32_t lane, long depthLimit = -1)
{
	if (n <= 16) {
		if (lane == 0 && n > 1) insertionSort<T>(a, 0, (long)n, less);
		__threadfence_block();
		__syncthreads();
		return;
	}
	const uint32_t cap = waveSortRangeCap(n);
	uint32_t* const listL = scratch;
	uint32_t* const listR = scratch + n;            // ascending indices; stop k from the right is listR[nR - 1 - k]
	uint32_t* const ranges = scratch + 2 * n;       // [0]/[1]: cooperative ranges of this / the next level, [2]/[3]: one-lane ranges
	uint32_t* const count = ranges + 12 * cap;
	auto rangeList = [&](uint32_t which) { return ranges + 3 * cap * which; };
	long depth0 = 0;
	for (uint32_t m = n; m > 1; m >>= 1) depth0++;   // __lg(n)
	depth0 *= 2;
	if (depthLimit >= 0) depth0 = depthLimit;
	if (lane == 0) {
		count[0] = count[1] = count[2] = count[3] = 0;
		const uint32_t which = n > GC_SORT_COOP_MIN ? 0u : 2u;
		uint32_t* r = rangeList(which);
		r[0] = 0; r[1] = n; r[2] = (uint32_t)depth0;
		count[which] = 1;
	}
	__threadfence_block();
	__syncthreads();
	// a child range goes to the next level's cooperative list, to its one-lane list, or - a leaf - is insertion-sorted on the spot: its share of __final_insertion_sort
	auto emit = [&](uint32_t nxt, long first, long last, long depth) {
		const long size = last - first;
		if (size <= 16) { if (size > 1) insertionSort<T>(a, first, last, less); return; }
		const uint32_t which = (size > GC_SORT_COOP_MIN ? 0u : 2u) + nxt;
		uint32_t* r = rangeList(which) + 3u * atomicAdd(&count[which], 1u);
		r[0] = (uint32_t)first; r[1] = (uint32_t)last; r[2] = (uint32_t)depth;
	};
	uint32_t cur = 0;
	while (true) {
		const uint32_t nBig = count[cur], nSmall = count[2 + cur], nxt = cur ^ 1u;
		if (nBig == 0 && nSmall == 0) break;
		// ---- ranges the whole wave partitions, one after the other (every value below is the same in all lanes unless it says `lane`)
		for (uint32_t b = 0; b < nBig; b++) {
			const uint32_t* rg = rangeList(cur) + 3 * b;
			const long first = rg[0], last = rg[1];
			long depth = rg[2];
			if (depth == 0) {   // the depth limit: __partial_sort(first, last, last), a heapsort - serial, and rare (2 floor(log2 n) levels of bad pivots)
				if (lane == 0) heapSort<T>(a, first, last, less);
				__threadfence_block();
				__syncthreads();
				continue;
			}
			--depth;
			const long pick = medianToFirstPick<T>(a, first, last, less);
			const T pivot = a[pick];
			__syncthreads();
			if (lane == 0) { const T t = a[first]; a[first] = pivot; a[pick] = t; }
			__threadfence_block();
			__syncthreads();
			// the stops of the two scans of __unguarded_partition, from the range as it is now: 64 elements per step, a ballot each
			uint32_t nL = 0, nR = 0;
			for (long base = first; base < last; base += 64) {
				const long i = base + lane;
				const bool in = i < last;
				T v = pivot;
				if (in) v = a[i];
				const bool ge = in && i > first && !less(v, pivot);   // where `while (a[lo] < pivot) ++lo` stops: lo starts at first + 1
				const bool le = in && !less(pivot, v);                // where `while (pivot < a[hi]) --hi` stops: a[first], the pivot itself, at the latest
				const unsigned long long bl = __ballot(ge), br = __ballot(le);
				const unsigned long long below = (1ull << lane) - 1ull;
				if (ge) listL[nL + (uint32_t)__popcll(bl & below)] = (uint32_t)i;
				if (le) listR[nR + (uint32_t)__popcll(br & below)] = (uint32_t)i;
				nL += (uint32_t)__popcll(bl);
				nR += (uint32_t)__popcll(br);
			}
			__threadfence_block();
			__syncthreads();
			// K = the pairs (stop k from the left, stop k from the right) with the left one still left of the right one: they are swapped (the condition is monotone in k)
			uint32_t lo = 0, hi = nL < nR ? nL : nR;
			while (lo < hi) {
				const uint32_t mid = (lo + hi) >> 1;
				if (listL[mid] < listR[nR - 1 - mid]) lo = mid + 1; else hi = mid;
			}
			const uint32_t K = lo;
			for (uint32_t k = lane; k < K; k += 64) {
				const uint32_t x = listL[k], y = listR[nR - 1 - k];
				const T t = a[x]; a[x] = a[y]; a[y] = t;
			}
			// the scan from the left that follows the last swap stops at its own next stop, or at the right side's last swapped position (it holds a >= element now)
			uint32_t cut = 0xffffffffu;
			if (K < nL) cut = listL[K];
			if (K >= 1) { const uint32_t y = listR[nR - K]; cut = y < cut ? y : cut; }
			__threadfence_block();
			__syncthreads();
			if (lane == 0) { emit(nxt, (long)cut, last, depth); emit(nxt, first, (long)cut, depth); }
			__threadfence_block();
			__syncthreads();
		}
		// ---- ranges of one lane each: one step of __introsort_loop per range and level, the plain two-pointer partition
		for (uint32_t s = lane; s < nSmall; s += 64) {
			const uint32_t* rg = rangeList(2 + cur) + 3 * s;
			const long first = rg[0], last = rg[1];
			long depth = rg[2];
			if (depth == 0) { heapSort<T>(a, first, last, less); continue; }
			--depth;
			const long pick = medianToFirstPick<T>(a, first, last, less);
			{ const T t = a[first]; a[first] = a[pick]; a[pick] = t; }
			const T pivot = a[first];
			long l = first + 1, h = last;
			while (true) {
				while (less(a[l], pivot)) ++l;
				--h;
				while (less(pivot, a[h])) --h;
				if (!(l < h)) break;
				const T t = a[l]; a[l] = a[h]; a[h] = t;
				++l;
			}
			emit(nxt, l, last, depth);
			emit(nxt, first, l, depth);
		}
		__threadfence_block();
		__syncthreads();
		if (lane == 0) { count[cur] = 0; count[2 + cur] = 0; }
		__threadfence_block();
		__syncthreads();
		cur = nxt;
	}
}

} // namespace gcsort
