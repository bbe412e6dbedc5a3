// libstdc++'s std::sort, restated so that a GPU lane can run it: the reference feeds three UNSTABLE std::sort calls into order-sensitive logic
// (matches by count src/MinimizerSeeder.cpp:497, seeds by goodness src/GraphAligner.h:293, seeds by seqPos src/Aligner.cpp:667), so the
// permutation std::sort happens to produce for equal keys is part of the reference's behaviour. That permutation is a deterministic function
// of the comparison outcomes: introsort (median of three of first+1 / middle / last-1 moved to first, unguarded Hoare partition, recursion
// on the right part, depth limit 2 * floor(log2 n) with heapsort beyond it, threshold 16) followed by one insertion sort pass
// (bits/stl_algo.h: __sort, __introsort_loop, __unguarded_partition_pivot, __move_median_to_first, __unguarded_partition,
// __final_insertion_sort, __insertion_sort, __unguarded_linear_insert; bits/stl_heap.h: __make_heap, __pop_heap, __adjust_heap, __push_heap).
// The same algorithm has shipped in every GCC since 4.x. tests/test_host_logic.py::test_stdsort_clone_equals_libstdcxx compiles this header with
// g++ and checks it, permutation for permutation, against the local libstdc++ on keys with many ties.
//
// gcStdSort(a, n, less): sorts the n elements a[0..n) (any trivially copyable T; `a` is anything indexable: a pointer, an LDS view).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GC_SORT_FN __device__ __host__ inline
#else
#define GC_SORT_FN inline
#endif

namespace gcsort {

template <class T, class A, class Less>
GC_SORT_FN void adjustHeap(A& a, long first, long holeIndex, long len, T value, Less less)
{
	const long topIndex = holeIndex;
	long secondChild = holeIndex;
	while (secondChild < (len - 1) / 2) {
		secondChild = 2 * (secondChild + 1);
		if (less(a[first + secondChild], a[first + (secondChild - 1)])) secondChild--;
		a[first + holeIndex] = a[first + secondChild];
		holeIndex = secondChild;
	}
	if ((len & 1) == 0 && secondChild == (len - 2) / 2) {
		secondChild = 2 * (secondChild + 1);
		a[first + holeIndex] = a[first + (secondChild - 1)];
		holeIndex = secondChild - 1;
	}
	// __push_heap
	long parent = (holeIndex - 1) / 2;
	while (holeIndex > topIndex && less(a[first + parent], value)) {
		a[first + holeIndex] = a[first + parent];
		holeIndex = parent;
		parent = (holeIndex - 1) / 2;
	}
	a[first + holeIndex] = value;
}

// __partial_sort(first, last, last): make_heap + sort_heap
template <class T, class A, class Less>
GC_SORT_FN void heapSort(A& a, long first, long last, Less less)
{
	const long len = last - first;
	if (len >= 2) {
		long parent = (len - 2) / 2;
		while (true) {
			T value = a[first + parent];
			adjustHeap<T>(a, first, parent, len, value, less);
			if (parent == 0) break;
			parent--;
		}
	}
	while (last - first > 1) {
		--last;
		T value = a[last];
		a[last] = a[first];
		adjustHeap<T>(a, first, 0, last - first, value, less);
	}
}

template <class T, class A, class Less>
GC_SORT_FN void unguardedLinearInsert(A& a, long last, Less less)
{
	T val = a[last];
	long next = last - 1;
	while (less(val, a[next])) { a[last] = a[next]; last = next; --next; }
	a[last] = val;
}

template <class T, class A, class Less>
GC_SORT_FN void insertionSort(A& a, long first, long last, Less less)
{
	if (first == last) return;
	for (long i = first + 1; i != last; ++i) {
		if (less(a[i], a[first])) {
			T val = a[i];
			for (long k = i; k > first; k--) a[k] = a[k - 1];   // move_backward(first, i, i + 1)
			a[first] = val;
		} else unguardedLinearInsert<T>(a, i, less);
	}
}

template <class T, class A, class Less>
GC_SORT_FN void gcStdSort(A& a, long n, Less less, long depthLimit = -1)   // depthLimit: test hook (the heapsort path), -1 = the reference's 2 * floor(log2 n)
{
	if (n <= 0) return;
	// __introsort_loop with the recursion on the right part kept on an explicit stack (its depth is bounded by the depth limit)
	long stackFirst[64], stackLast[64], stackDepth[64];
	int sp = 0;
	long depth0 = 0;
	for (long m = n; m > 1; m >>= 1) depth0++;   // __lg(n)
	depth0 *= 2;
	if (depthLimit >= 0) depth0 = depthLimit;
	stackFirst[0] = 0; stackLast[0] = n; stackDepth[0] = depth0; sp = 1;
	while (sp > 0) {
		sp--;
		long first = stackFirst[sp], last = stackLast[sp], depth = stackDepth[sp];
		while (last - first > 16) {
			if (depth == 0) { heapSort<T>(a, first, last, less); break; }
			--depth;
			// __unguarded_partition_pivot
			const long mid = first + (last - first) / 2;
			{
				const long ia = first + 1, ib = mid, ic = last - 1;
				long pick;
				if (less(a[ia], a[ib])) {
					if (less(a[ib], a[ic])) pick = ib;
					else if (less(a[ia], a[ic])) pick = ic;
					else pick = ia;
				} else if (less(a[ia], a[ic])) pick = ia;
				else if (less(a[ib], a[ic])) pick = ic;
				else pick = ib;
				T t = a[first]; a[first] = a[pick]; a[pick] = t;
			}
			long lo = first + 1, hi = last;
			while (true) {
				while (less(a[lo], a[first])) ++lo;
				--hi;
				while (less(a[first], a[hi])) --hi;
				if (!(lo < hi)) break;
				T t = a[lo]; a[lo] = a[hi]; a[hi] = t;
				++lo;
			}
			const long cut = lo;
			// the reference recurses into [cut, last) FIRST and then continues with [first, cut): the two ranges are disjoint, so the order in
			// which they are finished does not change either of them; the right part goes on the stack
			stackFirst[sp] = cut; stackLast[sp] = last; stackDepth[sp] = depth; sp++;
			last = cut;
		}
	}
	// __final_insertion_sort
	if (n > 16) {
		insertionSort<T>(a, 0, 16, less);
		for (long i = 16; i != n; ++i) unguardedLinearInsert<T>(a, i, less);
	} else insertionSort<T>(a, 0, n, less);
}

// ---- the same sort as a set of INDEPENDENT steps (r5: what gc_stdsort_wave.hpp runs on the 64 lanes of a wave; kept here as plain C++ so that tests/stdsort can hold it
// against the local libstdc++ permutation for permutation) --------------------------------------------------------------------------------------------------------
// (1) __unguarded_partition as two lists. The loop  { while (a[lo] < pivot) ++lo; --hi; while (pivot < a[hi]) --hi; if (!(lo < hi)) return lo; swap; ++lo; }  never reads a
//     position it has written: lo only moves right of its own swaps and stays left of hi's, hi the mirror image. So its stops are those of the ORIGINAL array: L = the indices
//     in (first, last) with a[i] >= pivot, ascending; R = the indices in [first, last) with a[i] <= pivot, descending (a[first] is the pivot: the scan from the right stops there at
//     the latest). Stop k of either side is swapped with stop k of the other while L[k] < R[k]; K = the number of such k (the condition is monotone). The scan that then runs on
//     stops either at its own next stop or at the other side's last swapped position (which now holds an element of its kind), so cut = min(L[K], R[K - 1]) (a missing term = infinity).
// (2) The ranges [cut, last) and [first, cut) are disjoint: any order of working them off gives the same array (the reference recurses into the right one and loops on the left).
// (3) __final_insertion_sort over the whole array = an insertion sort of every leaf range (<= 16 elements, where introsort stops) on its own: every element of an earlier range
//     is <= every element of a later one, and the insertion stops at the first element that is not greater.
template <class T, class A, class Less>
GC_SORT_FN long partitionByLists(A& a, long first, long last, Less less, long* listL, long* listR)   // pivot already at a[first]; returns the cut
{
	long nL = 0, nR = 0;
	for (long i = first + 1; i < last; i++) if (!less(a[i], a[first])) listL[nL++] = i;
	for (long i = last - 1; i >= first; i--) if (!less(a[first], a[i])) listR[nR++] = i;
	long K = 0;
	while (K < nL && K < nR && listL[K] < listR[K]) K++;
	for (long k = 0; k < K; k++) { T t = a[listL[k]]; a[listL[k]] = a[listR[k]]; a[listR[k]] = t; }
	long cut = -1;
	if (K < nL) cut = listL[K];
	if (K >= 1 && (cut < 0 || listR[K - 1] < cut)) cut = listR[K - 1];
	return cut;
}

template <class T, class A, class Less>
GC_SORT_FN void gcStdSortBySteps(A& a, long n, Less less, long* work /* 2 n + 3 (n / 16 + 2) words */, long depthLimit = -1)
{
	if (n <= 0) return;
	if (n <= 16) { insertionSort<T>(a, 0, n, less); return; }
	long depth0 = 0;
	for (long m = n; m > 1; m >>= 1) depth0++;
	depth0 *= 2;
	if (depthLimit >= 0) depth0 = depthLimit;
	long* listL = work; long* listR = work + n; long* ranges = work + 2 * n;   // pending ranges, three words each, worked off in ANY order (here: last in, first out)
	long nRanges = 0;
	ranges[0] = 0; ranges[1] = n; ranges[2] = depth0; nRanges = 1;
	while (nRanges > 0) {
		nRanges--;
		const long first = ranges[3 * nRanges], last = ranges[3 * nRanges + 1];
		long depth = ranges[3 * nRanges + 2];
		if (depth == 0) { heapSort<T>(a, first, last, less); continue; }
		--depth;
		const long mid = first + (last - first) / 2;
		{
			const long ia = first + 1, ib = mid, ic = last - 1;
			long pick;
			if (less(a[ia], a[ib])) {
				if (less(a[ib], a[ic])) pick = ib;
				else if (less(a[ia], a[ic])) pick = ic;
				else pick = ia;
			} else if (less(a[ia], a[ic])) pick = ia;
			else if (less(a[ib], a[ic])) pick = ic;
			else pick = ib;
			T t = a[first]; a[first] = a[pick]; a[pick] = t;
		}
		const long cut = partitionByLists<T>(a, first, last, less, listL, listR);
		const long childFirst[2] = { cut, first }, childLast[2] = { last, cut };
		for (int c = 0; c < 2; c++) {
			if (childLast[c] - childFirst[c] > 16) { ranges[3 * nRanges] = childFirst[c]; ranges[3 * nRanges + 1] = childLast[c]; ranges[3 * nRanges + 2] = depth; nRanges++; }
			else insertionSort<T>(a, childFirst[c], childLast[c], less);   // a leaf: its share of the final insertion pass
		}
	}
}

} // namespace gcsort
