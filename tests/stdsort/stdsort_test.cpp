#include "../../graphchainer_amd/csrc/hip/gc_stdsort.hpp"
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>
struct E { uint32_t key, id; };
int main() {
	std::mt19937 rng(5);
	long checked = 0;
	for (int iter = 0; iter < 20000; iter++) {
		int n = iter < 200 ? iter : (int)(rng() % 3000);
		int mode = rng() % 6;
		std::vector<E> v(n);
		for (int i = 0; i < n; i++) {
			uint32_t k;
			switch (mode) { case 0: k = 1; break; case 1: k = rng() % 3; break; case 2: k = rng() % 50; break; case 3: k = rng(); break; case 4: k = i / 7; break; default: k = (uint32_t)(n - i) / 3; }
			v[i] = { k, (uint32_t)i };
		}
		if (mode == 5 && n > 100 && (iter & 1)) { /* adversarial-ish: organ pipe */ for (int i = 0; i < n; i++) v[i].key = (uint32_t)std::min(i, n - i); }
		std::vector<E> a = v, b = v;
		std::sort(a.begin(), a.end(), [](const E& l, const E& r) { return l.key < r.key; });
		E* p = b.data();
		gcsort::gcStdSort<E>(p, (long)n, [](const E& l, const E& r) { return l.key < r.key; });
		for (int i = 0; i < n; i++) if (a[i].id != b[i].id) { printf("MISMATCH iter %d n %d mode %d at %d\n", iter, n, mode, i); return 1; }
		// the same sort as independent steps (partition by stop lists, ranges in any order, leaf-wise insertion): what the wave-cooperative device sort runs
		std::vector<E> c = v;
		std::vector<long> work(2 * n + 3 * (n / 16 + 2) + 8);
		E* q = c.data();
		gcsort::gcStdSortBySteps<E>(q, (long)n, [](const E& l, const E& r) { return l.key < r.key; }, work.data());
		for (int i = 0; i < n; i++) if (a[i].id != c[i].id) { printf("MISMATCH (by steps) iter %d n %d mode %d at %d\n", iter, n, mode, i); return 1; }
		checked += n;
	}
	// a depth-limit case: median-of-three killer sequences are hard to make by hand; force the heapsort path through a tiny depth by sorting
	// a big array of the pattern that degrades median-of-3 (sawtooth)
	for (int n : { 5000, 20000, 100000 }) {
		std::vector<E> v(n);
		for (int i = 0; i < n; i++) v[i] = { (uint32_t)((i % 2) ? i : n - i), (uint32_t)i };
		std::vector<E> a = v, b = v;
		std::sort(a.begin(), a.end(), [](const E& l, const E& r) { return l.key < r.key; });
		E* p = b.data();
		gcsort::gcStdSort<E>(p, (long)n, [](const E& l, const E& r) { return l.key < r.key; });
		for (int i = 0; i < n; i++) if (a[i].id != b[i].id) { printf("MISMATCH big n %d at %d\n", n, i); return 1; }
		std::vector<E> c = v;
		std::vector<long> work(2 * n + 3 * (n / 16 + 2) + 8);
		E* q = c.data();
		gcsort::gcStdSortBySteps<E>(q, (long)n, [](const E& l, const E& r) { return l.key < r.key; }, work.data());
		for (int i = 0; i < n; i++) if (a[i].id != c[i].id) { printf("MISMATCH (by steps) big n %d at %d\n", n, i); return 1; }
	}
	for (int iter = 0; iter < 3000; iter++) {   // the heapsort path: libstdc++'s own internals with a small depth limit
		int n = 17 + (int)(rng() % 2000);
		long depth = rng() % 4;
		std::vector<E> v(n);
		for (int i = 0; i < n; i++) v[i] = { (uint32_t)(rng() % (iter % 2 ? 5 : 100000)), (uint32_t)i };
		std::vector<E> a = v, b = v;
		auto cmp = [](const E& l, const E& r) { return l.key < r.key; };
		std::__introsort_loop(a.begin(), a.end(), depth, __gnu_cxx::__ops::__iter_comp_iter(cmp));
		std::__final_insertion_sort(a.begin(), a.end(), __gnu_cxx::__ops::__iter_comp_iter(cmp));
		E* p = b.data();
		gcsort::gcStdSort<E>(p, (long)n, cmp, depth);
		for (int i = 0; i < n; i++) if (a[i].id != b[i].id) { printf("MISMATCH heap iter %d n %d depth %ld at %d\n", iter, n, depth, i); return 1; }
		std::vector<E> c = v;
		std::vector<long> work(2 * n + 3 * (n / 16 + 2) + 8);
		E* q = c.data();
		gcsort::gcStdSortBySteps<E>(q, (long)n, cmp, work.data(), depth);
		for (int i = 0; i < n; i++) if (a[i].id != c[i].id) { printf("MISMATCH (by steps) heap iter %d n %d depth %ld at %d\n", iter, n, depth, i); return 1; }
	}
	printf("STDSORT_OK %ld elements\n", checked);
	return 0;
}
