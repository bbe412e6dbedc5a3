// The permutation libstdc++'s std::sort gives an array when only a key is compared (`left.key < right.key`): what the reference's three order-critical unstable
// sorts do (src/MinimizerSeeder.cpp:497, src/GraphAligner.h:293, src/Aligner.cpp:667). Built by tests/test_seeding_model.py with the local g++ as a shared library
// (ctypes); nothing of the product or of the oracle is compiled in - this is the real std::sort.
#include <algorithm>
#include <cstdint>
#include <vector>
extern "C" void std_sort_perm(const uint64_t* keys, int64_t n, int64_t* perm)
{
	struct E { uint64_t key; int64_t id; };
	std::vector<E> v((size_t)n);
	for (int64_t i = 0; i < n; i++) v[(size_t)i] = E { keys[i], i };
	std::sort(v.begin(), v.end(), [](const E& l, const E& r) { return l.key < r.key; });
	for (int64_t i = 0; i < n; i++) perm[i] = v[(size_t)i].id;
}
