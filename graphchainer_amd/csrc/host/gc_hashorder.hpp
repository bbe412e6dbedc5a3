// The iteration order of a libstdc++ unordered container, without the container.
//
// The reference numbers its split nodes by iterating std::unordered_map<int, std::string> (src/BigraphToDigraph.cpp:229), adds the edges by iterating
// std::unordered_map<NodePos, std::vector<NodePos>> (:251) and enumerates the minimizers' nodes by iterating std::unordered_map<int, std::vector<size_t>>
// (src/MinimizerSeeder.cpp:354-357): the graph the kernels see depends on the order in which libstdc++'s hash table links its nodes. r1-r3 reproduced it by
// building the same containers with the same insertion sequence - 65 M node-based inserts per map for a human-genome graph, serial, the larger part of a
// first build. The order itself is a function of the sequence of hash values and of the rehash policy only:
//   - a new element goes to the front of its bucket's run if the bucket is occupied, else to the front of the whole list (_M_insert_bucket_begin);
//   - a rehash walks the list and re-inserts every node by the same rule into the new bucket array (_M_rehash_aux, unique keys);
//   - bucket = hash % bucket count; the counts are the policy's primes (std::__detail::_Prime_rehash_policy, whose _M_need_rehash is called before every
//     insertion exactly as _Hashtable::_M_insert_unique_node does).
// HashOrder replays that on two integer arrays. It uses the library's own policy object, so the bucket counts are this libstdc++'s by construction;
// tests/hashorder/hashorder_test.cpp compares it with the real containers (the three key types above, growth across many rehashes, erased keys).
#pragma once
#include <cstddef>
#include <cstdint>
#include <unordered_map>   // <bits/hashtable_policy.h>: std::__detail::_Prime_rehash_policy
#include <vector>

namespace gc {

class HashOrder {
public:
	// the hash value of the next NEW key (the caller keeps keys unique, as operator[] / insert do); returns the element's index (0, 1, 2, ...)
	uint32_t insert(size_t hash)
	{
		const uint32_t n = (uint32_t)hashOf.size();
		const std::pair<bool, size_t> grow = policy.need(bucketCount, n, 1);
		if (grow.first) rehash(grow.second);
		hashOf.push_back(hash);
		next.push_back(NIL);
		const size_t b = hash % bucketCount;
		if (before[b] != EMPTY) {
			next[n] = nextOf(before[b]);
			setNext(before[b], n);
		} else {
			next[n] = head;
			head = n;
			if (next[n] != NIL) before[hashOf[next[n]] % bucketCount] = n;
			before[b] = BEFORE_BEGIN;
		}
		return n;
	}
	size_t size() const { return hashOf.size(); }
	// element indices in the container's iteration order (begin() to end())
	std::vector<uint32_t> order() const
	{
		std::vector<uint32_t> out;
		out.reserve(hashOf.size());
		for (uint32_t p = head; p != NIL; p = next[p]) out.push_back(p);
		return out;
	}

private:
	static constexpr uint32_t NIL = 0xffffffffu, EMPTY = 0xfffffffeu, BEFORE_BEGIN = 0xfffffffdu;
	struct Policy : std::__detail::_Prime_rehash_policy {
		std::pair<bool, size_t> need(size_t buckets, size_t elements, size_t inserting) const { return _M_need_rehash(buckets, elements, inserting); }
	};
	Policy policy;
	size_t bucketCount = 1;
	std::vector<uint32_t> before { EMPTY };   // per bucket: the element BEFORE its first element (BEFORE_BEGIN: the list's head sentinel), EMPTY: no element
	std::vector<uint32_t> next;                // per element
	std::vector<size_t> hashOf;
	uint32_t head = NIL;

	uint32_t nextOf(uint32_t p) const { return p == BEFORE_BEGIN ? head : next[p]; }
	void setNext(uint32_t p, uint32_t n) { if (p == BEFORE_BEGIN) head = n; else next[p] = n; }
	void rehash(size_t count)
	{
		std::vector<uint32_t> fresh(count, EMPTY);
		uint32_t p = head;
		head = NIL;
		size_t headBucket = 0;
		while (p != NIL) {
			const uint32_t following = next[p];
			const size_t b = hashOf[p] % count;
			if (fresh[b] == EMPTY) {
				next[p] = head;
				head = p;
				fresh[b] = BEFORE_BEGIN;
				if (next[p] != NIL) fresh[headBucket] = p;
				headBucket = b;
			} else {
				next[p] = nextOf2(fresh[b]);
				setNext(fresh[b], p);
			}
			p = following;
		}
		before.swap(fresh);
		bucketCount = count;
	}
	uint32_t nextOf2(uint32_t p) const { return p == BEFORE_BEGIN ? head : next[p]; }
};

} // namespace gc
