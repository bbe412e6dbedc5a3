// Shared by frag_host_test.cpp and frag_wave_sim.cpp: the fragment extension core compiled for the host, a flat copy of the oracle's graph with the kernel's per-node records,
// and a lane's working memory in the kernel's word layout. Test infrastructure.
#pragma once
#define __HIP_PLATFORM_AMD__ 1
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __clzll(long long x) { return __builtin_clzll(x); }
#include "../../oracle/pipeline.hpp"
#include "../../graphchainer_amd/csrc/hip/gc_frag_core.hpp"
#include "../../graphchainer_amd/csrc/host/gc_correctness.hpp"
#include <cstdio>
#include <fstream>

using namespace oracle;

struct FlatGraph {
	std::vector<uint8_t> nodeLength;
	std::vector<uint64_t> nodeSeq, ambSeq;
	std::vector<uint32_t> inOff, inAdj, outOff, outAdj, componentNumber;
	std::vector<gcdev::NodeRec> rec;
	gcdev::DGraph d {};
	explicit FlatGraph(const AlignmentGraph& h)
	{
		const size_t n = h.NodeSize();
		nodeLength.resize(n); componentNumber.resize(n);
		for (size_t i = 0; i < n; i++) { nodeLength[i] = (uint8_t)h.nodeLength[i]; componentNumber[i] = (uint32_t)h.componentNumber[i]; }
		nodeSeq.resize(2 * h.firstAmbiguous + 2); ambSeq.resize(4 * (n - h.firstAmbiguous) + 4);
		for (size_t i = 0; i < h.firstAmbiguous; i++) { nodeSeq[2 * i] = h.nodeSequences[i][0]; nodeSeq[2 * i + 1] = h.nodeSequences[i][1]; }
		auto csr = [&](const std::vector<std::vector<size_t>>& adj, std::vector<uint32_t>& off, std::vector<uint32_t>& flat) {
			off.assign(n + 1, 0);
			for (size_t i = 0; i < n; i++) off[i + 1] = off[i] + (uint32_t)adj[i].size();
			for (size_t i = 0; i < n; i++) for (size_t v : adj[i]) flat.push_back((uint32_t)v);
		};
		csr(h.inNeighbors, inOff, inAdj);
		csr(h.outNeighbors, outOff, outAdj);
		rec.resize(n);
		for (size_t i = 0; i < n; i++) {   // as uploadGraph builds them (gc_runtime.hpp)
			gcdev::NodeRec& r = rec[i];
			const uint32_t outDeg = outOff[i + 1] - outOff[i], inDeg = inOff[i + 1] - inOff[i];
			r.comp = componentNumber[i]; r.outOff = outOff[i]; r.inOff = inOff[i];
			r.meta = (uint32_t)nodeLength[i] | (i >= h.firstAmbiguous ? gcdev::NODEREC_SLOW : 0u) | (std::min(outDeg, 255u) << 8) | (std::min(inDeg, 255u) << 16);
			r.w0 = i < h.firstAmbiguous ? nodeSeq[2 * i] : 0; r.w1 = i < h.firstAmbiguous ? nodeSeq[2 * i + 1] : 0;
		}
		d.nNodes = (uint32_t)n; d.firstAmbiguous = (uint32_t)h.firstAmbiguous;
		d.nodeLength = nodeLength.data(); d.nodeSeq = nodeSeq.data(); d.ambSeq = ambSeq.data();
		d.inOff = inOff.data(); d.inAdj = inAdj.data(); d.outOff = outOff.data(); d.outAdj = outAdj.data();
		d.componentNumber = componentNumber.data();
		d.nodeRec = rec.data();
	}
};

struct HostStore {
	uint32_t words[gcfrag::FRAG_WORDS];
	struct Item { uint64_t sVP, sVN, eVP, eVN; int32_t sScore, eScore; uint32_t node; } items[gcfrag::FRAG_I];
	std::vector<gcdev::TraceCell> trace;
	uint32_t ld(uint32_t w) const { if (w >= gcfrag::FRAG_WORDS) { fprintf(stderr, "lane word %u out of range\n", w); exit(2); } return words[w]; }
	void st(uint32_t w, uint32_t v) { if (w >= gcfrag::FRAG_WORDS) { fprintf(stderr, "lane word %u out of range\n", w); exit(2); } words[w] = v; }
	void itemSetStart(uint32_t k, uint64_t VP, uint64_t VN, int32_t score, uint32_t node) { items[k].sVP = VP; items[k].sVN = VN; items[k].sScore = score; items[k].node = node; }
	void itemSetEnd(uint32_t k, uint64_t VP, uint64_t VN, int32_t score) { items[k].eVP = VP; items[k].eVN = VN; items[k].eScore = score; }
	gcdev::WS itemStart(uint32_t k) const { return gcdev::WS { items[k].sVP, items[k].sVN, items[k].sScore }; }
	gcdev::WS itemEnd(uint32_t k) const { return gcdev::WS { items[k].eVP, items[k].eVN, items[k].eScore }; }
	uint32_t itemNode(uint32_t k) const { return items[k].node; }
	void traceSet(uint64_t at, uint32_t node, int32_t seqPos, uint32_t offsetAndSwitch) { if (at >= trace.size()) { fprintf(stderr, "trace cell %llu beyond its reservation\n", (unsigned long long)at); exit(2); } trace[at] = gcdev::TraceCell { node, seqPos, offsetAndSwitch }; }
	void poison() { for (auto& w : words) w = 0xdeadbeefu; }
};


static inline void hostIupacTable(uint8_t* iupac)
{
	for (int i = 0; i < 256; i++) iupac[i] = 0;
	auto set = [&](const char* chars, uint8_t mask) { for (const char* c = chars; *c; c++) iupac[(uint8_t)*c] = mask; };
	set("Aa", 1); set("Cc", 2); set("Gg", 4); set("TtUu", 8);
	set("Rr", 1 | 4); set("Yy", 2 | 8); set("Kk", 4 | 8); set("Mm", 1 | 2); set("Ss", 2 | 4); set("Ww", 1 | 8);
	set("Bb", 2 | 4 | 8); set("Dd", 1 | 4 | 8); set("Hh", 1 | 2 | 8); set("Vv", 1 | 2 | 4); set("Nn", 15);
}
