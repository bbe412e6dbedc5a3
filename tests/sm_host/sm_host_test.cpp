// Host-side check of the state-machine extension core (graphchainer_amd/csrc/hip/gc_sm_core.hpp): the same phase functions the kernel
// k_long_extend_sm runs per lane are compiled here with g++ and ONE lane is driven on the CPU, extension by extension, against the
// oracle's getReverseTraceFromSeed (oracle/bitvector_aligner.hpp, restating src/GraphAlignerBitvectorBanded.h:46-71): status, score and
// every trace cell must be equal. Test infrastructure: nothing in the library runs these functions on the host.
//
// usage: sm_host_test graph.gfa reads.txt [seedsPerRead] [bandwidth]
#define __HIP_PLATFORM_AMD__ 1
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __clzll(long long x) { return __builtin_clzll(x); }
#include "../../oracle/pipeline.hpp"
#include "../../graphchainer_amd/csrc/hip/gc_sm_core.hpp"
#include "../../graphchainer_amd/csrc/host/gc_correctness.hpp"
#include <cstdio>
#include <fstream>

using namespace oracle;

struct HostTables {
	uint32_t* w;
	uint32_t ld(uint32_t i) const { return w[i]; }
	void st(uint32_t i, uint32_t v) const { w[i] = v; }
};

struct FlatGraph {
	std::vector<uint8_t> nodeLength;
	std::vector<uint64_t> nodeSeq, ambSeq;
	std::vector<uint32_t> inOff, inAdj, outOff, outAdj, componentNumber;
	gcdev::DGraph d {};
	explicit FlatGraph(const AlignmentGraph& h)
	{
		const size_t n = h.NodeSize();
		nodeLength.resize(n); componentNumber.resize(n);
		for (size_t i = 0; i < n; i++) { nodeLength[i] = (uint8_t)h.nodeLength[i]; componentNumber[i] = (uint32_t)h.componentNumber[i]; }
		nodeSeq.resize(2 * h.firstAmbiguous + 2); ambSeq.resize(4 * (n - h.firstAmbiguous) + 4);
		for (size_t i = 0; i < h.firstAmbiguous; i++) { nodeSeq[2 * i] = h.nodeSequences[i][0]; nodeSeq[2 * i + 1] = h.nodeSequences[i][1]; }
		for (size_t i = h.firstAmbiguous; i < n; i++) {
			const gc::AmbiguousSeq& s = h.ambiguousNodeSequences[i - h.firstAmbiguous];
			size_t at = 4 * (i - h.firstAmbiguous);
			ambSeq[at] = s.A; ambSeq[at + 1] = s.C; ambSeq[at + 2] = s.G; ambSeq[at + 3] = s.T;
		}
		auto csr = [&](const std::vector<std::vector<size_t>>& adj, std::vector<uint32_t>& off, std::vector<uint32_t>& flat) {
			off.assign(n + 1, 0);
			for (size_t i = 0; i < n; i++) off[i + 1] = off[i] + (uint32_t)adj[i].size();
			for (size_t i = 0; i < n; i++) for (size_t v : adj[i]) flat.push_back((uint32_t)v);
		};
		csr(h.inNeighbors, inOff, inAdj);
		csr(h.outNeighbors, outOff, outAdj);
		d.nNodes = (uint32_t)n; d.firstAmbiguous = (uint32_t)h.firstAmbiguous;
		d.nodeLength = nodeLength.data(); d.nodeSeq = nodeSeq.data(); d.ambSeq = ambSeq.data();
		d.inOff = inOff.data(); d.inAdj = inAdj.data(); d.outOff = outOff.data(); d.outAdj = outAdj.data();
		d.componentNumber = componentNumber.data();
	}
};

// match-mask bit vectors of a sequence, as k_pack_read_masks builds them: [A,C,G,T][words], bit i set when position i matches the base
static std::vector<uint64_t> matchMasks(std::string_view seq, uint32_t& words)
{
	words = (uint32_t)(seq.size() + 63) / 64 + 1;
	std::vector<uint64_t> m(4ull * words, 0);
	const char bases[4] = { 'A', 'C', 'G', 'T' };
	for (size_t i = 0; i < seq.size(); i++)
		for (int b = 0; b < 4; b++)
			if (characterMatch(seq[i], bases[b])) m[(size_t)b * words + i / 64] |= 1ull << (i % 64);
	return m;
}

struct SmRun { uint32_t status; int32_t score; std::vector<unsigned long long> trace; gcdev::ExtCounters cnt; unsigned long long phases[8]; };

static SmRun runLane(const gcdev::DGraph& g, const gcdev::CorrectnessTables& ct, const gcsm::SmParams& P, std::string_view seq, uint32_t node, uint32_t offset, uint32_t startBitShift)
{
	using namespace gcsm;
	// the sequence sits at bit `startBitShift` of the bit vectors (extensions of a read start anywhere in its masks)
	std::string padded(startBitShift, 'A');
	padded.append(seq);
	uint32_t words = 0;
	std::vector<uint64_t> masks = matchMasks(padded, words);
	std::vector<uint8_t> slab(smSlabBytes(P));
	std::vector<uint32_t> tables(SM_LANE_WORDS, 0xdeadbeefu);
	SmRun out {};
	out.trace.resize(seq.size() + seq.size() / 2 + 512);
	SmLane L {};
	L.len = (int32_t)seq.size();
	L.startNode = node; L.startOffset = offset;
	L.masks = masks.data(); L.maskWords = words; L.startBit = startBitShift;
	L.trace = out.trace.data(); L.traceCap = (uint32_t)out.trace.size();
	L.items = (gcdev::NodeItem*)slab.data();
	L.slices = (SmSlice*)(slab.data() + (size_t)P.maxItems * sizeof(gcdev::NodeItem));
	L.cols = (SmWalkCol*)(slab.data() + (size_t)P.maxItems * sizeof(gcdev::NodeItem) + (size_t)P.maxSlices * sizeof(SmSlice));
	HostTables t { tables.data() };
	smBegin(g, ct, P, L, t);
	while (L.state != SM_RETIRE) {
		out.phases[L.state]++;
		switch (L.state) {
			case SM_B: smPhaseB(g, ct, P, L, t); break;
			case SM_COL: smPhaseCol(g, L); break;
			case SM_BT: smPhaseBt(g, P, L, t); break;
			case SM_WALK: smPhaseWalk(L); break;
			default: fprintf(stderr, "bad state %u\n", L.state); exit(2);
		}
	}
	out.status = L.status; out.score = L.score; out.cnt = L.cnt;
	out.trace.resize(L.status == gcdev::EXT_OK ? L.nTrace : 0);
	return out;
}

int main(int argc, char** argv)
{
	if (argc < 3) { fprintf(stderr, "usage: %s graph.gfa reads.txt [seedsPerRead] [bandwidth]\n", argv[0]); return 2; }
	const int seedsPerRead = argc > 3 ? atoi(argv[3]) : 3;
	const int bandwidth = argc > 4 ? atoi(argv[4]) : 10;
	gc::GfaGraph gfa = gc::GfaGraph::LoadFromFile(argv[1]);
	AlignmentGraph graph = AlignmentGraph::BuildFromGFA(gfa);
	gc::MinimizerIndex index = gc::MinimizerIndex::Build(graph, 15, 20, 1.0 - 0.001);
	FlatGraph flat(graph);
	gcdev::CorrectnessTables ct;
	buildCorrectnessTables(ct);
	BitvectorAligner bv(graph, (size_t)bandwidth);
	AlignerState state(graph);
	std::ifstream in(argv[2]);
	std::string read;
	size_t nExt = 0, nOk = 0, nFailed = 0, nAssert = 0, nDeclined = 0, nCells = 0;
	unsigned long long phases[8] = { 0 };
	gcdev::ExtCounters total {};
	while (std::getline(in, read)) {
		if (read.empty()) continue;
		std::vector<SeedHit> seeds = getSeeds(graph, index, read, 10.0);
		const std::string rev = gc::ReverseComplement(read);
		const size_t step = std::max<size_t>(1, seeds.size() / (size_t)std::max(1, seedsPerRead));
		for (size_t si = 0; si < seeds.size(); si += step) {
			const SeedHit& sd = seeds[si];
			const int forwardNodeId = sd.nodeID * 2 + (sd.reverse ? 1 : 0);
			for (int dir = 0; dir < 2; dir++) {
				std::string_view part;
				int bigraphId; size_t offset; uint32_t shift;
				if (dir == 0) {
					if (sd.seqPos == 0) continue;
					part = std::string_view(rev.data() + rev.size() - sd.seqPos, sd.seqPos);
					auto reversePos = graph.GetReversePosition(forwardNodeId, sd.nodeOffset);
					bigraphId = forwardNodeId ^ 1; offset = reversePos.second;
					shift = (uint32_t)(rev.size() - sd.seqPos) % 64 + 64 * (uint32_t)(si % 3);   // the kernel's startBit is any read position
				} else {
					if (sd.seqPos + 1 >= read.size()) continue;
					part = std::string_view(read.data() + sd.seqPos + 1, read.size() - sd.seqPos - 1);
					bigraphId = forwardNodeId; offset = sd.nodeOffset;
					shift = (uint32_t)(sd.seqPos + 1) % 64;
				}
				const size_t splitNode = graph.GetUnitigNode(bigraphId, offset);
				const size_t splitOffset = offset - graph.nodeOffset[splitNode];
				uint32_t wantStatus = gcdev::EXT_OK;
				OnewayTrace want;
				try { want = bv.getReverseTraceFromSeed(part, bigraphId, offset, state); if (want.failed()) wantStatus = gcdev::EXT_FAILED; }
				catch (const AssertionFailure&) { wantStatus = gcdev::EXT_ASSERT; state.clear(); }
				gcsm::SmParams P { bandwidth, (uint32_t)std::max<size_t>(8192, (part.size() / 64 + 3) * 24), (uint32_t)(part.size() / 64 + 3) };
				SmRun got = runLane(flat.d, ct, P, part, (uint32_t)splitNode, (uint32_t)splitOffset, shift);
				nExt++;
				for (int k = 0; k < 8; k++) phases[k] += got.phases[k];
				if (got.status == gcsm::EXT_SM_DECLINED) { nDeclined++; continue; }   // the layout's tables were too small: the kernel hands such items to the one-extension-per-wave kernel
				auto failHere = [&](const char* what, size_t i) {
					fprintf(stderr, "MISMATCH %s: read seed %zu dir %d len %zu node %zu off %zu: status %u/%u score %d/%d trace %zu/%zu at %zu\n", what, si, dir, part.size(), splitNode, splitOffset,
						got.status, wantStatus, got.score, wantStatus == gcdev::EXT_OK ? want.score : 0, got.trace.size(), want.trace.size(), i);
					exit(1);
				};
				if (got.status != wantStatus) failHere("status", 0);
				if (wantStatus == gcdev::EXT_FAILED) { nFailed++; continue; }
				if (wantStatus == gcdev::EXT_ASSERT) { nAssert++; continue; }
				if (got.score != want.score) failHere("score", 0);
				if (got.trace.size() != want.trace.size()) failHere("trace length", 0);
				for (size_t i = 0; i < want.trace.size(); i++) {
					const TraceItem& t = want.trace[i];
					const unsigned long long cell = gcsm::smPackCell((uint32_t)t.DPposition.node, (uint32_t)t.DPposition.nodeOffset, (int32_t)(int64_t)t.DPposition.seqPos, t.nodeSwitch);
					if (cell != got.trace[i]) failHere("trace cell", i);
				}
				nOk++; nCells += want.trace.size();
				total.dpTiles += got.cnt.dpTiles; total.columnSteps += got.cnt.columnSteps; total.backtraceTiles += got.cnt.backtraceTiles;
			}
		}
	}
	printf("SM_HOST_OK extensions %zu equal %zu failed %zu asserted %zu declined %zu cells %zu | per extension: B %.0f COL %.0f BT %.0f WALK %.0f, dp tiles %.0f\n", nExt, nOk, nFailed, nAssert, nDeclined, nCells,
		(double)phases[gcsm::SM_B] / nExt, (double)phases[gcsm::SM_COL] / nExt, (double)phases[gcsm::SM_BT] / nExt, (double)phases[gcsm::SM_WALK] / nExt, (double)total.dpTiles / std::max<size_t>(1, nOk));
	return 0;
}
