// Host-side check of the fragment extension core (graphchainer_amd/csrc/hip/gc_frag_core.hpp): the phase functions the kernel k_extend runs per lane are
// compiled here with g++ and ONE lane is driven on the CPU, extension by extension, against the oracle's getReverseTraceFromSeed
// (oracle/bitvector_aligner.hpp, restating src/GraphAlignerBitvectorBanded.h:46-71): status, score, every trace cell, the tie flag and the work counters must be
// equal. The lane's working memory is the kernel's word layout (queue and ring + ids overlaid), poisoned at every hand-over between the two parts.
// Test infrastructure: nothing in the library runs these functions on the host.
//
// usage: frag_host_test graph.gfa reads.txt [seedsPerRead] [bandwidth]
#include "frag_host_common.hpp"

struct Run { uint32_t status; int32_t score; uint32_t tie; std::vector<gcdev::TraceCell> trace; uint32_t dpTiles, dpCols, btTiles, btCols, maxPending, nItems, traceCap; unsigned long long turns[8]; };

static Run runLane(const gcdev::DGraph& g, const gcfrag::FragParams& P, const uint8_t* iupac, std::string_view seq, uint32_t node, uint32_t offset)
{
	using namespace gcfrag;
	Run out {};
	FragMem<HostStore> m;
	m.poison();
	Lane L {};
	L.phase = PH_FETCH;
	fragBegin(g, P, L, m, 0, (uint32_t)seq.size(), node, offset, gcdev::EqFromBases { seq.data(), iupac });
	while (L.phase != PH_FETCH) {
		out.turns[L.phase]++;
		if (L.phase == PH_COLS && (L.tileFlags & TF_WALK)) out.turns[7]++;
		out.maxPending = std::max(out.maxPending, L.nPending);
		switch (L.phase) {
			case PH_POP: fragPop(g, P, L, m); break;
			case PH_COLS: fragColumn(L, m); break;
			case PH_TILE_END: fragTileEnd(g, P, L, m); break;
			case PH_FINISH:
				if (fragFinish(P, L)) {
					m.trace.assign(L.traceCap, gcdev::TraceCell { 0xffffffffu, -7, 0xffffffffu });
					L.traceBase = 0;
					m.poison();   // the queue is dead: the walk's ring and ids take its words
					fragWalkBegin(L, m);
				}
				break;
			case PH_WALK: fragWalkStep(g, P, L, m); break;
			default: fprintf(stderr, "bad phase %u\n", L.phase); exit(2);
		}
	}
	out.status = L.status; out.score = L.resultScore; out.tie = L.status == gcdev::EXT_OK ? L.tie : 0;
	out.dpTiles = L.cntTiles & 0xffffu; out.dpCols = L.cntCols & 0xffffu; out.btTiles = L.cntTiles >> 16; out.btCols = L.cntCols >> 16; out.nItems = L.nItems; out.traceCap = L.traceCap;
	if (L.status == gcdev::EXT_OK) out.trace.assign(m.trace.begin(), m.trace.begin() + L.nTrace);
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
	gcfrag::FragParams P { bandwidth, 0 };
	for (int i = 0; i < 64; i++) if (gcfrag::fragSliceKept(ct, i)) P.keepMask |= 1ull << i;
	uint8_t iupac[256];
	hostIupacTable(iupac);
	BitvectorAligner bv(graph, (size_t)bandwidth);
	AlignerState state(graph);
	std::ifstream in(argv[2]);
	std::string read;
	size_t nExt = 0, nOk = 0, nFailed = 0, nAssert = 0, nDeclined = 0, nCells = 0, nTies = 0, capCells = 0;
	unsigned long long turns[8] = { 0 }, pendingHist[16] = { 0 }, itemHist[32] = { 0 };
	size_t readNo = 0; long maxExcess = -1000; unsigned long long fragItemHist[32] = { 0 };
	while (std::getline(in, read)) {
		if (read.empty()) continue;
		readNo++;
		std::vector<SeedHit> seeds = getSeeds(graph, index, read, 10.0);
		const std::string rev = gc::ReverseComplement(read);
		const size_t step = std::max<size_t>(1, seeds.size() / (size_t)std::max(1, seedsPerRead));
		for (size_t si = 0; si < seeds.size(); si += step) {
			const SeedHit& sd = seeds[si];
			const int forwardNodeId = sd.nodeID * 2 + (sd.reverse ? 1 : 0);
			for (int dir = 0; dir < 2; dir++) {
				std::string_view part;
				int bigraphId; size_t offset;
				// lengths as the fragment pass makes them (0..34 beside a seed inside a 35-base window) and everything else the core must answer: 1..64 rows, and beyond
				const size_t k = nExt % 16;
				const size_t wantLen = k < 8 ? 1 + (nExt * 7) % 34 : k < 12 ? 35 + (nExt * 5) % 30 : k == 12 ? 64 : k == 13 ? 63 : k == 14 ? 65 + nExt % 40 : 1 + nExt % 3;
				if (dir == 0) {
					if (sd.seqPos == 0) continue;
					const size_t len = std::min<size_t>(wantLen, sd.seqPos);
					part = std::string_view(rev.data() + rev.size() - sd.seqPos, len);
					auto reversePos = graph.GetReversePosition(forwardNodeId, sd.nodeOffset);
					bigraphId = forwardNodeId ^ 1; offset = reversePos.second;
				} else {
					if (sd.seqPos + 1 >= read.size()) continue;
					const size_t len = std::min<size_t>(wantLen, read.size() - sd.seqPos - 1);
					part = std::string_view(read.data() + sd.seqPos + 1, len);
					bigraphId = forwardNodeId; offset = sd.nodeOffset;
				}
				const size_t splitNode = graph.GetUnitigNode(bigraphId, offset);
				const size_t splitOffset = offset - graph.nodeOffset[splitNode];
				uint32_t wantStatus = gcdev::EXT_OK;
				OnewayTrace want;
				const AlignerCounters before = state.counters;
				try { want = bv.getReverseTraceFromSeed(part, bigraphId, offset, state); if (want.failed()) wantStatus = gcdev::EXT_FAILED; }
				catch (const AssertionFailure&) { wantStatus = gcdev::EXT_ASSERT; state.clear(); }
				Run got = runLane(flat.d, P, iupac, part, (uint32_t)splitNode, (uint32_t)splitOffset);
				nExt++;
				for (int q = 0; q < 8; q++) turns[q] += got.turns[q];
				pendingHist[std::min<uint32_t>(got.maxPending, 15)]++;
				itemHist[std::min<uint32_t>(got.nItems, 31)]++;
				auto failHere = [&](const char* what, size_t i) {
					fprintf(stderr, "MISMATCH %s: read %zu seed %zu dir %d len %zu node %zu off %zu: status %u/%u score %d/%d trace %zu/%zu at %zu\n", what, readNo, si, dir, part.size(), splitNode, splitOffset,
						got.status, wantStatus, got.score, wantStatus == gcdev::EXT_OK ? want.score : 0, got.trace.size(), want.trace.size(), i);
					exit(1);
				};
				if (got.status == gcdev::EXT_OVERFLOW) {   // handed to the plain-layout core: allowed for what the core declares out of its range
					if (part.size() <= 64 && wantStatus == gcdev::EXT_OK && got.nItems < gcfrag::FRAG_I && got.maxPending < gcfrag::FRAG_Q && want.trace.size() <= part.size() + (size_t)want.score + 2) failHere("declined without a reason", 0);
					nDeclined++;
					continue;
				}
				if (part.size() > 64) failHere("more than one slice not declined", 0);
				if (got.status != wantStatus) failHere("status", 0);
				if (wantStatus == gcdev::EXT_FAILED) { nFailed++; continue; }
				if (wantStatus == gcdev::EXT_ASSERT) { nAssert++; continue; }
				if (got.score != want.score) failHere("score", 0);
				if (got.trace.size() != want.trace.size()) failHere("trace length", 0);
				for (size_t i = 0; i < want.trace.size(); i++) {
					const TraceItem& t = want.trace[i];
					const gcdev::TraceCell& c = got.trace[i];
					if (c.node != (uint32_t)t.DPposition.node || (c.offsetAndSwitch & 255u) != (uint32_t)t.DPposition.nodeOffset || c.seqPos != (int32_t)(int64_t)t.DPposition.seqPos || ((c.offsetAndSwitch >> 8) & 1u) != (t.nodeSwitch ? 1u : 0u)) failHere("trace cell", i);
				}
				const AlignerCounters& after = state.counters;
				if (got.tie != (uint32_t)(after.flattenTies - before.flattenTies)) failHere("tie flag", 0);
				const bool flat = part.size() < 64;
				if (got.dpTiles != after.dpTiles - before.dpTiles) failHere("dp tiles", got.dpTiles);
				if ((flat ? got.dpTiles : 0) + got.btTiles != after.recomputeTiles - before.recomputeTiles) failHere("recomputed tiles", got.btTiles);
				if ((flat ? 2 : 1) * (uint64_t)got.dpCols + got.btCols != after.columnSteps - before.columnSteps) failHere("column steps", got.dpCols);
				nOk++; nCells += want.trace.size(); nTies += got.tie; capCells += got.traceCap;
				maxExcess = std::max(maxExcess, (long)want.trace.size() - (long)(part.size() + 1 + (size_t)want.score));
				if (part.size() <= 34) fragItemHist[std::min<uint32_t>(got.nItems, 31)]++;
			}
		}
	}
	printf("FRAG_HOST_OK extensions %zu equal %zu failed %zu asserted %zu declined %zu cells %zu reserved %zu ties %zu | turns per extension: POP %.1f COLS %.1f (of which walk refills %.1f) TILE_END %.1f WALK %.1f\n", nExt, nOk, nFailed, nAssert, nDeclined, nCells, capCells, nTies,
		(double)turns[gcfrag::PH_POP] / nExt, (double)turns[gcfrag::PH_COLS] / nExt, (double)turns[7] / nExt, (double)turns[gcfrag::PH_TILE_END] / nExt, (double)turns[gcfrag::PH_WALK] / nExt);
	printf("most pending nodes:");
	for (int i = 0; i < 16; i++) printf(" %llu", pendingHist[i]);
	printf("\ntiles per slice:");
	for (int i = 0; i < 32; i++) printf(" %llu", itemHist[i]);
	printf("\ntiles per slice, at most 34 rows:");
	for (int i = 0; i < 32; i++) printf(" %llu", fragItemHist[i]);
	printf("\nmost trace cells beyond rows + 1 + score: %ld\n", maxExcess);
	return 0;
}
