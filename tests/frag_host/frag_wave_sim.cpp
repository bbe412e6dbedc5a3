// A wave of the fragment extension kernel on the CPU: 64 lanes of gc_frag_core.hpp under the kernel's own schedule (column loop until SWEEP lanes wait, then one sweep
// of the handlers), fed with the extensions the fragment pass makes of a read's seeds. Counts what a wave EXECUTES - column steps, handler runs, the turns of the loops
// inside the handlers - against what its lanes USE of it, i.e. the lockstep efficiency that decides the kernel's vector instruction count, and checks
// every result against the one-lane run. A development tool for the schedule (thresholds, handler order): not a test, not in the product.
//
// usage: frag_wave_sim graph.gfa reads.txt [sweepAt] [claim]
static unsigned long long g_ticks[8];
#define GC_LOOP_TICK(i) (g_ticks[i]++)
#include "frag_host_common.hpp"

struct Item { std::string seq; uint32_t node, offset; };
struct Ref { uint32_t status; int32_t score; std::vector<gcdev::TraceCell> trace; };

static Ref runOne(const gcdev::DGraph& g, const gcfrag::FragParams& P, const uint8_t* iupac, const Item& it)
{
	using namespace gcfrag;
	FragMem<HostStore> m; m.poison();
	Lane L {}; L.phase = PH_FETCH;
	fragBegin(g, P, L, m, 0, (uint32_t)it.seq.size(), it.node, it.offset, gcdev::EqFromBases { it.seq.data(), iupac });
	while (L.phase != PH_FETCH) {
		switch (L.phase) {
			case PH_POP: fragPop(g, P, L, m); break;
			case PH_COLS: fragColumn(L, m); break;
			case PH_TILE_END: fragTileEnd(g, P, L, m); break;
			case PH_FINISH: if (fragFinish(P, L)) { m.trace.assign(L.traceCap, gcdev::TraceCell {}); L.traceBase = 0; fragWalkBegin(L, m); } break;
			case PH_WALK: fragWalkStep(g, P, L, m); break;
		}
	}
	Ref r { L.status, L.resultScore, {} };
	if (L.status == gcdev::EXT_OK) r.trace.assign(m.trace.begin(), m.trace.begin() + L.nTrace);
	return r;
}

int main(int argc, char** argv)
{
	using namespace gcfrag;
	if (argc < 3) { fprintf(stderr, "usage: %s graph.gfa reads.txt [sweepAt] [claim]\n", argv[0]); return 2; }
	const uint32_t sweepAtMax = argc > 3 ? (uint32_t)atoi(argv[3]) : 24;
	const size_t claim = argc > 4 ? (size_t)atoi(argv[4]) : 256;
	const uint32_t walkAtMax = argc > 5 ? (uint32_t)atoi(argv[5]) : 0;   // > 0: the walk's handler has a trigger of its own
	gc::GfaGraph gfa = gc::GfaGraph::LoadFromFile(argv[1]);
	AlignmentGraph graph = AlignmentGraph::BuildFromGFA(gfa);
	gc::MinimizerIndex index = gc::MinimizerIndex::Build(graph, 15, 20, 1.0 - 0.001);
	FlatGraph flat(graph);
	gcdev::CorrectnessTables ct;
	buildCorrectnessTables(ct);
	FragParams P { 10, 0 };
	for (int i = 0; i < 64; i++) if (fragSliceKept(ct, i)) P.keepMask |= 1ull << i;
	uint8_t iupac[256];
	hostIupacTable(iupac);
	std::vector<Item> items;
	std::ifstream in(argv[2]);
	std::string read;
	while (std::getline(in, read)) {
		if (read.empty()) continue;
		std::vector<SeedHit> seeds = getSeeds(graph, index, read, 10.0);
		std::sort(seeds.begin(), seeds.end(), [](const SeedHit& a, const SeedHit& b) { return a.seqPos < b.seqPos; });
		const std::string rev = gc::ReverseComplement(read);
		for (const SeedHit& sd : seeds) {
			const int forwardNodeId = sd.nodeID * 2 + (sd.reverse ? 1 : 0);
			const size_t p = sd.seqPos % 35;   // the seed's place in a 35-base window
			if (sd.seqPos + 35 - p > read.size()) continue;
			if (p > 0) {
				auto reversePos = graph.GetReversePosition(forwardNodeId, sd.nodeOffset);
				const size_t node = graph.GetUnitigNode(forwardNodeId ^ 1, reversePos.second);
				items.push_back(Item { std::string(rev.data() + rev.size() - sd.seqPos, p), (uint32_t)node, (uint32_t)(reversePos.second - graph.nodeOffset[node]) });
			}
			if (p < 34) {
				const size_t node = graph.GetUnitigNode(forwardNodeId, sd.nodeOffset);
				items.push_back(Item { std::string(read.data() + sd.seqPos + 1, 34 - p), (uint32_t)node, (uint32_t)(sd.nodeOffset - graph.nodeOffset[node]) });
			}
		}
	}
	// ---- waves of 64 lanes, one after the other (each claims `claim` items at a time from the common cursor, as the kernel's waves do)
	const size_t nWaves = std::max<size_t>(1, items.size() / 1100);   // ~1 100 items per wave: cfg2's 4.4 M extensions on 3 840 waves
	std::vector<size_t> waveOf(items.size());
	{
		// a static picture of the dynamic claims: blocks of `claim` items dealt round-robin
		size_t b = 0;
		for (size_t i = 0; i < items.size(); i += claim, b++) for (size_t k = i; k < std::min(items.size(), i + claim); k++) waveOf[k] = b % nWaves;
	}
	unsigned long long tickMax[8][8] = { { 0 } }, tickSum[8][8] = { { 0 } };   // [handler][loop]: per run the most turns of a lane, and all lanes' turns
	unsigned long long colSteps = 0, colLaneSteps = 0, sweeps = 0, run[8] = { 0 }, lanesIn[8] = { 0 }, walkTurns = 0, walkLaneTurns = 0, popScan = 0, pushes = 0, pushLanes = 0, checked = 0, declined = 0, extensions = 0;
	for (size_t w = 0; w < nWaves; w++) {
		std::vector<size_t> mine;
		for (size_t i = 0; i < items.size(); i++) if (waveOf[i] == w) mine.push_back(i);
		size_t next = 0;
		struct LaneState { Lane L; FragMem<HostStore> m; size_t item; };
		std::vector<LaneState> lane(64);
		for (auto& l : lane) { l.L = Lane {}; l.L.phase = PH_FETCH; l.L.work = 0xffffffffu; l.m.poison(); }
		for (;;) {
			uint32_t nIdle = 0, nCols = 0, nWalk = 0;
			for (auto& l : lane) { nIdle += l.L.phase == PH_IDLE; nCols += l.L.phase == PH_COLS; nWalk += l.L.phase == PH_WALK; }
			if (nIdle == 64) break;
			const uint32_t nActive = 64 - nIdle, sweepAt = std::min(nActive, sweepAtMax);
			bool runDp = true, runWalk = true;
			if (!walkAtMax) {
				uint32_t nWait = nActive - nCols;
				while (nWait < sweepAt) {
					colSteps++;
					nCols = 0;
					for (auto& l : lane) if (l.L.phase == PH_COLS) { colLaneSteps++; fragColumn(l.L, l.m); nCols += l.L.phase == PH_COLS; }
					nWait = nActive - nCols;
				}
			} else {
				for (;;) {
					nCols = nWalk = 0;
					for (auto& l : lane) { nCols += l.L.phase == PH_COLS; nWalk += l.L.phase == PH_WALK; }
					const uint32_t nDp = nActive - nCols - nWalk;
					runDp = nDp >= std::min(sweepAt, nActive - nWalk) && nDp > 0; runWalk = nWalk >= std::min(walkAtMax, nActive) ;
					if (runDp || runWalk) break;
					if (nCols == 0) { runDp = nDp >= nWalk; runWalk = !runDp; break; }
					colSteps++;
					for (auto& l : lane) if (l.L.phase == PH_COLS) { colLaneSteps++; fragColumn(l.L, l.m); }
				}
			}
			sweeps++;
			unsigned long long runMax[8]; auto runBegin = [&]() { for (auto& x : runMax) x = 0; };
			auto laneCall = [&](auto&& f) { unsigned long long b[8]; for (int i = 0; i < 8; i++) b[i] = g_ticks[i]; f(); for (int i = 0; i < 8; i++) runMax[i] = std::max(runMax[i], g_ticks[i] - b[i]); };
			auto runEnd = [&](int ph) { for (int i = 0; i < 8; i++) tickMax[ph][i] += runMax[i]; };
			auto count = [&](uint32_t ph) { uint32_t c = 0; for (auto& l : lane) c += l.L.phase == ph; return c; };
			if (uint32_t c = runDp ? count(PH_FETCH) : 0) {
				run[PH_FETCH]++; lanesIn[PH_FETCH] += c;
				for (auto& l : lane) if (l.L.phase == PH_FETCH) {
					if (l.L.work != 0xffffffffu) {
						extensions++;
						if (l.L.status == gcdev::EXT_OVERFLOW) declined++;
						else {
							const Ref want = runOne(flat.d, P, iupac, items[l.item]);
							bool same = want.status == l.L.status && (want.status != gcdev::EXT_OK || (want.score == l.L.resultScore && want.trace.size() == l.L.nTrace));
							for (size_t i = 0; same && i < want.trace.size(); i++) same = want.trace[i].node == l.m.trace[i].node && want.trace[i].seqPos == l.m.trace[i].seqPos && want.trace[i].offsetAndSwitch == l.m.trace[i].offsetAndSwitch;
							if (!same) { fprintf(stderr, "the wave's result of item %zu differs from the one-lane run\n", l.item); return 1; }
							checked++;
						}
						l.L.work = 0xffffffffu;
					}
					if (next >= mine.size()) { l.L.phase = PH_IDLE; continue; }
					l.item = mine[next++];
					const Item& it = items[l.item];
					fragBegin(flat.d, P, l.L, l.m, (uint32_t)l.item, (uint32_t)it.seq.size(), it.node, it.offset, gcdev::EqFromBases { it.seq.data(), iupac });
				}
			}
			if (uint32_t c = runDp ? count(PH_TILE_END) : 0) {
				run[PH_TILE_END]++; lanesIn[PH_TILE_END] += c;
				uint32_t most = 0;
				runBegin();
				for (auto& l : lane) if (l.L.phase == PH_TILE_END) { laneCall([&]() { fragTileEnd(flat.d, P, l.L, l.m); }); most = std::max(most, l.L.outDeg); pushLanes += l.L.outDeg; }
				runEnd(PH_TILE_END);
				pushes += most;
			}
			if (uint32_t c = runDp ? count(PH_POP) : 0) {
				run[PH_POP]++; lanesIn[PH_POP] += c;
				uint32_t most = 0;
				runBegin();
				for (auto& l : lane) if (l.L.phase == PH_POP) { most = std::max(most, l.L.nPending); laneCall([&]() { fragPop(flat.d, P, l.L, l.m); }); }
				runEnd(PH_POP);
				popScan += most;
			}
			if (uint32_t c = runDp ? count(PH_FINISH) : 0) {
				run[PH_FINISH]++; lanesIn[PH_FINISH] += c;
				for (auto& l : lane) if (l.L.phase == PH_FINISH && fragFinish(P, l.L)) { l.m.trace.assign(l.L.traceCap, gcdev::TraceCell {}); l.L.traceBase = 0; fragWalkBegin(l.L, l.m); }
			}
			if (uint32_t c = runWalk ? count(PH_WALK) : 0) {
				run[PH_WALK]++; lanesIn[PH_WALK] += c;
				for (;;) {
					uint32_t walking = 0;
					runBegin();
					for (auto& l : lane) if (l.L.phase == PH_WALK) { walking++; laneCall([&]() { fragWalkStep(flat.d, P, l.L, l.m); }); }
					runEnd(PH_WALK);
					if (!walking) break;
					walkTurns++; walkLaneTurns += walking;
				}
			}
		}
	}
	const double per = (double)extensions / 64.0;
	printf("extensions %llu (checked %llu, declined %llu) in %zu waves; per 64 extensions:\n", extensions, checked, declined, nWaves);
	printf("  column steps %.0f with %.1f lanes in them\n", colSteps / per, (double)colLaneSteps / std::max(1ull, colSteps));
	printf("  sweeps %.1f\n", sweeps / per);
	const char* names[8] = { "", "FETCH", "POP", "", "TILE_END", "FINISH", "WALK", "" };
	for (int ph : { (int)PH_FETCH, (int)PH_TILE_END, (int)PH_POP, (int)PH_FINISH, (int)PH_WALK })
		printf("  %-8s runs %.1f with %.1f lanes\n", names[ph], run[ph] / per, (double)lanesIn[ph] / std::max(1ull, run[ph]));
	printf("  walk turns %.1f with %.1f lanes; pop scans %.1f entries; push rounds %.1f (lanes' pushes %.1f)\n", walkTurns / per, (double)walkLaneTurns / std::max(1ull, walkTurns), popScan / per, pushes / per, pushLanes / per);
	printf("  loop turns a wave pays (most of a lane per run): TILE_END merge %.0f columnMin %.0f find-slot %.0f | POP scan %.0f | WALK find-item %.0f inside %.0f merge %.0f\n", tickMax[PH_TILE_END][0] / per, tickMax[PH_TILE_END][1] / per, tickMax[PH_TILE_END][3] / per,
		tickMax[PH_POP][2] / per, tickMax[PH_WALK][4] / per, tickMax[PH_WALK][5] / per, tickMax[PH_WALK][0] / per);
	// a rough price in vector instructions (the handlers' static sizes from the kernel's ISA, loops by their turns): what the schedule is tuned against
	const double cost = 75.0 * colSteps + 350.0 * run[PH_FETCH] + 150.0 * run[PH_TILE_END] + 120.0 * pushes + 30.0 * tickMax[PH_TILE_END][0] + 15.0 * tickMax[PH_TILE_END][1] + 6.0 * tickMax[PH_TILE_END][3]
		+ 200.0 * run[PH_POP] + 8.0 * tickMax[PH_POP][2] + 200.0 * run[PH_FINISH] + 350.0 * walkTurns + 60.0 * tickMax[PH_WALK][5] + 10.0 * tickMax[PH_WALK][4];
	printf("  priced: %.0f vector instructions per 64 extensions (columns %.0f, walk %.0f)\n", cost / per, 75.0 * colSteps / per, (350.0 * walkTurns + 60.0 * tickMax[PH_WALK][5] + 10.0 * tickMax[PH_WALK][4]) / per);
	return 0;
}
