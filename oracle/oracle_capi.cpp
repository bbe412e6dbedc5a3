// ORACLE (test infrastructure, not product code): per-read driver restating src/Aligner.cpp:601-922 and a
// small C interface so tests/, smoke() and bench.py's cpu_baseline leg can run it through ctypes.
// Nothing in graphchainer_amd/ may link or call this file.
#include "pipeline.hpp"
#include "output.hpp"
#include "edlib_path.hpp"
#include "evalue.hpp"
#include <cstdlib>
#include <atomic>
#include <thread>
#include <chrono>
#include <cstring>
#include <map>
#include <sstream>

namespace oracle {

struct ReadResult {
	std::vector<SeedHit> seeds;                 // fragment-pass order (after the seqPos sort, src/Aligner.cpp:667)
	std::vector<std::array<size_t, 3>> fragments;   // (l, sl, sr) of every fragment that had seeds
	std::vector<gc::Anchor> anchors;
	std::vector<std::array<MatrixPosition, 2>> apos;   // first/last trace cell of each anchor, unitig coords (:722-728)
	std::vector<std::vector<TraceItem>> anchorTraces;
	std::vector<int32_t> anchorScores;
	std::vector<size_t> chain;
	size_t chainScore = 0;
	std::vector<AlignmentItem> longAlignments;  // after GreedyLength selection (:638-640)
	std::vector<AlignmentItem> longAll;         // before selection
	std::vector<MatrixPosition> longest;        // stitched path cells (node = split node), :754-822
	size_t longEditDistance = SIZE_MAX, chainEditDistance = SIZE_MAX;
	std::vector<unsigned char> chainOps;        // edlib's op string for (stitched path, read), :845
	std::vector<AlignmentItem> chainAlignments; // the chained alignment (:878-897), after the selection of :904
	bool chainedBetter = false;
	bool failedAssertion = false;
	size_t seedsExtended = 0;
	// r5: flattenLastSliceEnd calls of this read whose minimum was attained in more than one node (the one rule whose tie order is defined, not reproduced:
	// oracle/bitvector_aligner.hpp header), in the fragment pass and in the whole-read pass
	size_t flattenTies = 0, flattenTiesLong = 0;
};

// reference: src/AlignmentSelection.cpp (GreedySelectAlignments with alignmentLengthCompare, :42-50,
// and the incompatibility rule :12-33)
// reference: SelectECutoff, src/AlignmentSelection.cpp:91-99 (first step of SelectAlignments when --E-cutoff is given, :57-61)
static std::vector<AlignmentItem> selectECutoff(const std::vector<AlignmentItem>& all, size_t graphSize, size_t readSize, double cutoff, const EValueCalc& calc)
{
	if (cutoff == -1) return all;
	std::vector<AlignmentItem> result;
	for (const auto& aln : all) if (calc.getEValue(graphSize, readSize, aln.alignmentEnd - aln.alignmentStart, aln.alignmentScore) <= cutoff) result.push_back(aln);
	return result;
}

static std::vector<AlignmentItem> selectGreedyLength(const std::vector<AlignmentItem>& all)
{
	std::vector<AlignmentItem> sorted = all;
	auto better = [](const AlignmentItem& l, const AlignmentItem& r) {
		if ((l.alignmentEnd - l.alignmentStart) > (r.alignmentEnd - r.alignmentStart)) return true;
		if ((r.alignmentEnd - r.alignmentStart) > (l.alignmentEnd - l.alignmentStart)) return false;
		return l.alignmentScore < r.alignmentScore;
	};
	std::sort(sorted.begin(), sorted.end(), better);
	auto incompatible = [](const AlignmentItem& l, const AlignmentItem& r) {
		float minOverlapLen = std::min(l.alignmentEnd - l.alignmentStart, r.alignmentEnd - r.alignmentStart) * 0.05f;
		size_t ls = l.alignmentStart, le = l.alignmentEnd, rs = r.alignmentStart, re = r.alignmentEnd;
		if (ls > rs) { std::swap(ls, rs); std::swap(le, re); }
		int overlap = 0;
		if (le > rs) overlap = (int)(le - rs);
		return overlap > minOverlapLen;
	};
	std::vector<AlignmentItem> result;
	for (const auto& aln : sorted) {
		bool ok = true;
		for (const auto& kept : result) if (incompatible(aln, kept)) { ok = false; break; }
		if (ok) result.push_back(aln);
	}
	return result;
}

// reference: src/Aligner.cpp:376-408,425-428 (traceToPoses / traceToSequence)
static std::string traceToSequence(const AlignmentGraph& g, const AlignmentItem& aln)
{
	std::string ret;
	const auto& trace = aln.trace->trace;
	size_t lastNode = 0, lastOffset = 0, lastLength = 0;
	for (size_t j = 0; j < trace.size(); j++) {
		size_t node = g.GetUnitigNode((int)trace[j].DPposition.node, trace[j].DPposition.nodeOffset);
		size_t off = trace[j].DPposition.nodeOffset - g.NodeOffset(node);
		if (j == 0) {
			lastNode = node; lastOffset = off; lastLength = g.NodeLength(node);
			ret.push_back(g.NodeSequences(lastNode, lastOffset));
			lastOffset++;
		} else {
			if (node != lastNode) {
				while (lastOffset < lastLength) { ret.push_back(g.NodeSequences(lastNode, lastOffset)); lastOffset++; }
				lastNode = node; lastLength = g.NodeLength(node); lastOffset = 0;
			}
			while (lastOffset <= off) { ret.push_back(g.NodeSequences(lastNode, lastOffset)); lastOffset++; }
		}
	}
	return ret;
}

// reference: src/Aligner.cpp:409-424
static std::vector<MatrixPosition> pathToTrace(const AlignmentGraph& g, const std::vector<size_t>& path, size_t firstNodeOffset, size_t lastNodeOffset)
{
	std::vector<MatrixPosition> ret;
	for (size_t node : path) {
		size_t S = 0, L = g.NodeLength(node);
		if (node == path[0]) S = firstNodeOffset;
		else if (node == path.back()) L = lastNodeOffset + 1;
		for (size_t o = S; o < L; o++) ret.push_back(MatrixPosition { node, o, 0 });
	}
	return ret;
}

class Oracle {
public:
	gc::AlignmentGraph graph;
	gc::MinimizerIndex index;
	Params params;
	AlignerCounters counters;
	EValueCalc evalueCalc { .7 };   // src/Aligner.cpp:478-482 (precise clipping off)
	double stageSeconds[5] = { 0, 0, 0, 0, 0 };   // seed, long pass, fragments, chaining, stitch+edit distance

	// forceLongAssertion: test hook, makes the whole-read pass end as if one of the reference's live asserts had thrown
	ReadResult alignRead(const std::string& sequence, AlignerState& state, bool forceLongAssertion = false, double* stageOut = nullptr)
	{
		double* stageSeconds = stageOut ? stageOut : this->stageSeconds;
		typedef std::chrono::steady_clock clk;
		auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
		ReadResult res;
		// `cont` is declared once per read (src/Aligner.cpp:529), set by align_fn's catch (:591) and by "no seed hits" (:551), and
		// tested after every fragment (:702-703): once the whole-read pass has thrown, no fragment of the read adds an anchor.
		bool cont = false;
		// ---- A. whole-read pass (src/Aligner.cpp:630-654 -> align_fn :531-594)
		const uint64_t tiesAtStart = state.counters.flattenTies;
		if (params.longPass) {
			auto t0 = clk::now();
			std::vector<SeedHit> seeds = getSeeds(graph, index, sequence, params.seedDensity);
			auto t1 = clk::now();
			stageSeconds[0] += secs(t0, t1);
			if (!seeds.empty()) {
				try {
					orderSeedsByChaining(graph, seeds);
					GraphAligner aligner(graph, params, true);
					AlignmentResult r = aligner.AlignOneWay(sequence, seeds, state, 0, seeds.size(), 0);
					if (forceLongAssertion) throw AssertionFailure("forced by the test hook");
					res.longAll = r.alignments;
				} catch (const AssertionFailure&) {
					state.clear();
					res.failedAssertion = true;
					res.longAll.clear();
					cont = true;   // :591
				}
			} else cont = true;   // :551
			if (!res.longAll.empty()) res.longAlignments = selectGreedyLength(selectECutoff(res.longAll, graph.SizeInBP(), sequence.size(), params.eCutoff, evalueCalc));
			if (!res.longAlignments.empty()) res.longEditDistance = editDistanceNW(traceToSequence(graph, res.longAlignments[0]), sequence);
			stageSeconds[1] += secs(t1, clk::now());
		}
		res.flattenTiesLong = state.counters.flattenTies - tiesAtStart;
		// ---- B. fragment pass (:658-730)
		auto t2 = clk::now();
		res.seeds = getSeeds(graph, index, sequence, params.seedDensity);
		auto t3 = clk::now();
		stageSeconds[0] += secs(t2, t3);
		if (res.seeds.empty()) return res;
		orderSeedsByChaining(graph, res.seeds);
		std::sort(res.seeds.begin(), res.seeds.end(), [](const SeedHit& l, const SeedHit& r) { return l.seqPos < r.seqPos; });
		size_t len = params.splitLen, sep = params.splitGap;
		size_t sl = 0, sr = 0;
		// (cont is also never reset once a fragment has failed, src/Aligner.cpp:695-703)
		GraphAligner fragmentAligner(graph, params, false);
		for (size_t l = 0; l + len <= sequence.size(); l += sep) {
			while (sr < res.seeds.size() && res.seeds[sr].seqPos + res.seeds[sr].matchLen <= l + len) sr++;
			while (sl < sr && res.seeds[sl].seqPos < l) sl++;
			if (sl >= sr) continue;
			res.fragments.push_back({ l, sl, sr });
			std::string seq = sequence.substr(l, len);
			AlignmentResult alignments;
			try {
				alignments = fragmentAligner.AlignOneWay(seq, res.seeds, state, sl, sr, l);
			} catch (const AssertionFailure&) {
				state.clear();
				res.failedAssertion = true;
				cont = true;
			}
			if (cont) continue;
			res.seedsExtended += alignments.seedsExtended;
			for (auto& alignment : alignments.alignments) {
				if (alignment.alignmentFailed()) continue;
				const auto& trace = alignment.trace->trace;
				if (trace.empty()) continue;
				gc::Anchor anchor { {}, l, l + len - 1 };
				for (const TraceItem& t : trace) {
					size_t node = graph.GetUnitigNode((int)t.DPposition.node, t.DPposition.nodeOffset);
					if (anchor.path.empty() || node != anchor.path.back()) anchor.path.push_back(node);
				}
				res.anchors.push_back(anchor);
				std::array<MatrixPosition, 2> ends { trace[0].DPposition, trace.back().DPposition };
				for (auto& p : ends) {
					p.seqPos += l;
					size_t bigraphOffset = p.nodeOffset;
					p.node = graph.GetUnitigNode((int)p.node, bigraphOffset);
					p.nodeOffset = bigraphOffset - graph.NodeOffset(p.node);
				}
				res.apos.push_back(ends);
				res.anchorTraces.push_back(trace);
				res.anchorScores.push_back(alignment.trace->score);
			}
		}
		auto t4 = clk::now();
		stageSeconds[2] += secs(t3, t4);
		res.flattenTies = state.counters.flattenTies - tiesAtStart - res.flattenTiesLong;
		// ---- chaining (:735)
		auto chained = colinearChaining(graph, res.anchors);
		res.chain = chained.first;
		res.chainScore = chained.second;
		auto t5 = clk::now();
		stageSeconds[3] += secs(t4, t5);
		// ---- C. chain -> path (:754-822)
		stitch(res, sequence);
		stageSeconds[4] += secs(t5, clk::now());
		return res;
	}

private:
	void stitch(ReadResult& res, const std::string& sequence)
	{
		const auto& A = res.anchors;
		const auto& Apos = res.apos;
		std::vector<MatrixPosition> longest, tmp;
		std::vector<size_t> pos_path;
		std::unordered_set<size_t> nodes;
		size_t firstNodeOffset = 0, lastNodeOffset = 0;
		for (size_t ai : res.chain) {
			const gc::Anchor& anchor = A[ai];
			if (pos_path.empty()) {
				pos_path = anchor.path;
				firstNodeOffset = Apos[ai][0].nodeOffset;
				lastNodeOffset = Apos[ai][1].nodeOffset;
				for (size_t j : pos_path) nodes.insert(j);
			} else {
				bool gap = anchor.path[0] == pos_path.back() && params.colinearGap != -1 && (long long)Apos[ai][0].nodeOffset - (long long)lastNodeOffset > params.colinearGap + 1;
				std::vector<size_t> path;
				if (!nodes.count(anchor.path[0]) && pos_path.back() != Apos[ai][0].node) {
					long long gapLimit = params.colinearGap;
					if (gapLimit != -1) gapLimit -= (long long)Apos[ai][0].nodeOffset + (long long)(graph.NodeLength(pos_path.back()) - (long long)lastNodeOffset - 1);
					path = graph.getChainPath(pos_path.back(), Apos[ai][0].node, gapLimit);
					if (path.empty()) gap = true;
				}
				if (gap) {
					tmp = pathToTrace(graph, pos_path, firstNodeOffset, lastNodeOffset);
					if (longest.size() < tmp.size()) longest.swap(tmp);
					nodes.clear();
					pos_path.clear();
					firstNodeOffset = Apos[ai][0].nodeOffset;
				} else {
					for (size_t j : path) if (!nodes.count(j)) { nodes.insert(j); pos_path.push_back(j); }
				}
				for (size_t j : anchor.path) if (!nodes.count(j)) { nodes.insert(j); pos_path.push_back(j); }
				lastNodeOffset = Apos[ai][1].nodeOffset;
			}
		}
		if (!pos_path.empty()) {
			tmp = pathToTrace(graph, pos_path, firstNodeOffset, lastNodeOffset);
			if (longest.size() < tmp.size()) longest.swap(tmp);
		}
		res.longest = longest;
		std::string pathseq;
		for (const auto& p : longest) pathseq.push_back(graph.NodeSequences(p.node, p.nodeOffset));
		// :845 edlibAlign(pathseq, read, NW, PATH) -> :848-876 the op string walked over `longest` and the read (indices clamped to the
		// last valid cell) -> :878-897 trace items in output coordinates, the alignment item -> :901-905 selection and the decision.
		// edlibAlign on an empty path returns no alignment; an empty `longest` produces no alignment item anyway (:890).
		if (!longest.empty()) {
			res.chainEditDistance = (size_t)edlibPathNW(pathseq, sequence, res.chainOps);
			std::vector<MatrixPosition> cells;
			size_t pos_i = 0, seq_i = 0;
			for (unsigned char c : res.chainOps) {
				cells.push_back(MatrixPosition { longest[pos_i].node, longest[pos_i].nodeOffset, seq_i });
				if (c == 0 || c == 3) { pos_i++; seq_i++; }
				else if (c == 1) pos_i++;
				else if (c == 2) seq_i++;
				seq_i = std::min(seq_i, sequence.size() - 1);
				pos_i = std::min(pos_i, longest.size() - 1);
			}
			auto trace = std::make_shared<OnewayTrace>();   // trace.score stays 0: the reference never sets it (:739,891-893)
			for (size_t i = 0; i < cells.size(); i++) {
				bool nodeSwitch = i + 1 < cells.size() && cells[i].node != cells[i + 1].node;
				TraceItem item { cells[i], nodeSwitch, cells[i].seqPos < sequence.size() ? sequence[cells[i].seqPos] : '-', graph.NodeSequences(cells[i].node, cells[i].nodeOffset) };
				item.DPposition.nodeOffset += graph.NodeOffset(item.DPposition.node);
				item.DPposition.node = (size_t)graph.nodeIDs[item.DPposition.node];
				trace->trace.push_back(item);
			}
			if (!trace->trace.empty()) {
				AlignmentItem item;
				item.trace = trace;
				item.alignmentScore = res.chainEditDistance;
				item.alignmentStart = trace->trace[0].DPposition.seqPos;
				item.alignmentEnd = trace->trace.back().DPposition.seqPos + 1;
				res.chainAlignments = selectECutoff({ item }, graph.SizeInBP(), sequence.size(), params.eCutoff, evalueCalc);   // :904, method All
			}
			if (!res.chainAlignments.empty()) res.chainedBetter = res.longAlignments.empty() || res.longEditDistance > res.chainAlignments.front().alignmentScore;   // :905
		}
	}
};

// ---- flat export -----------------------------------------------------------------------------------

struct Export {
	std::map<std::string, std::vector<int64_t>> arrays;
	void clear() { arrays.clear(); }
	std::vector<int64_t>& operator[](const std::string& k) { return arrays[k]; }
};

} // namespace oracle

using namespace oracle;

struct OracleHandle {
	Oracle o;
	Export ex;
	std::string gaf[2];   // GAF text of the last gco_align call (read ids r0, r1, ...): [0] =/X cigar, [1] merged M cigar
	std::string json;     // JSON lines of the same alignments
	std::string gam;      // the same alignments as the INFLATED GAM stream: per read with output one group (varint count, then varint size + vg::Alignment bytes each), src/Aligner.cpp:261-281
	std::vector<uint64_t> gamGroupOff { 0 };   // group boundaries in gam (the reference deflates every group into its own gzip member)
	std::string error;
	int tieOrder = 0;     // flattenLastSliceEnd's node order (AlignerState::tieOrder): 0 the defined order, 1 reversed (sensitivity runs)
};

extern "C" {

void* gco_create(const char* gfaPath, int k, int w, double density, double discardFraction, int bandwidth, int splitLen, int splitGap, long long colinearGap, int longPass, int shrinkMpc)
{
	OracleHandle* h = new OracleHandle();
	try {
		gc::GfaGraph gfa = gc::GfaGraph::LoadFromFile(gfaPath);
		h->o.graph = gc::AlignmentGraph::BuildFromGFA(gfa);
		h->o.graph.buildMPC(shrinkMpc != 0);
		h->o.params.k = k; h->o.params.w = w; h->o.params.seedDensity = density;
		h->o.params.discardMostNumerousFraction = discardFraction;
		h->o.params.bandwidth = bandwidth; h->o.params.splitLen = splitLen; h->o.params.splitGap = splitGap;
		h->o.params.colinearGap = colinearGap; h->o.params.longPass = longPass != 0;
		h->o.index = gc::MinimizerIndex::Build(h->o.graph, k, w, 1.0 - discardFraction);
	} catch (const std::exception& e) {
		h->error = e.what();
	}
	return h;
}

const char* gco_error(void* hv) { return ((OracleHandle*)hv)->error.c_str(); }
void gco_destroy(void* hv) { delete (OracleHandle*)hv; }

// Aligns n reads (concatenated in `bases`, read i = [off[i], off[i+1])) and stores flat result arrays.
int gco_align(void* hv, const char* bases, const uint64_t* off, int n)
{
	OracleHandle* h = (OracleHandle*)hv;
	Export& ex = h->ex;
	ex.clear();
	AlignerState state(h->o.graph);
	state.tieOrder = h->tieOrder;
	h->o.counters = AlignerCounters();
	for (double& s : h->o.stageSeconds) s = 0;
	h->gaf[0].clear(); h->gaf[1].clear(); h->json.clear(); h->gam.clear(); h->gamGroupOff.assign(1, 0);
	const char* names[] = { "read_seed_off", "read_frag_off", "read_anchor_off", "read_chain_off", "read_long_off", "read_longall_off", "read_path_off", "anchor_path_off", "anchor_trace_off", "long_trace_off", "read_chain_ops_off", "read_chain_trace_off" };
	for (const char* nm : names) ex[nm].push_back(0);
	for (int r = 0; r < n; r++) {
		std::string seq(bases + off[r], bases + off[r + 1]);
		ReadResult res;
		try {
			const char* failLong = getenv("GC_TEST_FAIL_LONG");   // test hook shared with the product: index of a read whose whole-read pass "asserts"
			res = h->o.alignRead(seq, state, failLong && atoi(failLong) == r);
		} catch (const std::exception& e) {
			h->error = std::string("read ") + std::to_string(r) + ": " + e.what();
			return 1;
		}
		for (const SeedHit& s : res.seeds) {
			ex["seed_node"].push_back((int64_t)s.alignmentGraphNodeId);
			ex["seed_offset"].push_back((int64_t)s.alignmentGraphNodeOffset);
			ex["seed_seqpos"].push_back((int64_t)s.seqPos);
			ex["seed_goodness"].push_back((int64_t)s.seedGoodness);
		}
		ex["read_seed_off"].push_back((int64_t)ex["seed_node"].size());
		for (auto& f : res.fragments) { ex["frag_l"].push_back(f[0]); ex["frag_sl"].push_back(f[1]); ex["frag_sr"].push_back(f[2]); }
		ex["read_frag_off"].push_back((int64_t)ex["frag_l"].size());
		for (size_t a = 0; a < res.anchors.size(); a++) {
			ex["anchor_x"].push_back(res.anchors[a].x);
			ex["anchor_y"].push_back(res.anchors[a].y);
			for (size_t p : res.anchors[a].path) ex["anchor_path"].push_back((int64_t)p);
			ex["anchor_path_off"].push_back((int64_t)ex["anchor_path"].size());
			for (int e = 0; e < 2; e++) {
				ex[e ? "anchor_last_node" : "anchor_first_node"].push_back((int64_t)res.apos[a][e].node);
				ex[e ? "anchor_last_offset" : "anchor_first_offset"].push_back((int64_t)res.apos[a][e].nodeOffset);
				ex[e ? "anchor_last_seqpos" : "anchor_first_seqpos"].push_back((int64_t)res.apos[a][e].seqPos);
			}
			ex["anchor_score"].push_back(res.anchorScores[a]);
			for (const TraceItem& t : res.anchorTraces[a]) {
				ex["anchor_trace_node"].push_back((int64_t)t.DPposition.node);
				ex["anchor_trace_offset"].push_back((int64_t)t.DPposition.nodeOffset);
				ex["anchor_trace_seqpos"].push_back((int64_t)t.DPposition.seqPos);
				ex["anchor_trace_switch"].push_back(t.nodeSwitch ? 1 : 0);
			}
			ex["anchor_trace_off"].push_back((int64_t)ex["anchor_trace_node"].size());
		}
		ex["read_anchor_off"].push_back((int64_t)ex["anchor_x"].size());
		for (size_t c : res.chain) ex["chain"].push_back((int64_t)c);
		ex["read_chain_off"].push_back((int64_t)ex["chain"].size());
		ex["chain_score"].push_back((int64_t)res.chainScore);
		// final alignments of the read (src/Aligner.cpp:901-920): the chained alignment when it won, else the selected whole-read
		// alignments; AddAlignment / AddGAFLine each (:1006-1019), sorted by alignmentStart (:1003,:1023), one line each (:300-311)
		{
			std::vector<AlignmentItem> finalAlns = res.chainedBetter ? res.chainAlignments : res.longAlignments;
			auto byStart = [](const AlignmentItem& l, const AlignmentItem& rr) { return l.alignmentStart < rr.alignmentStart; };
			std::sort(finalAlns.begin(), finalAlns.end(), byStart);   // :1003
			std::sort(finalAlns.begin(), finalAlns.end(), byStart);   // :1023 (AddAlignment / AddGAFLine in between do not reorder)
			std::vector<std::string> messages;
			for (const AlignmentItem& a : finalAlns) {
				for (int m = 0; m < 2; m++) { h->gaf[m] += traceToGaf(h->o.graph, "r" + std::to_string(r), seq, *a.trace, m == 1); h->gaf[m] += '\n'; }
				OraAlignment vg = buildVgAlignment(h->o.graph, "r" + std::to_string(r), seq, a);
				h->json += vgAlignmentToJson(vg);
				h->json += '\n';
				messages.push_back(vgAlignmentToProto(vg));
			}
			if (!messages.empty()) {   // a read without alignments leaves runComponentMappings before the writers (src/Aligner.cpp:977-992)
				h->gam += gamGroup(messages);
				h->gamGroupOff.push_back(h->gam.size());
			}
		}
		for (unsigned char c : res.chainOps) ex["chain_ops"].push_back(c);
		ex["read_chain_ops_off"].push_back((int64_t)ex["chain_ops"].size());
		for (const AlignmentItem& a : res.chainAlignments) {
			for (const TraceItem& t : a.trace->trace) {
				ex["chain_trace_node"].push_back((int64_t)t.DPposition.node);
				ex["chain_trace_offset"].push_back((int64_t)t.DPposition.nodeOffset);
				ex["chain_trace_seqpos"].push_back((int64_t)t.DPposition.seqPos);
				ex["chain_trace_switch"].push_back(t.nodeSwitch ? 1 : 0);
			}
		}
		ex["read_chain_trace_off"].push_back((int64_t)ex["chain_trace_node"].size());
		ex["chain_aln_start"].push_back(res.chainAlignments.empty() ? -1 : (int64_t)res.chainAlignments[0].alignmentStart);
		ex["chain_aln_end"].push_back(res.chainAlignments.empty() ? -1 : (int64_t)res.chainAlignments[0].alignmentEnd);
		auto dumpAlns = [&](const std::vector<AlignmentItem>& alns, const std::string& prefix, bool traces) {
			for (const AlignmentItem& aln : alns) {
				ex[prefix + "_start"].push_back((int64_t)aln.alignmentStart);
				ex[prefix + "_end"].push_back((int64_t)aln.alignmentEnd);
				ex[prefix + "_score"].push_back((int64_t)aln.alignmentScore);
				if (!traces) continue;
				for (const TraceItem& t : aln.trace->trace) {
					ex["long_trace_node"].push_back((int64_t)t.DPposition.node);
					ex["long_trace_offset"].push_back((int64_t)t.DPposition.nodeOffset);
					ex["long_trace_seqpos"].push_back((int64_t)t.DPposition.seqPos);
					ex["long_trace_switch"].push_back(t.nodeSwitch ? 1 : 0);
				}
				ex["long_trace_off"].push_back((int64_t)ex["long_trace_node"].size());
			}
		};
		dumpAlns(res.longAlignments, "long", false);
		ex["read_long_off"].push_back((int64_t)ex["long_start"].size());
		dumpAlns(res.longAll, "longall", true);
		ex["read_longall_off"].push_back((int64_t)ex["longall_start"].size());
		for (const auto& p : res.longest) { ex["path_node"].push_back((int64_t)p.node); ex["path_offset"].push_back((int64_t)p.nodeOffset); }
		ex["read_path_off"].push_back((int64_t)ex["path_node"].size());
		ex["long_edit_distance"].push_back(res.longEditDistance == SIZE_MAX ? -1 : (int64_t)res.longEditDistance);
		ex["chain_edit_distance"].push_back(res.chainEditDistance == SIZE_MAX ? -1 : (int64_t)res.chainEditDistance);
		ex["chained_better"].push_back(res.chainedBetter ? 1 : 0);
		ex["failed_assertion"].push_back(res.failedAssertion ? 1 : 0);
		ex["seeds_extended"].push_back((int64_t)res.seedsExtended);
		ex["flatten_ties"].push_back((int64_t)res.flattenTies);
		ex["flatten_ties_long"].push_back((int64_t)res.flattenTiesLong);
	}
	const AlignerCounters& c = state.counters;
	ex["counters"] = { (int64_t)c.dpTiles, (int64_t)c.recomputeTiles, (int64_t)c.columnSteps, (int64_t)c.traceItems, (int64_t)c.extensions };
	ex["flatten_counters"] = { (int64_t)c.flattenCalls, (int64_t)c.flattenTies };   // calls of flattenLastSliceEnd; extensions whose backtrace started from a tied minimum
	for (double s : h->o.stageSeconds) ex["stage_microseconds"].push_back((int64_t)(s * 1e6));
	return 0;
}

// CPU baseline leg of bench.py: the reference's threading model (src/Aligner.cpp:1267-1270: one worker per thread over a shared
// read queue, each with its own reusable state). Returns the wall seconds; stage5 (may be NULL) receives the per-stage CPU seconds
// summed over the workers, in the order seed / whole-read pass / fragments / chaining / stitch + edlib. `summary` (may be NULL)
// receives 14 values per read - what bench.py's parity sample compares with the timed GPU output after the timed region
// (src/Aligner.cpp:630-654,735,901-905): anchors, chain length, chain hash, chain score, whole-read NW distance, chain NW distance,
// chained_better, whole-read alignments, their (start, end, score) hash, selected alignments, their hash, failed assertion, and (r5) the read's
// flattenLastSliceEnd ties in the fragment pass and in the whole-read pass (ReadResult::flattenTies / flattenTiesLong).
// hash of a list v[0..m) = sum (v[i] + 1) * (i + 1) * 2654435761 mod 2^64 (bench.py computes the same with numpy).
static uint64_t listHashStep(uint64_t h, uint64_t index, int64_t v) { return h + ((uint64_t)v + 1) * ((index + 1) * 2654435761ull); }
// gafHash (optional, r4): per read one more value - the hash of the GAF lines the reference would write for it (the final alignments in output order, =/X CIGAR;
// every line from its first TAB to its newline, i.e. without the read name): line hash = list hash of its bytes, read hash = list hash of its lines' hashes.
// bench.py compares it with the text gc_format_gaf returns in the end-to-end leg: traces, paths and CIGARs of the timed mode, not just summary values.
double gco_align_summary2(void* hv, const char* bases, const uint64_t* off, int n, int threads, double* stage5, int64_t* summary, int64_t* gafHash);
double gco_align_summary(void* hv, const char* bases, const uint64_t* off, int n, int threads, double* stage5, int64_t* summary) { return gco_align_summary2(hv, bases, off, n, threads, stage5, summary, nullptr); }
double gco_align_summary2(void* hv, const char* bases, const uint64_t* off, int n, int threads, double* stage5, int64_t* summary, int64_t* gafHash)
{
	OracleHandle* h = (OracleHandle*)hv;
	if (threads < 1) threads = 1;
	std::atomic<int> next { 0 };
	std::vector<std::array<double, 5>> stages(threads, std::array<double, 5> { 0, 0, 0, 0, 0 });
	auto t0 = std::chrono::steady_clock::now();
	auto worker = [&](int t) {
		AlignerState state(h->o.graph);
		state.tieOrder = h->tieOrder;
		for (int r; (r = next.fetch_add(1)) < n;) {
			std::string seq(bases + off[r], bases + off[r + 1]);
			int64_t* out = summary ? summary + 14 * (size_t)r : nullptr;
			try {
				ReadResult res = h->o.alignRead(seq, state, false, stages[t].data());
				if (gafHash) {
					std::vector<AlignmentItem> finalAlns = res.chainedBetter ? res.chainAlignments : res.longAlignments;
					auto byStart = [](const AlignmentItem& l, const AlignmentItem& rr) { return l.alignmentStart < rr.alignmentStart; };
					std::sort(finalAlns.begin(), finalAlns.end(), byStart);   // src/Aligner.cpp:1003
					std::sort(finalAlns.begin(), finalAlns.end(), byStart);   // :1023
					uint64_t readHash = 0, line = 0;
					for (const AlignmentItem& a : finalAlns) {
						std::string text = traceToGaf(h->o.graph, "r", seq, *a.trace, false);
						text += '\n';
						uint64_t lineHash = 0;
						for (size_t i = text.find('\t'), k = 0; i < text.size(); i++, k++) lineHash = listHashStep(lineHash, k, (int64_t)(unsigned char)text[i]);
						readHash = listHashStep(readHash, line++, (int64_t)lineHash);
					}
					gafHash[r] = (int64_t)readHash;
				}
				if (!out) continue;
				uint64_t hc = 0, ha = 0, hs = 0;
				for (size_t i = 0; i < res.chain.size(); i++) hc = listHashStep(hc, i, (int64_t)res.chain[i]);
				auto alnHash = [](const std::vector<AlignmentItem>& alns) {
					uint64_t x = 0;
					for (size_t i = 0; i < alns.size(); i++) { x = listHashStep(x, 3 * i, (int64_t)alns[i].alignmentStart); x = listHashStep(x, 3 * i + 1, (int64_t)alns[i].alignmentEnd); x = listHashStep(x, 3 * i + 2, (int64_t)alns[i].alignmentScore); }
					return x;
				};
				ha = alnHash(res.longAll); hs = alnHash(res.longAlignments);
				out[0] = (int64_t)res.anchors.size(); out[1] = (int64_t)res.chain.size(); out[2] = (int64_t)hc; out[3] = (int64_t)res.chainScore;
				out[4] = res.longEditDistance == SIZE_MAX ? -1 : (int64_t)res.longEditDistance;
				out[5] = res.chainEditDistance == SIZE_MAX ? -1 : (int64_t)res.chainEditDistance;
				out[6] = res.chainedBetter ? 1 : 0; out[7] = (int64_t)res.longAll.size(); out[8] = (int64_t)ha;
				out[9] = (int64_t)res.longAlignments.size(); out[10] = (int64_t)hs; out[11] = res.failedAssertion ? 1 : 0;
				out[12] = (int64_t)res.flattenTies; out[13] = (int64_t)res.flattenTiesLong;
			} catch (const std::exception&) { state.clear(); if (out) { for (int k = 0; k < 14; k++) out[k] = 0; out[11] = 2; } if (gafHash) gafHash[r] = 0; }
		}
	};
	std::vector<std::thread> pool;
	for (int t = 1; t < threads; t++) pool.emplace_back(worker, t);
	worker(0);
	for (auto& th : pool) th.join();
	double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	if (stage5) for (int k = 0; k < 5; k++) { stage5[k] = 0; for (int t = 0; t < threads; t++) stage5[k] += stages[t][k]; }
	return wall;
}
double gco_align_timed(void* hv, const char* bases, const uint64_t* off, int n, int threads, double* stage5) { return gco_align_summary(hv, bases, off, n, threads, stage5, nullptr); }

const char* gco_gaf(void* hv, int merge) { return ((OracleHandle*)hv)->gaf[merge ? 1 : 0].c_str(); }
const char* gco_json(void* hv) { return ((OracleHandle*)hv)->json.c_str(); }
// the inflated GAM stream of the last gco_align call; *groupOff / *nGroups: the groups' boundaries (nGroups + 1 offsets)
const char* gco_gam(void* hv, uint64_t* len, const uint64_t** groupOff, uint64_t* nGroups)
{
	OracleHandle* h = (OracleHandle*)hv;
	*len = h->gam.size(); *groupOff = h->gamGroupOff.data(); *nGroups = h->gamGroupOff.size() - 1;
	return h->gam.data();
}

const int64_t* gco_array(void* hv, const char* name, uint64_t* count)
{
	OracleHandle* h = (OracleHandle*)hv;
	auto it = h->ex.arrays.find(name);
	if (it == h->ex.arrays.end()) { *count = 0; return nullptr; }
	*count = it->second.size();
	return it->second.data();
}

// ---- graph introspection for tests (arrays of the A0 data) ------------------------------------------
int gco_graph_array(void* hv, const char* name)
{
	OracleHandle* h = (OracleHandle*)hv;
	const gc::AlignmentGraph& g = h->o.graph;
	std::vector<int64_t>& out = h->ex[std::string("graph_") + name];
	out.clear();
	std::string nm = name;
	size_t n = g.NodeSize();
	if (nm == "nodeLength") for (size_t i = 0; i < n; i++) out.push_back(g.nodeLength[i]);
	else if (nm == "nodeOffset") for (size_t i = 0; i < n; i++) out.push_back(g.nodeOffset[i]);
	else if (nm == "nodeIDs") for (size_t i = 0; i < n; i++) out.push_back(g.nodeIDs[i]);
	else if (nm == "reverse") for (size_t i = 0; i < n; i++) out.push_back(g.reverse[i]);
	else if (nm == "linearizable") for (size_t i = 0; i < n; i++) out.push_back(g.linearizable[i]);
	else if (nm == "componentNumber") for (size_t i = 0; i < n; i++) out.push_back(g.componentNumber[i]);
	else if (nm == "chainNumber") for (size_t i = 0; i < n; i++) out.push_back(g.chainNumber[i]);
	else if (nm == "chainApproxPos") for (size_t i = 0; i < n; i++) out.push_back(g.chainApproxPos[i]);
	else if (nm == "component_map") for (size_t i = 0; i < n; i++) out.push_back(g.component_map[i]);
	else if (nm == "out_off") { out.push_back(0); for (size_t i = 0; i < n; i++) out.push_back(out.back() + (int64_t)g.outNeighbors[i].size()); }
	else if (nm == "out_adj") for (size_t i = 0; i < n; i++) for (size_t v : g.outNeighbors[i]) out.push_back(v);
	else if (nm == "in_off") { out.push_back(0); for (size_t i = 0; i < n; i++) out.push_back(out.back() + (int64_t)g.inNeighbors[i].size()); }
	else if (nm == "in_adj") for (size_t i = 0; i < n; i++) for (size_t v : g.inNeighbors[i]) out.push_back(v);
	else if (nm == "sequence") for (size_t i = 0; i < n; i++) for (size_t j = 0; j < g.nodeLength[i]; j++) out.push_back(g.NodeSequences(i, j));
	else if (nm == "mpc_width") for (size_t c = 0; c < g.mpc.size(); c++) out.push_back(g.mpc[c].size());
	else if (nm == "component_idx") for (size_t i = 0; i < n; i++) out.push_back(g.component_idx[i]);
	else if (nm == "topo_id") for (size_t i = 0; i < n; i++) out.push_back(g.topo_ids[g.component_map[i]][g.component_idx[i]]);
	// the path cover: mpc_path_comp[p] = component of path p, mpc_path_off / mpc_path_nodes = its nodes (global ids) in order
	else if (nm == "mpc_path_comp") { for (size_t c = 0; c < g.mpc.size(); c++) for (size_t k = 0; k < g.mpc[c].size(); k++) out.push_back(c); }
	else if (nm == "mpc_path_off") { out.push_back(0); for (size_t c = 0; c < g.mpc.size(); c++) for (const auto& p : g.mpc[c]) out.push_back(out.back() + (int64_t)p.size()); }
	else if (nm == "mpc_path_nodes") { for (size_t c = 0; c < g.mpc.size(); c++) for (const auto& p : g.mpc[c]) for (size_t x : p) out.push_back(x); }
	// per node (global id): the path ids through it (local to its component), and the backward links (node = global id, path id)
	else if (nm == "paths_off") { out.push_back(0); for (size_t i = 0; i < n; i++) out.push_back(out.back() + (int64_t)g.paths[g.component_map[i]][g.component_idx[i]].size()); }
	else if (nm == "paths") { for (size_t i = 0; i < n; i++) for (size_t k : g.paths[g.component_map[i]][g.component_idx[i]]) out.push_back(k); }
	else if (nm == "back_off") { out.push_back(0); for (size_t i = 0; i < n; i++) out.push_back(out.back() + (int64_t)g.backwards[g.component_map[i]][g.component_idx[i]].size()); }
	else if (nm == "back_node") { for (size_t i = 0; i < n; i++) for (const auto& b : g.backwards[g.component_map[i]][g.component_idx[i]]) out.push_back(g.component_ids[g.component_map[i]][b.first]); }
	else if (nm == "back_path") { for (size_t i = 0; i < n; i++) for (const auto& b : g.backwards[g.component_map[i]][g.component_idx[i]]) out.push_back(b.second); }
	else if (nm == "index_kmers") for (uint64_t v : h->o.index.kmers) out.push_back((int64_t)v);
	else if (nm == "index_start") for (uint64_t v : h->o.index.startPos) out.push_back((int64_t)v);
	else if (nm == "index_positions") for (uint64_t v : h->o.index.positions) out.push_back((int64_t)v);
	else if (nm == "index_maxcount") out.push_back((int64_t)h->o.index.maxCount);
	else return 1;
	return 0;
}

// ---- unit-level entry points (checked against oracle/_ref in tests/test_oracle_units.py) -----------
void gco_merge(uint64_t avp, uint64_t avn, int32_t as, uint64_t bvp, uint64_t bvn, int32_t bs, uint64_t* vp, uint64_t* vn, int32_t* s)
{
	WordSlice r = mergeTwoSlices(WordSlice(avp, avn, as), WordSlice(bvp, bvn, bs));
	*vp = r.VP; *vn = r.VN; *s = r.scoreEnd;
}
int32_t gco_changed_min_score(uint64_t avp, uint64_t avn, int32_t as, uint64_t bvp, uint64_t bvn, int32_t bs) { return changedMinScore(WordSlice(avp, avn, as), WordSlice(bvp, bvn, bs)); }
int32_t gco_get_value(uint64_t vp, uint64_t vn, int32_t s, int row) { return WordSlice(vp, vn, s).getValue(row); }
int32_t gco_score_before_start(uint64_t vp, uint64_t vn, int32_t s) { return WordSlice(vp, vn, s).getScoreBeforeStart(); }
void gco_next_slice(uint64_t eq, uint64_t vp, uint64_t vn, int32_t s, uint64_t hinP, uint64_t hinN, uint64_t* ovp, uint64_t* ovn, int32_t* os, uint64_t* houtP, uint64_t* houtN)
{
	StepResult r = getNextSlice(eq, WordSlice(vp, vn, s), hinP, hinN);
	*ovp = r.ws.VP; *ovn = r.ws.VN; *os = r.ws.scoreEnd; *houtP = r.houtP; *houtN = r.houtN;
}
// runs the correctness HMM over a series of per-slice mismatch counts; out[i] = {correctLogOdds, falseLogOdds, flags}
void gco_correctness_series(const int* mismatches, int n, double* correct, double* wrong, int* flags)
{
	CorrectnessState st;
	for (int i = 0; i < n; i++) {
		st = st.NextState(mismatches[i]);
		correct[i] = st.correctLogOdds;
		wrong[i] = st.falseLogOdds;
		flags[i] = (st.CurrentlyCorrect() ? 1 : 0) | (st.CorrectFromCorrect() ? 2 : 0) | (st.FalseFromCorrect() ? 4 : 0);
	}
}
// edlibAlign(a, b, NW, PATH) restated (oracle/edlib_path.hpp): returns the op count (0 when no alignment), -2 if cap is too small
long long gco_edit_path(const char* a, uint64_t na, const char* b, uint64_t nb, unsigned char* ops, uint64_t cap, long long* distance)
{
	std::vector<unsigned char> v;
	*distance = edlibPathNW(std::string(a, a + na), std::string(b, b + nb), v);
	if (v.size() > cap) return -2;
	std::copy(v.begin(), v.end(), ops);
	return (long long)v.size();
}
void gco_evalue(double minIdentity, uint64_t databaseSize, uint64_t querySize, uint64_t alignmentLength, uint64_t numEdits, double* out)
{
	EValueCalc calc(minIdentity);
	out[0] = calc.getAlignmentScore(alignmentLength, numEdits);
	out[1] = calc.getEValue(databaseSize, querySize, alignmentLength, numEdits);
}
// One extension (src/GraphAlignerBitvectorBanded.h:46-71) laid open for tests/extension_model.py: per kept slice (initial slice first, after the trim of
// ...Common.h:1231-1241) its minimum score, minimum cell and node set, and the trace cells. Returns 0, 1 = the extension failed (no slice survived), 2 = assertion.
int gco_extend(void* hv, const char* seq, uint64_t len, int bigraphNodeId, uint64_t nodeOffset)
{
	OracleHandle* h = (OracleHandle*)hv;
	for (const char* name : { "ext_slice_min", "ext_slice_minnode", "ext_slice_minoffset", "ext_slice_off", "ext_slice_nodes", "ext_trace", "ext_score" }) h->ex[name].clear();
	try {
		BitvectorAligner bv(h->o.graph, h->o.params.bandwidth);
		AlignerState state(h->o.graph);
		std::string_view sequence(seq, len);
		size_t numSlices = (len + 63) / 64;
		// the same three calls getReverseTraceFromSeed makes, with the table kept
		OnewayTrace whole = bv.getReverseTraceFromSeed(sequence, bigraphNodeId, nodeOffset, state);
		AlignerState state2(h->o.graph);
		DPTable table = bv.getViterbiSlices(sequence, bv.initialSlice(bigraphNodeId, nodeOffset), numSlices, state2);
		BitvectorAligner::removeWronglyAlignedEnd(table);
		h->ex["ext_slice_off"].push_back(0);
		for (const DPSlice& sl : table.slices) {
			h->ex["ext_slice_min"].push_back(sl.minScore);
			h->ex["ext_slice_minnode"].push_back((int64_t)sl.minScoreNode);
			h->ex["ext_slice_minoffset"].push_back((int64_t)sl.minScoreNodeOffset);
			std::vector<int64_t> nodes;
			for (const auto& item : sl.scores.items) nodes.push_back((int64_t)item.first);
			std::sort(nodes.begin(), nodes.end());
			for (int64_t x : nodes) h->ex["ext_slice_nodes"].push_back(x);
			h->ex["ext_slice_off"].push_back((int64_t)h->ex["ext_slice_nodes"].size());
		}
		if (whole.failed()) return 1;
		h->ex["ext_score"].push_back(whole.score);
		for (const TraceItem& t : whole.trace) {
			h->ex["ext_trace"].push_back((int64_t)t.DPposition.node);
			h->ex["ext_trace"].push_back((int64_t)t.DPposition.nodeOffset);
			h->ex["ext_trace"].push_back((int64_t)t.DPposition.seqPos);   // (size_t)-1 -> -1
		}
		return 0;
	} catch (const AssertionFailure& e) {
		h->error = e.what();
		return 2;
	}
}
void gco_set_tie_order(void* hv, int order) { ((OracleHandle*)hv)->tieOrder = order; }   // 0: band-entry order (the defined order), 1: reversed
void gco_set_e_cutoff(void* hv, double cutoff) { ((OracleHandle*)hv)->o.params.eCutoff = cutoff; }
uint64_t gco_edit_distance(const char* a, uint64_t na, const char* b, uint64_t nb) { return editDistanceNW(std::string(a, a + na), std::string(b, b + nb)); }
uint64_t gco_minimizer_hash(uint64_t k) { return gc::minimizerHash(k); }

} // extern "C"
