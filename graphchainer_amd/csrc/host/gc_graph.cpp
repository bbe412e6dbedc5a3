// Host-side construction of the alignment graph ("A0" data).
// Restates the behaviour of the reference's loaders; every function cites the lines it follows.
#include "gc_graph.hpp"
#include "gc_stageclock.hpp"
#include <cstring>
#include <cstdio>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstdlib>
#include <exception>
#include <thread>
#include <algorithm>
#include <cassert>
#include <cmath>
#include <deque>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <tuple>

namespace gc {

// ------------------------------------------------------------------ small utilities

char Complement(char c)   // reference: src/CommonUtils.cpp:78-134
{
	switch (c) {
		case 'A': case 'a': return 'T';
		case 'C': case 'c': return 'G';
		case 'G': case 'g': return 'C';
		case 'T': case 't': case 'U': case 'u': return 'A';
		case 'N': case 'n': return 'N';
		case 'R': case 'r': return 'Y';
		case 'Y': case 'y': return 'R';
		case 'K': case 'k': return 'M';
		case 'M': case 'm': return 'K';
		case 'S': case 's': return 'S';
		case 'W': case 'w': return 'W';
		case 'B': case 'b': return 'V';
		case 'V': case 'v': return 'B';
		case 'D': case 'd': return 'H';
		case 'H': case 'h': return 'D';
	}
	throw std::runtime_error(std::string("Complement: invalid nucleotide '") + c + "'");
}

std::string ReverseComplement(const std::string& s)   // reference: src/CommonUtils.cpp:67-76
{
	std::string out(s.size(), 'N');
	for (size_t i = 0; i < s.size(); i++) out[i] = Complement(s[s.size() - 1 - i]);
	return out;
}


// ------------------------------------------------------------------ GFA parsing

// reference: src/GfaGraph.cpp:212-370 (LoadFromStream, called with allowVaryingOverlaps=true from
// src/Aligner.cpp:1100). Segment names map to integer ids in order of first appearance on an S or
// L line (getNameId, :146-156); numberBackToIntegers is disabled in the reference (:325-329).
GfaGraph GfaGraph::LoadFromStream(std::istream& in)
{
	GfaGraph g;
	std::unordered_map<std::string, int> ids;
	auto nameId = [&ids](const std::string& name) {
		auto it = ids.find(name);
		if (it != ids.end()) return it->second;
		int id = (int)ids.size();
		ids[name] = id;
		return id;
	};
	std::string line;
	// whitespace-separated fields of `line`, like operator>> on a stringstream (the reference's tokenizer) without building one per line:
	// the stream objects were a third of the start-up time on a 7 M-segment graph
	size_t cursor = 0;
	auto nextField = [&]() -> std::string {
		while (cursor < line.size() && (line[cursor] == ' ' || line[cursor] == '\t' || line[cursor] == '\r' || line[cursor] == '\v' || line[cursor] == '\f')) cursor++;
		size_t begin = cursor;
		while (cursor < line.size() && !(line[cursor] == ' ' || line[cursor] == '\t' || line[cursor] == '\r' || line[cursor] == '\v' || line[cursor] == '\f')) cursor++;
		return line.substr(begin, cursor - begin);
	};
	while (in.good()) {
		std::getline(in, line);
		if (!in.good()) break;   // like the reference (:219-223), a last line without '\n' is dropped
		if (line.empty()) continue;
		cursor = 0;
		if (line[0] == 'S') {
			std::string tag = nextField(), name = nextField();
			int id = nameId(name);
			std::string seq = nextField();
			if (seq == "*") throw std::runtime_error("Nodes without sequence (*) are not currently supported (nodeid " + name + ")");
			if (seq.empty()) throw std::runtime_error("empty S line for node " + name);
			g.nodes[id] = std::move(seq);
		} else if (line[0] == 'L') {
			std::string tag = nextField(), from = nextField();
			int fromId = nameId(from);
			std::string fromOri = nextField(), to = nextField();
			int toId = nameId(to);
			std::string toOri = nextField(), cigar = nextField();
			int overlap = 0;
			char unit = 'M';
			{   // "<int><unit>" as `ss >> overlap >> unit` reads it (a missing or malformed number leaves 0, as a failed extraction does)
				size_t at = 0;
				bool negative = false;
				if (at < cigar.size() && (cigar[at] == '-' || cigar[at] == '+')) { negative = cigar[at] == '-'; at++; }
				long long value = 0;
				bool digits = false;
				while (at < cigar.size() && cigar[at] >= '0' && cigar[at] <= '9') { value = value * 10 + (cigar[at] - '0'); if (value > 2000000000ll) value = 2000000000ll; at++; digits = true; }
				if (digits) { overlap = (int)(negative ? -value : value); if (at < cigar.size()) unit = cigar[at]; }
			}
			(void)unit;
			if ((fromOri != "+" && fromOri != "-") || (toOri != "+" && toOri != "-")) throw std::runtime_error("bad L line orientation: " + line);
			if (overlap < 0) throw std::runtime_error("Edge overlap cannot be negative. Fix the graph");
			NodePos a { fromId, fromOri == "+" };
			NodePos b { toId, toOri == "+" };
			g.edges[a].push_back(b);
			g.overlaps.push_back({ { a, b }, (size_t)overlap });
		}
	}
	for (auto& kv : ids) g.originalNodeName[kv.second] = kv.first;
	// edges that touch a segment without an S line are dropped (reference: src/GfaGraph.cpp:329-368)
	std::vector<NodePos> orphanSources;
	for (auto& e : g.edges) {
		if (g.nodes.count(e.first.id) == 0) { orphanSources.push_back(e.first); continue; }
		for (size_t i = e.second.size(); i-- > 0;)
			if (g.nodes.count(e.second[i].id) == 0) e.second.erase(e.second.begin() + i);
	}
	for (auto& p : orphanSources) g.edges.erase(p);
	return g;
}

GfaGraph GfaGraph::LoadFromFile(const std::string& path)
{
	std::ifstream f(path);
	if (!f.good()) throw std::runtime_error("cannot open GFA file " + path);
	auto t0 = std::chrono::steady_clock::now();
	GfaGraph g = LoadFromStream(f);
	if (getenv("GC_DEBUG_TIMES")) fprintf(stderr, "[gc build] %-28s %8.1f ms\n", "GFA parse", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
	return g;
}

// ------------------------------------------------------------------ bigraph -> split-node digraph

static bool allowedNucleotide(char c)   // reference: src/BigraphToDigraph.cpp:9-47
{
	switch (c) {
		case 'a': case 'A': case 'c': case 'C': case 'g': case 'G': case 't': case 'T': case 'u': case 'U':
		case 'y': case 'Y': case 'r': case 'R': case 'w': case 'W': case 's': case 'S': case 'k': case 'K':
		case 'm': case 'M': case 'd': case 'D': case 'v': case 'V': case 'h': case 'H': case 'b': case 'B':
		case 'n': case 'N': return true;
	}
	return false;
}

// reference: src/BigraphToDigraph.cpp:215-267. GFA segment n becomes forward node 2n and
// reverse-complement node 2n+1 (:101-104); link a(+/-) -> b(+/-) becomes two directed edges (:106-132).
AlignmentGraph AlignmentGraph::BuildFromGFA(const GfaGraph& gfa)
{
	StageClock clock;
	AlignmentGraph g;
	for (auto& ov : gfa.overlaps)
		if (ov.second != 0) throw std::runtime_error("edge overlaps other than 0M are outside this build's scope (variation-graph DAGs only)");
	{
		int maxId = -1;
		for (const auto& node : gfa.nodes) maxId = std::max(maxId, node.first);
		const size_t ids = 2 * ((size_t)(maxId + 1));   // (the id-keyed tables are vectors: sized up front, the ids come in hash order)
		g.nodeLookup.denseUpTo(ids); g.originalNodeSize.denseUpTo(ids); g.originalNodeName.denseUpTo(ids);
	}
	for (const auto& node : gfa.nodes) {    // libstdc++ unordered_map order == reference order
		for (char c : node.second)
			if (!allowedNucleotide(c)) throw std::runtime_error(std::string("Invalid sequence character: ") + c);
		auto nameIt = gfa.originalNodeName.find(node.first);
		std::string name = nameIt == gfa.originalNodeName.end() ? std::string() : nameIt->second;
		std::vector<size_t> breakpoints { 0, node.second.size() };
		g.AddNode(node.first * 2, node.second, name, false, breakpoints);
		g.AddNode(node.first * 2 + 1, ReverseComplement(node.second), name, true, breakpoints);
	}
	clock.lap("split nodes");
	for (const auto& edge : gfa.edges) {
		for (const auto& target : edge.second) {
			int from = edge.first.id, to = target.id;
			// ConvertGFAEdgeToEdges, src/BigraphToDigraph.cpp:106-132
			size_t fromLeft, fromRight, toLeft, toRight;
			if (!edge.first.end) { fromLeft = from * 2; fromRight = from * 2 + 1; } else { fromLeft = from * 2 + 1; fromRight = from * 2; }
			if (!target.end) { toLeft = to * 2; toRight = to * 2 + 1; } else { toLeft = to * 2 + 1; toRight = to * 2; }
			g.AddEdgeNodeId((int)fromRight, (int)toRight, 0);
			g.AddEdgeNodeId((int)toLeft, (int)fromLeft, 0);
		}
	}
	clock.lap("edges");
	// the reference's container into the product's flat storage, its iteration order recorded (what MinimizerIndex::Build follows)
	for (const auto& kv : g.buildLookup) {
		g.nodeLookupOrder.push_back(kv.first);
		g.nodeLookup.add(kv.first, kv.second.data(), kv.second.size());
	}
	g.buildLookup = std::unordered_map<int, std::vector<size_t>>();
	g.Finalize();
	return g;
}

// reference: src/AlignmentGraph.cpp:51-85. Cuts a bigraph node into <=64 bp pieces, chained by edges.
void AlignmentGraph::AddNode(int nodeId, const std::string& sequence, const std::string& name, bool reverseNode, const std::vector<size_t>& breakpoints)
{
	if (buildLookup.count(nodeId) != 0) return;
	originalNodeSize[nodeId] = sequence.size();
	originalNodeName[nodeId] = name;
	for (size_t b = 1; b < breakpoints.size(); b++) {
		if (breakpoints[b] == breakpoints[b - 1]) continue;
		for (size_t offset = breakpoints[b - 1]; offset < breakpoints[b]; offset += SPLIT_NODE_SIZE) {
			size_t size = std::min((size_t)SPLIT_NODE_SIZE, breakpoints[b] - offset);
			AddSplitNode(nodeId, (int)offset, sequence.substr(offset, size), reverseNode);
			if (offset > 0) {
				size_t last = outNeighbors.size() - 1;
				outNeighbors[last - 1].push_back(last);
				inNeighbors[last].push_back(last - 1);
			}
		}
	}
}

// reference: src/AlignmentGraph.cpp:87-231. Sequence packing: 2 bits per base, base i in word i/32 at
// bit (i%32)*2, A=0 C=1 G=2 T/U=3; IUPAC codes make the node "ambiguous" (one-hot words instead).
void AlignmentGraph::AddSplitNode(int nodeId, int offset, const std::string& sequence, bool reverseNode)
{
	assert(sequence.size() <= (size_t)SPLIT_NODE_SIZE);
	bpSize += sequence.size();
	buildLookup[nodeId].push_back(nodeLength.size());
	nodeLength.push_back(sequence.size());
	nodeIDs.push_back(nodeId);
	inNeighbors.emplace_back();
	outNeighbors.emplace_back();
	reverse.push_back(reverseNode);
	nodeOffset.push_back(offset);
	std::array<uint64_t, 2> packed { 0, 0 };
	AmbiguousSeq amb { 0, 0, 0, 0 };
	bool ambiguous = false;
	for (size_t i = 0; i < sequence.size(); i++) {
		uint64_t bit = (uint64_t)1 << i;
		int code = -1;
		bool a = false, c = false, gg = false, t = false;
		switch (sequence[i]) {
			case 'a': case 'A': a = true; code = 0; break;
			case 'c': case 'C': c = true; code = 1; break;
			case 'g': case 'G': gg = true; code = 2; break;
			case 't': case 'T': case 'u': case 'U': t = true; code = 3; break;
			case 'r': case 'R': a = gg = true; break;
			case 'y': case 'Y': c = t = true; break;
			case 's': case 'S': gg = c = true; break;
			case 'w': case 'W': a = t = true; break;
			case 'k': case 'K': gg = t = true; break;
			case 'm': case 'M': a = c = true; break;
			case 'b': case 'B': c = gg = t = true; break;
			case 'd': case 'D': a = gg = t = true; break;
			case 'h': case 'H': a = c = t = true; break;
			case 'v': case 'V': a = c = gg = true; break;
			case 'n': case 'N': a = c = gg = t = true; break;
			default: throw std::runtime_error("invalid nucleotide in graph");
		}
		if (a) amb.A |= bit;
		if (c) amb.C |= bit;
		if (gg) amb.G |= bit;
		if (t) amb.T |= bit;
		if (code >= 0) packed[i / BP_IN_CHUNK] |= (uint64_t)code << ((i % BP_IN_CHUNK) * 2);
		else ambiguous = true;
	}
	ambiguousNodes.push_back(ambiguous);
	if (ambiguous) ambiguousNodeSequences.push_back(amb);
	else nodeSequences.push_back(packed);
}

// reference: src/AlignmentGraph.cpp:233-253
void AlignmentGraph::AddEdgeNodeId(int fromId, int toId, size_t startOffset)
{
	size_t from = buildLookup.at(fromId).back();
	size_t to = SIZE_MAX;
	for (size_t node : buildLookup.at(toId))
		if (nodeOffset[node] == startOffset) to = node;
	if (to == SIZE_MAX) throw std::runtime_error("edge target offset not found");
	if (std::find(inNeighbors[to].begin(), inNeighbors[to].end(), from) == inNeighbors[to].end()) inNeighbors[to].push_back(from);
	if (std::find(outNeighbors[from].begin(), outNeighbors[from].end(), to) == outNeighbors[from].end()) outNeighbors[from].push_back(to);
}

// reference: src/AlignmentGraph.cpp:255-307
void AlignmentGraph::Finalize()
{
	StageClock clock;
	RenumberAmbiguousToEnd();
	ambiguousNodes.clear();
	clock.lap("renumber ambiguous");
	findLinearizable();
	clock.lap("linearizable");
	doComponentOrder();
	clock.lap("component order (Tarjan)");
	findChains();
	clock.lap("chains");
	finalized = true;
}

// reference: src/AlignmentGraph.cpp:919-974 (+ the two helpers :885-917). Non-ambiguous nodes keep
// their relative order; ambiguous node j (in creation order) moves to index N-1-j.
void AlignmentGraph::RenumberAmbiguousToEnd()
{
	size_t n = ambiguousNodes.size();
	std::vector<size_t> newIndex(n);
	size_t plain = 0, amb = 0;
	for (size_t i = 0; i < n; i++) {
		if (!ambiguousNodes[i]) newIndex[i] = plain++;
		else newIndex[i] = n - 1 - amb++;
	}
	firstAmbiguous = plain;
	if (amb == 0) return;
	std::reverse(ambiguousNodeSequences.begin(), ambiguousNodeSequences.end());
	auto permute = [&](auto& vec) {
		auto copy = vec;
		for (size_t i = 0; i < n; i++) vec[newIndex[i]] = copy[i];
	};
	permute(nodeLength);
	permute(nodeOffset);
	permute(nodeIDs);
	permute(inNeighbors);
	permute(outNeighbors);
	permute(reverse);
	for (size_t& v : nodeLookup.nodes()) v = newIndex[v];
	for (size_t i = 0; i < n; i++) {
		for (auto& v : inNeighbors[i]) v = newIndex[v];
		for (auto& v : outNeighbors[i]) v = newIndex[v];
	}
}

// reference: src/AlignmentGraph.cpp:644-736. Restated literally, including the fact that the start
// node is marked checked before the walk begins (:665), which makes the walk stop on its first
// iteration (:689-701) for every in-degree-1 node: on any graph the result is all-false. The walk is
// kept (instead of a constant) so a change in the reference's rule shows up as a diff here.
void AlignmentGraph::findLinearizable()
{
	size_t n = nodeLength.size();
	linearizable.assign(n, false);
	std::vector<bool> checked(n, false), onStack(n, false);
	std::vector<size_t> stack;
	for (size_t node = 0; node < n; node++) {
		if (checked[node]) continue;
		checked[node] = true;
		if (inNeighbors[node].size() != 1) continue;
		stack.assign(1, node);
		onStack[node] = true;
		while (true) {
			size_t top = stack.back();
			if (inNeighbors[top].size() != 1 || checked[top]) {
				for (size_t i = 0; i + 1 < stack.size(); i++) { checked[stack[i]] = true; linearizable[stack[i]] = true; onStack[stack[i]] = false; }
				linearizable[top] = false; checked[top] = true; onStack[top] = false;
				break;
			}
			size_t neighbor = inNeighbors[top][0];
			if (neighbor == node || onStack[neighbor]) {
				// only reachable on cyclic graphs, which the MPC build rejects (src/AlignmentGraph.cpp:1298)
				for (size_t v : stack) { checked[v] = true; linearizable[v] = false; onStack[v] = false; }
				break;
			}
			stack.push_back(neighbor);
			onStack[neighbor] = true;
		}
		stack.clear();
	}
}

// reference: src/AlignmentGraph.cpp:1008-1115. Tarjan SCC with roots in ascending node order and
// children in outNeighbors order; components are numbered in completion order and then reversed so
// that componentNumber is a topological rank (every edge goes to an equal-or-higher number).
void AlignmentGraph::doComponentOrder()
{
	size_t n = nodeLength.size();
	const size_t NONE = SIZE_MAX;
	std::vector<size_t> index(n, NONE), lowlink(n, NONE), sccStack;
	std::vector<bool> onStack(n, false);
	std::vector<std::pair<size_t, size_t>> dfs;   // (node, next child position)
	componentNumber.assign(n, NONE);
	size_t counter = 0, nextComponent = 0;
	for (size_t root = 0; root < n; root++) {
		if (index[root] != NONE) continue;
		dfs.emplace_back(root, 0);
		index[root] = lowlink[root] = counter++;
		sccStack.push_back(root);
		onStack[root] = true;
		while (!dfs.empty()) {
			size_t v = dfs.back().first;
			if (dfs.back().second < outNeighbors[v].size()) {
				size_t w = outNeighbors[v][dfs.back().second++];
				if (index[w] == NONE) {
					index[w] = lowlink[w] = counter++;
					sccStack.push_back(w);
					onStack[w] = true;
					dfs.emplace_back(w, 0);
				} else if (onStack[w]) {
					lowlink[v] = std::min(lowlink[v], index[w]);
				}
				continue;
			}
			dfs.pop_back();
			if (!dfs.empty()) lowlink[dfs.back().first] = std::min(lowlink[dfs.back().first], lowlink[v]);
			if (lowlink[v] == index[v]) {
				size_t w;
				do {
					w = sccStack.back();
					sccStack.pop_back();
					onStack[w] = false;
					componentNumber[w] = nextComponent;
				} while (w != v);
				nextComponent++;
			}
		}
	}
	for (size_t i = 0; i < n; i++) componentNumber[i] = nextComponent - 1 - componentNumber[i];
}

// reference: src/AlignmentGraph.cpp:583-642 (findChains) with chainTips :433-535, chainCycles :537-581,
// chainBubble :378-405. On an acyclic graph every node is both a "forward tip" and a "backward tip"
// (induction over the topological order in chainTips :445-462 and :473-490), so chainTips unions
// every edge and each weakly connected component becomes one chain; all later unions stay inside a
// component. Only the partition is consumed downstream (seed clustering, src/GraphAligner.h:261, and
// fixChainApproxPos), never the representative, so the chain label here is the smallest node index of
// the component. Cyclic graphs never reach alignment (rejected at src/AlignmentGraph.cpp:1298-1302).
void AlignmentGraph::findChains()
{
	size_t n = nodeLength.size();
	chainNumber.assign(n, SIZE_MAX);
	std::vector<size_t> queue;
	for (size_t s = 0; s < n; s++) {
		if (chainNumber[s] != SIZE_MAX) continue;
		chainNumber[s] = s;
		queue.assign(1, s);
		for (size_t qi = 0; qi < queue.size(); qi++) {
			size_t v = queue[qi];
			for (size_t u : outNeighbors[v]) if (chainNumber[u] == SIZE_MAX) { chainNumber[u] = s; queue.push_back(u); }
			for (size_t u : inNeighbors[v]) if (chainNumber[u] == SIZE_MAX) { chainNumber[u] = s; queue.push_back(u); }
		}
	}
	chainApproxPos.assign(n, SIZE_MAX);
	for (size_t i = 0; i < n; i++)
		if (chainApproxPos[i] == SIZE_MAX) fixChainApproxPos(i);
}

// reference: src/AlignmentGraph.cpp:407-431. Depth-first labelling with an explicit LIFO stack;
// the first value that reaches a node wins, so the push order (out-neighbours, then in-neighbours,
// each in adjacency order) is part of the result.
void AlignmentGraph::fixChainApproxPos(size_t start)
{
	std::vector<std::pair<size_t, size_t>> stack;
	size_t chain = chainNumber[start];
	stack.emplace_back(start, (nodeLength.size() + 5) * SPLIT_NODE_SIZE);
	while (!stack.empty()) {
		auto [v, dist] = stack.back();
		stack.pop_back();
		if (chainApproxPos[v] != SIZE_MAX) continue;
		chainApproxPos[v] = dist;
		for (size_t u : outNeighbors[v]) {
			if (chainNumber[u] != chain || chainApproxPos[u] != SIZE_MAX) continue;
			stack.emplace_back(u, dist + nodeLength[u]);
		}
		for (size_t u : inNeighbors[v]) {
			if (chainNumber[u] != chain || chainApproxPos[u] != SIZE_MAX) continue;
			stack.emplace_back(u, dist - nodeLength[v]);
		}
	}
}

// ------------------------------------------------------------------ accessors

char AlignmentGraph::NodeSequences(size_t node, size_t pos) const   // reference: src/AlignmentGraph.cpp:751-796
{
	if (node < firstAmbiguous) return "ACGT"[(nodeSequences[node][pos / BP_IN_CHUNK] >> ((pos % BP_IN_CHUNK) * 2)) & 3];
	const AmbiguousSeq& s = ambiguousNodeSequences[node - firstAmbiguous];
	int mask = (int)((s.A >> pos) & 1) | (int)(((s.C >> pos) & 1) << 1) | (int)(((s.G >> pos) & 1) << 2) | (int)(((s.T >> pos) & 1) << 3);
	// index = A | C<<1 | G<<2 | T<<3
	static const char iupac[16] = { 'N', 'A', 'C', 'M', 'G', 'R', 'S', 'V', 'T', 'W', 'Y', 'H', 'K', 'D', 'B', 'N' };
	return iupac[mask];
}

size_t AlignmentGraph::GetUnitigNode(int nodeId, size_t offset) const   // reference: src/AlignmentGraph.cpp:832-848
{
	const NodeLookup::Span nodes = nodeLookup.at(nodeId);
	size_t index = (size_t)(nodes.size() * ((double)offset / (double)originalNodeSize.at(nodeId)));
	if (index >= nodes.size()) index = nodes.size() - 1;
	while (index < nodes.size() - 1 && nodeOffset[nodes[index]] + nodeLength[nodes[index]] <= offset) index++;
	while (index > 0 && nodeOffset[nodes[index]] > offset) index--;
	return nodes[index];
}

std::pair<int, size_t> AlignmentGraph::GetReversePosition(int nodeId, size_t offset) const   // :850-868
{
	size_t originalSize = originalNodeSize.at(nodeId);
	return { nodeId ^ 1, originalSize - offset - 1 };
}

std::string AlignmentGraph::OriginalNodeName(int nodeId) const
{
	const std::string* name = originalNodeName.find(nodeId);
	return name ? *name : std::string();
}

// ------------------------------------------------------------------ MPC index

// reference: src/AlignmentGraph.cpp:1430-1463. Weakly connected components by BFS from ascending
// seeds, expanding out-neighbours then in-neighbours; component ids and the in-component index of
// each node (its BFS position) follow from that order.
void AlignmentGraph::buildComponentsMap()
{
	size_t n = NodeSize();
	component_map.assign(n, n + 1);
	component_idx.assign(n, n + 1);
	component_ids.clear();
	std::vector<size_t> q;
	for (size_t s = 0; s < n; s++) {
		if (component_map[s] != n + 1) continue;
		size_t c = component_ids.size();
		q.assign(1, s);
		component_map[s] = c;
		component_idx[s] = 0;
		for (size_t i = 0; i < q.size(); i++) {
			size_t v = q[i];
			for (size_t t : outNeighbors[v]) if (component_map[t] == n + 1) { component_map[t] = c; component_idx[t] = q.size(); q.push_back(t); }
			for (size_t t : inNeighbors[v]) if (component_map[t] == n + 1) { component_map[t] = c; component_idx[t] = q.size(); q.push_back(t); }
		}
		component_ids.push_back(q);
	}
}

// reference: src/AlignmentGraph.cpp:1267-1326. Repeatedly takes the source-to-sink path with the most
// still-uncovered nodes (ties towards the larger (count, predecessor) pair), trimmed of covered ends.
std::vector<std::vector<size_t>> AlignmentGraph::greedyCover(size_t cid) const
{
	const std::vector<size_t>& cids = component_ids[cid];
	size_t n = cids.size();
	std::vector<std::vector<size_t>> cover;
	std::vector<size_t> coveredTimes(n, 0), indeg(n), order(n);
	std::vector<std::pair<size_t, size_t>> best(n);   // (uncovered nodes on best path ending here, predecessor)
	size_t coveredCount = 0;
	while (coveredCount < n) {
		size_t qn = 0;
		for (size_t i = 0; i < n; i++) {
			best[i] = { 0, i };
			indeg[i] = inNeighbors[cids[i]].size();
			if (indeg[i] == 0) order[qn++] = i;
		}
		std::pair<size_t, size_t> top { 0, 0 };
		for (size_t qi = 0; qi < qn; qi++) {
			size_t s = order[qi];
			if (coveredTimes[s] == 0) best[s].first++;
			top = std::max(top, std::make_pair(best[s].first, s));
			for (size_t tid : outNeighbors[cids[s]]) {
				size_t t = component_idx[tid];
				best[t] = std::max(best[t], std::make_pair(best[s].first, s));
				if (--indeg[t] == 0) order[qn++] = t;
			}
		}
		if (qn < n) throw std::runtime_error("The input sequence graph has a directed cycle. The current version of GraphChainer only supports DAGs.");
		std::vector<size_t> walk;
		for (size_t i = top.second;; i = best[i].second) {
			walk.push_back(i);
			if (best[i].second == i) break;
		}
		std::reverse(walk.begin(), walk.end());
		size_t l = 0, r = walk.size() - 1;
		while (coveredTimes[walk[l]]) l++;
		while (coveredTimes[walk[r]]) r--;
		std::vector<size_t> path;
		for (size_t i = l; i <= r; i++) {
			path.push_back(cids[walk[i]]);
			if (coveredTimes[walk[i]]++ == 0) coveredCount++;
		}
		cover.push_back(std::move(path));
	}
	return cover;
}

// reference: src/AlignmentGraph.cpp:1157-1265. Reduces a path cover to minimum width: min-flow with
// lower bound 1 on every node, solved by cancelling flow along augmenting paths in the residual graph
// of the cover, then decomposing the remaining flow into paths. The resulting cover may differ from
// the reference's in which minimum cover is picked; chaining results do not depend on that choice
// (any valid cover yields the same reachability).
std::vector<std::vector<size_t>> AlignmentGraph::shrink(size_t cid, const std::vector<std::vector<size_t>>& pc)
{
	typedef long long LL;
	const std::vector<size_t>& cids = component_ids[cid];
	LL n = (LL)cids.size();
	LL width = (LL)pc.size(), inf = (LL)pc.size();
	std::vector<LL> nodeFlow(n, 0), startFlow(n, 0), endFlow(n, 0);
	std::map<std::pair<LL, LL>, LL> edgeFlow;
	for (const auto& path : pc) {
		for (size_t i = 0; i < path.size(); i++) {
			nodeFlow[component_idx[path[i]]]++;
			if (i > 0) edgeFlow[{ (LL)component_idx[path[i - 1]], (LL)component_idx[path[i]] }]++;
		}
		startFlow[component_idx[path[0]]]++;
		endFlow[component_idx[path.back()]]++;
	}
	// residual network: vertex i_in = i, i_out = i + n, S = 2n, T = 2n+1; arcs stored in pairs (e, e^1)
	LL V = 2 * n + 2, S = 2 * n, T = 2 * n + 1;
	std::vector<LL> head(V, 0), to(2, 0), next(2, 0), cap(2, 0);
	auto addArc = [&](LL a, LL b, LL c) { to.push_back(b); next.push_back(head[a]); cap.push_back(c); head[a] = (LL)to.size() - 1; };
	// arc e: capacity to *decrease* flow on (a,b) down to its lower bound; arc e^1: capacity to increase it
	auto addEdge = [&](LL a, LL b, LL upper, LL lower, LL flow) { addArc(a, b, flow - lower); addArc(b, a, upper - flow); };
	for (LL i = 0; i < n; i++)
		for (size_t jid : outNeighbors[cids[i]]) {
			LL j = (LL)component_idx[jid];
			auto it = edgeFlow.find({ i, j });
			addEdge(i + n, j, inf, 0, it == edgeFlow.end() ? 0 : it->second);
		}
	for (LL i = 0; i < n; i++) {
		addEdge(i, i + n, inf, 1, nodeFlow[i]);
		addEdge(S, i, inf, 0, startFlow[i]);
		addEdge(i + n, T, inf, 0, endFlow[i]);
	}
	std::vector<LL> queue(V), via(V);
	std::vector<char> seen(V);
	while (true) {
		std::fill(seen.begin(), seen.end(), 0);
		std::fill(via.begin(), via.end(), -1);
		LL qn = 0;
		queue[qn++] = S;
		seen[S] = 1;
		for (LL qi = 0; qi < qn && !seen[T]; qi++) {
			LL v = queue[qi];
			for (LL e = head[v]; e; e = next[e])
				if (cap[e] > 0 && !seen[to[e]]) { seen[to[e]] = 1; via[to[e]] = e; queue[qn++] = to[e]; }
		}
		if (!seen[T]) break;
		LL push = cap[via[T]];
		for (LL v = T; via[v] != -1; v = to[via[v] ^ 1]) push = std::min(push, cap[via[v]]);
		for (LL v = T; via[v] != -1; v = to[via[v] ^ 1]) { cap[via[v]] -= push; cap[via[v] ^ 1] += push; }
		if (push == 0) throw std::runtime_error("path cover shrink: zero augmentation");
		width -= push;
	}
	std::vector<std::vector<size_t>> out;
	for (LL it = 0; it < width; it++) {
		std::vector<size_t> path;
		for (LL v = S; v != T;) {
			if (v >= 0 && v < n) path.push_back(cids[v]);
			LL nxt = -1;
			for (LL e = head[v]; e; e = next[e]) {
				if (e & 1) continue;
				LL remaining = cap[e] + ((v < n && to[e] == v + n) ? 1 : 0);   // remaining flow = residual + lower bound
				if (remaining > 0) { nxt = to[e]; cap[e]--; break; }
			}
			if (nxt == -1) throw std::runtime_error("path cover shrink: flow decomposition failed");
			v = nxt;
		}
		out.push_back(std::move(path));
	}
	return out;
}

// reference: src/AlignmentGraph.cpp:1328-1401. For every node v and path k: backwards[v] holds
// (u, k) where u is the LAST node of path k that strictly reaches v; paths[v] lists the paths through v;
// topo_ids is a Kahn order of the component (queue seeded with sources in in-component index order).
void AlignmentGraph::computeMPCIndex(size_t cid, const std::vector<std::vector<size_t>>& pc)
{
	typedef long long LL;
	const std::vector<size_t>& cids = component_ids[cid];
	size_t n = cids.size();
	size_t K = pc.size();
	backwards[cid].assign(n, {});
	paths[cid].assign(n, {});
	std::vector<LL> last2reach(n * K, -1);
	for (size_t k = 0; k < K; k++)
		for (size_t j = 0; j < pc[k].size(); j++) {
			size_t x = component_idx[pc[k][j]];
			last2reach[x * K + k] = (LL)j;
			paths[cid][x].push_back(k);
		}
	std::vector<size_t> indeg(n), order;
	order.reserve(n);
	for (size_t i = 0; i < n; i++) {
		indeg[i] = inNeighbors[cids[i]].size();
		if (indeg[i] == 0) order.push_back(i);
	}
	topo_ids[cid].assign(n, 0);
	topo[cid].clear();
	for (size_t qi = 0; qi < order.size(); qi++) {
		size_t s = order[qi];
		for (size_t tid : outNeighbors[cids[s]]) {
			size_t t = component_idx[tid];
			if (--indeg[t] == 0) order.push_back(t);
		}
		topo_ids[cid][s] = topo[cid].size();
		topo[cid].push_back(s);
	}
	for (size_t i : order)
		for (size_t jid : outNeighbors[cids[i]]) {
			size_t j = component_idx[jid];
			for (size_t k = 0; k < K; k++) last2reach[j * K + k] = std::max(last2reach[j * K + k], last2reach[i * K + k]);
		}
	for (size_t i = 0; i < n; i++)
		for (size_t k = 0; k < K; k++) {
			LL idx = last2reach[i * K + k];
			if (idx != -1 && component_idx[pc[k][idx]] == i) idx--;
			if (idx != -1) backwards[cid][i].push_back({ component_idx[pc[k][idx]], k });
		}
}

// CPUs' worth of bandwidth the cgroup grants this process (cgroup v2 cpu.max, v1 cfs quota / period); 0 = no limit known.
// A container may show all hardware threads of the box and be throttled once a burst of workers has spent the period's quota.
double cpuQuota()
{
	auto readNumbers = [](const char* path, double& a, double& b) -> int {
		FILE* f = fopen(path, "r");
		if (!f) return 0;
		char first[64] = { 0 };
		int got = fscanf(f, "%63s %lf", first, &b);
		fclose(f);
		if (got < 1 || !strcmp(first, "max")) return -1;
		a = atof(first);
		return got;
	};
	double quota = 0, period = 0;
	int got = readNumbers("/sys/fs/cgroup/cpu.max", quota, period);
	if (got == 2 && quota > 0 && period > 0) return quota / period;
	if (got == 0) {
		double q = 0, p = 0, unused = 0;
		if (readNumbers("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", q, unused) >= 1 && q > 0 && readNumbers("/sys/fs/cgroup/cpu/cpu.cfs_period_us", p, unused) >= 1 && p > 0) return q / p;
	}
	return 0;
}

size_t buildThreads()
{
	if (const char* env = getenv("GC_BUILD_THREADS")) { long v = atol(env); if (v >= 1) return (size_t)v; }
	size_t n = std::max(1u, std::thread::hardware_concurrency());
	const double quota = cpuQuota();
	if (quota > 0) n = std::min<size_t>(n, std::max<size_t>(1, (size_t)(quota + 0.5)));
	return n;
}

void AlignmentGraph::buildMPC(bool shrinkToMinimum)   // reference: src/AlignmentGraph.cpp:1465-1489
{
	buildComponentsMap();
	size_t C = component_ids.size();
	mpc.assign(C, {});
	topo.assign(C, {});
	topo_ids.assign(C, {});
	paths.assign(C, {});
	backwards.assign(C, {});
	// Components are independent (forward and reverse strand of a connected graph are two) and every step below touches
	// only its own component's slots, so they are built side by side; the result does not depend on the schedule.
	auto buildComponent = [&](size_t cid) {
		StageClock clock;
		mpc[cid] = greedyCover(cid);
		clock.lap("greedy path cover");
		if (shrinkToMinimum) mpc[cid] = shrink(cid, mpc[cid]);
		clock.lap("shrink (max flow)");
		computeMPCIndex(cid, mpc[cid]);
		clock.lap("MPC index");
	};
	size_t workers = std::min<size_t>(C, buildThreads());
	if (workers <= 1) {
		for (size_t cid = 0; cid < C; cid++) buildComponent(cid);
		return;
	}
	std::atomic<size_t> next { 0 };
	std::vector<std::exception_ptr> errors(workers);
	std::vector<std::thread> threads;
	for (size_t t = 0; t < workers; t++)
		threads.emplace_back([&, t]() {
			try {
				for (size_t cid = next++; cid < C; cid = next++) buildComponent(cid);
			} catch (...) {
				errors[t] = std::current_exception();
			}
		});
	for (auto& th : threads) th.join();
	for (auto& e : errors) if (e) std::rethrow_exception(e);
}

// reference: src/AlignmentGraph.cpp:1866-1916. Unweighted BFS (fewest hops) from S until T is seen;
// nodes farther than sepLimit bp from S are not expanded. A negative sepLimit compares as a huge
// unsigned value (size_t > long long promotion, :1897), i.e. no limit - kept on purpose.
std::vector<size_t> AlignmentGraph::getChainPath(size_t S, size_t T, long long sepLimit) const
{
	// The reference marks visited nodes in arrays as long as the graph (src/AlignmentGraph.cpp:1866-1916). Here (r4) they live in a per-thread open-addressing table that
	// grows with the search - three graph-sized arrays per host thread are 3.5 GB each on a 1 Gbp graph - and the search leaves out what cannot reach T: componentNumber
	// never decreases along an edge (it is the topological rank of the node's strongly connected component, :1008), so a node ranked above T has no path to T, nor has
	// anything first discovered through it. The queue order, distances and predecessors of all other nodes are unchanged (k_stitch prunes the same way).
	struct Seen { size_t node, dis; uint32_t pre, generation; };   // pre: queue index of the node it was first reached from
	static thread_local std::vector<Seen> table;
	static thread_local std::vector<uint32_t> queueSlot;           // queue position -> table slot
	static thread_local uint32_t generation = 0;
	std::vector<size_t> out;
	const size_t rankT = componentNumber.empty() ? SIZE_MAX : componentNumber[T];
	if (!componentNumber.empty() && componentNumber[S] > rankT) return out;
	if (table.empty()) table.assign(1024, Seen { 0, 0, 0, 0 });
	if (++generation == 0) { for (Seen& e : table) e.generation = 0; generation = 1; }
	queueSlot.clear();
	size_t mask = table.size() - 1;
	auto slotOf = [&](size_t node) -> size_t {   // the slot that holds `node`, or the empty slot where it goes
		for (size_t h = (node * 0x9E3779B97F4A7C15ull >> 20) & mask;; h = (h + 1) & mask) if (table[h].generation != generation || table[h].node == node) return h;
	};
	auto grow = [&]() {
		std::vector<Seen> old(table.size() * 2, Seen { 0, 0, 0, 0 });
		old.swap(table);
		mask = table.size() - 1;
		for (uint32_t& q : queueSlot) { const Seen e = old[q]; const size_t h = slotOf(e.node); table[h] = e; q = (uint32_t)h; }
	};
	auto add = [&](size_t node, size_t dis, uint32_t pre) {
		if (2 * (queueSlot.size() + 1) > table.size()) grow();
		const size_t h = slotOf(node);
		table[h] = Seen { node, dis, pre, generation };
		queueSlot.push_back((uint32_t)h);
	};
	add(S, 0, 0);
	bool found = S == T;
	for (size_t i = 0; !found && i < queueSlot.size(); i++) {
		const Seen s = table[queueSlot[i]];
		if (s.dis > (size_t)sepLimit) continue;
		for (size_t t : outNeighbors[s.node]) {
			if (!componentNumber.empty() && componentNumber[t] > rankT) continue;
			const size_t h = slotOf(t);
			if (table[h].generation == generation) continue;   // seen before
			add(t, s.dis + nodeLength[t], (uint32_t)i);
			if (t == T) { found = true; break; }                // (the reference's loop ends once T has been discovered; what it still adds from this node's list changes nothing)
		}
	}
	if (!found) return out;
	for (size_t q = queueSlot.size() - 1;;) {   // T is the last entry
		const Seen& e = table[queueSlot[q]];
		out.push_back(e.node);
		if (e.node == S) break;
		q = e.pre;
	}
	std::reverse(out.begin(), out.end());
	return out;
}

} // namespace gc
