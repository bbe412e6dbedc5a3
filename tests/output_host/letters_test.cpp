// The encoders' split-node cursor (gc::GraphLetters, host/gc_output.hpp) against the two calls it stands for - GetUnitigNode + NodeSequences, what the reference's TraceItem
// constructor does per cell (src/GraphAlignerCommon.h:148-153) - on every letter of every original node of a graph, walked forwards, backwards, and at random with node changes.
// usage: letters_test graph.gfa
#include "gc_output.hpp"
#include <cstdio>
#include <random>

int main(int argc, char** argv)
{
	if (argc < 2) return 2;
	gc::GfaGraph gfa = gc::GfaGraph::LoadFromFile(argv[1]);
	gc::AlignmentGraph graph = gc::AlignmentGraph::BuildFromGFA(gfa);
	auto plain = [&](int id, size_t offset) { size_t split = graph.GetUnitigNode(id, offset); return graph.NodeSequences(split, offset - graph.NodeOffset(split)); };
	std::vector<std::pair<int, size_t>> nodes;
	graph.originalNodeSize.forEach([&](int id, size_t size) { nodes.emplace_back(id, size); });
	size_t checked = 0;
	gc::GraphLetters cursor(graph);
	for (const auto& node : nodes) {
		for (size_t o = 0; o < node.second; o++, checked++) if (cursor.at(node.first, o) != plain(node.first, o)) { printf("MISMATCH forward %d %zu\n", node.first, o); return 1; }
		for (size_t o = node.second; o-- > 0; checked++) if (cursor.at(node.first, o) != plain(node.first, o)) { printf("MISMATCH backward %d %zu\n", node.first, o); return 1; }
	}
	std::mt19937_64 rng(7);
	for (int i = 0; i < 2000000; i++, checked++) {
		const auto& node = nodes[rng() % nodes.size()];
		size_t o = rng() % node.second;
		for (int k = 0; k < 4 && o < node.second; k++, o += rng() % 70) if (cursor.at(node.first, o) != plain(node.first, o)) { printf("MISMATCH random %d %zu\n", node.first, o); return 1; }
	}
	printf("OK %zu\n", checked);
	return 0;
}
