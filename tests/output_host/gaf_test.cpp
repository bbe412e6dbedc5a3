// The product's GAF encoder (gc::formatGafLine, csrc/host/gc_output.cpp) on the CPU: formats the alignments of a dump file
//   (per alignment: "name merge n score start end" / the read / n rows "node offset seqPos nodeSwitch") against a graph and prints one GAF line each
//   (merge 0 / 1: the two CIGAR styles) or, with merge 2, the protobuf-JSON line of gc::buildVgAlignment + gc::vgToJson.
// tests/test_host_logic.py feeds it the oracle's whole-read alignments and compares the text with the oracle's own encoder.
// usage: gaf_test graph.gfa alignments.txt
#include "gc_output.hpp"
#include <cstdio>
#include <fstream>
#include <iostream>

int main(int argc, char** argv)
{
	if (argc < 3) return 2;
	gc::GfaGraph gfa = gc::GfaGraph::LoadFromFile(argv[1]);
	gc::AlignmentGraph graph = gc::AlignmentGraph::BuildFromGFA(gfa);
	std::ifstream in(argv[2]);
	std::string name, read;
	int merge = 0;
	uint64_t n = 0;
	long long score = 0, start = 0, end = 0;
	while (in >> name >> merge >> n >> score >> start >> end >> read) {
		std::vector<int32_t> node(n);
		std::vector<uint32_t> offset(n), seqPos(n);
		std::vector<uint8_t> nodeSwitch(n);
		for (uint64_t i = 0; i < n; i++) { long long a, b, c, d; in >> a >> b >> c >> d; node[i] = (int32_t)a; offset[i] = (uint32_t)b; seqPos[i] = (uint32_t)c; nodeSwitch[i] = (uint8_t)d; }
		gc::TraceView tv { node.data(), offset.data(), seqPos.data(), nodeSwitch.data(), n };
		if (merge == 2) std::cout << gc::vgToJson(gc::buildVgAlignment(graph, name, read.data(), read.size(), tv, (int32_t)score, (uint64_t)start, (uint64_t)end)) << "\n";
		else std::cout << gc::formatGafLine(graph, name, read.data(), read.size(), tv, merge != 0) << "\n";
	}
	return 0;
}
