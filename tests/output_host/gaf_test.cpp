// The product's GAF encoder (gc::formatGafLine, csrc/host/gc_output.cpp) on the CPU: formats the alignments of a dump file
//   (per alignment: "name merge n score start end" / the read / n rows "node offset seqPos nodeSwitch") against a graph and prints one GAF line each
//   (merge 0 / 1: the two CIGAR styles) or, with merge 2, the protobuf-JSON line of gc::buildVgAlignment + gc::vgToJson.
// tests/test_host_logic.py feeds it the oracle's whole-read alignments and compares the text with the oracle's own encoder.
// usage: gaf_test graph.gfa alignments.txt
#include "gc_output.hpp"
#include <cstdio>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

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
		if (merge == 3 || merge == 4) {
			// r4, the host's half of the device-encoded output (gc::appendGafLine, gc::vgProtobufFromEncoded, gc::vgFromEncoded): the pieces the device would
			// hand over are cut out of the host encoders' own output here - path and CIGAR columns and the counts; the vg::Path field of the message - and the
			// line / message put together from them must be the original
			gc::EncodedAlignment ea;
			const std::string line = gc::formatGafLine(graph, name, read.data(), read.size(), tv, false);
			std::vector<std::string> col;
			for (size_t at = 0; at <= line.size();) { size_t tab = line.find('\t', at); if (tab == std::string::npos) tab = line.size(); col.push_back(line.substr(at, tab - at)); at = tab + 1; }
			ea.alignmentStart = std::stoull(col[2]); ea.alignmentEnd = std::stoull(col[3]);
			ea.path = col[5].data(); ea.pathLen = col[5].size();
			ea.nodePathLen = std::stoull(col[6]); ea.nodePathStart = std::stoull(col[7]); ea.nodePathEnd = std::stoull(col[8]);
			ea.matches = std::stoull(col[9]); ea.cells = std::stoull(col[10]); ea.mismatches = std::stoull(col[12].substr(5));
			ea.cigar = col[15].data() + 5; ea.cigarLen = col[15].size() - 5;
			ea.score = (int32_t)score;
			if (merge == 3) { std::string out; gc::appendGafLine(out, name, read.size(), ea); std::cout << out << "\n"; continue; }
			const gc::VgAlignment aln = gc::buildVgAlignment(graph, name, read.data(), read.size(), tv, (int32_t)score, (uint64_t)start, (uint64_t)end);
			const std::string message = gc::vgToProtobuf(aln);
			// field 1 (sequence) then field 2 (path): skip the first, cut the second out
			size_t at = 0;
			auto varint = [&]() { uint64_t v = 0; int shift = 0; while (true) { unsigned char b = (unsigned char)message[at++]; v |= (uint64_t)(b & 0x7f) << shift; if (!(b & 0x80)) return v; shift += 7; } };
			if (message[at] == 0x0A) { at++; uint64_t len = varint(); at += len; }
			if (message[at] != 0x12) { std::cout << "NO PATH FIELD\n"; return 1; }
			at++;
			const uint64_t pathLen = varint();
			ea.vgPath = (const uint8_t*)message.data() + at; ea.vgPathLen = pathLen;
			if (gc::vgProtobufFromEncoded(name, read.data(), ea) != message) { std::cout << "MESSAGE DIFFERS\n"; return 1; }
			std::cout << gc::vgToJson(gc::vgFromEncoded(name, read.data(), ea)) << "\n";
			continue;
		}
		if (merge == 2) std::cout << gc::vgToJson(gc::buildVgAlignment(graph, name, read.data(), read.size(), tv, (int32_t)score, (uint64_t)start, (uint64_t)end)) << "\n";
		else std::cout << gc::formatGafLine(graph, name, read.data(), read.size(), tv, merge != 0) << "\n";
	}
	return 0;
}
