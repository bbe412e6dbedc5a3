// The two graph builders of graphchainer_amd/csrc/host side by side on one GFA file: GfaGraph::LoadFromFile + AlignmentGraph::BuildFromGFA (the reference's own containers,
// gc_graph.cpp) and AlignmentGraph::BuildFromGFAFile (flat arrays + the replayed container orders, gc_graph_fast.cpp). Every array the kernels, the minimizer index and the
// output encoders read must be equal; when the file is refused, both must refuse it with the same message. Prints "SAME ..." or the first difference.
#include "gc_graph.hpp"
#include <cstdio>
#include <cstdlib>
#include <string>

template <class A> static bool same(const char* what, const A& a, const A& b)
{
	if (a == b) return true;
	printf("DIFFERENT %s\n", what);
	return false;
}

int main(int argc, char** argv)
{
	if (argc < 2) return 2;
	std::string errLiteral, errFast;
	gc::AlignmentGraph a, b;
	bool okA = true, okB = true;
	try { gc::GfaGraph gfa = gc::GfaGraph::LoadFromFile(argv[1]); a = gc::AlignmentGraph::BuildFromGFA(gfa); } catch (const std::exception& e) { okA = false; errLiteral = e.what(); }
	try { b = gc::AlignmentGraph::BuildFromGFAFile(argv[1]); } catch (const std::exception& e) { okB = false; errFast = e.what(); }
	if (!okA || !okB) {
		if (okA != okB || errLiteral != errFast) { printf("DIFFERENT outcome: literal [%s] fast [%s]\n", okA ? "built" : errLiteral.c_str(), okB ? "built" : errFast.c_str()); return 1; }
		printf("SAME error: %s\n", errLiteral.c_str());
		return 0;
	}
	bool ok = true;
	ok = same("nodeLength", a.nodeLength, b.nodeLength) && ok;
	ok = same("nodeOffset", a.nodeOffset, b.nodeOffset) && ok;
	ok = same("nodeIDs", a.nodeIDs, b.nodeIDs) && ok;
	ok = same("inNeighbors", a.inNeighbors, b.inNeighbors) && ok;
	ok = same("outNeighbors", a.outNeighbors, b.outNeighbors) && ok;
	ok = same("reverse", a.reverse, b.reverse) && ok;
	ok = same("linearizable", a.linearizable, b.linearizable) && ok;
	ok = same("nodeSequences", a.nodeSequences, b.nodeSequences) && ok;
	ok = same("componentNumber", a.componentNumber, b.componentNumber) && ok;
	ok = same("chainNumber", a.chainNumber, b.chainNumber) && ok;
	ok = same("chainApproxPos", a.chainApproxPos, b.chainApproxPos) && ok;
	ok = same("nodeLookupOrder", a.nodeLookupOrder, b.nodeLookupOrder) && ok;
	ok = same("bpSize", a.bpSize, b.bpSize) && ok;
	ok = same("firstAmbiguous", a.firstAmbiguous, b.firstAmbiguous) && ok;
	if (a.ambiguousNodeSequences.size() != b.ambiguousNodeSequences.size()) { printf("DIFFERENT ambiguous count\n"); ok = false; }
	else for (size_t i = 0; i < a.ambiguousNodeSequences.size(); i++) {
		const gc::AmbiguousSeq &x = a.ambiguousNodeSequences[i], &y = b.ambiguousNodeSequences[i];
		if (x.A != y.A || x.C != y.C || x.G != y.G || x.T != y.T) { printf("DIFFERENT ambiguous sequence %zu\n", i); ok = false; break; }
	}
	if (a.nodeLookup.size() != b.nodeLookup.size() || a.originalNodeSize.size() != b.originalNodeSize.size() || a.originalNodeName.size() != b.originalNodeName.size()) { printf("DIFFERENT table sizes\n"); ok = false; }
	if (ok) for (int id : a.nodeLookupOrder) {
		if (!b.nodeLookup.contains(id)) { printf("DIFFERENT lookup: id %d missing\n", id); ok = false; break; }
		const gc::NodeLookup::Span x = a.nodeLookup.at(id), y = b.nodeLookup.at(id);
		if (x.size() != y.size() || !std::equal(x.begin(), x.end(), y.begin())) { printf("DIFFERENT lookup of id %d\n", id); ok = false; break; }
		if (a.originalNodeSize.at(id) != b.originalNodeSize.at(id) || a.OriginalNodeName(id) != b.OriginalNodeName(id)) { printf("DIFFERENT size / name of id %d\n", id); ok = false; break; }
	}
	// ids no S line named must be absent from both
	if (ok) for (int id = -2; id < (int)(2 * a.nodeLookupOrder.size() + 64); id++) if (a.nodeLookup.contains(id) != b.nodeLookup.contains(id) || a.originalNodeSize.count(id) != b.originalNodeSize.count(id) || a.originalNodeName.count(id) != b.originalNodeName.count(id)) { printf("DIFFERENT presence of id %d\n", id); ok = false; break; }
	if (!ok) return 1;
	printf("SAME graph: %zu split nodes, %zu bigraph nodes, %zu ambiguous\n", a.NodeSize(), a.nodeLookup.size(), a.ambiguousNodeSequences.size());
	return 0;
}
