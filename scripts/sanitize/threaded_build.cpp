#include "gc_graph.hpp"
#include <cstdio>
using namespace gc;
int main(int argc,char**argv){
	GfaGraph gfa=GfaGraph::LoadFromFile(argv[1]);
	AlignmentGraph g=AlignmentGraph::BuildFromGFA(gfa); g.buildMPC(true);
	MinimizerIndex idx=MinimizerIndex::Build(g,15,20,0.999);
	printf("nodes %zu kmers %zu positions %zu\n", g.NodeSize(), idx.kmers.size(), idx.positions.size());
}
