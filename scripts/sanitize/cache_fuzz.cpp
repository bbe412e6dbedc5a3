#include "gc_index_cache.hpp"
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <vector>
using namespace gc;
static uint64_t fnv(const std::vector<unsigned char>& b, size_t n){ uint64_t h=0xcbf29ce484222325ull; for(size_t i=0;i<n;i++){h^=b[i];h*=0x100000001b3ull;} return h; }
int main(int argc,char**argv){
	GfaGraph gfa=GfaGraph::LoadFromFile(argv[1]);
	AlignmentGraph g=AlignmentGraph::BuildFromGFA(gfa); g.buildMPC(true);
	MinimizerIndex idx=MinimizerIndex::Build(g,15,20,0.999);
	SaveIndexCache("/tmp/gc_sanitize/c.gcidx", g, &idx);
	IndexCacheInfo info=CheckIndexCache("/tmp/gc_sanitize/c.gcidx");
	printf("nodes %llu kmers %llu\n",(unsigned long long)info.nodes,(unsigned long long)info.kmers);
	AlignmentGraph g2; MinimizerIndex i2; LoadIndexCache("/tmp/gc_sanitize/c.gcidx", g2, i2);
	MinimizerIndex i3=MinimizerIndex::Build(g2,15,20,0.999);
	if(i3.positions!=idx.positions||i2.positions!=idx.positions){puts("MISMATCH");return 1;}
	std::ifstream f("/tmp/gc_sanitize/c.gcidx",std::ios::binary); std::vector<unsigned char> data((std::istreambuf_iterator<char>(f)),{});
	std::mt19937_64 rng(3); int refused=0,accepted=0; int trials=atoi(argv[2]);
	for(int t=0;t<trials;t++){
		std::vector<unsigned char> b(data.begin(),data.end()-8);
		size_t at=9+rng()%(b.size()-9);
		int kind=t%3;
		if(kind==0) b[at]^=(unsigned char)(1u<<(rng()%8)); else if(kind==1){ static const unsigned char v[4]={0,0x7f,0x80,0xff}; b[at]=v[rng()%4]; } else b.erase(b.begin()+at,b.begin()+std::min(b.size(),at+1+rng()%8));
		uint64_t h=fnv(b,b.size()); for(int i=0;i<8;i++) b.push_back((unsigned char)(h>>(8*i)));
		{ std::ofstream o("/tmp/gc_sanitize/m.gcidx",std::ios::binary); o.write((const char*)b.data(),b.size()); }
		try{ CheckIndexCache("/tmp/gc_sanitize/m.gcidx"); accepted++; }catch(const std::exception&){ refused++; }
	}
	printf("refused %d accepted %d\n",refused,accepted);
}
