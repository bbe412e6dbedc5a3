// One process, several GPUs: what the reference's alignReads (src/Aligner.cpp:1124-1310: one process, `-t N` worker threads over one read queue, src/Aligner.cpp:1267-1270)
// becomes on a node with several MI355X, written against the C ABI only (include/graphchainer_amd.h) - INTEGRATION.md §8 walks through it.
//   - the start-up work is done ONCE, on the host: gc_index_build writes graph + MPC index + minimizer index to a cache file (no GPU needed);
//   - every device gets its own replica: the first worker thread of a device calls gc_set_device + gc_index_load (the graph is immutable and shared by that device's threads);
//   - every worker thread has its own gc_stream (the reference's one AlignerGraphsizedState per thread, src/Aligner.cpp:469) and takes batches from ONE shared atomic cursor
//     until the reads run out: reads shard over the devices with no collective and no RCCL, a device that finishes early takes more;
//   - results come back per batch and are written in read order here (the reference's output order is whatever its writer threads see first).
// usage: multi_gpu_host graph.gfa reads.txt <logical devices> <threads per device> <reads per batch>
// Logical device d runs on physical device d % gc_device_count(): on a one-GPU box two logical devices are two replicas of the graph on the same GPU, which is how the GPU test
// drives the multi-device code path (two gc_graph handles, four streams, one process). Prints one line per read; exits 0 with NO_DEVICE when there is no GPU (no CPU fallback).
#include "graphchainer_amd.h"
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

struct Device {
	int physical = 0;
	std::once_flag loaded;
	gc_graph* graph = nullptr;
	gc_seeder* seeder = nullptr;
	std::string error;
};

int main(int argc, char** argv)
{
	if (argc < 6) { fprintf(stderr, "usage: multi_gpu_host graph.gfa reads.txt <logical devices> <threads per device> <reads per batch>\n"); return 2; }
	const int nPhysical = gc_device_count();
	if (nPhysical < 1) { printf("NO_DEVICE\n"); return 0; }
	const int nDevices = atoi(argv[3]), perDevice = atoi(argv[4]);
	const size_t batchReads = (size_t)atol(argv[5]);
	std::vector<std::string> reads;
	{
		std::ifstream in(argv[2]);
		for (std::string line; std::getline(in, line);) if (!line.empty()) reads.push_back(line);
	}
	// start-up, once, host only (src/Aligner.cpp:1137-1162 builds the same objects on every run; the cache is what its empty saveMPC / loadMPC would hold)
	const std::string cache = std::string(argv[2]) + ".gcidx";
	if (gc_index_build(argv[1], 15, 20, 1.0 - 0.001, cache.c_str()) != GC_OK) { fprintf(stderr, "gc_index_build: %s\n", gc_last_error()); return 1; }
	std::vector<Device> devices(nDevices);
	for (int d = 0; d < nDevices; d++) devices[d].physical = d % nPhysical;
	gc_params params;
	gc_params_default(&params);
	params.long_pass = 1; params.stitch = 1; params.edit_distances = 1;
	const size_t nBatches = (reads.size() + batchReads - 1) / batchReads;
	std::atomic<size_t> cursor { 0 };          // the shared batch cursor: the reference's read queue
	std::vector<std::string> lines(reads.size());
	std::vector<int> batchDevice(nBatches, -1);
	std::atomic<int> failures { 0 };
	auto worker = [&](int d, int t) {
		Device& dev = devices[d];
		if (gc_set_device(dev.physical) != GC_OK) { failures++; return; }             // the current device is per host thread
		std::call_once(dev.loaded, [&]() { if (gc_index_load(cache.c_str(), &dev.graph, &dev.seeder) != GC_OK) dev.error = gc_last_error(); });
		if (!dev.error.empty() || !dev.graph) { fprintf(stderr, "device %d: %s\n", d, dev.error.c_str()); failures++; return; }
		gc_stream* stream = nullptr;
		if (gc_stream_create(&stream) != GC_OK) { fprintf(stderr, "gc_stream_create: %s\n", gc_last_error()); failures++; return; }
		for (size_t b; (b = cursor.fetch_add(1)) < nBatches;) {
			const size_t r0 = b * batchReads, r1 = std::min(reads.size(), r0 + batchReads);
			std::string bases;
			std::vector<uint64_t> offsets { 0 };
			for (size_t r = r0; r < r1; r++) { bases += reads[r]; offsets.push_back(bases.size()); }
			gc_reads* batch = nullptr;
			gc_result* res = nullptr;
			if (gc_reads_upload(bases.data(), offsets.data(), r1 - r0, &batch) != GC_OK || gc_align_batch(dev.graph, dev.seeder, stream, batch, &params, &res) != GC_OK) {
				fprintf(stderr, "batch %zu on device %d thread %d: %s\n", b, d, t, gc_last_error());
				failures++;
				gc_reads_destroy(batch);
				break;
			}
			batchDevice[b] = d;
			for (size_t i = 0; i < r1 - r0; i++) {
				char buf[256];
				unsigned long long chainHash = 0;
				for (uint64_t c = res->read_chain_off[i]; c < res->read_chain_off[i + 1]; c++) chainHash = chainHash * 1000003ull + res->chain[c] + 1;
				snprintf(buf, sizeof buf, "read %zu anchors %llu chain %llu %llu score %llu long %llu dist %lld %lld better %d ties %u %u", r0 + i,
					(unsigned long long)(res->read_anchor_off[i + 1] - res->read_anchor_off[i]), (unsigned long long)(res->read_chain_off[i + 1] - res->read_chain_off[i]), chainHash,
					(unsigned long long)res->chain_score[i], (unsigned long long)(res->read_longall_off[i + 1] - res->read_longall_off[i]), (long long)res->long_edit_distance[i],
					(long long)res->chain_edit_distance[i], (int)res->chained_better[i], res->flatten_ties[i], res->flatten_ties_long[i]);
				lines[r0 + i] = buf;
			}
			gc_result_free(res);
			gc_reads_destroy(batch);
		}
		gc_stream_destroy(stream);
	};
	std::vector<std::thread> threads;
	for (int d = 0; d < nDevices; d++) for (int t = 0; t < perDevice; t++) threads.emplace_back(worker, d, t);
	for (auto& th : threads) th.join();
	for (auto& dev : devices) { if (dev.seeder) gc_seeder_destroy(dev.seeder); if (dev.graph) gc_graph_destroy(dev.graph); }
	remove(cache.c_str());
	if (failures) return 1;
	std::vector<size_t> perDev(nDevices, 0);
	for (int d : batchDevice) if (d >= 0) perDev[d]++;
	for (const std::string& l : lines) printf("%s\n", l.c_str());
	printf("batches");
	for (int d = 0; d < nDevices; d++) printf(" %zu", perDev[d]);
	printf("\n");
	return 0;
}
