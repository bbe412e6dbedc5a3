// Shared by the two translation units of the library's host side (r4: gc_capi.hip - the C entry points - and gc_batch.hip - gc_align_batch and the batch pipeline;
// one 3 500-line file until then): error reporting, device / pinned buffers, the whole-read tokens and the device's shared scratch, the worker pool, the block caches, the
// opaque handle types of the C ABI, graph upload, the edit-distance launcher. Everything here has ONE definition in the library (inline functions and variables).
#pragma once
// C ABI of the MI355X GraphChainer hot path (include/graphchainer_amd.h) and the batched host pipeline that
// drives the HIP kernels. No CPU fallback: every entry point that needs the device fails with GC_ERR_DEVICE
// when HIP is unavailable.
#include "../../include/graphchainer_amd.h"
#include "hip/gc_kernels.hpp"
#include "host/gc_graph.hpp"
#include "host/gc_hashorder.hpp"
#include "host/gc_glue.hpp"
#include "host/gc_output.hpp"
#include "host/gc_index_cache.hpp"
#include "host/gc_correctness.hpp"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

using namespace gcdev;

inline thread_local std::string g_lastError;
inline int fail(int code, const std::string& msg) { g_lastError = msg; return code; }

struct DeviceError : std::runtime_error { using std::runtime_error::runtime_error; };
#define HIP_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) throw DeviceError(std::string("HIP error: ") + hipGetErrorString(e_) + " at " #expr); } while (0)

// HIP maps its streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default), and kernels of different streams that share a hardware queue
// run one after the other. A batch in flight uses a dozen streams (fragment pipeline, whole-read rounds, edit-distance classes), two batches
// twice that: on 4 queues the whole-read pass's rounds wait behind the other batch's k_extend / k_chain / k_stitch (kernel trace: 60 ms of
// foreign kernels between two rounds). 16 queues: 228 -> 209 ms per batch on cfg2; 32 oversubscribe the command processor (290 ms).
// The HIP runtime reads the variable at its first call and it configures the whole process, so it is the HOST's to set (bench.py and the
// scripts set GPU_MAX_HW_QUEUES=16 before anything touches HIP); the library only says so, once, when a second stream is created without it.
inline void noteHardwareQueues(int streamsAlive)
{
	static std::atomic<bool> said { false };
	const char* e = getenv("GPU_MAX_HW_QUEUES");
	if (streamsAlive >= 2 && (!e || atoi(e) < 8) && !said.exchange(true))
		fprintf(stderr, "[graphchainer_amd] note: GPU_MAX_HW_QUEUES is %s; with several gc_streams per device set it to 16 before the process's first HIP call (INTEGRATION.md §7), or batches in flight serialise on HIP's 4 default hardware queues\n", e ? e : "unset");
}

namespace gcrt {

template <typename T>
T* uploadVector(const std::vector<T>& v)
{
	T* d = nullptr;
	size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
	HIP_CHECK(hipMalloc((void**)&d, bytes));
	if (!v.empty()) HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
	return d;
}

struct DeviceBuffer {   // growable device allocation owned by a stream object
	void* ptr = nullptr;
	size_t bytes = 0;
	template <typename T> T* reserve(size_t count, bool tight = false)   // tight: no head room (the pools that are sized by use bring their own slack)
	{
		size_t need = std::max<size_t>(count, 1) * sizeof(T);
		if (need > bytes) {
			if (ptr) HIP_CHECK(hipFree(ptr));
			ptr = nullptr;
			bytes = 0;
			size_t want = tight ? need + 256 : need + need / 8 + 256;
			HIP_CHECK(hipMalloc(&ptr, want));
			bytes = want;
		}
		return (T*)ptr;
	}
	void release() { if (ptr) (void)hipFree(ptr); ptr = nullptr; bytes = 0; }
	// a pool that a rerun sized from an overshooting request count comes back to what its batches use: called once, by the batch after the rerun (hipFree drains the device)
	void shrinkTo(size_t needBytes)
	{
		const char* env = getenv("GC_TEST_POOL_SHRINK_FLOOR");   // (test hook, read per call - the tests set it mid-process: small pools shrink too)
		const size_t floor = env ? (size_t)std::max(0ll, atoll(env)) : (size_t)(64u << 20);
		if (bytes > needBytes + needBytes / 4 + floor) release();
	}
	~DeviceBuffer() { if (ptr) (void)hipFree(ptr); }
};

// One whole-read pass at a time per device: its rounds saturate the scalar issue ports of the whole chip, so two passes side by side
// (two batches in flight on two gc_streams) only take turns at a finer grain and both finish late. With the token the second batch's
// seeding, host glue and fragment pipeline overlap the first batch's whole-read pass, and its own pass starts the moment the first
// one ends - a software pipeline over batches (GC_LONG_TOKEN=0 turns it off).
// r4 / r5: two tokens per device, each with a scratch of its own, for passes that cannot fill the chip (longTokenCount below)
inline constexpr int LONG_TOKENS_MAX = 2;
struct PassTokens {
	std::mutex m;
	std::condition_variable cv;
	bool busy[LONG_TOKENS_MAX] = { false, false };
	bool exclusive = false;       // slot 0 is held by a pass that fills the chip: nobody beside it
	int exclusiveWaiting = 0;     // such passes waiting (sharers let them go first: a stream of small batches must not starve a large one)
	// n: the slots the pass may take (1 or 2). alone: the pass fills the chip and runs with NO other pass beside it (r5 decided the count per batch but a count of 1 only
	// restricted the caller's choice of slot, so a large batch could run beside a small one holding slot 1 - ADVICE r5); a sharer never starts beside such a pass
	int acquire(int n, bool alone)
	{
		std::unique_lock<std::mutex> l(m);
		if (alone) {
			exclusiveWaiting++;
			cv.wait(l, [&]() { for (int s = 0; s < LONG_TOKENS_MAX; s++) if (busy[s]) return false; return true; });
			exclusiveWaiting--;
			busy[0] = true; exclusive = true;
			return 0;
		}
		int slot = -1;
		cv.wait(l, [&]() { if (exclusive || exclusiveWaiting > 0) return false; for (int s = 0; s < n; s++) if (!busy[s]) { slot = s; return true; } return false; });
		busy[slot] = true;
		return slot;
	}
	void release(int slot) { { std::lock_guard<std::mutex> l(m); busy[slot] = false; if (slot == 0) exclusive = false; } cv.notify_all(); }
};
inline PassTokens g_longPassToken[16];
// what std::unique_lock was for the single token: released when the holder goes out of scope
struct TokenHold {
	PassTokens* tokens = nullptr;
	int slot = -1;
	void lock(PassTokens& t, int n, bool alone) { tokens = &t; slot = t.acquire(n, alone); }
	bool owns_lock() const { return slot >= 0; }
	void unlock() { if (slot >= 0) { tokens->release(slot); slot = -1; } }
	~TokenHold() { unlock(); }
};
// Switches of measured-and-rejected alternatives (a token per round, read groups, the lane-per-extension kernel, ...) exist only in the experiments build
// (`make -C graphchainer_amd/csrc experiments`, -DGC_EXPERIMENTS): the product library does not read them. INTEGRATION.md §7 lists the switches that remain.
#ifdef GC_EXPERIMENTS
inline const char* expEnv(const char* name) { return getenv(name); }
#else
inline const char* expEnv(const char*) { return nullptr; }
#endif
// r5: how many passes may run side by side on a device is decided per batch. A pass whose rounds cannot fill the chip - round 0 holds two work items per read, k_long_extend<1> has
// 5 120 wave slots (256 CUs x 4 SIMDs x 5 waves) - shares the device with a second one: 2 000 x 50 kb reads on a 192 Mbp graph 4.0-4.2 k -> 5.0-5.3 k reads/s (`gpurun_out/r5_cfg5_tok`);
// a pass that fills it (10 k x 10 kb: 20 000 items) keeps the device to itself (two side by side measured slower in r4). GC_LONG_TOKENS=1|2 overrides (read per batch: the tests switch
// inside one process). The second token has a scratch of its own and is only taken when the device has the memory for it.
inline constexpr uint64_t LONG_WAVE_SLOTS = 5120;
// r6: ... and only while the fragment pipeline leaves the device room for it, told by the fragment extensions the stream's previous batch ran per read base (a count, not a time:
// times under five batches in flight are residencies, and a rule on them would feed back on itself). On a 192 Mbp graph that is 0.09 and two passes side by side gave 4.2 -> 5.0 k
// reads/s (r5); at 960 Mbp, where chance hits of 15-mers make it 0.42, the second pass that r6's freed memory suddenly had room for cost 4.31 -> 3.63 k (`gpurun_out/r6_cfg5_960p` / `_960q`)
inline int longTokenCount(uint64_t nReads, uint64_t streamBatchesDone, double extensionsPerBase)
{
	if (const char* e = getenv("GC_LONG_TOKENS")) return std::max(1, std::min(LONG_TOKENS_MAX, atoi(e)));
	// (a stream's first batch sizes its buffers - pools that rerun and grow: a second scratch taken while the device still looks empty cost config 5 at 960 Mbp its fifth stream)
	return streamBatchesDone > 0 && 2 * nReads + 128 <= LONG_WAVE_SLOTS && extensionsPerBase < 0.2 ? 2 : 1;
}
inline std::mutex g_longRoundToken[16];   // GC_LONG_TOKEN=2 (experiment): the token handed over per round
// The pass's extension scratch (up to 48 GB: one region per resident wave) is only touched while the token is held, so the gc_streams of a device share ONE
// (r3: 85 -> 37 GB per stream for 10 k x 10 kb batches, which is what lets five batches be in flight on a 288 GB device instead of three). It belongs to the
// token: reserved (grown) by the pass that holds it, freed when the device's last gc_stream goes.
struct SharedLongScratch { DeviceBuffer buffer[LONG_TOKENS_MAX]; int streams = 0; };
inline SharedLongScratch g_longScratch[16];
inline std::mutex g_longScratchCount;

inline double nowUs() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// CPU time of the whole process (all threads), for GC_DEBUG_TIMES' host budget lines
inline double threadCpuMs() { timespec ts {}; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6; }
inline double processCpuMs() { timespec ts {}; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6; }

// Waiting for a stream. hipStreamSynchronize spins on a CPU, and a batch has two host threads waiting most of its 200 ms (two batches in
// flight: four CPUs' worth of spinning, half of what a batch costs the host). Modes (GC_SPIN_SYNC): 2 (default) polls hipStreamQuery and
// sleeps 40 us between polls - the waits are tens of microseconds late and cost next to no CPU; 1 spins (the r2 behaviour); 0 sleeps on a
// hipEventBlockingSync event (interrupt-driven; measured slower than spinning on this pool's boxes: 255 against 237 ms per batch).
inline void syncStream(hipStream_t q)
{
	static const int mode = getenv("GC_SPIN_SYNC") ? atoi(getenv("GC_SPIN_SYNC")) : 2;
	if (mode == 1) { HIP_CHECK(hipStreamSynchronize(q)); return; }
	if (mode == 2) {
		// r5: the sleep between polls backs off from 40 us (GC_SYNC_POLL_US) to four times that once a wait has lasted a millisecond: a batch's threads wait ~1 s in all per 150 ms
		// step, mostly for kernels of tens of milliseconds, and every poll is a runtime call and a nanosleep
		static const int pollUs = getenv("GC_SYNC_POLL_US") ? std::max(1, atoi(getenv("GC_SYNC_POLL_US"))) : 40;
		int sleepUs = pollUs;
		for (int spins = 0;; spins++) {
			const hipError_t e = hipStreamQuery(q);
			if (e == hipSuccess) return;
			if (e != hipErrorNotReady) HIP_CHECK(e);
			if (spins >= 4) {   // (the first few polls back to back: many waits are for kernels of a few microseconds)
				std::this_thread::sleep_for(std::chrono::microseconds(sleepUs));
				if (spins >= 12 && sleepUs < 4 * pollUs) sleepUs += sleepUs / 4 + 1;
			}
		}
	}
	// one blocking-sync event per device this thread has waited on, destroyed with the thread (the whole-read pass threads live for one batch)
	struct Events { hipEvent_t e[16] = {}; ~Events() { for (auto& x : e) if (x) (void)hipEventDestroy(x); } };
	static thread_local Events events;
	int device = 0;
	HIP_CHECK(hipGetDevice(&device));
	hipEvent_t& e = events.e[device & 15];
	if (!e) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventBlockingSync | hipEventDisableTiming));
	HIP_CHECK(hipEventRecord(e, q));
	HIP_CHECK(hipEventSynchronize(e));
}

// The same for one event (the round token of the whole-read pass is released the moment its extension kernel has finished).
inline void syncEvent(hipEvent_t ev)
{
	static const int pollUs = getenv("GC_SYNC_POLL_US") ? std::max(1, atoi(getenv("GC_SYNC_POLL_US"))) : 40;
	int sleepUs = pollUs;
	for (int spins = 0;; spins++) {
		const hipError_t e = hipEventQuery(ev);
		if (e == hipSuccess) return;
		if (e != hipErrorNotReady) HIP_CHECK(e);
		if (spins >= 4) {
			std::this_thread::sleep_for(std::chrono::microseconds(sleepUs));
			if (spins >= 12 && sleepUs < 4 * pollUs) sleepUs += sleepUs / 4 + 1;
		}
	}
}

// Persistent worker pool for the per-read host glue (threads are created once per process).
class WorkerPool {
public:
	static WorkerPool& instance() { static WorkerPool p(0); return p; }
	// r5: the batch pipeline's own, smaller pool. Its jobs became short (the seed glue, the stitching and the anchor arrays are kernels now: what is left is a few ms of per-read
	// bookkeeping and memcpy per stage), and a job wakes every thread of its pool: with 32 workers a batch cost 0.29 CPU-seconds, with 8 0.185 at the same rate (`gpurun_out/r5_hostcpu`).
	// The output encoders (GAF text, zlib) keep the wide pool.
	static WorkerPool& batch() { static WorkerPool p(1); return p; }
	size_t size() const { return workers.size() + 1; }
	// runs body(i, worker) for i in [0, n); worker in [0, size())
	void run(size_t n, const std::function<void(size_t, size_t)>& body)
	{
		if (n == 0) return;
		if (workers.empty() || n < 4) { for (size_t i = 0; i < n; i++) body(i, 0); return; }
		std::lock_guard<std::mutex> oneJob(runMutex);   // batches in flight on different gc_streams take turns on the pool
		{
			std::unique_lock<std::mutex> lock(mutex);
			job = &body;
			total = n;
			next.store(0);
			pending = workers.size();
			failure = nullptr;
			generation++;
		}
		wake.notify_all();
		work(0);
		std::unique_lock<std::mutex> lock(mutex);
		done.wait(lock, [&]() { return pending == 0; });
		job = nullptr;
		if (failure) { std::exception_ptr e = failure; failure = nullptr; std::rethrow_exception(e); }   // (an exception in a pool thread used to end the process)
	}
private:
	explicit WorkerPool(int kind)
	{
		size_t n = std::max(1u, std::thread::hardware_concurrency());
		n = std::min<size_t>(n, 96);   // the glue is memory-bound; more threads stop helping
		// A container with a CPU bandwidth quota (cgroup cpu.max) shows all of the machine's threads but is throttled for the rest of the
		// 100 ms period once a burst of workers has spent the quota - measured on a 16-CPU quota: 96 workers finish a stage in 10 ms and the
		// whole process (the whole-read pass's round loop included) then stalls for 50-60 ms. Twice the quota keeps the bursts inside it.
		const double quota = gc::cpuQuota();
		if (quota > 0) n = std::min<size_t>(n, std::max<size_t>(4, (size_t)(2 * quota + 0.5)));
		if (const char* env = getenv("GC_HOST_THREADS")) n = (size_t)std::max(1, atoi(env));
		if (kind == 1) {
			n = std::min<size_t>(n, quota > 0 ? std::max<size_t>(4, (size_t)(quota / 2 + 0.5)) : 8);
			if (const char* env = getenv("GC_BATCH_THREADS")) n = (size_t)std::max(1, atoi(env));
		}
		for (size_t t = 1; t < n; t++) workers.emplace_back([this, t]() { loop(t); });
	}
	~WorkerPool()
	{
		{ std::unique_lock<std::mutex> lock(mutex); stop = true; generation++; }
		wake.notify_all();
		for (auto& w : workers) w.join();
	}
	void work(size_t id)
	{
		const size_t chunk = 4;
		static const bool account = getenv("GC_DEBUG_TIMES") != nullptr;
		const double cpu0 = account ? threadCpuMs() : 0;
		try {
			for (size_t i; (i = next.fetch_add(chunk)) < total;)
				for (size_t k = i; k < std::min(total, i + chunk); k++) (*job)(k, id);
			if (account) cpuUs.fetch_add((uint64_t)((threadCpuMs() - cpu0) * 1e3));
		} catch (...) {
			next.store(total);   // the first failure ends the job: the other threads stop fetching, the caller rethrows
			std::unique_lock<std::mutex> lock(mutex);
			if (!failure) failure = std::current_exception();
		}
	}
	void loop(size_t id)
	{
		size_t seen = 0;
		while (true) {
			{
				std::unique_lock<std::mutex> lock(mutex);
				wake.wait(lock, [&]() { return generation != seen; });
				seen = generation;
				if (stop) return;
			}
			work(id);
			std::unique_lock<std::mutex> lock(mutex);
			if (--pending == 0) done.notify_one();
		}
	}
	std::vector<std::thread> workers;
	std::mutex mutex, runMutex;
	std::condition_variable wake, done;
	const std::function<void(size_t, size_t)>* job = nullptr;
	std::atomic<size_t> next { 0 };
public:
	std::atomic<uint64_t> cpuUs { 0 };   // GC_DEBUG_TIMES: CPU time spent inside jobs by all threads (what a stage costs beside its own thread)
private:
	size_t total = 0, pending = 0, generation = 0;
	std::exception_ptr failure;
	bool stop = false;
};

struct PinnedBuffer {   // growable page-locked host staging buffer (full-rate PCIe copies)
	void* ptr = nullptr;
	size_t bytes = 0;
	template <typename T> T* reserve(size_t count)
	{
		size_t need = std::max<size_t>(count, 1) * sizeof(T);
		if (need > bytes) {
			if (ptr) HIP_CHECK(hipHostFree(ptr));
			ptr = nullptr;
			bytes = 0;
			size_t want = need + need / 8 + 4096;
			HIP_CHECK(hipHostMalloc(&ptr, want, hipHostMallocDefault));
			bytes = want;
		}
		return (T*)ptr;
	}
	~PinnedBuffer() { if (ptr) (void)hipHostFree(ptr); }
};

template <typename T> inline T* mallocArray(size_t n) { return (T*)malloc(std::max<size_t>(n, 1) * sizeof(T)); }

// The arrays of a gc_result. The big ones (the trace arrays of keep_traces: 0.1-0.5 GB each for 10 k x 10 kb reads) come from a small cache of blocks that
// gc_result_free gives back: fresh memory of that size is mapped and zero-filled page by page on first touch, every batch again (r3: ~1.5 of the 3.4 CPU-seconds the
// assembly of a traced batch cost). Every array carries a 64-byte header with its size, so gc_result_free knows what it holds.
struct ResultBlockCache {
	static constexpr size_t HEADER = 64, MAX_HELD = 24ull << 30;
	// test hooks: GC_RESULT_CACHE_MIN=bytes recycles arrays from that size on (default 32 MB), GC_TEST_RESULT_CACHE_POISON=1 fills every array with 0xA5 when it is handed out -
	// together they show any reader that counts on an array's unwritten part being zero (fresh pages are, recycled ones are not)
	const size_t BIG = getenv("GC_RESULT_CACHE_MIN") ? (size_t)std::max(1ll, atoll(getenv("GC_RESULT_CACHE_MIN"))) : (32ull << 20);
	const bool poison = getenv("GC_TEST_RESULT_CACHE_POISON") != nullptr;
	std::mutex mutex;
	std::vector<std::pair<char*, size_t>> blocks;   // (base, capacity in bytes without the header)
	size_t held = 0;
	void* get(size_t bytes)
	{
		if (bytes >= BIG) {
			std::lock_guard<std::mutex> lock(mutex);
			size_t best = blocks.size();
			for (size_t i = 0; i < blocks.size(); i++)
				if (blocks[i].second >= bytes && blocks[i].second <= 2 * bytes && (best == blocks.size() || blocks[i].second < blocks[best].second)) best = i;
			if (best < blocks.size()) {
				char* base = blocks[best].first;
				held -= blocks[best].second;
				const size_t capacityHeld = blocks[best].second;
				blocks.erase(blocks.begin() + (long)best);
				if (poison) memset(base + HEADER, 0xA5, capacityHeld);
				return base + HEADER;
			}
		}
		const size_t capacity = bytes >= BIG ? bytes + bytes / 16 : bytes;   // (a little slack: the next batch's arrays are about, not exactly, this size)
		char* base = (char*)malloc(capacity + HEADER);
		if (!base) throw std::bad_alloc();
		*(size_t*)base = capacity;
		if (poison) memset(base + HEADER, 0xA5, capacity);
		return base + HEADER;
	}
	void put(void* p)
	{
		if (!p) return;
		char* base = (char*)p - HEADER;
		const size_t capacity = *(size_t*)base;
		if (capacity >= BIG) {
			std::lock_guard<std::mutex> lock(mutex);
			if (held + capacity <= MAX_HELD && blocks.size() < 64) { blocks.emplace_back(base, capacity); held += capacity; return; }
		}
		free(base);
	}
	void trim()   // gives every held block back to the allocator (gc_result_cache_trim)
	{
		std::lock_guard<std::mutex> lock(mutex);
		for (auto& b : blocks) free(b.first);
		blocks.clear();
		held = 0;
	}
};
// deliberately never destroyed: a language runtime's finalizers may still call gc_result_free while the process's static destructors run
inline ResultBlockCache& g_resultBlocks = *new ResultBlockCache();
template <typename T> inline T* resultArray(size_t n) { return (T*)g_resultBlocks.get(std::max<size_t>(n, 1) * sizeof(T)); }

} // namespace gcrt
using namespace gcrt;

// ----------------------------------------------------------------------------------------------------
struct gc_graph {
	gc::AlignmentGraph host;
	bool hostMpcTrimmed = false;   // gc_graph_trim_host: host.mpc / paths / backwards / topo / topo_ids / component_ids are empty
	// dense-by-bigraph-node-id copies of the twin lookup tables (the same arrays the device gets): original node size, and the
	// split nodes of every bigraph node in offset order (chunk k covers offsets [64k, 64k+64))
	std::vector<uint32_t> hOrigSize, hLookupOff, hLookup;
	// reverse-strand twin of (split node, offset): GetReversePosition + GetUnitigNode (src/AlignmentGraph.cpp:832-868) without the hash maps
	inline void twinOf(uint32_t node, uint32_t offset, uint32_t& twinNode, uint32_t& twinOffset) const
	{
		uint32_t id = (uint32_t)host.nodeIDs[node];
		uint32_t rev = hOrigSize[id] - 1 - ((uint32_t)host.nodeOffset[node] + offset);
		twinNode = hLookup[hLookupOff[id ^ 1] + rev / 64];
		twinOffset = rev - (uint32_t)host.nodeOffset[twinNode];
	}
	DGraph dev {};
	OutNames devNames {};            // GFA segment names by bigraph node id (output encoding on the device, gc_output.hip)
	std::vector<void*> allocations;
	CorrectnessTables* devTables = nullptr;
	uint8_t* devIupac = nullptr;
	uint32_t maxMpcWidth = 0, maxPathsPerNode = 1, maxBackPerNode = 1;
	~gc_graph() { for (void* p : allocations) (void)hipFree(p); }
	template <typename T> const T* up(const std::vector<T>& v) { T* d = uploadVector(v); allocations.push_back(d); return d; }
};

struct gc_seeder {
	gc::MinimizerIndex host;
	SeedIndex dev {};
	std::vector<void*> allocations;
	~gc_seeder() { for (void* p : allocations) (void)hipFree(p); }
};

// Device and pinned blocks of read batches, kept for the next batch (r4). gc_reads_upload used to hipMalloc nine arrays per batch and gc_reads_destroy to hipFree them:
// hipFree waits for the device to drain, so with five batches in flight every destroy stalled its host thread for the length of whatever was queued (and the upload of the
// next batch behind it: 245 ms per batch in the end-to-end leg, `gpurun_out/r4_ab3`). One block per batch now, carved into the arrays, returned to a small cache.
struct BlockCache {
	bool pinned;
	std::mutex mutex;
	struct Block { void* ptr; size_t bytes; int device; };
	std::vector<Block> blocks;
	static constexpr size_t MAX_BLOCKS = 12;
	size_t maxBytes;          // r5: the cache holds back at most this many bytes in all (a block that would exceed it is freed instead)
	size_t heldBytes = 0;
	explicit BlockCache(bool pinned, size_t maxBytes = (size_t)24 << 30) : pinned(pinned), maxBytes(maxBytes) {}
	void* get(size_t bytes, int device, size_t& capacity)
	{
		{
			std::lock_guard<std::mutex> lock(mutex);
			size_t best = blocks.size();
			for (size_t i = 0; i < blocks.size(); i++)
				if (blocks[i].device == device && blocks[i].bytes >= bytes && blocks[i].bytes <= 2 * bytes + (1u << 20) && (best == blocks.size() || blocks[i].bytes < blocks[best].bytes)) best = i;
			if (best < blocks.size()) { Block b = blocks[best]; blocks.erase(blocks.begin() + (long)best); heldBytes -= b.bytes; capacity = b.bytes; return b.ptr; }
		}
		capacity = bytes + bytes / 8 + 4096;   // (a little slack: the next batch is about, not exactly, this size)
		void* p = nullptr;
		if (pinned) HIP_CHECK(hipHostMalloc(&p, capacity, hipHostMallocDefault));
		else HIP_CHECK(hipMalloc(&p, capacity));
		return p;
	}
	void put(void* p, size_t bytes, int device)
	{
		if (!p) return;
		{
			std::lock_guard<std::mutex> lock(mutex);
			if (blocks.size() < MAX_BLOCKS && heldBytes + bytes <= maxBytes) { blocks.push_back(Block { p, bytes, device }); heldBytes += bytes; return; }
		}
		if (pinned) (void)hipHostFree(p); else (void)hipFree(p);
	}
	void trim()   // gives everything held back to the allocator (gc_result_cache_trim)
	{
		std::vector<Block> mine;
		{ std::lock_guard<std::mutex> lock(mutex); mine.swap(blocks); heldBytes = 0; }
		for (const Block& b : mine) { if (pinned) (void)hipHostFree(b.ptr); else (void)hipFree(b.ptr); }
	}
};
inline BlockCache& g_readDeviceBlocks = *new BlockCache(false);   // (leaked on purpose, like the result cache: finalizers may run late)
inline BlockCache& g_readPinnedBlocks = *new BlockCache(true);
// r5: the GAM deflate's staging (a batch's inflated groups plus the output bound: hundreds of MB) has caches of its own - in the read batches' caches its other
// size class evicted read blocks, and an evicted block is a hipFree / hipHostFree: the device-wide stall the caches exist to avoid
inline BlockCache& g_deflateDeviceBlocks = *new BlockCache(false, (size_t)4 << 30);
inline BlockCache& g_deflatePinnedBlocks = *new BlockCache(true, (size_t)4 << 30);

// a stream of the calling thread's own for its uploads and small jobs (created on first use, recreated when the thread changes device)
inline hipStream_t threadStream(int device)
{
	static thread_local struct ThreadStream { hipStream_t q = nullptr; int device = -1; ~ThreadStream() { if (q) (void)hipStreamDestroy(q); } } ts;
	if (!ts.q || ts.device != device) { if (ts.q) (void)hipStreamDestroy(ts.q); ts.q = nullptr; HIP_CHECK(hipStreamCreateWithFlags(&ts.q, hipStreamNonBlocking)); ts.device = device; }
	return ts.q;
}

struct gc_reads {
	void* deviceBlock = nullptr; size_t deviceBlockBytes = 0; int device = 0;   // every device array below is carved from this one block
	std::vector<uint64_t> offsets;   // host copy [n+1]
	uint64_t totalBases = 0;
	std::vector<uint8_t> invalid;    // read has a character outside the IUPAC alphabet (the reference's Complement() asserts)
	char* devBases = nullptr;        // [2*totalBases]: all reads forward, then every read reverse-complemented in place
	uint64_t* devOffsets = nullptr;
	// per read: match-mask bit vectors [strand fwd/rc][A,C,G,T][words] (bit i set: read position i matches that base)
	uint64_t* devMasks = nullptr;
	std::vector<uint64_t> maskOff;   // [n] word offset of read r's masks
	std::vector<uint32_t> maskWords; // [n] words per bit vector
	const uint64_t* devMaskOff = nullptr; const uint32_t* devMaskWords = nullptr;   // the two on the device (the fragment extension kernel reads its rows' masks through them)
	// exact-match bit vectors of the forward strand [A,C,G,T][words] and the per-read records of the NW kernel (rows = read bases)
	uint64_t* devEqMasks = nullptr;
	EdRead* devEdReads = nullptr;
	uint32_t* devChunkRead = nullptr;   // read containing the first base of every 64-base chunk of the concatenated forward bases
	uint64_t* devPacked = nullptr;      // the forward bases, 2 bits each, big-endian inside 64-bit words (for the seed kernel's k-mers)
	uint64_t* devInvalid = nullptr;     // one bit per forward base: not A, C, G or T (same big-endian convention)
	uint8_t* devReadInvalid = nullptr;  // [n] the device's copy of `invalid`
	~gc_reads() { g_readDeviceBlocks.put(deviceBlock, deviceBlockBytes, device); }
};

struct StitchedPath { std::vector<uint32_t> nodes; uint32_t firstOffset = 0, lastOffset = 0; uint64_t cells = 0; };

struct ReadGlue {
	std::vector<gc::SeedRec> seeds;       // fragment-pass order (by seqPos)
	std::vector<gc::SeedRec> longSeeds;   // whole-read pass order (by goodness), only with long_pass
	std::vector<gc::FragmentWindow> windows;
	std::vector<LongAln> longAlns;        // final order (the reference's repeated sort by alignmentStart)
	uint64_t longBegin = 0, longTraceBegin = 0, longSeedBegin = 0;
	bool failed = false;
	bool longFailed = false;              // the whole-read pass asserted: no anchors, chain or alignment for this read
	bool capacityExceeded = false;        // a capacity of this library (not of the reference) was exceeded while processing this read
	bool capacityExceededLong = false;    // same, raised by the whole-read pass (its own thread; joined into capacityExceeded after the pass)
	uint64_t slotBegin = 0, fragBegin = 0;
	uint32_t nSeedsR = 0, nWindows = 0;   // seeds of the read (fragment order, at seedBegin) and fragments that hold seeds
	uint64_t nAnchors = 0, nPath = 0, nTrace = 0, anchorBegin = 0, pathBegin = 0, traceBegin = 0, seedBegin = 0, chainBegin = 0;
	StitchedPath stitched;                // chain stitching result
	bool stitchedOnDevice = false;        // its nodes are also in the device's stitch regions
	uint64_t stitchedBegin = 0;
	std::vector<uint32_t> longSelected;   // GreedyLength selection (src/Aligner.cpp:636-639): indices into longAlns
	uint64_t longSelectedBegin = 0;
	int64_t longEditDistance = -1, chainEditDistance = -1;
	// the chained alignment (src/Aligner.cpp:845-897): trace in output coordinates, alignmentStart / alignmentEnd
	std::vector<int32_t> chainTraceNode; std::vector<uint32_t> chainTraceOffset, chainTraceSeqPos; std::vector<uint8_t> chainTraceSwitch;
	uint32_t chainAlnStart = 0, chainAlnEnd = 0;
	bool hasChainAlignment = false, chainWins = false;
	uint64_t chainTraceBegin = 0;
	// back to the state of a fresh record, keeping the vectors' storage: the records live in the gc_stream and are reused by
	// every batch (allocating and destroying 10 k x 6 vectors per batch cost ~10 ms of teardown plus the allocations)
	void reset()
	{
		seeds.clear(); longSeeds.clear(); windows.clear(); longAlns.clear(); longSelected.clear();
		stitched.nodes.clear(); stitched.firstOffset = stitched.lastOffset = 0; stitched.cells = 0;
		longBegin = longTraceBegin = longSeedBegin = 0;
		failed = longFailed = capacityExceeded = capacityExceededLong = false;
		slotBegin = fragBegin = 0;
		nSeedsR = nWindows = 0;
		nAnchors = nPath = nTrace = anchorBegin = pathBegin = traceBegin = seedBegin = chainBegin = 0;
		stitchedBegin = longSelectedBegin = 0;
		longEditDistance = chainEditDistance = -1;
		chainTraceNode.clear(); chainTraceOffset.clear(); chainTraceSeqPos.clear(); chainTraceSwitch.clear();
		chainAlnStart = chainAlnEnd = 0;
		hasChainAlignment = chainWins = false;
		chainTraceBegin = 0;
	}
};

struct EditDistanceRun {
	hipStream_t streams[8] {};         // one per kernel class: three pairs per wave, two pairs per wave, units of 1, 2, 4, 8, 16 blocks, a workgroup per pair (the whole matrix)
	uint32_t begin[9] {};              // the classes' ranges in the grouped order
	std::vector<uint32_t> lens;        // the pairs' read lengths, grouped order (the reruns choose their kernel by them)
	hipEvent_t ready = nullptr;
	std::vector<uint32_t> perm;        // position in the grouped order -> original pair index
	std::vector<int64_t> grouped;      // results in grouped order (pinned not needed: small)
	~EditDistanceRun() { for (auto& q : streams) if (q) (void)hipStreamDestroy(q); if (ready) (void)hipEventDestroy(ready); }
};

inline constexpr int LONG_EVENT_RING = 8;
#ifndef GC_LONG_PLAN_DEFAULT
#define GC_LONG_PLAN_DEFAULT "1"   // candidates per read and round of the whole-read pass (GC_LONG_PLAN; see runLongGroup)
#endif
struct gc_stream {
	std::vector<ReadGlue> glue;   // per-read host records of the batch in flight (storage reused)
	int device = 0;             // the device the stream was created on; gc_align_batch selects it for the calling thread
	hipStream_t stream = nullptr;
	hipEvent_t ev[12] {};
	hipEvent_t fragEv[16] {};   // r5: a (begin, end) pair around each of the lazy rounds' k_extend launches [0..7] and k_build_anchors launches [8..15]: kernel_us[1] / [2] are sums of exactly those
	DeviceBuffer tmp, matches, readMatchOff, readMatchCount, cursors, readSeeds, fragFirstSeed, longRetryList, extLists, fragItems, fragRetryList, fragClaims, pendingFrags, fragNext, roundCounts, work, results, scratch, scratchRetry, tracePool, frags, fragSeeds, anchors, fragStatus, fragExtended, readTies, pathPool, jobs, chainOut, chainLen, chainScore, chainStatus, chainScratch, counters;
	PinnedBuffer hFragDeclined, hMatches, hReadSeeds, hFragFirstSeed, hFrags, hJobs, hAnchors, hFragStatus, hFragExtended, hReadTies, hChainOut, hChainLen, hChainScore, hChainStatus, hPathPool, hSmall;
	// whole-read pass: runs on its own stream, concurrently with the fragment kernels
	hipStream_t longStream = nullptr;
	hipEvent_t longEv[2] {};
	hipStream_t splitStream = nullptr;   // experiments build (GC_LONG_SPLIT): the multi-lane share of a whole-read round
	hipEvent_t splitEv[2] {};
	DeviceBuffer edPathNodes, edJobs, edLetters, edLettersLen, edPairs, edOut;
	PinnedBuffer hEdPathNodes, hEdJobs, hEdPairs, hEdOut;
	DeviceBuffer outJobs, outRecs, outOffsets, outMapSizes, outPathText, outCigarText, outVgBytes, outTotals;   // output encoding on the device (gc_output.hip)
	PinnedBuffer hOutJobs, hOutRecs, hOutOffsets, hOutPathText, hOutCigarText, hOutVgBytes, hOutTotals;
	DeviceBuffer stitchSlotOf, stitchRegions, stitchNodes, stitchInfo, stitchCursor, stitchSpill;   // chain stitching on the device (gc_stitch.hip)
	DeviceBuffer anchorPerRead, anchorSlotEnd, anchorOff, anchorDense;   // the result's dense anchor arrays made on the device (gc_results.hip)
	PinnedBuffer hAnchorPerRead, hAnchorOff, hAnchorDense;
	PinnedBuffer hStitchNodes, hStitchInfo, hStitchCursor;
	EditDistanceRun edChainRun;
	DeviceBuffer edPathJobs, edPathOps, edPathLen, edPathScratch;   // alignment path of the chained alignment (gc_edpath.hip)
	PinnedBuffer hEdPathJobs, hEdPathOps, hEdPathLen;
	// whole-read decision (selection + edit distance of the best alignment)
	struct LongDecision {
		PinnedBuffer hJobs, hPairs, hOut;
		DeviceBuffer jobs, letters, lettersLen, pairs, out;
		EditDistanceRun run;
		std::vector<uint32_t> pairRead;
		uint32_t nPairs = 0;
	} edLong[2];
	uint64_t longCellsPerBase = 4;           // merged-trace cells per read base the whole-read pass reserves (10 kb ONT-like reads use 1.1, 50 kb CLR-like reads on a genome with repeats 9-10; grows by what a batch asks for)
	double fragShare = 0;                     // fragment extensions per read base in the stream's last batch (longTokenCount)
	uint64_t batchesDone = 0;                 // batches this stream has finished (a second whole-read token is only taken from the second batch on: the first sizes the stream's buffers)
	bool poolsRerun = false;                  // the last batch ran its fragment pipeline again with larger pools: the next one gives back what that overshot
	double traceCellsPerSlot = 0, pathWordsPerSlot = 0;   // what this stream's batches have used of the fragment pipeline's trace pool / anchor path pool per anchor slot (0: no batch yet)
	std::vector<hipStream_t> groupStreams;   // read groups of the whole-read pass run their round loops concurrently
	std::vector<hipEvent_t> groupEvents;     // 2 * LONG_EVENT_RING per group
	DeviceBuffer longSeeds, longJobs, longAlns, longResults, longScratch, longCells, longCursor, longJobsFallback, longResultsFallback, longScratchFallback;
	DeviceBuffer gluePerRead, glueCursors, glueOut, glueSeedCap, glueSeedOff, glueWinCapOff, glueU32[8], glueSort, gluePos, glueWin;   // seed glue on the device (gc_seedglue.hip)
	PinnedBuffer hGlueOut, hGlueWinCapOff, hGlueSmall;
	DeviceBuffer longState, longWork, longWorkResults, longRoundTrace, longCandSeed, longWorkLen, longOrder, longRoundInfo;
	PinnedBuffer hLongRoundInfo;
	PinnedBuffer hLongSeeds, hLongJobs, hLongAlns, hLongResults, hLongSmall, hLongCells;
	// every device allocation of the stream with its size (GC_DEBUG_TIMES: "[gc mem]" lines; the whole-read decision's and the edit-distance runs' own buffers are listed by their owners)
	template <typename F> void forEachDeviceBuffer(F f) const
	{
		f("tmp", tmp.bytes); f("matches", matches.bytes); f("readMatchOff", readMatchOff.bytes); f("readMatchCount", readMatchCount.bytes); f("cursors", cursors.bytes); f("readSeeds", readSeeds.bytes);
		f("fragFirstSeed", fragFirstSeed.bytes); f("longRetryList", longRetryList.bytes); f("extLists", extLists.bytes); f("pendingFrags", pendingFrags.bytes); f("fragNext", fragNext.bytes);
		f("roundCounts", roundCounts.bytes); f("fragItems", fragItems.bytes); f("fragRetryList", fragRetryList.bytes); f("work", work.bytes); f("results", results.bytes); f("scratch", scratch.bytes); f("scratchRetry", scratchRetry.bytes); f("tracePool", tracePool.bytes);
		f("frags", frags.bytes); f("fragSeeds", fragSeeds.bytes); f("anchors", anchors.bytes); f("fragStatus", fragStatus.bytes); f("fragExtended", fragExtended.bytes); f("readTies", readTies.bytes);
		f("pathPool", pathPool.bytes); f("jobs", jobs.bytes); f("chainOut", chainOut.bytes); f("chainLen", chainLen.bytes); f("chainScore", chainScore.bytes); f("chainStatus", chainStatus.bytes);
		f("chainScratch", chainScratch.bytes); f("counters", counters.bytes); f("edPathNodes", edPathNodes.bytes); f("edJobs", edJobs.bytes); f("edLetters", edLetters.bytes);
		f("edLettersLen", edLettersLen.bytes); f("edPairs", edPairs.bytes); f("edOut", edOut.bytes); f("outJobs", outJobs.bytes); f("outRecs", outRecs.bytes); f("outOffsets", outOffsets.bytes);
		f("outMapSizes", outMapSizes.bytes); f("outPathText", outPathText.bytes); f("outCigarText", outCigarText.bytes); f("outVgBytes", outVgBytes.bytes); f("outTotals", outTotals.bytes);
		f("stitchSlotOf", stitchSlotOf.bytes); f("stitchRegions", stitchRegions.bytes); f("stitchNodes", stitchNodes.bytes); f("stitchInfo", stitchInfo.bytes); f("stitchCursor", stitchCursor.bytes); f("stitchSpill", stitchSpill.bytes);
		f("anchorPerRead", anchorPerRead.bytes); f("anchorSlotEnd", anchorSlotEnd.bytes); f("anchorOff", anchorOff.bytes); f("anchorDense", anchorDense.bytes);
		f("edPathJobs", edPathJobs.bytes); f("edPathOps", edPathOps.bytes); f("edPathLen", edPathLen.bytes); f("edPathScratch", edPathScratch.bytes); f("longSeeds", longSeeds.bytes);
		f("longJobs", longJobs.bytes); f("longAlns", longAlns.bytes); f("longResults", longResults.bytes); f("longScratch", longScratch.bytes); f("longCells", longCells.bytes);
		f("longCursor", longCursor.bytes); f("longJobsFallback", longJobsFallback.bytes); f("longResultsFallback", longResultsFallback.bytes); f("longScratchFallback", longScratchFallback.bytes);
		f("gluePerRead", gluePerRead.bytes); f("glueCursors", glueCursors.bytes); f("glueOut", glueOut.bytes); f("glueSeedCap", glueSeedCap.bytes); f("glueSeedOff", glueSeedOff.bytes);
		f("glueWinCapOff", glueWinCapOff.bytes); f("glueU32[0]", glueU32[0].bytes); f("glueU32[1]", glueU32[1].bytes); f("glueU32[2]", glueU32[2].bytes); f("glueU32[3]", glueU32[3].bytes);
		f("glueU32[4]", glueU32[4].bytes); f("glueU32[5]", glueU32[5].bytes); f("glueU32[6]", glueU32[6].bytes); f("glueU32[7]", glueU32[7].bytes); f("glueSort", glueSort.bytes);
		f("gluePos", gluePos.bytes); f("glueWin", glueWin.bytes); f("longState", longState.bytes); f("longWork", longWork.bytes); f("longWorkResults", longWorkResults.bytes);
		f("longRoundTrace", longRoundTrace.bytes); f("longCandSeed", longCandSeed.bytes); f("longWorkLen", longWorkLen.bytes); f("longOrder", longOrder.bytes); f("longRoundInfo", longRoundInfo.bytes);
		for (int k = 0; k < 2; k++) { f("edLong.letters", edLong[k].letters.bytes); f("edLong.jobs+pairs+out", edLong[k].jobs.bytes + edLong[k].lettersLen.bytes + edLong[k].pairs.bytes + edLong[k].out.bytes); }
	}
	~gc_stream()
	{
		for (auto& e : ev) if (e) (void)hipEventDestroy(e);
		for (auto& e : fragEv) if (e) (void)hipEventDestroy(e);
		for (auto& e : longEv) if (e) (void)hipEventDestroy(e);
		for (auto& e : groupEvents) if (e) (void)hipEventDestroy(e);
		for (auto& q : groupStreams) if (q) (void)hipStreamDestroy(q);
		if (stream) (void)hipStreamDestroy(stream);
		if (longStream) (void)hipStreamDestroy(longStream);
		if (splitStream) (void)hipStreamDestroy(splitStream);
		for (auto& e : splitEv) if (e) (void)hipEventDestroy(e);
	}
};

// ----------------------------------------------------------------------------------------------------
// set of bases (A=1,C=2,G=4,T=8) a read character can stand for; 0 = matches nothing.
// reference: characterMatch / ambiguousMatch, src/GraphAlignerCommon.h:190-297
inline void buildIupacTable(uint8_t* t)
{
	memset(t, 0, 256);
	auto set = [&](const char* chars, uint8_t m) { for (const char* c = chars; *c; c++) t[(uint8_t)*c] = m; };
	set("Aa", 1); set("Cc", 2); set("Gg", 4); set("TtUu", 8);
	set("Rr", 1 | 4); set("Yy", 2 | 8); set("Kk", 4 | 8); set("Mm", 1 | 2); set("Ss", 2 | 4); set("Ww", 1 | 8);
	set("Bb", 2 | 4 | 8); set("Dd", 1 | 4 | 8); set("Hh", 1 | 2 | 8); set("Vv", 1 | 2 | 4); set("Nn", 15);
}

inline void uploadGraph(gc_graph* G)
{
	const gc::AlignmentGraph& h = G->host;
	size_t n = h.NodeSize();
	if (n >= 0xfffffff0ull) throw std::runtime_error("graph too large for 32-bit node ids");
	// Every staging array is uploaded and released before the next one is made: at 3.1 Gbp the flattened copies are ~90 GB together (28 B per graph base) beside a host graph
	// of ~230 GB - too much for a 300 GiB host when they all live until the function returns, as they did up to r5
	DGraph& d = G->dev;
	d.nNodes = (uint32_t)n;
	d.firstAmbiguous = (uint32_t)h.firstAmbiguous;
	int maxId = -1;
	{
		std::vector<uint8_t> nodeLength(n);
		std::vector<uint32_t> nodeOffset(n);
		std::vector<int32_t> nodeIDs(n);
		for (size_t i = 0; i < n; i++) {
			nodeLength[i] = (uint8_t)h.nodeLength[i];
			nodeOffset[i] = (uint32_t)h.nodeOffset[i];
			nodeIDs[i] = h.nodeIDs[i];
			maxId = std::max(maxId, h.nodeIDs[i]);
		}
		d.nodeLength = G->up(nodeLength);
		d.nodeOffset = G->up(nodeOffset);
		d.nodeIDs = G->up(nodeIDs);
	}
	{
		std::vector<uint32_t> componentNumber(n), componentMap(n), topoId(n);
		for (size_t i = 0; i < n; i++) {
			componentNumber[i] = (uint32_t)h.componentNumber[i];
			componentMap[i] = (uint32_t)h.component_map[i];
			topoId[i] = (uint32_t)h.topo_ids[h.component_map[i]][h.component_idx[i]];
		}
		d.componentNumber = G->up(componentNumber);
		d.componentMap = G->up(componentMap);
		d.topoId = G->up(topoId);
	}
	{
		std::vector<uint64_t> nodeSeq(2 * h.firstAmbiguous);
		for (size_t i = 0; i < h.firstAmbiguous; i++) { nodeSeq[2 * i] = h.nodeSequences[i][0]; nodeSeq[2 * i + 1] = h.nodeSequences[i][1]; }
		d.nodeSeq = G->up(nodeSeq);
	}
	{
		std::vector<uint64_t> ambSeq(4 * (n - h.firstAmbiguous));
		for (size_t i = h.firstAmbiguous; i < n; i++) {
			const gc::AmbiguousSeq& s = h.ambiguousNodeSequences[i - h.firstAmbiguous];
			size_t at = 4 * (i - h.firstAmbiguous);
			ambSeq[at] = s.A; ambSeq[at + 1] = s.C; ambSeq[at + 2] = s.G; ambSeq[at + 3] = s.T;
		}
		d.ambSeq = G->up(ambSeq);
	}
	auto csr = [&](const std::vector<std::vector<size_t>>& adj, const uint32_t*& devOff, const uint32_t*& devFlat) {
		std::vector<uint32_t> off(n + 1, 0), flat;
		for (size_t i = 0; i < n; i++) off[i + 1] = off[i] + (uint32_t)adj[i].size();
		flat.reserve(off[n]);
		for (size_t i = 0; i < n; i++) for (size_t v : adj[i]) flat.push_back((uint32_t)v);
		devOff = G->up(off); devFlat = G->up(flat);
	};
	csr(h.inNeighbors, d.inOff, d.inAdj);
	csr(h.outNeighbors, d.outOff, d.outAdj);
	const size_t nB = (size_t)maxId + 1;
	{
		std::vector<uint32_t> origSize(nB, 0), lookupOff(nB + 1, 0), lookup;
		lookup.reserve(n);
		for (size_t id = 0; id < nB; id++) {
			const bool known = h.nodeLookup.contains((int)id);
			lookupOff[id + 1] = lookupOff[id] + (known ? (uint32_t)h.nodeLookup.at((int)id).size() : 0u);
			if (known) {
				origSize[id] = (uint32_t)h.originalNodeSize.at((int)id);
				for (size_t s : h.nodeLookup.at((int)id)) lookup.push_back((uint32_t)s);
			}
		}
		for (size_t id = 0; id < nB; id++)
			for (uint32_t k = lookupOff[id]; k < lookupOff[id + 1]; k++)
				if (h.nodeOffset[lookup[k]] != 64ull * (k - lookupOff[id])) throw std::runtime_error("split nodes are not 64-aligned chunks of their original node");
		d.origSize = G->up(origSize); d.lookupOff = G->up(lookupOff); d.lookup = G->up(lookup);
		G->hOrigSize = std::move(origSize); G->hLookupOff = std::move(lookupOff); G->hLookup = std::move(lookup);
	}
	{
		// MPC index, flattened to global node ids. The per-node path lists stay on the host while the backward links - 12 bytes each, 4.5 per node - are made and sent in slices of
		// nodes: all of them at once were 37 GB of staging at 3.1 Gbp, the top of the host's peak (235 GiB of 300)
		std::vector<uint32_t> pathsOff(n + 1, 0), pathsFlat, pathsPos, backOff(n + 1, 0), mpcWidth(h.mpc.size());
		for (size_t c = 0; c < h.mpc.size(); c++) { mpcWidth[c] = (uint32_t)h.mpc[c].size(); G->maxMpcWidth = std::max(G->maxMpcWidth, mpcWidth[c]); }
		size_t nPaths = 0, nBack = 0;
		for (size_t i = 0; i < n; i++) { const size_t c = h.component_map[i], x = h.component_idx[i]; nPaths += h.paths[c][x].size(); nBack += h.backwards[c][x].size(); }
		if (nPaths >= 0xffffffffull || nBack >= 0xffffffffull) throw std::runtime_error("MPC index too large for 32-bit offsets");
		pathsFlat.reserve(nPaths);
		for (size_t i = 0; i < n; i++) {
			size_t c = h.component_map[i], x = h.component_idx[i];
			for (size_t k : h.paths[c][x]) pathsFlat.push_back((uint32_t)k);
			pathsOff[i + 1] = (uint32_t)pathsFlat.size();
			G->maxPathsPerNode = std::max(G->maxPathsPerNode, pathsOff[i + 1] - pathsOff[i]);
			backOff[i + 1] = backOff[i] + (uint32_t)h.backwards[c][x].size();
			G->maxBackPerNode = std::max(G->maxBackPerNode, backOff[i + 1] - backOff[i]);
		}
		// position of every node on every path through it (paths[v] lists path ids in ascending order, and a path visits
		// its nodes in order, so walking path k in order fills the (v,k) entries)
		pathsPos.assign(pathsFlat.size(), 0);
		for (size_t c = 0; c < h.mpc.size(); c++)
			for (size_t k = 0; k < h.mpc[c].size(); k++)
				for (size_t j = 0; j < h.mpc[c][k].size(); j++) {
					size_t node = h.mpc[c][k][j];
					for (uint32_t e = pathsOff[node]; e < pathsOff[node + 1]; e++) if (pathsFlat[e] == k) pathsPos[e] = (uint32_t)j;   // last visit wins, as in last2reach (:1340-1345)
				}
		auto posOf = [&](uint32_t node, uint32_t k) -> uint32_t {
			for (uint32_t e = pathsOff[node]; e < pathsOff[node + 1]; e++) if (pathsFlat[e] == k) return pathsPos[e];
			throw std::runtime_error("MPC index: node not on path");
		};
		uint32_t *dBackNode = nullptr, *dBackPath = nullptr, *dBackPos = nullptr;
		for (uint32_t** p : { &dBackNode, &dBackPath, &dBackPos }) { HIP_CHECK(hipMalloc((void**)p, std::max<size_t>(nBack, 1) * sizeof(uint32_t))); G->allocations.push_back(*p); }
		size_t sliceLinks = (size_t)64 << 20;
		if (const char* env = getenv("GC_TEST_UPLOAD_SLICE")) sliceLinks = (size_t)std::max(1, atoi(env));   // test hook: many small slices
		std::vector<uint32_t> backNode, backPath, backPos;
		for (size_t first = 0; first < n;) {
			size_t last = first;
			while (last < n && (last == first || (size_t)(backOff[last + 1] - backOff[first]) <= sliceLinks)) last++;
			backNode.clear(); backPath.clear(); backPos.clear();
			for (size_t i = first; i < last; i++) {
				const size_t c = h.component_map[i], x = h.component_idx[i];
				for (const auto& b : h.backwards[c][x]) {
					const uint32_t node = (uint32_t)h.component_ids[c][b.first], path = (uint32_t)b.second;
					backNode.push_back(node); backPath.push_back(path); backPos.push_back(posOf(node, path));
				}
			}
			if (backNode.size() != (size_t)(backOff[last] - backOff[first])) throw std::runtime_error("MPC index: backward links counted differently");
			if (!backNode.empty()) {
				HIP_CHECK(hipMemcpy(dBackNode + backOff[first], backNode.data(), backNode.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
				HIP_CHECK(hipMemcpy(dBackPath + backOff[first], backPath.data(), backPath.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
				HIP_CHECK(hipMemcpy(dBackPos + backOff[first], backPos.data(), backPos.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
			}
			first = last;
		}
		d.pathsOff = G->up(pathsOff); d.paths = G->up(pathsFlat); d.pathsPos = G->up(pathsPos);
		d.backOff = G->up(backOff); d.backNode = dBackNode; d.backPath = dBackPath; d.backPos = dBackPos;
		d.mpcWidth = G->up(mpcWidth);
	}
	{
		std::vector<uint32_t> chainNumber(n);
		std::vector<uint64_t> chainApproxPos(n);
		for (size_t i = 0; i < n; i++) {
			if (h.chainNumber[i] >= 0xffffffffull) throw std::runtime_error("too many chains for 32-bit chain numbers");
			chainNumber[i] = (uint32_t)h.chainNumber[i];
			chainApproxPos[i] = (uint64_t)h.chainApproxPos[i];
		}
		d.chainNumber = G->up(chainNumber);
		d.chainApproxPos = G->up(chainApproxPos);
	}
	{
		// the names the output encoders print (OriginalNodeName; empty: they print id / 2), by bigraph node id
		std::vector<uint32_t> nameOff(nB + 1, 0);
		std::vector<char> nameBytes;
		{
			size_t total = 1;
			for (size_t id = 0; id < nB; id++) if (const std::string* name = h.originalNodeName.find((int)id)) total += name->size();
			nameBytes.reserve(total);
		}
		for (size_t id = 0; id < nB; id++) {
			const std::string* name = h.originalNodeName.find((int)id);
			if (name) nameBytes.insert(nameBytes.end(), name->begin(), name->end());
			if (nameBytes.size() >= 0xffffffffull) throw std::runtime_error("node names exceed 4 GB");
			nameOff[id + 1] = (uint32_t)nameBytes.size();
		}
		nameBytes.push_back(0);
		G->devNames.nameOff = G->up(nameOff);
		G->devNames.nameBytes = G->up(nameBytes);
	}
	{
		NodeRec* recs = nullptr;
		HIP_CHECK(hipMalloc((void**)&recs, std::max<size_t>(1, n) * sizeof(NodeRec)));
		G->allocations.push_back(recs);
		launchBuildNodeRecs(nullptr, d, recs);
		HIP_CHECK(hipGetLastError());
		HIP_CHECK(hipDeviceSynchronize());
		d.nodeRec = recs;
	}
	CorrectnessTables t;
	buildCorrectnessTables(t);
	HIP_CHECK(hipMalloc((void**)&G->devTables, sizeof(t)));
	G->allocations.push_back(G->devTables);
	HIP_CHECK(hipMemcpy(G->devTables, &t, sizeof(t), hipMemcpyHostToDevice));
	uint8_t iupac[256];
	buildIupacTable(iupac);
	HIP_CHECK(hipMalloc((void**)&G->devIupac, 256));
	G->allocations.push_back(G->devIupac);
	HIP_CHECK(hipMemcpy(G->devIupac, iupac, 256, hipMemcpyHostToDevice));
}

// One capacity: the GC_* environment variable (experiments, test hooks) wins over gc_params::capacity, 0 there means automatic.
inline int64_t capacityOr(const char* envName, int64_t param, int64_t automatic)
{
	if (const char* env = getenv(envName)) return atoll(env);
	return param != 0 ? param : automatic;
}

template <typename F>
inline int guarded(F&& f)
{
	try {
		return f();
	} catch (const DeviceError& e) {
		return fail(GC_ERR_DEVICE, e.what());
	} catch (const std::exception& e) {
		return fail(GC_ERR_INTERNAL, e.what());
	}
}

inline void requireDevice()
{
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0) throw DeviceError("no HIP device available: the product path has no CPU fallback");
}

namespace gcrt {

// Chain stitching, reference: src/Aligner.cpp:754-822 (+ pathToTrace :409-424, getChainPath src/AlignmentGraph.cpp:1866-1916).
// The chain's anchor paths are concatenated; consecutive anchors that are not adjacent are bridged by the fewest-hops
// path (bounded BFS); where no bridge exists within --colinear-gap the path is cut and the longest piece (most graph
// bases, the size of the reference's pathToTrace vector) is kept. `slots` = this read's kept anchors in anchor-index
// order. The piece is returned as its node path plus the offsets of its first and last base; pathToTrace's cell list
// (one entry per base) follows from those and is not materialised here.
inline void stitchChain(const gc::AlignmentGraph& graph, long long colinearGap, const uint32_t* chain, uint32_t chainLen, const uint32_t* slots,
	const AnchorRec* anchors, const uint32_t* pathPool, StitchedPath& longest)
{
	std::vector<size_t> posPath;
	std::unordered_set<size_t> nodes;
	size_t firstNodeOffset = 0, lastNodeOffset = 0;
	longest = StitchedPath();
	auto keepIfLonger = [&]() {
		uint64_t cells = 0;   // size of pathToTrace(posPath, firstNodeOffset, lastNodeOffset), src/Aligner.cpp:409-424
		for (size_t node : posPath) {
			size_t S = 0, L = graph.NodeLength(node);
			if (node == posPath[0]) S = firstNodeOffset;
			else if (node == posPath.back()) L = lastNodeOffset + 1;
			cells += L > S ? L - S : 0;
		}
		if (longest.cells < cells) {
			longest.nodes.assign(posPath.begin(), posPath.end());
			longest.firstOffset = (uint32_t)firstNodeOffset;
			longest.lastOffset = (uint32_t)lastNodeOffset;
			longest.cells = cells;
		}
	};
	for (uint32_t c = 0; c < chainLen; c++) {
		const AnchorRec& a = anchors[slots[chain[c]]];
		const uint32_t* apath = pathPool + a.pathOff;
		if (posPath.empty()) {
			posPath.assign(apath, apath + a.pathLen);
			firstNodeOffset = a.firstOffset;
			lastNodeOffset = a.lastOffset;
			for (size_t j : posPath) nodes.insert(j);
		} else {
			bool gap = apath[0] == posPath.back() && colinearGap != -1 && (long long)a.firstOffset - (long long)lastNodeOffset > colinearGap + 1;
			std::vector<size_t> bridge;
			if (!nodes.count(apath[0]) && posPath.back() != a.firstNode) {
				long long gapLimit = colinearGap;
				if (gapLimit != -1) gapLimit -= (long long)a.firstOffset + (long long)(graph.NodeLength(posPath.back()) - (long long)lastNodeOffset - 1);
				bridge = graph.getChainPath(posPath.back(), a.firstNode, gapLimit);
				if (bridge.empty()) gap = true;
			}
			if (gap) {
				keepIfLonger();
				nodes.clear();
				posPath.clear();
				firstNodeOffset = a.firstOffset;
			} else {
				for (size_t j : bridge) if (!nodes.count(j)) { nodes.insert(j); posPath.push_back(j); }
			}
			for (uint32_t k = 0; k < a.pathLen; k++) { size_t j = apath[k]; if (!nodes.count(j)) { nodes.insert(j); posPath.push_back(j); } }
			lastNodeOffset = a.lastOffset;
		}
	}
	if (!posPath.empty()) keepIfLonger();
}

// Exact-match bit vectors of a read for the NW kernel: [A,C,G,T][words], bit i set when base i is exactly that letter.
inline void buildEqMasks(const char* seq, uint64_t len, uint64_t words, uint64_t* out)
{
	for (uint64_t i = 0; i < len; i++) {
		int b = seq[i] == 'A' ? 0 : seq[i] == 'C' ? 1 : seq[i] == 'G' ? 2 : seq[i] == 'T' ? 3 : -1;
		if (b >= 0) out[(uint64_t)b * words + (i >> 6)] |= 1ull << (i & 63);
	}
}

// Runs the NW kernel over `pairs`. The rows-per-lane unit (1, 2, 4, 8, 16 blocks) a pair needs follows from its band
// half-width k and its read length (gc_editdist.hip); pairs are grouped by unit, every group runs on its own stream (a
// group of a few wide-band pairs is one long-running wave each and would otherwise hold up the others), and pairs
// whose k had to grow past their unit's limit are rerun with the next unit.
// hPairs/hOut: pinned host staging; dPairs/dOut: device arrays of at least nPairs elements; readLen: host read lengths.
inline uint32_t editDistanceUnit(uint32_t k, uint32_t readLen)
{
	uint32_t unit = 1;
	while (unit < 16 && k >= editDistanceMaxK(unit) && (readLen + 64 * unit - 1) / (64 * unit) > 64) unit *= 2;
	return unit;
}
// Streams of the fragment pipeline / the edit distances (role 0) and of the whole-read rounds (role 1). GC_STREAM_PRIORITY=frag|long raises one
// side's queue priority (experiment): the whole-read kernel holds 7 of a SIMD's 8 wave slots for milliseconds per wave, so
// whatever shares the device with it runs on what is left.
inline void createStream(hipStream_t* q, int role)
{
	static const int mode = []() { const char* e = expEnv("GC_STREAM_PRIORITY"); return !e ? 0 : !strcmp(e, "frag") ? 1 : !strcmp(e, "long") ? 2 : 0; }();
	int least = 0, greatest = 0;
	if (mode && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest) {
		const bool high = (mode == 1 && role == 0) || (mode == 2 && role == 1);
		HIP_CHECK(hipStreamCreateWithPriority(q, hipStreamNonBlocking, high ? greatest : least));
		return;
	}
	HIP_CHECK(hipStreamCreateWithFlags(q, hipStreamNonBlocking));
}

inline void launchEditDistances(EditDistanceRun& run, hipStream_t stream, EdPair* hPairs, int64_t* hOut, uint32_t nPairs, EdPair* dPairs, int64_t* dOut, const EdRead* dReads, const char* dBases,
	const uint64_t* dEqMasks, const char* dLetters, const uint32_t* dLettersLen, const std::function<uint32_t(uint32_t)>& readLen, bool kIsBound = false)
{
	if (!nPairs) return;
	if (!run.ready) HIP_CHECK(hipEventCreateWithFlags(&run.ready, hipEventDisableTiming));
	// (a class's stream is created when the class is first used: the batch's streams share the device's 16 hardware queues, and on cfg2 only the
	// two-pairs-per-wave class and the one-block class ever hold pairs)
	std::vector<uint32_t> cls(nPairs), lenOf(nPairs);
	uint32_t count[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
	uint32_t blockThreads = 64;        // class 7: threads of a workgroup = the 64-row blocks of the class's longest read
	uint32_t* begin = run.begin;
	static const bool halfWaves = !(getenv("GC_ED_HALF") && atoi(getenv("GC_ED_HALF")) == 0);
	static const bool thirdWaves = halfWaves && !(getenv("GC_ED_THIRD") && atoi(getenv("GC_ED_THIRD")) == 0);
	for (uint32_t i = 0; i < nPairs; i++) {
		const uint32_t len = readLen(hPairs[i].read);
		lenOf[i] = len;
		const uint32_t askedK = hPairs[i].k;   // the caller's band, before the team classes widen it to what their lanes hold anyway (class 7 below is judged on this one: ADVICE r4)
		uint32_t unit = editDistanceUnit(hPairs[i].k, len), c = 0;
		while ((1u << c) < unit) c++;
		c += 2;                                                               // classes 2..6: one pair per wave, units of 1..16 blocks
		if (halfWaves && unit == 1 && hPairs[i].k < editDistanceMaxK(0) && len <= 131072) {   // class 1: two pairs per wave (small first band)
			c = 1;
			// class 0 (r4): three pairs per wave, bands below 1290 - for pairs whose k is a bound (a whole-read pair's k comes from the alignment itself: one sweep, always enough).
			// A chain pair's k is a guess; tried there whenever length difference + 10 % of the shorter sequence fit, 45 % of cfg2's chain pairs (distances 1 100-1 300) came back
			// for a second sweep and the two kernels together took 352 ms per nine batches against 311 (`gpurun_out/r4_ring`): chain pairs stay with two per wave
			if (thirdWaves && kIsBound && len <= 65536 && hPairs[i].k < editDistanceTeamMaxK(3)) c = 0;
			// a sweep of these kernels takes columns + units steps whatever the band, as long as the band fits the team's lanes - so the first guess may as well be the widest band that
			// does (any k >= the distance gives the distance): a chain pair whose guess (length difference + 14 %) was a little short used to pay a failed sweep here, a second
			// failed sweep with the same guess in the one-pair-per-wave kernel and a third with the doubled band (r3)
			hPairs[i].k = c == 0 ? editDistanceTeamMaxK(3) - 1 : std::max(hPairs[i].k, editDistanceMaxK(0) - 1);
		}
		// class 7 (r4): a band of half the read or more covers most of the matrix - a chain whose path spells a fraction of its read, a whole-read alignment of a sliver: the pair
		// gets a workgroup with one thread per 64-row block and the whole matrix (exact, no retry) instead of one wave with up to sixteen blocks per lane and step
		static const bool blockPairs = !(getenv("GC_ED_BLOCK") && atoi(getenv("GC_ED_BLOCK")) == 0);
		if (blockPairs && len >= 1 && len <= editDistanceBlockMaxRows() && 2ull * askedK >= len) { hPairs[i].k = askedK; c = 7; blockThreads = std::max(blockThreads, (len + 63u) / 64u); }
		cls[i] = c;
		count[c]++;
	}
	begin[0] = 0;
	for (int c = 0; c < 8; c++) begin[c + 1] = begin[c] + count[c];
	run.perm.resize(nPairs);
	{
		uint32_t at[8] = { begin[0], begin[1], begin[2], begin[3], begin[4], begin[5], begin[6], begin[7] };
		std::vector<EdPair> grouped(nPairs);
		run.lens.resize(nPairs);
		for (uint32_t i = 0; i < nPairs; i++) { grouped[at[cls[i]]] = hPairs[i]; run.lens[at[cls[i]]] = lenOf[i]; run.perm[at[cls[i]]++] = i; }
		memcpy(hPairs, grouped.data(), (size_t)nPairs * sizeof(EdPair));   // hPairs is now in grouped order
	}
	HIP_CHECK(hipMemcpyAsync(dPairs, hPairs, (size_t)nPairs * sizeof(EdPair), hipMemcpyHostToDevice, stream));
	HIP_CHECK(hipEventRecord(run.ready, stream));
	for (int c = 0; c < 8; c++) {
		if (!count[c]) continue;
		if (!run.streams[c]) createStream(&run.streams[c], 0);
		HIP_CHECK(hipStreamWaitEvent(run.streams[c], run.ready, 0));
		if (c == 7) launchEditDistanceBlock(run.streams[c], blockThreads, dPairs + begin[c], count[c], dReads, dBases, dEqMasks, dLetters, dLettersLen, dOut + begin[c]);
		else if (c < 2) launchEditDistanceTeam(run.streams[c], c == 0 ? 3u : 2u, dPairs + begin[c], count[c], dReads, dBases, dEqMasks, dLetters, dLettersLen, dOut + begin[c]);
		else launchEditDistance(run.streams[c], 1u << (c - 2), dPairs + begin[c], count[c], dReads, dBases, dEqMasks, dLetters, dLettersLen, dOut + begin[c]);
		HIP_CHECK(hipMemcpyAsync(hOut + begin[c], dOut + begin[c], (size_t)count[c] * sizeof(int64_t), hipMemcpyDeviceToHost, run.streams[c]));
	}
	if (getenv("GC_DEBUG_TIMES")) fprintf(stderr, "[gc times] edit distance classes (3 per wave, 2 per wave, units 1..16, workgroup per pair): %u %u %u %u %u %u %u %u\n", count[0], count[1], count[2], count[3], count[4], count[5], count[6], count[7]);
}
inline void finishEditDistances(EditDistanceRun& run, hipStream_t stream, EdPair* hPairs, int64_t* hOut, uint32_t nPairs, EdPair* dPairs, int64_t* dOut, const EdRead* dReads, const char* dBases,
	const uint64_t* dEqMasks, const char* dLetters, const uint32_t* dLettersLen)
{
	if (!nPairs) return;
	for (auto& q : run.streams) if (q) syncStream(q);
	// reruns for the pairs whose band outgrew their unit (grouped order throughout)
	std::vector<uint32_t> todo;
	for (uint32_t i = 0; i < nPairs; i++) {
		if (hOut[i] == -2) todo.push_back(i);   // (-3: the path letters overflowed their slot - stays, the caller flags the read)
	}
	std::vector<EdPair> sub;
	std::vector<int64_t> subOut;
	if (getenv("GC_DEBUG_TIMES")) fprintf(stderr, "[gc times] edit distance reruns: %zu of %u pairs\n", todo.size(), nPairs);
	// unit 0 here: the two-pairs-per-wave kernel for what the three-pairs-per-wave kernel handed back (its pairs come first in the grouped order)
	for (uint32_t unit = 0; unit <= 16 && !todo.empty(); unit = unit ? unit * 2 : 1) {   // (unit 1 for what the two-pairs-per-wave kernel handed back)
		std::vector<uint32_t> later;
		if (unit == 0) {
			std::vector<uint32_t> now;
			for (uint32_t i : todo) (i < run.begin[1] ? now : later).push_back(i);
			todo.swap(now);
			if (todo.empty()) { todo.swap(later); continue; }
		}
		// r4: from unit 4 on a rerun goes to the workgroup-per-pair kernel when the read fits it (<= 65 536 bases): columns + blocks steps whatever the band, exact - a pair that has failed the
		// narrower bands is far from its path, and the wide units give it one wave with 4-16 blocks per lane and step (config 5: 145 ms at unit 8, 304 ms at unit 16 per launch)
		static const bool blockReruns = !(getenv("GC_ED_BLOCK") && atoi(getenv("GC_ED_BLOCK")) == 0);
		if (unit >= 4 && blockReruns) {
			std::vector<uint32_t> wide, rest;
			uint32_t threads = 64;
			for (uint32_t i : todo) {
				if (run.lens[i] >= 1 && run.lens[i] <= editDistanceBlockMaxRows()) { wide.push_back(i); threads = std::max(threads, (run.lens[i] + 63u) / 64u); }
				else rest.push_back(i);
			}
			if (!wide.empty()) {
				sub.resize(wide.size());
				subOut.resize(wide.size());
				for (size_t i = 0; i < wide.size(); i++) sub[i] = hPairs[wide[i]];
				HIP_CHECK(hipMemcpyAsync(dPairs, sub.data(), sub.size() * sizeof(EdPair), hipMemcpyHostToDevice, stream));
				launchEditDistanceBlock(stream, threads, dPairs, (uint32_t)sub.size(), dReads, dBases, dEqMasks, dLetters, dLettersLen, dOut);
				HIP_CHECK(hipMemcpyAsync(subOut.data(), dOut, sub.size() * sizeof(int64_t), hipMemcpyDeviceToHost, stream));
				syncStream(stream);
				for (size_t i = 0; i < wide.size(); i++) hOut[wide[i]] = subOut[i];
			}
			todo.swap(rest);
			if (todo.empty()) break;
		}
		sub.resize(todo.size());
		subOut.resize(todo.size());
		// (what reaches unit U has failed every band below it: the team kernels' limits for U = 0 and 1, the limit of unit U / 2 otherwise - start there, not at the first guess)
		const uint32_t failedBelow = unit == 0 ? editDistanceMaxK(0) - 1 : unit == 1 ? editDistanceMaxK(0) : editDistanceMaxK(unit / 2);
		for (size_t i = 0; i < todo.size(); i++) { sub[i] = hPairs[todo[i]]; sub[i].k = std::max(sub[i].k, failedBelow); }
		HIP_CHECK(hipMemcpyAsync(dPairs, sub.data(), sub.size() * sizeof(EdPair), hipMemcpyHostToDevice, stream));
		launchEditDistance(stream, unit, dPairs, (uint32_t)sub.size(), dReads, dBases, dEqMasks, dLetters, dLettersLen, dOut);
		HIP_CHECK(hipMemcpyAsync(subOut.data(), dOut, sub.size() * sizeof(int64_t), hipMemcpyDeviceToHost, stream));
		syncStream(stream);
		std::vector<uint32_t> next;
		for (size_t i = 0; i < todo.size(); i++) { hOut[todo[i]] = subOut[i]; if (subOut[i] == -2) next.push_back(todo[i]); }
		todo.swap(next);
		todo.insert(todo.end(), later.begin(), later.end());
	}
	// still -2: the band is too wide for the NW kernel (a read longer than 65536 bases more than 32256 edits away from its path); the caller flags the read
	// back to the caller's order
	run.grouped.assign(hOut, hOut + nPairs);
	for (uint32_t i = 0; i < nPairs; i++) hOut[run.perm[i]] = run.grouped[i];
}

template <typename T> inline T* copyOut(const std::vector<T>& v)
{
	T* p = mallocArray<T>(v.size());
	if (!v.empty()) memcpy(p, v.data(), v.size() * sizeof(T));
	return p;
}

} // namespace
