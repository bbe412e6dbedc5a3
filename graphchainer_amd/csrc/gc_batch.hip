// gc_align_batch: the batch pipeline (BatchRun) - seeding, the whole-read pass on its own thread, the fragment pipeline, chaining, stitching, edit distances, the decision, output encoding, result assembly.
#include "gc_runtime.hpp"

extern "C" {

struct BatchRun {
	// ---- the call
	const gc_graph* const G; const gc_seeder* const S; gc_stream* const st; const gc_reads* const R; const gc_params* const P; gc_result* const res;
	const double tCall, cpuCall;
	double cpuJoined;
	const uint64_t n;                        // reads in the batch
	const gc::AlignmentGraph& hg;
	WorkerPool& pool;
	std::vector<ReadGlue>& glue;             // per-read host records, storage reused across batches
	const hipStream_t stream;                // the fragment pipeline's stream (the whole-read pass has st->longStream / st->groupStreams)
	int evIdx = 0;
	double tTotal = 0;
	// ---- what the stages hand on (set by the stage named in the comment of each group)
	// seeds()
	bool deviceGlue = true;
	double tGlue = 0;
	uint64_t nSlots = 0, nFrags = 0, nSeedsTotal = 0, traceBudget = 0;
	uint32_t maxSlotsPerRead = 1, maxWindowSeeds = 0;
	Fragment* frags = nullptr; ReadChainJob* jobs = nullptr;                 // host copies
	FragSeed* readSeeds = nullptr; uint32_t* fragFirstSeed = nullptr;        // host copies (device glue: only with keep_seeds / keep_traces)
	Fragment* dFrags = nullptr; uint32_t* dFragFirstSeed = nullptr; FragSeed* dReadSeeds = nullptr; ReadChainJob* dJobs = nullptr;
	LongSeed* dLongSeeds = nullptr;
	hipEvent_t glueCopied = nullptr;   // device glue: the host copies of frags / seeds have arrived (waited for before the result assembly)
	double tOrdered = 0;
	unsigned long long* hSmall = nullptr;
	unsigned long long* dCursors = nullptr;
	unsigned long long* dCounters = nullptr;
	// prepareWholeReadPass()
	uint32_t maxAlignments = 0;
	LongAln* hLongAlns = nullptr;
	unsigned long long* hLongSmall = nullptr;
	LongReadResult* hLongResults = nullptr;
	LongCell* dLongCells = nullptr;
	uint64_t cellBudget = 0;                      // capacity of the merged-trace cell pool (grown and the pass rerun when a batch overflows it)
	unsigned long long* longScratchOfToken = nullptr;   // the device's shared extension scratch, set by the pass once it holds the token
	// r4: the token is taken when the pass's FIRST extension kernel is about to be queued and given back when the last round's count (zero) has come down: a pass's first
	// init / select / order / publish and the host's wait for the work count (each a launch that queues among the other batches' kernels), and its k_long_finish at the
	// end, no longer sit between two passes' extension kernels (GC_LONG_TOKEN_EARLY=1: around the whole pass, as before)
	std::function<void()> longTokenTake, longTokenDrop;
	uint64_t longScratchWords = 0;
	bool shareLongScratch = false;
	uint32_t longGroups = 0;
	std::vector<double> groupExtendUs; std::vector<uint32_t> groupRounds; std::vector<uint64_t> groupBegin, groupTraceBegin;   // (the pass thread works through pointers into these)
	const gc::EValueModel evalueModel { 0.7 };   // src/Aligner.cpp:478-482 (precise clipping is out of scope)
	struct DecisionPointers { EdPair* hPairs = nullptr; int64_t* hOut = nullptr; EdPair* dPairs = nullptr; int64_t* dOut = nullptr; char* dLetters = nullptr; uint32_t* dLettersLen = nullptr; } decisionPtr[2];
	bool longPostInThread = false;
	// ... the whole-read pass's own buffers and sizes (prepareWholeReadPass sets them; runLongGroup / growLongCells / longFallback / finishLongGroups run on the pass thread)
	LongJob* hJobs = nullptr;
	bool cellPoolPinned = 0;
	uint64_t waveWords = 0;
	LongJob* dLongJobs = nullptr;
	LongAln* dLongAlns = nullptr;
	uint32_t nGroups = 0;
	uint32_t cursorWords = 0;
	unsigned long long* dLongCursor = nullptr;
	LongState* dLongState = nullptr;
	LongWork* dLongWork = nullptr;
	LongWorkResult* dLongWorkResults = nullptr;
	uint32_t* dCandSeed = nullptr;
	uint32_t* dWorkLen = nullptr;
	uint32_t* dRetryList = nullptr;
	uint32_t* dOrder = nullptr;
	unsigned long long* dRoundTrace = nullptr;
	uint64_t scratchLanes = 0;
	hipStream_t ls = nullptr;
	ExtendConfig lcfg;
	unsigned long long* dLongScratchOwn = nullptr;   // (this stream's own scratch: only without the one-pass-at-a-time token)
	uint64_t maxReadLen = 1;
	// startWholeReadPass()
	std::vector<std::thread> longThreads;
	double passThreadCpuMs = 0;   // GC_DEBUG_TIMES: CPU time of the pass thread (read after the join)
	std::vector<std::exception_ptr> longErrors;
	double tLongWall0 = 0;
	std::atomic<double> longWallBeginUs { 0.0 };   // when the pass got the device's token (waiting for another batch's pass is not its own time)
	std::atomic<double> longWallEndUs { 0.0 };
	// fragmentPipeline()
	double tDev = 0;
	uint32_t nWork = 0;
	ExtResult* dResults = nullptr;
	PoolCell* dTrace = nullptr;
	AnchorRec* dAnchors = nullptr;
	uint32_t* dFragStatus = nullptr;
	uint32_t* dFragExtended = nullptr;
	uint64_t traceWorst = 0, pathWorst = 0;   // the pools' worst-case sizes (every slot's extensions at full length, 24 path words per slot)
	bool poolsSized = false;
	uint32_t nExtendPairs = 0, nAnchorPairs = 0;   // event pairs recorded around the rounds' k_extend / k_build_anchors launches
	uint32_t poolReruns = 0;
	uint32_t* dReadTies = nullptr;     // per read: fragment extensions whose flattenLastSliceEnd minimum was tied between nodes (k_build_anchors adds them up; gc_result::flatten_ties)
	uint64_t pathCapacity = 0;
	uint32_t* dPathPool = nullptr;
	uint32_t* dChainOut = nullptr;
	uint32_t* dChainLen = nullptr;
	unsigned long long* dChainScore = nullptr;
	uint32_t* dChainStatus = nullptr;
	bool deviceStitch = false;
	StitchInfo* stitchInfo = nullptr;
	uint32_t* hStitchNodes = nullptr;
	uint32_t* dStitchNodes = nullptr;
	uint64_t stitchDenseCap = 0;
	unsigned long long* hStitchCursor = nullptr;
	// resultsBack()
	AnchorRec* anchors = nullptr;
	uint32_t* fragStatus = nullptr;
	uint32_t* fragExtended = nullptr;
	uint32_t* readTies = nullptr;
	uint32_t* chainOut = nullptr;
	uint32_t* chainLen = nullptr;
	unsigned long long* chainScore = nullptr;
	uint32_t* chainStatus = nullptr;
	uint32_t* pathPool = nullptr;
	// r5: the result's dense anchor arrays are made on the device (gc_results.hip) unless the anchors' traces are asked for (keep_traces == 1: the host walks the slots for them anyway)
	bool deviceAnchors = false;
	uint4* hAnchorPerRead = nullptr;                  // (anchors kept, path words, seeds extended, flags) per read
	unsigned long long* hAnchorOff = nullptr;         // [2 r] anchors, [2 r + 1] path words before read r; totals at [2 n], [2 n + 1]
	uint8_t* hAnchorDense = nullptr;                  // the dense arrays as they came down: nine 4-byte arrays, the 8-byte path offsets, the path words
	uint64_t denseAnchors = 0, densePathWords = 0;
	std::vector<ExtResult> extResults;
	std::vector<PoolCell> tracePool;
	bool anchorTraces = false;
	bool stitchNodesPending = false;
	// stitchAndChainDistances()
	const PathSeqJob* chainLetterJobs = nullptr;   // per read: where its stitched path's letters are in dChainLetters
	const char* dChainLetters = nullptr;
	// joinWholeReadPass()
	double tJoined = 0;
	const LongCell* longCells = nullptr;   // keep_traces: the merged traces in pinned staging (a pageable destination made this copy 2-3 s per 10 k reads)

	BatchRun(const gc_graph* G, const gc_seeder* S, gc_stream* st, const gc_reads* R, const gc_params* P, gc_result* res, double tCall, double cpuCall)
		: G(G), S(S), st(st), R(R), P(P), res(res), tCall(tCall), cpuCall(cpuCall), cpuJoined(cpuCall), n(R->offsets.size() - 1), hg(G->host), pool(WorkerPool::batch()), glue(st->glue),
		  stream(st->stream), longErrors(16) {}
	~BatchRun() { for (auto& t : longThreads) if (t.joinable()) t.join(); }   // (an exception on the main thread must not leave the pass thread behind with dangling state)
	BatchRun(const BatchRun&) = delete;
	BatchRun& operator=(const BatchRun&) = delete;

	void mark() { HIP_CHECK(hipEventRecord(st->ev[evIdx++], stream)); }
	double elapsedUs(int a, int b) { float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, st->ev[a], st->ev[b])); return (double)ms * 1000.0; }

	void run()
	{
		res->n_reads = n;
		tTotal = nowUs();
		// GC_DEBUG_TIMES: CPU time of THIS thread per stage (the waits poll: what a stage costs the host is not its wall time)
		const bool cpuStages = getenv("GC_DEBUG_TIMES") != nullptr;
		double cpuAt = cpuStages ? threadCpuMs() : 0, cpuStage[10] = {}, poolStage[10] = {};
		uint64_t poolAt = pool.cpuUs.load();
		const double processAt = processCpuMs();
		auto stageDone = [&](int k) {
			if (!cpuStages) return;
			const double now = threadCpuMs(); const uint64_t poolNow = pool.cpuUs.load();
			cpuStage[k] += now - cpuAt; poolStage[k] += (poolNow - poolAt) / 1e3;
			cpuAt = now; poolAt = poolNow;
		};
		seeds(); stageDone(0);
		prepareWholeReadPass(); stageDone(1);
		startWholeReadPass(); stageDone(2);
		fragmentPipeline(); stageDone(3);
		resultsBack(); stageDone(4);
		// r5: the trace pool and the anchor path pool are sized by what the stream's batches have used, not by every slot's worst case (a 2 000 x 50 kb batch on a 960 Mbp
		// graph has 21 M slots: 26 GB of trace pool by worst case); a batch that needs more than its stream has seen so far runs its fragment pipeline again with the room it asked for
		while (fragmentPoolsOverflowed()) { fragmentPipeline(); resultsBack(); stageDone(3); }
		res->counters[6] = poolReruns;
		compactAnchors(); stageDone(4);
		stitchAndChainDistances(); stageDone(5);
		joinWholeReadPass(); stageDone(6);
		chainedAlignments(); stageDone(7);
		encodeOutput(); stageDone(8);
		assemble(); stageDone(9);
		st->fragShare = (double)res->counters[4] / (double)std::max<uint64_t>(1, R->totalBases);
		st->batchesDone++;
		if (cpuStages)
			fprintf(stderr, "[gc cpu] main thread, ms of its own CPU: seeds %.1f, whole-read set-up %.1f + start %.1f, fragment pipeline %.1f, results back %.1f, stitching + chain distances %.1f, join %.1f, chained alignments %.1f, output %.1f, assembly %.1f; pass thread %.1f\n",
				cpuStage[0], cpuStage[1], cpuStage[2], cpuStage[3], cpuStage[4], cpuStage[5], cpuStage[6], cpuStage[7], cpuStage[8], cpuStage[9], passThreadCpuMs);
		if (cpuStages)   // (with one batch in flight: this call's own; the pool's figure includes this thread's share of the jobs, which the line above counts too)
			fprintf(stderr, "[gc cpu] worker pool jobs, ms of CPU over all threads, by stage: %.1f %.1f %.1f %.1f %.1f %.1f %.1f %.1f %.1f %.1f; the process in all %.1f\n",
				poolStage[0], poolStage[1], poolStage[2], poolStage[3], poolStage[4], poolStage[5], poolStage[6], poolStage[7], poolStage[8], poolStage[9], processCpuMs() - processAt);
	}

#include "batch/gc_batch_seeds.inc"   // K1 seed lookup and the glue between it and the extension kernels
#include "batch/gc_batch_long.inc"   // the whole-read pass: set-up, the round loop, fallback reruns, selection and NW distance, its own host thread
#include "batch/gc_batch_fragments.inc"   // the fragment pipeline: windows, work items, extension and anchors in lazy rounds, chaining, stitching
#include "batch/gc_batch_results.inc"   // what comes down: anchors, chains, stitched paths, the chains' NW distances; the pass thread's join
#include "batch/gc_batch_output.inc"   // the chained alignments' traces, the output encoders on the device, the flat result
};

int gc_align_batch(const gc_graph* G, const gc_seeder* S, gc_stream* st, const gc_reads* R, const gc_params* P, gc_result** out)
{
	if (!G || !S || !st || !R || !P || !out) return fail(GC_ERR_INVALID, "null argument");
	if (P->split_len < 16 || P->split_len > 64 || P->split_gap < 1) return fail(GC_ERR_INVALID, "split_len must be in [16,64] (one 64-row slice per fragment extension) and split_gap >= 1");
	{
		const gc_capacities& c = P->capacity;
		if (c.reserved[0] || c.reserved[1] || c.reserved[2]) return fail(GC_ERR_INVALID, "gc_params::capacity.reserved must be 0 (was the struct initialised with gc_params_default?)");
		const int64_t v[] = { c.ext_max_items, c.ext_max_pending, c.ext_max_trace, c.long_max_items, c.long_cells_per_base, c.long_scratch_bytes, c.stitch_set_max, c.stitch_bfs_cap };
		for (int64_t x : v) if (x < 0 || x > (1ll << 40)) return fail(GC_ERR_INVALID, "gc_params::capacity: a size is negative or absurd (0 = automatic)");
		// what the consumers can hold: the tables are indexed with 32 bits, and the retry launch takes 16x the fragment sizes
		if (c.ext_max_items > (1ll << 24) || c.ext_max_pending > (1ll << 24) || c.ext_max_trace > (1ll << 24)) return fail(GC_ERR_INVALID, "gc_params::capacity.ext_*: at most 2^24 (the retry launch reserves 16x)");
		if (c.long_max_items > (1ll << 28)) return fail(GC_ERR_INVALID, "gc_params::capacity.long_max_items: at most 2^28");
		if (c.long_cells_per_base > 4096) return fail(GC_ERR_INVALID, "gc_params::capacity.long_cells_per_base: at most 4096");
		if (c.stitch_set_max > (1ll << 31) - 1 || c.stitch_bfs_cap > (1ll << 31) - 1) return fail(GC_ERR_INVALID, "gc_params::capacity.stitch_*: at most 2^31 - 1");
		if (c.long_column_store < -1 || c.long_column_store > (1ll << 31)) return fail(GC_ERR_INVALID, "gc_params::capacity.long_column_store: -1 (none), 0 (automatic) or a column count");
	}
	if (P->device_output < 0 || P->device_output > 7 || (P->device_output & 3) == 3) return fail(GC_ERR_INVALID, "gc_params::device_output: 1 or 2 (GAF pieces with = / X or with M), optionally + 4 (vg::Path bytes)");
	if (P->device_output && !(P->long_pass && P->edit_distances)) return fail(GC_ERR_INVALID, "gc_params::device_output needs long_pass and edit_distances (the final alignments are what it encodes)");
	*out = nullptr;
	const double tCall = nowUs();
	const double cpuCall = processCpuMs();
	gc_result* res = (gc_result*)calloc(1, sizeof(gc_result));
	int rc = guarded([&]() {
		HIP_CHECK(hipSetDevice(st->device));   // the current device is per host thread
		// without the chained alignment's trace the decision comes from the two distances alone, which skips --E-cutoff's test of the chained alignment
		// (src/Aligner.cpp:904): with a cut-off set the traces are made whatever chain_traces says
		gc_params effective = *P;
		if (effective.chain_traces == 0 && effective.e_cutoff >= 0 && effective.stitch && effective.edit_distances) effective.chain_traces = 1;
		BatchRun batch(G, S, st, R, &effective, res, tCall, cpuCall);
		batch.run();
		return (int)GC_OK;
	});
	if (getenv("GC_DEBUG_TIMES")) fprintf(stderr, "[gc times] gc_align_batch returned after %.1f ms\n", (nowUs() - tCall) / 1e3);
	if (getenv("GC_DEBUG_TIMES")) fprintf(stderr, "[gc token] stream %p call returned %.1f\n", (void*)st, nowUs() / 1e3);
	if (rc != GC_OK) { gc_result_free(res); return rc; }
	*out = res;
	return GC_OK;
}

} // extern "C"

