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
		st->batchesDone++;
		if (cpuStages)
			fprintf(stderr, "[gc cpu] main thread, ms of its own CPU: seeds %.1f, whole-read set-up %.1f + start %.1f, fragment pipeline %.1f, results back %.1f, stitching + chain distances %.1f, join %.1f, chained alignments %.1f, output %.1f, assembly %.1f; pass thread %.1f\n",
				cpuStage[0], cpuStage[1], cpuStage[2], cpuStage[3], cpuStage[4], cpuStage[5], cpuStage[6], cpuStage[7], cpuStage[8], cpuStage[9], passThreadCpuMs);
		if (cpuStages)   // (with one batch in flight: this call's own; the pool's figure includes this thread's share of the jobs, which the line above counts too)
			fprintf(stderr, "[gc cpu] worker pool jobs, ms of CPU over all threads, by stage: %.1f %.1f %.1f %.1f %.1f %.1f %.1f %.1f %.1f %.1f; the process in all %.1f\n",
				poolStage[0], poolStage[1], poolStage[2], poolStage[3], poolStage[4], poolStage[5], poolStage[6], poolStage[7], poolStage[8], poolStage[9], processCpuMs() - processAt);
	}

	// ---------------- K1 seed lookup, then the glue between it and the extension kernels (on the device; GC_DEVICE_GLUE=0: on the host)
	void seeds()
	{
		// ---------------- K1: seed lookup
		uint32_t* dTmp = st->tmp.reserve<uint32_t>(R->totalBases);
		uint2* dMatches = st->matches.reserve<uint2>(R->totalBases);
		uint32_t* dReadMatchOff = st->readMatchOff.reserve<uint32_t>(n);
		uint32_t* dReadMatchCount = st->readMatchCount.reserve<uint32_t>(n);
		dCursors = st->cursors.reserve<unsigned long long>(8);
		dCounters = st->counters.reserve<unsigned long long>(8);
		hSmall = st->hSmall.reserve<unsigned long long>(16 + 2 * n);
		uint32_t* readMatchOff = (uint32_t*)(hSmall + 16);
		uint32_t* readMatchCount = readMatchOff + n;
		HIP_CHECK(hipMemsetAsync(dCursors, 0, 8 * sizeof(unsigned long long), stream));
		HIP_CHECK(hipMemsetAsync(dCounters, 0, 8 * sizeof(unsigned long long), stream));
		mark();   // 0
		launchSeedLookup(stream, S->dev, R->devBases, R->devOffsets, (uint32_t)n, (uint64_t*)dCursors, dReadMatchOff, dReadMatchCount, dMatches, R->totalBases, dTmp, R->totalBases, R->devChunkRead, R->devPacked, R->devInvalid);
		mark();   // 1
		// The glue between the seed lookup and the extension kernels (hit expansion, seed ordering, fragment windows) runs on the device
		// (gc_seedglue.hip: one wave per read, the reference's three unstable sorts replayed with libstdc++'s own algorithm); GC_DEVICE_GLUE=0
		// keeps the r2 host path (host/gc_glue.cpp: same results, 1 CPU-second and two bulk transfers per 10 k reads).
		deviceGlue = !(getenv("GC_DEVICE_GLUE") && atoi(getenv("GC_DEVICE_GLUE")) == 0);
		if (glue.size() < n) glue.resize(n);
		gc::KmerMatch* matches = nullptr;
		tGlue = 0;
		// what both paths leave behind for the rest of the batch
		if (deviceGlue) {
			unsigned long long* dGlueCursors = st->glueCursors.reserve<unsigned long long>(8);
			uint32_t* dSeedCap = st->glueSeedCap.reserve<uint32_t>(n);
			uint32_t* dSeedOff = st->glueSeedOff.reserve<uint32_t>(n + 1);
			unsigned long long* hGlueSmall = st->hGlueSmall.reserve<unsigned long long>(8);
			HIP_CHECK(hipMemsetAsync(dGlueCursors, 0, 8 * sizeof(unsigned long long), stream));
			launchSeedCaps(stream, S->dev, (uint32_t)n, R->devReadInvalid, dMatches, dReadMatchOff, dReadMatchCount, dSeedCap, dSeedOff, dGlueCursors + 5);
			HIP_CHECK(hipMemcpyAsync(hSmall, dCursors, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
			HIP_CHECK(hipMemcpyAsync(hGlueSmall, dGlueCursors, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
			// capacity of the per-read window staging: one window per fragment position (host: the lengths are known)
			uint32_t* hWinCapOff = st->hGlueWinCapOff.reserve<uint32_t>(n + 1);
			uint64_t winCap = 0;
			for (uint64_t r = 0; r < n; r++) {
				const uint64_t len = R->offsets[r + 1] - R->offsets[r];
				hWinCapOff[r] = (uint32_t)winCap;
				winCap += len >= (uint64_t)P->split_len ? (len - P->split_len) / P->split_gap + 1 : 1;
			}
			hWinCapOff[n] = (uint32_t)winCap;
			if (winCap >= 0xffffffffull) throw std::runtime_error("batch too large: more than 2^32 fragment positions; split the batch");
			uint32_t* dWinCapOff = st->glueWinCapOff.reserve<uint32_t>(n + 1);
			HIP_CHECK(hipMemcpyAsync(dWinCapOff, hWinCapOff, (n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
			syncStream(stream);
			res->kernel_us[0] = elapsedUs(0, 1);
			res->host_us[2] = nowUs() - tTotal;   // K1 + its transfers, wall
			tGlue = nowUs();
			const uint64_t nMatchesDev = hSmall[0], seedCap = hGlueSmall[5];
			if (nMatchesDev > R->totalBases) throw std::runtime_error("seed lookup overflowed its buffer");
			if (seedCap >= 0xfffffff0ull) throw std::runtime_error("batch too large: more than 2^32 seed occurrences; split the batch");
			nSeedsTotal = seedCap;
			GlueStaging stg;
			uint32_t** u32s[8] = { &stg.mPos, &stg.mStartLo, &stg.mStartHi, &stg.sSeqPos, &stg.sNode, &stg.sOffset, &stg.sGood, &stg.sCluster };
			for (int k = 0; k < 8; k++) *u32s[k] = st->glueU32[k].reserve<uint32_t>(seedCap);
			stg.sortBuf = (GlueElem*)st->glueSort.reserve<uint8_t>(seedCap * glueElemBytes());
			stg.sortScratch = st->gluePos.reserve<uint32_t>(3 * seedCap + 64 * n + 64);
			stg.winBuf = st->glueWin.reserve<uint32_t>(4 * winCap);
			dLongSeeds = st->longSeeds.reserve<LongSeed>(P->long_pass ? seedCap : 0);
			dReadSeeds = st->readSeeds.reserve<FragSeed>(seedCap);
			dFrags = st->frags.reserve<Fragment>(winCap);
			dFragFirstSeed = st->fragFirstSeed.reserve<uint32_t>(winCap);
			dJobs = st->jobs.reserve<ReadChainJob>(n);
			GlueRead* dGlueOut = st->glueOut.reserve<GlueRead>(n);
			GlueRead* hGlueOut = st->hGlueOut.reserve<GlueRead>(n);
			jobs = st->hJobs.reserve<ReadChainJob>(n);
			launchSeedGlue(stream, S->dev, G->dev, R->devOffsets, (uint32_t)n, R->devReadInvalid, dMatches, dReadMatchOff, dReadMatchCount, dSeedOff, dWinCapOff, P->seed_density,
				(uint32_t)P->split_len, (uint32_t)P->split_gap, P->long_pass != 0, stg, st->gluePerRead.reserve<uint32_t>(6 * (n + 1)), dLongSeeds, dReadSeeds, dFrags, dFragFirstSeed, dJobs, dGlueOut, dGlueCursors);
			if (n) HIP_CHECK(hipMemcpyAsync(hGlueOut, dGlueOut, n * sizeof(GlueRead), hipMemcpyDeviceToHost, stream));
			if (n) HIP_CHECK(hipMemcpyAsync(jobs, dJobs, n * sizeof(ReadChainJob), hipMemcpyDeviceToHost, stream));
			HIP_CHECK(hipMemcpyAsync(hGlueSmall, dGlueCursors, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
			syncStream(stream);
			nFrags = hGlueSmall[0]; nSlots = hGlueSmall[1]; traceBudget = hGlueSmall[2];
			maxSlotsPerRead = (uint32_t)std::max<uint64_t>(1, hGlueSmall[3]); maxWindowSeeds = (uint32_t)hGlueSmall[4];
			for (uint64_t r = 0; r < n; r++) {
				ReadGlue& gl = glue[r];
				gl.reset();
				const GlueRead& g = hGlueOut[r];
				gl.failed = g.failed != 0;
				gl.nSeedsR = g.nSeeds; gl.nWindows = g.nFrags;
				gl.seedBegin = g.seedOff; gl.longSeedBegin = g.seedOff;
				gl.fragBegin = g.fragBegin; gl.slotBegin = g.slotBegin;
			}
			// host copies for the result assembly: the fragments always; the seeds only for the seed_* arrays / the anchor traces
			frags = st->hFrags.reserve<Fragment>(nFrags);
			if (nFrags) HIP_CHECK(hipMemcpyAsync(frags, dFrags, nFrags * sizeof(Fragment), hipMemcpyDeviceToHost, stream));
			if (P->keep_seeds || P->keep_traces == 1) {
				readSeeds = st->hReadSeeds.reserve<FragSeed>(seedCap);
				fragFirstSeed = st->hFragFirstSeed.reserve<uint32_t>(nFrags);
				if (seedCap) HIP_CHECK(hipMemcpyAsync(readSeeds, dReadSeeds, seedCap * sizeof(FragSeed), hipMemcpyDeviceToHost, stream));
				if (nFrags) HIP_CHECK(hipMemcpyAsync(fragFirstSeed, dFragFirstSeed, nFrags * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
			}
			glueCopied = st->ev[11];
			HIP_CHECK(hipEventRecord(glueCopied, stream));
		} else {
		if (n) HIP_CHECK(hipMemcpyAsync(readMatchOff, dReadMatchOff, n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
		if (n) HIP_CHECK(hipMemcpyAsync(readMatchCount, dReadMatchCount, n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
		HIP_CHECK(hipMemcpyAsync(hSmall, dCursors, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
		syncStream(stream);
		uint64_t nMatches = hSmall[0];
		matches = st->hMatches.reserve<gc::KmerMatch>(nMatches);
		if (nMatches) HIP_CHECK(hipMemcpyAsync(matches, dMatches, nMatches * sizeof(uint2), hipMemcpyDeviceToHost, stream));
		syncStream(stream);
		res->kernel_us[0] = elapsedUs(0, 1);
		res->host_us[2] = nowUs() - tTotal;   // K1 + its transfers, wall

		// ---------------- host glue: order-critical sorts and fragment windows (see host/gc_glue.hpp)
		tGlue = nowUs();
		pool.run(n, [&](size_t r, size_t) { glue[r].reset(); });
		std::vector<gc::GlueScratch> scratch(pool.size());
		pool.run(n, [&](size_t r, size_t worker) {
			ReadGlue& gl = glue[r];
			size_t len = R->offsets[r + 1] - R->offsets[r];
			if (R->invalid[r]) { gl.failed = true; return; }
			gc::expandSeeds(S->host, matches + readMatchOff[r], readMatchCount[r], len, P->seed_density, gl.seeds, scratch[worker]);
			if (gl.seeds.empty()) return;
			if (!gc::orderSeedsByChaining(hg, gl.seeds, scratch[worker])) {
				gl.failed = true;
				gl.seeds.clear();
				return;
			}
			if (P->long_pass) gl.longSeeds = gl.seeds;
		});
		}
		tOrdered = nowUs();
	}

	// ---------------- K3-long set-up: buffers, the round loop (runLongGroup), the rerun rules; nothing runs yet
	void prepareWholeReadPass()
	{
		// ---------------- K3-long: whole-read pass on its own stream (src/Aligner.cpp:630-654)
		uint64_t nLongSeeds = 0;
		maxReadLen = 1;
		for (uint64_t r = 0; r < n; r++) maxReadLen = std::max<uint64_t>(maxReadLen, R->offsets[r + 1] - R->offsets[r]);
		// alignments kept per read: the reference has no limit; 32 is far above what 10 kb reads produce (3.6 seeds extended on average), longer
		// and noisier reads get room in proportion. A read that still exceeds it is flagged (capacity_exceeded), the batch goes on.
		maxAlignments = (uint32_t)std::max<uint64_t>(32, maxReadLen / 512);
		// Decision for a set of reads whose whole-read alignments are final: the reference's alignment order, the GreedyLength
		// selection, and (queued, not awaited) the path letters + NW edit distance of the best alignment. (Tried: deciding the
		// reads that are already finished when the rounds turn latency-bound, so these kernels run beside the last rounds - the
		// rounds slow down by more than the 11 ms the tail saves: 304-319 -> 318-337 ms per batch. So: all reads, after the rounds.)
		if (P->long_pass) {
			if (deviceGlue) nLongSeeds = nSeedsTotal;   // (the device's seed lists sit at the reads' capacity offsets)
			else for (uint64_t r = 0; r < n; r++) { glue[r].longSeedBegin = nLongSeeds; nLongSeeds += glue[r].longSeeds.size(); }
			if (nLongSeeds >= 0xffffffffull) throw std::runtime_error("batch too large for the whole-read pass");
			LongSeed* hSeeds = st->hLongSeeds.reserve<LongSeed>(deviceGlue ? 0 : nLongSeeds);
			hJobs = st->hLongJobs.reserve<LongJob>(n);
			// merged-trace cells per read base: 8 hold the few partial alignments a 10 kb ONT read collects before its end-to-end one (cfg2 uses ~1.1);
			// noisy 50 kb CLR reads on a genome with repeats collect 8-9 alignments each and overflowed it (a quarter of the reads flagged, which reads
			// depending on timing). The stream remembers what its batches needed, and a batch that overflows reruns its pass with three times the room.
			cellPoolPinned = getenv("GC_LONG_CELLS_PER_BASE") || P->capacity.long_cells_per_base > 0;
			uint64_t cellsPerBase = (uint64_t)std::max<int64_t>(2, capacityOr("GC_LONG_CELLS_PER_BASE", P->capacity.long_cells_per_base, (int64_t)st->longCellsPerBase));
			if (getenv("GC_LONG_FORCE_FALLBACK") && !cellPoolPinned) cellsPerBase *= 3;   // (test hook: every read's alignments are made twice, by the rounds and by the fallback kernel, into the same pool)
			cellBudget = cellBudgetFor(cellsPerBase);
			pool.run(n, [&](size_t r, size_t) {
				const ReadGlue& gl = glue[r];
				uint64_t at = gl.longSeedBegin;
				if (deviceGlue) at += gl.nSeedsR;
				else for (const gc::SeedRec& s : gl.longSeeds) {
					hSeeds[at++] = LongSeed { s.node, s.seqPos, s.goodness, s.clusterSize, s.offset, 0 };
				}
				LongJob& j = hJobs[r];
				j.maskOff = R->maskOff[r];
				j.maskWords = R->maskWords[r];
				j.pad = 0;
				j.readOff = R->offsets[r];
				j.readLen = (uint32_t)(R->offsets[r + 1] - R->offsets[r]);
				j.seedBegin = (uint32_t)gl.longSeedBegin;
				j.seedEnd = (uint32_t)at;
				j.alnBegin = (uint32_t)(r * maxAlignments);
			});
			lcfg.bandwidth = P->bandwidth;
			lcfg.maxSlices = (uint32_t)(maxReadLen / 64 + 3);
			lcfg.maxItems = (uint32_t)std::max<uint64_t>(8192, (maxReadLen / 64 + 3) * 24);   // (slice, node) tiles of one extension: ~8 per slice on cfg2, room for 24
			lcfg.maxPending = 96;
			lcfg.maxTrace = (uint32_t)(maxReadLen + maxReadLen / 2 + 512);
			// column store of the one-extension-per-wave kernel: the DP keeps every column (16 B) so that the backtrace loads its tiles' columns back instead of
			// recomputing them (45 % of the kernel's column steps). ~2.1 columns per read row on cfg2; an extension that needs more than this room ends
			// with EXT_OVERFLOW and its read goes to the plain-layout kernel, which recomputes. GC_LONG_MAX_COLS=0: no store (the r2 behaviour).
			lcfg.maxCols = (uint32_t)std::max<int64_t>(0, capacityOr("GC_LONG_MAX_COLS", P->capacity.long_column_store, (int64_t)(3 * maxReadLen + 4096)));   // (-1 in the parameters, 0 in the environment: no store)
			lcfg.maxItems = (uint32_t)std::max<int64_t>(64, capacityOr("GC_LONG_MAX_ITEMS", P->capacity.long_max_items, lcfg.maxItems));
			if (const char* env = getenv("GC_LONG_REG_CAP")) lcfg.regCap = (uint32_t)std::max(1, std::min(64, atoi(env)));   // test hook: force the LDS-table retry
			waveWords = longWaveWordsPerLane(lcfg);
			if (!deviceGlue) dLongSeeds = st->longSeeds.reserve<LongSeed>(nLongSeeds);
			dLongJobs = st->longJobs.reserve<LongJob>(n);
			dLongAlns = st->longAlns.reserve<LongAln>(n * maxAlignments);
			LongReadResult* dLongResults = st->longResults.reserve<LongReadResult>(n);
			dLongCells = st->longCells.reserve<LongCell>(cellBudget, true);
			// Read groups: the rounds of one group are serial (select -> extend -> merge, host decides when to stop). Groups can
			// run their round loops concurrently, each on its own stream and host thread (GC_LONG_GROUPS). Measured on cfg2
			// (10k reads): 1 group 367 ms/step, 2 groups 517, 4 groups 477, 8 groups 607 - the groups' big rounds coincide and
			// their tails too, so nothing overlaps usefully and the kernels slow each other down. Default: one group.
			nGroups = 1;
			if (const char* env = expEnv("GC_LONG_GROUPS")) nGroups = (uint32_t)std::max(1, std::min(16, atoi(env)));   // (experiments build only)
			if (n < 64ull * nGroups) nGroups = 1;
			while (st->groupStreams.size() < nGroups) {
				hipStream_t q = nullptr;
				createStream(&q, 1);
				st->groupStreams.push_back(q);
				for (int k = 0; k < 2 * LONG_EVENT_RING; k++) { hipEvent_t e = nullptr; HIP_CHECK(hipEventCreate(&e)); st->groupEvents.push_back(e); }   // a ring of (begin, end) pairs around the rounds' extension launches
			}
			// cursors: [0] cell pool, [8..15] counters (+ [16..31] profiling stamps), per group g at 32+8g: [+0] work count, [+1] round trace cursor
			cursorWords = 32 + 8 * 16;
			dLongCursor = st->longCursor.reserve<unsigned long long>(cursorWords);
			hLongAlns = st->hLongAlns.reserve<LongAln>(n * maxAlignments);
			hLongResults = st->hLongResults.reserve<LongReadResult>(n);
			hLongSmall = st->hLongSmall.reserve<unsigned long long>(cursorWords);
			ls = st->longStream;
			HIP_CHECK(hipMemsetAsync(dLongCursor, 0, cursorWords * sizeof(unsigned long long), ls));
			if (nLongSeeds && !deviceGlue) HIP_CHECK(hipMemcpyAsync(dLongSeeds, hSeeds, nLongSeeds * sizeof(LongSeed), hipMemcpyHostToDevice, ls));
			if (n) HIP_CHECK(hipMemcpyAsync(dLongJobs, hJobs, n * sizeof(LongJob), hipMemcpyHostToDevice, ls));
			syncStream(ls);   // the group streams start from uploaded inputs
			// rounds: select -> extend -> merge until no read has a seed left to extend (see gc_kernels.hip, "K3-long in rounds")
			dLongState = st->longState.reserve<LongState>(n);
			const uint64_t workCapacity = 8 * n + 64ull * nGroups;   // all groups together; group g owns the slice for its reads
			dLongWork = st->longWork.reserve<LongWork>(workCapacity);
			dLongWorkResults = st->longWorkResults.reserve<LongWorkResult>(workCapacity);
			dCandSeed = st->longCandSeed.reserve<uint32_t>(2 * workCapacity);   // (two halves: k_long_round alternates them by the round's parity)
			dWorkLen = st->longWorkLen.reserve<uint32_t>(workCapacity);   // written by k_long_select, sorted into dOrder by k_long_order: the host only
			dRetryList = st->longRetryList.reserve<uint32_t>(workCapacity);   // work items whose band outgrew the register tables (per round)
			dOrder = st->longOrder.reserve<uint32_t>(workCapacity);       // learns the round's work count (k_publish: no copy-engine transfer in the round loop)
			groupBegin.assign(nGroups + 1, 0); groupTraceBegin.assign(nGroups + 1, 0);
			for (uint32_t g = 0; g <= nGroups; g++) groupBegin[g] = n * g / nGroups;
			for (uint32_t g = 0; g < nGroups; g++) {
				uint64_t budget = 0;
				for (uint64_t r = groupBegin[g]; r < groupBegin[g + 1]; r++) { uint64_t len = R->offsets[r + 1] - R->offsets[r]; budget += 4 * (len + len / 2 + 1024); }   // up to four candidate seeds' worth per read (the speculation rule below keeps rounds within it)
				groupTraceBegin[g + 1] = groupTraceBegin[g] + budget;
			}
			dRoundTrace = st->longRoundTrace.reserve<unsigned long long>(groupTraceBegin[nGroups]);
			// extension scratch: one region per lane of a resident wave (persistent waves fetch work items), per read group
			// (bounded by a memory budget: 0.8 MB per lane for 10 kb reads, 2.4 MB for 50 kb reads; GC_LONG_SCRATCH_GB overrides the 48 GB)
			uint64_t scratchBudget = P->capacity.long_scratch_bytes > 0 ? (uint64_t)P->capacity.long_scratch_bytes : (48ull << 30) / (uint64_t)longTokenCount(n, st->batchesDone);
			if (const char* env = getenv("GC_LONG_SCRATCH_GB")) scratchBudget = (uint64_t)std::max(1, atoi(env)) << 30;
			// (r5: no more lanes than a round can hold without speculation - two work items per read; the late rounds' speculation stays below that, and a round that does exceed
			// it runs persistent waves. A 2 000 x 50 kb batch reserved 48 GB for rounds of 4 000 extensions, a 10 k x 10 kb batch 48 GB for 20 000: now 20 and 27 GB)
			scratchLanes = std::min<uint64_t>(std::min<uint64_t>(workCapacity + 64, 2 * n + 128), std::max<uint64_t>(2048, std::min<uint64_t>(65536 + 64, scratchBudget / (waveWords * 8))));
			// one pass at a time (the default) works in the device's shared scratch; the experiments that let passes overlap keep a scratch per stream
			longScratchWords = (uint64_t)nGroups * scratchLanes * waveWords;
			shareLongScratch = nGroups == 1 && (getenv("GC_LONG_TOKEN") ? atoi(getenv("GC_LONG_TOKEN")) : 1) >= 1;   // (token per pass or per round: whoever holds it owns the scratch)
			if (!shareLongScratch) dLongScratchOwn = st->longScratch.reserve<unsigned long long>(longScratchWords);
			groupExtendUs.assign(nGroups, 0.0);
			groupRounds.assign(nGroups, 0);
			longGroups = nGroups;
			// reads whose band did not fit the LDS tables (status 5) are rerun with the plain-layout kernel
		}
		// What follows the rounds: fallback reruns, the reference's `cont` rule, selection and the NW distance of the best whole-read alignment.
		// With one read group it runs on the pass's own thread right after the rounds, beside the tail of the fragment pipeline (which ends
		// 20-30 ms after the pass on cfg2, starved by it), instead of after the join: 16 ms off the batch's critical path.
		// It writes the reads' long* fields and capacityExceededLong only; the fragment pipeline does not touch those.
		longPostInThread = P->long_pass && longGroups == 1;
		// The whole-read pass is the longest leg of the batch: its round loop runs on its own host thread and stream from
		// here on, while this thread prepares and runs the fragment pipeline.
	}

	uint64_t cellBudgetFor(uint64_t perBase) const { uint64_t b = 0; for (uint64_t r = 0; r < n; r++) b += perBase * (R->offsets[r + 1] - R->offsets[r]) + 1024; return b; }   // the merged-trace cell pool: cells per read base + slack per read

	bool growLongCells()   // whole-read pass thread: the cell pool was too small -> enlarge it, reset the pass's cursors; false when it cannot grow
	{
		bool overflowed = false;
		for (uint64_t r = 0; r < n && !overflowed; r++) overflowed = hLongResults[r].status == 4;
		if (!overflowed || cellPoolPinned) return false;   // (a pinned pool flags the reads instead: the caller asked for that much and no more)
		// r5: by what the pass asked for, not three times the last size (8 -> 24 cells per read base put 38 GB into a 2 000 x 50 kb batch in flight): the pool's cursor counts every
		// request, refused ones included; a read that was refused stops asking, so the count is a lower bound - half as much again, and the loop comes back if that is still short
		HIP_CHECK(hipMemcpyAsync(hLongSmall, dLongCursor, sizeof(unsigned long long), hipMemcpyDeviceToHost, ls));
		syncStream(ls);
		const uint64_t asked = hLongSmall[0], bases = std::max<uint64_t>(1, R->totalBases);
		const uint64_t next = std::max<uint64_t>(st->longCellsPerBase + 2, (asked + asked / 2 + bases - 1) / bases);
		if (getenv("GC_DEBUG_TIMES")) fprintf(stderr, "[gc mem] the whole-read pass runs again: %.2f G merged-trace cells asked for, %.2f G reserved (%llu per read base), now %llu per base\n", asked / 1e9, cellBudget / 1e9, (unsigned long long)st->longCellsPerBase, (unsigned long long)next);
		if (next > 256 || cellBudgetFor(next) * sizeof(LongCell) > (64ull << 30)) return false;
		st->longCellsPerBase = next;
		cellBudget = cellBudgetFor(next);
		dLongCells = st->longCells.reserve<LongCell>(cellBudget, true);
		HIP_CHECK(hipMemsetAsync(dLongCursor, 0, cursorWords * sizeof(unsigned long long), ls));
		syncStream(ls);
		return true;
	}

#ifdef GC_EXPERIMENTS
	// The round loop without a host round trip per round (r4; an experiment, GC_LONG_ROUNDS=1 - see roundsOnDevice): per round ONE kernel between two extension launches - k_long_round: the previous round's merge,
	// this round's select, the execution order, the work count to the device and to pinned host memory - and the extension kernel takes its item count from the device
	// (its grid is sized by a bound: 2 items per read, which the device-side speculation rule respects). Rounds are queued several at a time; the host looks at the published
	// counts only at the end of a chunk (a round after the last one finds nothing to do and costs a few empty launches). r3's loop queued zero / select / order / publish,
	// waited for the count, then extend / retry / merge: with five batches in flight each of those launches waited for a wave slot among the other batches' kernels and the
	// pass took 149 ms for 121 ms of extension kernels.
	bool roundsOnDevice(uint32_t g) const
	{
		if (nGroups != 1 || longExtendTeamSize(1) != 1) return false;
		for (const char* name : { "GC_LONG_SM", "GC_LONG_LANE", "GC_LONG_MAX_BLOCKS", "GC_LONG_PLAN" }) if (getenv(name)) return false;   // experiments and test hooks of the host-driven loop
		if (getenv("GC_LONG_TOKEN") && atoi(getenv("GC_LONG_TOKEN")) == 2) return false;
		if (!(getenv("GC_LONG_ROUNDS") && atoi(getenv("GC_LONG_ROUNDS")) == 1)) return false;   // GC_LONG_ROUNDS=1 selects it: measured 4-6 % SLOWER than the host-driven loop, which stays the default
		(void)g;
		return true;
	}
	void runLongGroupOnDevice(uint32_t g)
	{
		const uint64_t r0 = groupBegin[g], nG = groupBegin[g + 1] - r0;
		if (longTokenTake) longTokenTake();   // (rounds are queued ahead here: the token covers the whole loop)
		unsigned long long* dLongScratch = shareLongScratch ? longScratchOfToken : dLongScratchOwn;
		hipStream_t q = st->groupStreams[g];
		hipEvent_t* ring = st->groupEvents.data() + (size_t)2 * LONG_EVENT_RING * g;
		auto collect = [&](int slot) { float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, ring[2 * slot], ring[2 * slot + 1])); groupExtendUs[g] += (double)ms * 1000.0; };
		const uint64_t w0 = 8 * r0 + 64ull * g, capacity = 8 * nG + 64;
		unsigned long long* cursorSets = dLongCursor + 32 + 8 * g;   // two sets of four words: [0] work count, [1] round trace cursor, [2] next work slot, [3] length of the retry list
		const uint64_t traceBudget = groupTraceBegin[g + 1] - groupTraceBegin[g];
		const int MAX_ROUNDS = 250;
		unsigned long long* dRoundInfo = st->longRoundInfo.reserve<unsigned long long>(MAX_ROUNDS + 8);   // [0] ticket, [1 + round] work items of the round
		volatile unsigned long long* hInfo = st->hLongRoundInfo.reserve<unsigned long long>(MAX_ROUNDS + 8);   // [0] rounds published, [2 + round] work items of the round
		hInfo[0] = 0;
		const double dbgT0 = nowUs();
		double dbgWaitUs = 0;
		launchLongInit(q, dLongJobs + r0, (uint32_t)nG, dLongState + r0);
		launchZeroWords(q, dRoundInfo, 1);
		launchZeroWords(q, cursorSets, 8);
		// the extension launch's grid: every lane of the scratch. Two items per read cover a round without speculation (a round never holds more items than the one before it),
		// and the device-side rule keeps speculation within gridLimit; a batch whose 2 nG exceed the scratch's lanes runs persistent waves instead
		const uint64_t laneLimit = std::max<uint64_t>(1, scratchLanes - 64);
		const bool gridCovers = 2 * nG <= laneLimit;
		const uint32_t gridLimit = (uint32_t)std::min<uint64_t>(capacity, laneLimit);   // (as many lanes as the scratch has: the late rounds' speculation rule may use them)
		uint32_t forceCand = 0;
		if (const char* env = getenv("GC_LONG_SPECULATE")) forceCand = (uint32_t)std::min(2, std::max(1, atoi(env)));   // test hook: speculate from round 0
		const char* orderEnv = getenv("GC_LONG_ORDER");
		const uint32_t orderMode = orderEnv ? (uint32_t)atoi(orderEnv) : 1u;
		int queued = 0, timed = 0, done = -1;
		while (done < 0 && queued < MAX_ROUNDS) {
			const int chunk = queued == 0 ? 6 : 2;   // cfg2 needs six rounds; beyond that two at a time
			for (int k = 0; k < chunk && queued < MAX_ROUNDS; k++, queued++) {
				const uint32_t round = (uint32_t)queued;
				unsigned long long* cur = cursorSets + 4 * (round & 1u);
				launchLongRound(q, G->dev, dLongJobs + r0, (uint32_t)nG, dLongSeeds, (uint32_t)P->min_cluster_size, round, forceCand, gridLimit, dLongState + r0, dLongAlns, dLongCells, dLongCursor, cellBudget, maxAlignments,
					dLongWork + w0, dWorkLen + w0, dCandSeed + w0, dLongWorkResults + w0, dRoundTrace + groupTraceBegin[g], cursorSets, dRoundInfo + 1, dRoundInfo, dOrder + w0, (uint32_t)maxReadLen, orderMode,
					(unsigned long long*)hInfo, capacity);
				if (timed >= LONG_EVENT_RING) collect(timed % LONG_EVENT_RING);
				hipEvent_t ev0 = ring[2 * (timed % LONG_EVENT_RING)], ev1 = ring[2 * (timed % LONG_EVENT_RING) + 1];
				HIP_CHECK(hipEventRecord(ev0, q));
				launchLongExtend(q, G->dev, G->devTables, R->devMasks, lcfg, dLongWork + w0, dOrder + w0, gridCovers ? gridLimit : (uint32_t)std::min<uint64_t>(capacity, 0xffffffffull), dLongScratch + (uint64_t)g * scratchLanes * waveWords, 1, gridLimit,
					dRoundTrace + groupTraceBegin[g], cur + 1, traceBudget, dLongWorkResults + w0, dLongCursor + 8, cur + 2, 0, cur, dRetryList + w0, cur + 3, gridCovers);
				// extensions whose band outgrew the 64-entry register tables: second try with the LDS/HBM tables (two lanes per wave); the list is almost always empty
				if (!gridCovers) launchZeroWords(q, cur + 2, 1);   // (persistent waves used the slot counter)
				const uint32_t retryBlocks = std::min<uint32_t>(16, (uint32_t)std::max<uint64_t>(1, laneLimit / 2));
				launchLongExtend(q, G->dev, G->devTables, R->devMasks, lcfg, dLongWork + w0, dRetryList + w0, (uint32_t)std::min<uint64_t>(capacity, 0xffffffffull), dLongScratch + (uint64_t)g * scratchLanes * waveWords, 2, retryBlocks,
					dRoundTrace + groupTraceBegin[g], cur + 1, traceBudget, dLongWorkResults + w0, dLongCursor + 8, cur + 2, EXT_LDS_CAP, cur + 3);
				HIP_CHECK(hipEventRecord(ev1, q));
				timed++;
			}
			const double tWait0 = nowUs();
			syncStream(q);
			dbgWaitUs += nowUs() - tWait0;
			if ((int)hInfo[0] != queued) throw std::runtime_error("internal: the whole-read rounds did not publish their counts");
			for (int r = 0; r < queued && done < 0; r++) if (hInfo[2 + r] == 0) done = r;   // round `done` found no seed left to extend (its merge of the round before ran)
		}
		if (done < 0) throw std::runtime_error("whole-read pass: more rounds than the round loop queues");
		groupRounds[g] += (uint32_t)done;
		launchLongFinish(q, (uint32_t)nG, dLongState + r0, hLongResults + r0);
		syncStream(q);
		for (int k = std::max(0, timed - LONG_EVENT_RING); k < timed; k++) collect(k % LONG_EVENT_RING);
		if (getenv("GC_DEBUG_TIMES")) fprintf(stderr, "[gc times] whole-read rounds (queued ahead): %.1f ms in all, %.1f ms waiting at the chunks' ends, %d rounds with work, %d queued\n", (nowUs() - dbgT0) / 1e3, dbgWaitUs / 1e3, done, queued);
	}

#endif

	void runLongGroup(uint32_t g)   // the round loop of one read group: select -> extend -> merge until no read has a seed left to extend (see gc_kernels.hip, "K3-long in rounds")
	{
		const uint64_t r0 = groupBegin[g], nG = groupBegin[g + 1] - r0;
		if (nG == 0) return;
#ifdef GC_EXPERIMENTS
		if (roundsOnDevice(g)) { runLongGroupOnDevice(g); return; }
#endif
		unsigned long long* dLongScratch = nullptr;   // (set when the token is taken - before the first extension launch; round token: under the lock, every round)
		hipStream_t q = st->groupStreams[g];
		hipEvent_t* ring = st->groupEvents.data() + (size_t)2 * LONG_EVENT_RING * g;
		// the rounds' extension time: read a ring slot's pair before the slot is reused (its round is complete by then: every round's
		// work count has been awaited since) and what is left after the last round
		auto collect = [&](int slot) { float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, ring[2 * slot], ring[2 * slot + 1])); groupExtendUs[g] += (double)ms * 1000.0; };
		int timedRounds = 0;
		const uint64_t w0 = 8 * r0 + 64ull * g, capacity = 8 * nG + 64;   // this group's slice of the work arrays
		unsigned long long* cursor = dLongCursor + 32 + 8 * g;
		volatile unsigned long long* hCursor = hLongSmall + 32 + 8 * g;
		const uint64_t traceBudget = groupTraceBegin[g + 1] - groupTraceBegin[g];
		double dbgWaitUs = 0;
		const double dbgT0 = nowUs();
		launchLongInit(q, dLongJobs + r0, (uint32_t)nG, dLongState + r0);
		uint32_t lastWork = 0xffffffffu;
		// GC_LONG_TOKEN=2: the token (and with it the device's extension scratch) is held per round - from the moment a round's extension kernel is queued until
		// that kernel has finished - so that the small kernels and the host round trip between two rounds of one batch run beside another batch's extension kernel
		int deviceNow = 0;
		HIP_CHECK(hipGetDevice(&deviceNow));
		const bool roundToken = expEnv("GC_LONG_TOKEN") && atoi(expEnv("GC_LONG_TOKEN")) == 2;   // (experiments build only; read per batch: the tests switch modes inside one process)
		std::unique_lock<std::mutex> roundLock(g_longRoundToken[deviceNow & 15], std::defer_lock);
		hipEvent_t roundExtendDone = nullptr;

		for (int round = 0; round < 4096; round++) {
			launchZeroWords(q, cursor, 4);   // [0] work count, [1] round trace cursor, [2] next work slot, [3] length of the retry list
			// tail rounds: once fewer than a quarter of the reads are still active the chip is mostly idle, so the
			// remaining reads try several seeds per round (exact: k_long_merge re-checks them in order)
			// (the number of work items stays below what round 0 had: active reads x candidates <= n)
			uint32_t maxCand = 1;
			if (round > 0 && lastWork > 0) maxCand = (uint32_t)std::min<uint64_t>(8, std::max<uint64_t>(1, (2 * nG) / lastWork));
			if (round > 0 && lastWork < 8192) maxCand = 8;   // fewer work items than wave slots: the round costs one extension's latency whatever it holds
			// at most lastWork/2 reads are still active, so this keeps the round within the work arrays (8 per read) and the trace budget (4 seeds' worth per read)
			if (round > 0 && lastWork > 0) maxCand = (uint32_t)std::min<uint64_t>(maxCand, std::max<uint64_t>(1, (8 * nG) / lastWork));
			if (const char* env = getenv("GC_LONG_SPECULATE")) maxCand = (uint32_t)std::min(2, std::max(1, atoi(env)));   // test hook: speculate from round 0
			static const uint32_t candCap = getenv("GC_LONG_CAND_MAX") ? (uint32_t)std::max(1, std::min(8, atoi(getenv("GC_LONG_CAND_MAX")))) : 8u;   // measurement hook: fewer speculated seeds per read and round
			maxCand = std::min(maxCand, candCap);
			// speculation plan (r4): candidates per read in rounds 0, 1, 2, ... (the last entry repeats), a floor under the rule above; still bounded by the work arrays
			// and the trace budget (at most nG / 2... reads x candidates <= 4 nG). Why: rounds 3-5 of cfg2 hold fewer work items than the chip has wave slots and cost one
			// extension's latency (~17 ms) each - 98 % of the reads extend a second seed and 81 % a third, so asking for two seeds per read from round 0 on
			// merges rounds at a few per cent of wasted extensions (k_long_merge drops a candidate that an alignment accepted before it explains).
#ifdef GC_EXPERIMENTS
			{
				static const std::vector<int> plan = []() { std::vector<int> v; const char* e = getenv("GC_LONG_PLAN"); std::string t = e ? e : GC_LONG_PLAN_DEFAULT; size_t at = 0; while (at < t.size()) { v.push_back(std::max(1, std::min(8, atoi(t.c_str() + at)))); size_t c = t.find(',', at); if (c == std::string::npos) break; at = c + 1; } if (v.empty()) v.push_back(1); return v; }();
				const uint32_t floorCand = (uint32_t)plan[std::min<size_t>((size_t)round, plan.size() - 1)];
				const uint64_t active = round == 0 ? nG : std::max<uint64_t>(1, std::min<uint64_t>(nG, lastWork / 2));
				if (!getenv("GC_LONG_SPECULATE")) maxCand = (uint32_t)std::min<uint64_t>(std::max(maxCand, floorCand), std::max<uint64_t>(1, (4 * nG) / active));
			}
#endif
			launchLongSelect(q, G->dev, dLongJobs + r0, (uint32_t)nG, dLongSeeds, R->totalBases, (uint32_t)P->min_cluster_size, maxCand, dLongState + r0, dLongAlns, dLongCells, dLongWork + w0, dWorkLen + w0, dCandSeed + w0, cursor, capacity);
			{
				// execution order: longest extensions first, so the round's tail is made of short ones (GC_LONG_ORDER=0: as emitted)
				const char* mode = getenv("GC_LONG_ORDER");
				launchLongOrder(q, dWorkLen + w0, cursor, dOrder + w0, (uint32_t)maxReadLen, mode ? (uint32_t)atoi(mode) : 1u);
			}
			launchPublish(q, cursor, (unsigned long long*)hCursor, 2);
			const double tWait0 = nowUs();
			if (roundLock.owns_lock()) { syncEvent(roundExtendDone); roundLock.unlock(); }   // the previous round's extension kernel has finished: the merge and this round's set-up need no token
			syncStream(q);
			dbgWaitUs += nowUs() - tWait0;
			uint32_t nWorkItems = (uint32_t)hCursor[0];
			if (nWorkItems == 0) break;
			if (!dLongScratch) {
				if (longTokenTake) longTokenTake();
				dLongScratch = shareLongScratch ? longScratchOfToken : dLongScratchOwn;
			}
			if (roundToken && nGroups == 1) {
				roundLock.lock();
				if (shareLongScratch) dLongScratch = g_longScratch[deviceNow & 15].buffer[0].reserve<unsigned long long>(longScratchWords);
			}
			uint32_t team = longExtendTeamSize(nWorkItems);
			uint32_t blocks = std::min<uint32_t>((nWorkItems + team - 1) / team, (uint32_t)std::max<uint64_t>(1, (scratchLanes - 64) / team));
			if (const char* env = getenv("GC_LONG_MAX_BLOCKS")) blocks = std::min<uint32_t>(blocks, (uint32_t)std::max(1, atoi(env)));   // test hook: force persistent waves
			if (timedRounds >= LONG_EVENT_RING) collect(timedRounds % LONG_EVENT_RING);
			hipEvent_t ev0 = ring[2 * (timedRounds % LONG_EVENT_RING)], ev1 = ring[2 * (timedRounds % LONG_EVENT_RING) + 1];
			HIP_CHECK(hipEventRecord(ev0, q));
			// The experiments build (`make -C graphchainer_amd/csrc experiments`) can replace the extension step by one of the two measured-and-rejected layouts:
			// GC_LONG_SM=1 (one extension per LANE as per-lane state machines, k_long_extend_sm in gc_sm.hip, 6x slower; what outgrows its tables -
			// EXT_SM_DECLINED - is listed and rerun one extension per wave) or GC_LONG_LANE=1 (one extension per LANE with the plain-layout core and a per-lane HBM slab, 6.9x slower)
#ifdef GC_EXPERIMENTS
			const bool useSm = team == 1 && getenv("GC_LONG_SM") && atoi(getenv("GC_LONG_SM")) == 1;
			const bool useLane = !useSm && team == 1 && getenv("GC_LONG_LANE") && atoi(getenv("GC_LONG_LANE")) == 1;
			// GC_LONG_SPLIT=p (r5 experiment, VERDICT r4 item 3: "use the idle vector issue port"): the round's last p % of the work items (the shortest - the list is sorted
			// longest first) go to the multi-lane instantiation (GC_LONG_SPLIT_TEAM lanes per wave, 16: divergent lanes, i.e. vector instructions, LDS tables) on a second
			// stream, beside the one-extension-per-wave kernel that saturates the CUs' scalar units
			uint32_t nVector = 0, vectorTeam = 16;
			if (const char* env = getenv("GC_LONG_SPLIT_TEAM")) { const int v = atoi(env); if (v == 2 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64) vectorTeam = (uint32_t)v; }
			if (const char* env = getenv("GC_LONG_SPLIT")) {
				const uint64_t want = (uint64_t)nWorkItems * (uint64_t)std::max(0, std::min(90, atoi(env))) / 100 / vectorTeam * vectorTeam;
				if (team == 1 && nGroups == 1 && nWorkItems >= 4096 && want >= vectorTeam && (uint64_t)nWorkItems + vectorTeam + 64 <= scratchLanes && (uint64_t)blocks * team >= nWorkItems) nVector = (uint32_t)want;
			}
			if (useSm) {
				launchLongExtendSm(q, G->dev, G->devTables, R->devMasks, lcfg, dLongWork + w0, dOrder + w0, nWorkItems, (uint8_t*)(dLongScratch + (uint64_t)g * scratchLanes * waveWords), scratchLanes * waveWords * 8,
					dRoundTrace + groupTraceBegin[g], cursor + 1, traceBudget, dLongWorkResults + w0, dLongCursor + 8, cursor + 2);
				launchZeroWords(q, cursor + 2, 2);   // [2] next slot, [3] length of the list
				launchLongRetryList(q, dLongWorkResults + w0, nWorkItems, 6u /* EXT_SM_DECLINED */, dRetryList + w0, cursor + 3);
				const uint32_t declinedBlocks = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(nWorkItems, 8192), std::max<uint64_t>(1, scratchLanes - 64));
				launchLongExtend(q, G->dev, G->devTables, R->devMasks, lcfg, dLongWork + w0, dRetryList + w0, nWorkItems, dLongScratch + (uint64_t)g * scratchLanes * waveWords, 1, declinedBlocks,
					dRoundTrace + groupTraceBegin[g], cursor + 1, traceBudget, dLongWorkResults + w0, dLongCursor + 8, cursor + 2, 6u, cursor + 3);
			} else if (useLane) {
				launchLongExtendLane(q, G->dev, G->devTables, R->devMasks, lcfg, dLongWork + w0, dOrder + w0, nWorkItems, (uint8_t*)(dLongScratch + (uint64_t)g * scratchLanes * waveWords), scratchLanes * waveWords * 8,
					dRoundTrace + groupTraceBegin[g], cursor + 1, traceBudget, dLongWorkResults + w0, dLongCursor + 8);
			} else if (nVector) {
				if (!st->splitStream) { createStream(&st->splitStream, 1); for (auto& e : st->splitEv) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); }
				const uint32_t nScalar = nWorkItems - nVector;
				HIP_CHECK(hipEventRecord(st->splitEv[0], q));
				HIP_CHECK(hipStreamWaitEvent(st->splitStream, st->splitEv[0], 0));
				launchLongExtend(st->splitStream, G->dev, G->devTables, R->devMasks, lcfg, dLongWork + w0, dOrder + w0 + nScalar, nVector, dLongScratch + (uint64_t)nScalar * waveWords, vectorTeam, nVector / vectorTeam,
					dRoundTrace + groupTraceBegin[g], cursor + 1, traceBudget, dLongWorkResults + w0, dLongCursor + 8, cursor + 2, 0, nullptr, nullptr, nullptr);
				HIP_CHECK(hipEventRecord(st->splitEv[1], st->splitStream));
				launchLongExtend(q, G->dev, G->devTables, R->devMasks, lcfg, dLongWork + w0, dOrder + w0, nScalar, dLongScratch, 1, nScalar,
					dRoundTrace + groupTraceBegin[g], cursor + 1, traceBudget, dLongWorkResults + w0, dLongCursor + 8, cursor + 2, 0, nullptr, dRetryList + w0, cursor + 3);
				HIP_CHECK(hipStreamWaitEvent(q, st->splitEv[1], 0));
			} else
#else
			const bool useSm = false, useLane = false;
#endif
			launchLongExtend(q, G->dev, G->devTables, R->devMasks, lcfg, dLongWork + w0, dOrder + w0, nWorkItems, dLongScratch + (uint64_t)g * scratchLanes * waveWords, team, blocks,
				dRoundTrace + groupTraceBegin[g], cursor + 1, traceBudget, dLongWorkResults + w0, dLongCursor + 8, cursor + 2, 0, nullptr, team == 1 ? dRetryList + w0 : nullptr, cursor + 3);
			if (team == 1 && !useLane) {
				// extensions whose band outgrew the 64-entry register tables: second try with the LDS/HBM tables (two lanes per wave,
				// 28 + 228 entries); waves whose items are fine leave at once. Beyond that the read goes to the plain-layout fallback.
				// (those items are listed first - almost always none - so that the retry is a handful of waves that fetch from the list, not
				// one wave per pair of work items that looks at a status and leaves: that cost 0.5-2 ms of every round)
				if (useSm || (uint64_t)blocks * team < nWorkItems) {   // (the list is the extension kernel's own, unless the state-machine path or persistent waves used the slot counter)
					launchZeroWords(q, cursor + 2, useSm ? 2 : 1);   // [2] next slot, [3] length of the retry list
#ifdef GC_EXPERIMENTS
					if (useSm) launchLongRetryList(q, dLongWorkResults + w0, nWorkItems, EXT_LDS_CAP, dRetryList + w0, cursor + 3);
#endif
				}
				uint32_t retryBlocks = std::min<uint32_t>(128, (uint32_t)std::max<uint64_t>(1, (scratchLanes - 64) / 2));   // (persistent waves over a list that is almost always empty on 10 kb reads; 50 kb CLR-like reads on a genome-sized graph list a few dozen per round, and one such extension lasts 10-50 ms)
				launchLongExtend(q, G->dev, G->devTables, R->devMasks, lcfg, dLongWork + w0, dRetryList + w0, nWorkItems, dLongScratch + (uint64_t)g * scratchLanes * waveWords, 2, retryBlocks,
					dRoundTrace + groupTraceBegin[g], cursor + 1, traceBudget, dLongWorkResults + w0, dLongCursor + 8, cursor + 2, EXT_LDS_CAP, cursor + 3);
			}
			HIP_CHECK(hipEventRecord(ev1, q));
			roundExtendDone = ev1;
			launchLongMerge(q, G->dev, dLongJobs + r0, (uint32_t)nG, dLongSeeds, dCandSeed + w0, dLongWorkResults + w0, dRoundTrace + groupTraceBegin[g], maxAlignments, dLongState + r0, dLongAlns, dLongCells, dLongCursor, cellBudget);
			lastWork = nWorkItems;
			// no wait here: the next round's select / order / publish queue up right behind the merge, and the only host round trip per
			// round is the work count above (with a second wait after the merge the stream drained twice per round, and each refill
			// waited behind whatever other batches had queued on the device)
			timedRounds++;
			groupRounds[g]++;
		}
		// (the last round's count has come down: every extension kernel of the pass is complete, the scratch is free)
		if (longTokenDrop) longTokenDrop();
		// the per-read results go straight into pinned host memory (the kernel writes them across PCIe): a copy-engine transfer here queued behind
		// the other batch's bulk downloads for 30-50 ms while this pass still held the device's whole-read token
		launchLongFinish(q, (uint32_t)nG, dLongState + r0, hLongResults + r0);
		const double dbgT1 = nowUs();
		syncStream(q);
		if (getenv("GC_DEBUG_TIMES")) fprintf(stderr, "[gc times] whole-read rounds: %.1f ms in all, %.1f ms waiting for the rounds' work counts, %.1f ms in the last wait, %d rounds\n", (nowUs() - dbgT0) / 1e3, dbgWaitUs / 1e3, (nowUs() - dbgT1) / 1e3, timedRounds);
		for (int k = std::max(0, timedRounds - LONG_EVENT_RING); k < timedRounds; k++) collect(k % LONG_EVENT_RING);
	}

	void finishLongGroups()   // after the group threads joined
	{
		double us = 0; uint32_t rounds = 0;
		for (uint32_t g = 0; g < nGroups; g++) { us += groupExtendUs[g]; rounds = std::max(rounds, groupRounds[g]); }
		res->kernel_us[4] = us;
		res->counters_long[6] = rounds;
	}

	uint64_t longFallback()   // reads whose band did not fit the wave layout's tables are rerun with the plain-layout kernel; returns how many
	{
		syncStream(ls);
		std::vector<uint32_t> redo;
		const bool forceAll = getenv("GC_LONG_FORCE_FALLBACK") != nullptr;   // test hook: run every read through the plain-layout kernel too
		// status 5: a slice with more nodes than the wave tables hold; status 2: an extension with more tiles / trace cells than its scratch
		// (the plain-layout kernel below gets four times the room)
		for (uint64_t r = 0; r < n; r++) if (hLongResults[r].status == 5 || hLongResults[r].status == 2 || forceAll) redo.push_back((uint32_t)r);
		if (!redo.empty()) {
			ExtendConfig fcfg = lcfg;
			fcfg.maxItems = 4 * lcfg.maxItems; fcfg.maxTrace = 2 * lcfg.maxTrace; fcfg.maxPending = 4 * lcfg.maxPending;
			uint64_t lslab = longSlabBytes(fcfg);
			// The reruns append their cells to the pool the rounds have filled, and the pool is sized by use: a rerun that finds it full answers status 4. The round loop's
			// overflow runs the pass again (growLongCells); here the pool grows IN PLACE - a larger block, the cells already written copied over - and only the reads that were
			// refused run again, in this batch (r5 flagged them and gave the room to the stream's next batch: a read's output depended on its stream's history - ADVICE r5)
			std::vector<uint32_t> now = redo;
			for (int attempt = 0; !now.empty(); attempt++) {
				std::vector<LongJob> subJobs(now.size());
				for (size_t i = 0; i < now.size(); i++) subJobs[i] = hJobs[now[i]];
				uint64_t lanes = (now.size() + 63) / 64 * 64;
				LongJob* dSubJobs = st->longJobsFallback.reserve<LongJob>(now.size());
				LongReadResult* dSubResults = st->longResultsFallback.reserve<LongReadResult>(now.size());
				uint8_t* dSlab = st->longScratchFallback.reserve<uint8_t>(lanes * lslab);
				HIP_CHECK(hipMemcpyAsync(dSubJobs, subJobs.data(), now.size() * sizeof(LongJob), hipMemcpyHostToDevice, ls));
				launchLongPass(ls, G->dev, G->devTables, G->devIupac, fcfg, dSubJobs, (uint32_t)now.size(), dLongSeeds, R->devBases, R->totalBases, (uint32_t)P->min_cluster_size, maxAlignments,
					dSlab, lslab, dLongCells, dLongCursor, cellBudget, dLongAlns, dSubResults, dLongCursor + 8);
				std::vector<LongReadResult> subResults(now.size());
				HIP_CHECK(hipMemcpyAsync(subResults.data(), dSubResults, now.size() * sizeof(LongReadResult), hipMemcpyDeviceToHost, ls));
				HIP_CHECK(hipMemcpyAsync(hLongSmall, dLongCursor, sizeof(unsigned long long), hipMemcpyDeviceToHost, ls));
				syncStream(ls);
				std::vector<uint32_t> refused;
				for (size_t i = 0; i < now.size(); i++) { hLongResults[now[i]] = subResults[i]; if (subResults[i].status == 4) refused.push_back(now[i]); }
				if (refused.empty() || cellPoolPinned || attempt >= 3) break;   // (a pinned pool flags the reads: the caller asked for that much and no more)
				const uint64_t used = std::min<uint64_t>(hLongSmall[0], cellBudget), next = std::min<uint64_t>(256, st->longCellsPerBase * 2);
				if (next == st->longCellsPerBase || cellBudgetFor(next) * sizeof(LongCell) > (64ull << 30)) break;
				if (getenv("GC_DEBUG_TIMES")) fprintf(stderr, "[gc mem] the plain-layout reruns of %zu reads found the merged-trace pool full: %llu -> %llu cells per read base, %.2f G cells kept\n", refused.size(), (unsigned long long)st->longCellsPerBase, (unsigned long long)next, used / 1e9);
				DeviceBuffer larger;
				LongCell* dLarger = larger.reserve<LongCell>(cellBudgetFor(next), true);
				if (used) HIP_CHECK(hipMemcpyAsync(dLarger, dLongCells, used * sizeof(LongCell), hipMemcpyDeviceToDevice, ls));
				const unsigned long long cursor = used;   // (the refused requests are given back: the reruns ask again)
				HIP_CHECK(hipMemcpyAsync(dLongCursor, &cursor, sizeof(cursor), hipMemcpyHostToDevice, ls));
				syncStream(ls);
				std::swap(st->longCells.ptr, larger.ptr); std::swap(st->longCells.bytes, larger.bytes);
				st->longCellsPerBase = next;
				cellBudget = cellBudgetFor(next);
				dLongCells = (LongCell*)st->longCells.ptr;
				now.swap(refused);
			}
		}
		if (n) HIP_CHECK(hipMemcpyAsync(hLongAlns, dLongAlns, n * maxAlignments * sizeof(LongAln), hipMemcpyDeviceToHost, ls));
		HIP_CHECK(hipMemcpyAsync(hLongSmall, dLongCursor, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ls));
		syncStream(ls);
		return (uint64_t)redo.size();
	}

	// selection of a subset of reads' whole-read alignments and the NW distance of the best one (src/Aligner.cpp:636-654): launch ...
	void decideLongReads(const std::vector<uint32_t>& subset, int slot, hipStream_t q, const std::function<uint32_t(uint32_t)>& nAlnOf, bool usePool = true)
	{
		// the reference re-sorts its alignment list by alignmentStart after every accepted alignment
		// (src/GraphAligner.h:183); replaying that on the acceptance-ordered list gives its final order. Then the
		// GreedyLength selection (src/Aligner.cpp:636-639, src/AlignmentSelection.cpp:12-50) with the same unstable sort.
		auto selectOne = [&](size_t i, size_t) {
			const uint32_t r = subset[i];
			ReadGlue& gl = glue[r];
			gl.longAlns.clear();
			gl.longSelected.clear();
			const uint32_t nAln = nAlnOf(r);
			for (uint32_t a = 0; a < nAln; a++) {
				gl.longAlns.push_back(hLongAlns[(uint64_t)r * maxAlignments + a]);
				std::sort(gl.longAlns.begin(), gl.longAlns.end(), [](const LongAln& l, const LongAln& rr) { return l.start < rr.start; });
			}
			struct Item { uint32_t start, end, score, index; };
			std::vector<Item> sorted;
			const size_t readLen = R->offsets[r + 1] - R->offsets[r];
			for (uint32_t a = 0; a < gl.longAlns.size(); a++) {
				// --E-cutoff: SelectECutoff runs before the greedy selection and keeps the list's order (src/AlignmentSelection.cpp:57-61,91-99)
				if (!evalueModel.keeps(P->e_cutoff, hg.SizeInBP(), readLen, gl.longAlns[a].end - gl.longAlns[a].start, gl.longAlns[a].score)) continue;
				sorted.push_back(Item { gl.longAlns[a].start, gl.longAlns[a].end, gl.longAlns[a].score, a });
			}
			std::sort(sorted.begin(), sorted.end(), [](const Item& l, const Item& rr) {
				if ((l.end - l.start) > (rr.end - rr.start)) return true;
				if ((rr.end - rr.start) > (l.end - l.start)) return false;
				return l.score < rr.score;
			});
			auto incompatible = [](const Item& l, const Item& rr) {
				float minOverlapLen = std::min(l.end - l.start, rr.end - rr.start) * 0.05f;
				size_t ls = l.start, le = l.end, rs = rr.start, re = rr.end;
				if (ls > rs) { std::swap(ls, rs); std::swap(le, re); }
				int overlap = 0;
				if (le > rs) overlap = (int)(le - rs);
				return overlap > minOverlapLen;
			};
			std::vector<Item> kept;
			for (const Item& it : sorted) {
				bool ok = true;
				for (const Item& k : kept) if (incompatible(it, k)) { ok = false; break; }
				if (ok) { kept.push_back(it); gl.longSelected.push_back(it.index); }
			}
		};
		if (usePool) pool.run(subset.size(), selectOne); else for (size_t i = 0; i < subset.size(); i++) selectOne(i, 0);
		auto& D = st->edLong[slot];
		D.nPairs = 0;
		D.pairRead.clear();
		if (!P->edit_distances || subset.empty()) return;
		// edit distance of the best whole-read alignment's path against the read (edlibAlign at src/Aligner.cpp:645)
		const size_t m = subset.size();
		PathSeqJob* hJobsPS = D.hJobs.reserve<PathSeqJob>(m);
		EdPair* hPairs = D.hPairs.reserve<EdPair>(m);
		int64_t* hOut = D.hOut.reserve<int64_t>(m);
		uint64_t nLetters = 0;
		uint32_t nPairs = 0;
		for (size_t i = 0; i < m; i++) {
			const uint32_t r = subset[i];
			const ReadGlue& gl = glue[r];
			if (gl.longSelected.empty()) { hJobsPS[i] = PathSeqJob { 0, nLetters, 0, 0, 0, 0 }; continue; }
			const LongAln& al = gl.longAlns[gl.longSelected[0]];
			uint32_t len = (uint32_t)(R->offsets[r + 1] - R->offsets[r]);
			uint32_t cap = 2 * al.traceLen + 256;
			hJobsPS[i] = PathSeqJob { al.traceOff, nLetters, al.traceLen, cap, 0, 0 };
			// the alignment itself bounds the distance: its edits plus the unaligned read ends
			hPairs[nPairs++] = EdPair { nLetters, 0, (uint32_t)i, r, al.score + al.start + (len - std::min(len, al.end)) + 8 };
			D.pairRead.push_back(r);
			nLetters += cap;
		}
		PathSeqJob* dJobsPS = D.jobs.reserve<PathSeqJob>(m);
		char* dLetters = D.letters.reserve<char>(nLetters);
		uint32_t* dLettersLen = D.lettersLen.reserve<uint32_t>(m);
		EdPair* dPairs = D.pairs.reserve<EdPair>(m);
		int64_t* dOut = D.out.reserve<int64_t>(m);
		HIP_CHECK(hipMemcpyAsync(dJobsPS, hJobsPS, m * sizeof(PathSeqJob), hipMemcpyHostToDevice, q));
		launchLongPathSeq(q, G->dev, dJobsPS, (uint32_t)m, dLongCells, dLetters, dLettersLen);
		auto readLenOf = [R = R](uint32_t r) { return (uint32_t)(R->offsets[r + 1] - R->offsets[r]); };
		launchEditDistances(D.run, q, hPairs, hOut, nPairs, dPairs, dOut, R->devEdReads, R->devBases, R->devEqMasks, dLetters, dLettersLen, readLenOf, true);   // (k: the alignment's own bound)
		D.nPairs = nPairs;
		decisionPtr[slot] = DecisionPointers { hPairs, hOut, dPairs, dOut, dLetters, dLettersLen };
	}

	// ... and collect
	void finishLongDecision(int slot)
	{
		auto& D = st->edLong[slot];
		if (!D.nPairs) return;
		const DecisionPointers& p = decisionPtr[slot];
		finishEditDistances(D.run, st->longStream, p.hPairs, p.hOut, D.nPairs, p.dPairs, p.dOut, R->devEdReads, R->devBases, R->devEqMasks, p.dLetters, p.dLettersLen);
		for (uint32_t i = 0; i < D.nPairs; i++) {
			ReadGlue& gl = glue[D.pairRead[i]];
			gl.longEditDistance = p.hOut[i];
			if (p.hOut[i] < -1) { gl.longEditDistance = -1; gl.capacityExceededLong = true; }   // outside the NW kernel's range: flagged, no distance
		}
		D.nPairs = 0;
	}

	// What follows the rounds: fallback reruns, the reference's `cont` rule, selection and the NW distance of the best whole-read alignment.
	void afterLongPass()
	{
		uint64_t rerun = longFallback();
		res->counters_long[7] = rerun;   // reads that needed the plain-layout fallback kernel
		for (int i = 0; i < 6; i++) res->counters_long[i] = hLongSmall[8 + i];   // same units as counters[]
#ifdef GC_STAMPS
		{
			static const char* names[11] = { "slice prologue", "pop+prev lookup", "tile columns", "item store", "edge pushes", "slice epilogue", "bt slice change", "bt item loads", "bt recompute", "bt corner", "bt walk" };
			double total = 0;
			for (int i = 0; i < 11; i++) total += (double)hLongSmall[16 + i];
			for (int i = 0; i < 11; i++) fprintf(stderr, "[gc stamps] %-16s %6.2f%%  %.3e lane-cycles\n", names[i], 100.0 * hLongSmall[16 + i] / (total > 0 ? total : 1), (double)hLongSmall[16 + i]);
		}
#endif
#ifdef GC_SM_STAMPS
		{
			static const char* names[5] = { "B (tile boundary)", "COL (column)", "BT (bt boundary)", "WALK (cell)", "housekeeping+vote" };
			double total = 0;
			for (int i = 0; i < 5; i++) total += (double)hLongSmall[16 + i];
			for (int i = 0; i < 5; i++) fprintf(stderr, "[gc sm stamps] %-18s %6.2f%% of wave-cycles, %.3e executions, %.0f cycles each, %.2f lanes served per execution\n", names[i], 100.0 * hLongSmall[16 + i] / (total > 0 ? total : 1),
				(double)hLongSmall[21 + i], (double)hLongSmall[16 + i] / std::max<double>(1, (double)hLongSmall[21 + i]), (double)hLongSmall[26 + i] / std::max<double>(1, (double)hLongSmall[21 + i]));
		}
#endif
		if (const char* env = getenv("GC_TEST_FAIL_LONG")) {   // test hook shared with the oracle: this read's whole-read pass "asserts"
			long idx = atol(env);
			if (idx >= 0 && (uint64_t)idx < n) hLongResults[idx].status = 1;
		}
		// A whole-read pass that trips one of the reference's live asserts leaves the read with nothing: align_fn's catch sets
		// `cont` (src/Aligner.cpp:591), which is declared once per read (:529) and makes the fragment loop skip every anchor
		// (:702-703); the alignments found before the throw are lost with the exception.
		for (uint64_t r = 0; r < n; r++) if (hLongResults[r].status == 1) { hLongResults[r].nAlignments = 0; glue[r].longFailed = true; }
		// capacities of this library, not of the reference: 2 extension scratch (even with the fallback's four-fold room), 3 more alignments than
		// maxAlignments, 4 the merged-trace cell pool (GC_LONG_CELLS_PER_BASE). The read keeps what was found up to there and is flagged.
		for (uint64_t r = 0; r < n; r++) if (hLongResults[r].status >= 2 && hLongResults[r].status <= 4) glue[r].capacityExceededLong = true;
		{
			std::vector<uint32_t> all(n);
			for (uint64_t r = 0; r < n; r++) all[r] = (uint32_t)r;
			decideLongReads(all, 0, st->longStream, [&](uint32_t r) { return hLongResults[r].nAlignments; });
			encodeOutputStart();   // (the selection is known; its waits overlap the NW kernels the decision has just queued)
			finishLongDecision(0);
		}
	}

	// ---------------- the pass gets its own host thread and stream from here on
	void startWholeReadPass()
	{
		tLongWall0 = nowUs();
		if (P->long_pass) {
			int device = 0;
			HIP_CHECK(hipGetDevice(&device));
			for (uint32_t g = 0; g < longGroups; g++)
				longThreads.emplace_back([&, device, g]() {
					// (declared outside the try block: on an exception the catch below waits for the pass's kernels BEFORE the token - and with it the device's shared scratch - is released)
					TokenHold token;
					try {
						HIP_CHECK(hipSetDevice(device));
						int tokenMode = getenv("GC_LONG_TOKEN") ? atoi(getenv("GC_LONG_TOKEN")) : 1;   // 0 none, 1 one pass at a time (2, experiments build: one round's extension kernel at a time)
#ifndef GC_EXPERIMENTS
						if (tokenMode != 0) tokenMode = 1;
#endif
						const double tTokenAsk = nowUs();
						const bool early = expEnv("GC_LONG_TOKEN_EARLY") && atoi(expEnv("GC_LONG_TOKEN_EARLY")) == 1;
						bool held = false;   // between take and drop (with or without a token to hold: GC_LONG_TOKEN=0 has none)
						auto stampBegin = [&]() { double now = nowUs(), seen = longWallBeginUs.load(); while ((seen == 0.0 || now < seen) && !longWallBeginUs.compare_exchange_weak(seen, now)) {} };
						auto stampEnd = [&]() { double now = nowUs(), seen = longWallEndUs.load(); while (now > seen && !longWallEndUs.compare_exchange_weak(seen, now)) {} };
						auto take = [&, device, tokenMode, tTokenAsk]() {
							if (held) return;
							const double tAsk = nowUs();
							if (tokenMode == 1 && longGroups == 1) {
								const int passesSideBySide = longTokenCount(n, st->batchesDone);
								token.lock(g_longPassToken[device & 15], passesSideBySide, passesSideBySide == 1);   // (a pass that fills the chip: alone on the device)
								if (token.slot > 0 && shareLongScratch) {   // the second token's scratch is only grown when the device has the room: otherwise this pass waits for the first token like any other
									const DeviceBuffer& have = g_longScratch[device & 15].buffer[token.slot];
									size_t freeBytes = 0, totalBytes = 0;
									const uint64_t need = longScratchWords * sizeof(unsigned long long);
									if (have.bytes < need && (hipMemGetInfo(&freeBytes, &totalBytes) != hipSuccess || freeBytes + have.bytes < need + need / 8 + (6ull << 30))) { token.unlock(); token.lock(g_longPassToken[device & 15], 1, false); }
								}
							}
							if (shareLongScratch && tokenMode == 1) {
								if (!token.owns_lock()) throw std::runtime_error("internal: shared whole-read scratch without the token");
								try {
									longScratchOfToken = g_longScratch[device & 15].buffer[token.slot].reserve<unsigned long long>(longScratchWords);
								} catch (const DeviceError&) {
									// (the check above and this reservation are not one step: another stream may have taken the memory in between) - the second token's scratch does not
									// fit after all: this pass takes its turn on the first token's instead of failing the batch
									if (token.slot == 0) throw;
									(void)hipGetLastError();
									token.unlock();
									token.lock(g_longPassToken[device & 15], 1, false);
									longScratchOfToken = g_longScratch[device & 15].buffer[0].reserve<unsigned long long>(longScratchWords);
								}
							}
							held = true;
							if (getenv("GC_DEBUG_TIMES")) fprintf(stderr, "[gc token] stream %p pass began %.1f asked %.1f got %.1f (call began %.1f)\n", (void*)st, tTokenAsk / 1e3, tAsk / 1e3, nowUs() / 1e3, tCall / 1e3);
							stampBegin();   // (whole_read_pass_wall: from the token to its release)
						};
						auto drop = [&]() {
							if (!held) return;
							held = false;
							stampEnd();
							if (getenv("GC_DEBUG_TIMES")) fprintf(stderr, "[gc token] stream %p released %.1f\n", (void*)st, nowUs() / 1e3);
							token.unlock();   // the next batch's pass may start; what follows is this batch's own tail
						};
						if (longGroups == 1 && !early) { longTokenTake = take; longTokenDrop = drop; }
						else if (longGroups == 1) take();
						else stampBegin();
						runLongGroup(g);
						while (longGroups == 1 && growLongCells()) runLongGroup(g);   // the cell pool overflowed: again, with room
						if (longGroups == 1) { longTokenTake = nullptr; longTokenDrop = nullptr; drop(); }
						else stampEnd();
						if (longPostInThread) afterLongPass();
						passThreadCpuMs = threadCpuMs();   // (the thread's whole life: it is created per batch)
					} catch (...) {
						longErrors[g] = std::current_exception();
						if (longGroups == 1) { longTokenTake = nullptr; longTokenDrop = nullptr; }   // (they refer to this thread's locals)
						// the token is released when this lambda returns: a kernel of this pass may still be writing to the shared scratch
						if (g < st->groupStreams.size()) (void)hipStreamSynchronize(st->groupStreams[g]);
						(void)hipStreamSynchronize(st->longStream);
					}
				});
		}
	}

	// ---------------- fragment windows, k_build_fragment_work, K3 / K3b in lazy rounds, K4, k_stitch - queued on the main stream
	void fragmentPipeline()
	{
		double tLongStarted = nowUs();
		double tWindows = tLongStarted, tReserved = tLongStarted;
		if (!deviceGlue && !poolsSized) {   // (this stage may run twice - fragmentPoolsOverflowed - and the block below builds sums and arrays that the second run finds as they are: ADVICE r5)
		pool.run(n, [&](size_t r, size_t) {
			ReadGlue& gl = glue[r];
			if (gl.seeds.empty()) return;
			gc::fragmentWindows(gl.seeds, R->offsets[r + 1] - R->offsets[r], (size_t)P->split_len, (size_t)P->split_gap, gl.windows);
		});
		tWindows = nowUs();
		for (uint64_t r = 0; r < n; r++) {
			glue[r].slotBegin = nSlots;
			glue[r].fragBegin = nFrags;
			glue[r].seedBegin = nSeedsTotal;
			glue[r].nSeedsR = (uint32_t)glue[r].seeds.size();
			glue[r].nWindows = (uint32_t)glue[r].windows.size();
			for (const auto& w : glue[r].windows) nSlots += w.sr - w.sl;
			nFrags += glue[r].windows.size();
			nSeedsTotal += glue[r].seeds.size();
		}
		// The per-slot records (seed in fragment order + its two extensions) are expanded on the device (k_build_fragment_work) from what the
		// host decides: the read's seeds in the reference's order after its sort by position, and the windows.
		frags = st->hFrags.reserve<Fragment>(nFrags);
		fragFirstSeed = st->hFragFirstSeed.reserve<uint32_t>(nFrags);
		readSeeds = st->hReadSeeds.reserve<FragSeed>(nSeedsTotal);
		jobs = st->hJobs.reserve<ReadChainJob>(n);
		std::vector<uint64_t> traceBudgets(pool.size(), 0);
		std::vector<uint32_t> windowSeeds(pool.size(), 0);   // per worker: the most seeds a window holds
		tReserved = nowUs();
		pool.run(n, [&](size_t r, size_t worker) {
			const ReadGlue& gl = glue[r];
			size_t len = R->offsets[r + 1] - R->offsets[r];
			for (size_t k = 0; k < gl.seeds.size(); k++) readSeeds[gl.seedBegin + k] = FragSeed { gl.seeds[k].node, gl.seeds[k].offset, gl.seeds[k].seqPos, gl.seeds[k].goodness };
			uint64_t slot = gl.slotBegin;
			uint64_t budget = 0;
			for (size_t f = 0; f < gl.windows.size(); f++) {
				const gc::FragmentWindow& w = gl.windows[f];
				Fragment& fr = frags[gl.fragBegin + f];
				fr.read = (uint32_t)r;
				fr.l = w.l;
				fr.seedBegin = (uint32_t)slot;
				fragFirstSeed[gl.fragBegin + f] = (uint32_t)(gl.seedBegin + w.sl);
				for (uint32_t k = w.sl; k < w.sr; k++, slot++) {
					// trace cells the two extensions of this seed may need: backward p rows, forward split_len - 1 - p (src/GraphAligner.h:499-511)
					const uint32_t p = gl.seeds[k].seqPos - w.l, q = (uint32_t)P->split_len - 1 - p;
					budget += (p ? p + 24 : 0) + (q ? q + 24 : 0);
				}
				fr.seedEnd = (uint32_t)slot;
				windowSeeds[worker] = std::max(windowSeeds[worker], w.sr - w.sl);
			}
			traceBudgets[worker] += budget;
			ReadChainJob& job = jobs[r];
			job.slotBegin = (uint32_t)gl.slotBegin;
			job.nSlots = (uint32_t)(slot - gl.slotBegin);
			job.chainBegin = (uint32_t)gl.slotBegin;
			job.nKeys = len >= (size_t)P->split_len ? (uint32_t)((len - P->split_len) / P->split_gap + 1) : 1;
			job.fragBegin = (uint32_t)gl.fragBegin;
			job.nFrags = (uint32_t)gl.windows.size();
		});
		for (uint64_t b : traceBudgets) traceBudget += b;
		for (uint32_t m : windowSeeds) maxWindowSeeds = std::max(maxWindowSeeds, m);
		for (uint64_t r = 0; r < n; r++) maxSlotsPerRead = std::max(maxSlotsPerRead, jobs[r].nSlots);
		}
		if (2 * nSlots >= 0xffffffffull) throw std::runtime_error("batch too large: more than 2^31 fragment seeds; split the batch");
		if (!poolsSized) {
			traceWorst = traceBudget + traceBudget / 4 + (1u << 20);   // every slot's two extensions at full length + room for the extensions that only fit the retry launch's larger trace buffers
			pathWorst = nSlots * 24 + 4096;
			// by use (device glue; GC_POOLS_WORST_CASE=1 and the host glue path keep the worst case): what the stream's earlier batches needed per slot, with 15 % of slack; a
			// stream's first batch starts from a low guess (36 trace cells - 8 bytes each since r6 - and 4 path words per slot of the 103 and 24 the worst case reserves) and
			// runs its fragment pipeline again with what it asked for when that was short - in the warm-up batch, once per stream - so that the pools are never larger than a batch needs
			const bool byUse = deviceGlue && !(getenv("GC_POOLS_WORST_CASE") && atoi(getenv("GC_POOLS_WORST_CASE")) == 1);
			double traceGuess = 36.0, pathGuess = 4.0, slackCells = (double)(1u << 20), slackWords = 4096.0;
			if (const char* env = getenv("GC_POOL_FIRST_GUESS")) { traceGuess = std::max(0.0, atof(env)); pathGuess = traceGuess / 8; slackCells = slackWords = 64; }   // test hook: a stream's first batch outgrows its pools
			traceBudget = byUse ? std::min<uint64_t>(traceWorst, (uint64_t)((double)nSlots * (st->traceCellsPerSlot > 0 ? st->traceCellsPerSlot * 1.15 : traceGuess) + slackCells)) : traceWorst;
			pathCapacity = byUse ? std::min<uint64_t>(pathWorst, (uint64_t)((double)nSlots * (st->pathWordsPerSlot > 0 ? st->pathWordsPerSlot * 1.15 : pathGuess) + slackWords)) : pathWorst;
			poolsSized = true;
		}
		ChainCaps caps { 1, 1, 1, 1 };
		caps.capAnchors = std::max(1u, maxSlotsPerRead);
		caps.capEndpoints = (uint32_t)std::min<uint64_t>(0x7fffffffull, (uint64_t)caps.capAnchors * G->maxPathsPerNode);   // entries: one per path through an anchor's end node
		caps.capTable = std::max(1u, G->maxMpcWidth);
		caps.capBack = (uint32_t)std::min<uint64_t>(0x7fffffffull, (uint64_t)caps.capAnchors * ((uint64_t)G->maxBackPerNode + G->maxPathsPerNode));   // threshold lists: backward links + paths of the start node
		res->host_us[0] = nowUs() - tGlue;
		if (getenv("GC_DEBUG_TIMES")) fprintf(stderr, "[gc cpu] %.0f ms of process CPU up to the end of the host glue\n", processCpuMs() - cpuCall);
		if (getenv("GC_DEBUG_TIMES")) fprintf(stderr, "[gc times] seed expand+order %.1f ms, whole-read setup %.1f ms, fragment windows+arrays %.1f ms (windows %.1f, sizes+buffers %.1f, arrays %.1f)\n", (tOrdered - tGlue) / 1e3, (tLongStarted - tOrdered) / 1e3, (nowUs() - tLongStarted) / 1e3,
			(tWindows - tLongStarted) / 1e3, (tReserved - tWindows) / 1e3, (nowUs() - tReserved) / 1e3);

		// ---------------- K3 / K3b / K4
		tDev = nowUs();
		evIdx = 2;                                   // (this stage may run twice: see fragmentPoolsOverflowed)
		launchZeroWords(stream, dCursors + 1, 2);   // [1] trace pool cursor, [2] anchor path pool cursor ([0] is the seed lookup's)
		launchZeroWords(stream, dCounters, 8);
		ExtendConfig cfg;
		cfg.bandwidth = P->bandwidth;
		cfg.maxSlices = 3;
		cfg.maxItems = 72;
		cfg.maxPending = 48;
		cfg.maxTrace = 192;
		cfg.maxItems = (uint32_t)std::max<int64_t>(8, capacityOr("GC_EXT_MAX_ITEMS", P->capacity.ext_max_items, cfg.maxItems));
		cfg.maxPending = (uint32_t)std::max<int64_t>(8, capacityOr("GC_EXT_MAX_PENDING", P->capacity.ext_max_pending, cfg.maxPending));
		cfg.maxTrace = (uint32_t)std::max<int64_t>(64, capacityOr("GC_EXT_MAX_TRACE", P->capacity.ext_max_trace, cfg.maxTrace));
		nWork = (uint32_t)(2 * nSlots);
		uint64_t slabBytes = extendSlabBytes(cfg);
		uint32_t lanes = extendGridLanes(nWork);
		ExtItem* dWork = st->work.reserve<ExtItem>(nWork);
		dResults = st->results.reserve<ExtResult>(nWork);
		// r6: fragments of up to 65 bases (one slice per extension) go through the lockstep kernel (gc_extend_frag.hip); what it declines and every longer fragment through the
		// plain-layout kernel on per-lane slabs. GC_EXTEND_SLAB=1: the plain-layout kernel for everything, as up to r5 (A/B)
		const bool fragKernel = P->split_len <= 65 && !(getenv("GC_EXTEND_SLAB") && atoi(getenv("GC_EXTEND_SLAB")) == 1);
		uint8_t* dScratch = fragKernel ? nullptr : st->scratch.reserve<uint8_t>((uint64_t)lanes * slabBytes);
		const uint32_t fragWaves = fragKernel ? extendFragWaves() : 0;
		uint4* dFragItems = fragKernel ? st->fragItems.reserve<uint4>(extendFragScratchBytes(fragWaves) / sizeof(uint4)) : nullptr;
		uint32_t* dFragRetry = fragKernel ? st->fragRetryList.reserve<uint32_t>(std::max<uint32_t>(1, nWork)) : nullptr;
		unsigned long long* dFragClaims = st->fragClaims.reserve<unsigned long long>(16);   // ([8..15]: the profiling build's section cycles)   // per extension round: [2 k] the waves' claim cursor, [2 k + 1] the number of declined items
		if (fragKernel) launchZeroWords(stream, dFragClaims, 16);
		unsigned long long* hFragDeclined = st->hFragDeclined.reserve<unsigned long long>(1);
		const FragReads fragReads { R->devMasks, R->devMaskOff, R->devMaskWords, R->devOffsets, R->totalBases };
		if (st->poolsRerun && poolReruns == 0) {   // (the batch after a rerun: see DeviceBuffer::shrinkTo)
			st->tracePool.shrinkTo(traceBudget * sizeof(PoolCell));
			st->pathPool.shrinkTo(pathCapacity * sizeof(uint32_t));
			st->poolsRerun = false;
		}
		dTrace = st->tracePool.reserve<PoolCell>(traceBudget, true);
		if (!deviceGlue) dFrags = st->frags.reserve<Fragment>(nFrags);
		FragSeed* dFragSeeds = st->fragSeeds.reserve<FragSeed>(nSlots);
		dAnchors = st->anchors.reserve<AnchorRec>(nSlots);
		dFragStatus = st->fragStatus.reserve<uint32_t>(nFrags);
		dFragExtended = st->fragExtended.reserve<uint32_t>(nFrags);
		dReadTies = st->readTies.reserve<uint32_t>(n);
		if (n) HIP_CHECK(hipMemsetAsync(dReadTies, 0, n * sizeof(uint32_t), stream));
		dPathPool = st->pathPool.reserve<uint32_t>(pathCapacity, true);
		if (!deviceGlue) dJobs = st->jobs.reserve<ReadChainJob>(n);
		dChainOut = st->chainOut.reserve<uint32_t>(nSlots);
		dChainLen = st->chainLen.reserve<uint32_t>(n);
		dChainScore = st->chainScore.reserve<unsigned long long>(n);
		dChainStatus = st->chainStatus.reserve<uint32_t>(n);
		uint32_t fewestSlots = 0xffffffffu;
		for (uint64_t r = 0; r < n; r++) fewestSlots = std::min(fewestSlots, jobs[r].nSlots);
		if (!n) fewestSlots = 0;
		const bool forceChainScratch = getenv("GC_CHAIN_FORCE_SCRATCH") != nullptr;
		// both launches index the scratch by block (the LDS launch keeps its threshold lists there); a batch of long reads has no LDS launch
		uint32_t chainBlocks = chainLdsLaunch(fewestSlots, forceChainScratch) ? std::max(chainGridBlocks((uint32_t)n), chainScratchBlocks((uint32_t)n)) : chainScratchBlocks((uint32_t)n);
		uint8_t* dChainScratch = st->chainScratch.reserve<uint8_t>((uint64_t)std::max(1u, chainBlocks) * chainScratchBytes(caps));
		if (!deviceGlue) {
			dReadSeeds = st->readSeeds.reserve<FragSeed>(nSeedsTotal);
			dFragFirstSeed = st->fragFirstSeed.reserve<uint32_t>(nFrags);
			if (nFrags) HIP_CHECK(hipMemcpyAsync(dFrags, frags, nFrags * sizeof(Fragment), hipMemcpyHostToDevice, stream));
			if (nFrags) HIP_CHECK(hipMemcpyAsync(dFragFirstSeed, fragFirstSeed, nFrags * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
			if (nSeedsTotal) HIP_CHECK(hipMemcpyAsync(dReadSeeds, readSeeds, nSeedsTotal * sizeof(FragSeed), hipMemcpyHostToDevice, stream));
		}
		// Lazy extension (default): a seed is extended only when the reference would extend it - when it does not lie on an earlier alignment of its
		// fragment (src/GraphAligner.h:163-173). Round 0 extends every fragment's first seed; k_build_anchors parks the fragments that reach another
		// seed they must extend and queues that seed for the next round (the launches size themselves from counts on the device, no host round
		// trip); three rounds at most, the last parking round queues everything its fragments have left. On cfg2 the reference extends 47 % of the
		// seeds the windows hold. GC_EXT_LAZY=0: every seed is extended up front.
		const bool lazyExtend = !(getenv("GC_EXT_LAZY") && atoi(getenv("GC_EXT_LAZY")) == 0) && nFrags > 0;
		launchBuildFragmentWork(stream, G->dev, dFrags, dFragFirstSeed, (uint32_t)nFrags, dReadSeeds, R->devOffsets, R->totalBases, (uint32_t)P->split_len, dFragSeeds, dWork, lazyExtend ? dResults : nullptr);
		if (n && !deviceGlue) HIP_CHECK(hipMemcpyAsync(dJobs, jobs, n * sizeof(ReadChainJob), hipMemcpyHostToDevice, stream));
		mark();   // 2
		// extensions that outgrew their slab (a dense variant cluster: more tiles, queue entries or trace cells than the common case is sized
		// for) run again in a small grid with 16x the room; lanes whose items are fine only read the status array. What overflows even
		// that is flagged per read (capacity_exceeded), never a failed call.
		ExtendConfig big = cfg;
		auto times16 = [](uint32_t v) { return (uint32_t)std::min<uint64_t>(16ull * v, 0xffffffffull); };   // (saturating: the GC_EXT_* variables are not range-checked like gc_params::capacity)
		big.maxItems = times16(cfg.maxItems); big.maxPending = times16(cfg.maxPending); big.maxTrace = times16(cfg.maxTrace); big.maxSlices = cfg.maxSlices;
		if (const char* env = getenv("GC_EXT_RETRY_MAX_ITEMS")) big.maxItems = (uint32_t)std::max(8, atoi(env));   // test hook: make the retry overflow too
		const uint32_t retryLanes = 2048;
		uint8_t* dRetryScratch = st->scratchRetry.reserve<uint8_t>((uint64_t)retryLanes * extendSlabBytes(big));
		// (r5: every round's extension launches and every k_build_anchors launch sit between an event pair of their own - r4 bracketed "round 0's extensions" and "everything up to
		// the chaining kernel", so the later rounds' k_extend launches were charged to the anchors stage and roofline_other did not follow from the kernel trace)
		nExtendPairs = nAnchorPairs = 0;
		uint32_t nExtendRounds = 0;
		static const uint32_t extendChunkItems = getenv("GC_EXTEND_CHUNK") ? (uint32_t)std::max(0, atoi(getenv("GC_EXTEND_CHUNK"))) : 0u;   // extensions per k_extend launch (0: one launch per round)
		auto extendRound = [&](const ExtSelection& sel) {
			if (nExtendPairs < 4) HIP_CHECK(hipEventRecord(st->fragEv[2 * nExtendPairs], stream));
			if (fragKernel) {
				unsigned long long* claims = dFragClaims + 2 * (nExtendRounds++ & 3u);
				if (nExtendRounds > 4) launchZeroWords(stream, claims, 2);
				launchExtendFrag(stream, G->dev, G->devTables, cfg.bandwidth, dWork, nWork, fragReads, dResults, dFragItems, fragWaves, dTrace, dCursors + 1, traceBudget, dCounters, sel, claims, dFragRetry, claims + 1, dFragClaims + 8);
				ExtSelection declined;
				declined.mode = 2; declined.list = dFragRetry; declined.listCount = claims + 1;
				// What the kernel declined (233 of cfg2's 4.4 M extensions) goes to the plain-layout kernel on the large slabs - when there is anything: its count comes to the host
				// first. A launch of that kernel with nothing to do still waits 2-7 ms for a SIMD to free a quarter of its registers beside the whole-read kernel's resident waves
				// (`gpurun_out/r6_f`: 12 ms per batch for six near-empty launches), the round trip costs the batch a few tens of microseconds while the other batches' kernels run
				HIP_CHECK(hipMemcpyAsync(hFragDeclined, claims + 1, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
				syncStream(stream);
				if (*hFragDeclined > 0)
					launchExtend(stream, G->dev, G->devTables, G->devIupac, big, dWork, nWork, R->devBases, dResults, dRetryScratch, extendSlabBytes(big), dTrace, dCursors + 1, traceBudget, dCounters, EXT_OVERFLOW, retryLanes, declined);
			} else {
				launchExtend(stream, G->dev, G->devTables, G->devIupac, cfg, dWork, nWork, R->devBases, dResults, dScratch, slabBytes, dTrace, dCursors + 1, traceBudget, dCounters, 0, 4096, sel, extendChunkItems);
				launchExtend(stream, G->dev, G->devTables, G->devIupac, big, dWork, nWork, R->devBases, dResults, dRetryScratch, extendSlabBytes(big), dTrace, dCursors + 1, traceBudget, dCounters, EXT_OVERFLOW, retryLanes, sel);
			}
			if (nExtendPairs < 4) { HIP_CHECK(hipEventRecord(st->fragEv[2 * nExtendPairs + 1], stream)); nExtendPairs++; }
		};
		auto anchorsTimed = [&](const AnchorRounds& ar) {
			if (nAnchorPairs < 4) HIP_CHECK(hipEventRecord(st->fragEv[8 + 2 * nAnchorPairs], stream));
			launchBuildAnchors(stream, G->dev, dFrags, (uint32_t)nFrags, dFragSeeds, dResults, dTrace, P->split_len, dAnchors, dFragStatus, dFragExtended, dPathPool, dCursors + 2, pathCapacity, ar, dReadTies);
			if (nAnchorPairs < 4) { HIP_CHECK(hipEventRecord(st->fragEv[8 + 2 * nAnchorPairs + 1], stream)); nAnchorPairs++; }
		};
		if (!lazyExtend) {
			extendRound(ExtSelection());
			mark();   // 3
			anchorsTimed(AnchorRounds());
		} else {
			uint32_t* dLists = st->extLists.reserve<uint32_t>(2ull * nWork);           // two work lists, used in turn
			uint32_t* dPending = st->pendingFrags.reserve<uint32_t>(2ull * nFrags);     // two pending-fragment lists
			uint32_t* dFragNext = st->fragNext.reserve<uint32_t>(nFrags);
			unsigned long long* dRoundCounts = st->roundCounts.reserve<unsigned long long>(4);   // [2k] work list k, [2k+1] pending list k
			launchZeroWords(stream, dRoundCounts, 4);
			ExtSelection first;
			first.mode = 1; first.frags = dFrags; first.nFrags = (uint32_t)nFrags;
			extendRound(first);
			mark();   // 3 (round 0's extensions; the later rounds are charged to the anchors stage)
			const uint32_t nRounds = std::min<uint32_t>(maxWindowSeeds, 3);   // first seeds; the next seed each parked fragment needs; then all that is left of the few still parked
			for (uint32_t round = 0; round < nRounds; round++) {
				const uint32_t cur = round & 1u, nxt = cur ^ 1u;
				if (round > 0) {
					ExtSelection sel;
					sel.mode = 2; sel.list = dLists + (uint64_t)cur * nWork; sel.listCount = dRoundCounts + 2 * cur;
					extendRound(sel);
				}
				launchZeroWords(stream, dRoundCounts + 2 * nxt, 2);
				AnchorRounds ar;
				ar.lazy = 1; ar.round = round; ar.parkAll = round + 2 >= nRounds ? 1 : 0;
				ar.pending = dPending + (uint64_t)cur * nFrags; ar.pendingCount = dRoundCounts + 2 * cur + 1;
				ar.nextList = dLists + (uint64_t)nxt * nWork; ar.nextListCount = dRoundCounts + 2 * nxt;
				ar.nextPending = dPending + (uint64_t)nxt * nFrags; ar.nextPendingCount = dRoundCounts + 2 * nxt + 1;
				ar.fragNext = dFragNext;
				anchorsTimed(ar);
			}
		}
		mark();   // 4
		launchChain(stream, G->dev, dJobs, (uint32_t)n, dAnchors, dFrags, dFragStatus, P->split_len, P->split_gap, caps, dChainScratch, dChainOut, dChainLen, dChainScore, dChainStatus, forceChainScratch, fewestSlots);
		mark();   // 5
		// chain stitching (src/Aligner.cpp:754-822) on the device, right behind the chaining kernel; GC_HOST_STITCH=1 keeps it on the
		// host workers (the path also taken by reads that do not fit the kernel's tables)
		deviceStitch = P->stitch && n > 0 && !(getenv("GC_HOST_STITCH") && atoi(getenv("GC_HOST_STITCH")) != 0);
		if (deviceStitch) {
			stitchDenseCap = stitchDenseWords(nSlots, n);
			uint32_t* dSlotOf = st->stitchSlotOf.reserve<uint32_t>(std::max<uint64_t>(1, nSlots));
			uint32_t* dRegions = st->stitchRegions.reserve<uint32_t>(stitchRegionWords(nSlots, n));
			dStitchNodes = st->stitchNodes.reserve<uint32_t>(stitchDenseCap);
			StitchInfo* dStitchInfo = st->stitchInfo.reserve<StitchInfo>(n);
			unsigned long long* dCursor = st->stitchCursor.reserve<unsigned long long>(1);
			stitchInfo = st->hStitchInfo.reserve<StitchInfo>(n);
			hStitchCursor = st->hStitchCursor.reserve<unsigned long long>(1);
			HIP_CHECK(hipMemsetAsync(dCursor, 0, sizeof(unsigned long long), stream));
			int stitchClass = maxReadLen > 16384 ? 3 : 0;
			if (const char* env = getenv("GC_STITCH_CLASS")) stitchClass = atoi(env) == 3 ? 3 : 0;
#ifdef GC_EXPERIMENTS
			if (getenv("GC_STITCH_SMALL") && atoi(getenv("GC_STITCH_SMALL")) && maxReadLen <= 16384) stitchClass = 1;   // (r4: the half-size search tables, measured and not kept)
#endif
			launchStitch(stream, G->dev, dJobs, (uint32_t)n, dAnchors, dFrags, dFragStatus, dChainOut, dChainLen, dChainStatus, dPathPool, pathCapacity, (long long)P->colinear_gap, dSlotOf,
				dRegions, dStitchNodes, stitchDenseCap, dCursor, dStitchInfo,
				(uint32_t)capacityOr("GC_STITCH_SET_MAX", P->capacity.stitch_set_max, 0), (uint32_t)capacityOr("GC_STITCH_BFS_CAP", P->capacity.stitch_bfs_cap, 0),
				// reads beyond 16 kb: the class whose node set and wide bridge searches live in HBM scratch (r5; gc_stitch.hip) - a 50 kb read's piece holds ~2 500 split nodes, more than the
				// default class's LDS node set, and every read of config 5 used to be stitched by the host. GC_STITCH_CLASS=0 / 3 forces a class (tests, A/B)
				stitchClass, stitchClass == 3 ? st->stitchSpill.reserve<unsigned long long>((uint64_t)stitchSpillBlocks((uint32_t)n) * stitchSpillWordsPerBlock()) : nullptr);
			HIP_CHECK(hipMemcpyAsync(stitchInfo, dStitchInfo, n * sizeof(StitchInfo), hipMemcpyDeviceToHost, stream));
			HIP_CHECK(hipMemcpyAsync(hStitchCursor, dCursor, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
		}

	}

	// the stream learns what its batches use of the two pools; true: this batch needed more than it was given (and less than the worst case): sized again, the stage runs again
	bool fragmentPoolsOverflowed()
	{
		const uint64_t traceNeed = hSmall[1], pathNeed = hSmall[2];
		const bool traceShort = traceNeed > traceBudget && traceBudget < traceWorst, pathShort = pathNeed > pathCapacity && pathCapacity < pathWorst;
		// what the stream learns is a run's use when the pool held it: the cursor of a run that overflowed also counts the requests of the retry launch, which runs every refused
		// extension again (config 5 at 960 Mbp learned 44 cells per slot from such a run where 36 are used: 2.4 GB per batch in flight)
		if (nSlots && !traceShort) st->traceCellsPerSlot = std::max(st->traceCellsPerSlot, (double)std::min(traceNeed, traceWorst) / (double)nSlots);
		if (nSlots && !pathShort) st->pathWordsPerSlot = std::max(st->pathWordsPerSlot, (double)std::min(pathNeed, pathWorst) / (double)nSlots);
		if (!traceShort && !pathShort) return false;
		// (the cursors count every request, the refused ones included - but a fragment whose extension was refused stops asking, so the need seen is a lower bound: a fifth more, and the loop comes back when that is still short)
		if (traceShort) traceBudget = std::min<uint64_t>(traceWorst, traceNeed + traceNeed / 5 + (1u << 20));
		if (pathShort) pathCapacity = std::min<uint64_t>(pathWorst, pathNeed + pathNeed / 5 + 4096);
		if (getenv("GC_DEBUG_TIMES")) fprintf(stderr, "[gc mem] the fragment pipeline runs again: trace pool %.2f G cells needed (now %.2f), anchor path pool %.2f G words needed (now %.2f)\n", traceNeed / 1e9, traceBudget / 1e9, pathNeed / 1e9, pathCapacity / 1e9);
		poolReruns++;
		st->poolsRerun = true;
		return true;
	}

	// ---------------- anchors, chains and stitched paths come down (pinned staging)
	void resultsBack()
	{
		// ---------------- results back (pinned staging)
		deviceAnchors = P->keep_traces != 1 && !(getenv("GC_HOST_ANCHORS") && atoi(getenv("GC_HOST_ANCHORS")) == 1);   // (GC_HOST_ANCHORS=1: test hook, the host's walk over the slots as before r5)
		anchors = st->hAnchors.reserve<AnchorRec>(deviceAnchors ? 1 : nSlots);
		fragStatus = st->hFragStatus.reserve<uint32_t>(deviceAnchors ? 1 : nFrags);
		fragExtended = st->hFragExtended.reserve<uint32_t>(deviceAnchors ? 1 : nFrags);
		readTies = st->hReadTies.reserve<uint32_t>(n);
		chainOut = st->hChainOut.reserve<uint32_t>(nSlots);
		chainLen = st->hChainLen.reserve<uint32_t>(n);
		chainScore = st->hChainScore.reserve<unsigned long long>(n);
		chainStatus = st->hChainStatus.reserve<uint32_t>(n);
		if (!deviceAnchors) {
			if (nSlots) HIP_CHECK(hipMemcpyAsync(anchors, dAnchors, nSlots * sizeof(AnchorRec), hipMemcpyDeviceToHost, stream));
			if (nFrags) HIP_CHECK(hipMemcpyAsync(fragStatus, dFragStatus, nFrags * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
			if (nFrags) HIP_CHECK(hipMemcpyAsync(fragExtended, dFragExtended, nFrags * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
		}
		if (n) HIP_CHECK(hipMemcpyAsync(readTies, dReadTies, n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
		if (nSlots) HIP_CHECK(hipMemcpyAsync(chainOut, dChainOut, nSlots * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
		if (n) HIP_CHECK(hipMemcpyAsync(chainLen, dChainLen, n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
		if (n) HIP_CHECK(hipMemcpyAsync(chainStatus, dChainStatus, n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
		if (n) HIP_CHECK(hipMemcpyAsync(chainScore, dChainScore, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
		HIP_CHECK(hipMemcpyAsync(hSmall, dCursors, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
		HIP_CHECK(hipMemcpyAsync(hSmall + 8, dCounters, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
		syncStream(stream);
		{
			auto pairUs = [&](int k) { float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, st->fragEv[2 * k], st->fragEv[2 * k + 1])); return (double)ms * 1000.0; };
			res->kernel_us[1] = res->kernel_us[2] = 0;
			for (uint32_t k = 0; k < nExtendPairs; k++) res->kernel_us[1] += pairUs((int)k);          // k_extend, all rounds (with their retry launches)
			for (uint32_t k = 0; k < nAnchorPairs; k++) res->kernel_us[2] += pairUs(4 + (int)k);      // k_build_anchors, all rounds
		}
		res->kernel_us[3] = elapsedUs(4, 5);
		for (int i = 0; i < 8; i++) res->counters[i] = hSmall[8 + i];
		if (getenv("GC_DEBUG_TIMES")) fprintf(stderr, "[gc frag] %llu of %llu fragment extensions handed to the plain-layout kernel\n", hSmall[8 + 6], hSmall[8 + 4]);
#ifdef GC_FRAG_STAMPS
		{
			unsigned long long cyc[8] = { 0 };
			HIP_CHECK(hipMemcpy(cyc, (unsigned long long*)st->fragClaims.ptr + 8, sizeof(cyc), hipMemcpyDeviceToHost));
			fprintf(stderr, "[gc frag stamps] wave-cycles: columns %llu fetch %llu tile_end %llu pop %llu finish %llu walk %llu\n", cyc[0], cyc[1], cyc[2], cyc[3], cyc[4], cyc[5]);
		}
#endif
		// (the cursors overshoot when a pool is full: the extensions / fragments that did not fit carry an overflow status and their reads are flagged)
		uint64_t traceUsed = std::min<uint64_t>(hSmall[1], traceBudget), pathUsed = std::min<uint64_t>(hSmall[2], pathCapacity);
		pathPool = st->hPathPool.reserve<uint32_t>(deviceAnchors ? 1 : pathUsed);
		if (pathUsed && !deviceAnchors) HIP_CHECK(hipMemcpyAsync(pathPool, dPathPool, pathUsed * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
		anchorTraces = P->keep_traces == 1;   // (keep_traces == 2: the alignments' traces only - what the output encoders read)
		if (anchorTraces) {
			extResults.resize(nWork);
			tracePool.resize(traceUsed);
			if (nWork) HIP_CHECK(hipMemcpyAsync(extResults.data(), dResults, (size_t)nWork * sizeof(ExtResult), hipMemcpyDeviceToHost, stream));
			if (traceUsed) HIP_CHECK(hipMemcpyAsync(tracePool.data(), dTrace, traceUsed * sizeof(PoolCell), hipMemcpyDeviceToHost, stream));
		}
		syncStream(stream);
		res->host_us[3] = nowUs() - tDev;   // K3..K4 + their transfers, wall
		// the stitched node paths come down behind the kernels that follow on this stream; they are only needed for the result arrays
		if (deviceStitch) {
			uint64_t used = std::min<uint64_t>(*hStitchCursor, stitchDenseCap);
			hStitchNodes = st->hStitchNodes.reserve<uint32_t>(std::max<uint64_t>(1, used));
			if (used) HIP_CHECK(hipMemcpyAsync(hStitchNodes, dStitchNodes, used * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
			stitchNodesPending = true;
		}

	}

	// ---------------- r5: the anchors the reference keeps of every read as the result's dense arrays, made on the device (gc_results.hip)
	void compactAnchors()
	{
		if (!deviceAnchors) return;
		uint4* dPerRead = st->anchorPerRead.reserve<uint4>(n);
		uint32_t* dSlotEnd = st->anchorSlotEnd.reserve<uint32_t>(n);
		unsigned long long* dOff = st->anchorOff.reserve<unsigned long long>(2 * n + 2);
		hAnchorPerRead = st->hAnchorPerRead.reserve<uint4>(n);
		hAnchorOff = st->hAnchorOff.reserve<unsigned long long>(2 * n + 4);
		unsigned long long* hTotals = hAnchorOff + 2 * n + 2;
		hTotals[0] = hTotals[1] = 0;
		launchAnchorCounts(stream, dJobs, (uint32_t)n, dFrags, dFragStatus, dFragExtended, dAnchors, dPerRead, dSlotEnd, dOff, hTotals);
		if (n) HIP_CHECK(hipMemcpyAsync(hAnchorPerRead, dPerRead, n * sizeof(uint4), hipMemcpyDeviceToHost, stream));
		if (n) HIP_CHECK(hipMemcpyAsync(hAnchorOff, dOff, (2 * n + 2) * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
		syncStream(stream);
		denseAnchors = n ? hAnchorOff[2 * n] : 0; densePathWords = n ? hAnchorOff[2 * n + 1] : 0;
		// one block for the eleven arrays: nine of 4 bytes per anchor, the 8-byte path offsets, the path words
		const uint64_t A = denseAnchors, bytes = A * 44 + densePathWords * 4 + 64;
		uint8_t* dDense = st->anchorDense.reserve<uint8_t>(bytes);
		hAnchorDense = st->hAnchorDense.reserve<uint8_t>(bytes);
		AnchorArrays out;
		out.pathOff = (unsigned long long*)dDense;
		uint32_t* w = (uint32_t*)(dDense + 8 * A);
		out.x = w; out.y = w + A; out.firstNode = w + 2 * A; out.firstOffset = w + 3 * A; out.firstSeqPos = w + 4 * A; out.lastNode = w + 5 * A; out.lastOffset = w + 6 * A; out.lastSeqPos = w + 7 * A;
		out.score = (int32_t*)(w + 8 * A); out.path = w + 9 * A;
		launchAnchorCompact(stream, dJobs, (uint32_t)n, dAnchors, dPathPool, dSlotEnd, dOff, out);
		if (A) HIP_CHECK(hipMemcpyAsync(hAnchorDense, dDense, A * 44 + densePathWords * 4, hipMemcpyDeviceToHost, stream));
		syncStream(stream);
	}
	// the dense arrays in the staging block (the layout compactAnchors gave the device's)
	const unsigned long long* densePathOff() const { return (const unsigned long long*)hAnchorDense; }
	const uint32_t* denseWords(int k) const { return (const uint32_t*)(hAnchorDense + 8 * denseAnchors) + (uint64_t)k * denseAnchors; }   // 0 x, 1 y, 2-4 first node / offset / seqPos, 5-7 last, 8 score, 9 the path words

	// ---------------- host stitching of what the kernel declined; NW distance of every stitched path against its read
	void stitchAndChainDistances()
	{
		// ---------------- chain stitching (src/Aligner.cpp:754-822) on the host workers, while the whole-read pass still runs
		double tStitch = nowUs();
		std::atomic<uint64_t> hostStitched { 0 };
		if (P->stitch) {
			pool.run(n, [&](size_t r, size_t) {
				ReadGlue& gl = glue[r];
				gl.stitchedOnDevice = false;
				if (chainStatus[r] != 0 || chainLen[r] == 0) return;
				if (deviceStitch && stitchInfo[r].status == 0) {
					const StitchInfo& si = stitchInfo[r];
					gl.stitched.nodes.clear();   // filled once the download has finished (below)
					gl.stitched.firstOffset = si.firstOffset; gl.stitched.lastOffset = si.lastOffset; gl.stitched.cells = si.cells;
					gl.stitchedOnDevice = true;
					return;
				}
				hostStitched++;
				if (deviceStitch && getenv("GC_DEBUG_TIMES")) fprintf(stderr, "[gc stitch] read %zu goes to the host: reason %u, chain of %u anchors\n", r, stitchInfo[r].status, chainLen[r]);
				std::vector<uint32_t> slots;
				if (deviceAnchors) {   // the read's kept anchors are the dense arrays' [a0, a1): chain index = dense index
					const uint64_t a0 = hAnchorOff[2 * r], a1 = hAnchorOff[2 * r + 2];
					std::vector<AnchorRec> recs(a1 - a0);
					slots.resize(a1 - a0);
					for (uint64_t a = a0; a < a1; a++) {
						AnchorRec& rec = recs[a - a0];
						rec.valid = 1; rec.x = denseWords(0)[a]; rec.y = denseWords(1)[a];
						rec.firstNode = denseWords(2)[a]; rec.firstOffset = denseWords(3)[a]; rec.firstSeqPos = denseWords(4)[a] - rec.x;
						rec.lastNode = denseWords(5)[a]; rec.lastOffset = denseWords(6)[a]; rec.lastSeqPos = denseWords(7)[a] - rec.x;
						rec.score = (int32_t)denseWords(8)[a];
						rec.pathOff = densePathOff()[a];
						rec.pathLen = (uint32_t)((a + 1 < denseAnchors ? densePathOff()[a + 1] : densePathWords) - densePathOff()[a]);
						rec.pad = 0;
						slots[a - a0] = (uint32_t)(a - a0);
					}
					stitchChain(hg, (long long)P->colinear_gap, chainOut + jobs[r].chainBegin, chainLen[r], slots.data(), recs.data(), denseWords(9), gl.stitched);
					return;
				}
				uint64_t slot = gl.slotBegin;
				for (size_t f = 0; f < gl.nWindows; f++) {
					uint64_t F = gl.fragBegin + f;
					uint32_t nS = frags[F].seedEnd - frags[F].seedBegin;
					if (fragStatus[F] == 1) break;   // `cont` is never reset (src/Aligner.cpp:695-703)
					for (uint32_t k = 0; k < nS; k++) if (anchors[slot + k].valid) slots.push_back((uint32_t)(slot + k - gl.slotBegin));
					slot += nS;
				}
				stitchChain(hg, (long long)P->colinear_gap, chainOut + jobs[r].chainBegin, chainLen[r], slots.data(), anchors + gl.slotBegin, pathPool, gl.stitched);
			});
		}
		// ---------------- edit distance of the stitched path against the read (edlibAlign at src/Aligner.cpp:845, value only):
		// path letters are spelled out on the device from the node path, then the NW kernel; still behind the whole-read pass
		std::function<void()> finishChainEditDistances;
		std::vector<uint32_t> pairRead;   // chain pairs -> read
		if (P->stitch && P->edit_distances) {
			uint64_t nNodesTotal = 0, nCells = 0;
			uint32_t nPairs = 0;
			// node paths stitched on the device are read where k_stitch left them; only the ones stitched here go up
			for (uint64_t r = 0; r < n; r++) { glue[r].stitchedBegin = nNodesTotal; if (!glue[r].stitchedOnDevice) nNodesTotal += glue[r].stitched.nodes.size(); }
			uint32_t* hNodes = st->hEdPathNodes.reserve<uint32_t>(nNodesTotal);
			PathSeqJob* hJobsPS = st->hEdJobs.reserve<PathSeqJob>(n);
			EdPair* hPairs = st->hEdPairs.reserve<EdPair>(n);
			int64_t* hOut = st->hEdOut.reserve<int64_t>(n);
			for (uint64_t r = 0; r < n; r++) {
				const StitchedPath& sp = glue[r].stitched;
				const bool onDevice = glue[r].stitchedOnDevice;
				if (!onDevice && !sp.nodes.empty()) memcpy(hNodes + glue[r].stitchedBegin, sp.nodes.data(), sp.nodes.size() * sizeof(uint32_t));
				if (sp.cells >= 0x7fffffffull) throw std::runtime_error("stitched path too long");
				hJobsPS[r] = PathSeqJob { onDevice ? stitchInfo[r].start : ((1ull << 63) | glue[r].stitchedBegin), nCells, onDevice ? stitchInfo[r].len : (uint32_t)sp.nodes.size(), (uint32_t)sp.cells, sp.firstOffset, sp.lastOffset };
				if (sp.cells) {
					uint32_t len = (uint32_t)(R->offsets[r + 1] - R->offsets[r]);
					// first band: the length difference plus ~14 % of the shorter sequence (ONT-like error rates pass in one sweep) - widened to 20 %
					// where that still fits the two-pairs-per-wave kernel: a wave's time follows the number of columns, not the band, so the wider
					// band is free there and spares the pairs above 14 % their second sweep
					uint32_t cells = (uint32_t)sp.cells, shorter = std::min(cells, len), longer = std::max(cells, len);
					uint32_t firstBand = (longer - shorter) + std::max<uint32_t>(64, shorter / 7);
					const uint32_t halfLimit = editDistanceMaxK(0);
					if (firstBand < halfLimit) firstBand = std::max(firstBand, std::min<uint32_t>(halfLimit - 1, (longer - shorter) + shorter / 5));
					hPairs[nPairs++] = EdPair { nCells, cells, (uint32_t)r, (uint32_t)r, firstBand };
					pairRead.push_back((uint32_t)r);
				}
				nCells += sp.cells;
			}
			uint32_t* dNodes = st->edPathNodes.reserve<uint32_t>(nNodesTotal);
			PathSeqJob* dJobsPS = st->edJobs.reserve<PathSeqJob>(n);
			char* dLetters = st->edLetters.reserve<char>(nCells);
			uint32_t* dLettersLen = st->edLettersLen.reserve<uint32_t>(n);
			EdPair* dPairs = st->edPairs.reserve<EdPair>(n);
			int64_t* dOut = st->edOut.reserve<int64_t>(n);
			if (nNodesTotal) HIP_CHECK(hipMemcpyAsync(dNodes, hNodes, nNodesTotal * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
			if (n) HIP_CHECK(hipMemcpyAsync(dJobsPS, hJobsPS, n * sizeof(PathSeqJob), hipMemcpyHostToDevice, stream));
			launchChainPathSeq(stream, G->dev, dJobsPS, (uint32_t)n, dStitchNodes, dNodes, dLetters, dLettersLen);
			chainLetterJobs = hJobsPS;
			dChainLetters = dLetters;
			auto readLenOf = [R = R](uint32_t r) { return (uint32_t)(R->offsets[r + 1] - R->offsets[r]); };
			launchEditDistances(st->edChainRun, stream, hPairs, hOut, nPairs, dPairs, dOut, R->devEdReads, R->devBases, R->devEqMasks, dLetters, dLettersLen, readLenOf);
			finishChainEditDistances = [=, &pairRead]() {   // waits for the kernels (they run beside the whole-read pass) and reruns the few pairs that need a wider band
				finishEditDistances(st->edChainRun, stream, hPairs, hOut, nPairs, dPairs, dOut, R->devEdReads, R->devBases, R->devEqMasks, dLetters, dLettersLen);
				if (getenv("GC_DEBUG_ED")) for (uint32_t i = 0; i < nPairs && i < 400; i++) {
					const uint32_t r = pairRead[i];
					fprintf(stderr, "[gc ed] read %u len %llu path %llu chain %u scoreSum %u onDevice %d distance %lld\n", r, (unsigned long long)(R->offsets[r + 1] - R->offsets[r]), (unsigned long long)glue[r].stitched.cells,
						chainLen[r], deviceStitch ? stitchInfo[r].scoreSum : 0u, (int)glue[r].stitchedOnDevice, (long long)hOut[i]);
				}
				for (uint32_t i = 0; i < nPairs; i++) {
					ReadGlue& gl = glue[pairRead[i]];
					gl.chainEditDistance = hOut[i];
					if (hOut[i] < -1) { gl.chainEditDistance = -1; gl.capacityExceeded = true; }
				}
			};
		}
		if (finishChainEditDistances) finishChainEditDistances();   // this thread would only wait for the whole-read pass otherwise
		if (stitchNodesPending) {
			syncStream(stream);
			pool.run(n, [&](size_t r, size_t) {
				if (!glue[r].stitchedOnDevice) return;
				const StitchInfo& si = stitchInfo[r];
				glue[r].stitched.nodes.assign(hStitchNodes + si.start, hStitchNodes + si.start + si.len);
			});
		}
		double stitchUs = nowUs() - tStitch;
		if (getenv("GC_DEBUG_TIMES")) fprintf(stderr, "[gc cpu] %.0f ms up to the end of stitching + chain edit distances\n", processCpuMs() - cpuCall);
		if (getenv("GC_DEBUG_TIMES")) fprintf(stderr, "[gc times] chain stitching + its edit distances %.1f ms (%llu reads stitched on the host)\n", stitchUs / 1e3, (unsigned long long)hostStitched.load());
		res->counters[7] = hostStitched.load();   // reads whose chain was stitched on the host

	}

	// ---------------- the pass thread ends (its after-pass stage included unless it ran on this thread)
	void joinWholeReadPass()
	{
		// ---------------- whole-read pass results
		tJoined = nowUs();
		if (P->long_pass) {
			double tJoin0 = nowUs();
			for (auto& t : longThreads) t.join();
			tJoined = nowUs();
			cpuJoined = processCpuMs();
			if (getenv("GC_DEBUG_TIMES")) fprintf(stderr, "[gc times] main thread waited %.1f ms for the whole-read pass\n", (tJoined - tJoin0) / 1e3);
			for (auto& e : longErrors) if (e) std::rethrow_exception(e);
			finishLongGroups();
			res->kernel_us[5] = longWallEndUs.load() - (longWallBeginUs.load() > 0.0 ? longWallBeginUs.load() : tLongWall0);   // whole-read pass, wall clock from the first group's start to the last group's end
			if (!longPostInThread) afterLongPass();
			for (uint64_t r = 0; r < n; r++) if (glue[r].capacityExceededLong) glue[r].capacityExceeded = true;
			if (P->keep_traces) {
				const uint64_t cellsUsed = std::min<uint64_t>(hLongSmall[0], cellBudget);   // (the cursor counts refused requests too: a full pool leaves it beyond the pool's end)
				LongCell* staged = st->hLongCells.reserve<LongCell>(cellsUsed);
				if (cellsUsed) HIP_CHECK(hipMemcpyAsync(staged, dLongCells, cellsUsed * sizeof(LongCell), hipMemcpyDeviceToHost, st->longStream));
				syncStream(st->longStream);
				longCells = staged;
			}
		}

		if (getenv("GC_DEBUG_TIMES")) fprintf(stderr, "[gc times] after the whole-read pass: selection + its edit distances %.1f ms\n", (nowUs() - tJoined) / 1e3);

	}

	// ---------------- edlib path + trace of the chained alignments that are wanted (src/Aligner.cpp:845-905)
	void chainedAlignments()
	{
		// ---------------- the chained alignment (src/Aligner.cpp:845-905): edlib's alignment path of (stitched path, read) from k_edit_path,
		// walked over the path cells and the read into the trace; then the decision. Only reads whose chained alignment can still win
		// (or all with chain_traces == 2) are traced: the path does not change the edit distance that decides.
		double tChainTrace = nowUs();
		uint64_t nChainTraced = 0;
		if (P->stitch && P->edit_distances && chainLetterJobs) {
			auto beats = [&](const ReadGlue& gl) { return gl.longSelected.empty() || gl.longEditDistance > gl.chainEditDistance; };   // :905
			std::vector<uint32_t> cand;
			for (uint64_t r = 0; r < n; r++) {
				const ReadGlue& gl = glue[r];
				if (gl.longFailed || gl.stitched.cells == 0 || gl.chainEditDistance < 0) continue;
				if (P->chain_traces >= 2 || (P->chain_traces == 1 && beats(gl))) cand.push_back((uint32_t)r);
			}
			nChainTraced = cand.size();
			if (!cand.empty()) {
				const size_t m = cand.size();
				EdPathJob* hJobsEP = st->hEdPathJobs.reserve<EdPathJob>(m);
				uint64_t opsTotal = 0;
				uint32_t maxQ = 1, maxT = 1;
				for (size_t i = 0; i < m; i++) {
					const uint32_t r = cand[i];
					const uint32_t q = (uint32_t)glue[r].stitched.cells, t = (uint32_t)(R->offsets[r + 1] - R->offsets[r]);
					hJobsEP[i] = EdPathJob { chainLetterJobs[r].outOff, R->offsets[r], opsTotal, q, t, (int32_t)glue[r].chainEditDistance, 0 };
					opsTotal += (uint64_t)q + t;
					maxQ = std::max(maxQ, q); maxT = std::max(maxT, t);
				}
				EdPathJob* dJobsEP = st->edPathJobs.reserve<EdPathJob>(m);
				uint8_t* dOps = st->edPathOps.reserve<uint8_t>(opsTotal);
				uint32_t* dOpsLen = st->edPathLen.reserve<uint32_t>(m);
				uint8_t* dScratchEP = st->edPathScratch.reserve<uint8_t>((uint64_t)editPathGridBlocks((uint32_t)m) * editPathScratchBytes(maxQ, maxT));
				uint8_t* hOps = st->hEdPathOps.reserve<uint8_t>(opsTotal);
				uint32_t* hOpsLen = st->hEdPathLen.reserve<uint32_t>(m);
				HIP_CHECK(hipMemcpyAsync(dJobsEP, hJobsEP, m * sizeof(EdPathJob), hipMemcpyHostToDevice, stream));
				launchEditPath(stream, dJobsEP, (uint32_t)m, dChainLetters, R->devBases, dScratchEP, maxQ, maxT, dOps, dOpsLen);
				HIP_CHECK(hipMemcpyAsync(hOpsLen, dOpsLen, m * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
				HIP_CHECK(hipMemcpyAsync(hOps, dOps, opsTotal, hipMemcpyDeviceToHost, stream));
				syncStream(stream);
				pool.run(m, [&](size_t i, size_t) {
					const uint32_t r = cand[i];
					ReadGlue& gl = glue[r];
					const uint64_t readLen = R->offsets[r + 1] - R->offsets[r];
					// `longest`: one (node, offset) per base of the stitched piece (pathToTrace, src/Aligner.cpp:409-424)
					std::vector<std::pair<uint32_t, uint32_t>> cells;
					cells.reserve(gl.stitched.cells);
					for (uint32_t node : gl.stitched.nodes) {
						uint32_t S = 0, L = (uint32_t)hg.NodeLength(node);
						if (node == gl.stitched.nodes[0]) S = gl.stitched.firstOffset;
						else if (node == gl.stitched.nodes.back()) L = gl.stitched.lastOffset + 1;
						for (uint32_t o = S; o < L; o++) cells.emplace_back(node, o);
					}
					const uint8_t* ops = hOps + hJobsEP[i].opsOff;
					const uint32_t nOps = hOpsLen[i];
					if (cells.empty() || nOps == 0) return;   // no alignment from edlib: no chained alignment item (:890)
					// :848-876: one trace cell per op, recorded before the op advances; indices clamped to the last cell / base
					uint64_t pos_i = 0, seq_i = 0;
					gl.chainTraceNode.resize(nOps); gl.chainTraceOffset.resize(nOps); gl.chainTraceSeqPos.resize(nOps); gl.chainTraceSwitch.assign(nOps, 0);
					uint32_t prevSplit = 0;
					for (uint32_t j = 0; j < nOps; j++) {
						const uint32_t node = cells[pos_i].first, off = cells[pos_i].second;
						gl.chainTraceNode[j] = hg.nodeIDs[node];                       // :886-887 output coordinates
						gl.chainTraceOffset[j] = (uint32_t)(off + hg.nodeOffset[node]);
						gl.chainTraceSeqPos[j] = (uint32_t)seq_i;
						if (j > 0 && node != prevSplit) gl.chainTraceSwitch[j - 1] = 1;      // :880-882 (split nodes compared)
						prevSplit = node;
						const uint8_t c = ops[j];
						if (c == 0 || c == 3) { pos_i++; seq_i++; }
						else if (c == 1) pos_i++;
						else if (c == 2) seq_i++;
						seq_i = std::min<uint64_t>(seq_i, readLen - 1);
						pos_i = std::min<uint64_t>(pos_i, cells.size() - 1);
					}
					gl.chainAlnStart = gl.chainTraceSeqPos[0];
					gl.chainAlnEnd = gl.chainTraceSeqPos[nOps - 1] + 1;
					// :904 SelectAlignments(method All) still applies --E-cutoff; :905 the decision
					gl.hasChainAlignment = evalueModel.keeps(P->e_cutoff, hg.SizeInBP(), readLen, gl.chainAlnEnd - gl.chainAlnStart, (size_t)gl.chainEditDistance);
					if (!gl.hasChainAlignment) { gl.chainTraceNode.clear(); gl.chainTraceOffset.clear(); gl.chainTraceSeqPos.clear(); gl.chainTraceSwitch.clear(); gl.chainAlnStart = gl.chainAlnEnd = 0; }
					gl.chainWins = gl.hasChainAlignment && beats(gl);
				});
			}
			if (P->chain_traces == 0) {
				// no traces asked for: the decision from the distances alone (an alignment edlib cannot build or --E-cutoff drops would differ)
				for (uint64_t r = 0; r < n; r++) { ReadGlue& gl = glue[r]; gl.chainWins = !gl.longFailed && gl.stitched.cells > 0 && gl.chainEditDistance >= 0 && beats(gl); }
			}
		}
		if (getenv("GC_DEBUG_TIMES")) fprintf(stderr, "[gc times] chained alignment traces: %llu reads, %.1f ms\n", (unsigned long long)nChainTraced, (nowUs() - tChainTrace) / 1e3);

	}

	// ---------------- the final alignments encoded where their traces are (params->device_output; gc_output.hip): the GAF path / CIGAR text and the vg::Path bytes of
	// every alignment the reference would write for the batch (src/Aligner.cpp:901-920,1003-1023), so that no trace cell has to come down for the writers
	struct OutEntry { uint32_t read, aln; uint8_t source; };   // aln: index into the read's longAlns (source 0); source 1: the read's chained alignment, encoded by the host from its trace
	std::vector<OutEntry> outEntries;
	std::vector<uint64_t> readOutOff;
	const OutRec* hOutRecs = nullptr; const uint64_t* hOutOffsets = nullptr;
	const char* hOutPath = nullptr; const char* hOutCigar = nullptr; const uint8_t* hOutVg = nullptr;
	uint64_t nOutJobs = 0;
	std::vector<uint64_t> outJobOfEntry;
	// Started by the whole-read pass's own thread as soon as the selection is known (the end of afterLongPass): the main thread is still in the fragment pipeline, the stitching and the
	// chain distances then, so the two passes of the encoder, their two waits and the download of the text overlap that work instead of following it (r4: with the encoder after the
	// join, every batch's latency grew by the encoder's launches waiting for wave slots among the other batches' kernels, and five streams in flight completed a batch every 255-268 ms
	// end to end against 160 for the hot path). Every read's selected alignments are encoded; a read whose chained alignment wins in the end (known only after the join) drops its pieces
	// in encodeOutput - wasted work in proportion to the winners.
	std::vector<uint64_t> readJobBegin;   // first job of every read (jobs of a read: its selected alignments sorted by alignmentStart); [n] = number of jobs
	std::vector<uint32_t> jobAln;         // the alignment (index into the read's longAlns) of every job
	void encodeOutputStart()
	{
		if (!P->device_output) return;
		const double t0 = nowUs();
		readJobBegin.assign(n + 1, 0);
		jobAln.clear();
		for (uint64_t r = 0; r < n; r++) {
			const ReadGlue& gl = glue[r];
			readJobBegin[r] = jobAln.size();
			if (gl.longFailed) continue;
			struct Item { uint32_t start; uint32_t aln; };
			std::vector<Item> items;
			for (uint32_t k : gl.longSelected) items.push_back(Item { gl.longAlns[k].start, k });
			auto byStart = [](const Item& l, const Item& rr) { return l.start < rr.start; };
			std::sort(items.begin(), items.end(), byStart);   // src/Aligner.cpp:1003
			std::sort(items.begin(), items.end(), byStart);   // :1023 (an unstable sort may move ties even in a sorted list)
			for (const Item& it : items) jobAln.push_back(it.aln);
		}
		readJobBegin[n] = jobAln.size();
		nOutJobs = jobAln.size();
		if (nOutJobs == 0) return;
		if (nOutJobs >= 0xffffffffull) throw std::runtime_error("too many alignments in one batch for the output encoder");
		OutJob* hJobsOut = st->hOutJobs.reserve<OutJob>(nOutJobs);
		const uint32_t flags = ((P->device_output & 2) ? 1u : 0u) | ((P->device_output & 3) ? 2u : 0u) | ((P->device_output & 4) ? 4u : 0u);
		for (uint64_t r = 0; r < n; r++)
			for (uint64_t k = readJobBegin[r]; k < readJobBegin[r + 1]; k++) {
				const LongAln& al = glue[r].longAlns[jobAln[k]];
				hJobsOut[k] = OutJob { al.traceOff, R->offsets[r], al.traceLen, (uint32_t)(R->offsets[r + 1] - R->offsets[r]), flags, 0 };
			}
		hipStream_t q = st->longStream;   // (the pass and its decision are done with it)
		OutJob* dJobsOut = st->outJobs.reserve<OutJob>(nOutJobs);
		OutRec* dRecs = st->outRecs.reserve<OutRec>(nOutJobs);
		uint64_t* dOffsets = st->outOffsets.reserve<uint64_t>(3 * (nOutJobs + 1));
		unsigned long long* dTotals = st->outTotals.reserve<unsigned long long>(4);
		uint32_t* dMapSizes = (P->device_output & 4) ? st->outMapSizes.reserve<uint32_t>(std::max<uint64_t>(1, hLongSmall[0]))   /* one word per cell of the pool in use */ : nullptr;
		unsigned long long* hTotals = st->hOutTotals.reserve<unsigned long long>(4);
		HIP_CHECK(hipMemcpyAsync(dJobsOut, hJobsOut, nOutJobs * sizeof(OutJob), hipMemcpyHostToDevice, q));
		launchOutCount(q, G->dev, G->devNames, G->devIupac, dJobsOut, (uint32_t)nOutJobs, dLongCells, R->devBases, dRecs, dOffsets, dMapSizes, dTotals);
		HIP_CHECK(hipMemcpyAsync(hTotals, dTotals, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, q));
		syncStream(q);
		const uint64_t pathBytes = hTotals[0], cigarBytes = hTotals[1], vgBytes = hTotals[2];
		char* dPath = st->outPathText.reserve<char>(pathBytes + 1);
		char* dCigar = st->outCigarText.reserve<char>(cigarBytes + 1);
		uint8_t* dVg = st->outVgBytes.reserve<uint8_t>(vgBytes + 1);
		launchOutWrite(q, G->dev, G->devNames, G->devIupac, dJobsOut, (uint32_t)nOutJobs, dLongCells, R->devBases, dRecs, dOffsets, dMapSizes, dPath, dCigar, dVg);
		OutRec* recs = st->hOutRecs.reserve<OutRec>(nOutJobs);
		uint64_t* offs = st->hOutOffsets.reserve<uint64_t>(3 * (nOutJobs + 1));
		char* pathText = st->hOutPathText.reserve<char>(pathBytes + 1);
		char* cigarText = st->hOutCigarText.reserve<char>(cigarBytes + 1);
		uint8_t* vg = st->hOutVgBytes.reserve<uint8_t>(vgBytes + 1);
		HIP_CHECK(hipMemcpyAsync(recs, dRecs, nOutJobs * sizeof(OutRec), hipMemcpyDeviceToHost, q));
		HIP_CHECK(hipMemcpyAsync(offs, dOffsets, 3 * (nOutJobs + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, q));
		if (pathBytes) HIP_CHECK(hipMemcpyAsync(pathText, dPath, pathBytes, hipMemcpyDeviceToHost, q));
		if (cigarBytes) HIP_CHECK(hipMemcpyAsync(cigarText, dCigar, cigarBytes, hipMemcpyDeviceToHost, q));
		if (vgBytes) HIP_CHECK(hipMemcpyAsync(vg, dVg, vgBytes, hipMemcpyDeviceToHost, q));
		syncStream(q);
		for (uint64_t k = 0; k < nOutJobs; k++) if (recs[k].steps == 0xffffffffu) throw std::runtime_error("internal: the output encoder's two passes disagree on an alignment's size");
		hOutRecs = recs; hOutOffsets = offs; hOutPath = pathText; hOutCigar = cigarText; hOutVg = vg;
		if (getenv("GC_DEBUG_TIMES")) fprintf(stderr, "[gc times] output encoding on the device: %llu alignments, %.1f MB of path text, %.1f MB of CIGAR, %.1f MB of vg::Path bytes, %.1f ms (on the whole-read pass's thread, beside the fragment pipeline)\n",
			(unsigned long long)nOutJobs, pathBytes / 1e6, cigarBytes / 1e6, vgBytes / 1e6, (nowUs() - t0) / 1e3);
	}

	// After the decision: which of the encoded alignments are the batch's output, in the reference's order
	void encodeOutput()
	{
		if (!P->device_output) return;
		readOutOff.assign(n + 1, 0);
		outEntries.clear();
		outJobOfEntry.clear();
		for (uint64_t r = 0; r < n; r++) {
			const ReadGlue& gl = glue[r];
			readOutOff[r] = outEntries.size();
			if (gl.longFailed) continue;
			if (gl.chainWins) { outEntries.push_back(OutEntry { (uint32_t)r, 0, 1 }); outJobOfEntry.push_back(~0ull); continue; }
			for (uint64_t k = readJobBegin[r]; k < readJobBegin[r + 1]; k++) { outEntries.push_back(OutEntry { (uint32_t)r, jobAln[k], 0 }); outJobOfEntry.push_back(k); }
		}
		readOutOff[n] = outEntries.size();
	}

	void assembleOutput()   // the pieces into the result (entries of chained winners stay empty: source 1)
	{
		res->device_output = P->device_output;
		if (!P->device_output) return;
		const uint64_t nOut = outEntries.size();
		res->read_out_off = resultArray<uint64_t>(n + 1);
		memcpy(res->read_out_off, readOutOff.data(), (n + 1) * sizeof(uint64_t));
		res->out_source = resultArray<uint8_t>(nOut);
		res->out_numbers = resultArray<uint64_t>(12 * nOut);
		res->out_path_off = resultArray<uint64_t>(nOut + 1); res->out_cigar_off = resultArray<uint64_t>(nOut + 1); res->out_vg_off = resultArray<uint64_t>(nOut + 1);
		const uint64_t stride = nOutJobs + 1;
		// where every entry's pieces go: the kept jobs' bytes back to back (all of them, in order, unless chained alignments won)
		uint64_t pathBytes = 0, cigarBytes = 0, vgBytes = 0;
		for (uint64_t e = 0; e < nOut; e++) {
			res->out_path_off[e] = pathBytes; res->out_cigar_off[e] = cigarBytes; res->out_vg_off[e] = vgBytes;
			const uint64_t k = outJobOfEntry[e];
			if (k == ~0ull) continue;
			pathBytes += hOutOffsets[k + 1] - hOutOffsets[k]; cigarBytes += hOutOffsets[stride + k + 1] - hOutOffsets[stride + k]; vgBytes += hOutOffsets[2 * stride + k + 1] - hOutOffsets[2 * stride + k];
		}
		res->out_path_off[nOut] = pathBytes; res->out_cigar_off[nOut] = cigarBytes; res->out_vg_off[nOut] = vgBytes;
		res->out_path_text = resultArray<char>(pathBytes + 1); res->out_cigar_text = resultArray<char>(cigarBytes + 1); res->out_vg_path = resultArray<uint8_t>(vgBytes + 1);
		bool allKept = nOut == nOutJobs;   // no chained winner: the device's blobs are the result's, job k is entry k
		for (uint64_t e = 0; e < nOut && allKept; e++) allKept = outJobOfEntry[e] == e;   // (a winner's one entry can stand where a read's one dropped job was: the counts alone do not tell)
		const size_t parts = 16;
		if (allKept) pool.run(3 * parts, [&](size_t i, size_t) {
			const size_t which = i / parts, part = i % parts;
			const uint64_t total = which == 0 ? pathBytes : which == 1 ? cigarBytes : vgBytes;
			const char* src = which == 0 ? hOutPath : which == 1 ? hOutCigar : (const char*)hOutVg;
			char* dst = which == 0 ? res->out_path_text : which == 1 ? res->out_cigar_text : (char*)res->out_vg_path;
			const uint64_t b = total * part / parts, e = total * (part + 1) / parts;
			if (e > b) memcpy(dst + b, src + b, e - b);
		});
		else pool.run(nOut, [&](size_t e, size_t) {
			const uint64_t k = outJobOfEntry[e];
			if (k == ~0ull) return;
			memcpy(res->out_path_text + res->out_path_off[e], hOutPath + hOutOffsets[k], hOutOffsets[k + 1] - hOutOffsets[k]);
			memcpy(res->out_cigar_text + res->out_cigar_off[e], hOutCigar + hOutOffsets[stride + k], hOutOffsets[stride + k + 1] - hOutOffsets[stride + k]);
			memcpy(res->out_vg_path + res->out_vg_off[e], hOutVg + hOutOffsets[2 * stride + k], hOutOffsets[2 * stride + k + 1] - hOutOffsets[2 * stride + k]);
		});
		res->out_path_text[pathBytes] = 0; res->out_cigar_text[cigarBytes] = 0;
		for (uint64_t e = 0; e < nOut; e++) {
			const OutEntry& en = outEntries[e];
			res->out_source[e] = en.source;
			uint64_t* num = res->out_numbers + 12 * e;
			if (en.source == 0) {
				const OutRec& rec = hOutRecs[outJobOfEntry[e]];
				const LongAln& al = glue[en.read].longAlns[en.aln];
				num[0] = rec.nodePathLen; num[1] = rec.nodePathStart; num[2] = rec.nodePathEnd; num[3] = rec.matches; num[4] = rec.mismatches; num[5] = rec.insertions; num[6] = rec.deletions;
				num[7] = al.traceLen; num[8] = al.start; num[9] = al.end; num[10] = rec.steps; num[11] = al.score;
			} else for (int i = 0; i < 12; i++) num[i] = 0;
		}
	}

	// ---------------- the flat result: count per read, prefix-sum, fill in parallel
	void assemble()
	{
		// ---------------- assemble the flat result: count per read, prefix-sum, fill in parallel
		double tAsm = nowUs();
		if (glueCopied) HIP_CHECK(hipEventSynchronize(glueCopied));   // (long since done: the copies were queued before the fragment pipeline)
		std::vector<uint8_t> failedAssertion(n, 0);
		std::vector<uint64_t> seedsExtended(n, 0), seedsExtendedLong(n, 0);
		// read position of the seed in a fragment-pass slot (the device holds the per-slot records; the host keeps seeds and windows)
		auto slotSeqPos = [&](uint64_t r, uint64_t slot, uint64_t F) -> uint32_t {
			(void)r;
			return readSeeds[fragFirstSeed[F] + (slot - frags[F].seedBegin)].seqPos;   // (host copies: kept whenever keep_traces asks for this)
		};
		auto forEachAnchor = [&](uint64_t r, auto&& visit) {   // visit(slotIndex, fragmentIndex) for every anchor the reference would keep
			const ReadGlue& gl = glue[r];
			if (gl.longFailed) return;   // `cont` was already set by the whole-read pass (src/Aligner.cpp:529,591,702)
			uint64_t slot = gl.slotBegin;
			for (size_t f = 0; f < gl.nWindows; f++) {
				uint64_t F = gl.fragBegin + f;
				uint32_t nS = frags[F].seedEnd - frags[F].seedBegin;
				if (fragStatus[F] == 1) return;   // `cont` is never reset: later fragments add nothing (src/Aligner.cpp:695-703)
				for (uint32_t k = 0; k < nS; k++) if (anchors[slot + k].valid) visit(slot + k, F);
				slot += nS;
			}
		};
		pool.run(n, [&](size_t r, size_t) {
			ReadGlue& gl = glue[r];
			failedAssertion[r] = gl.failed ? 1 : 0;
			if (gl.longFailed) {   // the fragment pipeline ran beside the whole-read pass; what it found for this read is dropped
				failedAssertion[r] = 1;
				chainLen[r] = 0; chainScore[r] = 0; chainStatus[r] = 0;
				gl.stitched = StitchedPath();
				gl.chainEditDistance = -1;
				if (P->long_pass) seedsExtendedLong[r] = hLongResults[r].seedsExtended;
				return;
			}
			if (chainStatus[r] != 0) { gl.capacityExceeded = true; chainLen[r] = 0; chainScore[r] = 0; }
			if (deviceAnchors) {   // k_anchor_counts walked the read's fragments and slots
				const uint4 pr = hAnchorPerRead[r];
				if (pr.w & 2u) gl.capacityExceeded = true;
				if (pr.w & 1u) failedAssertion[r] = 1;
				seedsExtended[r] += pr.z;
				gl.nAnchors = pr.x; gl.nPath = pr.y;
			} else
			for (size_t f = 0; f < gl.nWindows; f++) {
				uint64_t F = gl.fragBegin + f;
				if (fragStatus[F] == 2) gl.capacityExceeded = true;   // an extension or the anchor path pool overflowed even in the retry: this fragment gave no anchors
				if (fragStatus[F] == 1) { failedAssertion[r] = 1; break; }
				seedsExtended[r] += fragExtended[F];
			}
			if (P->long_pass) {
				seedsExtendedLong[r] = hLongResults[r].seedsExtended;
				if (hLongResults[r].status == 1) failedAssertion[r] = 1;
			}
			if (!deviceAnchors) forEachAnchor(r, [&](uint64_t slot, uint64_t F) {
				gl.nAnchors++;
				gl.nPath += anchors[slot].pathLen;
				if (anchorTraces) {
					const ExtResult& eb = extResults[2 * slot];
					const ExtResult& ef = extResults[2 * slot + 1];
					uint32_t p = slotSeqPos(r, slot, F) - (anchors[slot].x);
					bool hasB = p > 0 && eb.status == EXT_OK, hasF = p < (uint32_t)P->split_len - 1 && ef.status == EXT_OK;
					gl.nTrace += (hasB ? (hasF ? eb.traceLen - 1 : eb.traceLen) : 0) + (hasF ? ef.traceLen : 0);
				}
			});
		});
		uint64_t nAnchors = 0, nPath = 0, nTrace = 0, nChain = 0, nLong = 0, nLongTrace = 0, nStitched = 0, nLongSelected = 0, nChainTrace = 0;
		for (uint64_t r = 0; r < n; r++) {
			glue[r].anchorBegin = nAnchors; glue[r].pathBegin = nPath; glue[r].traceBegin = nTrace; glue[r].chainBegin = nChain;
			glue[r].longBegin = nLong; glue[r].longTraceBegin = nLongTrace;
			nAnchors += glue[r].nAnchors; nPath += glue[r].nPath; nTrace += glue[r].nTrace; nChain += chainLen[r];
			nLong += glue[r].longAlns.size();
			glue[r].stitchedBegin = nStitched; nStitched += glue[r].stitched.nodes.size();
			glue[r].longSelectedBegin = nLongSelected; nLongSelected += glue[r].longSelected.size();
			glue[r].chainTraceBegin = nChainTrace; nChainTrace += glue[r].chainTraceNode.size();
			if (P->keep_traces) for (const LongAln& a : glue[r].longAlns) nLongTrace += a.traceLen;
		}
		const bool keepSeeds = P->keep_seeds != 0;
		std::vector<uint64_t> seedOutBegin(n + 1, 0);   // the result's seed lists are dense (the device's sit at capacity offsets)
		for (uint64_t r = 0; r < n; r++) seedOutBegin[r + 1] = seedOutBegin[r] + (keepSeeds ? glue[r].nSeedsR : 0);
		const uint64_t nSeedsOut = seedOutBegin[n];
		res->read_seed_off = resultArray<uint64_t>(n + 1);
		res->seed_node = resultArray<uint32_t>(nSeedsOut); res->seed_offset = resultArray<uint32_t>(nSeedsOut);
		res->seed_seqpos = resultArray<uint32_t>(nSeedsOut); res->seed_goodness = resultArray<uint64_t>(nSeedsOut);
		res->read_anchor_off = resultArray<uint64_t>(n + 1);
		res->anchor_x = resultArray<uint32_t>(nAnchors); res->anchor_y = resultArray<uint32_t>(nAnchors);
		res->anchor_path_off = resultArray<uint64_t>(nAnchors + 1); res->anchor_path = resultArray<uint32_t>(nPath);
		res->anchor_first_node = resultArray<uint32_t>(nAnchors); res->anchor_first_offset = resultArray<uint32_t>(nAnchors); res->anchor_first_seqpos = resultArray<uint32_t>(nAnchors);
		res->anchor_last_node = resultArray<uint32_t>(nAnchors); res->anchor_last_offset = resultArray<uint32_t>(nAnchors); res->anchor_last_seqpos = resultArray<uint32_t>(nAnchors);
		res->anchor_score = resultArray<int32_t>(nAnchors);
		if (anchorTraces) {
			res->anchor_trace_off = resultArray<uint64_t>(nAnchors + 1);
			res->anchor_trace_node = resultArray<int32_t>(nTrace); res->anchor_trace_offset = resultArray<uint32_t>(nTrace);
			res->anchor_trace_seqpos = resultArray<uint32_t>(nTrace); res->anchor_trace_switch = resultArray<uint8_t>(nTrace);
			res->anchor_trace_off[nAnchors] = nTrace;
		}
		res->read_chain_off = resultArray<uint64_t>(n + 1);
		res->chain = resultArray<uint32_t>(nChain);
		res->chain_score = resultArray<uint64_t>(n);
		res->read_longall_off = resultArray<uint64_t>(n + 1);
		res->read_longall_off[n] = nLong;
		res->longall_start = resultArray<uint32_t>(nLong); res->longall_end = resultArray<uint32_t>(nLong); res->longall_score = resultArray<uint32_t>(nLong);
		res->long_trace_off = resultArray<uint64_t>(nLong + 1);
		res->long_trace_off[nLong] = nLongTrace;
		res->long_trace_node = resultArray<int32_t>(nLongTrace); res->long_trace_offset = resultArray<uint32_t>(nLongTrace);
		res->long_trace_seqpos = resultArray<uint32_t>(nLongTrace); res->long_trace_switch = resultArray<uint8_t>(nLongTrace);
		res->seeds_extended_long = resultArray<uint64_t>(n);
		res->read_long_off = resultArray<uint64_t>(n + 1);
		res->read_long_off[n] = nLongSelected;
		res->long_index = resultArray<uint32_t>(nLongSelected);
		res->long_edit_distance = resultArray<int64_t>(n); res->chain_edit_distance = resultArray<int64_t>(n); res->chained_better = resultArray<uint8_t>(n);
		res->read_path_off = resultArray<uint64_t>(n + 1);
		res->read_path_off[n] = nStitched;
		res->path_node = resultArray<uint32_t>(nStitched);
		res->path_first_offset = resultArray<uint32_t>(n); res->path_last_offset = resultArray<uint32_t>(n); res->path_cells = resultArray<uint64_t>(n);
		res->read_chain_trace_off = resultArray<uint64_t>(n + 1);
		res->read_chain_trace_off[n] = nChainTrace;
		res->chain_trace_node = resultArray<int32_t>(nChainTrace); res->chain_trace_offset = resultArray<uint32_t>(nChainTrace);
		res->chain_trace_seqpos = resultArray<uint32_t>(nChainTrace); res->chain_trace_switch = resultArray<uint8_t>(nChainTrace);
		res->chain_aln_start = resultArray<uint32_t>(n); res->chain_aln_end = resultArray<uint32_t>(n);
		res->failed_assertion = resultArray<uint8_t>(n);
		res->capacity_exceeded = resultArray<uint8_t>(n);
		res->seeds_extended = resultArray<uint64_t>(n);
		res->flatten_ties = resultArray<uint32_t>(n); res->flatten_ties_long = resultArray<uint32_t>(n);
		res->read_seed_off[n] = nSeedsOut; res->read_anchor_off[n] = nAnchors; res->anchor_path_off[nAnchors] = nPath; res->read_chain_off[n] = nChain;
		pool.run(n, [&](size_t r, size_t) {
			const ReadGlue& gl = glue[r];
			res->read_seed_off[r] = seedOutBegin[r];
			if (keepSeeds) {
				uint64_t at = seedOutBegin[r];
				for (uint32_t k = 0; k < gl.nSeedsR; k++, at++) {
					const FragSeed& s = readSeeds[gl.seedBegin + k];   // fragment-pass order; pad = seedGoodness
					res->seed_node[at] = s.node; res->seed_offset[at] = s.offset; res->seed_seqpos[at] = s.seqPos; res->seed_goodness[at] = s.pad;
				}
			}
			res->read_anchor_off[r] = gl.anchorBegin;
			res->read_chain_off[r] = gl.chainBegin;
			res->chain_score[r] = chainScore[r];
			res->failed_assertion[r] = failedAssertion[r];
			res->capacity_exceeded[r] = gl.capacityExceeded ? 1 : 0;
			res->seeds_extended[r] = seedsExtended[r];
			res->seeds_extended_long[r] = seedsExtendedLong[r];
			res->flatten_ties[r] = readTies[r];                                        // (counted for every extension the reference would have run, whatever became of the read)
			res->flatten_ties_long[r] = P->long_pass ? hLongResults[r].pad : 0;
			res->read_longall_off[r] = gl.longBegin;
			res->read_path_off[r] = gl.stitchedBegin;
			res->read_long_off[r] = gl.longSelectedBegin;
			for (size_t i = 0; i < gl.longSelected.size(); i++) res->long_index[gl.longSelectedBegin + i] = gl.longSelected[i];
			res->long_edit_distance[r] = gl.longEditDistance;
			res->chain_edit_distance[r] = gl.chainEditDistance;
			// src/Aligner.cpp:901-905: the chained alignment wins when there is no whole-read alignment or its edit distance is larger
			res->chained_better[r] = gl.chainWins ? 1 : 0;
			res->read_chain_trace_off[r] = gl.chainTraceBegin;
			if (!gl.chainTraceNode.empty()) {
				memcpy(res->chain_trace_node + gl.chainTraceBegin, gl.chainTraceNode.data(), gl.chainTraceNode.size() * sizeof(int32_t));
				memcpy(res->chain_trace_offset + gl.chainTraceBegin, gl.chainTraceOffset.data(), gl.chainTraceOffset.size() * sizeof(uint32_t));
				memcpy(res->chain_trace_seqpos + gl.chainTraceBegin, gl.chainTraceSeqPos.data(), gl.chainTraceSeqPos.size() * sizeof(uint32_t));
				memcpy(res->chain_trace_switch + gl.chainTraceBegin, gl.chainTraceSwitch.data(), gl.chainTraceSwitch.size());
			}
			res->chain_aln_start[r] = gl.chainAlnStart; res->chain_aln_end[r] = gl.chainAlnEnd;
			for (size_t i = 0; i < gl.stitched.nodes.size(); i++) res->path_node[gl.stitchedBegin + i] = gl.stitched.nodes[i];
			res->path_first_offset[r] = gl.stitched.firstOffset; res->path_last_offset[r] = gl.stitched.lastOffset; res->path_cells[r] = gl.stitched.cells;
			{
				uint64_t la = gl.longBegin, lt = gl.longTraceBegin;
				for (const LongAln& al : gl.longAlns) {
					res->longall_start[la] = al.start; res->longall_end[la] = al.end; res->longall_score[la] = al.score;
					res->long_trace_off[la] = P->keep_traces ? lt : 0;
					if (P->keep_traces) for (uint32_t i = 0; i < al.traceLen; i++, lt++) {
						const LongCell& c = longCells[al.traceOff + i];
						res->long_trace_node[lt] = c.node; res->long_trace_offset[lt] = c.offset; res->long_trace_seqpos[lt] = c.seqPos; res->long_trace_switch[lt] = (uint8_t)c.nodeSwitch;
					}
					la++;
				}
			}
			for (uint32_t i = 0; i < chainLen[r]; i++) res->chain[gl.chainBegin + i] = chainOut[jobs[r].chainBegin + i];
			uint64_t a = gl.anchorBegin, pathAt = gl.pathBegin, traceAt = gl.traceBegin;
			if (deviceAnchors && gl.nAnchors) {   // the read's share of the dense arrays, to its place (the same place unless an earlier read of the batch lost its anchors to the whole-read pass)
				const uint64_t a0 = hAnchorOff[2 * r], w0 = hAnchorOff[2 * r + 1], cnt = gl.nAnchors;
				uint32_t* const to[9] = { res->anchor_x, res->anchor_y, res->anchor_first_node, res->anchor_first_offset, res->anchor_first_seqpos, res->anchor_last_node, res->anchor_last_offset, res->anchor_last_seqpos, (uint32_t*)res->anchor_score };
				for (int k = 0; k < 9; k++) memcpy(to[k] + gl.anchorBegin, denseWords(k) + a0, cnt * sizeof(uint32_t));
				const unsigned long long* off = densePathOff() + a0;
				for (uint64_t i = 0; i < cnt; i++) res->anchor_path_off[gl.anchorBegin + i] = off[i] - w0 + gl.pathBegin;
				memcpy(res->anchor_path + gl.pathBegin, denseWords(9) + w0, gl.nPath * sizeof(uint32_t));
			}
			if (!deviceAnchors) forEachAnchor(r, [&](uint64_t slot, uint64_t F) {
				const AnchorRec& rec = anchors[slot];
				res->anchor_x[a] = rec.x; res->anchor_y[a] = rec.y;
				res->anchor_path_off[a] = pathAt;
				for (uint32_t i = 0; i < rec.pathLen; i++) res->anchor_path[pathAt++] = pathPool[rec.pathOff + i];
				res->anchor_first_node[a] = rec.firstNode; res->anchor_first_offset[a] = rec.firstOffset; res->anchor_first_seqpos[a] = rec.firstSeqPos + frags[F].l;
				res->anchor_last_node[a] = rec.lastNode; res->anchor_last_offset[a] = rec.lastOffset; res->anchor_last_seqpos[a] = rec.lastSeqPos + frags[F].l;
				res->anchor_score[a] = rec.score;
				if (anchorTraces) {
					// merged trace in the reference's output coordinates (bigraph node id, offset in original node),
					// src/GraphAligner.h:527-565,590-608
					res->anchor_trace_off[a] = traceAt;
					const ExtResult& eb = extResults[2 * slot];
					const ExtResult& ef = extResults[2 * slot + 1];
					uint32_t p = slotSeqPos(r, slot, F) - frags[F].l;
					bool hasB = p > 0 && eb.status == EXT_OK, hasF = p < (uint32_t)P->split_len - 1 && ef.status == EXT_OK;
					if (hasB) {
						uint32_t use = hasF ? eb.traceLen - 1 : eb.traceLen;
						for (uint32_t i = 0; i < use; i++) {
							const PoolCell& c = tracePool[eb.traceOff + i];
							uint32_t off = c.offsetAndSwitch & 255u;
							auto rev = hg.GetReversePosition(hg.nodeIDs[c.node], hg.nodeOffset[c.node] + off);
							res->anchor_trace_node[traceAt] = rev.first;
							res->anchor_trace_offset[traceAt] = (uint32_t)rev.second;
							res->anchor_trace_seqpos[traceAt] = (uint32_t)((int32_t)p - 1 - c.seqPos);
							bool sw = i + 1 < eb.traceLen ? ((tracePool[eb.traceOff + i + 1].offsetAndSwitch >> 8) & 1) : false;
							res->anchor_trace_switch[traceAt] = sw ? 1 : 0;
							traceAt++;
						}
					}
					if (hasF) {
						for (uint32_t i = ef.traceLen; i-- > 0;) {
							const PoolCell& c = tracePool[ef.traceOff + i];
							uint32_t off = c.offsetAndSwitch & 255u;
							res->anchor_trace_node[traceAt] = hg.nodeIDs[c.node];
							res->anchor_trace_offset[traceAt] = (uint32_t)(hg.nodeOffset[c.node] + off);
							res->anchor_trace_seqpos[traceAt] = (uint32_t)((int32_t)p + 1 + c.seqPos);
							res->anchor_trace_switch[traceAt] = (c.offsetAndSwitch >> 8) & 1;
							traceAt++;
						}
					}
				}
				a++;
			});
		});
		assembleOutput();
		res->host_us[1] = nowUs() - tAsm;
		if (getenv("GC_DEBUG_TIMES")) {
			// what the stream holds on the device, largest first, and how much of the pools this batch used
			std::vector<std::pair<size_t, const char*>> sizes;
			size_t total = 0;
			st->forEachDeviceBuffer([&](const char* name, size_t bytes) { if (bytes) sizes.emplace_back(bytes, name); total += bytes; });
			std::sort(sizes.begin(), sizes.end(), [](const auto& l, const auto& r) { return l.first > r.first; });
			std::string line;
			for (size_t i = 0; i < sizes.size() && i < 14; i++) { char buf[96]; snprintf(buf, sizeof buf, " %s %.2f", sizes[i].second, sizes[i].first / 1073741824.0); line += buf; }
			fprintf(stderr, "[gc mem] stream %p holds %.2f GB on the device:%s\n", (void*)st, total / 1073741824.0, line.c_str());
			fprintf(stderr, "[gc mem] %llu reads, %llu seed occurrences, %llu fragments, %llu anchor slots (%llu extensions run); trace pool %.2f of %.2f G cells, anchor path pool %.2f of %.2f G words, whole-read cells %.2f of %.2f G\n",
				(unsigned long long)n, (unsigned long long)nSeedsTotal, (unsigned long long)nFrags, (unsigned long long)nSlots, (unsigned long long)res->counters[4], hSmall[1] / 1e9, traceBudget / 1e9, hSmall[2] / 1e9, pathCapacity / 1e9,
				P->long_pass ? hLongSmall[0] / 1e9 : 0.0, cellBudget / 1e9);
		}
		if (getenv("GC_DEBUG_TIMES")) fprintf(stderr, "[gc cpu] %.0f ms up to the end of the batch (the join came at %.0f)\n", processCpuMs() - cpuCall, cpuJoined - cpuCall);
		if (getenv("GC_DEBUG_TIMES")) fprintf(stderr, "[gc times] batch timeline (ms from the call): whole-read pass started %.1f, joined %.1f, assembly began %.1f, done %.1f\n", (tLongWall0 - tTotal) / 1e3, (tJoined - tTotal) / 1e3, (tAsm - tTotal) / 1e3, (nowUs() - tTotal) / 1e3);
	}
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

