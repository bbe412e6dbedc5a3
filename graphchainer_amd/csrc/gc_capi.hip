// The C entry points of include/graphchainer_amd.h except gc_align_batch (gc_batch.hip): graphs, seeders, index cache, read batches, streams, edit distances, output formats.
#include "gc_runtime.hpp"
#include "host/gc_stageclock.hpp"
#include <malloc.h>

extern "C" {

const char* gc_last_error(void) { return g_lastError.c_str(); }
void gc_free(void* p) { free(p); }

// Global (NW) edit distances of n_pairs string pairs on the GPU: what edlibAlign(a, b, EDLIB_MODE_NW, EDLIB_TASK_DISTANCE)
// returns at src/Aligner.cpp:645,845.
int gc_edit_distance(const char* a, const uint64_t* a_off, const char* b, const uint64_t* b_off, uint64_t n_pairs, int64_t* out)
{
	if ((!a && n_pairs) || !a_off || (!b && n_pairs) || !b_off || !out) return fail(GC_ERR_INVALID, "null argument");
	return guarded([&]() {
		requireDevice();
		if (n_pairs == 0) return (int)GC_OK;
		if (n_pairs >= 0xffffffffull) throw std::runtime_error("too many pairs");
		const uint64_t aBytes = a_off[n_pairs], bBytes = b_off[n_pairs];
		std::vector<EdRead> reads(n_pairs);
		uint64_t words = 0;
		for (uint64_t i = 0; i < n_pairs; i++) {
			uint64_t len = b_off[i + 1] - b_off[i];
			if (len >= 0x7fffffffull || a_off[i + 1] - a_off[i] >= 0x7fffffffull) throw std::runtime_error("sequence too long");
			reads[i] = EdRead { b_off[i], words, (uint32_t)len, (uint32_t)((len + 63) / 64 + 1) };
			words += 4ull * reads[i].words;
		}
		std::vector<uint64_t> masks(words, 0);
		for (uint64_t i = 0; i < n_pairs; i++) buildEqMasks(b + b_off[i], reads[i].len, reads[i].words, masks.data() + reads[i].eqOff);
		std::vector<EdPair> pairs(n_pairs);
		for (uint64_t i = 0; i < n_pairs; i++) {
			uint32_t m = (uint32_t)(a_off[i + 1] - a_off[i]);
			pairs[i] = EdPair { a_off[i], m, 0, (uint32_t)i, std::max<uint32_t>(64, (m + reads[i].len) / 16) };
			if (const char* env = getenv("GC_ED_FIRST_K")) pairs[i].k = (uint32_t)std::max(1, atoi(env));   // test hook: the first band, as small as a whole-read alignment's own bound (score + clipped ends + 8)
		}
		DeviceBuffer dA, dB, dMasks, dReads, dPairs, dOut;
		char* pa = dA.reserve<char>(aBytes); char* pb = dB.reserve<char>(bBytes);
		uint64_t* pm = dMasks.reserve<uint64_t>(words);
		EdRead* pr = dReads.reserve<EdRead>(n_pairs);
		EdPair* pp = dPairs.reserve<EdPair>(n_pairs);
		int64_t* po = dOut.reserve<int64_t>(n_pairs);
		if (aBytes) HIP_CHECK(hipMemcpy(pa, a, aBytes, hipMemcpyHostToDevice));
		if (bBytes) HIP_CHECK(hipMemcpy(pb, b, bBytes, hipMemcpyHostToDevice));
		if (words) HIP_CHECK(hipMemcpy(pm, masks.data(), words * sizeof(uint64_t), hipMemcpyHostToDevice));
		HIP_CHECK(hipMemcpy(pr, reads.data(), n_pairs * sizeof(EdRead), hipMemcpyHostToDevice));
		EditDistanceRun run;
		auto readLenOf = [&](uint32_t r) { return reads[r].len; };
		launchEditDistances(run, nullptr, pairs.data(), out, (uint32_t)n_pairs, pp, po, pr, pb, pm, pa, nullptr, readLenOf, getenv("GC_ED_FIRST_K") != nullptr);   // (the test hook's band picks the kernel class as a whole-read pair's own bound does)
		finishEditDistances(run, nullptr, pairs.data(), out, (uint32_t)n_pairs, pp, po, pr, pb, pm, pa, nullptr);
		return (int)GC_OK;
	});
}

// The alignment path edlib returns for EDLIB_TASK_PATH (src/Aligner.cpp:845), for arbitrary string pairs: k_edit_distance for the
// distance, k_edit_path for the ops. Rows = a (the reference passes the path letters first), columns = b (the read).
int gc_edit_path(const char* a, const uint64_t* a_off, const char* b, const uint64_t* b_off, uint64_t n_pairs, const uint64_t* ops_off, uint8_t* ops, uint32_t* ops_len, int64_t* distance)
{
	if ((!a && n_pairs) || !a_off || (!b && n_pairs) || !b_off || !ops_off || !ops || !ops_len || !distance) return fail(GC_ERR_INVALID, "null argument");
	int rc = gc_edit_distance(a, a_off, b, b_off, n_pairs, distance);
	if (rc != GC_OK) return rc;
	return guarded([&]() {
		if (n_pairs == 0) return (int)GC_OK;
		const uint64_t aBytes = a_off[n_pairs], bBytes = b_off[n_pairs];
		std::vector<EdPathJob> jobs(n_pairs);
		uint32_t maxQ = 1, maxT = 1;
		uint64_t opsEnd = 0;
		for (uint64_t i = 0; i < n_pairs; i++) {
			const uint32_t q = (uint32_t)(a_off[i + 1] - a_off[i]), t = (uint32_t)(b_off[i + 1] - b_off[i]);
			jobs[i] = EdPathJob { a_off[i], b_off[i], ops_off[i], q, t, (int32_t)distance[i], 0 };
			maxQ = std::max(maxQ, q); maxT = std::max(maxT, t);
			opsEnd = std::max<uint64_t>(opsEnd, ops_off[i] + q + t);
		}
		DeviceBuffer dA, dB, dJobs, dOps, dLen, dScratch;
		char* pa = dA.reserve<char>(aBytes); char* pb = dB.reserve<char>(bBytes);
		EdPathJob* pj = dJobs.reserve<EdPathJob>(n_pairs);
		uint8_t* po = dOps.reserve<uint8_t>(opsEnd);
		uint32_t* pl = dLen.reserve<uint32_t>(n_pairs);
		uint8_t* ps = dScratch.reserve<uint8_t>((uint64_t)editPathGridBlocks((uint32_t)n_pairs) * editPathScratchBytes(maxQ, maxT));
		if (aBytes) HIP_CHECK(hipMemcpy(pa, a, aBytes, hipMemcpyHostToDevice));
		if (bBytes) HIP_CHECK(hipMemcpy(pb, b, bBytes, hipMemcpyHostToDevice));
		HIP_CHECK(hipMemcpy(pj, jobs.data(), n_pairs * sizeof(EdPathJob), hipMemcpyHostToDevice));
		launchEditPath(nullptr, pj, (uint32_t)n_pairs, pa, pb, ps, maxQ, maxT, po, pl);
		HIP_CHECK(hipDeviceSynchronize());
		HIP_CHECK(hipMemcpy(ops_len, pl, n_pairs * sizeof(uint32_t), hipMemcpyDeviceToHost));
		if (opsEnd) HIP_CHECK(hipMemcpy(ops, po, opsEnd, hipMemcpyDeviceToHost));
		return (int)GC_OK;
	});
}

// E-value model of --E-cutoff (host only): out2 = {alignment score, E-value}; what SelectECutoff compares with the cut-off
// (src/AlignmentSelection.cpp:91-99, src/EValue.cpp:35-48).
int gc_evalue(double min_identity, uint64_t database_size, uint64_t query_size, uint64_t alignment_length, uint64_t num_edits, double* out2)
{
	if (!out2) return fail(GC_ERR_INVALID, "null argument");
	gc::EValueModel model(min_identity);
	out2[0] = model.alignmentScore(alignment_length, num_edits);
	out2[1] = model.evalue(database_size, query_size, alignment_length, num_edits);
	return GC_OK;
}

// Output of a batch's final alignments in the reference's formats (src/Aligner.cpp:1003-1023: the read's list sorted by
// alignmentStart, AddAlignment / AddGAFLine per alignment, sorted again; writeGAMToQueue :261-281, writeJSONToQueue :283-298,
// writeGAFToQueue :300-311). Needs a result produced with long_pass, keep_traces and edit_distances; a read whose chained
// alignment won (chained_better) is written from its chain_trace_* arrays (chain_traces >= 1), and counted as skipped only
// when the result carries no trace for it.
enum OutputKind { OUT_GAF, OUT_JSON, OUT_GAM };

// GC_GAM_DEVICE_HUFFMAN: every non-empty element of `groups` (a read's uncompressed group) becomes its gzip member, deflated on the device (hip/gc_deflate.hip: one
// dynamic-Huffman block of literals per read). The host stages the bytes in pinned memory, frames the members and computes their CRC-32s.
static void gzipGroupsOnDevice(std::vector<std::string>& groups, bool skipEmpty = true, bool lz = false)   // lz (r6, GC_GAM_DEVICE_LZ): LZ77 matches in front of the Huffman stage
{
	std::vector<uint32_t> which;
	std::vector<uint64_t> rawOff { 0 };
	for (size_t i = 0; i < groups.size(); i++) if (!skipEmpty || !groups[i].empty()) { which.push_back((uint32_t)i); rawOff.push_back(rawOff.back() + groups[i].size()); }
	const size_t m = which.size();
	if (!m) return;
	const uint64_t rawTotal = rawOff.back();
	requireDevice();
	int device = 0;
	HIP_CHECK(hipGetDevice(&device));
	const uint64_t outBound = rawTotal + 5 * (rawTotal / 65535 + m) + 4 * m;   // stored blocks are the worst case the plan accepts; 4-byte placement
	size_t at = 0;
	auto part = [&](size_t bytes) { const size_t here = at; at += (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255; return here; };
	const size_t oRaw = part(rawTotal), oRawOff = part((m + 1) * sizeof(uint64_t)), oLens = part(m * (size_t)std::max(260u, deflateLzLensStride())), oPlan = part(m * sizeof(uint2)), oOutOff = part((m + 1) * sizeof(uint64_t)), oOut = part(outBound);
	size_t deviceBytes = 0, pinnedBytes = 0;
	hipStream_t q = threadStream(device);
	// (the guards wait for the stream before a block goes back to its cache: on an exception between a launch and the wait below the block would otherwise be handed to the next caller
	// while this thread's copies and kernels still use it - ADVICE r4)
	char* D = (char*)g_deflateDeviceBlocks.get(at, device, deviceBytes);
	struct DeviceReturn { char* p; size_t bytes; int device; hipStream_t q; ~DeviceReturn() { (void)hipStreamSynchronize(q); g_deflateDeviceBlocks.put(p, bytes, device); } } deviceReturn { D, deviceBytes, device, q };
	size_t hat = 0;
	auto hpart = [&](size_t bytes) { const size_t here = hat; hat += (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255; return here; };
	const size_t hRaw = hpart(std::max<uint64_t>(rawTotal, outBound)), hRawOff = hpart((m + 1) * sizeof(uint64_t)), hPlan = hpart(m * sizeof(uint2)), hOutOff = hpart((m + 1) * sizeof(uint64_t));
	char* H = (char*)g_deflatePinnedBlocks.get(hat, device, pinnedBytes);
	struct PinnedReturn { char* p; size_t bytes; int device; hipStream_t q; ~PinnedReturn() { (void)hipStreamSynchronize(q); g_deflatePinnedBlocks.put(p, bytes, device); } } pinnedReturn { H, pinnedBytes, device, q };
	WorkerPool::instance().run(m, [&](size_t k, size_t) { memcpy(H + hRaw + rawOff[k], groups[which[k]].data(), groups[which[k]].size()); });
	memcpy(H + hRawOff, rawOff.data(), (m + 1) * sizeof(uint64_t));
	if (rawTotal) HIP_CHECK(hipMemcpyAsync(D + oRaw, H + hRaw, rawTotal, hipMemcpyHostToDevice, q));
	HIP_CHECK(hipMemcpyAsync(D + oRawOff, H + hRawOff, (m + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, q));
	(lz ? launchDeflateLzPlan : launchDeflatePlan)(q, (const uint8_t*)(D + oRaw), (const uint64_t*)(D + oRawOff), (uint32_t)m, (uint8_t*)(D + oLens), (uint2*)(D + oPlan));
	HIP_CHECK(hipMemcpyAsync(H + hPlan, D + oPlan, m * sizeof(uint2), hipMemcpyDeviceToHost, q));
	HIP_CHECK(hipStreamSynchronize(q));
	const uint2* plan = (const uint2*)(H + hPlan);
	uint64_t* outOff = (uint64_t*)(H + hOutOff);
	outOff[0] = 0;
	for (size_t k = 0; k < m; k++) outOff[k + 1] = outOff[k] + (((uint64_t)plan[k].x + 3) & ~(uint64_t)3);
	if (outOff[m] > outBound) throw std::runtime_error("device deflate: planned sizes exceed the bound");
	HIP_CHECK(hipMemcpyAsync(D + oOutOff, outOff, (m + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, q));
	(lz ? launchDeflateLzWrite : launchDeflateWrite)(q, (const uint8_t*)(D + oRaw), (const uint64_t*)(D + oRawOff), (uint32_t)m, (const uint8_t*)(D + oLens), (const uint2*)(D + oPlan), (uint8_t*)(D + oOut), (const uint64_t*)(D + oOutOff));
	HIP_CHECK(hipMemcpyAsync(H + hRaw, D + oOut, outOff[m], hipMemcpyDeviceToHost, q));
	HIP_CHECK(hipStreamSynchronize(q));
	WorkerPool::instance().run(m, [&](size_t k, size_t) { std::string& g = groups[which[k]]; g = gc::gzipMember((const uint8_t*)(H + hRaw + outOff[k]), plan[k].x, g); });
}

static int formatBatch(const gc_graph* G, const gc_result* r, const char* const* read_names, const char* bases, const uint64_t* offsets, OutputKind kind, int cigar_match_mismatch_merge,
	char** out_text, uint64_t* out_len, uint64_t* n_chained_skipped, int gamLevel = -1)
{
	if (!G || !r || !read_names || !offsets || !out_text || !out_len) return fail(GC_ERR_INVALID, "null argument");
	const bool pieces = r->read_out_off != nullptr;   // the result carries the alignments as the device encoded them (gc_params::device_output)
	if (pieces) {
		// the pieces' CIGAR style (= / X or M) was fixed when the batch was aligned (device_output bit 0 / 1); a call that asks for the other style would get the device's pieces in one
		// style and the chained winners' lines (encoded here, on the host) in the other: refused
		if (kind == OUT_GAF && (r->device_output & 3) && ((r->device_output & 2) != 0) != (cigar_match_mismatch_merge != 0))
			return fail(GC_ERR_INVALID, "gc_format_gaf: cigar_match_mismatch_merge differs from the style the result's pieces were encoded in (gc_params::device_output 1: = / X, 2: M)");
		if (kind == OUT_GAF && !r->out_cigar_off) return fail(GC_ERR_INVALID, "the result holds no GAF pieces");
		if (kind == OUT_GAF && r->out_cigar_off[r->read_out_off[r->n_reads]] == 0 && r->out_vg_off[r->read_out_off[r->n_reads]] != 0) return fail(GC_ERR_INVALID, "the result was produced without the GAF pieces (gc_params::device_output & 3)");
		if (kind != OUT_GAF && r->out_vg_off[r->read_out_off[r->n_reads]] == 0 && r->out_cigar_off[r->read_out_off[r->n_reads]] != 0) return fail(GC_ERR_INVALID, "the result was produced without the vg::Path bytes (gc_params::device_output & 4)");
	}
	if (!pieces && (!r->long_trace_off || !r->read_long_off || !r->long_index)) return fail(GC_ERR_INVALID, "the output encoders need a result with long_pass, keep_traces and edit_distances");
	return guarded([&]() {
		const uint64_t n = r->n_reads;
		std::vector<std::string> perRead(n);
		std::atomic<uint64_t> skipped { 0 };
		const bool onDevice = gamLevel == GC_GAM_DEVICE_HUFFMAN || gamLevel == GC_GAM_DEVICE_LZ;
		auto group = [&](const std::vector<std::string>& messages) { return onDevice ? gc::gamGroupRaw(messages) : gc::gamGroup(messages, gamLevel); };
		WorkerPool::instance().run(n, [&](size_t i, size_t) {
			std::string& text = perRead[i];
			std::vector<std::string> messages;
			const std::string name = read_names[i] ? read_names[i] : "";
			const uint64_t len = offsets[i + 1] - offsets[i];
			if (pieces && !r->chained_better[i]) {
				// put together from what the device wrote: the path and CIGAR columns, or the vg::Path bytes, plus the numbers of the other columns / fields
				for (uint64_t e = r->read_out_off[i]; e < r->read_out_off[i + 1]; e++) {
					const uint64_t* num = r->out_numbers + 12 * e;
					gc::EncodedAlignment ea;
					ea.nodePathLen = num[0]; ea.nodePathStart = num[1]; ea.nodePathEnd = num[2]; ea.matches = num[3]; ea.mismatches = num[4]; ea.insertions = num[5]; ea.deletions = num[6];
					ea.cells = num[7]; ea.alignmentStart = num[8]; ea.alignmentEnd = num[9]; ea.score = (int32_t)num[11];
					if (kind == OUT_GAF) {
						ea.path = r->out_path_text + r->out_path_off[e]; ea.pathLen = r->out_path_off[e + 1] - r->out_path_off[e];
						ea.cigar = r->out_cigar_text + r->out_cigar_off[e]; ea.cigarLen = r->out_cigar_off[e + 1] - r->out_cigar_off[e];
						gc::appendGafLine(text, name, len, ea);
						text += '\n';
					} else {
						ea.vgPath = r->out_vg_path + r->out_vg_off[e]; ea.vgPathLen = r->out_vg_off[e + 1] - r->out_vg_off[e];
						if (kind == OUT_JSON) { text += gc::vgToJson(gc::vgFromEncoded(name, bases + offsets[i], ea)); text += '\n'; }
						else messages.push_back(gc::vgProtobufFromEncoded(name, bases + offsets[i], ea));
					}
				}
				if (kind == OUT_GAM && !messages.empty()) text = group(messages);
				return;
			}
			if (r->chained_better[i]) {
				// the chained alignment replaces the whole-read ones (src/Aligner.cpp:901-920): a single item, trace score 0
				const uint64_t t0 = r->read_chain_trace_off ? r->read_chain_trace_off[i] : 0, t1 = r->read_chain_trace_off ? r->read_chain_trace_off[i + 1] : 0;
				if (t1 == t0) { skipped++; return; }   // result produced without chain_traces
				gc::TraceView tv { r->chain_trace_node + t0, r->chain_trace_offset + t0, r->chain_trace_seqpos + t0, r->chain_trace_switch + t0, t1 - t0 };
				if (kind == OUT_GAF) {
					text += gc::formatGafLine(G->host, name, bases + offsets[i], len, tv, cigar_match_mismatch_merge != 0);
					text += '\n';
				} else {
					gc::VgAlignment aln = gc::buildVgAlignment(G->host, name, bases + offsets[i], len, tv, 0, r->chain_aln_start[i], r->chain_aln_end[i]);
					if (kind == OUT_JSON) { text += gc::vgToJson(aln); text += '\n'; }
					else text = group({ gc::vgToProtobuf(aln) });
				}
				return;
			}
			if (pieces) return;   // (a read without alignments)
			struct Item { uint32_t start; uint64_t aln; };
			std::vector<Item> items;
			for (uint64_t k = r->read_long_off[i]; k < r->read_long_off[i + 1]; k++) {
				uint64_t a = r->read_longall_off[i] + r->long_index[k];
				items.push_back(Item { r->longall_start[a], a });
			}
			if (items.empty()) return;
			auto byStart = [](const Item& l, const Item& rr) { return l.start < rr.start; };
			std::sort(items.begin(), items.end(), byStart);   // src/Aligner.cpp:1003
			std::sort(items.begin(), items.end(), byStart);   // :1023 (an unstable sort may move ties even in a sorted list)
			for (const Item& it : items) {
				uint64_t t0 = r->long_trace_off[it.aln], t1 = r->long_trace_off[it.aln + 1];
				gc::TraceView tv { r->long_trace_node + t0, r->long_trace_offset + t0, r->long_trace_seqpos + t0, r->long_trace_switch + t0, t1 - t0 };
				if (kind == OUT_GAF) {
					text += gc::formatGafLine(G->host, name, bases + offsets[i], len, tv, cigar_match_mismatch_merge != 0);
					text += '\n';
				} else {
					gc::VgAlignment aln = gc::buildVgAlignment(G->host, name, bases + offsets[i], len, tv, (int32_t)r->longall_score[it.aln], r->longall_start[it.aln], r->longall_end[it.aln]);
					if (kind == OUT_JSON) { text += gc::vgToJson(aln); text += '\n'; }
					else messages.push_back(gc::vgToProtobuf(aln));
				}
			}
			if (kind == OUT_GAM) text = group(messages);
		});
		if (kind == OUT_GAM && onDevice) gzipGroupsOnDevice(perRead, true, gamLevel == GC_GAM_DEVICE_LZ);
		uint64_t total = 0;
		for (const auto& t : perRead) total += t.size();
		char* buf = (char*)malloc(total + 1);
		if (!buf) throw std::runtime_error("out of memory");
		uint64_t at = 0;
		for (const auto& t : perRead) { memcpy(buf + at, t.data(), t.size()); at += t.size(); }
		buf[total] = 0;
		*out_text = buf;
		*out_len = total;
		if (n_chained_skipped) *n_chained_skipped = skipped.load();
		return (int)GC_OK;
	});
}

int gc_format_gaf(const gc_graph* G, const gc_result* r, const char* const* read_names, const char* bases, const uint64_t* offsets, int cigar_match_mismatch_merge,
	char** out_text, uint64_t* out_len, uint64_t* n_chained_skipped)
{
	return formatBatch(G, r, read_names, bases, offsets, OUT_GAF, cigar_match_mismatch_merge, out_text, out_len, n_chained_skipped);
}
int gc_format_json(const gc_graph* G, const gc_result* r, const char* const* read_names, const char* bases, const uint64_t* offsets, char** out_text, uint64_t* out_len, uint64_t* n_chained_skipped)
{
	return formatBatch(G, r, read_names, bases, offsets, OUT_JSON, 0, out_text, out_len, n_chained_skipped);
}
int gc_format_gam(const gc_graph* G, const gc_result* r, const char* const* read_names, const char* bases, const uint64_t* offsets, char** out_bytes, uint64_t* out_len, uint64_t* n_chained_skipped)
{
	// r6: without a level the members are deflated on the device with LZ77 matches (GC_GAM_DEVICE_LZ): the inflated stream is the contract - the reference's own bytes depend on its
	// zlib build - and zlib's default level on the host is a fifth of the hot path's rate (gc_format_gam_level(-1) still gives it)
	return formatBatch(G, r, read_names, bases, offsets, OUT_GAM, 0, out_bytes, out_len, n_chained_skipped, GC_GAM_DEVICE_LZ);
}

// gc_result_free keeps the large arrays of freed results (up to 24 GB) for the next batch instead of returning them to the allocator; this returns them.
void gc_result_cache_trim(void)
{
	g_resultBlocks.trim();
	// r5: the device / pinned blocks held back by gc_reads_destroy and by the device deflate's staging as well (blocks in use are not in the caches and are unaffected)
	g_readDeviceBlocks.trim(); g_readPinnedBlocks.trim(); g_deflateDeviceBlocks.trim(); g_deflatePinnedBlocks.trim();
}

// One alignment at a time, for a host that keeps the reference's own per-alignment calls (include/graphchainer_amd_shim.hpp: AddGAFLine / AddAlignment,
// src/GraphAlignerWrapper.h:43-44): the trace in output coordinates in, the GAF line / the vg::Alignment message out. Host code (no device work).
int gc_format_gaf_trace(const gc_graph* G, const char* read_name, const char* sequence, uint64_t sequence_len, const int32_t* node, const uint32_t* offset, const uint32_t* seqpos, const uint8_t* node_switch,
	uint64_t n, int cigar_match_mismatch_merge, char** out_text, uint64_t* out_len)
{
	if (!G || !sequence || !out_text || !out_len || (n && (!node || !offset || !seqpos || !node_switch))) return fail(GC_ERR_INVALID, "null argument");
	return guarded([&]() {
		gc::TraceView tv { node, offset, seqpos, node_switch, n };
		const std::string line = gc::formatGafLine(G->host, read_name ? read_name : "", sequence, sequence_len, tv, cigar_match_mismatch_merge != 0);
		char* buf = (char*)malloc(line.size() + 1);
		if (!buf) throw std::runtime_error("out of memory");
		memcpy(buf, line.data(), line.size());
		buf[line.size()] = 0;
		*out_text = buf; *out_len = line.size();
		return (int)GC_OK;
	});
}
static int formatVgTrace(const gc_graph* G, const char* read_name, const char* sequence, uint64_t sequence_len, const int32_t* node, const uint32_t* offset, const uint32_t* seqpos, const uint8_t* node_switch,
	uint64_t n, int32_t score, uint64_t alignment_start, uint64_t alignment_end, char** out_bytes, uint64_t* out_len, bool digraphIds);
int gc_format_vg_trace(const gc_graph* G, const char* read_name, const char* sequence, uint64_t sequence_len, const int32_t* node, const uint32_t* offset, const uint32_t* seqpos, const uint8_t* node_switch,
	uint64_t n, int32_t score, uint64_t alignment_start, uint64_t alignment_end, char** out_bytes, uint64_t* out_len)
{
	return formatVgTrace(G, read_name, sequence, sequence_len, node, offset, seqpos, node_switch, n, score, alignment_start, alignment_end, out_bytes, out_len, false);
}
// the message as AddAlignment leaves it (src/GraphAligner.h:205-212): digraph node ids (2 x segment index + strand), no names - for a host that calls
// replaceDigraphNodeIdsWithOriginalNodeIds itself right after (src/Aligner.cpp:1009), as the reference's own loop does
int gc_format_vg_trace_digraph(const gc_graph* G, const char* read_name, const char* sequence, uint64_t sequence_len, const int32_t* node, const uint32_t* offset, const uint32_t* seqpos, const uint8_t* node_switch,
	uint64_t n, int32_t score, uint64_t alignment_start, uint64_t alignment_end, char** out_bytes, uint64_t* out_len)
{
	return formatVgTrace(G, read_name, sequence, sequence_len, node, offset, seqpos, node_switch, n, score, alignment_start, alignment_end, out_bytes, out_len, true);
}
static int formatVgTrace(const gc_graph* G, const char* read_name, const char* sequence, uint64_t sequence_len, const int32_t* node, const uint32_t* offset, const uint32_t* seqpos, const uint8_t* node_switch,
	uint64_t n, int32_t score, uint64_t alignment_start, uint64_t alignment_end, char** out_bytes, uint64_t* out_len, bool digraphIds)
{
	if (!G || !sequence || !out_bytes || !out_len || (n && (!node || !offset || !seqpos || !node_switch))) return fail(GC_ERR_INVALID, "null argument");
	if (alignment_end < alignment_start || alignment_end > sequence_len) return fail(GC_ERR_INVALID, "gc_format_vg_trace: alignment_start / alignment_end outside the read");
	return guarded([&]() {
		gc::TraceView tv { node, offset, seqpos, node_switch, n };
		gc::VgAlignment aln = gc::buildVgAlignment(G->host, read_name ? read_name : "", sequence, sequence_len, tv, score, alignment_start, alignment_end);
		if (digraphIds) for (gc::VgMapping& m : aln.mappings) { m.nodeId = 2 * m.nodeId + (m.isReverse ? 1 : 0); m.name.clear(); }   // the inverse of replaceDigraphNodeIdsWithOriginalNodeIds (src/Aligner.cpp:152-165)
		const std::string msg = gc::vgToProtobuf(aln);
		char* buf = (char*)malloc(msg.size() + 1);
		if (!buf) throw std::runtime_error("out of memory");
		memcpy(buf, msg.data(), msg.size());
		buf[msg.size()] = 0;
		*out_bytes = buf; *out_len = msg.size();
		return (int)GC_OK;
	});
}
// The graph letter under each (bigraph node id, offset in the original node): what the reference's TraceItem constructor looks up (src/GraphAlignerCommon.h:148-153)
int gc_graph_letters(const gc_graph* G, const int32_t* node, const uint32_t* offset, uint64_t n, char* out)
{
	if (!G || !out || (n && (!node || !offset))) return fail(GC_ERR_INVALID, "null argument");
	return guarded([&]() {
		gc::GraphLetters letters(G->host);
		for (uint64_t i = 0; i < n; i++) {
			const size_t* size = G->host.originalNodeSize.find(node[i]);
			if (!size || offset[i] >= *size) throw std::runtime_error("gc_graph_letters: no such node / offset");
			out[i] = letters.at(node[i], offset[i]);
		}
		return (int)GC_OK;
	});
}

int gc_format_gam_level(const gc_graph* G, const gc_result* r, const char* const* read_names, const char* bases, const uint64_t* offsets, int level, char** out_bytes, uint64_t* out_len, uint64_t* n_chained_skipped)
{
	if ((level < -1 || level > 9) && level != GC_GAM_DEVICE_HUFFMAN && level != GC_GAM_DEVICE_LZ) return fail(GC_ERR_INVALID, "gc_format_gam_level: zlib levels are -1 (default) and 0..9, or GC_GAM_DEVICE_HUFFMAN / GC_GAM_DEVICE_LZ");
	return formatBatch(G, r, read_names, bases, offsets, OUT_GAM, 0, out_bytes, out_len, n_chained_skipped, level);
}

// gzip members of independent byte streams, deflated on the device as GC_GAM_DEVICE_HUFFMAN / GC_GAM_DEVICE_LZ do for the GAM groups (an empty stream gives an empty member)
static int gzipStreams(const uint8_t* bytes, const uint64_t* offsets, uint64_t n, char** out_bytes, uint64_t* out_offsets, bool lz);
int gc_gzip_streams(const uint8_t* bytes, const uint64_t* offsets, uint64_t n, char** out_bytes, uint64_t* out_offsets) { return gzipStreams(bytes, offsets, n, out_bytes, out_offsets, false); }
int gc_gzip_streams_lz(const uint8_t* bytes, const uint64_t* offsets, uint64_t n, char** out_bytes, uint64_t* out_offsets) { return gzipStreams(bytes, offsets, n, out_bytes, out_offsets, true); }
static int gzipStreams(const uint8_t* bytes, const uint64_t* offsets, uint64_t n, char** out_bytes, uint64_t* out_offsets, bool lz)
{
	if (!offsets || !out_bytes || !out_offsets || (offsets[n] && !bytes)) return fail(GC_ERR_INVALID, "null argument");
	for (uint64_t i = 0; i < n; i++) if (offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[i] >= (1ull << 32)) return fail(GC_ERR_INVALID, "gc_gzip_streams: offsets must ascend, streams below 4 GB");
	return guarded([&]() {
		std::vector<std::string> groups(n);
		for (uint64_t i = 0; i < n; i++) groups[i].assign((const char*)bytes + offsets[i], (const char*)bytes + offsets[i + 1]);
		gzipGroupsOnDevice(groups, false, lz);
		out_offsets[0] = 0;
		for (uint64_t i = 0; i < n; i++) out_offsets[i + 1] = out_offsets[i] + groups[i].size();
		char* buf = (char*)malloc(out_offsets[n] + 1);
		if (!buf) throw std::runtime_error("out of memory");
		for (uint64_t i = 0; i < n; i++) memcpy(buf + out_offsets[i], groups[i].data(), groups[i].size());
		*out_bytes = buf;
		return (int)GC_OK;
	});
}

// Test entry: the permutation the device's replay of libstdc++'s std::sort (hip/gc_stdsort_wave.hpp, what k_seed_glue runs for the reference's three order-critical unstable sorts)
// gives for arrays of 32-bit keys: array s = keys[offsets[s] .. offsets[s + 1]), perm_out likewise - perm_out[offsets[s] + i] = the index (inside its array) of the element
// that ends at place i. depth_limit < 0: the reference's 2 floor(log2 n); smaller values force the heapsort path.
int gc_std_sort_permutations(const uint32_t* keys, const uint64_t* offsets, uint64_t n_arrays, int64_t depth_limit, uint32_t* perm_out)
{
	if (!offsets || !perm_out || (!keys && n_arrays && offsets[n_arrays] > 0)) return fail(GC_ERR_INVALID, "null argument");
	return guarded([&]() {
		requireDevice();
		const uint64_t total = offsets[n_arrays];
		if (n_arrays >= 0xffffffffull) throw std::runtime_error("too many arrays");
		std::vector<unsigned long long> elems(total);
		for (uint64_t s = 0; s < n_arrays; s++) {
			if (offsets[s + 1] < offsets[s] || offsets[s + 1] - offsets[s] >= 0xffffffffull) throw std::runtime_error("bad offsets");
			for (uint64_t i = offsets[s]; i < offsets[s + 1]; i++) elems[i] = ((unsigned long long)keys[i] << 32) | (i - offsets[s]);
		}
		DeviceBuffer dElems, dOff, dScratch;
		unsigned long long* de = dElems.reserve<unsigned long long>(total);
		uint64_t* dof = dOff.reserve<uint64_t>(n_arrays + 1);
		uint32_t* ds = dScratch.reserve<uint32_t>(3 * total + 64 * n_arrays + 64);
		if (total) HIP_CHECK(hipMemcpy(de, elems.data(), total * sizeof(unsigned long long), hipMemcpyHostToDevice));
		HIP_CHECK(hipMemcpy(dof, offsets, (n_arrays + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
		launchTestStdSort(nullptr, de, dof, (uint32_t)n_arrays, ds, (long)depth_limit);
		HIP_CHECK(hipDeviceSynchronize());
		if (total) HIP_CHECK(hipMemcpy(elems.data(), de, total * sizeof(unsigned long long), hipMemcpyDeviceToHost));
		for (uint64_t i = 0; i < total; i++) perm_out[i] = (uint32_t)elems[i];
		return (int)GC_OK;
	});
}

int gc_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

// free and total bytes of the current device's memory (hipMemGetInfo): what a host sizing its batches - or a benchmark reporting its peak - needs
int gc_device_memory(uint64_t* free_bytes, uint64_t* total_bytes)
{
	if (!free_bytes || !total_bytes) return fail(GC_ERR_INVALID, "null argument");
	return guarded([&]() {
		requireDevice();
		size_t f = 0, t = 0;
		HIP_CHECK(hipMemGetInfo(&f, &t));
		*free_bytes = f; *total_bytes = t;
		return (int)GC_OK;
	});
}

int gc_set_device(int device)
{
	return guarded([&]() { HIP_CHECK(hipSetDevice(device)); return (int)GC_OK; });
}

void gc_params_default(gc_params* p)
{
	if (!p) return;
	p->bandwidth = 10;
	p->split_len = 35;
	p->split_gap = 35;
	p->colinear_gap = 10000;
	p->seed_density = 10;
	p->min_cluster_size = 1;
	p->long_pass = 0;
	p->keep_traces = 0;
	p->keep_seeds = 0;
	p->stitch = 1;
	p->edit_distances = 1;
	p->chain_traces = 1;
	p->device_output = 0;
	p->e_cutoff = -1;
	memset(&p->capacity, 0, sizeof(p->capacity));   // automatic
}

int gc_graph_create_from_gfa(const char* gfa_path, gc_graph** out)
{
	if (!gfa_path || !out) return fail(GC_ERR_INVALID, "null argument");
	*out = nullptr;
	gc_graph* G = new gc_graph();
	try {
		requireDevice();
		gc::StageClock clock;
		G->host = gc::AlignmentGraph::BuildFromGFAFile(gfa_path);
		G->host.buildMPC(true);
		malloc_trim(0);   // (the builders' freed small blocks back to the system: the host graph stays, and at config 5's sizes the host's memory is what bounds the graph)
		clock.lap("graph + MPC built");
		uploadGraph(G);
		clock.lap("graph uploaded");
	} catch (const DeviceError& e) {
		delete G;
		return fail(GC_ERR_DEVICE, e.what());
	} catch (const std::exception& e) {
		delete G;
		return fail(GC_ERR_GRAPH, e.what());
	}
	*out = G;
	return GC_OK;
}

int gc_graph_create(const gc_graph_desc* desc, gc_graph** out)
{
	if (!desc || !out) return fail(GC_ERR_INVALID, "null argument");
	*out = nullptr;
	gc_graph* G = new gc_graph();
	try {
		requireDevice();
		gc::AlignmentGraph& h = G->host;
		size_t n = desc->n_nodes;
		h.firstAmbiguous = desc->first_ambiguous;
		h.nodeLength.resize(n); h.nodeOffset.resize(n); h.nodeIDs.resize(n); h.reverse.resize(n); h.linearizable.assign(n, false);
		h.inNeighbors.resize(n); h.outNeighbors.resize(n);
		h.componentNumber.resize(n); h.chainNumber.resize(n); h.chainApproxPos.resize(n);
		for (size_t i = 0; i < n; i++) {
			h.nodeLength[i] = desc->node_length[i];
			h.nodeOffset[i] = desc->node_offset[i];
			h.nodeIDs[i] = desc->node_ids[i];
			h.reverse[i] = desc->node_ids[i] & 1;   // reverse-complement strand nodes have odd bigraph ids (src/BigraphToDigraph.cpp:101-104)
			h.bpSize += desc->node_length[i];
			for (uint64_t e = desc->in_off[i]; e < desc->in_off[i + 1]; e++) h.inNeighbors[i].push_back(desc->in_adj[e]);
			for (uint64_t e = desc->out_off[i]; e < desc->out_off[i + 1]; e++) h.outNeighbors[i].push_back(desc->out_adj[e]);
			h.componentNumber[i] = desc->component_number[i];
			h.chainNumber[i] = desc->chain_number[i];
			h.chainApproxPos[i] = desc->chain_approx_pos[i];
		}
		h.nodeSequences.resize(h.firstAmbiguous);
		for (size_t i = 0; i < h.firstAmbiguous; i++) h.nodeSequences[i] = { desc->node_seq[2 * i], desc->node_seq[2 * i + 1] };
		h.ambiguousNodeSequences.resize(n - h.firstAmbiguous);
		for (size_t i = 0; i < n - h.firstAmbiguous; i++) h.ambiguousNodeSequences[i] = { desc->ambiguous_seq[4 * i], desc->ambiguous_seq[4 * i + 1], desc->ambiguous_seq[4 * i + 2], desc->ambiguous_seq[4 * i + 3] };
		// nodeLookup: split nodes of each bigraph node in offset order
		std::vector<size_t> order(n);
		for (size_t i = 0; i < n; i++) order[i] = i;
		std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return h.nodeIDs[a] != h.nodeIDs[b] ? h.nodeIDs[a] < h.nodeIDs[b] : h.nodeOffset[a] < h.nodeOffset[b]; });
		// (without a lookup_order from the caller the minimizer index follows the order an unordered_map filled with the ids in ascending order iterates in - as r1-r3 built it)
		gc::HashOrder ascending;
		std::vector<int> ascendingIds;
		for (size_t at = 0; at < n;) {
			size_t end = at;
			while (end < n && h.nodeIDs[order[end]] == h.nodeIDs[order[at]]) end++;
			h.nodeLookup.add(h.nodeIDs[order[at]], order.data() + at, end - at);
			for (size_t k = at; k < end; k++) h.originalNodeSize[h.nodeIDs[order[at]]] += h.nodeLength[order[k]];
			ascending.insert(std::hash<int>()(h.nodeIDs[order[at]]));
			ascendingIds.push_back(h.nodeIDs[order[at]]);
			at = end;
		}
		for (uint32_t e : ascending.order()) h.nodeLookupOrder.push_back(ascendingIds[e]);
		if (desc->lookup_order) {
			// the host's nodeLookup iteration order: the minimizer index enumerates nodes in it (src/MinimizerSeeder.cpp:354-357)
			if (desc->n_lookup != h.nodeLookup.size()) throw std::runtime_error("lookup_order must list every bigraph node id once");
			std::unordered_set<int> seen;
			for (uint64_t i = 0; i < desc->n_lookup; i++)
				if (!h.nodeLookup.count(desc->lookup_order[i]) || !seen.insert(desc->lookup_order[i]).second) throw std::runtime_error("lookup_order must list every bigraph node id once");
			h.nodeLookupOrder.assign(desc->lookup_order, desc->lookup_order + desc->n_lookup);
		}
		h.finalized = true;
		h.buildMPC(true);
		uploadGraph(G);
	} catch (const DeviceError& e) {
		delete G;
		return fail(GC_ERR_DEVICE, e.what());
	} catch (const std::exception& e) {
		delete G;
		return fail(GC_ERR_GRAPH, e.what());
	}
	*out = G;
	return GC_OK;
}

void gc_graph_destroy(gc_graph* g) { delete g; malloc_trim(0); }   // (the host graph's many small blocks back to the system, not to the allocator's free lists: a graph loaded next would sit beside them)
uint64_t gc_graph_num_nodes(const gc_graph* g) { return g ? g->host.NodeSize() : 0; }
uint64_t gc_graph_size_bp(const gc_graph* g) { return g ? g->host.SizeInBP() : 0; }

int gc_graph_trim_host(gc_graph* G)
{
	if (!G) return fail(GC_ERR_INVALID, "null argument");
	gc::AlignmentGraph& h = G->host;
	std::vector<std::vector<std::vector<size_t>>>().swap(h.mpc);
	std::vector<std::vector<std::vector<size_t>>>().swap(h.paths);
	std::vector<std::vector<std::vector<std::pair<size_t, size_t>>>>().swap(h.backwards);
	std::vector<std::vector<size_t>>().swap(h.topo);
	std::vector<std::vector<size_t>>().swap(h.topo_ids);
	std::vector<std::vector<size_t>>().swap(h.component_ids);
	G->hostMpcTrimmed = true;
	malloc_trim(0);
	return GC_OK;
}

int gc_graph_array(const gc_graph* G, const char* name, int64_t** out, uint64_t* count)
{
	if (!G || !name || !out || !count) return fail(GC_ERR_INVALID, "null argument");
	if (G->hostMpcTrimmed && (strncmp(name, "mpc_", 4) == 0 || strncmp(name, "paths", 5) == 0 || strncmp(name, "back_", 5) == 0 || strcmp(name, "topo_id") == 0))
		return fail(GC_ERR_INVALID, "the host copy of the MPC index was released (gc_graph_trim_host)");
	const gc::AlignmentGraph& g = G->host;
	std::vector<int64_t> v;
	std::string nm = name;
	size_t n = g.NodeSize();
	if (nm == "nodeLength") for (size_t i = 0; i < n; i++) v.push_back(g.nodeLength[i]);
	else if (nm == "nodeOffset") for (size_t i = 0; i < n; i++) v.push_back(g.nodeOffset[i]);
	else if (nm == "nodeIDs") for (size_t i = 0; i < n; i++) v.push_back(g.nodeIDs[i]);
	else if (nm == "reverse") for (size_t i = 0; i < n; i++) v.push_back(g.reverse[i]);
	else if (nm == "componentNumber") for (size_t i = 0; i < n; i++) v.push_back(g.componentNumber[i]);
	else if (nm == "chainNumber") for (size_t i = 0; i < n; i++) v.push_back(g.chainNumber[i]);
	else if (nm == "chainApproxPos") for (size_t i = 0; i < n; i++) v.push_back(g.chainApproxPos[i]);
	else if (nm == "component_map") for (size_t i = 0; i < n; i++) v.push_back(g.component_map[i]);
	else if (nm == "out_off") { v.push_back(0); for (size_t i = 0; i < n; i++) v.push_back(v.back() + (int64_t)g.outNeighbors[i].size()); }
	else if (nm == "out_adj") for (size_t i = 0; i < n; i++) for (size_t x : g.outNeighbors[i]) v.push_back(x);
	else if (nm == "in_off") { v.push_back(0); for (size_t i = 0; i < n; i++) v.push_back(v.back() + (int64_t)g.inNeighbors[i].size()); }
	else if (nm == "in_adj") for (size_t i = 0; i < n; i++) for (size_t x : g.inNeighbors[i]) v.push_back(x);
	else if (nm == "mpc_width") for (size_t c = 0; c < g.mpc.size(); c++) v.push_back(g.mpc[c].size());
	else if (nm == "firstAmbiguous") v.push_back((int64_t)std::min(g.firstAmbiguous, n));
	else if (nm == "nodeSeq") for (size_t i = 0; i < n && i < g.firstAmbiguous; i++) { v.push_back((int64_t)g.nodeSequences[i][0]); v.push_back((int64_t)g.nodeSequences[i][1]); }   // bit patterns
	else if (nm == "ambiguousSeq") for (const gc::AmbiguousSeq& a : g.ambiguousNodeSequences) { v.push_back((int64_t)a.A); v.push_back((int64_t)a.T); v.push_back((int64_t)a.C); v.push_back((int64_t)a.G); }
	else if (nm == "component_idx") for (size_t i = 0; i < n; i++) v.push_back(g.component_idx[i]);
	else if (nm == "topo_id") for (size_t i = 0; i < n; i++) v.push_back(g.topo_ids[g.component_map[i]][g.component_idx[i]]);
	// the path cover: mpc_path_comp[p] = component of path p, mpc_path_off / mpc_path_nodes = its nodes (global ids) in order
	else if (nm == "mpc_path_comp") { for (size_t c = 0; c < g.mpc.size(); c++) for (size_t k = 0; k < g.mpc[c].size(); k++) v.push_back(c); }
	else if (nm == "mpc_path_off") { v.push_back(0); for (size_t c = 0; c < g.mpc.size(); c++) for (const auto& p : g.mpc[c]) v.push_back(v.back() + (int64_t)p.size()); }
	else if (nm == "mpc_path_nodes") { for (size_t c = 0; c < g.mpc.size(); c++) for (const auto& p : g.mpc[c]) for (size_t x : p) v.push_back(x); }
	// per node (global id): the path ids through it (local to its component), and the backward links (node = global id, path id)
	else if (nm == "paths_off") { v.push_back(0); for (size_t i = 0; i < n; i++) v.push_back(v.back() + (int64_t)g.paths[g.component_map[i]][g.component_idx[i]].size()); }
	else if (nm == "paths") { for (size_t i = 0; i < n; i++) for (size_t k : g.paths[g.component_map[i]][g.component_idx[i]]) v.push_back(k); }
	else if (nm == "back_off") { v.push_back(0); for (size_t i = 0; i < n; i++) v.push_back(v.back() + (int64_t)g.backwards[g.component_map[i]][g.component_idx[i]].size()); }
	else if (nm == "back_node") { for (size_t i = 0; i < n; i++) for (const auto& b : g.backwards[g.component_map[i]][g.component_idx[i]]) v.push_back(g.component_ids[g.component_map[i]][b.first]); }
	else if (nm == "back_path") { for (size_t i = 0; i < n; i++) for (const auto& b : g.backwards[g.component_map[i]][g.component_idx[i]]) v.push_back(b.second); }
	else if (nm == "lookupOrder") { for (int id : g.nodeLookupOrder) v.push_back(id); }
	else return fail(GC_ERR_INVALID, "unknown graph array " + nm);
	*out = mallocArray<int64_t>(v.size());
	memcpy(*out, v.data(), v.size() * sizeof(int64_t));
	*count = v.size();
	return GC_OK;
}

// ---- seeder -------------------------------------------------------------------------------------------
static inline uint32_t hostHashKmer(uint64_t kmer) { kmer *= 0x9E3779B97F4A7C15ull; return (uint32_t)(kmer >> 32); }

// hash table, membership filter and start offsets of a built minimizer index -> HBM
static void uploadSeeder(gc_seeder* S)
{
	size_t nKeys = S->host.kmers.size();
	size_t tableSize = 16;
	while (tableSize < 2 * nKeys) tableSize <<= 1;
	std::vector<uint64_t> table(tableSize, ~0ull);
	for (size_t i = 0; i < nKeys; i++) {
		uint32_t hsh = hostHashKmer(S->host.kmers[i]) & (uint32_t)(tableSize - 1);
		while (table[hsh] != ~0ull) hsh = (hsh + 1) & (uint32_t)(tableSize - 1);
		table[hsh] = ((S->host.kmers[i] & 0xffffffffull) << 32) | (uint64_t)i;   // (k <= 15: the whole k-mer; longer ones are verified against wideKmers)
	}
	if (nKeys >= 0xffffffffull) throw std::runtime_error("minimizer index too large for 32-bit key indices");
	S->dev.wideKmers = nullptr;
	if (S->host.k > 15) {
		uint64_t* dKmers = uploadVector(S->host.kmers);
		S->allocations.push_back(dKmers);
		S->dev.wideKmers = dKmers;
	}
	uint64_t* dTable = uploadVector(table);
	S->allocations.push_back(dTable);
	// membership filter in front of the table (see filterBit in gc_kernels.hip): ~8 bits per key, at most 2^28 bits; on cfg2 (5 M keys)
	// 2^25 bits = 4 MB, which stays in every XCD's L2 (a 32 MB filter missed to the fabric on every probe)
	uint32_t filterBits = 20;
	while (filterBits < 28 && (1ull << filterBits) < 8 * nKeys) filterBits++;
	if (const char* env = getenv("GC_TEST_SEED_FILTER_BITS")) filterBits = (uint32_t)std::max(10, std::min(30, atoi(env)));
	std::vector<uint32_t> filter((1ull << filterBits) / 32, 0);
	for (size_t i = 0; i < nKeys; i++) { uint32_t fb = (uint32_t)((S->host.kmers[i] * 0xD6E8FEB86659FD93ull) >> (64 - filterBits)); filter[fb >> 5] |= 1u << (fb & 31); }
	S->dev.filterShift = 64 - filterBits;
	uint32_t* dFilter = uploadVector(filter);
	S->allocations.push_back(dFilter);
	S->dev.filter = dFilter;
	uint64_t* dStart = uploadVector(S->host.startPos);
	S->allocations.push_back(dStart);
	S->dev.table = dTable;
	S->dev.tableMask = (uint32_t)(tableSize - 1);
	S->dev.startPos = dStart;
	uint64_t* dPositions = uploadVector(S->host.positions);   // the occurrence lists: the device expands the seeds itself (gc_seedglue.hip)
	S->allocations.push_back(dPositions);
	S->dev.positions = dPositions;
	S->dev.nKeys = (uint32_t)nKeys;
	S->dev.maxCount = (uint32_t)std::min<size_t>(S->host.maxCount, 0xffffffffu);
	S->dev.k = (int32_t)S->host.k;
	S->dev.w = (int32_t)S->host.w;
}

// The minimizer index built by the device (gc_minimizer.hip): window scan per bigraph node, one radix sort; the host only cuts the sorted
// pairs into the k-mer groups. Same index as gc::MinimizerIndex::Build (tests/test_index_cache.py compares them array by array).
// false = not applicable here (deque longer than the kernel's ring, GC_SEEDER_BUILD=host), the caller builds on the host.
static bool buildSeederOnDevice(const gc_graph* G, gc_seeder* S, size_t k, size_t w, double keepLeastFrequentFraction)
{
	if (const char* env = getenv("GC_SEEDER_BUILD")) if (!strcmp(env, "host")) return false;
	if (w - k + 2 > 32 || k > 15) return false;   // (the device build packs the k-mer into 30 bits of its sort key: longer minimizers are built on the host)
	const gc::AlignmentGraph& h = G->host;
	// minimizers ending inside an overlap prefix are skipped (src/MinimizerSeeder.cpp:323-340,369): never the case for the 0M graphs the
	// library loads, checked rather than assumed
	for (size_t i = 0; i < h.NodeSize(); i++) {
		if (h.nodeOffset[i] == 0) continue;
		for (size_t nb : h.inNeighbors[i]) if (h.nodeIDs[nb] != h.nodeIDs[i]) return false;
	}
	std::vector<int32_t> idOrder(h.nodeLookupOrder.begin(), h.nodeLookupOrder.end());   // arrival order at -t 1: nodeLookup iteration order (:354-357)
	if (idOrder.empty()) return false;
	int32_t* dOrder = uploadVector(idOrder);
	uint64_t *dKeys = nullptr, *dValues = nullptr;
	uint64_t n = gcdev::buildMinimizerPairsDevice(G->dev, dOrder, (uint32_t)idOrder.size(), (uint32_t)k, (uint32_t)w, &dKeys, &dValues);
	(void)hipFree(dOrder);
	if (n == ~0ull) { (void)hipGetLastError(); return false; }
	std::vector<uint64_t> keys(n);
	gc::MinimizerIndex& idx = S->host;
	idx = gc::MinimizerIndex();
	idx.k = k; idx.w = w;
	idx.positions.resize(n);
	if (n) {
		HIP_CHECK(hipMemcpy(keys.data(), dKeys, n * 8, hipMemcpyDeviceToHost));
		HIP_CHECK(hipMemcpy(idx.positions.data(), dValues, n * 8, hipMemcpyDeviceToHost));
		(void)hipFree(dKeys); (void)hipFree(dValues);
	}
	for (size_t i = 0; i < n; i++) {
		uint64_t kmer = keys[i] >> 34;
		if (idx.kmers.empty() || idx.kmers.back() != kmer) { idx.kmers.push_back(kmer); idx.startPos.push_back(i); }
	}
	idx.startPos.push_back(n);
	idx.maxCount = gc::minimizerMaxCount(idx.startPos, keepLeastFrequentFraction);
	return true;
}

static bool seederShapeOk(int64_t k, int64_t w) { return k >= 1 && k <= 31 && w >= k; }   // (the reference's range: src/AlignerMain.cpp:221,390)

int gc_seeder_create(const gc_graph* g, int32_t k, int32_t w, double keepFraction, gc_seeder** out)
{
	if (!g || !out) return fail(GC_ERR_INVALID, "null argument");
	if (!seederShapeOk(k, w)) return fail(GC_ERR_INVALID, "supported minimizer length is 1..31 with w >= k");
	*out = nullptr;
	gc_seeder* S = new gc_seeder();
	int rc = guarded([&]() {
		requireDevice();
		if (!buildSeederOnDevice(g, S, (size_t)k, (size_t)w, keepFraction)) S->host = gc::MinimizerIndex::Build(g->host, (size_t)k, (size_t)w, keepFraction);
		uploadSeeder(S);
		return (int)GC_OK;
	});
	if (rc != GC_OK) { delete S; return rc; }
	*out = S;
	return GC_OK;
}

// ---- index cache (host/gc_index_cache.hpp) -----------------------------------------------
int gc_index_build(const char* gfa_path, int32_t k, int32_t w, double keepFraction, const char* cache_path)
{
	if (!gfa_path || !cache_path) return fail(GC_ERR_INVALID, "null argument");
	if (k > 0 && !seederShapeOk(k, w)) return fail(GC_ERR_INVALID, "supported minimizer length is 1..31 with w >= k");
	try {
		gc::AlignmentGraph graph = gc::AlignmentGraph::BuildFromGFAFile(gfa_path);
		graph.buildMPC(true);
		if (k > 0) {
			gc::MinimizerIndex idx = gc::MinimizerIndex::Build(graph, (size_t)k, (size_t)w, keepFraction);
			gc::SaveIndexCache(cache_path, graph, &idx);
		} else {
			gc::SaveIndexCache(cache_path, graph, nullptr);
		}
	} catch (const std::exception& e) {
		return fail(GC_ERR_GRAPH, e.what());
	}
	return GC_OK;
}

int gc_index_save(const gc_graph* g, const gc_seeder* s, const char* cache_path)
{
	if (!g || !cache_path) return fail(GC_ERR_INVALID, "null argument");
	try {
		if (g->hostMpcTrimmed) return fail(GC_ERR_INVALID, "the host copy of the MPC index was released (gc_graph_trim_host): nothing to write the cache from");
		gc::SaveIndexCache(cache_path, g->host, s ? &s->host : nullptr);
	} catch (const std::exception& e) {
		return fail(GC_ERR_GRAPH, e.what());
	}
	return GC_OK;
}

static void fillIndexInfo(const gc::IndexCacheInfo& info, uint64_t* out)
{
	if (!out) return;
	out[0] = gc::INDEX_CACHE_VERSION; out[1] = info.nodes; out[2] = info.bp; out[3] = info.hasSeeder ? 1 : 0;
	out[4] = info.k; out[5] = info.w; out[6] = info.kmers; out[7] = info.positions;
}

int gc_index_check(const char* cache_path, uint64_t* info8)
{
	if (!cache_path) return fail(GC_ERR_INVALID, "null argument");
	try {
		fillIndexInfo(gc::CheckIndexCache(cache_path), info8);
	} catch (const std::exception& e) {
		return fail(GC_ERR_GRAPH, e.what());
	}
	return GC_OK;
}

int gc_index_load(const char* cache_path, gc_graph** graph_out, gc_seeder** seeder_out)
{
	if (!cache_path || !graph_out) return fail(GC_ERR_INVALID, "null argument");
	*graph_out = nullptr;
	if (seeder_out) *seeder_out = nullptr;
	gc_graph* G = new gc_graph();
	gc_seeder* S = new gc_seeder();
	try {
		requireDevice();
		gc::IndexCacheInfo info = gc::LoadIndexCache(cache_path, G->host, S->host);
		uploadGraph(G);
		if (info.hasSeeder && seeder_out) {
			if (!seederShapeOk((int64_t)S->host.k, (int64_t)S->host.w)) throw std::runtime_error("index cache holds a minimizer index this library cannot run");
			uploadSeeder(S);
		} else {
			delete S;
			S = nullptr;
		}
	} catch (const DeviceError& e) {
		delete G; delete S;
		return fail(GC_ERR_DEVICE, e.what());
	} catch (const std::exception& e) {
		delete G; delete S;
		return fail(GC_ERR_GRAPH, e.what());
	}
	*graph_out = G;
	if (seeder_out) *seeder_out = S;
	return GC_OK;
}

void gc_seeder_destroy(gc_seeder* s) { delete s; malloc_trim(0); }

int gc_seeder_array(const gc_seeder* s, const char* name, int64_t** out, uint64_t* count)
{
	if (!s || !name || !out || !count) return fail(GC_ERR_INVALID, "null argument");
	std::string nm = name;
	const std::vector<uint64_t>* src = nullptr;
	std::vector<uint64_t> single;
	if (nm == "kmers") src = &s->host.kmers;
	else if (nm == "start") src = &s->host.startPos;
	else if (nm == "positions") src = &s->host.positions;
	else if (nm == "maxcount") { single.push_back(s->host.maxCount); src = &single; }
	else if (nm == "k") { single.push_back((uint64_t)s->dev.k); src = &single; }
	else if (nm == "w") { single.push_back((uint64_t)s->dev.w); src = &single; }
	else return fail(GC_ERR_INVALID, "unknown seeder array " + nm);
	*out = mallocArray<int64_t>(src->size());
	for (size_t i = 0; i < src->size(); i++) (*out)[i] = (int64_t)(*src)[i];
	*count = src->size();
	return GC_OK;
}

// ---- streams / reads ------------------------------------------------------------------------------------
int gc_stream_create(gc_stream** out)
{
	if (!out) return fail(GC_ERR_INVALID, "null argument");
	*out = nullptr;
	gc_stream* st = new gc_stream();
	int rc = guarded([&]() {
		requireDevice();
		HIP_CHECK(hipGetDevice(&st->device));
		if (st->device < 0 || st->device >= 16) throw std::runtime_error("gc_stream_create: device index beyond the 16 per-device slots of the whole-read token and scratch");
		createStream(&st->stream, 0);       // non-blocking: uploads of another batch on the null stream do not serialise with this one
		createStream(&st->longStream, 0);
		for (auto& e : st->ev) HIP_CHECK(hipEventCreate(&e));
		for (auto& e : st->fragEv) HIP_CHECK(hipEventCreate(&e));
		for (auto& e : st->longEv) HIP_CHECK(hipEventCreate(&e));
		return (int)GC_OK;
	});
	if (rc != GC_OK) { delete st; return rc; }
	{ std::lock_guard<std::mutex> lock(g_longScratchCount); noteHardwareQueues(++g_longScratch[st->device & 15].streams); }
	*out = st;
	return GC_OK;
}
void gc_stream_destroy(gc_stream* st)
{
	if (!st) return;
	{
		std::lock_guard<std::mutex> count(g_longScratchCount);
		SharedLongScratch& shared = g_longScratch[st->device & 15];
		if (--shared.streams == 0) {   // the device's last stream: nobody can hold the token any more
			TokenHold alone;
			alone.lock(g_longPassToken[st->device & 15], 1, true);   // (every slot free and nobody beside: as a pass that fills the chip waits)
			int current = 0;
			if (hipGetDevice(&current) == hipSuccess) { (void)hipSetDevice(st->device); for (auto& b : shared.buffer) b.release(); (void)hipSetDevice(current); }
		}
	}
	delete st;
}

int gc_reads_upload(const char* bases, const uint64_t* offsets, uint64_t n, gc_reads** out)
{
	if (!offsets || !out || (!bases && n > 0 && offsets[n] > 0)) return fail(GC_ERR_INVALID, "null argument");
	*out = nullptr;
	gc_reads* R = new gc_reads();
	hipStream_t uploadStream = nullptr;
	int rc = guarded([&]() {
		requireDevice();
		R->offsets.assign(offsets, offsets + n + 1);
		R->totalBases = offsets[n];
		if (n >= 0xffffffffull) throw std::runtime_error("too many reads in one batch");
		// per-read layout of the bit vectors (host, O(n)); everything per base is derived on the device from one copy of the raw bases
		R->maskOff.assign(n, 0);
		R->maskWords.assign(n, 0);
		std::vector<EdRead> edReads(n);
		std::vector<uint64_t> eqOff(n);
		uint64_t totalWords = 0, eqWords = 0;
		for (uint64_t r = 0; r < n; r++) {
			R->maskOff[r] = totalWords;
			R->maskWords[r] = (uint32_t)((offsets[r + 1] - offsets[r] + 63) / 64 + 1);
			totalWords += 8ull * R->maskWords[r];
			edReads[r] = EdRead { offsets[r], eqWords, (uint32_t)(offsets[r + 1] - offsets[r]), R->maskWords[r] };
			eqOff[r] = eqWords;
			eqWords += 4ull * R->maskWords[r];
		}
		const uint64_t total = R->totalBases;
		// one device block for everything (256-byte aligned parts), one pinned block for what goes up; both come from small caches (see BlockCache)
		HIP_CHECK(hipGetDevice(&R->device));
		size_t at = 0;
		auto part = [&](size_t bytes) { const size_t here = at; at += (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255; return here; };
		const size_t oBases = part(2 * total), oOffsets = part((n + 1) * sizeof(uint64_t)), oMasks = part(totalWords * sizeof(uint64_t)), oEqMasks = part(eqWords * sizeof(uint64_t)),
			oEdReads = part(n * sizeof(EdRead)), oPacked = part(((total >> 5) + 1) * sizeof(uint64_t)), oInvalid = part(((total >> 6) + 1) * sizeof(uint64_t)), oChunkRead = part(((total >> 6) + 1) * sizeof(uint32_t)),
			oMaskOff = part(n * sizeof(uint64_t)), oMaskWords = part(n * sizeof(uint32_t)), oEqOff = part(n * sizeof(uint64_t)), oReadInvalid = part(n);
		R->deviceBlock = g_readDeviceBlocks.get(at, R->device, R->deviceBlockBytes);
		char* D = (char*)R->deviceBlock;
		R->devBases = D + oBases; R->devOffsets = (uint64_t*)(D + oOffsets); R->devMasks = (uint64_t*)(D + oMasks); R->devEqMasks = (uint64_t*)(D + oEqMasks); R->devEdReads = (EdRead*)(D + oEdReads);
		R->devPacked = (uint64_t*)(D + oPacked); R->devInvalid = (uint64_t*)(D + oInvalid); R->devChunkRead = (uint32_t*)(D + oChunkRead); R->devReadInvalid = (uint8_t*)(D + oReadInvalid);
		uint64_t* pMaskOff = (uint64_t*)(D + oMaskOff);
		uint32_t* pMaskWords = (uint32_t*)(D + oMaskWords);
		uint64_t* pEqOff = (uint64_t*)(D + oEqOff);
		uint8_t* pInvalid = R->devReadInvalid;
		R->devMaskOff = pMaskOff; R->devMaskWords = pMaskWords;
		// staging: the bases and the five small arrays in one pinned block, copied asynchronously on a stream of the calling thread's own (a synchronous copy from pageable
		// memory goes through the runtime's bounce buffers at a few GB/s and its null-stream semantics)
		size_t hat = 0;
		auto hpart = [&](size_t bytes) { const size_t here = hat; hat += (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255; return here; };
		const size_t hBases = hpart(total), hOffsets = hpart((n + 1) * sizeof(uint64_t)), hMaskOff = hpart(n * sizeof(uint64_t)), hMaskWords = hpart(n * sizeof(uint32_t)), hEqOff = hpart(n * sizeof(uint64_t)),
			hEdReads = hpart(n * sizeof(EdRead)), hInvalidBack = hpart(n);
		size_t pinnedBytes = 0;
		struct { hipStream_t q; } up { threadStream(R->device) };
		uploadStream = up.q;
		char* H = (char*)g_readPinnedBlocks.get(hat, R->device, pinnedBytes);
		// (waits for the stream before the block goes back: after an exception between the copies / kernels and the wait below they may still be reading it - ADVICE r4)
		struct PinnedReturn { char* p; size_t bytes; int device; hipStream_t q; ~PinnedReturn() { (void)hipStreamSynchronize(q); g_readPinnedBlocks.put(p, bytes, device); } } pinnedReturn { H, pinnedBytes, R->device, up.q };
		if (total) memcpy(H + hBases, bases, total);
		memcpy(H + hOffsets, offsets, (n + 1) * sizeof(uint64_t));
		if (n) {
			memcpy(H + hMaskOff, R->maskOff.data(), n * sizeof(uint64_t)); memcpy(H + hMaskWords, R->maskWords.data(), n * sizeof(uint32_t));
			memcpy(H + hEqOff, eqOff.data(), n * sizeof(uint64_t)); memcpy(H + hEdReads, edReads.data(), n * sizeof(EdRead));
		}
		if (total) HIP_CHECK(hipMemcpyAsync(R->devBases, H + hBases, total, hipMemcpyHostToDevice, up.q));
		HIP_CHECK(hipMemcpyAsync(R->devOffsets, H + hOffsets, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, up.q));
		if (n) {
			HIP_CHECK(hipMemcpyAsync(pMaskOff, H + hMaskOff, n * sizeof(uint64_t), hipMemcpyHostToDevice, up.q));
			HIP_CHECK(hipMemcpyAsync(pMaskWords, H + hMaskWords, n * sizeof(uint32_t), hipMemcpyHostToDevice, up.q));
			HIP_CHECK(hipMemcpyAsync(pEqOff, H + hEqOff, n * sizeof(uint64_t), hipMemcpyHostToDevice, up.q));
			HIP_CHECK(hipMemcpyAsync(R->devEdReads, H + hEdReads, n * sizeof(EdRead), hipMemcpyHostToDevice, up.q));
		}
		launchPackReads(up.q, R->devOffsets, (uint32_t)n, total, R->devBases, pMaskOff, pMaskWords, R->devMasks, pEqOff, R->devEqMasks, pInvalid, R->devPacked, R->devInvalid, R->devChunkRead);
		R->invalid.assign(n, 0);
		if (n) HIP_CHECK(hipMemcpyAsync(H + hInvalidBack, pInvalid, n, hipMemcpyDeviceToHost, up.q));
		HIP_CHECK(hipStreamSynchronize(up.q));
		if (n) memcpy(R->invalid.data(), H + hInvalidBack, n);
		return (int)GC_OK;
	});
	if (rc != GC_OK) {
		if (uploadStream) (void)hipStreamSynchronize(uploadStream);   // ~gc_reads returns the device block to its cache: nothing of this call may still be writing to it
		delete R;
		return rc;
	}
	*out = R;
	return GC_OK;
}
void gc_reads_destroy(gc_reads* r) { delete r; }

// ---- the batch pipeline -----------------------------------------------------------------------------------

void gc_result_free(gc_result* r)
{
	if (!r) return;
	void* ptrs[] = { r->read_seed_off, r->seed_node, r->seed_offset, r->seed_seqpos, r->seed_goodness, r->read_anchor_off, r->anchor_x, r->anchor_y, r->anchor_path_off,
		r->anchor_path, r->anchor_first_node, r->anchor_first_offset, r->anchor_first_seqpos, r->anchor_last_node, r->anchor_last_offset, r->anchor_last_seqpos, r->anchor_score,
		r->anchor_trace_off, r->anchor_trace_node, r->anchor_trace_offset, r->anchor_trace_seqpos, r->anchor_trace_switch, r->read_chain_off, r->chain, r->chain_score,
		r->read_longall_off, r->longall_start, r->longall_end, r->longall_score, r->long_trace_off, r->long_trace_node, r->long_trace_offset, r->long_trace_seqpos, r->long_trace_switch,
		r->failed_assertion, r->seeds_extended, r->seeds_extended_long, r->read_path_off, r->path_node, r->path_first_offset, r->path_last_offset, r->path_cells,
		r->read_long_off, r->long_index, r->long_edit_distance, r->chain_edit_distance, r->chained_better,
		r->capacity_exceeded, r->read_chain_trace_off, r->chain_trace_node, r->chain_trace_offset, r->chain_trace_seqpos, r->chain_trace_switch, r->chain_aln_start, r->chain_aln_end,
		r->read_out_off, r->out_source, r->out_numbers, r->out_path_off, r->out_path_text, r->out_cigar_off, r->out_cigar_text, r->out_vg_off, r->out_vg_path, r->flatten_ties, r->flatten_ties_long };
	for (void* p : ptrs) g_resultBlocks.put(p);
	free(r);
}

// One batch through the whole path, stage by stage (r3: this was one 1 200-line function). A BatchRun holds what the stages share - the call's arguments, the sizes the seed
// stage leaves behind, the device and pinned-host buffers a later stage reads again - and every stage is one member function, in the batch's order:
//   seeds -> prepareWholeReadPass -> startWholeReadPass (the pass runs on its own host thread and stream from there: runLongGroup, then afterLongPass) -> [main thread, meanwhile]
//   fragmentPipeline -> resultsBack -> stitchAndChainDistances -> joinWholeReadPass -> chainedAlignments -> assemble.
// Capacities are per read (flags in the result), errors of the reference's own making per read or fragment (failed_assertion); only invalid arguments and device errors fail
// the call (they throw; gc_align_batch turns that into its return code, and ~BatchRun joins the pass thread first).

} // extern "C"
