// Output encoding on the device - SURVEY.md §8 row f2 (r4).
//
// The reference writes every read's alignments inside the worker that aligned it (src/Aligner.cpp:1003-1049: AddGAFLine ->
// GraphAlignerGAFAlignment::traceToAlignment, src/GraphAlignerGAFAlignment.h:38-196; AddAlignment -> GraphAlignerVGAlignment::traceToAlignment,
// src/GraphAlignerVGAlignment.h:36-163, replaceDigraphNodeIdsWithOriginalNodeIds src/Aligner.cpp:152-165). Both walk the alignment's trace cell by
// cell. The merged traces of the whole-read pass already sit in HBM (LongCell pool), 1.4 GB of them per 10 k x 10 kb batch; bringing them down and
// walking them on the host was what held the end-to-end rate at 42 % of the hot path's (r3). Here one wave walks one alignment, 64 cells at a
// time, and writes
//   - the GAF path column (">12<7...") and the cg:Z: CIGAR as text, and the counts the line's other columns are made of;
//   - the vg::Path of the alignment in proto3 wire format (mappings with position, edits, rank: src/vg.proto:52-109) - the bulk of a GAM record,
// so that only text / message bytes come down; the host adds the read name, the numeric columns and the two float tags (iostream formatting), or
// wraps the path into the vg::Alignment message (sequence, name, score, query_position, identity).
//
// Parallel form of the reference's loop. Per cell pos >= 1 the loop decides (a) whether the cell begins a new path step ("insideNode" is false:
// the previous cell is flagged nodeSwitch and the cell is not further along in the step's own node, src/GraphAlignerGAFAlignment.h:101), which
// depends on the node and offset at which the CURRENT step began - resolved by a short uniform loop over the chunk's flagged cells, everything
// else follows from ballots; (b) the edit kind: deletion (read position unchanged), insertion (same step, offset unchanged), else match /
// mismatch by the IUPAC sets of the read character and the graph letter. CIGAR items are runs of equal kinds; vg edits are runs of equal kinds
// inside one step; a run is emitted by the lane that begins the NEXT run (it knows the length), the positions of the emissions in the output come
// from wave prefix sums. Two passes: <false> counts bytes (and stores every mapping's edit bytes, which the mapping's length prefix needs before
// its edits are written), an exclusive scan over the jobs places them, <true> writes.
#include "gc_kernels.hpp"
#include <hip/hip_runtime.h>

namespace gcdev {

namespace {

__device__ __forceinline__ uint32_t decDigits(uint64_t v) { uint32_t d = 1; while (v >= 10) { v /= 10; d++; } return d; }
__device__ __forceinline__ void writeDec(char* out, uint64_t v, uint32_t digits) { for (int i = (int)digits - 1; i >= 0; i--) { out[i] = (char)('0' + v % 10); v /= 10; } }
__device__ __forceinline__ uint32_t varintLen(uint64_t v) { uint32_t n = 1; while (v >= 0x80) { v >>= 7; n++; } return n; }
__device__ __forceinline__ uint8_t* putVarint(uint8_t* p, uint64_t v) { while (v >= 0x80) { *p++ = (uint8_t)(v | 0x80); v >>= 7; } *p++ = (uint8_t)v; return p; }
__device__ __forceinline__ int prevBit(uint64_t mask, uint32_t lane) { const uint64_t below = mask & ((1ull << lane) - 1); return below ? 63 - __clzll((long long)below) : -1; }
__device__ __forceinline__ uint32_t scanInclusive(uint32_t v, uint32_t lane)
{
	for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(v, d); if ((int)lane >= d) v += o; }
	return v;
}
__device__ __forceinline__ uint64_t waveSum64(uint64_t v)
{
	for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
	return v;
}

// set of plain bases (A 1, C 2, G 4, T 8) the graph letter under (bigraph node, offset in the original node) stands for: GetUnitigNode + NodeSequences
// (src/GraphAlignerCommon.h:148-153); the split nodes of an original node are its 64-letter chunks in offset order. An all-zero one-hot column reads 'N' (:751-796).
__device__ __forceinline__ uint32_t graphLetterSet(const DGraph& g, int32_t nodeId, uint32_t offset)
{
	const uint32_t split = g.lookup[g.lookupOff[nodeId] + (offset >> 6)];
	const uint32_t pos = offset & 63u;
	if (split < g.firstAmbiguous) return 1u << ((g.nodeSeq[2 * (size_t)split + (pos >> 5)] >> ((pos & 31u) * 2)) & 3u);
	const uint64_t* p = g.ambSeq + 4 * (size_t)(split - g.firstAmbiguous);
	const uint32_t mask = (uint32_t)((p[0] >> pos) & 1) | (uint32_t)(((p[1] >> pos) & 1) << 1) | (uint32_t)(((p[2] >> pos) & 1) << 2) | (uint32_t)(((p[3] >> pos) & 1) << 3);
	return mask ? mask : 15u;
}

// bytes of one vg::Edit inside its mapping (tag + length + body): from_length 1, to_length 2, sequence 3 (src/vg.proto:52-56); kinds: 0 match, 1 mismatch, 2 insertion, 3 deletion
__device__ __forceinline__ uint32_t editBody(uint32_t kind, uint32_t len)
{
	const uint32_t v = 1 + varintLen(len);
	uint32_t body = 0;
	if (kind != 2) body += v;                          // from_length
	if (kind != 3) body += v;                          // to_length
	if (kind == 1 || kind == 2) body += v + len;       // sequence
	return body;
}

} // namespace

template <bool WRITE>
__global__ void __launch_bounds__(64) k_out_encode(DGraph g, OutNames names, const uint8_t* __restrict__ iupac, const OutJob* __restrict__ jobs, uint32_t nJobs, const LongCell* __restrict__ cellPool,
	const char* __restrict__ bases, OutRec* __restrict__ recs, const uint64_t* __restrict__ offsets, uint32_t* __restrict__ mapSizeAtCell, char* __restrict__ pathText, char* __restrict__ cigarText, uint8_t* __restrict__ vgBytes)
{
	GC_RAISE_PRIO();
	const uint32_t j = blockIdx.x, lane = threadIdx.x;
	if (j >= nJobs) return;
	const OutJob job = jobs[j];
	const uint32_t n = job.cellLen;
	if (n == 0) { if (!WRITE && lane == 0) recs[j] = OutRec {}; return; }
	const LongCell* cells = cellPool + job.cellOff;
	const char* read = bases + job.readOff;
	const bool merge = (job.flags & 1u) != 0, wantGaf = (job.flags & 2u) != 0, wantVg = (job.flags & 4u) != 0;
	// where this job's bytes go: offsets = [path text | cigar | vg] x (nJobs + 1), filled by k_out_place after the counting pass
	const uint64_t stride = (uint64_t)nJobs + 1;
	char* outPath = WRITE ? pathText + offsets[j] : nullptr;
	char* outCigar = WRITE ? cigarText + offsets[stride + j] : nullptr;
	uint8_t* outVg = WRITE ? vgBytes + offsets[2 * stride + j] : nullptr;
	uint32_t* mapSizes = mapSizeAtCell + job.cellOff;   // size of every mapping, at the index of the cell that opens it: left by the counting pass for the writing pass

	int32_t stepNode = 0; uint32_t stepOff = 0;      // node and offset at which the current path step began (currentNode / currentOffset)
	LongCell carry {};                               // the cell before this chunk's first
	uint32_t kindCarry = 0, ckCarry = 0;
	uint32_t cigStart = 0, edStart = 0;              // positions where the current CIGAR item / vg edit began
	uint32_t steps = 0;                              // path steps begun so far
	uint32_t matches = 0, mismatches = 0, insertions = 0, deletions = 0;
	uint64_t nodePathLen = 0;
	uint32_t pathCur = 0, cigCur = 0, vgCur = 0;     // bytes of each stream so far
	uint32_t mapAccum = 0, stepOpenPos = 0, stepPosBody = 0;   // counting pass: edit bytes of the open mapping so far, the cell that opened it, its position message's size

	// bytes of a mapping around its edits: the position message (always present) and the rank (src/vg.proto:62-66,89-94)
	auto positionBody = [&](int32_t node, uint32_t offset, uint32_t nameLen) -> uint32_t {
		uint32_t b = 0;
		if (node / 2 != 0) b += 1 + varintLen((uint64_t)(node / 2));
		if (offset != 0) b += 1 + varintLen(offset);
		if (node % 2 == 1) b += 2;
		if (nameLen) b += 1 + varintLen(nameLen) + nameLen;
		return b;
	};
	auto mappingSize = [&](uint32_t posBody, uint32_t editBytes, uint32_t rank) -> uint32_t { return 1 + varintLen(posBody) + posBody + editBytes + (rank ? 1 + varintLen(rank) : 0); };
	// one closed vg edit -> bytes at p (tag, length, fields; the sequence of a mismatch / insertion is the read under its cells,
	// except that cell 0 of the alignment contributes sequence[0] whatever its read position: src/GraphAlignerVGAlignment.h:73)
	auto writeEdit = [&](uint8_t* p, uint32_t kind, uint32_t start, uint32_t len) {
		*p++ = 0x12;
		p = putVarint(p, editBody(kind, len));
		if (kind != 2) { *p++ = 0x08; p = putVarint(p, len); }
		if (kind != 3) { *p++ = 0x10; p = putVarint(p, len); }
		if (kind == 1 || kind == 2) {
			*p++ = 0x1A; p = putVarint(p, len);
			for (uint32_t k = 0; k < len; k++) {
				const uint32_t sp = cells[start + k].seqPos;
				*p++ = (uint8_t)((start + k == 0) ? read[0] : (sp < job.readLen ? read[sp] : '-'));
			}
		}
	};

	for (uint32_t base = 0; base < n; base += 64) {
		const uint32_t pos = base + lane;
		const bool valid = pos < n;
		const uint32_t nValid = n - base < 64 ? n - base : 64;
		const LongCell c = cells[valid ? pos : n - 1];
		LongCell p;   // the cell before mine
		p.node = __shfl_up(c.node, 1); p.offset = __shfl_up(c.offset, 1); p.seqPos = __shfl_up(c.seqPos, 1); p.nodeSwitch = __shfl_up(c.nodeSwitch, 1);
		if (lane == 0) p = carry;
		// ---- (a) step starts: the reference's insideNode test, in order, for the cells whose predecessor is flagged
		uint64_t flagged = __ballot(valid && pos > 0 && p.nodeSwitch != 0);
		uint64_t startMask = base == 0 ? 1ull : 0ull;
		if (base == 0) { stepNode = __shfl(c.node, 0); stepOff = __shfl(c.offset, 0); }
		const int32_t stepNodeIn = stepNode;   // the step that was open when this chunk began
		while (flagged) {
			const int b = __ffsll((unsigned long long)flagged) - 1;
			flagged &= flagged - 1;
			const int32_t nb = __shfl(c.node, b);
			const uint32_t ob = __shfl(c.offset, b);
			if (!(nb == stepNode && ob > stepOff)) { startMask |= 1ull << b; stepNode = nb; stepOff = ob; }
		}
		const bool isStart = valid && ((startMask >> lane) & 1);
		// ---- (b) edit kinds: 0 match, 1 mismatch, 2 insertion, 3 deletion (4: the merged 'M' of the CIGAR)
		uint32_t kind;
		{
			const uint32_t readSet = c.seqPos < job.readLen ? iupac[(uint8_t)read[c.seqPos]] : 0u;
			const bool m = (readSet & graphLetterSet(g, c.node, c.offset)) != 0;
			if (pos == 0) kind = m ? 0u : 1u;
			else if (p.seqPos == c.seqPos) kind = 3u;
			else if (!isStart && p.offset == c.offset) kind = 2u;
			else kind = m ? 0u : 1u;
		}
		const uint32_t ck = (merge && kind < 2) ? 4u : kind;
		matches += (uint32_t)__popcll(__ballot(valid && kind == 0)); mismatches += (uint32_t)__popcll(__ballot(valid && kind == 1));
		insertions += (uint32_t)__popcll(__ballot(valid && kind == 2)); deletions += (uint32_t)__popcll(__ballot(valid && kind == 3));
		uint32_t kindPrev = __shfl_up(kind, 1), ckPrev = __shfl_up(ck, 1);
		if (lane == 0) { kindPrev = kindCarry; ckPrev = ckCarry; }
		const bool edBegins = valid && (isStart || kind != kindPrev);           // a vg edit begins at my cell (cell 0 is a step start)
		const bool cigBegins = valid && (pos == 0 || ck != ckPrev);             // a CIGAR item begins at my cell
		const uint64_t edMask = __ballot(edBegins), cigMask = __ballot(cigBegins);
		// ---- path steps
		const int prevStartLane = prevBit(startMask, lane);
		const uint32_t myStep = steps + (uint32_t)__popcll(startMask & ((1ull << lane) - 1));   // start lanes: index of the step that begins here
		uint32_t nameLen = 0, nameBegin = 0, idDigits = 0;
		if (isStart) {
			nameBegin = names.nameOff[c.node]; nameLen = names.nameOff[c.node + 1] - nameBegin;
			if (nameLen == 0) idDigits = decDigits((uint64_t)(c.node / 2));
		}
		{
			// nodePathLen (src/GraphAlignerGAFAlignment.h:104-113), modulo 2^64 like the reference's size_t arithmetic
			const int32_t inChunk = __shfl(c.node, prevStartLane >= 0 ? prevStartLane : 0);
			uint64_t add = 0;
			if (isStart) {
				if (pos == 0) add = g.origSize[c.node];
				else {
					const int32_t prevStepNode = prevStartLane >= 0 ? inChunk : stepNodeIn;
					const uint64_t skippedBefore = (uint64_t)g.origSize[prevStepNode] - 1 - p.offset, skippedAfter = c.offset;
					add = (uint64_t)g.origSize[c.node] - (skippedBefore + skippedAfter);
				}
			}
			nodePathLen += waveSum64(add);
		}
		if (wantGaf) {
			// path text: '>' or '<' and the node's name (its number when the GFA gave none) per step
			const uint32_t mine = isStart ? 1 + (nameLen ? nameLen : idDigits) : 0;
			const uint32_t incl = scanInclusive(mine, lane);
			if (WRITE && isStart) {
				char* o = outPath + pathCur + (incl - mine);
				*o++ = (c.node % 2) == 1 ? '<' : '>';
				if (nameLen) for (uint32_t k = 0; k < nameLen; k++) o[k] = names.nameBytes[nameBegin + k];
				else writeDec(o, (uint64_t)(c.node / 2), idDigits);
			}
			pathCur += __shfl(incl, 63);
			// CIGAR: the item that ends where mine begins
			uint32_t len = 0;
			if (cigBegins && pos > 0) { const int b = prevBit(cigMask, lane); len = pos - (b >= 0 ? base + (uint32_t)b : cigStart); }
			const uint32_t cmine = len ? decDigits(len) + 1 : 0;
			const uint32_t cincl = scanInclusive(cmine, lane);
			if (WRITE && len) {
				char* o = outCigar + cigCur + (cincl - cmine);
				writeDec(o, len, cmine - 1);
				o[cmine - 1] = "=XIDM"[ckPrev];
			}
			cigCur += __shfl(cincl, 63);
			if (cigMask) cigStart = base + (63 - (uint32_t)__clzll((long long)cigMask));
		}
		if (wantVg) {
			// the edit that ends where mine begins (it belongs to the mapping of the cell before mine)
			uint32_t elen = 0, estart = 0;
			if (edBegins && pos > 0) { const int b = prevBit(edMask, lane); estart = b >= 0 ? base + (uint32_t)b : edStart; elen = pos - estart; }
			const uint32_t ebody = elen ? editBody(kindPrev, elen) : 0;
			const uint32_t ebytes = elen ? 1 + varintLen(ebody) + ebody : 0;
			const uint32_t posBody = isStart ? positionBody(c.node, c.offset, nameLen) : 0;
			// emissions at my lane, in stream order: [the closing edit][rank of the closing mapping][header of the mapping that opens]
			const uint32_t rankBytes = (isStart && pos > 0 && myStep - 1 != 0) ? 1 + varintLen(myStep - 1) : 0;
			if (!WRITE) {
				// a mapping's size is known where it closes: its edits are those emitted after its opening lane up to and including the closing lane's;
				// stored at the opening cell's index for the writing pass, which needs it in the mapping's length prefix before any edit is written
				const uint32_t eincl = scanInclusive(ebytes, lane);
				const uint32_t eAtOpen = __shfl(eincl, prevStartLane >= 0 ? prevStartLane : 0);
				const uint32_t bodyAtOpen = __shfl(posBody, prevStartLane >= 0 ? prevStartLane : 0);
				uint32_t closed = 0;
				if (isStart && pos > 0) {
					const uint32_t editBytes = prevStartLane >= 0 ? eincl - eAtOpen : mapAccum + eincl;
					const uint32_t openPos = prevStartLane >= 0 ? base + (uint32_t)prevStartLane : stepOpenPos;
					const uint32_t ms = mappingSize(prevStartLane >= 0 ? bodyAtOpen : stepPosBody, editBytes, myStep - 1);
					mapSizes[openPos] = ms;
					closed = 1 + varintLen(ms) + ms;
				}
				vgCur += (uint32_t)waveSum64(closed);
				const int lastStart = startMask ? 63 - __clzll((long long)startMask) : -1;
				const uint32_t total = __shfl(eincl, 63);
				if (lastStart >= 0) { mapAccum = total - __shfl(eincl, lastStart); stepOpenPos = base + (uint32_t)lastStart; stepPosBody = __shfl(posBody, lastStart); }
				else mapAccum += total;
			} else {
				uint32_t mapSize = 0, headBytes = 0;
				if (isStart) { mapSize = mapSizes[pos]; headBytes = 1 + varintLen(mapSize) + 1 + varintLen(posBody) + posBody; }
				const uint32_t mine = ebytes + rankBytes + headBytes;
				const uint32_t incl = scanInclusive(mine, lane);
				if (mine) {
					uint8_t* o = outVg + vgCur + (incl - mine);
					if (elen) { writeEdit(o, kindPrev, estart, elen); o += ebytes; }
					if (rankBytes) { *o++ = 0x28; o = putVarint(o, myStep - 1); }
					if (isStart) {
						*o++ = 0x12; o = putVarint(o, mapSize);
						*o++ = 0x0A; o = putVarint(o, posBody);
						if (c.node / 2 != 0) { *o++ = 0x08; o = putVarint(o, (uint64_t)(c.node / 2)); }
						if (c.offset != 0) { *o++ = 0x10; o = putVarint(o, c.offset); }
						if (c.node % 2 == 1) { *o++ = 0x20; *o++ = 0x01; }
						if (nameLen) { *o++ = 0x2A; o = putVarint(o, nameLen); for (uint32_t k = 0; k < nameLen; k++) *o++ = (uint8_t)names.nameBytes[nameBegin + k]; }
					}
				}
				vgCur += __shfl(incl, 63);
			}
			if (edMask) edStart = base + (63 - (uint32_t)__clzll((long long)edMask));
		}
		steps += (uint32_t)__popcll(startMask);
		carry.node = __shfl(c.node, (int)nValid - 1); carry.offset = __shfl(c.offset, (int)nValid - 1); carry.seqPos = __shfl(c.seqPos, (int)nValid - 1); carry.nodeSwitch = __shfl(c.nodeSwitch, (int)nValid - 1);
		kindCarry = __shfl(kind, (int)nValid - 1); ckCarry = __shfl(ck, (int)nValid - 1);
	}
	// ---- the runs that are still open end with the trace (all lanes hold the same values; lane 0 writes)
	if (wantGaf) {
		const uint32_t len = n - cigStart, digits = decDigits(len);
		if (WRITE && lane == 0) { writeDec(outCigar + cigCur, len, digits); outCigar[cigCur + digits] = "=XIDM"[ckCarry]; }
		cigCur += digits + 1;
	}
	if (wantVg) {
		const uint32_t elen = n - edStart, ebody = editBody(kindCarry, elen), ebytes = 1 + varintLen(ebody) + ebody;
		const uint32_t rank = steps - 1, rankBytes = rank ? 1 + varintLen(rank) : 0;
		if (!WRITE) {
			const uint32_t ms = mappingSize(stepPosBody, mapAccum + ebytes, rank);
			if (lane == 0) mapSizes[stepOpenPos] = ms;
			vgCur += 1 + varintLen(ms) + ms;
		} else {
			if (lane == 0) {
				uint8_t* o = outVg + vgCur;
				writeEdit(o, kindCarry, edStart, elen); o += ebytes;
				if (rankBytes) { *o++ = 0x28; o = putVarint(o, rank); }
			}
			vgCur += ebytes + rankBytes;
		}
	}
	if (!WRITE && lane == 0) {
		OutRec r;
		r.pathTextLen = pathCur; r.cigarLen = cigCur; r.vgLen = vgCur; r.steps = steps;
		r.nodePathLen = nodePathLen;
		r.nodePathStart = cells[0].offset;
		r.nodePathEnd = nodePathLen - ((uint64_t)g.origSize[carry.node] - 1 - carry.offset);
		r.matches = matches; r.mismatches = mismatches; r.insertions = insertions; r.deletions = deletions;
		r.readStart = cells[0].seqPos; r.readEnd = carry.seqPos + 1;
		recs[j] = r;
	}
	if (WRITE && lane == 0) {
		// the writing pass must land exactly where the counting pass said it would
		const OutRec r = recs[j];
		if ((wantGaf && (pathCur != r.pathTextLen || cigCur != r.cigarLen)) || (wantVg && vgCur != r.vgLen)) recs[j].steps = 0xffffffffu;
	}
}

// exclusive scans of the three byte counts over the jobs -> offsets[3][nJobs + 1] (last entries = totals); one block
__global__ void __launch_bounds__(1024) k_out_place(const OutRec* __restrict__ recs, uint32_t nJobs, uint64_t* __restrict__ offsets, unsigned long long* __restrict__ totals)
{
	GC_RAISE_PRIO();
	__shared__ unsigned long long part[3][1024];
	const uint32_t t = threadIdx.x;
	const uint32_t per = (nJobs + 1023) / 1024;
	const uint32_t b = t * per < nJobs ? t * per : nJobs, e = b + per < nJobs ? b + per : nJobs;
	unsigned long long s0 = 0, s1 = 0, s2 = 0;
	for (uint32_t i = b; i < e; i++) { s0 += recs[i].pathTextLen; s1 += recs[i].cigarLen; s2 += recs[i].vgLen; }
	part[0][t] = s0; part[1][t] = s1; part[2][t] = s2;
	__syncthreads();
	if (t < 3) {
		unsigned long long run = 0;
		for (uint32_t i = 0; i < 1024; i++) { const unsigned long long v = part[t][i]; part[t][i] = run; run += v; }
		offsets[(uint64_t)t * (nJobs + 1) + nJobs] = run;
		totals[t] = run;
	}
	__syncthreads();
	unsigned long long r0 = part[0][t], r1 = part[1][t], r2 = part[2][t];
	const uint64_t stride = (uint64_t)nJobs + 1;
	for (uint32_t i = b; i < e; i++) {
		offsets[i] = r0; offsets[stride + i] = r1; offsets[2 * stride + i] = r2;
		r0 += recs[i].pathTextLen; r1 += recs[i].cigarLen; r2 += recs[i].vgLen;
	}
}

void launchOutCount(hipStream_t stream, const DGraph& g, const OutNames& names, const uint8_t* iupac, const OutJob* jobs, uint32_t nJobs, const LongCell* cellPool, const char* bases,
	OutRec* recs, uint64_t* offsets, uint32_t* mapSizeAtCell, unsigned long long* totals)
{
	if (!nJobs) return;
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_out_encode<false>), dim3(nJobs), dim3(64), 0, stream, g, names, iupac, jobs, nJobs, cellPool, bases, recs, (const uint64_t*)offsets, mapSizeAtCell, (char*)nullptr, (char*)nullptr, (uint8_t*)nullptr);
	hipLaunchKernelGGL(k_out_place, dim3(1), dim3(1024), 0, stream, (const OutRec*)recs, nJobs, offsets, totals);
}

void launchOutWrite(hipStream_t stream, const DGraph& g, const OutNames& names, const uint8_t* iupac, const OutJob* jobs, uint32_t nJobs, const LongCell* cellPool, const char* bases,
	OutRec* recs, const uint64_t* offsets, uint32_t* mapSizeAtCell, char* pathText, char* cigarText, uint8_t* vgBytes)
{
	if (!nJobs) return;
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_out_encode<true>), dim3(nJobs), dim3(64), 0, stream, g, names, iupac, jobs, nJobs, cellPool, bases, recs, offsets, mapSizeAtCell, pathText, cigarText, vgBytes);
}

} // namespace gcdev
