// ORACLE support (test infrastructure): thin extern "C" shims over the few reference translation units
// that compile from their own sources with no third-party or generated headers:
//   /root/reference/src/WordSlice.h                          (header-only)
//   /root/reference/src/AlignmentCorrectnessEstimation.cpp   (+ ThreadReadAssertion.cpp)
//   /root/reference/edlib/src/edlib.cpp                      (vendored edlib)
//   /root/reference/src/EValue.cpp                           (Karlin-Altschul E-values of --E-cutoff)
// Built by oracle/Makefile into oracle/_ref/libref_units.so, only where /root/reference exists.
// No reference source is copied; the files are compiled where they lie.
#include <algorithm>
#include <cassert>
#include <cstdint>
#include <cstddef>
#include <limits>
#include <tuple>
#include <type_traits>
#include <utility>
#include "WordSlice.h"
#include "AlignmentCorrectnessEstimation.h"
#include "edlib.h"
#include "EValue.h"

typedef WordSlice<size_t, int32_t, uint64_t> RefWordSlice;

extern "C" {

void ref_merge(uint64_t avp, uint64_t avn, int32_t as, uint64_t bvp, uint64_t bvn, int32_t bs, uint64_t* vp, uint64_t* vn, int32_t* s)
{
	RefWordSlice r = RefWordSlice(avp, avn, as).mergeWith(RefWordSlice(bvp, bvn, bs));
	*vp = r.VP; *vn = r.VN; *s = r.scoreEnd;
}
int32_t ref_changed_min_score(uint64_t avp, uint64_t avn, int32_t as, uint64_t bvp, uint64_t bvn, int32_t bs) { return RefWordSlice(avp, avn, as).changedMinScore(RefWordSlice(bvp, bvn, bs)); }
int32_t ref_get_value(uint64_t vp, uint64_t vn, int32_t s, int row) { return RefWordSlice(vp, vn, s).getValue(row); }
int32_t ref_score_before_start(uint64_t vp, uint64_t vn, int32_t s) { return RefWordSlice(vp, vn, s).getScoreBeforeStart(); }

void ref_correctness_series(const int* mismatches, int n, double* correct, double* wrong, int* flags)
{
	AlignmentCorrectnessEstimationState st;
	for (int i = 0; i < n; i++) {
		st = st.NextState(mismatches[i], 64);
		correct[i] = st.CorrectLogOdds();
		wrong[i] = st.FalseLogOdds();
		flags[i] = (st.CurrentlyCorrect() ? 1 : 0) | (st.CorrectFromCorrect() ? 2 : 0) | (st.FalseFromCorrect() ? 4 : 0);
	}
}

// the two edlib calls on the per-read path: src/Aligner.cpp:645 (distance) and :845 (path)
long long ref_edit_distance(const char* a, uint64_t na, const char* b, uint64_t nb)
{
	EdlibAlignResult r = edlibAlign(a, (int)na, b, (int)nb, edlibNewAlignConfig(-1, EDLIB_MODE_NW, EDLIB_TASK_DISTANCE, NULL, 0));
	long long d = r.status == EDLIB_STATUS_OK ? r.editDistance : -1;
	edlibFreeAlignResult(r);
	return d;
}

// src/Aligner.cpp:845: edlibAlign(path letters, read, NW, PATH). Writes the op string (0 match, 1 insert = query/path letter
// alone, 2 delete = target/read letter alone, 3 mismatch) and returns its length; -1 on an edlib error, -2 when `cap` is too
// small. *distance = editDistance. An alignment edlib failed to build (obtainAlignment's status is ignored by edlibAlign,
// edlib/src/edlib.cpp:270) comes back as length 0 with a valid distance, exactly as the reference sees it.
long long ref_edit_path(const char* a, uint64_t na, const char* b, uint64_t nb, unsigned char* ops, uint64_t cap, long long* distance)
{
	EdlibAlignResult r = edlibAlign(a, (int)na, b, (int)nb, edlibNewAlignConfig(-1, EDLIB_MODE_NW, EDLIB_TASK_PATH, NULL, 0));
	long long n = -1;
	if (r.status == EDLIB_STATUS_OK) {
		*distance = r.editDistance;
		n = r.alignmentLength;
		if ((uint64_t)n > cap) n = -2;
		else for (long long i = 0; i < n; i++) ops[i] = r.alignment[i];
	}
	edlibFreeAlignResult(r);
	return n;
}

// src/EValue.cpp through the constructor the aligner uses (src/Aligner.cpp:476,481): out = {alignment score, E-value}
void ref_evalue(double minIdentity, uint64_t databaseSize, uint64_t querySize, uint64_t alignmentLength, uint64_t numEdits, double* out)
{
	EValueCalculator calc(minIdentity);
	out[0] = calc.getAlignmentScore(alignmentLength, numEdits);
	out[1] = calc.getEValue(databaseSize, querySize, alignmentLength, numEdits);
}

} // extern "C"
