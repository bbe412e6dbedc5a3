// ORACLE (test infrastructure, not product code): the Karlin-Altschul E-value behind --E-cutoff. Restates
// src/EValue.cpp:16-105 (EValueCalculator) and src/AlignmentSelection.cpp:91-99 (SelectECutoff); pinned bit for bit against
// the reference's own EValue.cpp compiled into oracle/_ref (tests/test_oracle_units.py). Same libm, same operation order.
#pragma once
#include <cmath>
#include <cstddef>
#include <vector>

namespace oracle {

class EValueCalc {
public:
	EValueCalc() = default;
	explicit EValueCalc(double minIdentity) : matchScore(1), mismatchScore(-minIdentity / (1.0 - minIdentity))   // :25-33
	{
		// lambda: root of (e^(l*match) + e^(l*mismatch)) / 2 = 1 in (0, 0.7), 100 bisection steps (:52-76)
		double lo = 0, hi = 0.7;
		for (int i = 0; i < 100; i++) {
			double mid = (lo + hi) * 0.5;
			double value = pow(E, mid * matchScore) * .5 + pow(E, mid * mismatchScore) * 0.5 - 1;
			if (value < 0) lo = mid;
			if (value > 0) hi = mid;
			if (value == 0) { lo = mid; hi = mid; break; }
			if (lo == hi) break;
		}
		lambda = (lo + hi) / 2;
		// K from the first nine terms of the ladder-point series over binomial match/mismatch counts (:78-105)
		double seriesSum = 0;
		std::vector<size_t> row { 1 };
		for (int k = 1; k < 10; k++) {
			std::vector<size_t> next(row.size() + 1, 0);
			for (size_t j = 0; j < row.size(); j++) { next[j] += row[j]; next[j + 1] += row[j]; }
			row = next;
			size_t total = 0;
			for (size_t n : row) total += n;
			double negativeExpectation = 0, greaterProbability = 0;
			for (size_t j = 0; j < row.size(); j++) {
				size_t matches = j, mismatches = row.size() - 1 - j;
				double score = (double)matches * matchScore + (double)mismatches * mismatchScore;
				double probability = (double)row[j] / (double)total;
				if (score < 0) negativeExpectation += pow(E, lambda * score) * probability;
				if (score >= 0) greaterProbability += probability;
			}
			seriesSum += (negativeExpectation + greaterProbability) / (double)k;
		}
		double expectation = .5 * matchScore * pow(E, lambda * matchScore) + .5 * mismatchScore * pow(E, lambda * mismatchScore);
		double Cstar = pow(E, -2 * seriesSum) / (lambda * expectation);
		K = Cstar * lambda / (1.0 - pow(E, -lambda));
	}
	double getAlignmentScore(size_t alignmentLength, size_t numEdits) const { return alignmentLength * matchScore - numEdits * (mismatchScore - matchScore); }   // :45-48 (sic: minus a negative difference)
	double getEValue(size_t databaseSize, size_t querySize, size_t alignmentLength, size_t numEdits) const   // :35-43
	{
		return K * databaseSize * querySize * pow(E, -lambda * getAlignmentScore(alignmentLength, numEdits));
	}
private:
	static constexpr double E = 2.71828182845904523536028747135266249775724709369995;
	double matchScore = -1, mismatchScore = -1, lambda = -1, K = -1;
};

} // namespace oracle
