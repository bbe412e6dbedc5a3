// The correctness HMM's tables (host, libm), shared by the library and by the host-side test of the state-machine extension core.
#pragma once
#include "../hip/gc_device.hpp"
#include <cmath>
#include <vector>

// reference: src/AlignmentCorrectnessEstimation.cpp:6-70,72-78. libm is only used here, on the host; the
// device does +, max and >= on these doubles.
inline void buildCorrectnessTables(gcdev::CorrectnessTables& t)
{
	const double correctMean = 0.1875, correctStddev = 0.0955, wrongMean = 0.5, wrongStddev = 0.0291;
	const int wordSize = 64;
	t.f2c = log(0.00001);
	t.f2f = log(1.0 - 0.00001);
	t.c2f = log(0.0000000001);
	t.c2c = log(1.0 - 0.0000000001);
	auto fill = [&](double* out, double mean, double stddev) {
		std::vector<double> v;
		for (int i = 0; i <= wordSize / 2; i++) { double val = i; v.push_back(-(val - mean) * (val - mean) / (2 * stddev * stddev)); }
		double sum = 0;
		for (double x : v) sum += exp(x);
		double add = log(1.0 / sum);
		for (double& x : v) x += add;
		for (int i = wordSize / 2; i < wordSize; i++) v.push_back(v.back());
		for (int i = 0; i < 64; i++) out[i] = v[i];
	};
	fill(t.correctOdds, correctMean * wordSize, correctStddev * wordSize);
	fill(t.wrongOdds, wrongMean * wordSize, wrongStddev * wordSize);
	t.initCorrect = log(0.8);
	t.initFalse = log(0.2);
}

