// Start-up stage timing on stderr (GC_DEBUG_TIMES): milliseconds per stage and the process's resident memory when the stage ends
// (what decides how large a graph a host can build: DESIGN.md §10, config 5).
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace gc {

inline double residentGiB()   // VmRSS of /proc/self/status; 0 when unreadable
{
	FILE* f = fopen("/proc/self/status", "r");
	if (!f) return 0;
	char line[256];
	double kb = 0;
	while (fgets(line, sizeof line, f)) if (strncmp(line, "VmRSS:", 6) == 0) { kb = atof(line + 6); break; }
	fclose(f);
	return kb / (1024.0 * 1024.0);
}

struct StageClock {
	const char* tag;
	std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
	const bool on = getenv("GC_DEBUG_TIMES") != nullptr;
	explicit StageClock(const char* tag_ = "gc build") : tag(tag_) {}
	void lap(const char* what)
	{
		auto now = std::chrono::steady_clock::now();
		if (on) fprintf(stderr, "[%s] %-28s %8.1f ms  %7.2f GiB resident\n", tag, what, std::chrono::duration<double, std::milli>(now - t).count(), residentGiB());
		t = now;
	}
};

} // namespace gc
