// host_multi_own_driver.cpp -- TEST HARNESS: asif_amd/csrc/multi_own.hpp (the ownership rules of
// asif_hip_create_multi) instantiated with counting stand-ins for the filter handle and the stream, g++ only.
// Returns, per scenario, how often each stand-in handle / stream was released.
#include <stdint.h>
#include "multi_own.hpp"

// fail_create_at / fail_stream_at: index whose creation fails (-1: none).  destroyed[i], unmade[i]: release counts.
extern "C" int multi_own_scenario(int n, int fail_create_at, int fail_stream_at, int32_t *destroyed, int32_t *unmade,
                                  int32_t *left_handles, int32_t *left_streams)
{
	std::vector<int> hs, ss;
	const int r = asif::create_all(
	    n, hs, ss,
	    [&](int i, int *h) { if (i == fail_create_at) return 77; *h = i + 1; return 0; },
	    [&](int h) { destroyed[h - 1]++; },
	    [&](int h, int *s) { if (h - 1 == fail_stream_at) return 88; *s = 100 + h; return 0; },
	    [&](int h, int s) { if (s == 100 + h) unmade[h - 1]++; else unmade[h - 1] += 1000; });
	*left_handles = (int32_t)hs.size();
	*left_streams = (int32_t)ss.size();
	return r;
}
