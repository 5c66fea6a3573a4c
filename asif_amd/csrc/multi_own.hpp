// multi_own.hpp -- who releases what when a multi-device handle cannot be completed (asif_hip_create_multi).
// The filter handles belong to the CALLER of adopt_with_streams until it returns success; on failure it releases the
// streams it made itself and nothing else, so every handle is destroyed exactly once, by its creator.
// Plain templates over the handle / stream types: tests/host_multi_own_driver.cpp instantiates them with counting
// stand-ins on the CPU (round 2 destroyed the handles twice on this path).
#pragma once
#include <stddef.h>
#include <vector>

namespace asif {

// make(handle, &stream) -> 0 or an error code; unmake(handle, stream) releases one stream made by make.
template <class H, class S, class Make, class Unmake>
int make_streams(const std::vector<H> &handles, std::vector<S> &streams, Make make, Unmake unmake)
{
	streams.assign(handles.size(), S());
	for (size_t i = 0; i < handles.size(); i++) {
		const int e = make(handles[i], &streams[i]);
		if (e) {
			for (size_t k = 0; k < i; k++) unmake(handles[k], streams[k]);
			streams.clear();
			return e;
		}
	}
	return 0;
}

// create(i, &handle) for i = 0 .. n-1, then the streams; on any failure everything made so far is released once.
template <class H, class S, class Create, class Destroy, class Make, class Unmake>
int create_all(int n, std::vector<H> &handles, std::vector<S> &streams, Create create, Destroy destroy, Make make,
               Unmake unmake)
{
	handles.clear();
	int e = 0;
	for (int i = 0; i < n && !e; i++) {
		H h = H();
		e = create(i, &h);
		if (!e) handles.push_back(h);
	}
	if (!e) e = make_streams(handles, streams, make, unmake);
	if (e) {
		for (H h : handles) destroy(h);
		handles.clear();
	}
	return e;
}

} // namespace asif
