// internal.h -- pieces shared by the translation units behind the C ABI (urmapx.hip, text_gpu.hip).  Not installed.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>

#include "../../include/urmapx.h"

namespace urx {

inline int hip_rc(hipError_t e) {
	if (e == hipSuccess) return URMAPX_OK;
	if (e == hipErrorOutOfMemory) return URMAPX_E_NOMEM;
	return URMAPX_E_NODEVICE;
}
#define HIP_TRY(x)                                    \
	do {                                              \
		hipError_t e_ = (x);                          \
		if (e_ != hipSuccess) return urx::hip_rc(e_); \
	} while (0)

// what allocation calls cost this process (urmapx_map_report.alloc_*): [0] hipMalloc / hipFree of device arrays, [1] hipHostMalloc /
// hipHostFree of page-locked chunk buffers; nanoseconds and calls.  Defined in urmapx.hip.
struct AllocClock {
	std::atomic<uint64_t> ns[2], calls[2];
};
AllocClock &alloc_clock();
struct AllocTimer {
	int k;
	std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
	explicit AllocTimer(int kind) : k(kind) {}
	~AllocTimer() {
		AllocClock &c = alloc_clock();
		c.ns[k].fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(), std::memory_order_relaxed);
		c.calls[k].fetch_add(1, std::memory_order_relaxed);
	}
};

// device array that only grows
template <class T>
struct DevBuf {
	T *p = nullptr;
	size_t cap = 0;
	int ensure(size_t n) {
		if (n <= cap) return URMAPX_OK;
		AllocTimer at(0);
		if (p) (void)hipFree(p);
		p = nullptr; cap = 0;
		size_t want = n + n / 4 + 64;
		hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
		if (e != hipSuccess) { p = nullptr; return hip_rc(e); }
		cap = want;
		return URMAPX_OK;
	}
	void release() {
		if (p) { AllocTimer at(0); (void)hipFree(p); }
		p = nullptr; cap = 0;
	}
};

// kernel classes of the pair kernel (pairs themselves: <= 279 bases per mate, flagged per read)
constexpr uint32_t MAX_QL_PE = 320;

// mapping contexts kept between urmapx_map_files calls (pipeline.cpp): those of an index are destroyed before the index is
void lane_pool_purge(const urmapx_index *I);  // nullptr: all

// what text_gpu.hip needs of a mapping context (defined in urmapx.hip)
hipStream_t ctx_stream(urmapx_ctx *);
int ctx_device(const urmapx_ctx *);
const urmapx_index *ctx_index(const urmapx_ctx *);

}  // namespace urx
