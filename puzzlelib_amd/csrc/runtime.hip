// Device runtime of libpuzzle_mi355: device selection, size-class pool allocator, copies, streams, events.
// Native counterpart of the reference "Driver" C extension (Cuda/Source/Core/{Device,Allocator,Buffer,Stream}.c),
// designed for one process per GPU and 288 GB of HBM: blocks are never split or coalesced, a released block is
// parked in its size class and reused in stream order by the next request of that class.
#include "common.h"
#include <chrono>
#include <thread>

#include <mutex>
#include <unordered_map>
#include <vector>

namespace pz {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

}  // namespace pz

extern "C" {

int pz_version(void) { return 100; }

const char *pz_last_error(void) { return pz::g_err; }

int pz_device_count(int *count) {
	PZ_REQUIRE(count != nullptr, "pz_device_count: null output");
	hipError_t e = hipGetDeviceCount(count);
	if (e != hipSuccess) {
		*count = 0;
		(void)hipGetLastError();
	}
	return PZ_OK;
}

int pz_init(int device) {
	int n = 0;
	pz_device_count(&n);
	PZ_REQUIRE(n > 0, "pz_init: no HIP device visible");
	PZ_REQUIRE(device >= 0 && device < n, "pz_init: device index %d out of range (0..%d)", device, n - 1);
	PZ_HIP(hipSetDevice(device));

	hipDeviceProp_t prop;
	PZ_HIP(hipGetDeviceProperties(&prop, device));
	PZ_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0,
	           "pz_init: device %d is %s; this library ships gfx950 (MI355X) code objects only", device,
	           prop.gcnArchName);
	return PZ_OK;
}

int pz_device_name(int device, char *buf, int buflen) {
	hipDeviceProp_t prop;
	PZ_HIP(hipGetDeviceProperties(&prop, device));
	if (prop.name[0])
		snprintf(buf, buflen, "%s", prop.name);
	else
		snprintf(buf, buflen, "AMD Instinct (%s, %d CUs)", prop.gcnArchName, prop.multiProcessorCount);
	return PZ_OK;
}

int pz_device_arch(int device, char *buf, int buflen) {
	hipDeviceProp_t prop;
	PZ_HIP(hipGetDeviceProperties(&prop, device));
	snprintf(buf, buflen, "%s", prop.gcnArchName);
	return PZ_OK;
}

int pz_device_num_cus(int device, int *cus) {
	hipDeviceProp_t prop;
	PZ_HIP(hipGetDeviceProperties(&prop, device));
	*cus = prop.multiProcessorCount;
	return PZ_OK;
}

int pz_device_sync(void) {
	PZ_HIP(hipDeviceSynchronize());
	return PZ_OK;
}

int pz_device_mem_info(size_t *free_bytes, size_t *total_bytes) {
	PZ_HIP(hipMemGetInfo(free_bytes, total_bytes));
	return PZ_OK;
}

// ---------------------------------------------------------------------------------------------- memory
int pz_malloc(void **ptr, size_t nbytes) {
	PZ_REQUIRE(ptr != nullptr, "pz_malloc: null output");
	*ptr = nullptr;
	if (nbytes == 0) return PZ_OK;
	PZ_HIP(hipMalloc(ptr, nbytes));
	return PZ_OK;
}

int pz_free(void *ptr) {
	if (ptr) PZ_HIP(hipFree(ptr));
	return PZ_OK;
}

int pz_host_alloc_pinned(void **h_ptr, size_t nbytes) {
	PZ_HIP(hipHostMalloc(h_ptr, nbytes, hipHostMallocDefault));
	return PZ_OK;
}

int pz_host_free_pinned(void *h_ptr) {
	if (h_ptr) PZ_HIP(hipHostFree(h_ptr));
	return PZ_OK;
}

}  // extern "C"

// Size classes: powers of two subdivided by 4 (2 mantissa bits, like the reference's Allocator.c:51-66, so the
// round-up waste is <= 25 %), minimum 256 B so every block is 256-B aligned relative to hipMalloc's base.
static size_t pool_class_size(size_t n) {
	const size_t kMin = 256;
	if (n <= kMin) return kMin;
	const int msb = 63 - __builtin_clzll((unsigned long long)(n - 1));   // n-1 in [2^msb, 2^(msb+1))
	size_t step = (size_t)1 << (msb >= 2 ? msb - 2 : 0);                 // 4 classes per octave
	if (step < kMin) step = kMin;
	return (n + step - 1) / step * step;
}

struct pz_pool {
	std::mutex mu;
	std::unordered_map<size_t, std::vector<void *>> held;   // class size -> parked blocks
	std::unordered_map<void *, size_t> live;                // block -> class size
	size_t held_bytes = 0, live_bytes = 0, n_held = 0;
};

extern "C" {

int pz_pool_create(pz_pool_t *pool) {
	PZ_REQUIRE(pool != nullptr, "pz_pool_create: null output");
	*pool = new pz_pool();
	return PZ_OK;
}

int pz_pool_free_held(pz_pool_t pool) {
	PZ_REQUIRE(pool != nullptr, "pz_pool_free_held: null pool");
	std::lock_guard<std::mutex> lock(pool->mu);
	// parked blocks may still be referenced by queued kernels: drain the device before returning them to the driver
	PZ_HIP(hipDeviceSynchronize());
	for (auto &kv : pool->held)
		for (void *p : kv.second) (void)hipFree(p);
	pool->held.clear();
	pool->held_bytes = 0;
	pool->n_held = 0;
	return PZ_OK;
}

int pz_pool_destroy(pz_pool_t pool) {
	if (!pool) return PZ_OK;
	pz_pool_free_held(pool);
	{
		std::lock_guard<std::mutex> lock(pool->mu);
		for (auto &kv : pool->live) (void)hipFree(kv.first);
	}
	delete pool;
	return PZ_OK;
}

static long g_pool_oom_events = 0;
static long g_pool_driver_allocs = 0;          // pool misses: blocks fetched from the driver
static double g_pool_driver_seconds = 0.0;     // host time spent inside those hipMalloc calls

int pz_pool_driver_allocs(long *count, double *seconds) {
	if (count) *count = g_pool_driver_allocs;
	if (seconds) *seconds = g_pool_driver_seconds;
	return PZ_OK;
}

int pz_pool_oom_events(long *count) {
	PZ_REQUIRE(count != nullptr, "pz_pool_oom_events: null output");
	*count = g_pool_oom_events;
	return PZ_OK;
}

int pz_pool_alloc(pz_pool_t pool, size_t nbytes, void **ptr) {
	PZ_REQUIRE(pool != nullptr && ptr != nullptr, "pz_pool_alloc: null argument");
	*ptr = nullptr;
	if (nbytes == 0) return PZ_OK;

	const size_t cls = pool_class_size(nbytes);
	std::lock_guard<std::mutex> lock(pool->mu);

	auto it = pool->held.find(cls);
	if (it != pool->held.end() && !it->second.empty()) {
		*ptr = it->second.back();
		it->second.pop_back();
		pool->held_bytes -= cls;
		pool->n_held -= 1;
	} else {
		const auto t0 = std::chrono::steady_clock::now();
		hipError_t e = hipMalloc(ptr, cls);
		g_pool_driver_allocs += 1;
		g_pool_driver_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		if (e == hipErrorOutOfMemory) {
			g_pool_oom_events += 1;
			(void)hipGetLastError();
			(void)hipDeviceSynchronize();
			// The device may be short of memory only for a moment: the driver reclaims the memory of a process that has
			// just exited asynchronously (a bench started right behind another 70 GB process sees "out of memory" for a
			// second or two). Wait for that first — dropping the parked blocks would make every following step re-allocate
			// them, each time through this path (measured: 5x slower steps until the reclaim is over).
			for (int attempt = 0; attempt < 40 && e == hipErrorOutOfMemory; ++attempt) {
				std::this_thread::sleep_for(std::chrono::milliseconds(50));
				e = hipMalloc(ptr, cls);
				if (e == hipErrorOutOfMemory) (void)hipGetLastError();
			}
		}
		if (e == hipErrorOutOfMemory) {
			// really full: give parked memory back to the driver once, then retry
			for (auto &kv : pool->held)
				for (void *p : kv.second) (void)hipFree(p);
			pool->held.clear();
			pool->held_bytes = 0;
			pool->n_held = 0;
			e = hipMalloc(ptr, cls);
		}
		if (e != hipSuccess) {
			pz::set_error("pz_pool_alloc: hipMalloc(%zu) failed: %s", cls, hipGetErrorString(e));
			*ptr = nullptr;
			return e == hipErrorOutOfMemory ? PZ_ERR_NOMEM : PZ_ERR_HIP;
		}
	}

	pool->live[*ptr] = cls;
	pool->live_bytes += cls;
	return PZ_OK;
}

int pz_pool_release(pz_pool_t pool, void *ptr) {
	if (!ptr) return PZ_OK;
	PZ_REQUIRE(pool != nullptr, "pz_pool_release: null pool");
	std::lock_guard<std::mutex> lock(pool->mu);

	auto it = pool->live.find(ptr);
	PZ_REQUIRE(it != pool->live.end(), "pz_pool_release: pointer %p does not belong to this pool", ptr);

	const size_t cls = it->second;
	pool->live.erase(it);
	pool->live_bytes -= cls;
	pool->held[cls].push_back(ptr);
	pool->held_bytes += cls;
	pool->n_held += 1;
	return PZ_OK;
}

int pz_pool_stats(pz_pool_t pool, size_t *held_bytes, size_t *live_bytes, size_t *n_held, size_t *n_live) {
	PZ_REQUIRE(pool != nullptr, "pz_pool_stats: null pool");
	std::lock_guard<std::mutex> lock(pool->mu);
	if (held_bytes) *held_bytes = pool->held_bytes;
	if (live_bytes) *live_bytes = pool->live_bytes;
	if (n_held) *n_held = pool->n_held;
	if (n_live) *n_live = pool->live.size();
	return PZ_OK;
}

// ---------------------------------------------------------------------------------------------- copies
int pz_memcpy_h2d(void *dst, const void *h_src, size_t nbytes, pz_stream_t stream) {
	if (nbytes == 0) return PZ_OK;
	PZ_HIP(hipMemcpyAsync(dst, h_src, nbytes, hipMemcpyHostToDevice, pz::as_stream(stream)));
	return PZ_OK;
}

int pz_memcpy_d2h(void *h_dst, const void *src, size_t nbytes, pz_stream_t stream) {
	if (nbytes == 0) return PZ_OK;
	PZ_HIP(hipMemcpyAsync(h_dst, src, nbytes, hipMemcpyDeviceToHost, pz::as_stream(stream)));
	return PZ_OK;
}

int pz_memcpy_d2d(void *dst, const void *src, size_t nbytes, pz_stream_t stream) {
	if (nbytes == 0) return PZ_OK;
	PZ_HIP(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, pz::as_stream(stream)));
	return PZ_OK;
}

int pz_memcpy_2d(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width_bytes, size_t height,
                 pz_stream_t stream) {
	if (width_bytes == 0 || height == 0) return PZ_OK;
	PZ_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, height, hipMemcpyDeviceToDevice,
	                        pz::as_stream(stream)));
	return PZ_OK;
}

}  // extern "C"

// 32-bit pattern fill: 16 B per lane per iteration, grid-stride
__global__ void __launch_bounds__(256) fill_d32_kernel(uint32_t *__restrict__ dst, uint32_t value, size_t count) {
	const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t nthreads = (size_t)gridDim.x * blockDim.x;

	// peel to 16-B alignment
	const size_t mis = ((uintptr_t)dst >> 2) & 3;
	const size_t head = mis ? (4 - mis < count ? 4 - mis : count) : 0;
	if (tid < head) dst[tid] = value;

	uint4 *d4 = reinterpret_cast<uint4 *>(dst + head);
	const size_t n4 = (count - head) >> 2;
	const uint4 v4 = make_uint4(value, value, value, value);
	for (size_t i = tid; i < n4; i += nthreads) d4[i] = v4;

	const size_t tail0 = head + (n4 << 2);
	if (tid < count - tail0) dst[tail0 + tid] = value;
}

struct StridedCopyArgs {
	int64_t shape[6], ds[6], ss[6];
	int ndim;
};

__global__ void __launch_bounds__(256) strided_copy_kernel(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src,
                                                            StridedCopyArgs a, size_t total) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
		size_t rem = i;
		int64_t so = 0, dof = 0;
#pragma unroll
		for (int d = 5; d >= 0; --d) {
			if (d < a.ndim) {
				const int64_t idx = (int64_t)(rem % (size_t)a.shape[d]);
				rem /= (size_t)a.shape[d];
				so += idx * a.ss[d];
				dof += idx * a.ds[d];
			}
		}
		dst[dof] = src[so];
	}
}

extern "C" {

int pz_memset_d32(void *dst, uint32_t value, size_t count, pz_stream_t stream) {
	if (count == 0) return PZ_OK;
	PZ_REQUIRE(((uintptr_t)dst & 3) == 0, "pz_memset_d32: destination not 4-byte aligned");
	const int grid = pz::stream_grid(count / 4 + 1, 256);
	fill_d32_kernel<<<grid, 256, 0, pz::as_stream(stream)>>>((uint32_t *)dst, value, count);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_strided_copy(void *dst, const int64_t *dst_strides, const void *src, const int64_t *src_strides,
                    const int64_t *shape, int ndim, pz_stream_t stream) {
	PZ_REQUIRE(ndim >= 0 && ndim <= 6, "pz_strided_copy: ndim %d > 6", ndim);
	StridedCopyArgs a;
	size_t total = 1;
	for (int d = 0; d < 6; ++d) {
		a.shape[d] = d < ndim ? shape[d] : 1;
		a.ds[d] = d < ndim ? dst_strides[d] : 0;   // strides in ELEMENTS
		a.ss[d] = d < ndim ? src_strides[d] : 0;
		total *= (size_t)a.shape[d];
	}
	a.ndim = ndim;
	if (total == 0) return PZ_OK;
	strided_copy_kernel<<<pz::stream_grid(total, 256), 256, 0, pz::as_stream(stream)>>>(
	    (uint32_t *)dst, (const uint32_t *)src, a, total);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

// ---------------------------------------------------------------------------------------------- streams / events
int pz_stream_create(pz_stream_t *stream) {
	hipStream_t s;
	PZ_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	*stream = s;
	return PZ_OK;
}

int pz_stream_create_priority(pz_stream_t *stream, int level) {
	// level < 0: the device's lowest priority, > 0: its highest, 0: as pz_stream_create
	int least = 0, greatest = 0;
	PZ_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
	hipStream_t s;
	PZ_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, level < 0 ? least : (level > 0 ? greatest : (least + greatest) / 2)));
	*stream = s;
	return PZ_OK;
}

int pz_stream_destroy(pz_stream_t stream) {
	if (stream) PZ_HIP(hipStreamDestroy(pz::as_stream(stream)));
	return PZ_OK;
}

int pz_stream_sync(pz_stream_t stream) {
	PZ_HIP(hipStreamSynchronize(pz::as_stream(stream)));
	return PZ_OK;
}

int pz_stream_wait_event(pz_stream_t stream, pz_event_t event) {
	PZ_HIP(hipStreamWaitEvent(pz::as_stream(stream), (hipEvent_t)event, 0));
	return PZ_OK;
}

int pz_event_create(pz_event_t *event) {
	hipEvent_t e;
	PZ_HIP(hipEventCreate(&e));
	*event = e;
	return PZ_OK;
}

int pz_event_destroy(pz_event_t event) {
	if (event) PZ_HIP(hipEventDestroy((hipEvent_t)event));
	return PZ_OK;
}

int pz_event_record(pz_event_t event, pz_stream_t stream) {
	PZ_HIP(hipEventRecord((hipEvent_t)event, pz::as_stream(stream)));
	return PZ_OK;
}

int pz_event_sync(pz_event_t event) {
	PZ_HIP(hipEventSynchronize((hipEvent_t)event));
	return PZ_OK;
}

int pz_event_query(pz_event_t event, int *done) {
	PZ_REQUIRE(event && done, "pz_event_query: null argument");
	const hipError_t rc = hipEventQuery((hipEvent_t)event);
	if (rc != hipSuccess && rc != hipErrorNotReady) PZ_HIP(rc);
	*done = rc == hipSuccess;
	return PZ_OK;
}

int pz_event_elapsed_ms(pz_event_t start, pz_event_t end, float *ms) {
	PZ_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)end));
	return PZ_OK;
}

}  // extern "C"
