// Probe: what does a streaming kernel (read 1 tensor, write 1 tensor; and read 2 / write 1) reach on this HBM, as a
// function of bytes in flight per lane, workgroups per CU and cache policy? Sets the ceiling the batch-norm family is
// measured against.  hipcc --offload-arch=gfx950 -O3 stream_bw.hip -o stream_bw && ./stream_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ void __launch_bounds__(256) copy_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n4) {
	const size_t stride = (size_t)gridDim.x * 256;
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	for (; i + (U - 1) * stride < n4; i += U * stride) {
		f32x4 v[U];
#pragma unroll
		for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const f32x4 r = v[u] * 2.0f;
			if (NT) __builtin_nontemporal_store(r, dst + i + u * stride); else dst[i + u * stride] = r;
		}
	}
	for (; i < n4; i += stride) dst[i] = src[i] * 2.0f;
}

template <int U, bool NT>
__global__ void __launch_bounds__(256) add_kernel(const f32x4 *__restrict__ a, const f32x4 *__restrict__ b, f32x4 *__restrict__ dst, size_t n4) {
	const size_t stride = (size_t)gridDim.x * 256;
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	for (; i + (U - 1) * stride < n4; i += U * stride) {
		f32x4 v[U], w[U];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			v[u] = NT ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
			w[u] = NT ? __builtin_nontemporal_load(b + i + u * stride) : b[i + u * stride];
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const f32x4 r = v[u] + w[u];
			if (NT) __builtin_nontemporal_store(r, dst + i + u * stride); else dst[i + u * stride] = r;
		}
	}
	for (; i < n4; i += stride) dst[i] = a[i] + b[i];
}

template <typename F>
float timed(F f) {
	hipEvent_t s, e;
	hipEventCreate(&s), hipEventCreate(&e);
	f();
	hipDeviceSynchronize();
	hipEventRecord(s);
	for (int r = 0; r < 10; ++r) f();
	hipEventRecord(e);
	hipEventSynchronize(e);
	float ms;
	hipEventElapsedTime(&ms, s, e);
	return ms / 10;
}

int main() {
	const size_t n4 = (size_t)1 << 26;       // 1 GiB per tensor
	f32x4 *a, *b, *c;
	hipMalloc(&a, n4 * 16), hipMalloc(&b, n4 * 16), hipMalloc(&c, n4 * 16);
	hipMemset(a, 0, n4 * 16), hipMemset(b, 0, n4 * 16);
	const int grids[] = {256 * 4, 256 * 8, 256 * 16, 256 * 32, (int)(n4 / 256)};
	for (int g : grids) {
		float t;
#define RUN(NAME, K, BYTES) t = timed([&] { K; }); printf("%-26s grid %8d: %.3f ms  %.2f TB/s\n", NAME, g, t, BYTES * n4 * 16.0 / t / 1e9);
		RUN("copy U1", (copy_kernel<1, false><<<g, 256>>>(a, c, n4)), 2)
		RUN("copy U4", (copy_kernel<4, false><<<g, 256>>>(a, c, n4)), 2)
		RUN("copy U8", (copy_kernel<8, false><<<g, 256>>>(a, c, n4)), 2)
		RUN("copy U4 nontemporal", (copy_kernel<4, true><<<g, 256>>>(a, c, n4)), 2)
		RUN("add  U2", (add_kernel<2, false><<<g, 256>>>(a, b, c, n4)), 3)
		RUN("add  U4", (add_kernel<4, false><<<g, 256>>>(a, b, c, n4)), 3)
		RUN("add  U4 nontemporal", (add_kernel<4, true><<<g, 256>>>(a, b, c, n4)), 3)
	}
	return 0;
}
