// Counter-based RNG (Philox4x32-10): fillInteger / fillUniform / fillNormal for Dropout masks and parameter init.
// Replaces RandomNumberGenerator — Cuda/Source/Libs/CuRand.c:231-234 (XORWOW there). Statistical parity only: the
// reference's stream is seeded from numpy (Cuda/GPUBackend.py:62-63) and is not reproducible across backends either.
// Stateless kernels: word i of call #k is philox(key = seed, counter = {i/4, k}), so fills never need a state array
// in HBM and every launch is a pure streaming write.
#include "common.h"

struct pz_rng {
	uint64_t seed;
	uint64_t calls;
};

namespace {

__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
	constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
	for (int r = 0; r < 10; ++r) {
		const uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
		const uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
		ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
		key.x += W0;
		key.y += W1;
	}
	return ctr;
}

__device__ __forceinline__ float u01(uint32_t w) { return ((float)(w >> 8) + 1.0f) * (1.0f / 16777216.0f); }   // (0, 1]

// MODE 0: raw words, 1: uniform (0,1], 2: normal(mean, std) via Box-Muller
template <int MODE>
__global__ void __launch_bounds__(256) rng_fill_kernel(uint32_t *__restrict__ out, size_t count, uint64_t seed, uint64_t call,
                                                        float mean, float stddev) {
	const size_t nquads = (count + 3) >> 2;
	for (size_t qd = (size_t)blockIdx.x * blockDim.x + threadIdx.x; qd < nquads; qd += (size_t)gridDim.x * blockDim.x) {
		const uint4 r = philox4x32_10(make_uint4((uint32_t)qd, (uint32_t)(qd >> 32), (uint32_t)call, (uint32_t)(call >> 32)),
		                              make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
		uint32_t w[4] = {r.x, r.y, r.z, r.w};

		if (MODE == 1) {
#pragma unroll
			for (int k = 0; k < 4; ++k) w[k] = __float_as_uint(u01(w[k]));
		} else if (MODE == 2) {
#pragma unroll
			for (int k = 0; k < 4; k += 2) {
				const float rad = sqrtf(-2.f * logf(u01(w[k]))), ang = 6.283185307179586f * u01(w[k + 1]);
				w[k] = __float_as_uint(mean + stddev * rad * cosf(ang));
				w[k + 1] = __float_as_uint(mean + stddev * rad * sinf(ang));
			}
		}

		const size_t base = qd << 2;
		if (base + 4 <= count && (((uintptr_t)out & 15) == 0)) {
			reinterpret_cast<uint4 *>(out)[qd] = make_uint4(w[0], w[1], w[2], w[3]);
		} else {
#pragma unroll
			for (int k = 0; k < 4; ++k)
				if (base + k < count) out[base + k] = w[k];
		}
	}
}

template <int MODE>
int fill(pz_rng_t rng, void *out, size_t count, float mean, float stddev, pz_stream_t stream) {
	PZ_REQUIRE(rng != nullptr, "rng: null generator");
	if (count == 0) return PZ_OK;
	PZ_REQUIRE(out != nullptr, "rng: null output");
	rng_fill_kernel<MODE><<<pz::stream_grid((count + 3) / 4, 256), 256, 0, pz::as_stream(stream)>>>(
	    (uint32_t *)out, count, rng->seed, rng->calls++, mean, stddev);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

}  // namespace

extern "C" {

int pz_rng_create(uint64_t seed, pz_rng_t *rng) {
	PZ_REQUIRE(rng != nullptr, "pz_rng_create: null output");
	*rng = new pz_rng{seed, 0};
	return PZ_OK;
}

int pz_rng_destroy(pz_rng_t rng) {
	delete rng;
	return PZ_OK;
}

int pz_rng_fill_u32(pz_rng_t rng, uint32_t *out, size_t count, pz_stream_t stream) { return fill<0>(rng, out, count, 0.f, 1.f, stream); }

int pz_rng_fill_uniform(pz_rng_t rng, float *out, size_t count, pz_stream_t stream) { return fill<1>(rng, out, count, 0.f, 1.f, stream); }

int pz_rng_fill_normal(pz_rng_t rng, float *out, size_t count, float mean, float stddev, pz_stream_t stream) {
	return fill<2>(rng, out, count, mean, stddev, stream);
}

}  // extern "C"
