// Probe: do packed-fp32 VALU instructions (v_pk_add/mul/fma_f32, as hipcc's SLP vectorizer emits them, including the
// op_sel forms) give wrong results while ANOTHER wave on the same SIMD runs v_mfma_f32_32x32x16_bf16?
// victim: per thread, two running sums over a stream of values, (a) written so that the SLP pass packs the pair into
// v_pk_* ops, (b) the same arithmetic kept scalar by empty asm barriers. Bitwise mismatches are counted.
// aggressor (second stream): a dense MFMA loop, bf16 32x32x16 or fp32 32x32x2.
// Build: hipcc --offload-arch=gfx950 -O3 -o pk_mfma_probe pk_mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void acc4(const f4u &q, const f4u &x, float mu, float &s1, float &s2) {
#pragma clang fp contract(off)
	s1 += (q[0] + q[1]) + (q[2] + q[3]);
	const float d0 = x[0] - mu, d1 = x[1] - mu, d2 = x[2] - mu, d3 = x[3] - mu;
	const float p01 = __builtin_fmaf(q[0], d0, q[1] * d1), p23 = __builtin_fmaf(q[2], d2, q[3] * d3);
	s2 += p01 + p23;
}
#define KEEP(v) asm volatile("" : "+v"(v))
__device__ __forceinline__ void acc4_scalar(const f4u &q, const f4u &x, float mu, float &s1, float &s2) {
#pragma clang fp contract(off)
	float t0 = q[0] + q[1]; KEEP(t0);
	float t1 = q[2] + q[3]; KEEP(t1);
	float t = t0 + t1; KEEP(t);
	s1 += t; KEEP(s1);
	float d0 = x[0] - mu; KEEP(d0);
	float d1 = x[1] - mu; KEEP(d1);
	float d2 = x[2] - mu; KEEP(d2);
	float d3 = x[3] - mu; KEEP(d3);
	float m1 = q[1] * d1; KEEP(m1);
	float p01 = __builtin_fmaf(q[0], d0, m1); KEEP(p01);
	float m3 = q[3] * d3; KEEP(m3);
	float p23 = __builtin_fmaf(q[2], d2, m3); KEEP(p23);
	float p = p01 + p23; KEEP(p);
	s2 += p; KEEP(s2);
}

__global__ void __launch_bounds__(256) victim(const float *__restrict__ g0, const float *__restrict__ g1, const float *__restrict__ xa,
                                              const float *__restrict__ xb, const float *__restrict__ ma, const float *__restrict__ mb,
                                              int per_block, unsigned *bad, float *out) {
	const int ch = blockIdx.x;
	const float mua = ma[ch], mub = mb[ch];
	float a1 = 0, a2 = 0, b1 = 0, b2 = 0, c1 = 0, c2 = 0, d1 = 0, d2 = 0;
	const size_t base = (size_t)ch * per_block;
	for (int i = threadIdx.x * 4; i < per_block; i += 1024) {
		const f4u v0 = *reinterpret_cast<const f4u *>(g0 + base + i), v1 = *reinterpret_cast<const f4u *>(g1 + base + i);
		const f4u va = *reinterpret_cast<const f4u *>(xa + base + i), vb = *reinterpret_cast<const f4u *>(xb + base + i);
		f4u q;
		for (int e = 0; e < 4; ++e) q[e] = (v0[e] + v1[e]) * (va[e] > 0.f ? 1.f : 0.f);
		acc4(q, va, mua, a1, a2);                // the SLP pass pairs these two calls lane by lane
		acc4(q, vb, mub, b1, b2);
		acc4_scalar(q, va, mua, c1, c2);
		acc4_scalar(q, vb, mub, d1, d2);
	}
	auto ne = [](float x, float y) { return __builtin_bit_cast(unsigned, x) != __builtin_bit_cast(unsigned, y); };
	if (ne(a1, c1) || ne(a2, c2) || ne(b1, d1) || ne(b2, d2)) atomicAdd(bad, 1u);
	out[(size_t)blockIdx.x * 256 + threadIdx.x] = a1 + a2 + b1 + b2;
}

__global__ void __launch_bounds__(256) mfma_busy(float *out, int iters, int use_bf16) {
	f32x16 acc[4];
	for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
	u16x8 a, b;
	for (int e = 0; e < 8; ++e) a[e] = (unsigned short)(0x3c00 + threadIdx.x + e), b[e] = (unsigned short)(0x3c10 + e);
	float fa = 1.0f + threadIdx.x * 1e-3f, fb = 0.5f;
	for (int it = 0; it < iters; ++it)
		for (int i = 0; i < 4; ++i) {
			if (use_bf16) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
			else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i], 0, 0, 0);
		}
	float s = 0.f;
	for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
	const int C = 8192, PB = 8192;
	float *g0, *g1, *xa, *xb, *ma, *mb, *out, *mout;
	unsigned *bad;
	const size_t N = (size_t)C * PB;
	hipMalloc(&g0, N * 4), hipMalloc(&g1, N * 4), hipMalloc(&xa, N * 4), hipMalloc(&xb, N * 4);
	hipMalloc(&ma, C * 4), hipMalloc(&mb, C * 4), hipMalloc(&out, (size_t)C * 256 * 4), hipMalloc(&bad, 4), hipMalloc(&mout, 1024 * 256 * 4);
	std::vector<float> h(N);
	for (size_t i = 0; i < N; ++i) h[i] = (float)(((i * 2654435761u) >> 7) & 0xffff) / 32768.f - 1.f;
	hipMemcpy(g0, h.data(), N * 4, hipMemcpyHostToDevice);
	for (size_t i = 0; i < N; ++i) h[i] = (float)(((i * 40503u + 7) >> 3) & 0xffff) / 32768.f - 1.f;
	hipMemcpy(g1, h.data(), N * 4, hipMemcpyHostToDevice), hipMemcpy(xb, h.data() + 1, (N - 1) * 4, hipMemcpyHostToDevice);
	for (size_t i = 0; i < N; ++i) h[i] = (float)(((i * 69069u + 3) >> 5) & 0xffff) / 32768.f - 1.f;
	hipMemcpy(xa, h.data(), N * 4, hipMemcpyHostToDevice);
	hipMemcpy(ma, h.data(), C * 4, hipMemcpyHostToDevice), hipMemcpy(mb, h.data() + C, C * 4, hipMemcpyHostToDevice);
	hipStream_t s2;
	hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
	const char *names[] = {"idle chip", "second stream: fp32 MFMA 32x32x2 loop", "second stream: bf16 MFMA 32x32x16 loop"};
	for (int mode = 0; mode < 3; ++mode) {
		hipMemset(bad, 0, 4);
		for (int it = 0; it < 10; ++it) {
			if (mode) mfma_busy<<<768, 256, 0, s2>>>(mout, 40000, mode == 2);
			victim<<<C, 256>>>(g0, g1, xa, xb, ma, mb, PB, bad, out);
		}
		hipDeviceSynchronize();
		unsigned hb = 0;
		hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
		printf("%-44s threads with packed != scalar sums: %u of %d x 10 launches\n", names[mode], hb, C * 256);
	}
	return 0;
}
