// Probe: fp32 products on the bf16 matrix pipe. An fp32 number is EXACTLY the sum of three bf16 numbers (24 significand
// bits = 8 + 8 + 8, truncation split), every bf16 x bf16 product is exact in fp32, and the MFMA accumulates in fp32 — so
// a*b = sum of 9 exact partial products. This probe measures (1) the sustained rate of v_mfma_f32_32x32x16_bf16 next to
// v_mfma_f32_32x32x2_f32, (2) how the bf16 MFMA rounds its 16-term sum, and (3) the error of a K-long dot product computed
// with 9 / 6 / 3 partial products against the fp32 MFMA and a float64 host reference.
// Build: hipcc --offload-arch=gfx950 -O3 -o split_probe split_probe.hip ; run: ./split_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------------------------------------- (1) rate
template <int ILP>
__global__ void __launch_bounds__(256) bf16_loop(float *out, int iters) {
	f32x16 acc[ILP];
	for (int i = 0; i < ILP; ++i)
		for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
	u16x8 a[4], b[4];
	unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
	for (int k = 0; k < 4; ++k)
		for (int e = 0; e < 8; ++e) {
			h = h * 1664525u + 1013904223u;
			a[k][e] = (unsigned short)(0x3c00u + ((h >> 9) & 0x3ffu) + ((h >> 31) << 15));
			h = h * 1664525u + 1013904223u;
			b[k][e] = (unsigned short)(0x3c00u + ((h >> 9) & 0x3ffu) + ((h >> 31) << 15));
		}
	for (int it = 0; it < iters / 4; ++it) {
#pragma unroll
		for (int k = 0; k < 4; ++k)
#pragma unroll
			for (int i = 0; i < ILP; ++i)
				acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[k]), __builtin_bit_cast(bf16x8, b[k]), acc[i], 0, 0, 0);
	}
	float s = 0.f;
	for (int i = 0; i < ILP; ++i)
		for (int r = 0; r < 16; ++r) s += acc[i][r];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the shape of a split main loop: per k16 step a wave reads 12 fragments (2 x 3 for each operand) with ds_read_b128 and
// issues 36 MFMAs (2 x 2 output tiles x 9 partial products)
template <int NPROD>
__global__ void __launch_bounds__(256) split_shape_loop(float *out, int iters) {
	__shared__ u16x8 lds[4][12][64];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
	for (int f = 0; f < 12; ++f) {
		u16x8 v;
		for (int e = 0; e < 8; ++e) {
			h = h * 1664525u + 1013904223u;
			v[e] = (unsigned short)(0x3c00u + ((h >> 9) & 0x3ffu) + ((h >> 31) << 15));
		}
		lds[wave][f][lane] = v;
	}
	__syncthreads();
	f32x16 acc[4];
	for (int i = 0; i < 4; ++i)
		for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
	for (int it = 0; it < iters; ++it) {
		bf16x8 fa[2][3], fb[2][3];
#pragma unroll
		for (int m = 0; m < 2; ++m)
#pragma unroll
			for (int s = 0; s < 3; ++s) {
				fa[m][s] = __builtin_bit_cast(bf16x8, lds[wave][m * 3 + s][(lane + it) & 63]);
				fb[m][s] = __builtin_bit_cast(bf16x8, lds[wave][6 + m * 3 + s][(lane + it) & 63]);
			}
#pragma unroll
		for (int sa = 2; sa >= 0; --sa)
#pragma unroll
			for (int sb = 2; sb >= 0; --sb) {
				if (NPROD == 6 && sa + sb > 2) continue;
				if (NPROD == 3 && sa + sb > 1) continue;
#pragma unroll
				for (int m = 0; m < 2; ++m)
#pragma unroll
					for (int n = 0; n < 2; ++n)
						acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m][sa], fb[n][sb], acc[m * 2 + n], 0, 0, 0);
			}
	}
	float s = 0.f;
	for (int i = 0; i < 4; ++i)
		for (int r = 0; r < 16; ++r) s += acc[i][r];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---------------------------------------------------------------------------------------------- (2) one MFMA's rounding
// D = A(32x16) * B(16x32) + C with A row i / B column j taken from the same 16-vectors: every (i, j) is an independent case
__global__ void one_mfma(const unsigned short *a, const unsigned short *b, const float *c, float *d) {
	const int lane = threadIdx.x;
	u16x8 fa, fb;
	for (int e = 0; e < 8; ++e) {
		fa[e] = a[(lane & 31) * 16 + (lane >> 5) * 8 + e];          // A[i = lane%32][k = 8*(lane/32) + e]
		fb[e] = b[(lane & 31) * 16 + (lane >> 5) * 8 + e];          // B[k][j = lane%32]
	}
	f32x16 acc;
	for (int r = 0; r < 16; ++r) acc[r] = c[(8 * (r / 4) + 4 * (lane >> 5) + (r & 3)) * 32 + (lane & 31)];
	acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), acc, 0, 0, 0);
	for (int r = 0; r < 16; ++r) d[(8 * (r / 4) + 4 * (lane >> 5) + (r & 3)) * 32 + (lane & 31)] = acc[r];
}

// ---------------------------------------------------------------------------------------------- (3) dot products
__device__ inline void split3(float x, unsigned short &h, unsigned short &m, unsigned short &l) {
	const unsigned xb = __builtin_bit_cast(unsigned, x);
	const float hi = __builtin_bit_cast(float, xb & 0xffff0000u);
	const float r1 = x - hi;
	const unsigned rb = __builtin_bit_cast(unsigned, r1);
	const float mid = __builtin_bit_cast(float, rb & 0xffff0000u);
	const float lo = r1 - mid;
	h = (unsigned short)(xb >> 16);
	m = (unsigned short)(rb >> 16);
	l = (unsigned short)(__builtin_bit_cast(unsigned, lo) >> 16);
}

// one wave: D(32x32) = A(32xK) * B(Kx32); A stored [i][k], B stored [j][k] (both k-contiguous), fp32
template <int NPROD>
__global__ void dot_split(const float *A, const float *B, float *D, int K) {
	const int lane = threadIdx.x;
	f32x16 acc;
	for (int r = 0; r < 16; ++r) acc[r] = 0.f;
	for (int k0 = 0; k0 < K; k0 += 16) {
		u16x8 fa[3], fb[3];
		for (int e = 0; e < 8; ++e) {
			const int k = k0 + (lane >> 5) * 8 + e;
			unsigned short h, m, l;
			split3(A[(lane & 31) * K + k], h, m, l);
			fa[0][e] = h, fa[1][e] = m, fa[2][e] = l;
			split3(B[(lane & 31) * K + k], h, m, l);
			fb[0][e] = h, fb[1][e] = m, fb[2][e] = l;
		}
		// smallest partial products first
		for (int order = 4; order >= 0; --order)
			for (int sa = 2; sa >= 0; --sa) {
				const int sb = order - sa;
				if (sb < 0 || sb > 2) continue;
				if (NPROD == 6 && order > 2) continue;
				if (NPROD == 3 && order > 1) continue;
				if (NPROD == 1 && order > 0) continue;
				acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[sa]), __builtin_bit_cast(bf16x8, fb[sb]), acc, 0, 0, 0);
			}
	}
	for (int r = 0; r < 16; ++r) D[(8 * (r / 4) + 4 * (lane >> 5) + (r & 3)) * 32 + (lane & 31)] = acc[r];
}

__global__ void dot_f32(const float *A, const float *B, float *D, int K) {
	const int lane = threadIdx.x;
	f32x16 acc;
	for (int r = 0; r < 16; ++r) acc[r] = 0.f;
	for (int k0 = 0; k0 < K; k0 += 2) {
		const int k = k0 + (lane >> 5);
		acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(lane & 31) * K + k], B[(lane & 31) * K + k], acc, 0, 0, 0);
	}
	for (int r = 0; r < 16; ++r) D[(8 * (r / 4) + 4 * (lane >> 5) + (r & 3)) * 32 + (lane & 31)] = acc[r];
}

static float bf16_to_f(unsigned short h) {
	unsigned u = (unsigned)h << 16;
	float f;
	memcpy(&f, &u, 4);
	return f;
}

template <class F>
static void rate(const char *what, F launch, double flop_per_launch) {
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	for (int rep = 0; rep < 3; ++rep) {
		hipEventRecord(e0);
		launch();
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms = 0;
		hipEventElapsedTime(&ms, e0, e1);
		printf("%-64s %8.2f ms  %8.1f TFLOP/s executed\n", what, ms, flop_per_launch / ms / 1e9);
	}
}

int main() {
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	float *out;
	hipMalloc(&out, (size_t)cus * 4 * 256 * 4);

	// (1)
	const int iters = 100000;
	const double per_mfma = 32.0 * 32 * 16 * 2;
	rate("bf16 32x32x16, 1 wave/SIMD, 4 independent accumulators", [&] { bf16_loop<4><<<cus, 256>>>(out, iters); },
	     (double)cus * 4 * iters * 4 * per_mfma);
	rate("bf16 32x32x16, 2 waves/SIMD, 4 independent accumulators", [&] { bf16_loop<4><<<cus * 2, 256>>>(out, iters); },
	     (double)cus * 2 * 4 * iters * 4 * per_mfma);
	const int it2 = 20000;
	rate("split shape: 12 ds_read_b128 + 36 MFMA per k16, 1 wave/SIMD", [&] { split_shape_loop<9><<<cus, 256>>>(out, it2); },
	     (double)cus * 4 * it2 * 36 * per_mfma);
	rate("split shape: 12 ds_read_b128 + 36 MFMA per k16, 2 waves/SIMD", [&] { split_shape_loop<9><<<cus * 2, 256>>>(out, it2); },
	     (double)cus * 2 * 4 * it2 * 36 * per_mfma);
	rate("split shape: 12 ds_read_b128 + 24 MFMA per k16, 1 wave/SIMD", [&] { split_shape_loop<6><<<cus, 256>>>(out, it2); },
	     (double)cus * 4 * it2 * 24 * per_mfma);
	rate("split shape: 12 ds_read_b128 + 24 MFMA per k16, 2 waves/SIMD", [&] { split_shape_loop<6><<<cus * 2, 256>>>(out, it2); },
	     (double)cus * 2 * 4 * it2 * 24 * per_mfma);

	// (2) rounding of one MFMA: random bf16 vectors with a wide exponent spread + a large accumulator
	{
		std::vector<unsigned short> ha(32 * 16), hb(32 * 16);
		std::vector<float> hc(32 * 32), hd(32 * 32);
		srand(7);
		double worst = 0, sumsq = 0;
		int n = 0, inexact = 0;
		unsigned short *da, *db;
		float *dc, *dd;
		hipMalloc(&da, 32 * 16 * 2), hipMalloc(&db, 32 * 16 * 2), hipMalloc(&dc, 32 * 32 * 4), hipMalloc(&dd, 32 * 32 * 4);
		for (int trial = 0; trial < 200; ++trial) {
			for (auto &v : ha) v = (unsigned short)(((rand() & 1) << 15) | ((120 + rand() % 14) << 7) | (rand() & 127));
			for (auto &v : hb) v = (unsigned short)(((rand() & 1) << 15) | ((120 + rand() % 14) << 7) | (rand() & 127));
			for (auto &v : hc) v = (float)((rand() % 2001 - 1000) * (trial % 2 ? 1e-3 : 1e-6));
			hipMemcpy(da, ha.data(), 32 * 16 * 2, hipMemcpyHostToDevice);
			hipMemcpy(db, hb.data(), 32 * 16 * 2, hipMemcpyHostToDevice);
			hipMemcpy(dc, hc.data(), 32 * 32 * 4, hipMemcpyHostToDevice);
			one_mfma<<<1, 64>>>(da, db, dc, dd);
			hipMemcpy(hd.data(), dd, 32 * 32 * 4, hipMemcpyDeviceToHost);
			for (int i = 0; i < 32; ++i)
				for (int j = 0; j < 32; ++j) {
					double exact = hc[i * 32 + j];
					for (int k = 0; k < 16; ++k) exact += (double)bf16_to_f(ha[i * 16 + k]) * (double)bf16_to_f(hb[j * 16 + k]);
					const float rn = (float)exact;                    // correctly rounded result
					const double ulp = ldexp(1.0, ilogb(fabs(exact) > 0 ? fabs(exact) : 1e-30) - 23);
					const double err = fabs((double)hd[i * 32 + j] - exact) / ulp;
					worst = err > worst ? err : worst;
					sumsq += err * err;
					inexact += hd[i * 32 + j] != rn;
					++n;
				}
		}
		printf("one bf16 MFMA (16 exact products + C) vs exact: worst %.3f ulp, rms %.3f ulp, %d of %d differ from the correctly rounded sum\n",
		       worst, sqrt(sumsq / n), inexact, n);
	}

	// (3) K-long dot products
	for (int K : {64, 576, 2048}) {
		std::vector<float> hA(32 * K), hB(32 * K), hD(32 * 32);
		srand(11 + K);
		for (auto &v : hA) v = (float)((rand() / (double)RAND_MAX - 0.5) * 2.0) * (1.0f + (rand() & 7));
		for (auto &v : hB) v = (float)((rand() / (double)RAND_MAX - 0.5) * 0.2);
		float *dA, *dB, *dD;
		hipMalloc(&dA, 32 * K * 4), hipMalloc(&dB, 32 * K * 4), hipMalloc(&dD, 32 * 32 * 4);
		hipMemcpy(dA, hA.data(), 32 * K * 4, hipMemcpyHostToDevice);
		hipMemcpy(dB, hB.data(), 32 * K * 4, hipMemcpyHostToDevice);
		std::vector<double> ref(32 * 32), scale(32 * 32);
		for (int i = 0; i < 32; ++i)
			for (int j = 0; j < 32; ++j) {
				double s = 0, m = 0;
				for (int k = 0; k < K; ++k) s += (double)hA[i * K + k] * hB[j * K + k], m += fabs((double)hA[i * K + k] * hB[j * K + k]);
				ref[i * 32 + j] = s, scale[i * 32 + j] = m;
			}
		auto report = [&](const char *what) {
			hipMemcpy(hD.data(), dD, 32 * 32 * 4, hipMemcpyDeviceToHost);
			double worst = 0, sumsq = 0;
			for (int i = 0; i < 1024; ++i) {
				const double e = fabs(hD[i] - ref[i]) / scale[i];          // relative to sum |a||b|
				worst = e > worst ? e : worst, sumsq += e * e;
			}
			printf("K=%4d %-28s max |err| / sum|a||b| = %.3e (%.2f x 2^-24), rms %.3e\n", K, what, worst, worst * 16777216.0, sqrt(sumsq / 1024));
		};
		dot_f32<<<1, 64>>>(dA, dB, dD, K);
		report("fp32 MFMA 32x32x2");
		dot_split<9><<<1, 64>>>(dA, dB, dD, K);
		report("bf16 split, 9 products");
		dot_split<6><<<1, 64>>>(dA, dB, dD, K);
		report("bf16 split, 6 products");
		dot_split<3><<<1, 64>>>(dA, dB, dD, K);
		report("bf16 split, 3 products");
		dot_split<1><<<1, 64>>>(dA, dB, dD, K);
		report("plain bf16 (1 product)");
	}
	return 0;
}
