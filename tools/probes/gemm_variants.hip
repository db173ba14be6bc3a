// Probe (round 4, session 6): the gather-free test bed the review asked for. One row-major fp32 GEMM (NN, 4096^3 by default,
// uniform [-1, 1) operands as MI355X_MICROARCH.md:389 quotes its 122 TFLOP/s) on v_mfma_f32_32x32x2_f32 in a grid of
// structures, so that ONE run on the GPU box says which structural knob moves a finite launch:
//   workgroup = WM x WN waves, wave tile = TM x TN MFMA tiles of 32 x 32 (BM = 32 WM TM, BN = 32 WN TN), k-tile BK;
//   PIPE 0  no software pipelining: global -> registers -> LDS, barrier, multiply, barrier (the guide's "untuned" kernel)
//   PIPE 1  the production order of csrc/gemm.hip: next tile's global loads issued in front of the multiply, parked in the
//           other LDS buffer behind it, one barrier per k-tile
//   PIPE 2  one barrier, two LDS buffers, no register staging across the multiply: the next tile is loaded AND parked in front
//           of the multiply (other waves cover the latency)
//   SWZ 0   block b -> tile b (m fastest);  SWZ 1  block b -> XCD b % 8 owns a contiguous range of tiles walked in bands
//           of 8 m-tiles (the operand panels of neighbouring tiles meet in that XCD's L2)
// Every variant must equal the k-ordered fmaf chain of a one-thread-per-output kernel bit for bit.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemm_variants gemm_variants.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NT, int ROWS, int BK, bool KMAJOR, int PAD = 32>
struct Loader {
	static constexpr int LD = ROWS + PAD;
	static constexpr int NV = ROWS * BK / 4 / NT;
	static_assert(NV >= 1 && NV * NT * 4 == ROWS * BK, "tile / thread count");
	f32x4 reg[NV];
	__device__ __forceinline__ void load(const float *__restrict__ src, int ld, int row0, int k0, int tid) {
#pragma unroll
		for (int i = 0; i < NV; ++i) {
			const int v = tid + NT * i;
			int r, k;
			if (KMAJOR) r = (v % (ROWS / 4)) * 4, k = v / (ROWS / 4);
			else r = v % ROWS, k = (v / ROWS) * 4;
			const float *p = KMAJOR ? src + (size_t)(k0 + k) * ld + row0 + r : src + (size_t)(row0 + r) * ld + k0 + k;
			reg[i] = *reinterpret_cast<const f32x4 *>(p);
		}
	}
	__device__ __forceinline__ void park(float *lds, int tid) const {
#pragma unroll
		for (int i = 0; i < NV; ++i) {
			const int v = tid + NT * i;
			if (KMAJOR) {
				const int r = (v % (ROWS / 4)) * 4, k = v / (ROWS / 4);
				*reinterpret_cast<f32x4 *>(&lds[k * LD + r]) = reg[i];
			} else {
				const int r = v % ROWS, k = (v / ROWS) * 4;
#pragma unroll
				for (int e = 0; e < 4; ++e) lds[(k + e) * LD + r] = reg[i][e];
			}
		}
	}
};

template <int WM, int WN, int TM, int TN, int BK, int PIPE, int SWZ, int PAD = 32, int SCH = 0>
__global__ void __launch_bounds__(64 * WM * WN) gv_kernel(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C,
                                                          int M, int N, int K) {
	constexpr int NT = 64 * WM * WN, BM = 32 * WM * TM, BN = 32 * WN * TN;
	using LA = Loader<NT, BM, BK, false, PAD>;
	using LB = Loader<NT, BN, BK, true, PAD>;
	constexpr int NBUF = PIPE == 0 ? 1 : 2;
	__shared__ __attribute__((aligned(16))) float As[NBUF][BK * LA::LD];
	__shared__ __attribute__((aligned(16))) float Bs[NBUF][BK * LB::LD];

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = wave / WN, wn = wave % WN;
	const int l31 = lane & 31, lhi = lane >> 5;
	const int tiles_m = M / BM, tiles_n = N / BN, tiles = tiles_m * tiles_n;
	int t = blockIdx.x;
	if (SWZ) {
		const int per = tiles / 8, rem = tiles % 8, x = blockIdx.x % 8;        // XCD x owns per (+1) consecutive tiles
		t = x * per + min(x, rem) + blockIdx.x / 8;
	}
	int tm, tn;
	if (SWZ) {
		constexpr int G = 8;
		const int band = t / (G * tiles_n), within = t % (G * tiles_n);
		const int gh = min(G, tiles_m - band * G);
		tm = band * G + within % gh, tn = within / gh;
	} else {
		tm = t % tiles_m, tn = t / tiles_m;
	}
	const int m0 = tm * BM, n0 = tn * BN;

	f32x16 acc[TM][TN];
#pragma unroll
	for (int i = 0; i < TM; ++i)
#pragma unroll
		for (int j = 0; j < TN; ++j)
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

	LA la;
	LB lb;
	auto multiply = [&](int buf) {
		const float *as = As[buf] + wm * (BM / WM) + l31, *bs = Bs[buf] + wn * (BN / WN) + l31;
		if (SCH == 0) {
#pragma unroll
			for (int ks = 0; ks < BK; ks += 2) {
				float av[TM], bv[TN];
#pragma unroll
				for (int i = 0; i < TM; ++i) av[i] = as[(ks + lhi) * LA::LD + i * 32];
#pragma unroll
				for (int j = 0; j < TN; ++j) bv[j] = bs[(ks + lhi) * LB::LD + j * 32];
#pragma unroll
				for (int i = 0; i < TM; ++i)
#pragma unroll
					for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
			}
		} else {            // the conv kernel's order: fragments of step j+1 read in front of the MFMAs of step j, pinned by sched barriers
			float av[2][TM], bv[2][TN];
#pragma unroll
			for (int i = 0; i < TM; ++i) av[0][i] = as[lhi * LA::LD + i * 32];
#pragma unroll
			for (int j = 0; j < TN; ++j) bv[0][j] = bs[lhi * LB::LD + j * 32];
#pragma unroll
			for (int s2 = 0; s2 < BK / 2; ++s2) {
				if (s2 + 1 < BK / 2) {
#pragma unroll
					for (int i = 0; i < TM; ++i) av[(s2 + 1) & 1][i] = as[(2 * (s2 + 1) + lhi) * LA::LD + i * 32];
#pragma unroll
					for (int j = 0; j < TN; ++j) bv[(s2 + 1) & 1][j] = bs[(2 * (s2 + 1) + lhi) * LB::LD + j * 32];
				}
				__builtin_amdgcn_sched_barrier(0);
#pragma unroll
				for (int i = 0; i < TM; ++i)
#pragma unroll
					for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s2 & 1][i], bv[s2 & 1][j], acc[i][j], 0, 0, 0);
				__builtin_amdgcn_sched_barrier(0);
			}
		}
	};

	if (PIPE == 0) {
		for (int k0 = 0; k0 < K; k0 += BK) {
			la.load(A, K, m0, k0, tid);
			lb.load(B, N, n0, k0, tid);
			la.park(As[0], tid);
			lb.park(Bs[0], tid);
			__syncthreads();
			multiply(0);
			__syncthreads();
		}
	} else if (PIPE == 1) {
		la.load(A, K, m0, 0, tid);
		lb.load(B, N, n0, 0, tid);
		la.park(As[0], tid);
		lb.park(Bs[0], tid);
		__syncthreads();
		int buf = 0;
		for (int k0 = 0; k0 < K; k0 += BK, buf ^= 1) {
			const bool more = k0 + BK < K;
			if (more) {
				la.load(A, K, m0, k0 + BK, tid);
				lb.load(B, N, n0, k0 + BK, tid);
			}
			multiply(buf);
			if (more) {
				la.park(As[buf ^ 1], tid);
				lb.park(Bs[buf ^ 1], tid);
			}
			__syncthreads();
		}
	} else {
		la.load(A, K, m0, 0, tid);
		lb.load(B, N, n0, 0, tid);
		la.park(As[0], tid);
		lb.park(Bs[0], tid);
		__syncthreads();
		int buf = 0;
		for (int k0 = 0; k0 < K; k0 += BK, buf ^= 1) {
			if (k0 + BK < K) {
				la.load(A, K, m0, k0 + BK, tid);
				lb.load(B, N, n0, k0 + BK, tid);
				la.park(As[buf ^ 1], tid);
				lb.park(Bs[buf ^ 1], tid);
			}
			multiply(buf);
			__syncthreads();
		}
	}

#pragma unroll
	for (int j = 0; j < TN; ++j) {
		const int n = n0 + wn * (BN / WN) + j * 32 + l31;
#pragma unroll
		for (int i = 0; i < TM; ++i)
#pragma unroll
			for (int r = 0; r < 16; ++r) {
				const int m = m0 + wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
				C[(size_t)m * N + n] = acc[i][j][r];
			}
	}
}

// The same 128 x 128 / 4-wave / BK 16 / PIPE 1 structure on v_mfma_f32_16x16x4_f32 (16 MFMAs of 32 cycles per wave and 4 reduction
// elements instead of 8 of 64 cycles per 2): same FLOP per cycle, half the accumulator registers read and written per MAC, twice
// the operand registers — a power question on a chip that runs these loops at its power limit.
typedef float f32x4acc __attribute__((ext_vector_type(4)));
template <int SWZ, int XLDS = 0>
__global__ void __launch_bounds__(256) gv16_kernel(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, int M, int N, int K) {
	constexpr int NT = 256, BM = 128, BN = 128, BK = 16;
	using LA = Loader<NT, BM, BK, false, 16>;        // row stride 144: the four k rows of a fragment read sit 16 banks apart
	using LB = Loader<NT, BN, BK, true, 16>;
	__shared__ __attribute__((aligned(16))) float As[2][BK * LA::LD];
	__shared__ __attribute__((aligned(16))) float Bs[2][BK * LB::LD];
	__shared__ float ballast[XLDS > 0 ? XLDS : 1];          // XLDS floats of unused LDS: fewer resident workgroups per CU at the same code
	if (XLDS > 0 && K < 0) ballast[threadIdx.x % XLDS] = 1.f;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = wave >> 1, wn = wave & 1;
	const int l15 = lane & 15, lq = lane >> 4;
	const int tiles_m = M / BM, tiles_n = N / BN, tiles = tiles_m * tiles_n;
	int t = blockIdx.x;
	if (SWZ) {
		const int per = tiles / 8, rem = tiles % 8, x = blockIdx.x % 8;
		t = x * per + min(x, rem) + blockIdx.x / 8;
	}
	const int tm = t % tiles_m, tn = t / tiles_m;
	const int m0 = tm * BM, n0 = tn * BN;
	f32x4acc acc[4][4];
#pragma unroll
	for (int i = 0; i < 4; ++i)
#pragma unroll
		for (int j = 0; j < 4; ++j) acc[i][j] = f32x4acc{0.f, 0.f, 0.f, 0.f};
	LA la;
	LB lb;
	la.load(A, K, m0, 0, tid);
	lb.load(B, N, n0, 0, tid);
	la.park(As[0], tid);
	lb.park(Bs[0], tid);
	__syncthreads();
	int buf = 0;
	for (int k0 = 0; k0 < K; k0 += BK, buf ^= 1) {
		const bool more = k0 + BK < K;
		if (more) {
			la.load(A, K, m0, k0 + BK, tid);
			lb.load(B, N, n0, k0 + BK, tid);
		}
		const float *as = As[buf] + wm * 64 + l15, *bs = Bs[buf] + wn * 64 + l15;
#pragma unroll
		for (int ks = 0; ks < BK; ks += 4) {
			float av[4], bv[4];
#pragma unroll
			for (int i = 0; i < 4; ++i) av[i] = as[(ks + lq) * LA::LD + i * 16];
#pragma unroll
			for (int j = 0; j < 4; ++j) bv[j] = bs[(ks + lq) * LB::LD + j * 16];
#pragma unroll
			for (int i = 0; i < 4; ++i)
#pragma unroll
				for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
		}
		if (more) {
			la.park(As[buf ^ 1], tid);
			lb.park(Bs[buf ^ 1], tid);
		}
		__syncthreads();
	}
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const int n = n0 + wn * 64 + j * 16 + l15;
#pragma unroll
		for (int i = 0; i < 4; ++i)
#pragma unroll
			for (int r = 0; r < 4; ++r) C[(size_t)(m0 + wm * 64 + i * 16 + 4 * lq + r) * N + n] = acc[i][j][r];
	}
}

// the k-ordered fmaf chain the MFMA result must equal bit for bit
__global__ void __launch_bounds__(256) ref_kernel(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, int M, int N, int K) {
	const int n = blockIdx.x * 64 + (threadIdx.x & 63), m = blockIdx.y * 4 + (threadIdx.x >> 6);
	float s = 0.f;
	for (int k = 0; k < K; ++k) s = __builtin_fmaf(A[(size_t)m * K + k], B[(size_t)k * N + n], s);
	C[(size_t)m * N + n] = s;
}

__global__ void diff_kernel(const float *a, const float *b, size_t n, unsigned *count) {
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n && __float_as_uint(a[i]) != __float_as_uint(b[i])) atomicAdd(count, 1u);
}

struct Ctx {
	float *A, *B, *C, *R;
	unsigned *cnt;
	int M, N, K, reps;
};

template <int WM, int WN, int TM, int TN, int BK, int PIPE, int SWZ, int PAD = 32, int SCH = 0>
void run(const Ctx &c) {
	constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN, NT = 64 * WM * WN;
	if (c.M % BM || c.N % BN || c.K % BK) return;
	auto kern = gv_kernel<WM, WN, TM, TN, BK, PIPE, SWZ, PAD, SCH>;
	hipFuncAttributes fa;
	CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kern)));
	const dim3 grid((c.M / BM) * (c.N / BN));
	CK(hipMemset(c.C, 0xff, (size_t)c.M * c.N * 4));
	kern<<<grid, NT>>>(c.A, c.B, c.C, c.M, c.N, c.K);
	CK(hipGetLastError());
	CK(hipMemset(c.cnt, 0, 4));
	diff_kernel<<<(unsigned)(((size_t)c.M * c.N + 255) / 256), 256>>>(c.C, c.R, (size_t)c.M * c.N, c.cnt);
	unsigned bad = 0;
	CK(hipMemcpy(&bad, c.cnt, 4, hipMemcpyDeviceToHost));
	for (int i = 0; i < 2 + c.reps / 4; ++i) kern<<<grid, NT>>>(c.A, c.B, c.C, c.M, c.N, c.K);      // warm: clocks settle under this kernel's load
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	CK(hipEventRecord(e0));
	for (int i = 0; i < c.reps; ++i) kern<<<grid, NT>>>(c.A, c.B, c.C, c.M, c.N, c.K);
	CK(hipEventRecord(e1));
	CK(hipEventSynchronize(e1));
	float ms;
	CK(hipEventElapsedTime(&ms, e0, e1));
	const double us = ms * 1e3 / c.reps, tf = 2.0 * c.M * c.N * c.K / (us * 1e-6) / 1e12;
	printf("%3dx%-3d %dx%d waves of %dx%d  BK %2d  PIPE %d  SWZ %d PAD %2d SCH %d | %3d vgpr %5zu B scratch %6zu B lds | %8.1f us %6.1f TF %5.3f | %s\n", BM, BN, WM, WN,
	       TM * 32, TN * 32, BK, PIPE, SWZ, PAD, SCH, fa.numRegs, (size_t)fa.localSizeBytes, (size_t)fa.sharedSizeBytes, us, tf, tf / 157.3,
	       bad ? "DIFFERS" : "bit-identical");
	fflush(stdout);
	CK(hipEventDestroy(e0));
	CK(hipEventDestroy(e1));
}

template <int SWZ, int XLDS = 0>
void run16(const Ctx &c) {
	if (c.M % 128 || c.N % 128 || c.K % 16) return;
	auto kern = gv16_kernel<SWZ, XLDS>;
	hipFuncAttributes fa;
	CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kern)));
	const dim3 grid((c.M / 128) * (c.N / 128));
	CK(hipMemset(c.C, 0xff, (size_t)c.M * c.N * 4));
	kern<<<grid, 256>>>(c.A, c.B, c.C, c.M, c.N, c.K);
	CK(hipGetLastError());
	CK(hipMemset(c.cnt, 0, 4));
	diff_kernel<<<(unsigned)(((size_t)c.M * c.N + 255) / 256), 256>>>(c.C, c.R, (size_t)c.M * c.N, c.cnt);
	unsigned bad = 0;
	CK(hipMemcpy(&bad, c.cnt, 4, hipMemcpyDeviceToHost));
	for (int i = 0; i < 2 + c.reps / 4; ++i) kern<<<grid, 256>>>(c.A, c.B, c.C, c.M, c.N, c.K);
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	CK(hipEventRecord(e0));
	for (int i = 0; i < c.reps; ++i) kern<<<grid, 256>>>(c.A, c.B, c.C, c.M, c.N, c.K);
	CK(hipEventRecord(e1));
	CK(hipEventSynchronize(e1));
	float ms;
	CK(hipEventElapsedTime(&ms, e0, e1));
	const double us = ms * 1e3 / c.reps, tf = 2.0 * c.M * c.N * c.K / (us * 1e-6) / 1e12;
	printf("128x128 2x2 waves of 64x64  BK 16  PIPE 1  SWZ %d on v_mfma_f32_16x16x4_f32 | %3d vgpr %6zu B lds | %8.1f us %6.1f TF %5.3f | %s\n", SWZ, fa.numRegs,
	       (size_t)fa.sharedSizeBytes, us, tf, tf / 157.3, bad ? "DIFFERS" : "bit-identical");
	fflush(stdout);
}

template <int WM, int WN, int TM, int TN>
void family(const Ctx &c) {
	run<WM, WN, TM, TN, 16, 1, 0>(c);
	run<WM, WN, TM, TN, 16, 1, 1>(c);
	run<WM, WN, TM, TN, 32, 1, 1>(c);
	run<WM, WN, TM, TN, 16, 0, 1>(c);
	run<WM, WN, TM, TN, 32, 0, 1>(c);
	run<WM, WN, TM, TN, 16, 2, 1>(c);
	run<WM, WN, TM, TN, 32, 2, 1>(c);
}

int main(int argc, char **argv) {
	Ctx c;
	c.M = argc > 1 ? atoi(argv[1]) : 4096, c.K = argc > 2 ? atoi(argv[2]) : 4096, c.N = argc > 3 ? atoi(argv[3]) : 4096;
	c.reps = argc > 4 ? atoi(argv[4]) : 20;
	const char *data = argc > 5 ? argv[5] : "uniform";
	std::vector<float> ha((size_t)c.M * c.K), hb((size_t)c.K * c.N);
	unsigned s = 12345u;
	auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffffff) / 8388608.0f - 1.0f; };
	const bool zeros = !strcmp(data, "zeros");
	for (auto &v : ha) v = zeros ? 0.f : rnd();
	for (auto &v : hb) v = zeros ? 0.f : rnd();
	CK(hipMalloc(&c.A, ha.size() * 4));
	CK(hipMalloc(&c.B, hb.size() * 4));
	CK(hipMalloc(&c.C, (size_t)c.M * c.N * 4));
	CK(hipMalloc(&c.R, (size_t)c.M * c.N * 4));
	CK(hipMalloc(&c.cnt, 4));
	CK(hipMemcpy(c.A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
	CK(hipMemcpy(c.B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
	ref_kernel<<<dim3(c.N / 64, c.M / 4), 256>>>(c.A, c.B, c.R, c.M, c.N, c.K);
	CK(hipDeviceSynchronize());
	printf("GEMM NN %d x %d x %d, %s operands, %d launches back to back; peak 157.3 TFLOP/s\n", c.M, c.K, c.N, data, c.reps);
	if (argc > 6 && !strcmp(argv[6], "bisect")) {      // which of the conv kernel's choices costs: unpadded LDS rows, its fragment schedule
		run<2, 2, 2, 2, 32, 0, 1, 32, 0>(c);
		run<2, 2, 2, 2, 32, 0, 1, 0, 0>(c);
		run<2, 2, 2, 2, 32, 0, 1, 32, 1>(c);
		run<2, 2, 2, 2, 32, 0, 1, 0, 1>(c);
		run<2, 2, 2, 2, 16, 1, 1, 32, 0>(c);
		run<2, 2, 2, 2, 16, 1, 1, 0, 0>(c);
		run<2, 2, 2, 2, 16, 1, 1, 32, 1>(c);
		run<2, 2, 2, 2, 16, 1, 1, 0, 1>(c);
		return 0;
	}
	if (argc > 6 && !strcmp(argv[6], "mfma16")) {      // instruction shape at equal structure, A B A B
		for (int rep = 0; rep < 2; ++rep) {
			run<2, 2, 2, 2, 16, 1, 1>(c);
			run16<1>(c);
			run16<1, 4096>(c);        // 52 KB: three workgroups per CU, as the 32x32x2 variant's 132 registers allow
			run16<1, 12288>(c);       // 84 KB: one workgroup per CU
		}
		return 0;
	}
	if (argc > 6 && !strcmp(argv[6], "key")) {         // the candidates, for long runs (reps >= 300: 20-launch bursts read 10 % off either way)
		run<2, 2, 2, 2, 16, 1, 0>(c);
		run<2, 2, 2, 2, 16, 1, 1>(c);
		run<2, 2, 2, 2, 32, 0, 1>(c);
		run<2, 2, 2, 2, 16, 2, 1>(c);
		run<2, 2, 2, 4, 16, 1, 1>(c);
		run<2, 2, 2, 4, 16, 0, 1>(c);
		run<4, 2, 2, 2, 16, 1, 1>(c);
		run<2, 4, 2, 2, 16, 1, 1>(c);
		run<2, 4, 2, 2, 16, 0, 1>(c);
		run<4, 4, 2, 2, 16, 1, 1>(c);
		run<4, 4, 2, 2, 32, 1, 1>(c);
		run<2, 4, 4, 2, 16, 1, 1>(c);
		run<4, 2, 2, 4, 16, 1, 1>(c);
		run<2, 2, 2, 2, 16, 1, 0>(c);                  // the first one again: drift over the run
		return 0;
	}
	family<2, 2, 2, 2>(c);      // 128 x 128, 4 waves (production shape)
	family<2, 2, 4, 2>(c);      // 256 x 128, 4 waves of 128 x 64
	family<2, 2, 2, 4>(c);      // 128 x 256, 4 waves of 64 x 128
	family<2, 2, 4, 4>(c);      // 256 x 256, 4 waves of 128 x 128 (256 accumulator registers)
	family<4, 2, 2, 2>(c);      // 256 x 128, 8 waves
	family<2, 4, 2, 2>(c);      // 128 x 256, 8 waves
	family<4, 4, 2, 2>(c);      // 256 x 256, 16 waves
	family<2, 4, 4, 2>(c);      // 256 x 256, 8 waves of 128 x 64
	family<4, 2, 2, 4>(c);      // 256 x 256, 8 waves of 64 x 128
	return 0;
}
