// Row-major fp32 GEMM on the f32 matrix cores: C[M,N] = alpha*op(A)*op(B) + beta*C, NN / NT / TN (not TT).
// Serves Linear forward (NN), dx (NT) and dW (TN with alpha=scale, beta=momentum) — Modules/Linear.py:36-54 via
// Blas.mulMatrixOnMatrix (Backend/Blas.py:60-61). Replaces BlasContext.gemm — Cuda/Source/Libs/CuBlas.c:327-402.
//
// 64x64 workgroup tile, 4 waves of one 32x32x2 MFMA tile each, BK=16. Both operands are parked in LDS as
// [row][BK+1] so that a fragment read (lane l -> row l&31, k = l>>5) touches 32 distinct banks whichever way
// the source matrix is laid out; global reads run along the contiguous axis of each operand.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct GemmArgs {
	const float *a, *b;
	float *c;
	int m, n, k, lda, ldb, ldc;
	float alpha, beta;
	int tiles_m;
};

template <bool TA, bool TB>
__global__ void __launch_bounds__(256) gemm_kernel(GemmArgs g) {
	constexpr int BM = 64, BN = 64, BK = 16, LD = BK + 1;
	__shared__ float As[BM * LD];
	__shared__ float Bs[BN * LD];

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = wave >> 1, wn = wave & 1;
	const int tm = blockIdx.x % g.tiles_m, tn = blockIdx.x / g.tiles_m;
	const int m0 = tm * BM, n0 = tn * BN;

	// per-thread element coordinates inside a tile (4 elements of A, 4 of B per k-step)
	int ar[4], ak[4], br[4], bk[4];
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		if (TA) { ak[i] = (tid >> 6) + 4 * i; ar[i] = tid & 63; }      // A stored [k][m]: lanes along m
		else    { ar[i] = (tid >> 4) + 16 * i; ak[i] = tid & 15; }     // A stored [m][k]: lanes along k
		if (TB) { br[i] = (tid >> 4) + 16 * i; bk[i] = tid & 15; }     // B stored [n][k]: lanes along k
		else    { bk[i] = (tid >> 6) + 4 * i; br[i] = tid & 63; }      // B stored [k][n]: lanes along n
	}

	f32x16 acc;
#pragma unroll
	for (int r = 0; r < 16; ++r) acc[r] = 0.f;

	float ra[4], rb[4];
	unsigned amask = 0, bmask = 0;

	auto load = [&](int k0) {
		amask = bmask = 0;
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int m = m0 + ar[i], k = k0 + ak[i];
			const unsigned ok = (unsigned)(m < g.m) & (unsigned)(k < g.k);
			const size_t off = TA ? (size_t)k * g.lda + m : (size_t)m * g.lda + k;
			ra[i] = g.a[ok ? off : 0];
			amask |= ok << i;
		}
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int n = n0 + br[i], k = k0 + bk[i];
			const unsigned ok = (unsigned)(n < g.n) & (unsigned)(k < g.k);
			const size_t off = TB ? (size_t)n * g.ldb + k : (size_t)k * g.ldb + n;
			rb[i] = g.b[ok ? off : 0];
			bmask |= ok << i;
		}
	};

	const int l31 = lane & 31, lhi = lane >> 5;
	load(0);

	for (int k0 = 0; k0 < g.k; k0 += BK) {
		__syncthreads();
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			As[ar[i] * LD + ak[i]] = (amask >> i) & 1u ? ra[i] : 0.f;
			Bs[br[i] * LD + bk[i]] = (bmask >> i) & 1u ? rb[i] : 0.f;
		}
		__syncthreads();

		if (k0 + BK < g.k) load(k0 + BK);

#pragma unroll
		for (int ks = 0; ks < BK; ks += 2) {
			const float av = As[(wm * 32 + l31) * LD + ks + lhi];
			const float bv = Bs[(wn * 32 + l31) * LD + ks + lhi];
			acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
		}
	}

	const int n = n0 + wn * 32 + l31;
	if (n < g.n) {
#pragma unroll
		for (int r = 0; r < 16; ++r) {
			const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
			if (m < g.m) {
				float *o = g.c + (size_t)m * g.ldc + n;
				*o = (g.beta == 0.f ? 0.f : g.beta * *o) + g.alpha * acc[r];
			}
		}
	}
}

}  // namespace

extern "C" int pz_gemm(int trans_a, int trans_b, int m, int n, int k, float alpha, const float *a, int lda,
                       const float *b, int ldb, float beta, float *c, int ldc, pz_stream_t stream) {
	PZ_REQUIRE(!(trans_a && trans_b), "pz_gemm: both operands transposed is not supported");
	PZ_REQUIRE(m > 0 && n > 0 && k > 0, "pz_gemm: non-positive dimension (%d, %d, %d)", m, n, k);
	PZ_REQUIRE(a && b && c, "pz_gemm: null matrix");
	PZ_REQUIRE(lda >= (trans_a ? m : k) && ldb >= (trans_b ? k : n) && ldc >= n, "pz_gemm: leading dimension too small");

	GemmArgs g{a, b, c, m, n, k, lda, ldb, ldc, alpha, beta, pz::ceil_div(m, 64)};
	const int grid = g.tiles_m * pz::ceil_div(n, 64);
	hipStream_t st = pz::as_stream(stream);

	if (trans_a)
		gemm_kernel<true, false><<<grid, 256, 0, st>>>(g);
	else if (trans_b)
		gemm_kernel<false, true><<<grid, 256, 0, st>>>(g);
	else
		gemm_kernel<false, false><<<grid, 256, 0, st>>>(g);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}
