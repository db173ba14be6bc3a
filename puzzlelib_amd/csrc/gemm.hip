// Row-major fp32 GEMM on the f32 matrix cores: C[M,N] = alpha*op(A)*op(B) + beta*C, NN / NT / TN (not TT).
// Serves Linear forward (NN), dx (NT) and dW (TN with alpha=scale, beta=momentum) — Modules/Linear.py:36-54 via
// Blas.mulMatrixOnMatrix (Backend/Blas.py:60-61). Replaces BlasContext.gemm — Cuda/Source/Libs/CuBlas.c:327-402.
//
// Workgroup = 4 waves (2 x 2), tile BM x BN in {64, 128}^2, each wave (BM/2) x (BN/2) = TM x TN tiles of
// v_mfma_f32_32x32x2_f32; BK = 16, LDS double buffer, one barrier per k-tile, the next tile's global loads in flight
// while the current one is multiplied.
// LDS holds both operands reduction-major, As[k][m] and Bs[k][n] with a row stride of BM + 32 floats: a fragment read
// (lane l -> column l & 31 of row k + (l >> 5)) touches 64 distinct banks. Global reads always run along the
// contiguous axis of the source, 16 bytes per lane: an operand whose contiguous axis is m (or n) is parked with one
// ds_write_b128, one whose contiguous axis is k with four ds_write_b32 (lanes along m: conflict-free). Operands that
// are not 16-byte aligned / whose K is not a multiple of 4 take the 4-byte loader.
// Small outputs with long reductions (the 256 x 1000 x 2048 classifier of ResNet-50 is 16 tiles on 256 CUs) are split
// along K over blockIdx.z; partial tiles go to slabs and gemm_reduce_kernel adds them in slab order with the alpha / beta
// epilogue — deterministic, no atomics. 2*M*N*K FLOP; MFMA-bound above ~128 x 128 x 1k.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

struct GemmArgs {
	const float *a, *b;
	float *c;                       // the output, or the slabs when splits > 1
	int m, n, k, lda, ldb, ldc;
	float alpha, beta;
	int tiles_m, tiles_n, splits, ksteps_per_split;
};

constexpr int BK = 16;

// Loads one operand tile (ROWS x BK, ROWS = BM or BN) into registers and parks it in LDS as [k][row].
// KMAJOR: the source is stored [k][row] (row contiguous); else [row][k] (k contiguous). VEC: 16-byte accesses.
template <int ROWS, bool KMAJOR, bool VEC>
struct Loader {
	static constexpr int LD = ROWS + 32;
	static constexpr int NV = ROWS * BK / 4 / 256;            // float4 per thread per tile (2 for 128 rows, 1 for 64)
	f32x4 reg[NV];

	__device__ __forceinline__ void load(const float *__restrict__ src, int ld, int row0, int nrows, int k0, int kend, int tid) {
#pragma unroll
		for (int i = 0; i < NV; ++i) {
			const int v = tid + 256 * i;
			int r, k;
			if (KMAJOR) { r = (v % (ROWS / 4)) * 4, k = v / (ROWS / 4); }       // 4 consecutive rows of one k
			else        { r = v % ROWS, k = (v / ROWS) * 4; }                     // 4 consecutive k of one row
			const int gr = row0 + r, gk = k0 + k;
			f32x4 x = {0.f, 0.f, 0.f, 0.f};
			if (KMAJOR) {
				if (gk < kend) {
					const float *p = src + (size_t)gk * ld + gr;
					if (VEC && gr + 3 < nrows) x = *reinterpret_cast<const f32x4 *>(p);
					else {
#pragma unroll
						for (int e = 0; e < 4; ++e) if (gr + e < nrows) x[e] = p[e];
					}
				}
			} else {
				if (gr < nrows) {
					const float *p = src + (size_t)gr * ld + gk;
					if (VEC && gk + 3 < kend) x = *reinterpret_cast<const f32x4 *>(p);
					else {
#pragma unroll
						for (int e = 0; e < 4; ++e) if (gk + e < kend) x[e] = p[e];
					}
				}
			}
			reg[i] = x;
		}
	}

	__device__ __forceinline__ void park(float *lds, int tid) const {
#pragma unroll
		for (int i = 0; i < NV; ++i) {
			const int v = tid + 256 * i;
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

template <int BM, int BN, bool TA, bool TB, bool VEC>
__global__ void __launch_bounds__(256) gemm_kernel(GemmArgs g) {
	constexpr int TM = BM / 64, TN = BN / 64;
	using LA = Loader<BM, TA, VEC>;             // A stored [k][m] when transposed
	using LB = Loader<BN, !TB, VEC>;            // B stored [k][n] unless transposed
	__shared__ __attribute__((aligned(16))) float As[2][BK * LA::LD];
	__shared__ __attribute__((aligned(16))) float Bs[2][BK * LB::LD];

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = wave >> 1, wn = wave & 1;
	const int tm = blockIdx.x % g.tiles_m, tn = blockIdx.x / g.tiles_m;
	const int m0 = tm * BM, n0 = tn * BN;
	const int split = blockIdx.z;
	const int kbeg = split * g.ksteps_per_split * BK;
	const int kend = min(g.k, kbeg + g.ksteps_per_split * BK);
	const int l31 = lane & 31, lhi = lane >> 5;

	f32x16 acc[TM][TN];
#pragma unroll
	for (int i = 0; i < TM; ++i)
#pragma unroll
		for (int j = 0; j < TN; ++j)
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

	LA la;
	LB lb;
	la.load(g.a, g.lda, m0, g.m, kbeg, kend, tid);
	lb.load(g.b, g.ldb, n0, g.n, kbeg, kend, tid);
	la.park(As[0], tid);
	lb.park(Bs[0], tid);
	__syncthreads();

	int buf = 0;
	for (int k0 = kbeg; k0 < kend; k0 += BK, buf ^= 1) {
		const bool more = k0 + BK < kend;
		if (more) {
			la.load(g.a, g.lda, m0, g.m, k0 + BK, kend, tid);
			lb.load(g.b, g.ldb, n0, g.n, k0 + BK, kend, tid);
		}

		const float *as = As[buf] + wm * (BM / 2) + l31, *bs = Bs[buf] + wn * (BN / 2) + l31;
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

		if (more) {
			la.park(As[buf ^ 1], tid);
			lb.park(Bs[buf ^ 1], tid);
		}
		__syncthreads();
	}

	// epilogue: lane = column, 16 rows per MFMA tile; split-K partials go to their slab untouched
	const bool direct = g.splits == 1;
	float *out = direct ? g.c : g.c + (size_t)split * g.m * g.n;
	const int ldo = direct ? g.ldc : g.n;
#pragma unroll
	for (int j = 0; j < TN; ++j) {
		const int n = n0 + wn * (BN / 2) + j * 32 + l31;
		if (n >= g.n) continue;
#pragma unroll
		for (int i = 0; i < TM; ++i)
#pragma unroll
			for (int r = 0; r < 16; ++r) {
				const int m = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
				if (m < g.m) {
					float *o = out + (size_t)m * ldo + n;
					*o = direct ? (g.beta == 0.f ? 0.f : g.beta * *o) + g.alpha * acc[i][j][r] : acc[i][j][r];
				}
			}
	}
}

// C = beta*C + alpha * (slab_0 + slab_1 + ...), slabs added in order
__global__ void __launch_bounds__(256) gemm_reduce_kernel(const float *__restrict__ slabs, float *__restrict__ c, int m, int n,
                                                           int ldc, int splits, float alpha, float beta) {
	const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)m * n;
	if (idx >= total) return;
	float s = slabs[idx];
	for (int i = 1; i < splits; ++i) s += slabs[(size_t)i * total + idx];
	float *o = c + (idx / n) * ldc + idx % n;
	*o = (beta == 0.f ? 0.f : beta * *o) + alpha * s;
}

struct GemmPlan {
	int bm, bn, tiles_m, tiles_n, splits, ksteps_per_split;
};

GemmPlan plan_gemm(int m, int n, int k) {
	GemmPlan p;
	p.bm = m > 64 ? 128 : 64;
	p.bn = n > 64 ? 128 : 64;
	// prefer the smaller tile when the larger one leaves the chip mostly idle and pads a lot
	if (p.bm == 128 && pz::ceil_div(m, 128) * pz::ceil_div(n, p.bn) < pz::kNumCU / 2 && m % 128 != 0 && m % 128 <= 64) p.bm = 64;
	if (p.bn == 128 && pz::ceil_div(m, p.bm) * pz::ceil_div(n, 128) < pz::kNumCU / 2 && n % 128 != 0 && n % 128 <= 64) p.bn = 64;
	p.tiles_m = pz::ceil_div(m, p.bm), p.tiles_n = pz::ceil_div(n, p.bn);
	const int tiles = p.tiles_m * p.tiles_n, ksteps = pz::ceil_div(k, BK);
	int splits = 1;
	if (tiles < pz::kNumCU) {                                   // fill the chip: one balanced round, >= 8 k-tiles per split
		splits = (2 * pz::kNumCU) / tiles;
		if (splits > ksteps / 8) splits = ksteps / 8;
		if (splits < 1) splits = 1;
		if (splits > 64) splits = 64;
	}
	p.ksteps_per_split = pz::ceil_div(ksteps, splits);
	p.splits = pz::ceil_div(ksteps, p.ksteps_per_split);
	return p;
}

template <bool TA, bool TB, bool VEC>
void launch(const GemmPlan &p, const GemmArgs &g, hipStream_t st) {
	const dim3 grid(p.tiles_m * p.tiles_n, 1, p.splits);
	if (p.bm == 128 && p.bn == 128) gemm_kernel<128, 128, TA, TB, VEC><<<grid, 256, 0, st>>>(g);
	else if (p.bm == 128) gemm_kernel<128, 64, TA, TB, VEC><<<grid, 256, 0, st>>>(g);
	else if (p.bn == 128) gemm_kernel<64, 128, TA, TB, VEC><<<grid, 256, 0, st>>>(g);
	else gemm_kernel<64, 64, TA, TB, VEC><<<grid, 256, 0, st>>>(g);
}

}  // namespace

extern "C" {

int pz_gemm_workspace_bytes(int m, int n, int k, size_t *nbytes) {
	PZ_REQUIRE(m > 0 && n > 0 && k > 0 && nbytes, "pz_gemm_workspace_bytes: bad arguments");
	const GemmPlan p = plan_gemm(m, n, k);
	*nbytes = p.splits > 1 ? (size_t)p.splits * m * n * sizeof(float) : 0;
	return PZ_OK;
}

int pz_gemm_ws(int trans_a, int trans_b, int m, int n, int k, float alpha, const float *a, int lda, const float *b, int ldb,
               float beta, float *c, int ldc, void *workspace, size_t ws_bytes, pz_stream_t stream) {
	PZ_REQUIRE(!(trans_a && trans_b), "pz_gemm: both operands transposed is not supported");
	PZ_REQUIRE(m > 0 && n > 0 && k > 0, "pz_gemm: non-positive dimension (%d, %d, %d)", m, n, k);
	PZ_REQUIRE(a && b && c, "pz_gemm: null matrix");
	PZ_REQUIRE(lda >= (trans_a ? m : k) && ldb >= (trans_b ? k : n) && ldc >= n, "pz_gemm: leading dimension too small");

	GemmPlan p = plan_gemm(m, n, k);
	const size_t need = p.splits > 1 ? (size_t)p.splits * m * n * sizeof(float) : 0;
	if (ws_bytes < need || (need > 0 && workspace == nullptr)) p.splits = 1, p.ksteps_per_split = pz::ceil_div(k, BK);   // no scratch: one pass over K

	GemmArgs g{a, b, p.splits > 1 ? (float *)workspace : c, m, n, k, lda, ldb, ldc, alpha, beta, p.tiles_m, p.tiles_n, p.splits,
	           p.ksteps_per_split};
	hipStream_t st = pz::as_stream(stream);
	const bool vec = ((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0) && lda % 4 == 0 && ldb % 4 == 0;

	if (trans_a) vec ? launch<true, false, true>(p, g, st) : launch<true, false, false>(p, g, st);
	else if (trans_b) vec ? launch<false, true, true>(p, g, st) : launch<false, true, false>(p, g, st);
	else vec ? launch<false, false, true>(p, g, st) : launch<false, false, false>(p, g, st);
	PZ_LAUNCH_CHECK();

	if (p.splits > 1) {
		gemm_reduce_kernel<<<pz::ceil_div((long)m * n, 256), 256, 0, st>>>((const float *)workspace, c, m, n, ldc, p.splits, alpha, beta);
		PZ_LAUNCH_CHECK();
	}
	return PZ_OK;
}

int pz_gemm(int trans_a, int trans_b, int m, int n, int k, float alpha, const float *a, int lda, const float *b, int ldb,
            float beta, float *c, int ldc, pz_stream_t stream) {
	return pz_gemm_ws(trans_a, trans_b, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, nullptr, 0, stream);
}

}  // extern "C"
