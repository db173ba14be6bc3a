// Row-major fp32 GEMM on the f32 matrix cores: C[M,N] = alpha*op(A)*op(B) + beta*C, NN / NT / TN (not TT).
// Serves Linear forward (NN), dx (NT) and dW (TN with alpha=scale, beta=momentum) — Modules/Linear.py:36-54 via
// Blas.mulMatrixOnMatrix (Backend/Blas.py:60-61). Replaces BlasContext.gemm — Cuda/Source/Libs/CuBlas.c:327-402.
//
// Workgroup = WM x WN waves, each wave (BM / WM) x (BN / WN) = TM x TN tiles of v_mfma_f32_32x32x2_f32: tiles {64, 128}^2 on 4
// waves (2 x 2; 98-104 registers: four workgroups per CU, +4 % over three), and 256 x 256 on 16 waves (4 x 4; a launch with at
// least one such tile per CU and K >= 1024 — 137 instead of 126 TFLOP/s at 4096^3: half the operand bytes per MFMA through L2 and LDS). BK = 16, LDS double buffer, one barrier per
// k-tile, the next tile's global loads in flight while the current one is multiplied.
// LDS holds both operands reduction-major, As[k][m] and Bs[k][n] with a row stride of BM + 32 floats: a fragment read
// (lane l -> column l & 31 of row k + (l >> 5)) touches 64 distinct banks. Global reads always run along the
// contiguous axis of the source, 16 bytes per lane: an operand whose contiguous axis is m (or n) is parked with one
// ds_write_b128, one whose contiguous axis is k with four ds_write_b32 (lanes along m: conflict-free).
// The 16-byte loads are BUFFER loads without a branch around them (round 4, session 6): the per-lane bounds tests of the
// first version (`if (row < rows) x = *p; else x = 0`) made the compiler wait for every load right behind its issue — the
// software pipeline never overlapped anything, 106 TFLOP/s at 4096^3 where tools/probes/gemm_variants.hip measured 126 for the
// same structure with unconditional loads. Now the descriptor covers exactly the matrix: rows beyond it read as zero (and
// rows >= M / columns >= N only feed outputs nobody stores), a quad beyond the reduction range takes an out-of-range
// offset (one v_cndmask per load). Operands that are not 16-byte aligned, whose K is not a multiple of 4 or that reach
// beyond 4 GiB take the 4-byte loader with explicit tests.
// Small outputs with long reductions (the 256 x 1000 x 2048 classifier of ResNet-50 is 16 tiles on 256 CUs) are split
// along K over blockIdx.z; partial tiles go to slabs and gemm_reduce_kernel adds them in slab order with the alpha / beta
// epilogue — deterministic, no atomics. 2*M*N*K FLOP; MFMA-bound above ~128 x 128 x 1k.
#include "common.h"

#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

struct GemmArgs {
	const float *a, *b;
	float *c;                       // the output, or the slabs when splits > 1
	int m, n, k, lda, ldb, ldc;
	float alpha, beta;
	int tiles_m, tiles_n, splits, ksteps_per_split;
	unsigned a_bytes, b_bytes;      // extents for the buffer descriptors of the 16-byte path
};

constexpr int BK = 16;
constexpr unsigned kOOB = 0xfffffff0u;      // buffer-load byte offset beyond every matrix: the hardware returns 0

// Loads one operand tile (ROWS x BK, ROWS = BM or BN) into registers and parks it in LDS as [k][row].
// KMAJOR: the source is stored [k][row] (row contiguous); else [row][k] (k contiguous). VEC: 16-byte buffer loads.
template <int NT, int ROWS, bool KMAJOR, bool VEC>
struct Loader {
	static constexpr int LD = ROWS + 32;
	static constexpr int NV = ROWS * BK / 4 / NT;             // float4 per thread per tile
	static_assert(NV >= 1 && NV * NT * 4 == ROWS * BK, "operand tile / thread count");
	f32x4 reg[NV];
	unsigned voff[NV];               // VEC: byte offset of this thread's quad in k-tile 0 of the launch (kOOB: row outside)
	int kq[NV];                      // VEC: its first k inside a k-tile

	__device__ __forceinline__ void init(int ld, int row0, int nrows, int kbeg, int tid) {
		if (!VEC) return;
#pragma unroll
		for (int i = 0; i < NV; ++i) {
			const int v = tid + NT * i;
			int r, k;
			if (KMAJOR) { r = (v % (ROWS / 4)) * 4, k = v / (ROWS / 4); }
			else        { r = v % ROWS, k = (v / ROWS) * 4; }
			const int gr = row0 + r;
			kq[i] = k;
			// a quad that starts inside the matrix and runs over its edge (KMAJOR, rows not a multiple of 4) reads the next
			// k-row's first elements or zeros beyond the descriptor: rows >= nrows only meet outputs that are not stored
			const size_t e = KMAJOR ? (size_t)(kbeg + k) * ld + gr : (size_t)gr * ld + kbeg + k;
			voff[i] = gr < nrows ? (unsigned)(e * 4) : kOOB;
		}
	}

	__device__ __forceinline__ void load(const float *__restrict__ src, __amdgpu_buffer_rsrc_t rs, int ld, int row0, int nrows, int k0,
	                                     int kend, int tid) {
#pragma unroll
		for (int i = 0; i < NV; ++i) {
			if (VEC) {
				const unsigned vo = k0 + kq[i] < kend ? voff[i] : kOOB;
				reg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0));
				voff[i] += KMAJOR ? (unsigned)(BK * ld) * 4u : (unsigned)BK * 4u;      // (kOOB stays beyond every matrix < 4 GiB - 64 K)
				continue;
			}
			const int v = tid + NT * i;
			int r, k;
			if (KMAJOR) { r = (v % (ROWS / 4)) * 4, k = v / (ROWS / 4); }       // 4 consecutive rows of one k
			else        { r = v % ROWS, k = (v / ROWS) * 4; }                     // 4 consecutive k of one row
			const int gr = row0 + r, gk = k0 + k;
			f32x4 x = {0.f, 0.f, 0.f, 0.f};
			if (KMAJOR) {
				if (gk < kend) {
					const float *p = src + (size_t)gk * ld + gr;
#pragma unroll
					for (int e = 0; e < 4; ++e) if (gr + e < nrows) x[e] = p[e];
				}
			} else {
				if (gr < nrows) {
					const float *p = src + (size_t)gr * ld + gk;
#pragma unroll
					for (int e = 0; e < 4; ++e) if (gk + e < kend) x[e] = p[e];
				}
			}
			reg[i] = x;
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

template <int BM, int BN, int WM, int WN, bool TA, bool TB, bool VEC>
__global__ void __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(4, 8))) gemm_kernel(GemmArgs g) {
	constexpr int NT = 64 * WM * WN, TM = BM / (32 * WM), TN = BN / (32 * WN);
	using LA = Loader<NT, BM, TA, VEC>;         // A stored [k][m] when transposed
	using LB = Loader<NT, BN, !TB, VEC>;        // B stored [k][n] unless transposed
	__shared__ __attribute__((aligned(16))) float As[2][BK * LA::LD];
	__shared__ __attribute__((aligned(16))) float Bs[2][BK * LB::LD];

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = wave / WN, wn = wave % WN;
	const int tm = blockIdx.x % g.tiles_m, tn = blockIdx.x / g.tiles_m;
	const int m0 = tm * BM, n0 = tn * BN;
	const int split = blockIdx.z;
	const int kbeg = split * g.ksteps_per_split * BK;
	const int kend = min(g.k, kbeg + g.ksteps_per_split * BK);
	const int l31 = lane & 31, lhi = lane >> 5;

	const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc((void *)g.a, 0, g.a_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc((void *)g.b, 0, g.b_bytes, 0x00020000);

	f32x16 acc[TM][TN];
#pragma unroll
	for (int i = 0; i < TM; ++i)
#pragma unroll
		for (int j = 0; j < TN; ++j)
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

	LA la;
	LB lb;
	la.init(g.lda, m0, g.m, kbeg, tid);
	lb.init(g.ldb, n0, g.n, kbeg, tid);
	la.load(g.a, ar, g.lda, m0, g.m, kbeg, kend, tid);
	lb.load(g.b, br, g.ldb, n0, g.n, kbeg, kend, tid);
	la.park(As[0], tid);
	lb.park(Bs[0], tid);
	__syncthreads();

	int buf = 0;
	for (int k0 = kbeg; k0 < kend; k0 += BK, buf ^= 1) {
		const bool more = k0 + BK < kend;
		if (more) {
			la.load(g.a, ar, g.lda, m0, g.m, k0 + BK, kend, tid);
			lb.load(g.b, br, g.ldb, n0, g.n, k0 + BK, kend, tid);
		}

		const float *as = As[buf] + wm * (BM / WM) + l31, *bs = Bs[buf] + wn * (BN / WN) + l31;
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

	// epilogue: lane = column, 16 rows per MFMA tile; split-K partials go to their slab untouched. The three cases (slab, beta = 0,
	// beta != 0) and whole / ragged tiles are told apart ONCE, outside the 64 stores of a lane (on short reductions the epilogue is a
	// fifth of the launch); same arithmetic as before: (beta == 0 ? 0 : beta * c) + alpha * acc
	const bool direct = g.splits == 1;
	float *out = direct ? g.c : g.c + (size_t)split * g.m * g.n;
	const int ldo = direct ? g.ldc : g.n;
	const bool whole = m0 + BM <= g.m && n0 + BN <= g.n;
	auto store_all = [&](auto mode, auto full) {
		constexpr int MODE = decltype(mode)::value;
		constexpr bool FULL = decltype(full)::value;
#pragma unroll
		for (int j = 0; j < TN; ++j) {
			const int n = n0 + wn * (BN / WN) + j * 32 + l31;
			if (!FULL && n >= g.n) continue;
#pragma unroll
			for (int i = 0; i < TM; ++i)
#pragma unroll
				for (int r = 0; r < 16; ++r) {
					const int m = m0 + wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
					if (FULL || m < g.m) {
						float *o = out + (size_t)m * ldo + n;
						if (MODE == 0) *o = acc[i][j][r];
						else if (MODE == 1) *o = 0.f + g.alpha * acc[i][j][r];
						else *o = g.beta * *o + g.alpha * acc[i][j][r];
					}
				}
		}
	};
	using M0 = std::integral_constant<int, 0>;
	using M1 = std::integral_constant<int, 1>;
	using M2 = std::integral_constant<int, 2>;
	if (!direct) whole ? store_all(M0{}, std::true_type{}) : store_all(M0{}, std::false_type{});
	else if (g.beta == 0.f) whole ? store_all(M1{}, std::true_type{}) : store_all(M1{}, std::false_type{});
	else whole ? store_all(M2{}, std::true_type{}) : store_all(M2{}, std::false_type{});
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

GemmPlan plan_gemm(int m, int n, int k, bool vec = true) {
	GemmPlan p;
	p.bm = m > 64 ? 128 : 64;
	p.bn = n > 64 ? 128 : 64;
	// prefer the smaller tile when the larger one leaves the chip mostly idle and pads a lot
	if (p.bm == 128 && pz::ceil_div(m, 128) * pz::ceil_div(n, p.bn) < pz::kNumCU / 2 && m % 128 != 0 && m % 128 <= 64) p.bm = 64;
	if (p.bn == 128 && pz::ceil_div(m, p.bm) * pz::ceil_div(n, 128) < pz::kNumCU / 2 && n % 128 != 0 && n % 128 <= 64) p.bn = 64;
	// one 256 x 256 tile per CU or more over a reduction of >= 1024 (eight per CU from 256): 16 waves per workgroup
	// (never split along K, with either tiling: the workspace does not depend on `vec`)
	// (on short reductions the big tile needs many rounds to pay: 1024 x 256 x 50176 = 3 tiles per CU runs 94 TFLOP/s on it, 114 on 128 x 128)
	const long tiles256 = (long)pz::ceil_div(m, 256) * pz::ceil_div(n, 256);
	if (vec && (k >= 1024 ? tiles256 >= pz::kNumCU : k >= 256 && tiles256 >= 8 * pz::kNumCU)) p.bm = p.bn = 256;
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

// false: the plan names a tile this loader has no kernel for (256 x 256 exists for the 16-byte loader only; plan_gemm(vec = false)
// never picks it) — the caller reports it instead of returning with C unwritten
template <bool TA, bool TB, bool VEC>
bool launch(const GemmPlan &p, const GemmArgs &g, hipStream_t st) {
	const dim3 grid(p.tiles_m * p.tiles_n, 1, p.splits);
	if (p.bm == 256) {
		if constexpr (VEC) gemm_kernel<256, 256, 4, 4, TA, TB, true><<<grid, 1024, 0, st>>>(g);
		else return false;
	} else if (p.bm == 128 && p.bn == 128) gemm_kernel<128, 128, 2, 2, TA, TB, VEC><<<grid, 256, 0, st>>>(g);
	else if (p.bm == 128) gemm_kernel<128, 64, 2, 2, TA, TB, VEC><<<grid, 256, 0, st>>>(g);
	else if (p.bn == 128) gemm_kernel<64, 128, 2, 2, TA, TB, VEC><<<grid, 256, 0, st>>>(g);
	else gemm_kernel<64, 64, 2, 2, TA, TB, VEC><<<grid, 256, 0, st>>>(g);
	return true;
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

	// 16-byte buffer loads: aligned operands, whole quads along K, extents a 32-bit byte offset can address
	const size_t a_ext = (trans_a ? (size_t)(k - 1) * lda + m : (size_t)(m - 1) * lda + k) * sizeof(float);
	const size_t b_ext = (trans_b ? (size_t)(n - 1) * ldb + k : (size_t)(k - 1) * ldb + n) * sizeof(float);
	const bool vec = ((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0) && lda % 4 == 0 && ldb % 4 == 0 && k % 4 == 0 &&
	                 a_ext < 0xffff0000ull && b_ext < 0xffff0000ull;

	GemmPlan p = plan_gemm(m, n, k, vec);
	const size_t need = p.splits > 1 ? (size_t)p.splits * m * n * sizeof(float) : 0;
	if (ws_bytes < need || (need > 0 && workspace == nullptr)) p.splits = 1, p.ksteps_per_split = pz::ceil_div(k, BK);   // no scratch: one pass over K

	GemmArgs g{a, b, p.splits > 1 ? (float *)workspace : c, m, n, k, lda, ldb, ldc, alpha, beta, p.tiles_m, p.tiles_n, p.splits,
	           p.ksteps_per_split, vec ? (unsigned)a_ext : 0u, vec ? (unsigned)b_ext : 0u};
	hipStream_t st = pz::as_stream(stream);

	bool launched;
	if (trans_a) launched = vec ? launch<true, false, true>(p, g, st) : launch<true, false, false>(p, g, st);
	else if (trans_b) launched = vec ? launch<false, true, true>(p, g, st) : launch<false, true, false>(p, g, st);
	else launched = vec ? launch<false, false, true>(p, g, st) : launch<false, false, false>(p, g, st);
	PZ_REQUIRE(launched, "pz_gemm: no kernel for a %d x %d tile with the 4-byte loader", p.bm, p.bn);
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
