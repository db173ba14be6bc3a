// Winograd F(4x4, 3x3) convolution for gfx950: the 3x3 / stride-1 / undilated / ungrouped forward and backward-data
// passes on maps large enough to fill 4x4 output tiles (ConvFwdAlgo.winograd of Hip/Wrappers/MIOpen.py:28). 4x fewer
// multiply-accumulates than the direct sum (F(2x2, 3x3) of wino.hip: 2.25x), all of them on v_mfma_f32_32x32x2_f32:
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      per 4x4 output tile, 6x6 input patch d, 3x3 filter g
//
// with the interpolation points 0, +-1, +-2, inf. Per output pixel everything but the transforms' own arithmetic shrinks by
// 36/16 : 16/4 = 0.56 against F(2x2): the multiplies, the patch elements gathered, the transformed values through LDS, the
// transformed filters read. The price is rounding: ~4e-6 relative L2 / 1e-5 of the output scale in fp32 (F(2x2): 5e-7), see
// DESIGN.md 3.1g and tests/test_gpu_0_ops.py::test_winograd_convolution.
//
// One launch; nothing transformed goes through HBM except the filters (36 values per 3x3 filter, pz_conv2d_prepack):
//   * a workgroup (4 waves, two workgroups per CU: 144 accumulator registers per lane) owns 32 tiles x 32 produced channels x
//     all 36 transform positions = 36 MFMA tiles; wave w accumulates positions 9w..9w+8;
//   * the reduction runs over chunks of 4 channels. Wave w gathers channel w of the chunk for all 32 tiles: a lane owns one
//     row of a 6x6 patch (b128 + b64 buffer loads, zero padding by out-of-range offsets, three rows per lane), applies
//     (.) B along the row and parks the result in a wave-private LDS block; then a lane owns one COLUMN of a patch, reads its
//     six values back, applies B^T and leaves V[pos][tile][c] in the stage the MFMAs of chunk + 2 will read (two stages of
//     18 KB) — the transform costs 18 long-lived registers instead of the 48 a whole half patch per lane would. A position
//     belongs to one wave, so the transformed filters never need LDS: every wave loads its own A fragments U[pos][k][c]
//     straight from global memory (L2-resident), and both fragment sets are reloaded in place — for the next chunk — right
//     after their last use; one barrier per chunk;
//   * the epilogue passes the accumulators through LDS 8 channels at a time so that one thread holds the 36 positions of a
//     (tile, channel), applies A^T . A, adds the bias, stores four rows of 16 bytes (lanes along the tile row) and, when
//     asked, leaves the shifted sums of its 32 tiles for the batch normalisation that follows (the strip format of wino.hip).
#include "common.h"

#include <cstdlib>
#include <type_traits>
#include <utility>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));


namespace {

constexpr unsigned kOOB = 0xffffff00u;      // buffer byte offset beyond every tensor (+16 does not wrap): loads return 0, stores are dropped

constexpr int TB = 32;       // tiles per workgroup
constexpr int KB = 32;       // produced channels per workgroup
constexpr int BC = 4;        // reduction channels per chunk
constexpr int NP = 36;       // transform positions
constexpr int kV = NP * TB * BC;     // floats of transformed patches per chunk (18 KB)
constexpr int kU = NP * KB * BC;     // floats of transformed filters per (channel block, chunk) (18 KB)

struct W4FilterArgs {
	const float *w;          // (K, C, R, R), R = 3 (F(4x4, 3x3)) or 5 (F(2x2, 5x5): the same six interpolation points)
	float *u;                // [kblocks][chunks][4 waves]{[4 position pairs][2][KB][2 positions][2], [2][KB][2]}: w4_u_index
	int mode;                // 0: forward (produced = K, reduction = C); 1: backward-data (produced = C, reduction = K, taps flipped)
	int K, C, R;             // dims of w
	int prod, red;
	int kblocks, chunks;
};

// Where U[pos][h][kk][ci] of a (channel block, chunk) lives, in floats: wave pos / 9 reads its nine positions as four 16-byte
// fragments (two positions each) and one 8-byte fragment per lane (h, kk)
__device__ __forceinline__ int w4_u_index(int pos, int h, int kk, int ci) {
	const int wave = pos / 9, local = pos % 9;
	const int base = wave * (9 * 2 * KB * 2);
	if (local < 8) return base + (((local >> 1) * 2 + h) * KB + kk) * 4 + (local & 1) * 2 + ci;
	return base + 4 * (2 * KB * 4) + (h * KB + kk) * 2 + ci;
}

// one row of G . (g0, g1, g2)
__device__ __forceinline__ void w4_g_row(float g0, float g1, float g2, float (&o)[6]) {
	const float s = g0 + g2;
	o[0] = 0.25f * g0;
	o[1] = (-1.f / 6.f) * (s + g1);
	o[2] = (-1.f / 6.f) * (s - g1);
	const float e = (1.f / 24.f) * g0 + (1.f / 6.f) * g2, f = (1.f / 12.f) * g1;
	o[3] = e + f;
	o[4] = e - f;
	o[5] = g2;
}

// one row of G5 . (g0 .. g4): the 6 x 5 matrix of F(2, 5) on the points 0, +-1, +-2, inf — row i = c_i (1, p_i, p_i^2, p_i^3, p_i^4)
// with the c_i of w4_g_row (they depend on the points alone), last row (0, 0, 0, 0, 1)
__device__ __forceinline__ void w5_g_row(float g0, float g1, float g2, float g3, float g4, float (&o)[6]) {
	const float ev = g0 + g2 + g4, od = g1 + g3;
	o[0] = 0.25f * g0;
	o[1] = (-1.f / 6.f) * (ev + od);
	o[2] = (-1.f / 6.f) * (ev - od);
	const float e = (1.f / 24.f) * g0 + (1.f / 6.f) * g2 + (2.f / 3.f) * g4, f = (1.f / 12.f) * g1 + (1.f / 3.f) * g3;
	o[3] = e + f;
	o[4] = e - f;
	o[5] = g4;
}

template <int R>
__device__ __forceinline__ void w4_filter_body_r(const W4FilterArgs &a, long first, long step);

__device__ __forceinline__ void w4_filter_body(const W4FilterArgs &a, long first, long step) {
	if (a.R == 5) {
		w4_filter_body_r<5>(a, first, step);
		return;
	}
	const long total = (long)a.kblocks * a.chunks * KB * BC;
	for (long i = first; i < total; i += step) {
		const int ci = (int)(i % 2), kk = (int)((i / 2) % KB), h = (int)((i / (2 * KB)) % 2);
		const long blk = i / (2 * KB * 2);
		const int chunk = (int)(blk % a.chunks), kb = (int)(blk / a.chunks);
		const int k = kb * KB + kk, c = chunk * BC + h * 2 + ci;

		float g[3][3];
#pragma unroll
		for (int r = 0; r < 3; ++r)
#pragma unroll
			for (int s = 0; s < 3; ++s) {
				float v = 0.f;
				if (k < a.prod && c < a.red)
					v = a.mode == 0 ? a.w[((long)k * a.C + c) * 9 + r * 3 + s] : a.w[((long)c * a.C + k) * 9 + (2 - r) * 3 + (2 - s)];
				g[r][s] = v;
			}

		float t[3][6];           // t[s][r] = (G g)[r][s]
#pragma unroll
		for (int s = 0; s < 3; ++s) w4_g_row(g[0][s], g[1][s], g[2][s], t[s]);
		float *dst = a.u + blk * kU;
#pragma unroll
		for (int r = 0; r < 6; ++r) {
			float u[6];
			w4_g_row(t[0][r], t[1][r], t[2][r], u);
#pragma unroll
			for (int s = 0; s < 6; ++s) dst[w4_u_index(r * 6 + s, h, kk, ci)] = u[s];
		}
	}
}

template <int R>
__device__ __forceinline__ void w4_filter_body_r(const W4FilterArgs &a, long first, long step) {
	static_assert(R == 5, "the 3-tap form is w4_filter_body itself");
	const long total = (long)a.kblocks * a.chunks * KB * BC;
	for (long i = first; i < total; i += step) {
		const int ci = (int)(i % 2), kk = (int)((i / 2) % KB), h = (int)((i / (2 * KB)) % 2);
		const long blk = i / (2 * KB * 2);
		const int chunk = (int)(blk % a.chunks), kb = (int)(blk / a.chunks);
		const int k = kb * KB + kk, c = chunk * BC + h * 2 + ci;

		float g[R][R];
#pragma unroll
		for (int r = 0; r < R; ++r)
#pragma unroll
			for (int t = 0; t < R; ++t) {
				float v = 0.f;
				if (k < a.prod && c < a.red)
					v = a.mode == 0 ? a.w[((long)k * a.C + c) * (R * R) + r * R + t] : a.w[((long)c * a.C + k) * (R * R) + (R - 1 - r) * R + (R - 1 - t)];
				g[r][t] = v;
			}

		float t6[R][6];          // t6[s][r] = (G g)[r][s]
#pragma unroll
		for (int t = 0; t < R; ++t) w5_g_row(g[0][t], g[1][t], g[2][t], g[3][t], g[4][t], t6[t]);
		float *dst = a.u + blk * kU;
#pragma unroll
		for (int r = 0; r < 6; ++r) {
			float u[6];
			w5_g_row(t6[0][r], t6[1][r], t6[2][r], t6[3][r], t6[4][r], u);
#pragma unroll
			for (int t = 0; t < 6; ++t) dst[w4_u_index(r * 6 + t, h, kk, ci)] = u[t];
		}
	}
}

__global__ void __launch_bounds__(256) wino4_filter_kernel(W4FilterArgs a) {
	w4_filter_body(a, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

struct W4FilterBatch {
	int n, start[pz::kWinoBatch + 1];
	W4FilterArgs job[pz::kWinoBatch];
};
static_assert(sizeof(W4FilterBatch) <= 4000, "kernel arguments");

__global__ void __launch_bounds__(256) wino4_filter_batch_kernel(W4FilterBatch b) {
	int j = 0;
	while (j + 1 < b.n && (int)blockIdx.x >= b.start[j + 1]) ++j;
	const int nb = b.start[j + 1] - b.start[j];
	w4_filter_body(b.job[j], (long)(blockIdx.x - b.start[j]) * 256 + threadIdx.x, (long)nb * 256);
}

struct W4Args {
	const float *x;          // gathered tensor (N, C, H, W)
	const float *u;          // transformed filters
	const float *bias;       // per produced channel or NULL
	float *y;                // (N, K, P, Q)
	int N, C, H, W, K, P, Q;
	int pad_h, pad_w;
	int TY, TX, tiles;       // output tiles per image column / row, N*TY*TX
	int ts;                  // outputs per tile side = tile step: 4 (3x3 filters) or 2 (5x5 filters); the patch is 6x6 either way
	int chunks, tblocks, kblocks;
	unsigned x_bytes, y_bytes;
	float4 *stats;           // optional [K][tblocks] {shift, sum(v - shift), sum((v - shift)^2), count}
	// backward-data launches (pz_conv2d_bwd_data_bnstats, see BnStatsOut in common.h): y is the gradient w.r.t. relu(gab.x * gx + gab.y);
	// per produced channel and tile block {sum q, sum q (gx - gmean)} with q = y * (relu(..) > 0) to gst[k * tblocks + tb]
	const float *gx;
	const float2 *gab;
	const float *gmean;
	float2 *gst;
	const float *v;          // PRE: transformed patches [tblocks][chunks][kV] in the order of an LDS stage (wino4_input_kernel)
};

template <int V>
using IC = std::integral_constant<int, V>;

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
	(f(IC<I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
	static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// B^T applied to six values along one axis
__device__ __forceinline__ void w4_bt(const float (&d)[6], float (&o)[6]) {
	o[0] = __builtin_fmaf(4.f, d[0], __builtin_fmaf(-5.f, d[2], d[4]));
	o[5] = __builtin_fmaf(4.f, d[1], __builtin_fmaf(-5.f, d[3], d[5]));
	const float u = __builtin_fmaf(-4.f, d[2], d[4]), v = __builtin_fmaf(-4.f, d[1], d[3]);
	o[1] = u + v, o[2] = u - v;
	const float p = d[4] - d[2], q = d[3] - d[1];
	o[3] = __builtin_fmaf(2.f, q, p), o[4] = __builtin_fmaf(-2.f, q, p);
}

// A^T applied to six values: four results
__device__ __forceinline__ void w4_at(const float (&m)[6], float (&o)[4]) {
	const float p12 = m[1] + m[2], d12 = m[1] - m[2], p34 = m[3] + m[4], d34 = m[3] - m[4];
	o[0] = m[0] + p12 + p34;
	o[1] = __builtin_fmaf(2.f, d34, d12);
	o[2] = __builtin_fmaf(4.f, p34, p12);
	o[3] = __builtin_fmaf(8.f, d34, d12) + m[5];
}

// ---- the patches' transform as a pass of its own (PRE): V = B^T d B of every (tile, channel), written once in the order of
// the main kernel's LDS stages — [tile block][chunk]{[position][channel pair][tile][channel of the pair]} — so that the main
// kernel's workgroups (one per 32 produced channels: 8-16 of them share a tile block) COPY a chunk instead of each gathering
// and transforming it: the main loop sheds its ~90 vector and ~30 LDS instructions per wave and chunk, which at two waves per
// SIMD do not hide behind the MFMAs. An experiment that did NOT pay (see wino4_input_bytes: off by default).
// Same operations in the same order as the fused kernel's row / column passes: bit-identical results.
// Block = 256 threads = 32 tiles x 4 channels x 2 chunks; lane = (tile, channel parity): a wave stores 256 contiguous bytes.
__global__ void __launch_bounds__(256) wino4_input_kernel(W4Args a, float *__restrict__ vout) {
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int pairs = (a.chunks + 1) / 2;
	const int tb = blockIdx.x / pairs, chunk = (blockIdx.x % pairs) * 2 + (wave >> 1);
	if (chunk >= a.chunks) return;
	const int chalf = wave & 1, j = lane >> 1, cpar = lane & 1;
	const int ch = chunk * BC + chalf * 2 + cpar;

	const int t = tb * TB + j;
	const bool tv = t < a.tiles;
	const int n = t / (a.TY * a.TX), r0 = t - n * (a.TY * a.TX);
	const int ty = r0 / a.TX, tx = r0 - ty * a.TX;
	const int row0 = a.ts * ty - a.pad_h, col0 = a.ts * tx - a.pad_w;
	const bool inner = col0 >= 0 && col0 + 5 < a.W;
	const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.x_bytes, 0x00020000);

	float rt[6][6];                             // rt[e][jj] = (d B)[e][jj]: row e of the patch through B^T along the row
#pragma unroll
	for (int e = 0; e < 6; ++e) {
		const int row = row0 + e;
		const bool ok = tv && (unsigned)row < (unsigned)a.H;
		float d[6];
		if (inner) {
			const unsigned off = ok ? (unsigned)(((((long)n * a.C + ch) * a.H + row) * a.W + col0) * 4) : kOOB;
			const f32x4 lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0));
			const f32x2 hi = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr, off + 16u, 0, 0));
			d[0] = lo[0], d[1] = lo[1], d[2] = lo[2], d[3] = lo[3], d[4] = hi[0], d[5] = hi[1];
		} else {
#pragma unroll
			for (int c = 0; c < 6; ++c) {
				const int col = col0 + c;
				const bool cok = ok && (unsigned)col < (unsigned)a.W;
				const unsigned off = cok ? (unsigned)(((((long)n * a.C + ch) * a.H + row) * a.W + col) * 4) : kOOB;
				d[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, off, 0, 0));
			}
		}
		w4_bt(d, rt[e]);
	}

	float *dst = vout + ((size_t)tb * a.chunks + chunk) * kV + (chalf * TB + j) * 2 + cpar;
#pragma unroll
	for (int c = 0; c < 6; ++c) {
		const float col[6] = {rt[0][c], rt[1][c], rt[2][c], rt[3][c], rt[4][c], rt[5][c]};
		float o[6];
		w4_bt(col, o);
#pragma unroll
		for (int i = 0; i < 6; ++i) dst[(i * 6 + c) * (TB * BC)] = o[i];
	}
}

__device__ __forceinline__ void w4_epilogue(const W4Args &a, f32x16 (&acc)[9], float *Ms, int kb, int tb, int tid, int pbase, int lane) {
	const int l31 = lane & 31, lhi = lane >> 5;
	const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void *)a.y, 0, a.y_bytes, 0x00020000);

	const int t = tb * TB + l31;
	const bool tv = t < a.tiles;
	const int n = t / (a.TY * a.TX), rr = t - n * (a.TY * a.TX);
	const int ty = rr / a.TX, tx = rr - ty * a.TX;
	bool rowok[4], colok[4];
#pragma unroll
	for (int i = 0; i < 4; ++i) rowok[i] = 4 * ty + i < a.P, colok[i] = 4 * tx + i < a.Q;
	const bool wide = __builtin_amdgcn_readfirstlane((int)(__builtin_amdgcn_ballot_w64(tv && !colok[3]) == 0ull)) != 0;
	const unsigned pq4 = (unsigned)(a.P * a.Q) * 4u, q4 = (unsigned)a.Q * 4u;
	const unsigned obase = (unsigned)((((long)n * a.K) * a.P + 4 * ty) * a.Q + 4 * tx) * 4u;
	const int kk = tid >> 5;                           // 256 threads = 32 tiles x 8 channels

#pragma unroll
	for (int qq = 0; qq < 4; ++qq) {
		// accumulator rows 8 qq .. 8 qq + 7 (registers 4 qq .. 4 qq + 3 of both lane halves) -> Ms[position][8 channels][32 tiles]
#pragma unroll
		for (int i = 0; i < 9; ++i)
#pragma unroll
			for (int r = 0; r < 4; ++r) Ms[((pbase + i) * 8 + 4 * lhi + r) * 32 + l31] = acc[i][4 * qq + r];
		__syncthreads();

		{
			const int k = kb * KB + qq * 8 + kk;
			float s[4][6];                             // A^T m, one column of m at a time
#pragma unroll
			for (int c = 0; c < 6; ++c) {
				float m[6], o[4];
#pragma unroll
				for (int r = 0; r < 6; ++r) m[r] = Ms[((r * 6 + c) * 8 + kk) * 32 + l31];
				w4_at(m, o);
#pragma unroll
				for (int i = 0; i < 4; ++i) s[i][c] = o[i];
			}
			const float b = (a.bias != nullptr && k < a.K) ? a.bias[k] : 0.f;
			const bool kv = tv && k < a.K;
			const unsigned o = obase + (unsigned)k * pq4;

			float shift = 0.f, s1 = 0.f, s2 = 0.f, cnt = 0.f;
			float g1 = 0.f, g2 = 0.f, ga = 0.f, gb = 0.f, gmu = 0.f;
			if (a.gst) {                               // (wave-uniform)
				const float2 ab = a.gab[min(k, a.K - 1)];
				ga = ab.x, gb = ab.y, gmu = a.gmean[min(k, a.K - 1)];
			}
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				float y[4];
				w4_at(s[i], y);
#pragma unroll
				for (int j = 0; j < 4; ++j) y[j] += b;

				const bool rv = kv && rowok[i];
				const unsigned orow = o + (unsigned)i * q4;
				if (a.gst) {
					// the BatchNorm input at the very addresses this row is stored to (same shape as y): gate with the forward's own
					// fma (bn_gate<true>'s form), sum the gated gradient and its product with the centred input
					const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc((void *)a.gx, 0, a.y_bytes, 0x00020000);
					float xv[4];
					if (wide) {
						const f32x4 t4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(gr, rv && colok[3] ? orow : kOOB, 0, 0));
						xv[0] = t4[0], xv[1] = t4[1], xv[2] = t4[2], xv[3] = t4[3];
					} else {
#pragma unroll
						for (int j = 0; j < 4; ++j)
							xv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gr, rv && colok[j] ? orow + 4u * j : kOOB, 0, 0));
					}
#pragma unroll
					for (int j = 0; j < 4; ++j) {
						const bool ok = rv && colok[j];
						const float q = (ok && __builtin_fmaf(xv[j], ga, gb) > 0.f) ? y[j] : 0.f;
						g1 += q;
						g2 = __builtin_fmaf(q, xv[j] - gmu, g2);
					}
				}
				// whole tile rows as 16 bytes; the tile at the right edge of a map whose width is no multiple of 4 word by word
				__builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{y[0], y[1], y[2], y[3]}), yr, rv && colok[3] ? orow : kOOB, 0, 0);
				if (!wide) {
					const bool edge = rv && !colok[3];
					__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y[0]), yr, edge && colok[0] ? orow : kOOB, 0, 0);
					__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y[1]), yr, edge && colok[1] ? orow + 4u : kOOB, 0, 0);
					__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y[2]), yr, edge && colok[2] ? orow + 8u : kOOB, 0, 0);
				}

				if (a.stats) {
					// the 32 lanes of a half-wave hold this channel's 32 tiles: shifted sums about the block's first output
					// (tile tb*TB always exists), fixed shuffle tree -> deterministic
					if (i == 0) shift = __shfl(y[0], lane & 32);
#pragma unroll
					for (int j = 0; j < 4; ++j) {
						const bool ok = rv && colok[j];
						const float dlt = ok ? y[j] - shift : 0.f;
						s1 += dlt, s2 = __builtin_fmaf(dlt, dlt, s2), cnt += ok ? 1.f : 0.f;
					}
				}
			}
			if (a.stats) {
#pragma unroll
				for (int msk = 16; msk > 0; msk >>= 1) s1 += __shfl_xor(s1, msk), s2 += __shfl_xor(s2, msk), cnt += __shfl_xor(cnt, msk);
				if (l31 == 0 && k < a.K) a.stats[(size_t)k * a.tblocks + tb] = make_float4(shift, s1, s2, cnt);
			}
			if (a.gst) {                               // the same fixed tree over the half-wave's 32 tiles
#pragma unroll
				for (int msk = 16; msk > 0; msk >>= 1) g1 += __shfl_xor(g1, msk), g2 += __shfl_xor(g2, msk);
				if (l31 == 0 && k < a.K) a.gst[(size_t)k * a.tblocks + tb] = make_float2(g1, g2);
			}
		}
		if (qq < 3) __syncthreads();
	}
}

// A^T of F(2, 5) applied to six values: two results (rows (1 1 1 1 1 0) and (0 1 -1 2 -2 1) — the first two rows of w4_at's matrix,
// the point at infinity moved to the last row there is)
__device__ __forceinline__ void w5_at(const float (&m)[6], float (&o)[2]) {
	const float p12 = m[1] + m[2], d12 = m[1] - m[2], p34 = m[3] + m[4], d34 = m[3] - m[4];
	o[0] = m[0] + p12 + p34;
	o[1] = __builtin_fmaf(2.f, d34, d12) + m[5];
}

// The epilogue of the 2x2-output form (5x5 filters): w4_epilogue with two output rows of two columns per tile — 8-byte stores, the
// tile at the right edge of an odd map word by word; statistics blocks as there.
__device__ __forceinline__ void w5_epilogue(const W4Args &a, f32x16 (&acc)[9], float *Ms, int kb, int tb, int tid, int pbase, int lane) {
	const int l31 = lane & 31, lhi = lane >> 5;
	const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void *)a.y, 0, a.y_bytes, 0x00020000);

	const int t = tb * TB + l31;
	const bool tv = t < a.tiles;
	const int n = t / (a.TY * a.TX), rr = t - n * (a.TY * a.TX);
	const int ty = rr / a.TX, tx = rr - ty * a.TX;
	bool rowok[2], colok[2];
#pragma unroll
	for (int i = 0; i < 2; ++i) rowok[i] = 2 * ty + i < a.P, colok[i] = 2 * tx + i < a.Q;
	const bool wide = __builtin_amdgcn_readfirstlane((int)(__builtin_amdgcn_ballot_w64(tv && !colok[1]) == 0ull)) != 0;
	const unsigned pq4 = (unsigned)(a.P * a.Q) * 4u, q4 = (unsigned)a.Q * 4u;
	const unsigned obase = (unsigned)((((long)n * a.K) * a.P + 2 * ty) * a.Q + 2 * tx) * 4u;
	const int kk = tid >> 5;                           // 256 threads = 32 tiles x 8 channels

#pragma unroll
	for (int qq = 0; qq < 4; ++qq) {
#pragma unroll
		for (int i = 0; i < 9; ++i)
#pragma unroll
			for (int r = 0; r < 4; ++r) Ms[((pbase + i) * 8 + 4 * lhi + r) * 32 + l31] = acc[i][4 * qq + r];
		__syncthreads();

		{
			const int k = kb * KB + qq * 8 + kk;
			float s[2][6];                             // A^T m, one column of m at a time
#pragma unroll
			for (int c = 0; c < 6; ++c) {
				float m[6], o[2];
#pragma unroll
				for (int r = 0; r < 6; ++r) m[r] = Ms[((r * 6 + c) * 8 + kk) * 32 + l31];
				w5_at(m, o);
				s[0][c] = o[0], s[1][c] = o[1];
			}
			const float b = (a.bias != nullptr && k < a.K) ? a.bias[k] : 0.f;
			const bool kv = tv && k < a.K;
			const unsigned o = obase + (unsigned)k * pq4;

			float shift = 0.f, s1 = 0.f, s2 = 0.f, cnt = 0.f;
			float g1 = 0.f, g2 = 0.f, ga = 0.f, gb = 0.f, gmu = 0.f;
			if (a.gst) {                               // (wave-uniform)
				const float2 ab = a.gab[min(k, a.K - 1)];
				ga = ab.x, gb = ab.y, gmu = a.gmean[min(k, a.K - 1)];
			}
#pragma unroll
			for (int i = 0; i < 2; ++i) {
				float y[2];
				w5_at(s[i], y);
				y[0] += b, y[1] += b;

				const bool rv = kv && rowok[i];
				const unsigned orow = o + (unsigned)i * q4;
				if (a.gst) {
					const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc((void *)a.gx, 0, a.y_bytes, 0x00020000);
					float xv[2];
					if (wide) {
						const f32x2 t2 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(gr, rv && colok[1] ? orow : kOOB, 0, 0));
						xv[0] = t2[0], xv[1] = t2[1];
					} else {
#pragma unroll
						for (int j = 0; j < 2; ++j)
							xv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gr, rv && colok[j] ? orow + 4u * j : kOOB, 0, 0));
					}
#pragma unroll
					for (int j = 0; j < 2; ++j) {
						const bool ok = rv && colok[j];
						const float q = (ok && __builtin_fmaf(xv[j], ga, gb) > 0.f) ? y[j] : 0.f;
						g1 += q;
						g2 = __builtin_fmaf(q, xv[j] - gmu, g2);
					}
				}
				__builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{y[0], y[1]}), yr, rv && colok[1] ? orow : kOOB, 0, 0);
				if (!wide)
					__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y[0]), yr, rv && !colok[1] && colok[0] ? orow : kOOB, 0, 0);

				if (a.stats) {
					if (i == 0) shift = __shfl(y[0], lane & 32);
#pragma unroll
					for (int j = 0; j < 2; ++j) {
						const bool ok = rv && colok[j];
						const float dlt = ok ? y[j] - shift : 0.f;
						s1 += dlt, s2 = __builtin_fmaf(dlt, dlt, s2), cnt += ok ? 1.f : 0.f;
					}
				}
			}
			if (a.stats) {
#pragma unroll
				for (int msk = 16; msk > 0; msk >>= 1) s1 += __shfl_xor(s1, msk), s2 += __shfl_xor(s2, msk), cnt += __shfl_xor(cnt, msk);
				if (l31 == 0 && k < a.K) a.stats[(size_t)k * a.tblocks + tb] = make_float4(shift, s1, s2, cnt);
			}
			if (a.gst) {
#pragma unroll
				for (int msk = 16; msk > 0; msk >>= 1) g1 += __shfl_xor(g1, msk), g2 += __shfl_xor(g2, msk);
				if (l31 == 0 && k < a.K) a.gst[(size_t)k * a.tblocks + tb] = make_float2(g1, g2);
			}
		}
		if (qq < 3) __syncthreads();
	}
}

// M = outputs per tile side: 4 (3x3 filters, w4_epilogue) or 2 (5x5 filters, w5_epilogue); everything in front of the epilogue —
// gathers, the B^T d B transform of the 6x6 patches, the 36 position products — is the same code (the tile step is W4Args::ts)
template <bool PRE, int M = 4>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) wino4_conv_kernel(W4Args a) {
	// two stages of V, then the waves' private blocks of row-transformed patches; the epilogue's block reuses all of it
	__shared__ __attribute__((aligned(16))) float smem[3 * kV];                  // 54 KB

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int l31 = lane & 31, lhi = lane >> 5;
	// block b runs on XCD b % 8: the channel blocks of one tile block go side by side to one XCD and share its patches in that
	// XCD's L2 (-3 % on the 55x55 and 28x28 layers; the grid is padded to whole groups of 8 tile blocks)
	const int xl = blockIdx.x >> 3, kb = xl % a.kblocks, tb = (xl / a.kblocks) * 8 + (blockIdx.x & 7);
	if (tb >= a.tblocks) return;

	const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.x_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(
	    (void *)(a.u + (size_t)kb * a.chunks * kU), 0, (unsigned)a.chunks * (kU * 4u), 0x00020000);
	const unsigned hw4 = (unsigned)(a.H * a.W) * 4u;

	// ---- patch role: channel `wave` of the chunk, tile l31; round r (0-2): image row 2 r + lhi of the patch, then column 2 r + lhi
	unsigned voff[3];                           // byte offsets of the three rows (below zero as two's complement: see the fixed variant)
	bool colok[6];
	bool anyfix = false;
	{
		const int t = tb * TB + l31;
		const bool tv = t < a.tiles;
		const int n = t / (a.TY * a.TX), r0 = t - n * (a.TY * a.TX);
		const int ty = r0 / a.TX, tx = r0 - ty * a.TX;
		const int col0 = a.ts * tx - a.pad_w;
#pragma unroll
		for (int r = 0; r < 3; ++r) {
			const int row = a.ts * ty - a.pad_h + 2 * r + lhi;
			const bool ok = tv && (unsigned)row < (unsigned)a.H;
			const long off = (((long)n * a.C + wave) * a.H + row) * a.W + col0;
			voff[r] = ok ? (unsigned)(off * 4) : kOOB;
			anyfix = anyfix || (ok && off < 0);
		}
#pragma unroll
		for (int j = 0; j < 6; ++j) colok[j] = (unsigned)(col0 + j) < (unsigned)a.W;
	}
	anyfix = __builtin_amdgcn_readfirstlane((int)(__builtin_amdgcn_ballot_w64(anyfix) != 0ull)) != 0;
	float *Rs = smem + 2 * kV + wave * (NP * TB) + l31;                                 // + (e * 6 + j) * 32
	const unsigned vdst = (unsigned)(((wave >> 1) * TB + l31) * 2 + (wave & 1));        // + pos * (TB * BC)

	// ---- MFMA role: positions 9 wave .. 9 wave + 8
	const int pbase = 9 * wave;
	const unsigned vfrag = (unsigned)(((pbase * 2 + lhi) * TB + l31) * 2);
	const unsigned uvoff = (unsigned)((lhi * KB + l31) * 16);      // within a pair's 1 KB; the ninth position's 8-byte fragments: half of it
	const unsigned uwave = (unsigned)(wave * (9 * 2 * KB * 2 * 4));

	f32x4 sa[3];
	f32x2 sb[3];                                // staged patch rows (6 floats each)
	float rt[6];                                // a transformed row / column on its way to LDS

	// PRE: a chunk of transformed patches is a plain 18 KB copy: four 16-byte and one 8-byte piece per thread
	f32x4 pv[4];
	f32x2 pt;
	const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(
	    (void *)(PRE ? a.v + (size_t)tb * a.chunks * kV : a.x), 0, PRE ? (unsigned)a.chunks * (kV * 4u) : 0u, 0x00020000);
	auto pre_load = [&](int chunk) {
		const unsigned soff = (unsigned)chunk * (kV * 4u);
#pragma unroll
		for (int i = 0; i < 4; ++i) pv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(vr, (unsigned)(tid + 256 * i) * 16u, soff, 0));
		pt = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(vr, 16384u + (unsigned)tid * 8u, soff, 0));
	};
	auto pre_park = [&](float *stg) {
#pragma unroll
		for (int i = 0; i < 4; ++i) reinterpret_cast<f32x4 *>(stg)[tid + 256 * i] = pv[i];
		reinterpret_cast<f32x2 *>(stg + 4096)[tid] = pt;
	};
	static_assert(kV == 4 * 256 * 4 + 256 * 2, "the copy covers a stage exactly");

	// row `r` of the next but two chunk. The one workgroup whose first patch starts in front of the tensor loads word by word
	// and turns offsets below zero into out-of-range ones: they read as the padding they are.
	auto issue_row = [&](auto fixed, auto rc, int chunk) {
		constexpr int R = decltype(rc)::value;
		const unsigned soff = (unsigned)chunk * (BC * hw4);
		if constexpr (decltype(fixed)::value) {
			float v[6];
#pragma unroll
			for (int j = 0; j < 6; ++j) {
				const int o = (int)voff[R] + 4 * j;              // this workgroup's rows sit in the tensor's first image: offsets fit an int
				v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, o < 0 ? kOOB : (unsigned)o, soff, 0));
			}
			sa[R] = f32x4{v[0], v[1], v[2], v[3]}, sb[R] = f32x2{v[4], v[5]};
		} else {
			sa[R] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, voff[R], soff, 0));
			sb[R] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr, voff[R] + 16u, soff, 0));
		}
	};

	// slot S (0-17, one per MFMA of the chunk) of the transform of the staged rows into stage `stg`:
	//   2 r: row r (.) B;  2 r + 1: parked in the wave's block, the row of chunk `reload` requested;
	//   6 + 3 c .. 8 + 3 c: column c read back, B^T, stored as V
	auto patch_slot = [&](auto fixed, auto slot, float *stg, int reload) {
		constexpr int S = decltype(slot)::value;
		if constexpr (S < 6 && S % 2 == 0) {
			constexpr int R = S / 2;
			float d[6];
#pragma unroll
			for (int j = 0; j < 6; ++j) d[j] = colok[j] ? (j < 4 ? sa[R][j & 3] : sb[R][j & 1]) : 0.f;
			w4_bt(d, rt);
		} else if constexpr (S < 6) {
			constexpr int R = S / 2;
#pragma unroll
			for (int j = 0; j < 6; ++j) Rs[(((2 * R) * 6 + j) * 32) + lhi * (6 * 32)] = rt[j];
			if (reload >= 0) issue_row(fixed, IC<R>{}, reload);
		} else if constexpr (S < 15) {
			constexpr int Cc = (S - 6) / 3, G = (S - 6) % 3;
			if constexpr (G == 0) {
#pragma unroll
				for (int e = 0; e < 6; ++e) rt[e] = Rs[(e * 6 + 2 * Cc) * 32 + lhi * 32];
			} else if constexpr (G == 1) {
				float o[6];
				w4_bt(rt, o);
#pragma unroll
				for (int e = 0; e < 6; ++e) rt[e] = o[e];
			} else {
				float *dst = stg + vdst + (unsigned)(2 * Cc) * (TB * BC) + lhi * (TB * BC);
#pragma unroll
				for (int i = 0; i < 6; ++i) dst[(i * 6) * (TB * BC)] = rt[i];
			}
		}
	};

	f32x16 acc[9];
	f32x2 bv[9], avs;
	f32x4 avp[4];                               // A fragments of positions (2 s, 2 s + 1), both reduction steps; avs: position 8

	auto load_b = [&](const float *stg, auto ic) {
		constexpr int I = decltype(ic)::value;
		bv[I] = *reinterpret_cast<const f32x2 *>(stg + vfrag + I * (2 * TB * 2));
	};
	auto load_a = [&](auto sc, int chunk) {            // s = 0-3: a pair of positions, 4: the ninth
		constexpr int S = decltype(sc)::value;
		const unsigned soff = (unsigned)chunk * (kU * 4u) + uwave;
		if constexpr (S < 4)
			avp[S] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ur, uvoff, soff + (unsigned)(S * 2 * KB * 16), 0));
		else
			avs = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(ur, uvoff >> 1, soff + 4u * (2 * KB * 16), 0));
	};
	auto a_of = [&](auto ic, auto s2c) -> float {
		constexpr int I = decltype(ic)::value, S2 = decltype(s2c)::value;
		if constexpr (I < 8)
			return avp[I / 2][(I & 1) * 2 + S2];
		else
			return avs[S2];
	};

	auto run = [&](auto fixed) {
		const int last = a.chunks - 1;

		// ---- prologue: V(0), V(1) into the two stages, the fragments of chunk 0 into registers
		static_for<5>([&](auto sc) { load_a(sc, 0); });
		if constexpr (PRE) {
			pre_load(0);
			pre_park(smem);
			pre_load(min(1, last));
			pre_park(smem + kV);
		} else {
			static_for<3>([&](auto rc) { issue_row(fixed, rc, 0); });
			static_for<15>([&](auto sc) { patch_slot(fixed, sc, smem, min(1, last)); });
			static_for<15>([&](auto sc) { patch_slot(fixed, sc, smem + kV, min(2, last)); });
		}
#pragma unroll
		for (int i = 0; i < 9; ++i)
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
		__syncthreads();
		static_for<9>([&](auto ic) { load_b(smem, ic); });
		__syncthreads();

		for (int ch = 0; ch <= last; ++ch) {
			const float *rd = smem + ((ch + 1) & 1) * kV;        // V(ch + 1): fragments for the next chunk
			float *wr = smem + (ch & 1) * kV;                    // V(ch + 2) goes where V(ch) was
			const int nxt = min(ch + 1, last), reload = min(ch + 3, last);

			static_for<3>([&](auto gc) {
				constexpr int G = decltype(gc)::value;
				static_for<6>([&](auto jc) {
					constexpr int J = decltype(jc)::value, I = 3 * G + J % 3, S2 = J / 3;
					acc[I] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_of(IC<I>{}, IC<S2>{}), bv[I][S2], acc[I], 0, 0, 0);
					if constexpr (PRE) {
						// V(ch + 2): requested behind the chunk's first MFMA, parked behind its last (a chunk of MFMAs covers the latency)
						if constexpr (6 * G + J == 0) pre_load(min(ch + 2, last));
						if constexpr (6 * G + J == 17) pre_park(wr);
					} else {
						patch_slot(fixed, IC<6 * G + J>{}, wr, reload);
					}
					__builtin_amdgcn_sched_barrier(0);
				});
				// the fragments are reloaded in place for chunk ch + 1 once their positions are through
				static_for<3>([&](auto jc) { load_b(rd, IC<3 * G + decltype(jc)::value>{}); });
				if constexpr (G == 0) {
					load_a(IC<0>{}, nxt);
				} else if constexpr (G == 1) {
					load_a(IC<1>{}, nxt);
					load_a(IC<2>{}, nxt);
				} else {
					load_a(IC<3>{}, nxt);
					load_a(IC<4>{}, nxt);
				}
			});
			__syncthreads();
		}

		if constexpr (M == 4)
			w4_epilogue(a, acc, smem, kb, tb, tid, pbase, lane);
		else
			w5_epilogue(a, acc, smem, kb, tb, tid, pbase, lane);
	};

	if constexpr (PRE) {
		run(IC<0>{});
	} else {
		if (anyfix)
			run(IC<1>{});
		else
			run(IC<0>{});
	}
}


}  // namespace

namespace pz {

static void w4_dims(const pz_conv_desc *d, int which, int *prod, int *red) {
	*prod = which == PZ_CONV_FWD ? d->k : d->c;
	*red = which == PZ_CONV_FWD ? d->c : d->k;
}

// F(4x4) where it multiplies less than F(2x2) once the ragged tiles at the right / bottom edge are counted
// (36 per 16 outputs against 16 per 4), on the same layers wino_eligible admits
int g_wino_tile = 0;      // pz_conv_winograd_tile_set

// outputs per tile side of this file's kernel for a layer: 4 under 3x3 filters, 2 under 5x5 filters (same 6x6 patches, same 36
// positions: F(2x2, 5x5) multiplies 36 times per 4 outputs where the direct sum needs 100)
static int w4_tile(const pz_conv_desc *d) { return d->r == 5 ? 2 : 4; }

bool wino4_pick(const pz_conv_desc *d, int which, int P, int Q) {
	if (d->r == 5) return true;          // (5x5 layers have no other Winograd form: wino_eligible decides whether they come here at all)
	const int mode = g_wino_tile;
	if (mode == 2) return false;
	const int OP = which == PZ_CONV_FWD ? P : d->h, OQ = which == PZ_CONV_FWD ? Q : d->w;      // produced map
	if (mode == 4) return true;
	const double c4 = 36.0 * ((OP + 3) / 4) * ((OQ + 3) / 4), c2 = 16.0 * ((OP + 1) / 2) * ((OQ + 1) / 2);
	if (!(c4 < 0.9 * c2)) return false;
	// ... and where the launch has a workgroup for every CU: four times fewer tiles make a small problem (NiN's 192-channel
	// 8x8 layer at batch 128: 96 workgroups) a few long serial chains — 0.055 ms against F(2x2)'s 0.041
	int prod, red;
	w4_dims(d, which, &prod, &red);
	const long wgs = (long)ceil_div((long)d->n * ((OP + 3) / 4) * ((OQ + 3) / 4), TB) * ceil_div(prod, KB);
	return wgs >= kNumCU;
}

// The patches' transform as its own pass (PRE), for launches whose transformed input is at most PUZZLE_MI355_WINO_PRE_MB
// megabytes. OFF by default (0): measured on the four ResNet-50 3x3 layers at batch 256 (profiles/r04_wino4_pretransform_ab.txt,
// ms per launch fused -> PRE, forward | backward-data): 55x55 0.244 -> 0.410 | 0.228 -> 0.398 (V = 462 MB), 28x28 0.204 -> 0.297 |
// 0.186 -> 0.286 (231 MB), 14x14 0.221 -> 0.256 | 0.202 -> 0.226 (151 MB), 7x7 0.220 -> 0.227 | 0.206 -> 0.209 (75 MB). Results are
// bit-identical (the Winograd and whole-tensor suites passed with the limit at 200), but 2.25 x the input through HBM / L2 and
// a second launch cost more than the vector work they take out of the main loop — the fused kernel stays.
size_t wino4_input_bytes(const pz_conv_desc *d, int which, int P, int Q) {
	static const long limit_mb = [] { const char *e = getenv("PUZZLE_MI355_WINO_PRE_MB"); return e ? atol(e) : 0L; }();
	if (!wino4_pick(d, which, P, Q) || d->r != 3) return 0;
	int prod, red;
	w4_dims(d, which, &prod, &red);
	const int OP = which == PZ_CONV_FWD ? P : d->h, OQ = which == PZ_CONV_FWD ? Q : d->w;
	const long tiles = (long)d->n * ((OP + 3) / 4) * ((OQ + 3) / 4);
	const size_t bytes = (size_t)ceil_div(tiles, TB) * (red / BC) * kV * sizeof(float);
	return bytes <= (size_t)limit_mb << 20 ? bytes : 0;
}

size_t wino4_workspace_bytes(const pz_conv_desc *d, int which) {
	int prod, red;
	w4_dims(d, which, &prod, &red);
	return (size_t)ceil_div(prod, KB) * (red / BC) * kU * sizeof(float);
}

int wino4_stats_strips(const pz_conv_desc *d, int P, int Q) {
	const int m = w4_tile(d);
	return ceil_div((long)d->n * ((P + m - 1) / m) * ((Q + m - 1) / m), TB);
}

static W4FilterArgs w4_filter_args(const pz_conv_desc *d, int which, const float *w, float *u) {
	int prod, red;
	w4_dims(d, which, &prod, &red);
	W4FilterArgs fa{};
	fa.w = w, fa.u = u, fa.mode = which == PZ_CONV_FWD ? 0 : 1;
	fa.K = d->k, fa.C = d->c, fa.R = d->r, fa.prod = prod, fa.red = red;
	fa.kblocks = ceil_div(prod, KB), fa.chunks = red / BC;
	return fa;
}

int wino4_filter_batch(const pz_conv_desc *const *descs, const int *which, const float *const *w, float *const *u, int n, hipStream_t st) {
	if (n == 0) return PZ_OK;
	W4FilterBatch b{};
	for (int i = 0; i < n; ++i) {
		b.job[i] = w4_filter_args(descs[i], which[i], w[i], u[i]);
		const long ftotal = (long)b.job[i].kblocks * b.job[i].chunks * KB * BC;
		b.start[i + 1] = b.start[i] + stream_grid(ftotal, 256);
	}
	b.n = n;
	wino4_filter_batch_kernel<<<b.start[n], 256, 0, st>>>(b);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int wino4_conv(const pz_conv_desc *d, int which, int P, int Q, const float *in, const float *w, const float *bias, float *out,
               void *workspace, hipStream_t st, float *stats, bool filters_ready, void *vscratch, const BnStatsOut *bst) {
	PZ_REQUIRE(bst == nullptr || which == PZ_CONV_BWD_DATA, "wino4_conv: the gated statistics belong to a backward-data launch");
	W4FilterArgs fa = w4_filter_args(d, which, w, (float *)workspace);
	if (!filters_ready) {
		const long ftotal = (long)fa.kblocks * fa.chunks * KB * BC;
		wino4_filter_kernel<<<stream_grid(ftotal, 256), 256, 0, st>>>(fa);
		PZ_LAUNCH_CHECK();
	}

	W4Args a{};
	a.x = in, a.u = (const float *)workspace, a.bias = bias, a.y = out;
	a.N = d->n, a.C = fa.red, a.K = fa.prod;
	if (which == PZ_CONV_FWD) {
		a.H = d->h, a.W = d->w, a.P = P, a.Q = Q, a.pad_h = d->pad_h, a.pad_w = d->pad_w;
	} else {
		a.H = P, a.W = Q, a.P = d->h, a.Q = d->w, a.pad_h = (d->r - 1) - d->pad_h, a.pad_w = (d->s - 1) - d->pad_w;
	}
	a.ts = w4_tile(d);
	a.TY = (a.P + a.ts - 1) / a.ts, a.TX = (a.Q + a.ts - 1) / a.ts, a.tiles = a.N * a.TY * a.TX;
	a.chunks = fa.chunks, a.tblocks = ceil_div(a.tiles, TB), a.kblocks = fa.kblocks;
	a.x_bytes = (unsigned)((size_t)a.N * a.C * a.H * a.W * 4);
	a.y_bytes = (unsigned)((size_t)a.N * a.K * a.P * a.Q * 4);
	a.stats = reinterpret_cast<float4 *>(stats);
	if (bst) a.gx = bst->gx, a.gab = reinterpret_cast<const float2 *>(bst->gab), a.gmean = bst->gmean, a.gst = reinterpret_cast<float2 *>(bst->gst);
	if (vscratch != nullptr && wino4_input_bytes(d, which, P, Q) > 0) {
		a.v = (const float *)vscratch;
		wino4_input_kernel<<<a.tblocks * ((a.chunks + 1) / 2), 256, 0, st>>>(a, (float *)vscratch);
		PZ_LAUNCH_CHECK();
		wino4_conv_kernel<true><<<ceil_div(a.tblocks, 8) * 8 * fa.kblocks, 256, 0, st>>>(a);
	} else if (a.ts == 2) {
		wino4_conv_kernel<false, 2><<<ceil_div(a.tblocks, 8) * 8 * fa.kblocks, 256, 0, st>>>(a);
	} else {
		wino4_conv_kernel<false><<<ceil_div(a.tblocks, 8) * 8 * fa.kblocks, 256, 0, st>>>(a);
	}
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}


}  // namespace pz

extern "C" {

int pz_conv_winograd_tile_set(int tile) {
	PZ_REQUIRE(tile == 0 || tile == 2 || tile == 4, "pz_conv_winograd_tile_set: %d is not one of 0 (by cost), 2, 4", tile);
	pz::g_wino_tile = tile;
	return PZ_OK;
}

int pz_conv_winograd_tile_get(int *tile) {
	PZ_REQUIRE(tile != nullptr, "pz_conv_winograd_tile_get: null output");
	*tile = pz::g_wino_tile;
	return PZ_OK;
}

}  // extern "C"
