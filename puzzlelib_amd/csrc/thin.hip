// Backward-data of a stride-2 (and, below, unit-stride) convolution with very few input maps — the stem of an ImageNet network (3 maps, 7x7 / 2,
// Models/Nets/ResNet.py:88): dx has 3 channels, so as an implicit GEMM its M side fills 3 of 64 tile rows (the 128x128 /
// 64x256 MFMA tiles of conv.hip spend 9.5 ms on 60 GFLOP of useful work at batch 256). This is the direct form, shaped
// for the vector ALU instead:
//
//   dx[n, c, 2i+a, 2j+b] = sum_k sum_{dr, ds} dy[n, k, i + dmin + dr, j + dmin + ds] * w[k, c, r(a, dr), s(b, ds)]
//   r(a, dr) = a + pad - 2*(dr + dmin)         (the filter taps that reach output row parity a from input row i + dmin + dr)
//
// A coarse pixel (i, j) of one image = the 2 x 2 x C outputs that share one WR x WS window of dy (4 x 4 for 7x7 / pad 3).
// One thread owns kThinCells vertically adjacent coarse pixels; per reduction channel k it loads their joint window once
// (5 x 4 loads for 2 cells, lanes along j: coalesced) and issues one fma per (output, valid tap) — 147 per cell for the stem, the two
// column parities of an output row as one v_pk_fma_f32 — whose weight operand is WAVE-UNIFORM: the packed filter is indexed by k and compile-time constants only, so
// the compiler fetches it with scalar loads and the fma takes it from an SGPR. No LDS, no barrier, 2 x 2 x C accumulators.
// Taps that do not exist for a parity (a = 0 meets 3 filter rows, a = 1 meets 4) are skipped at compile time.
// Work: 2*N*P*Q*K*C*R*S FLOP on the VALU (78 TFLOP/s of plain fp32 fma on 256 CUs); dy is read once from HBM.
#include "common.h"

#ifndef PZ_THIN_UNROLL
#define PZ_THIN_UNROLL 1           // reduction channels unrolled per thread (2 with two-cell threads: registers, 0.83 -> 0.99 ms)
#endif
#define PZ_PRAGMA_(x) _Pragma(#x)
#define PZ_UNROLL(n) PZ_PRAGMA_(unroll n)      // (a macro inside a plain #pragma does not survive -save-temps)

namespace {

constexpr unsigned kOOB = 0xfffffff0u;

// d = (a + pad - r) / 2 over the valid (a, r) pairs: the window of input rows an output row pair touches
constexpr int win_min(int R, int pad) {
	int m = 1 << 20;
	for (int a = 0; a < 2; ++a)
		for (int r = 0; r < R; ++r)
			if (((a + pad - r) & 1) == 0) m = (a + pad - r) / 2 < m ? (a + pad - r) / 2 : m;
	return m;
}

constexpr int win_max(int R, int pad) {
	int m = -(1 << 20);
	for (int a = 0; a < 2; ++a)
		for (int r = 0; r < R; ++r)
			if (((a + pad - r) & 1) == 0) m = (a + pad - r) / 2 > m ? (a + pad - r) / 2 : m;
	return m;
}

template <int R, int PAD>
struct Win {
	static constexpr int lo = win_min(R, PAD), hi = win_max(R, PAD), size = hi - lo + 1;
	static constexpr int tap(int a, int d) { return a + PAD - 2 * (d + lo); }                 // filter index, may be out of range
	static constexpr bool valid(int a, int d) { return tap(a, d) >= 0 && tap(a, d) < R; }
};

// wpk[k][dr][ds][a][c][b] = w[k][c][r(a, dr)][s(b, ds)], zero where the tap does not exist. The column parity b is the
// fastest axis: the two outputs (2i+a, 2j) and (2i+a, 2j+1) of a channel take one v_pk_fma_f32 whose weight pair is an
// aligned SGPR pair
template <int C, int R, int S, int PH, int PW>
__global__ void __launch_bounds__(256) thin_pack_kernel(const float *__restrict__ w, float *__restrict__ wpk, int K) {
	using WH = Win<R, PH>;
	using WW = Win<S, PW>;
	const int per_k = WH::size * WW::size * 4 * C;
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= K * per_k) return;
	int t = idx;
	const int b = t & 1;
	t >>= 1;
	const int c = t % C;
	t /= C;
	const int a = t & 1;
	t >>= 1;
	const int ds = t % WW::size;
	t /= WW::size;
	const int dr = t % WH::size, k = t / WH::size;
	const int r = a + PH - 2 * (dr + WH::lo), s = b + PW - 2 * (ds + WW::lo);
	wpk[idx] = (r >= 0 && r < R && s >= 0 && s < S) ? w[((k * C + c) * R + r) * S + s] : 0.f;
}

// NC vertically adjacent coarse pixels per thread: their windows overlap in WR - 1 rows, so WR + NC - 1 rows of WS
// columns are loaded for NC cells (10 loads per cell and reduction channel with NC = 2 instead of 16), every load still
// 256 contiguous bytes per wave (lanes along j). The kernel is bound by those 4-byte gathers next to its fmas: without
// them it takes 0.68 instead of 1.01 ms on the stem. Measured on the stem (ms): NC = 1: 1.01, 2: 0.83, 3: 0.86, 4: 0.87,
// 6: 0.85, 8: 0.89; cells side by side in a row instead (lane stride 16 B: four times the cache lines per load): 1.6.
#ifndef PZ_THIN_CELLS
#define PZ_THIN_CELLS 2
#endif
constexpr int kThinCells = PZ_THIN_CELLS;

template <int C, int R, int S, int PH, int PW>
__global__ void __launch_bounds__(256) thin_dgrad_kernel(const float *__restrict__ dy, const float *__restrict__ wpk,
                                                          float *__restrict__ dx, int K, int P, int Q, int H, int W, int Hg,
                                                          int Wc, unsigned dy_bytes) {
	using WH = Win<R, PH>;
	using WW = Win<S, PW>;
	constexpr int WR = WH::size, WS = WW::size, NC = kThinCells, WROWS = WR + NC - 1;

	const int grp = blockIdx.x * 256 + threadIdx.x;            // (group of NC coarse rows, coarse column), row-major: lanes along j
	const int n = blockIdx.y;
	const bool live = grp < Hg * Wc;
	const int ig = grp / Wc, j = grp - ig * Wc, i = ig * NC;    // first coarse row of the group

	// byte offsets of the window inside one (n, k) plane of dy; outside the map (or a dead thread) -> the hardware returns 0
	unsigned woff[WROWS][WS];
#pragma unroll
	for (int dr = 0; dr < WROWS; ++dr)
#pragma unroll
		for (int ds = 0; ds < WS; ++ds) {
			const int p = i + WH::lo + dr, q = j + WW::lo + ds;
			woff[dr][ds] = (live && (unsigned)p < (unsigned)P && (unsigned)q < (unsigned)Q) ? (unsigned)(p * Q + q) * 4u : kOOB;
		}

	const __amdgpu_buffer_rsrc_t dyr = __builtin_amdgcn_make_buffer_rsrc((void *)dy, 0, dy_bytes, 0x00020000);
	const unsigned plane = (unsigned)(P * Q) * 4u;
	unsigned soff = (unsigned)n * (unsigned)K * plane;          // scalar: start of this image's first plane

	typedef float f32x2 __attribute__((ext_vector_type(2)));
	f32x2 acc[NC][2][C];                                       // [cell][row parity a][channel] = the column pair (b = 0, 1)
#pragma unroll
	for (int e = 0; e < NC; ++e)
#pragma unroll
		for (int a = 0; a < 2; ++a)
#pragma unroll
			for (int c = 0; c < C; ++c) acc[e][a][c] = f32x2{0.f, 0.f};

	constexpr int per_k = WR * WS * 4 * C;
PZ_UNROLL(PZ_THIN_UNROLL)
	for (int k = 0; k < K; ++k, soff += plane) {
		float v[WROWS][WS];
#pragma unroll
		for (int dr = 0; dr < WROWS; ++dr)
#pragma unroll
			for (int ds = 0; ds < WS; ++ds)
				v[dr][ds] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dyr, woff[dr][ds], soff, 0));

		const f32x2 *wk = reinterpret_cast<const f32x2 *>(wpk + (size_t)k * per_k);      // wave-uniform: scalar loads
#pragma unroll
		for (int dr = 0; dr < WR; ++dr)
#pragma unroll
			for (int ds = 0; ds < WS; ++ds)
#pragma unroll
				for (int a = 0; a < 2; ++a)
					if (WH::valid(a, dr)) {
#pragma unroll
						for (int c = 0; c < C; ++c) {
							const f32x2 w2 = wk[((dr * WS + ds) * 2 + a) * C + c];
#pragma unroll
							for (int e = 0; e < NC; ++e) {
								const float x = v[dr + e][ds];
								// a tap that exists for one column parity only is a scalar fma on that half: no 0 * dy term
								// enters the other sum (it would turn a non-finite dy into NaNs the convolution does not produce)
								if (WW::valid(0, ds) && WW::valid(1, ds))
									acc[e][a][c] = __builtin_elementwise_fma(f32x2{x, x}, w2, acc[e][a][c]);
								else if (WW::valid(0, ds))
									acc[e][a][c][0] = __builtin_fmaf(x, w2[0], acc[e][a][c][0]);
								else if (WW::valid(1, ds))
									acc[e][a][c][1] = __builtin_fmaf(x, w2[1], acc[e][a][c][1]);
							}
						}
					}
	}

	if (!live) return;
	typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
	const int x0 = 2 * j;
#pragma unroll
	for (int c = 0; c < C; ++c)
#pragma unroll
		for (int e = 0; e < NC; ++e)
#pragma unroll
			for (int a = 0; a < 2; ++a) {
				const int h = 2 * (i + e) + a;
				if (h >= H) continue;
				float *row = dx + (((size_t)n * C + c) * H + h) * W + x0;
				if (x0 + 1 < W)
					*reinterpret_cast<f2u *>(row) = acc[e][a][c];
				else if (x0 < W)
					row[0] = acc[e][a][c][0];
			}
}

// ---- unit stride ---------------------------------------------------------------------------------------------------------
// The first layer of a CIFAR-sized network (config 3, TestLib/CnnCifar10NIN.py: 3 -> 192 maps, 5x5 / 1, pad 2) has the same
// problem without the stride: as an implicit GEMM its input gradient fills 3 of 64 tile rows and, with 512 tiles of 300
// k-tiles each, holds every CU for 0.68 ms of a 3 ms step (5.8 TFLOP/s) while the filter-gradient stream starves behind
// it. Direct form:   dx[n, c, h, w] = sum_k sum_{r, s} dy[n, k, h + pad - r, w + pad - s] * w[k, c, r, s]
// A thread owns NC vertically adjacent pixels of one column (lanes along w: every load is 256 contiguous bytes per wave);
// per reduction channel it loads their joint (R + NC - 1) x S window of dy and issues one fma per (pixel, tap, channel
// pair): the channels of a tap are an aligned pair of wave-uniform weights (wpk[k][R-1-r][S-1-s][c], channels padded to
// even), so a tap costs ceil(C / 2) v_pk_fma_f32 per pixel.
#ifndef PZ_THIN1_CELLS
#define PZ_THIN1_CELLS 4
#endif
constexpr int kThin1Cells = PZ_THIN1_CELLS;

template <int C, int R, int S>
__global__ void __launch_bounds__(256) thin1_pack_kernel(const float *__restrict__ w, float *__restrict__ wpk, int K) {
	constexpr int CP = (C + 1) / 2 * 2, per_k = R * S * CP;
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= K * per_k) return;
	int t = idx;
	const int c = t % CP;
	t /= CP;
	const int ds = t % S;
	t /= S;
	const int dr = t % R, k = t / R;
	wpk[idx] = c < C ? w[((k * C + c) * R + (R - 1 - dr)) * S + (S - 1 - ds)] : 0.f;
}

// KS waves of a workgroup share the reduction channels (wave w takes k = w, w + KS, ...: still wave-uniform weights) and
// add their partial sums through LDS in wave order: at batch 128 a 32x32 map is 32 768 threads of 4 pixels — 2 waves per
// CU, every one of them waiting on its own loads (0.28 ms); with 8 k-slices the same loads are spread over 16 waves per CU.
template <int C, int R, int S, int PH, int PW, int KS>
__global__ void __launch_bounds__(64 * KS) thin1_dgrad_kernel(const float *__restrict__ dy, const float *__restrict__ wpk,
                                                              float *__restrict__ dx, int K, int P, int Q, int H, int W, int Hg,
                                                              unsigned dy_bytes) {
	constexpr int NC = kThin1Cells, WROWS = R + NC - 1, CP2 = (C + 1) / 2;
	const int lane = threadIdx.x & 63, ks = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int grp = blockIdx.x * 64 + lane;                    // (group of NC rows, column), row-major: lanes along w
	const int n = blockIdx.y;
	const bool live = grp < Hg * W;
	const int ig = grp / W, x = grp - ig * W, h0 = ig * NC;

	unsigned woff[WROWS][S];
#pragma unroll
	for (int dr = 0; dr < WROWS; ++dr)
#pragma unroll
		for (int ds = 0; ds < S; ++ds) {
			const int p = h0 + PH - (R - 1) + dr, q = x + PW - (S - 1) + ds;
			woff[dr][ds] = (live && (unsigned)p < (unsigned)P && (unsigned)q < (unsigned)Q) ? (unsigned)(p * Q + q) * 4u : kOOB;
		}

	const __amdgpu_buffer_rsrc_t dyr = __builtin_amdgcn_make_buffer_rsrc((void *)dy, 0, dy_bytes, 0x00020000);
	const unsigned plane = (unsigned)(P * Q) * 4u;
	unsigned soff = ((unsigned)n * (unsigned)K + (unsigned)ks) * plane;

	typedef float f32x2 __attribute__((ext_vector_type(2)));
	f32x2 acc[NC][CP2];
#pragma unroll
	for (int e = 0; e < NC; ++e)
#pragma unroll
		for (int c = 0; c < CP2; ++c) acc[e][c] = f32x2{0.f, 0.f};

	constexpr int per_k = R * S * CP2 * 2;
	for (int k = ks; k < K; k += KS, soff += KS * plane) {
		float v[WROWS][S];
#pragma unroll
		for (int dr = 0; dr < WROWS; ++dr)
#pragma unroll
			for (int ds = 0; ds < S; ++ds)
				v[dr][ds] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dyr, woff[dr][ds], soff, 0));

		const f32x2 *wk = reinterpret_cast<const f32x2 *>(wpk + (size_t)k * per_k);      // wave-uniform: scalar loads
#pragma unroll
		for (int dr = 0; dr < R; ++dr)
#pragma unroll
			for (int ds = 0; ds < S; ++ds)
#pragma unroll
				for (int c = 0; c < CP2; ++c) {
					const f32x2 w2 = wk[(dr * S + ds) * CP2 + c];
#pragma unroll
					for (int e = 0; e < NC; ++e) {
						const float xv = v[dr + e][ds];
						if (2 * c + 1 < C)
							acc[e][c] = __builtin_elementwise_fma(f32x2{xv, xv}, w2, acc[e][c]);
						else                      // the odd last channel: no 0 * dy term for the padding slot
							acc[e][c][0] = __builtin_fmaf(xv, w2[0], acc[e][c][0]);
					}
				}
	}

	if constexpr (KS > 1) {
		__shared__ f32x2 part[KS - 1][NC * CP2][64];
		if (ks > 0) {
#pragma unroll
			for (int e = 0; e < NC; ++e)
#pragma unroll
				for (int c = 0; c < CP2; ++c) part[ks - 1][e * CP2 + c][lane] = acc[e][c];
		}
		__syncthreads();
		if (ks > 0) return;
#pragma unroll
		for (int q = 0; q < KS - 1; ++q)
#pragma unroll
			for (int e = 0; e < NC; ++e)
#pragma unroll
				for (int c = 0; c < CP2; ++c) acc[e][c] += part[q][e * CP2 + c][lane];
	}

	if (!live) return;
#pragma unroll
	for (int c = 0; c < C; ++c)
#pragma unroll
		for (int e = 0; e < NC; ++e) {
			const int h = h0 + e;
			if (h < H) dx[(((size_t)n * C + c) * H + h) * W + x] = acc[e][c / 2][c & 1];
		}
}

template <int C, int R, int S, int PH, int PW>
int thin1_launch(const pz_conv_desc *d, int P, int Q, const float *dy, const float *w, float *dx, void *workspace, hipStream_t st) {
	float *wpk = (float *)workspace;
	constexpr int per_k = R * S * ((C + 1) / 2 * 2);
	thin1_pack_kernel<C, R, S><<<pz::ceil_div((long)d->k * per_k, 256), 256, 0, st>>>(w, wpk, d->k);
	PZ_LAUNCH_CHECK();
	const int Hg = pz::ceil_div(d->h, kThin1Cells);
	dim3 grid(pz::ceil_div((long)Hg * d->w, 64), d->n);
	const unsigned dy_bytes = (unsigned)((size_t)d->n * d->k * P * Q * 4);
	// k-slices: enough waves for 16 per CU, at least 8 reduction channels per slice
	const long waves = (long)grid.x * grid.y;
	int ks = 1;
	while (ks < 8 && waves * ks < 16L * pz::kNumCU && d->k / (2 * ks) >= 8) ks *= 2;
#define PZ_THIN1_GO(KS) thin1_dgrad_kernel<C, R, S, PH, PW, KS><<<grid, 64 * KS, 0, st>>>(dy, wpk, dx, d->k, P, Q, d->h, d->w, Hg, dy_bytes)
	if (ks == 8) PZ_THIN1_GO(8);
	else if (ks == 4) PZ_THIN1_GO(4);
	else if (ks == 2) PZ_THIN1_GO(2);
	else PZ_THIN1_GO(1);
#undef PZ_THIN1_GO
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

// the shapes instantiated: (C, R, S, pad_h, pad_w)
template <int C, int R, int S, int PH, int PW>
bool thin_match(const pz_conv_desc *d) {
	return d->c == C && d->r == R && d->s == S && d->pad_h == PH && d->pad_w == PW;
}

template <int C, int R, int S, int PH, int PW>
int thin_launch(const pz_conv_desc *d, int P, int Q, const float *dy, const float *w, float *dx, void *workspace, hipStream_t st) {
	float *wpk = (float *)workspace;
	const int per_k = Win<R, PH>::size * Win<S, PW>::size * 4 * C;
	thin_pack_kernel<C, R, S, PH, PW><<<pz::ceil_div((long)d->k * per_k, 256), 256, 0, st>>>(w, wpk, d->k);
	PZ_LAUNCH_CHECK();
	const int Hg = pz::ceil_div((d->h + 1) / 2, kThinCells), Wc = (d->w + 1) / 2;      // groups of coarse rows, coarse columns
	dim3 grid(pz::ceil_div((long)Hg * Wc, 256), d->n);
	thin_dgrad_kernel<C, R, S, PH, PW><<<grid, 256, 0, st>>>(dy, wpk, dx, d->k, P, Q, d->h, d->w, Hg, Wc,
	                                                           (unsigned)((size_t)d->n * d->k * P * Q * 4));
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

}  // namespace

namespace pz {

#define PZ_THIN_SHAPES(X) X(3, 7, 7, 3, 3) X(3, 3, 3, 1, 1) X(1, 7, 7, 3, 3) X(3, 5, 5, 2, 2) X(4, 7, 7, 3, 3) X(1, 3, 3, 1, 1)
// ... and with unit stride
#define PZ_THIN1_SHAPES(X) X(3, 5, 5, 2, 2) X(3, 3, 3, 1, 1) X(1, 5, 5, 2, 2) X(1, 3, 3, 1, 1) X(3, 7, 7, 3, 3) X(1, 5, 5, 0, 0) X(3, 5, 5, 0, 0)

bool thin_dgrad_eligible(const pz_conv_desc *d, int P, int Q) {
	if (d->dil_h != 1 || d->dil_w != 1 || d->groups != 1 || d->stride_h != d->stride_w) return false;
	if ((size_t)d->n * d->k * P * Q * 4 >= 0xfffffff0ull || d->n > 65535) return false;
	if (d->stride_h == 2) {
#define X(C, R, S, PH, PW) if (thin_match<C, R, S, PH, PW>(d)) return true;
		PZ_THIN_SHAPES(X)
#undef X
	} else if (d->stride_h == 1) {
#define X(C, R, S, PH, PW) if (thin_match<C, R, S, PH, PW>(d)) return true;
		PZ_THIN1_SHAPES(X)
#undef X
	}
	return false;
}

size_t thin_dgrad_workspace_bytes(const pz_conv_desc *d) {
	if (d->stride_h == 1) return (size_t)d->k * d->r * d->s * ((d->c + 1) / 2 * 2) * sizeof(float);
	// window <= ceil(R/2)+1 per axis, 2x2 parities, C channels
	return (size_t)d->k * ((d->r + 1) / 2 + 1) * ((d->s + 1) / 2 + 1) * 4 * d->c * sizeof(float);
}

int thin_dgrad(const pz_conv_desc *d, int P, int Q, const float *dy, const float *w, float *dx, void *workspace, hipStream_t st) {
	if (d->stride_h == 1) {
#define X(C, R, S, PH, PW) if (thin_match<C, R, S, PH, PW>(d)) return thin1_launch<C, R, S, PH, PW>(d, P, Q, dy, w, dx, workspace, st);
		PZ_THIN1_SHAPES(X)
#undef X
	}
#define X(C, R, S, PH, PW) if (thin_match<C, R, S, PH, PW>(d)) return thin_launch<C, R, S, PH, PW>(d, P, Q, dy, w, dx, workspace, st);
	PZ_THIN_SHAPES(X)
#undef X
	pz::set_error("thin_dgrad: shape not instantiated");
	return PZ_ERR_INVALID;
}

}  // namespace pz
