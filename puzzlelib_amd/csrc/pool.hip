// 2-D pooling (max / average incl. padding / average excl. padding), NCHW fp32, HBM-bound gather kernels.
// Replaces DnnContext.poolNd / poolNdBackward — Hip/Wrappers/MIOpen.py:549-598. The training-mode "workspace" is
// this library's own format: one byte per output element holding the window-local arg-max (r*size_w + s, first
// maximum in row-major window order). Backward is a gather over the (few) windows covering each input pixel:
// no atomics, deterministic.
#include "common.h"
#include <cfloat>

namespace {

// One workgroup owns `group` consecutive (image, channel) planes (several when a plane has fewer outputs than threads),
// so all index arithmetic is 32-bit and a wave's accesses stay inside neighbouring rows of one plane.
struct PoolGeom {
	int P, Q, group;
	unsigned planes;
};

// SZ / ST: square window size / stride known at compile time (0 = read from the descriptor): the common 2x2/2, 3x3/2
// and 3x3/1 windows get unrolled taps and shift/multiply index arithmetic — the generic loops are ALU-bound
template <int MODE, int SZ, int ST>
__global__ void __launch_bounds__(256) pool_fwd_kernel(pz_pool_desc d, PoolGeom g, const float *__restrict__ x,
                                                        float *__restrict__ y, uint8_t *__restrict__ idx) {
	if (SZ) d.size_h = d.size_w = SZ;
	if (ST) d.stride_h = d.stride_w = ST;
	const unsigned PQ = (unsigned)(g.P * g.Q), HW = (unsigned)(d.h * d.w);
	const unsigned plane0 = blockIdx.x * (unsigned)g.group;
	const unsigned nplanes = min((unsigned)g.group, g.planes - plane0);

	for (unsigned j = threadIdx.x; j < nplanes * PQ; j += 256) {
		const unsigned pl = j / PQ, i = j - pl * PQ;
		const int p = (int)(i / (unsigned)g.Q), q = (int)(i - (unsigned)p * g.Q);
		const float *img = x + (size_t)(plane0 + pl) * HW;
		const size_t o = (size_t)(plane0 + pl) * PQ + i;
		const int h0 = p * d.stride_h - d.pad_h, w0 = q * d.stride_w - d.pad_w;
		const int r_lo = h0 < 0 ? -h0 : 0, r_hi = min(d.size_h, d.h - h0);
		const int s_lo = w0 < 0 ? -w0 : 0, s_hi = min(d.size_w, d.w - w0);

		if (MODE == 0) {
			float best = -INFINITY;                    // NumpyDnn.pool2d pads with -inf
			int bi = 0;
			bool found = false;
			if constexpr (SZ != 0) {
#pragma unroll
				for (int r = 0; r < SZ; ++r)
#pragma unroll
					for (int t = 0; t < SZ; ++t) {
						if (r < r_lo || r >= r_hi || t < s_lo || t >= s_hi) continue;
						const float v = img[(h0 + r) * d.w + w0 + t];
						if (!found || v > best) best = v, bi = r * SZ + t, found = true;       // first maximum in window order
					}
			} else {
				for (int r = r_lo; r < r_hi; ++r)
					for (int t = s_lo; t < s_hi; ++t) {
						const float v = img[(h0 + r) * d.w + w0 + t];
						if (!found || v > best) best = v, bi = r * d.size_w + t, found = true;
					}
			}
			y[o] = best;
			if (idx) idx[o] = (uint8_t)bi;
		} else {
			float sum = 0.f;
			for (int r = r_lo; r < r_hi; ++r)
				for (int t = s_lo; t < s_hi; ++t) sum += img[(h0 + r) * d.w + w0 + t];
			const int cnt = max(r_hi - r_lo, 0) * max(s_hi - s_lo, 0);
			const float div = MODE == 1 ? (float)(d.size_h * d.size_w) : (float)(cnt > 0 ? cnt : 1);
			y[o] = sum / div;
		}
	}
}

// number of in-bounds taps of window (p, q): divisor of avgNoPad
__device__ __forceinline__ int valid_taps(const pz_pool_desc &d, int p, int q) {
	const int h0 = p * d.stride_h - d.pad_h, w0 = q * d.stride_w - d.pad_w;
	const int hlo = h0 < 0 ? 0 : h0, hhi = h0 + d.size_h > d.h ? d.h : h0 + d.size_h;
	const int wlo = w0 < 0 ? 0 : w0, whi = w0 + d.size_w > d.w ? d.w : w0 + d.size_w;
	const int a = hhi - hlo, b = whi - wlo;
	return a > 0 && b > 0 ? a * b : 1;
}

// backward: a gather over the (few) windows covering each input pixel
template <int MODE, bool HAS_IDX, int SZ, int ST>
__global__ void __launch_bounds__(256) pool_bwd_kernel(pz_pool_desc d, PoolGeom g, const float *__restrict__ dy,
                                                        const float *__restrict__ x, const uint8_t *__restrict__ idx,
                                                        float *__restrict__ dx) {
	if (SZ) d.size_h = d.size_w = SZ;
	if (ST) d.stride_h = d.stride_w = ST;
	const unsigned PQ = (unsigned)(g.P * g.Q), HW = (unsigned)(d.h * d.w);
	const unsigned plane0 = blockIdx.x * (unsigned)g.group;
	const unsigned nplanes = min((unsigned)g.group, g.planes - plane0);

	for (unsigned j = threadIdx.x; j < nplanes * HW; j += 256) {
		const unsigned pl = j / HW, i = j - pl * HW;
		const int hh = (int)(i / (unsigned)d.w), ww = (int)(i - (unsigned)hh * d.w);
		const size_t plane = plane0 + pl;
		const float *gimg = dy + plane * PQ;

		// windows p with p*stride - pad <= hh < p*stride - pad + size
		const int hp = hh + d.pad_h, wp = ww + d.pad_w;
		int p_lo = hp - d.size_h + 1;
		p_lo = p_lo <= 0 ? 0 : (p_lo + d.stride_h - 1) / d.stride_h;
		const int p_hi = min(hp / d.stride_h, g.P - 1);
		int q_lo = wp - d.size_w + 1;
		q_lo = q_lo <= 0 ? 0 : (q_lo + d.stride_w - 1) / d.stride_w;
		const int q_hi = min(wp / d.stride_w, g.Q - 1);

		float sum = 0.f;
		for (int p = p_lo; p <= p_hi; ++p)
			for (int q = q_lo; q <= q_hi; ++q) {
				const float gv = gimg[p * g.Q + q];
				if (MODE == 0) {
					const int r = hp - p * d.stride_h, t = wp - q * d.stride_w;
					int win;
					if (HAS_IDX) {
						win = idx[plane * PQ + p * g.Q + q];
					} else {
						// recompute the first maximum of the window
						const float *img = x + plane * HW;
						const int h0 = p * d.stride_h - d.pad_h, w0 = q * d.stride_w - d.pad_w;
						float best = -INFINITY;
						bool found = false;
						win = 0;
						for (int rr = 0; rr < d.size_h; ++rr)
							for (int tt = 0; tt < d.size_w; ++tt) {
								const int a = h0 + rr, b = w0 + tt;
								if ((unsigned)a >= (unsigned)d.h || (unsigned)b >= (unsigned)d.w) continue;
								const float v = img[a * d.w + b];
								if (!found || v > best) best = v, win = rr * d.size_w + tt, found = true;
							}
					}
					if (win == r * d.size_w + t) sum += gv;
				} else if (MODE == 1) {
					sum += gv / (float)(d.size_h * d.size_w);
				} else {
					sum += gv / (float)valid_taps(d, p, q);
				}
			}
		dx[plane * HW + i] = sum;
	}
}

// Max pooling through LDS (3x3/2 on 112x112 -> 55x55 is the ResNet stem): the generic kernels issue one 4-byte load per
// tap and are bound by the texture-address unit, not by HBM. A workgroup owns a band of rows of one (image, channel)
// plane — small enough (<= 16 KB) that 8 workgroups share a CU and loads of one overlap the arithmetic of another:
//   forward : the band's input rows are read once with 16-byte loads into LDS, windows are evaluated out of LDS
//   backward: the dy rows and arg-max bytes covering the band are staged in LDS, dx is written 16 bytes per lane
constexpr int kPoolBandFloats = 3584;           // 14 KB
constexpr int kPoolFwdBand = 8;                 // output rows per workgroup (forward)
constexpr int kPoolBwdBand = 16;                // input rows per workgroup (backward)

// `coef` != nullptr: the input is a batch normalisation that was only described — the band is normalised while it is
// staged, v = fma(x, a[ch], b[ch]) and the fused ReLU's (v > 0 ? v : 0), exactly what pz_bn_apply_add would have written
// (same fma, same select), and the normalised tensor never exists in memory
template <int SZ, int ST>
__global__ void __launch_bounds__(256) maxpool_fwd_lds_kernel(pz_pool_desc d, PoolGeom g, const float *__restrict__ x,
                                                               float *__restrict__ y, uint8_t *__restrict__ idx,
                                                               const float *__restrict__ coef = nullptr, int relu = 0) {
	__shared__ __attribute__((aligned(16))) float rows[kPoolBandFloats];
	const int HW = d.h * d.w, PQ = g.P * g.Q;
	const int p0 = blockIdx.y * kPoolFwdBand, p1 = min(p0 + kPoolFwdBand, g.P);
	const int h_lo = max(p0 * ST - d.pad_h, 0), h_hi = min((p1 - 1) * ST - d.pad_h + SZ, d.h);     // input rows [h_lo, h_hi)
	const float *img = x + (size_t)blockIdx.x * HW + (size_t)h_lo * d.w;
	const int count = (h_hi - h_lo) * d.w;

	typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
	typedef float f4a __attribute__((ext_vector_type(4), may_alias));
	if (coef) {
		const int ch = (int)(blockIdx.x % (unsigned)d.c);
		const float a = coef[2 * ch], b = coef[2 * ch + 1];
		auto act = [&](float u) {
			const float v = __builtin_fmaf(u, a, b);
			return relu ? (v > 0.f ? v : 0.f) : v;
		};
		for (int i = threadIdx.x; i < (count >> 2); i += 256) {
			const f4u u = *reinterpret_cast<const f4u *>(img + 4 * i);
			*reinterpret_cast<f4a *>(&rows[4 * i]) = f4a{act(u[0]), act(u[1]), act(u[2]), act(u[3])};
		}
		for (int i = (count & ~3) + threadIdx.x; i < count; i += 256) rows[i] = act(img[i]);
	} else {
		for (int i = threadIdx.x; i < (count >> 2); i += 256) *reinterpret_cast<f4a *>(&rows[4 * i]) = *reinterpret_cast<const f4u *>(img + 4 * i);
		for (int i = (count & ~3) + threadIdx.x; i < count; i += 256) rows[i] = img[i];
	}
	__syncthreads();

	for (int i = threadIdx.x; i < (p1 - p0) * g.Q; i += 256) {
		const int p = p0 + i / g.Q, q = i % g.Q;
		const int h0 = p * ST - d.pad_h, w0 = q * ST - d.pad_w;
		float best = -INFINITY;
		int bi = 0;
		bool found = false;
#pragma unroll
		for (int r = 0; r < SZ; ++r)
#pragma unroll
			for (int t = 0; t < SZ; ++t) {
				const int hh = h0 + r, ww = w0 + t;
				if ((unsigned)hh >= (unsigned)d.h || (unsigned)ww >= (unsigned)d.w) continue;
				const float v = rows[(hh - h_lo) * d.w + ww];
				if (!found || v > best) best = v, bi = r * SZ + t, found = true;
			}
		const size_t o = (size_t)blockIdx.x * PQ + (size_t)p * g.Q + q;
		y[o] = best;
		if (idx) idx[o] = (uint8_t)bi;
	}
}

template <int SZ, int ST>
__global__ void __launch_bounds__(256) maxpool_bwd_lds_kernel(pz_pool_desc d, PoolGeom g, const float *__restrict__ dy,
                                                               const uint8_t *__restrict__ idx, float *__restrict__ dx) {
	__shared__ __attribute__((aligned(16))) float grad[kPoolBandFloats * 4 / 5];
	__shared__ uint8_t win[kPoolBandFloats * 4 / 5];
	const int HW = d.h * d.w, PQ = g.P * g.Q;
	const size_t plane = blockIdx.x;
	const int h0 = blockIdx.y * kPoolBwdBand, h1 = min(h0 + kPoolBwdBand, d.h);

	// output rows whose windows touch input rows [h0, h1)
	int pa = h0 + d.pad_h - SZ + 1;
	pa = pa <= 0 ? 0 : (pa + ST - 1) / ST;
	const int pb = min((h1 - 1 + d.pad_h) / ST, g.P - 1);
	const int count = max(pb - pa + 1, 0) * g.Q;
	for (int i = threadIdx.x; i < count; i += 256) {
		grad[i] = dy[plane * PQ + (size_t)pa * g.Q + i];
		win[i] = idx[plane * PQ + (size_t)pa * g.Q + i];
	}
	__syncthreads();

	auto pixel = [&](int hh, int ww) {
		const int hp = hh + d.pad_h, wp = ww + d.pad_w;
		int p_lo = hp - SZ + 1;
		p_lo = p_lo <= 0 ? 0 : (p_lo + ST - 1) / ST;
		const int p_hi = min(hp / ST, g.P - 1);
		int q_lo = wp - SZ + 1;
		q_lo = q_lo <= 0 ? 0 : (q_lo + ST - 1) / ST;
		const int q_hi = min(wp / ST, g.Q - 1);
		float sum = 0.f;
		for (int p = p_lo; p <= p_hi; ++p)
			for (int q = q_lo; q <= q_hi; ++q)
				if (win[(p - pa) * g.Q + q] == (hp - p * ST) * SZ + (wp - q * ST)) sum += grad[(p - pa) * g.Q + q];
		return sum;
	};

	typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
	float *out = dx + plane * HW;

	// 3x3 / stride 2 / unpadded (the ImageNet stem's pooling): which windows can have chosen an input pixel, and with which
	// tap, follows from the pixel's parities alone. A thread takes a 2 x 4 block of input pixels (rows 2i, 2i+1, columns
	// 4j..4j+3): the 2 x 3 windows (i-1, i) x (2j-1, 2j, 2j+1) around it are read ONCE (12 LDS reads for 8 outputs where the
	// per-pixel form below does 8 per output: the kernel was bound by its byte-wide LDS reads, 2.9 TB/s of HBM traffic).
	if (SZ == 3 && ST == 2 && d.pad_h == 0 && d.pad_w == 0 && (d.w & 3) == 0 && (h0 & 1) == 0) {
		const int w4 = d.w >> 2, rows2 = (h1 - h0 + 1) >> 1;
		for (int t = threadIdx.x; t < rows2 * w4; t += 256) {
			const int i = (h0 >> 1) + t / w4, j = t % w4;
			float gv[2][3];
			int wv[2][3];
#pragma unroll
			for (int a = 0; a < 2; ++a)
#pragma unroll
				for (int b = 0; b < 3; ++b) {
					const int p = i - 1 + a, q = 2 * j - 1 + b;
					const bool ok = p >= pa && p <= pb && q >= 0 && q < g.Q;
					const int at = ok ? (p - pa) * g.Q + q : 0;
					gv[a][b] = ok ? grad[at] : 0.f;
					wv[a][b] = ok ? (int)win[at] : -1;
				}
			auto take = [&](int a, int b, int r, int c) { return wv[a][b] == r * 3 + c ? gv[a][b] : 0.f; };
			// row 2i: window rows i-1 (tap row 2) and i (tap row 0); row 2i+1: window row i (tap row 1)
			// column 4j: windows 2j-1 (tap 2), 2j (tap 0); 4j+1: 2j (1); 4j+2: 2j (2), 2j+1 (0); 4j+3: 2j+1 (1)
			const f4u top = {(take(0, 0, 2, 2) + take(0, 1, 2, 0)) + (take(1, 0, 0, 2) + take(1, 1, 0, 0)),
			                 take(0, 1, 2, 1) + take(1, 1, 0, 1),
			                 (take(0, 1, 2, 2) + take(0, 2, 2, 0)) + (take(1, 1, 0, 2) + take(1, 2, 0, 0)),
			                 take(0, 2, 2, 1) + take(1, 2, 0, 1)};
			const f4u bot = {take(1, 0, 1, 2) + take(1, 1, 1, 0), take(1, 1, 1, 1), take(1, 1, 1, 2) + take(1, 2, 1, 0), take(1, 2, 1, 1)};
			const int hh = 2 * i;
			*reinterpret_cast<f4u *>(out + hh * d.w + 4 * j) = top;
			if (hh + 1 < h1) *reinterpret_cast<f4u *>(out + (hh + 1) * d.w + 4 * j) = bot;
		}
		return;
	}

	const int w4 = d.w >> 2;                                   // 4-pixel groups per row, then the row's remainder
	for (int i = threadIdx.x; i < (h1 - h0) * w4; i += 256) {
		const int hh = h0 + i / w4, ww = (i % w4) * 4;
		*reinterpret_cast<f4u *>(out + hh * d.w + ww) = f4u{pixel(hh, ww), pixel(hh, ww + 1), pixel(hh, ww + 2), pixel(hh, ww + 3)};
	}
	const int rem = d.w & 3;
	for (int i = threadIdx.x; i < (h1 - h0) * rem; i += 256) {
		const int hh = h0 + i / rem, ww = 4 * w4 + i % rem;
		out[hh * d.w + ww] = pixel(hh, ww);
	}
}

// whether the band of a plane of this geometry fits the LDS arrays above
inline bool pool_lds_fits(const pz_pool_desc *d, int Q, bool fwd) {
	const int st = d->stride_h, sz = d->size_h;
	if (fwd) return (size_t)((kPoolFwdBand - 1) * st + sz) * d->w <= kPoolBandFloats;
	return (size_t)(kPoolBwdBand / st + sz + 1) * Q <= kPoolBandFloats * 4 / 5;
}

// average over the whole plane (window = plane, no padding, one output). Planes are short (7 x 7 = 196 bytes at the end of
// ResNet-50, half a million of them at batch 256): one wave per plane left 15 of 64 lanes without an element and made
// every plane its own latency-bound trip (74 us for 26 MB). A group of LANES lanes takes a plane (LANES = the power of two
// covering hw / 4, at most 64), so a wave reads 64 / LANES consecutive planes = one contiguous stretch, and sums with a
// fixed shuffle tree inside the group.
template <int LANES>
__global__ void __launch_bounds__(256) pool_global_avg_fwd_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                                   unsigned planes, int hw) {
	constexpr int PER_BLOCK = 256 / LANES;
	const unsigned plane = blockIdx.x * (unsigned)PER_BLOCK + threadIdx.x / LANES;
	const int sub = threadIdx.x % LANES;
	float s = 0.f;
	if (plane < planes) {
		const float *img = x + (size_t)plane * hw;
		for (int i = sub; i < hw; i += LANES) s += img[i];
	}
#pragma unroll
	for (int m = LANES / 2; m > 0; m >>= 1) s += __shfl_xor(s, m, 64);
	if (sub == 0 && plane < planes) y[plane] = s / (float)hw;
}

// the backward pass writes dy[plane] / hw over every plane: a thread owns 4 consecutive elements (one 16-byte store when
// they lie in one plane and the address allows it; planes of 49 floats rarely do, so the store type is 4-byte aligned)
__global__ void __launch_bounds__(256) pool_global_avg_bwd_kernel(const float *__restrict__ dy, float *__restrict__ dx,
                                                                   size_t total, int hw, unsigned magic_hw) {
	typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
	const float inv = 1.f / (float)hw;
	// the workgroup's first element: plane and offset inside it by one (wave-uniform) 64-bit division; the 1 024 elements
	// behind it by multiply-high (exact: r < 2^16 + 1 024, magic_hw = ceil(2^32 / hw))
	const size_t first = (size_t)blockIdx.x * 1024;
	const size_t p_first = first / (size_t)hw;
	const unsigned r_first = (unsigned)(first - p_first * (size_t)hw);
	const size_t i = first + (size_t)threadIdx.x * 4;
	if (i >= total) return;
	float v[4];
#pragma unroll
	for (int e = 0; e < 4; ++e) {
		const unsigned r = r_first + threadIdx.x * 4u + e;
		v[e] = i + e < total ? dy[p_first + __umulhi(r, magic_hw)] * inv : 0.f;
	}
	if (i + 4 <= total)
		*reinterpret_cast<f4u *>(dx + i) = f4u{v[0], v[1], v[2], v[3]};
	else
		for (int e = 0; i + e < total; ++e) dx[i + e] = v[e];
}

// (a 1 x 1 plane is excluded: ceil(2^32 / 1) does not fit the 32-bit magic of the backward kernel, and the generic kernels
// copy such a tensor just as well)
inline bool pool_is_global_avg(const pz_pool_desc *d, int P, int Q) {
	return d->mode != 0 && P == 1 && Q == 1 && d->pad_h == 0 && d->pad_w == 0 && d->size_h == d->h && d->size_w == d->w &&
	       (size_t)d->n * d->c * d->h * d->w < ((size_t)1 << 32) && d->h * d->w < (1 << 16) && d->h * d->w >= 2;
}

// windows the LDS kernels are instantiated for: square 2x2/2, 3x3/2, 3x3/1
inline bool pool_lds_window(const pz_pool_desc *d) {
	const bool sq = d->size_h == d->size_w && d->stride_h == d->stride_w;
	return sq && ((d->size_h == 3 && (d->stride_h == 2 || d->stride_h == 1)) || (d->size_h == 2 && d->stride_h == 2));
}

// compile-time window variants
#define PZ_POOL_SPECIALISE(LAUNCH)                                                             \
	do {                                                                                       \
		const bool sq = d->size_h == d->size_w && d->stride_h == d->stride_w;                  \
		if (sq && d->size_h == 3 && d->stride_h == 2) { LAUNCH(3, 2); }                        \
		else if (sq && d->size_h == 2 && d->stride_h == 2) { LAUNCH(2, 2); }                   \
		else if (sq && d->size_h == 3 && d->stride_h == 1) { LAUNCH(3, 1); }                   \
		else { LAUNCH(0, 0); }                                                                 \
	} while (0)

PoolGeom pool_geom(const pz_pool_desc *d, int P, int Q, size_t per_plane) {
	PoolGeom g;
	g.P = P, g.Q = Q;
	g.planes = (unsigned)((size_t)d->n * d->c);
	g.group = per_plane >= 256 ? 1 : (int)(256 / per_plane);
	return g;
}

int pool_check(const pz_pool_desc *d, int *P, int *Q) {
	PZ_REQUIRE(d != nullptr, "pool: null descriptor");
	PZ_REQUIRE(d->n > 0 && d->c > 0 && d->h > 0 && d->w > 0, "pool: non-positive tensor dimension");
	PZ_REQUIRE(d->size_h > 0 && d->size_w > 0 && d->stride_h > 0 && d->stride_w > 0 && d->pad_h >= 0 && d->pad_w >= 0,
	           "pool: invalid window/stride/pad");
	PZ_REQUIRE(d->size_h * d->size_w <= 256, "pool: window larger than 256 taps");
	PZ_REQUIRE(d->mode >= 0 && d->mode <= 2, "pool: unknown mode %d", d->mode);
	PZ_REQUIRE((size_t)d->h * d->w < (1u << 30) && (size_t)d->n * d->c < (1u << 31), "pool: plane or plane count beyond 32-bit indexing");
	PZ_REQUIRE(d->h + 2 * d->pad_h >= d->size_h && d->w + 2 * d->pad_w >= d->size_w, "pool: window larger than padded input");
	*P = (d->h + 2 * d->pad_h - d->size_h) / d->stride_h + 1;
	*Q = (d->w + 2 * d->pad_w - d->size_w) / d->stride_w + 1;
	return PZ_OK;
}

}  // namespace

extern "C" {

int pz_pool2d_out_shape(const pz_pool_desc *d, int *p, int *q) { return pool_check(d, p, q); }

int pz_pool2d_fwd(const pz_pool_desc *d, const float *x, float *y, uint8_t *index_ws, pz_stream_t stream) {
	int P, Q;
	if (int rc = pool_check(d, &P, &Q)) return rc;
	PZ_REQUIRE(x && y, "pz_pool2d_fwd: null tensor");
	const PoolGeom g = pool_geom(d, P, Q, (size_t)P * Q);
	const unsigned blocks = (g.planes + g.group - 1) / g.group;
	hipStream_t st = pz::as_stream(stream);
	if (pool_is_global_avg(d, P, Q)) {
		const int hw = d->h * d->w;
		if (hw <= 16) pool_global_avg_fwd_kernel<4><<<(g.planes + 63) / 64, 256, 0, st>>>(x, y, g.planes, hw);
		else if (hw <= 64) pool_global_avg_fwd_kernel<16><<<(g.planes + 15) / 16, 256, 0, st>>>(x, y, g.planes, hw);
		else pool_global_avg_fwd_kernel<64><<<(g.planes + 3) / 4, 256, 0, st>>>(x, y, g.planes, hw);
	} else if (d->mode == 0 && pool_lds_window(d) && pool_lds_fits(d, Q, true) && g.planes < 65536u * 32768u) {
		const dim3 grid(g.planes, (P + kPoolFwdBand - 1) / kPoolFwdBand);
		if (d->size_h == 3 && d->stride_h == 2) maxpool_fwd_lds_kernel<3, 2><<<grid, 256, 0, st>>>(*d, g, x, y, index_ws);
		else if (d->size_h == 2) maxpool_fwd_lds_kernel<2, 2><<<grid, 256, 0, st>>>(*d, g, x, y, index_ws);
		else maxpool_fwd_lds_kernel<3, 1><<<grid, 256, 0, st>>>(*d, g, x, y, index_ws);
	} else if (d->mode == 0) {
#define PZ_L(SZ, ST) pool_fwd_kernel<0, SZ, ST><<<blocks, 256, 0, st>>>(*d, g, x, y, index_ws)
		PZ_POOL_SPECIALISE(PZ_L);
#undef PZ_L
	} else if (d->mode == 1) {
		pool_fwd_kernel<1, 0, 0><<<blocks, 256, 0, st>>>(*d, g, x, y, index_ws);
	} else {
		pool_fwd_kernel<2, 0, 0><<<blocks, 256, 0, st>>>(*d, g, x, y, index_ws);
	}
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

static bool pool_fwd_bn_ok(const pz_pool_desc *d, int Q, const PoolGeom &g) {
	return d->mode == 0 && pool_lds_window(d) && pool_lds_fits(d, Q, true) && g.planes < 65536u * 32768u;
}

int pz_pool2d_fwd_bn_supported(const pz_pool_desc *d, int *supported) {
	int P, Q;
	if (int rc = pool_check(d, &P, &Q)) return rc;
	PZ_REQUIRE(supported != nullptr, "pz_pool2d_fwd_bn_supported: null output");
	*supported = pool_fwd_bn_ok(d, Q, pool_geom(d, P, Q, (size_t)P * Q)) ? 1 : 0;
	return PZ_OK;
}

int pz_pool2d_fwd_bn(const pz_pool_desc *d, const float *x, const float *coef, int relu, float *y, uint8_t *index_ws,
                     pz_stream_t stream) {
	int P, Q;
	if (int rc = pool_check(d, &P, &Q)) return rc;
	PZ_REQUIRE(x && coef && y, "pz_pool2d_fwd_bn: null tensor");
	const PoolGeom g = pool_geom(d, P, Q, (size_t)P * Q);
	PZ_REQUIRE(pool_fwd_bn_ok(d, Q, g), "pz_pool2d_fwd_bn: this pooling does not run on the band kernel (pz_pool2d_fwd_bn_supported)");
	hipStream_t st = pz::as_stream(stream);
	const dim3 grid(g.planes, (P + kPoolFwdBand - 1) / kPoolFwdBand);
	if (d->size_h == 3 && d->stride_h == 2) maxpool_fwd_lds_kernel<3, 2><<<grid, 256, 0, st>>>(*d, g, x, y, index_ws, coef, relu);
	else if (d->size_h == 2) maxpool_fwd_lds_kernel<2, 2><<<grid, 256, 0, st>>>(*d, g, x, y, index_ws, coef, relu);
	else maxpool_fwd_lds_kernel<3, 1><<<grid, 256, 0, st>>>(*d, g, x, y, index_ws, coef, relu);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_pool2d_bwd(const pz_pool_desc *d, const float *dy, const float *x, const float *y, const uint8_t *index_ws, float *dx,
                  pz_stream_t stream) {
	int P, Q;
	if (int rc = pool_check(d, &P, &Q)) return rc;
	(void)y;
	PZ_REQUIRE(dy && dx, "pz_pool2d_bwd: null tensor");
	PZ_REQUIRE(d->mode != 0 || index_ws || x, "pz_pool2d_bwd: max pooling needs the index workspace or the input tensor");
	const PoolGeom g = pool_geom(d, P, Q, (size_t)d->h * d->w);
	const unsigned blocks = (g.planes + g.group - 1) / g.group;
	hipStream_t st = pz::as_stream(stream);
	if (pool_is_global_avg(d, P, Q)) {
		const size_t total = (size_t)g.planes * d->h * d->w;
		const int hw = d->h * d->w;
		pool_global_avg_bwd_kernel<<<(unsigned)((total + 1023) / 1024), 256, 0, st>>>(dy, dx, total, hw,
		                                                                               (unsigned)((((unsigned long long)1 << 32) + hw - 1) / hw));
	} else if (d->mode == 0 && index_ws && pool_lds_window(d) && pool_lds_fits(d, Q, false)) {
		const dim3 grid(g.planes, (d->h + kPoolBwdBand - 1) / kPoolBwdBand);
		if (d->size_h == 3 && d->stride_h == 2) maxpool_bwd_lds_kernel<3, 2><<<grid, 256, 0, st>>>(*d, g, dy, index_ws, dx);
		else if (d->size_h == 2) maxpool_bwd_lds_kernel<2, 2><<<grid, 256, 0, st>>>(*d, g, dy, index_ws, dx);
		else maxpool_bwd_lds_kernel<3, 1><<<grid, 256, 0, st>>>(*d, g, dy, index_ws, dx);
	} else if (d->mode == 0 && index_ws) {
#define PZ_L(SZ, ST) pool_bwd_kernel<0, true, SZ, ST><<<blocks, 256, 0, st>>>(*d, g, dy, x, index_ws, dx)
		PZ_POOL_SPECIALISE(PZ_L);
#undef PZ_L
	} else if (d->mode == 0) {
		pool_bwd_kernel<0, false, 0, 0><<<blocks, 256, 0, st>>>(*d, g, dy, x, index_ws, dx);
	} else if (d->mode == 1) {
		pool_bwd_kernel<1, false, 0, 0><<<blocks, 256, 0, st>>>(*d, g, dy, x, index_ws, dx);
	} else {
		pool_bwd_kernel<2, false, 0, 0><<<blocks, 256, 0, st>>>(*d, g, dy, x, index_ws, dx);
	}
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

}  // extern "C"
