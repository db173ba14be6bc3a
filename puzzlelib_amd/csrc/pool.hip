// 2-D pooling (max / average incl. padding / average excl. padding), NCHW fp32, HBM-bound gather kernels.
// Replaces DnnContext.poolNd / poolNdBackward — Hip/Wrappers/MIOpen.py:549-598. The training-mode "workspace" is
// this library's own format: one byte per output element holding the window-local arg-max (r*size_w + s, first
// maximum in row-major window order). Backward is a gather over the (few) windows covering each input pixel:
// no atomics, deterministic.
#include "common.h"
#include <cfloat>

namespace {

__global__ void __launch_bounds__(256) pool_fwd_kernel(pz_pool_desc d, int P, int Q, const float *__restrict__ x,
                                                        float *__restrict__ y, uint8_t *__restrict__ idx) {
	const size_t total = (size_t)d.n * d.c * P * Q;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
		const int q = (int)(i % Q), p = (int)((i / Q) % P);
		const size_t nc = i / ((size_t)Q * P);
		const float *img = x + nc * d.h * d.w;
		const int h0 = p * d.stride_h - d.pad_h, w0 = q * d.stride_w - d.pad_w;

		if (d.mode == 0) {
			float best = -FLT_MAX;
			int bi = 0;
			bool found = false;
			for (int r = 0; r < d.size_h; ++r) {
				const int hh = h0 + r;
				if ((unsigned)hh >= (unsigned)d.h) continue;
				for (int s = 0; s < d.size_w; ++s) {
					const int ww = w0 + s;
					if ((unsigned)ww >= (unsigned)d.w) continue;
					const float v = img[hh * d.w + ww];
					if (!found || v > best) { best = v; bi = r * d.size_w + s; found = true; }
				}
			}
			y[i] = found ? best : -INFINITY;          // NumpyDnn.pool2d pads with -inf
			if (idx) idx[i] = (uint8_t)bi;
		} else {
			float s = 0.f;
			int cnt = 0;
			for (int r = 0; r < d.size_h; ++r) {
				const int hh = h0 + r;
				if ((unsigned)hh >= (unsigned)d.h) continue;
				for (int t = 0; t < d.size_w; ++t) {
					const int ww = w0 + t;
					if ((unsigned)ww >= (unsigned)d.w) continue;
					s += img[hh * d.w + ww];
					++cnt;
				}
			}
			const float div = d.mode == 1 ? (float)(d.size_h * d.size_w) : (float)(cnt > 0 ? cnt : 1);
			y[i] = s / div;
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

__global__ void __launch_bounds__(256) pool_bwd_kernel(pz_pool_desc d, int P, int Q, const float *__restrict__ dy,
                                                        const float *__restrict__ x, const uint8_t *__restrict__ idx,
                                                        float *__restrict__ dx) {
	const size_t total = (size_t)d.n * d.c * d.h * d.w;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
		const int ww = (int)(i % d.w), hh = (int)((i / d.w) % d.h);
		const size_t nc = i / ((size_t)d.w * d.h);
		const float *gimg = dy + nc * P * Q;

		// windows p with p*stride - pad <= hh < p*stride - pad + size
		const int hp = hh + d.pad_h, wp = ww + d.pad_w;
		int p_lo = hp - d.size_h + 1;
		p_lo = p_lo <= 0 ? 0 : (p_lo + d.stride_h - 1) / d.stride_h;
		int p_hi = hp / d.stride_h;
		p_hi = p_hi >= P ? P - 1 : p_hi;
		int q_lo = wp - d.size_w + 1;
		q_lo = q_lo <= 0 ? 0 : (q_lo + d.stride_w - 1) / d.stride_w;
		int q_hi = wp / d.stride_w;
		q_hi = q_hi >= Q ? Q - 1 : q_hi;

		float s = 0.f;
		for (int p = p_lo; p <= p_hi; ++p)
			for (int q = q_lo; q <= q_hi; ++q) {
				const float g = gimg[p * Q + q];
				if (d.mode == 0) {
					const int r = hp - p * d.stride_h, t = wp - q * d.stride_w;
					int win;
					if (idx) {
						win = idx[nc * P * Q + p * Q + q];
					} else {
						// recompute the first maximum of the window
						const float *img = x + nc * d.h * d.w;
						const int h0 = p * d.stride_h - d.pad_h, w0 = q * d.stride_w - d.pad_w;
						float best = -FLT_MAX;
						bool found = false;
						win = 0;
						for (int rr = 0; rr < d.size_h; ++rr)
							for (int tt = 0; tt < d.size_w; ++tt) {
								const int a = h0 + rr, b = w0 + tt;
								if ((unsigned)a >= (unsigned)d.h || (unsigned)b >= (unsigned)d.w) continue;
								const float v = img[a * d.w + b];
								if (!found || v > best) { best = v; win = rr * d.size_w + tt; found = true; }
							}
					}
					if (win == r * d.size_w + t) s += g;
				} else if (d.mode == 1) {
					s += g / (float)(d.size_h * d.size_w);
				} else {
					s += g / (float)valid_taps(d, p, q);
				}
			}
		dx[i] = s;
	}
}

int pool_check(const pz_pool_desc *d, int *P, int *Q) {
	PZ_REQUIRE(d != nullptr, "pool: null descriptor");
	PZ_REQUIRE(d->n > 0 && d->c > 0 && d->h > 0 && d->w > 0, "pool: non-positive tensor dimension");
	PZ_REQUIRE(d->size_h > 0 && d->size_w > 0 && d->stride_h > 0 && d->stride_w > 0 && d->pad_h >= 0 && d->pad_w >= 0,
	           "pool: invalid window/stride/pad");
	PZ_REQUIRE(d->size_h * d->size_w <= 256, "pool: window larger than 256 taps");
	PZ_REQUIRE(d->mode >= 0 && d->mode <= 2, "pool: unknown mode %d", d->mode);
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
	const size_t total = (size_t)d->n * d->c * P * Q;
	pool_fwd_kernel<<<pz::stream_grid(total, 256), 256, 0, pz::as_stream(stream)>>>(*d, P, Q, x, y, index_ws);
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
	const size_t total = (size_t)d->n * d->c * d->h * d->w;
	pool_bwd_kernel<<<pz::stream_grid(total, 256), 256, 0, pz::as_stream(stream)>>>(*d, P, Q, dy, x, index_ws, dx);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

}  // extern "C"
