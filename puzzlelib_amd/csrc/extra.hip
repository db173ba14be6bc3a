// Operators beside the ResNet / NiN / LeNet path that other reference modules reach through the same backend object
// (SURVEY.md §8 f3): mask pooling (MaxPool2D(useMask=True) / MaxUnpool2D), local response normalisation, batched
// matrix-vector products, arg-min. Plain HBM-bound kernels: lanes along the contiguous axis, one output per thread or one
// wave per row; none of them is on a timed path, so they are written for clarity and parity.
//   mask pooling  — Cuda/Kernels/Pool.py:7-114 (poolmod.maxpool2d / maxpool2dBackward / maxunpool2d / maxunpool2dBackward):
//                   the mask holds, per pooled element, the flat index h*W + w of its maximum inside the input plane
//                   (-1 for an empty window); first maximum wins (strict >), as in the reference kernel
//   LRN           — Hip/Wrappers/MIOpen.py:691-751 (dnn.lrn / lrnBackward), formulas pinned by
//                   Cuda/Wrappers/CuDnnNorm.py:183-262 (mapLRN2dTest / crossMapLRN2dTest):
//                   y = x / s^beta, s = K + alpha/|window| * sum_window x^2, window = N x N pixels of one map ("map" mode,
//                   |window| = N^2) or N neighbouring maps of one pixel ("cross", |window| = N), clipped at the borders;
//                   dx = dy / s^beta - 2*alpha*beta/|window| * x * sum_window dy * x / s^(beta + 1)
//   matvec        — Cuda/Kernels/MatVec.py:302-345: out[z, i] = alpha*sum_j mat[z, i, j]*vec[z, j] + beta*out (axis 1),
//                   out[z, j] = alpha*sum_i mat[z, i, j]*vec[z, i] + beta*out (axis 0)
#include "common.h"
#include <cfloat>

namespace {

struct MaskPoolGeom {
	int maps, inh, inw, outh, outw, fh, fw, sh, sw, ph, pw;
};

__global__ void __launch_bounds__(256) maskpool_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, int32_t *__restrict__ mask,
                                                            MaskPoolGeom g, size_t total) {
	const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (idx >= total) return;
	const int pw = (int)(idx % g.outw), ph = (int)((idx / g.outw) % g.outh);
	const size_t plane = idx / ((size_t)g.outw * g.outh);
	int h0 = ph * g.sh - g.ph, w0 = pw * g.sw - g.pw;
	const int h1 = min(h0 + g.fh, g.inh), w1 = min(w0 + g.fw, g.inw);
	h0 = max(h0, 0), w0 = max(w0, 0);
	const float *slice = x + plane * g.inh * g.inw;
	float best = -FLT_MAX;
	int arg = -1;
	for (int h = h0; h < h1; ++h)
		for (int w = w0; w < w1; ++w) {
			const float v = slice[h * g.inw + w];
			if (v > best) best = v, arg = h * g.inw + w;
		}
	y[idx] = best;
	mask[idx] = arg;
}

// gather form: an input element collects the gradients of the pooled elements whose mask points at it
__global__ void __launch_bounds__(256) maskpool_bwd_kernel(const float *__restrict__ dy, const int32_t *__restrict__ mask,
                                                            float *__restrict__ dx, MaskPoolGeom g, size_t total) {
	const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (idx >= total) return;
	const int w = (int)(idx % g.inw), h = (int)((idx / g.inw) % g.inh);
	const size_t plane = idx / ((size_t)g.inw * g.inh);
	const int ph0 = (h + g.ph < g.fh) ? 0 : (h + g.ph - g.fh) / g.sh + 1, ph1 = min((h + g.ph) / g.sh + 1, g.outh);
	const int pw0 = (w + g.pw < g.fw) ? 0 : (w + g.pw - g.fw) / g.sw + 1, pw1 = min((w + g.pw) / g.sw + 1, g.outw);
	const size_t off = plane * g.outh * g.outw;
	float acc = 0.f;
	for (int ph = ph0; ph < ph1; ++ph)
		for (int pw = pw0; pw < pw1; ++pw)
			if (mask[off + ph * g.outw + pw] == h * g.inw + w) acc += dy[off + ph * g.outw + pw];
	dx[idx] = acc;
}

// unpool: out (zeroed by the caller) [plane][mask] = in; backward: dx = dy[plane][mask]
__global__ void __launch_bounds__(256) unpool_fwd_kernel(const float *__restrict__ x, const int32_t *__restrict__ mask, float *__restrict__ y,
                                                          size_t in_plane, size_t out_plane, size_t total) {
	const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (idx >= total) return;
	const int m = mask[idx];
	if (m >= 0) y[(idx / in_plane) * out_plane + m] = x[idx];
}

__global__ void __launch_bounds__(256) unpool_bwd_kernel(const float *__restrict__ dy, const int32_t *__restrict__ mask, float *__restrict__ dx,
                                                          size_t in_plane, size_t out_plane, size_t total) {
	const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (idx >= total) return;
	const int m = mask[idx];
	dx[idx] = m >= 0 ? dy[(idx / in_plane) * out_plane + m] : 0.f;
}

struct LrnGeom {
	int c, h, w, n, cross;
	float alpha, beta, k;
};

template <typename F>
__device__ __forceinline__ void lrn_window(const LrnGeom &g, int ch, int y, int x, F visit) {
	const int behind = (g.n - 1) / 2, ahead = g.n - behind;
	if (g.cross) {
		for (int cc = max(0, ch - behind); cc < min(g.c, ch + ahead); ++cc) visit((cc * g.h + y) * g.w + x);
	} else {
		for (int yy = max(0, y - behind); yy < min(g.h, y + ahead); ++yy)
			for (int xx = max(0, x - behind); xx < min(g.w, x + ahead); ++xx) visit((ch * g.h + yy) * g.w + xx);
	}
}

__global__ void __launch_bounds__(256) lrn_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, float *__restrict__ scale, LrnGeom g,
                                                       size_t total) {
	const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (idx >= total) return;
	const int xx = (int)(idx % g.w), yy = (int)((idx / g.w) % g.h), ch = (int)((idx / ((size_t)g.w * g.h)) % g.c);
	const float *img = x + (idx / ((size_t)g.w * g.h * g.c)) * ((size_t)g.c * g.h * g.w);
	float ss = 0.f;
	lrn_window(g, ch, yy, xx, [&](int off) { ss = __builtin_fmaf(img[off], img[off], ss); });
	const float s = g.k + ss * g.alpha / (g.cross ? (float)g.n : (float)(g.n * g.n));
	if (scale) scale[idx] = s;
	y[idx] = x[idx] * __powf(s, -g.beta);
}

__global__ void __launch_bounds__(256) lrn_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ scale,
                                                       float *__restrict__ dx, LrnGeom g, size_t total) {
	const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (idx >= total) return;
	const int xx = (int)(idx % g.w), yy = (int)((idx / g.w) % g.h), ch = (int)((idx / ((size_t)g.w * g.h)) % g.c);
	const size_t base = (idx / ((size_t)g.w * g.h * g.c)) * ((size_t)g.c * g.h * g.w);
	float acc = 0.f;
	lrn_window(g, ch, yy, xx, [&](int off) {
		const float s = scale[base + off];
		acc = __builtin_fmaf(dy[base + off] * x[base + off], __powf(s, -g.beta - 1.f), acc);
	});
	const float coef = 2.f * g.alpha * g.beta / (g.cross ? (float)g.n : (float)(g.n * g.n));
	dx[idx] = dy[idx] * __powf(scale[idx], -g.beta) - coef * x[idx] * acc;
}

// one wave per (z, row): dot of a matrix row with the batch's vector
__global__ void __launch_bounds__(256) matvec_rows_kernel(const float *__restrict__ mat, const float *__restrict__ vec, float *__restrict__ out,
                                                           int h, int w, float alpha, float beta, size_t rows) {
	const int lane = threadIdx.x & 63;
	const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (row >= rows) return;
	const float *r = mat + row * w, *v = vec + (row / h) * w;
	float acc = 0.f;
	for (int j = lane; j < w; j += 64) acc = __builtin_fmaf(r[j], v[j], acc);
	acc = wave_sum(acc);
	if (lane == 0) out[row] = (beta == 0.f ? 0.f : beta * out[row]) + alpha * acc;
}

// one thread per (z, column): lanes along the contiguous axis
__global__ void __launch_bounds__(256) matvec_cols_kernel(const float *__restrict__ mat, const float *__restrict__ vec, float *__restrict__ out,
                                                           int h, int w, float alpha, float beta) {
	const int col = blockIdx.x * 256 + threadIdx.x, z = blockIdx.z;
	if (col >= w) return;
	const float *base = mat + (size_t)z * h * w + col, *v = vec + (size_t)z * h;
	float acc = 0.f;
	for (int i = 0; i < h; ++i) acc = __builtin_fmaf(base[(size_t)i * w], v[i], acc);
	float *o = out + (size_t)z * w + col;
	*o = (beta == 0.f ? 0.f : beta * *o) + alpha * acc;
}

__device__ __forceinline__ void argmin_combine(float &v, int &i, float ov, int oi) {
	if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
}

__global__ void __launch_bounds__(256) argmin_rows_kernel(const float *__restrict__ t, int rows, int cols, int32_t *__restrict__ out) {
	const int lane = threadIdx.x & 63;
	const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (row >= rows) return;
	const float *r = t + (size_t)row * cols;
	float best = FLT_MAX;
	int idx = INT_MAX;
	for (int j = lane; j < cols; j += 64) argmin_combine(best, idx, r[j], j);
#pragma unroll
	for (int m = 32; m > 0; m >>= 1) {
		const float ov = __shfl_xor(best, m, 64);
		const int oi = __shfl_xor(idx, m, 64);
		argmin_combine(best, idx, ov, oi);
	}
	if (lane == 0) out[row] = idx == INT_MAX ? 0 : idx;
}

__global__ void __launch_bounds__(256) argmin_cols_kernel(const float *__restrict__ t, int h, int w, int32_t *__restrict__ out) {
	const int col = blockIdx.x * 256 + threadIdx.x, z = blockIdx.z;
	if (col >= w) return;
	const float *base = t + (size_t)z * h * w + col;
	float best = base[0];
	int idx = 0;
	for (int i = 1; i < h; ++i) {
		const float v = base[(size_t)i * w];
		if (v < best) { best = v; idx = i; }
	}
	out[(size_t)z * w + col] = idx;
}

inline int grid_for(size_t total) { return (int)((total + 255) / 256); }

}  // namespace

extern "C" {

int pz_maskpool2d_fwd(const pz_pool_desc *d, const float *x, float *y, int32_t *mask, pz_stream_t stream) {
	int p, q;
	if (int rc = pz_pool2d_out_shape(d, &p, &q)) return rc;
	PZ_REQUIRE(x && y && mask, "pz_maskpool2d_fwd: null tensor");
	const size_t total = (size_t)d->n * d->c * p * q;
	PZ_REQUIRE(total < ((size_t)1 << 31) * 256, "pz_maskpool2d_fwd: tensor too large");
	const MaskPoolGeom g{d->c, d->h, d->w, p, q, d->size_h, d->size_w, d->stride_h, d->stride_w, d->pad_h, d->pad_w};
	maskpool_fwd_kernel<<<grid_for(total), 256, 0, pz::as_stream(stream)>>>(x, y, mask, g, total);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_maskpool2d_bwd(const pz_pool_desc *d, const float *dy, const int32_t *mask, float *dx, pz_stream_t stream) {
	int p, q;
	if (int rc = pz_pool2d_out_shape(d, &p, &q)) return rc;
	PZ_REQUIRE(dy && dx && mask, "pz_maskpool2d_bwd: null tensor");
	const size_t total = (size_t)d->n * d->c * d->h * d->w;
	const MaskPoolGeom g{d->c, d->h, d->w, p, q, d->size_h, d->size_w, d->stride_h, d->stride_w, d->pad_h, d->pad_w};
	maskpool_bwd_kernel<<<grid_for(total), 256, 0, pz::as_stream(stream)>>>(dy, mask, dx, g, total);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_maxunpool2d_fwd(const float *x, const int32_t *mask, float *y, size_t planes, size_t in_plane, size_t out_plane,
                       pz_stream_t stream) {
	PZ_REQUIRE(x && mask && y && planes > 0 && in_plane > 0 && out_plane > 0, "pz_maxunpool2d_fwd: bad arguments");
	hipStream_t st = pz::as_stream(stream);
	PZ_HIP(hipMemsetAsync(y, 0, planes * out_plane * sizeof(float), st));
	unpool_fwd_kernel<<<grid_for(planes * in_plane), 256, 0, st>>>(x, mask, y, in_plane, out_plane, planes * in_plane);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_maxunpool2d_bwd(const float *dy, const int32_t *mask, float *dx, size_t planes, size_t in_plane, size_t out_plane,
                       pz_stream_t stream) {
	PZ_REQUIRE(dy && mask && dx && planes > 0 && in_plane > 0 && out_plane > 0, "pz_maxunpool2d_bwd: bad arguments");
	unpool_bwd_kernel<<<grid_for(planes * in_plane), 256, 0, pz::as_stream(stream)>>>(dy, mask, dx, in_plane, out_plane, planes * in_plane);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_lrn_fwd(const float *x, float *y, float *scale, int n, int c, int h, int w, int size, float alpha, float beta, float k,
               int cross, pz_stream_t stream) {
	PZ_REQUIRE(x && y && n > 0 && c > 0 && h > 0 && w > 0 && size > 0, "pz_lrn_fwd: bad arguments");
	const size_t total = (size_t)n * c * h * w;
	const LrnGeom g{c, h, w, size, cross, alpha, beta, k};
	lrn_fwd_kernel<<<grid_for(total), 256, 0, pz::as_stream(stream)>>>(x, y, scale, g, total);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_lrn_bwd(const float *x, const float *dy, const float *scale, float *dx, int n, int c, int h, int w, int size, float alpha,
               float beta, float k, int cross, pz_stream_t stream) {
	PZ_REQUIRE(x && dy && scale && dx && n > 0 && c > 0 && h > 0 && w > 0 && size > 0, "pz_lrn_bwd: bad arguments");
	const size_t total = (size_t)n * c * h * w;
	const LrnGeom g{c, h, w, size, cross, alpha, beta, k};
	lrn_bwd_kernel<<<grid_for(total), 256, 0, pz::as_stream(stream)>>>(x, dy, scale, dx, g, total);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_matvec(const float *mat, const float *vec, float *out, int z, int h, int w, int axis, float alpha, float beta, pz_stream_t stream) {
	PZ_REQUIRE(mat && vec && out && z > 0 && h > 0 && w > 0 && (axis == 0 || axis == 1) && z <= 65535, "pz_matvec: bad arguments");
	hipStream_t st = pz::as_stream(stream);
	if (axis == 1) {
		const size_t rows = (size_t)z * h;
		matvec_rows_kernel<<<(int)((rows + 3) / 4), 256, 0, st>>>(mat, vec, out, h, w, alpha, beta, rows);
	} else {
		matvec_cols_kernel<<<dim3(pz::ceil_div(w, 256), 1, z), 256, 0, st>>>(mat, vec, out, h, w, alpha, beta);
	}
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_argmin_rows(const float *t, int rows, int cols, int32_t *out, pz_stream_t stream) {
	PZ_REQUIRE(t && out && rows > 0 && cols > 0, "pz_argmin_rows: bad arguments");
	argmin_rows_kernel<<<pz::ceil_div(rows, 4), 256, 0, pz::as_stream(stream)>>>(t, rows, cols, out);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_argmin_cols(const float *t, int z, int h, int w, int32_t *out, pz_stream_t stream) {
	PZ_REQUIRE(t && out && z > 0 && h > 0 && w > 0 && z <= 65535, "pz_argmin_cols: bad arguments");
	argmin_cols_kernel<<<dim3(pz::ceil_div(w, 256), 1, z), 256, 0, pz::as_stream(stream)>>>(t, h, w, out);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

}  // extern "C"

// SVM cost (Cuda/Kernels/Costs.py:109-130,250-276): one-vs-all hinge over the class axis, L1 or squared (L2);
// grad per score, per-element error terms to `terms` (summed afterwards by pz_asum: deterministic, the reference
// uses atomicAdd). cls = +1 for the labelled class, -1 otherwise.
namespace {
__global__ void __launch_bounds__(256) svm_cost_kernel(const float *__restrict__ scores, const int32_t *__restrict__ labels, int cases,
                                                        int spatial, int samples, int squared, float *__restrict__ grad,
                                                        float *__restrict__ terms, size_t total) {
	const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (idx >= total) return;
	const size_t map_stride = (size_t)spatial * cases;
	const int b = (int)(idx / map_stride), m = (int)(idx % spatial), c = (int)((idx / spatial) % cases);
	const float score = scores[idx];
	const float cls = labels[(size_t)b * spatial + m] == c ? 1.f : -1.f;
	const float margin = fmaxf(0.f, 1.f - score * cls);
	if (squared) {
		grad[idx] = 2.f * cls * margin / cases / samples;
		terms[idx] = margin * margin / cases / spatial;
	} else {
		grad[idx] = score * cls < 1.f ? cls / cases / samples : 0.f;
		terms[idx] = margin / cases / spatial;
	}
}
}  // namespace

extern "C" int pz_svm_cost(const float *scores, const int32_t *labels, int samples, int cases, int spatial, int squared, float *grad,
                           float *terms, pz_stream_t stream) {
	PZ_REQUIRE(scores && labels && grad && terms && samples > 0 && cases > 0 && spatial > 0, "pz_svm_cost: bad arguments");
	const size_t total = (size_t)samples * cases * spatial;
	svm_cost_kernel<<<(int)((total + 255) / 256), 256, 0, pz::as_stream(stream)>>>(scores, labels, cases, spatial, samples, squared, grad,
	                                                                             terms, total);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}
