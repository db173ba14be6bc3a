// Reductions and matrix-vector helpers (wavefront-shuffle kernels, 64 lanes):
//   matsum rows/cols (bias gradient), argmax rows/cols, bias add, count_neq (accuracy), min/max, dot, asum.
// Replaces MatModule — Cuda/Kernels/MatVec.py:231-374 (kernels :8-171), the ReductionKernel users in
// Cuda/GPUArray.py:80-103 and Cuda/Kernels/Costs.py:178-182, and BlasContext.dot/l1norm — CuBlas.c:486-499.
// All multi-workgroup reductions are two-stage with a fixed order (deterministic), final result stays on device.
#include "common.h"
#include <cfloat>
#include <climits>

namespace {

// out[row] = beta*out[row] + alpha*sum_j t[row, j] : one wave per row, 4 rows per workgroup
__global__ void __launch_bounds__(256) sum_rows_kernel(const float *__restrict__ t, int rows, int cols, float *__restrict__ out,
                                                        float alpha, float beta) {
	const int lane = threadIdx.x & 63;
	const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (row >= rows) return;
	const float *r = t + (size_t)row * cols;
	float acc = 0.f;
	for (int j = lane; j < cols; j += 64) acc += r[j];
	acc = wave_sum(acc);
	if (lane == 0) out[row] = (beta == 0.f ? 0.f : beta * out[row]) + alpha * acc;
}

// tensor (z, h, w): out[z, col] = beta*out + alpha*sum_i t[z, i, col].
// 64 columns x 4 row-groups per workgroup, LDS combine -> coalesced along w, h split over the 4 waves.
__global__ void __launch_bounds__(256) sum_cols_kernel(const float *__restrict__ t, int h, int w, float *__restrict__ out,
                                                        float alpha, float beta) {
	__shared__ float part[4][64];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int col = blockIdx.x * 64 + lane, z = blockIdx.z;
	float acc = 0.f;
	if (col < w) {
		const float *base = t + (size_t)z * h * w + col;
		for (int i = wv; i < h; i += 4) acc += base[(size_t)i * w];
	}
	part[wv][lane] = acc;
	__syncthreads();
	if (wv == 0 && col < w) {
		const float s = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
		float *o = out + (size_t)z * w + col;
		*o = (beta == 0.f ? 0.f : beta * *o) + alpha * s;
	}
}

// first maximum wins (np.argmax semantics)
__device__ __forceinline__ void argmax_combine(float &v, int &i, float ov, int oi) {
	if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}

__global__ void __launch_bounds__(256) argmax_rows_kernel(const float *__restrict__ t, int rows, int cols, int32_t *__restrict__ out) {
	const int lane = threadIdx.x & 63;
	const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (row >= rows) return;
	const float *r = t + (size_t)row * cols;
	float best = -FLT_MAX;
	int idx = INT_MAX;
	for (int j = lane; j < cols; j += 64) argmax_combine(best, idx, r[j], j);
#pragma unroll
	for (int m = 32; m > 0; m >>= 1) {
		const float ov = __shfl_xor(best, m, 64);
		const int oi = __shfl_xor(idx, m, 64);
		argmax_combine(best, idx, ov, oi);
	}
	if (lane == 0) out[row] = idx == INT_MAX ? 0 : idx;
}

__global__ void __launch_bounds__(256) argmax_cols_kernel(const float *__restrict__ t, int h, int w, int32_t *__restrict__ out) {
	const int col = blockIdx.x * 256 + threadIdx.x, z = blockIdx.z;
	if (col >= w) return;
	const float *base = t + (size_t)z * h * w + col;
	float best = base[0];
	int idx = 0;
	for (int i = 1; i < h; ++i) {
		const float v = base[(size_t)i * w];
		if (v > best) { best = v; idx = i; }
	}
	out[(size_t)z * w + col] = idx;
}

// out[z, i, j] = mat[z, i, j] + (axis == 1 ? vec[z, j % veclen] : vec[z, i])
__global__ void __launch_bounds__(256) bias_add_kernel(float *out, const float *mat, const float *__restrict__ vec, int n, int m,
                                                        int veclen, int axis) {
	const int z = blockIdx.z;
	const size_t total = (size_t)n * m;
	const float *mz = mat + (size_t)z * total;
	float *oz = out + (size_t)z * total;
	const float *vz = vec + (size_t)z * veclen;
	for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
		const int i = (int)(e / m), j = (int)(e - (size_t)i * m);
		oz[e] = mz[e] + (axis == 1 ? vz[j % veclen] : vz[i % veclen]);
	}
}

// ---- generic two-stage scalar reductions -------------------------------------------------------
enum { RED_NEQ = 0, RED_DOT, RED_ASUM, RED_MIN_F, RED_MAX_F, RED_MIN_I, RED_MAX_I, RED_BCE_ACC, RED_L1HINGE_ACC };

template <int KIND>
__device__ __forceinline__ float red_map(const void *x, const void *y, size_t i) {
	if (KIND == RED_NEQ) return ((const int32_t *)x)[i] != ((const int32_t *)y)[i] ? 1.f : 0.f;
	if (KIND == RED_DOT) return ((const float *)x)[i] * ((const float *)y)[i];
	if (KIND == RED_ASUM) return fabsf(((const float *)x)[i]);
	// the reference's accuracy reductions (Cuda/Kernels/Costs.py:184-203): x = scores / distances, y = int32 labels
	if (KIND == RED_BCE_ACC) return (((const int32_t *)y)[i] == 1 ? ((const float *)x)[i] <= 0.f : ((const float *)x)[i] > 0.f) ? 1.f : 0.f;
	if (KIND == RED_L1HINGE_ACC) return ((int32_t)(((const float *)x)[i] <= 1.f) != ((const int32_t *)y)[i]) ? 1.f : 0.f;
	return ((const float *)x)[i];
}

// klDivergence (Cuda/Kernels/Costs.py:190-197): grad[i] = (y[i] - x[i]) * gradnorm on the way, sum of y (log y - log x) where y > 0
__global__ void __launch_bounds__(256) kl_stage1(const float *x, const float *y, float *grad, float gradnorm, size_t n, float *part) {
	__shared__ float red[16];
	float acc = 0.f;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const float xv = x[i], yv = y[i];
		grad[i] = (yv - xv) * gradnorm;
		acc += yv > 0.f ? yv * (logf(yv) - logf(xv)) : 0.f;
	}
	acc = block_sum(acc, red);
	if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

template <int KIND>
__global__ void __launch_bounds__(256) red_sum_stage1(const void *x, const void *y, size_t n, float *part) {
	__shared__ float red[16];
	float acc = 0.f;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
		acc += red_map<KIND>(x, y, i);
	acc = block_sum(acc, red);
	if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

__global__ void __launch_bounds__(256) red_sum_stage2(const float *part, int nparts, float *out) {
	__shared__ float red[16];
	float acc = 0.f;
	for (int i = threadIdx.x; i < nparts; i += blockDim.x) acc += part[i];
	acc = block_sum(acc, red);
	if (threadIdx.x == 0) *out = acc;
}

template <typename T, bool IS_MAX>
__global__ void __launch_bounds__(256) red_minmax_kernel(const T *x, size_t n, T *part, T neutral) {
	__shared__ T red[256];
	T acc = neutral;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const T v = x[i];
		acc = IS_MAX ? (v > acc ? v : acc) : (v < acc ? v : acc);
	}
	red[threadIdx.x] = acc;
	__syncthreads();
	for (int s = 128; s > 0; s >>= 1) {
		if ((int)threadIdx.x < s) {
			const T o = red[threadIdx.x + s];
			red[threadIdx.x] = IS_MAX ? (o > red[threadIdx.x] ? o : red[threadIdx.x]) : (o < red[threadIdx.x] ? o : red[threadIdx.x]);
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// scratch for stage-1 partials: one small per-device buffer, reused in stream order
float *g_scratch = nullptr;
constexpr int kMaxParts = 1024;

int scratch(float **p) {
	if (!g_scratch) PZ_HIP(hipMalloc((void **)&g_scratch, kMaxParts * sizeof(float)));
	*p = g_scratch;
	return PZ_OK;
}

template <int KIND>
int reduce_sum(const void *x, const void *y, size_t n, float *out, hipStream_t st) {
	PZ_REQUIRE(out != nullptr, "reduce: null output");
	if (n == 0) {
		PZ_HIP(hipMemsetAsync(out, 0, sizeof(float), st));
		return PZ_OK;
	}
	float *part;
	if (int rc = scratch(&part)) return rc;
	int blocks = pz::stream_grid(n, 256 * 8);
	if (blocks > kMaxParts) blocks = kMaxParts;
	red_sum_stage1<KIND><<<blocks, 256, 0, st>>>(x, y, n, part);
	PZ_LAUNCH_CHECK();
	red_sum_stage2<<<1, 256, 0, st>>>(part, blocks, out);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

template <typename T, bool IS_MAX>
int reduce_minmax(const T *x, size_t n, T *out, T neutral, hipStream_t st) {
	PZ_REQUIRE(x && out && n > 0, "min/max: empty input");
	float *partf;
	if (int rc = scratch(&partf)) return rc;
	T *part = reinterpret_cast<T *>(partf);
	int blocks = pz::stream_grid(n, 256 * 8);
	if (blocks > kMaxParts) blocks = kMaxParts;
	red_minmax_kernel<T, IS_MAX><<<blocks, 256, 0, st>>>(x, n, part, neutral);
	PZ_LAUNCH_CHECK();
	red_minmax_kernel<T, IS_MAX><<<1, 256, 0, st>>>(part, (size_t)blocks, out, neutral);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

}  // namespace

extern "C" {

int pz_reduce_sum_rows(const float *t, int rows, int cols, float *out, float alpha, float beta, pz_stream_t stream) {
	PZ_REQUIRE(t && out && rows > 0 && cols > 0, "pz_reduce_sum_rows: bad arguments");
	sum_rows_kernel<<<pz::ceil_div(rows, 4), 256, 0, pz::as_stream(stream)>>>(t, rows, cols, out, alpha, beta);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_reduce_sum_cols(const float *t, int z, int h, int w, float *out, float alpha, float beta, pz_stream_t stream) {
	PZ_REQUIRE(t && out && z > 0 && h > 0 && w > 0 && z <= 65535, "pz_reduce_sum_cols: bad arguments");
	sum_cols_kernel<<<dim3(pz::ceil_div(w, 64), 1, z), 256, 0, pz::as_stream(stream)>>>(t, h, w, out, alpha, beta);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_argmax_rows(const float *t, int rows, int cols, int32_t *out, pz_stream_t stream) {
	PZ_REQUIRE(t && out && rows > 0 && cols > 0, "pz_argmax_rows: bad arguments");
	argmax_rows_kernel<<<pz::ceil_div(rows, 4), 256, 0, pz::as_stream(stream)>>>(t, rows, cols, out);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_argmax_cols(const float *t, int z, int h, int w, int32_t *out, pz_stream_t stream) {
	PZ_REQUIRE(t && out && z > 0 && h > 0 && w > 0 && z <= 65535, "pz_argmax_cols: bad arguments");
	argmax_cols_kernel<<<dim3(pz::ceil_div(w, 256), 1, z), 256, 0, pz::as_stream(stream)>>>(t, h, w, out);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_bias_add(float *out, const float *mat, const float *vec, int z, int n, int m, int veclen, int axis, pz_stream_t stream) {
	PZ_REQUIRE(out && mat && vec && z > 0 && n > 0 && m > 0 && z <= 65535, "pz_bias_add: bad arguments");
	PZ_REQUIRE(veclen > 0 && ((axis == 0 && n % veclen == 0) || (axis == 1 && m % veclen == 0)),
	           "pz_bias_add: vector length %d does not tile the %d x %d matrix along axis %d", veclen, n, m, axis);
	bias_add_kernel<<<dim3(pz::stream_grid((size_t)n * m, 256), 1, z), 256, 0, pz::as_stream(stream)>>>(out, mat, vec, n, m, veclen, axis);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_count_neq_i32(const int32_t *x, const int32_t *y, size_t count, float *out, pz_stream_t stream) {
	return reduce_sum<RED_NEQ>(x, y, count, out, pz::as_stream(stream));
}

int pz_cost_accuracy(int kind, const float *x, const int32_t *labels, size_t count, float *out, pz_stream_t stream) {
	PZ_REQUIRE(kind == 0 || kind == 1, "pz_cost_accuracy: kind %d", kind);
	return kind == 0 ? reduce_sum<RED_BCE_ACC>(x, labels, count, out, pz::as_stream(stream))
	                 : reduce_sum<RED_L1HINGE_ACC>(x, labels, count, out, pz::as_stream(stream));
}

int pz_kl_divergence(const float *x, const float *y, float *grad, float gradnorm, size_t count, float *out, pz_stream_t stream) {
	PZ_REQUIRE(x && y && grad && out, "pz_kl_divergence: null tensor");
	hipStream_t st = pz::as_stream(stream);
	if (count == 0) {
		PZ_HIP(hipMemsetAsync(out, 0, sizeof(float), st));
		return PZ_OK;
	}
	float *part;
	if (int rc = scratch(&part)) return rc;
	int blocks = pz::stream_grid(count, 256 * 8);
	if (blocks > kMaxParts) blocks = kMaxParts;
	kl_stage1<<<blocks, 256, 0, st>>>(x, y, grad, gradnorm, count, part);
	PZ_LAUNCH_CHECK();
	red_sum_stage2<<<1, 256, 0, st>>>(part, blocks, out);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_dot(const float *x, const float *y, size_t count, float *out, pz_stream_t stream) {
	return reduce_sum<RED_DOT>(x, y, count, out, pz::as_stream(stream));
}

int pz_asum(const float *x, size_t count, float *out, pz_stream_t stream) {
	return reduce_sum<RED_ASUM>(x, nullptr, count, out, pz::as_stream(stream));
}

int pz_reduce_minmax_f32(const float *x, size_t count, int is_max, float *out, pz_stream_t stream) {
	return is_max ? reduce_minmax<float, true>(x, count, out, -FLT_MAX, pz::as_stream(stream))
	              : reduce_minmax<float, false>(x, count, out, FLT_MAX, pz::as_stream(stream));
}

int pz_reduce_minmax_i32(const int32_t *x, size_t count, int is_max, int32_t *out, pz_stream_t stream) {
	return is_max ? reduce_minmax<int32_t, true>(x, count, out, INT_MIN, pz::as_stream(stream))
	              : reduce_minmax<int32_t, false>(x, count, out, INT_MAX, pz::as_stream(stream));
}

}  // extern "C"
