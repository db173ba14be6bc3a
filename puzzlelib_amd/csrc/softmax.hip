// Channel softmax (forward/backward) and the fused softmax + cross-entropy gradient/error kernel.
// Tensors are viewed as (n, c, spatial); the softmax runs over c for every (n, spatial) position.
// Replaces DnnContext.softmaxNd/softmaxNdBackward — Hip/Wrappers/MIOpen.py:601-631 ("accurate" = max-subtracted,
// channel mode) and CostModule.crossEntropy — Cuda/Kernels/Costs.py:79-106,213-247. The reference accumulates the
// error with atomicAdd; here per-position losses go to a workspace and are summed in a fixed order (deterministic).
#include "common.h"
#include <cfloat>

namespace {

// one wave per (n, spatial) position, lanes stride over channels (stride `spatial` elements in memory)
__global__ void __launch_bounds__(256) softmax_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, int n, int c, int sp) {
	const int lane = threadIdx.x & 63;
	const long pos = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (pos >= (long)n * sp) return;
	const int b = (int)(pos / sp), m = (int)(pos % sp);
	const float *xi = x + (size_t)b * c * sp + m;
	float *yi = y + (size_t)b * c * sp + m;

	float mx = -FLT_MAX;
	for (int j = lane; j < c; j += 64) mx = fmaxf(mx, xi[(size_t)j * sp]);
	mx = wave_max(mx);

	float sum = 0.f;
	for (int j = lane; j < c; j += 64) sum += expf(xi[(size_t)j * sp] - mx);
	sum = wave_sum(sum);

	for (int j = lane; j < c; j += 64) yi[(size_t)j * sp] = expf(xi[(size_t)j * sp] - mx) / sum;
}

// dx = y * (dy - sum_c(y*dy))
__global__ void __launch_bounds__(256) softmax_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                           float *__restrict__ dx, int n, int c, int sp) {
	const int lane = threadIdx.x & 63;
	const long pos = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (pos >= (long)n * sp) return;
	const int b = (int)(pos / sp), m = (int)(pos % sp);
	const size_t base = (size_t)b * c * sp + m;

	float dot = 0.f;
	for (int j = lane; j < c; j += 64) dot += y[base + (size_t)j * sp] * dy[base + (size_t)j * sp];
	dot = wave_sum(dot);

	for (int j = lane; j < c; j += 64) {
		const size_t o = base + (size_t)j * sp;
		dx[o] = y[o] * (dy[o] - dot);
	}
}

// grad[b,cls,m] = w[cls]*((cls==label) - p)/n ; loss[pos] = -w[label]*log(p[label])/spatial
__global__ void __launch_bounds__(256) cross_entropy_kernel(const float *__restrict__ scores, const int32_t *__restrict__ labels,
                                                             const float *__restrict__ weights, int n, int c, int sp,
                                                             float *__restrict__ grad, float *__restrict__ loss) {
	const int lane = threadIdx.x & 63;
	const long pos = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (pos >= (long)n * sp) return;
	const int b = (int)(pos / sp), m = (int)(pos % sp);
	const size_t base = (size_t)b * c * sp + m;
	const int label = labels[(size_t)b * sp + m];

	float mx = -FLT_MAX;
	for (int j = lane; j < c; j += 64) mx = fmaxf(mx, scores[base + (size_t)j * sp]);
	mx = wave_max(mx);

	float sum = 0.f;
	for (int j = lane; j < c; j += 64) sum += expf(scores[base + (size_t)j * sp] - mx);
	sum = wave_sum(sum);

	const float inv_n = 1.f / (float)n;
	float mine = 0.f;
	for (int j = lane; j < c; j += 64) {
		const float p = expf(scores[base + (size_t)j * sp] - mx) / sum;
		const float w = weights ? weights[j] : 1.f;
		grad[base + (size_t)j * sp] = w * ((j == label ? 1.f : 0.f) - p) * inv_n;
		if (j == label) mine = -w * logf(p) / (float)sp;
	}
	mine = wave_sum(mine);
	if (lane == 0) loss[pos] = mine;
}

__global__ void __launch_bounds__(256) sum_to_scalar_kernel(const float *__restrict__ v, long count, float *__restrict__ out) {
	__shared__ float red[16];
	float acc = 0.f;
	for (long i = threadIdx.x; i < count; i += blockDim.x) acc += v[i];
	acc = block_sum(acc, red);
	if (threadIdx.x == 0) *out = acc;
}

int sm_check(int n, int c, int sp) {
	PZ_REQUIRE(n > 0 && c > 0 && sp > 0, "softmax: non-positive dimension (%d, %d, %d)", n, c, sp);
	return PZ_OK;
}

}  // namespace

extern "C" {

int pz_softmax_fwd(const float *x, float *y, int n, int c, int spatial, pz_stream_t stream) {
	if (int rc = sm_check(n, c, spatial)) return rc;
	PZ_REQUIRE(x && y, "pz_softmax_fwd: null tensor");
	softmax_fwd_kernel<<<pz::ceil_div((long)n * spatial, 4), 256, 0, pz::as_stream(stream)>>>(x, y, n, c, spatial);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_softmax_bwd(const float *dy, const float *y, float *dx, int n, int c, int spatial, pz_stream_t stream) {
	if (int rc = sm_check(n, c, spatial)) return rc;
	PZ_REQUIRE(dy && y && dx, "pz_softmax_bwd: null tensor");
	softmax_bwd_kernel<<<pz::ceil_div((long)n * spatial, 4), 256, 0, pz::as_stream(stream)>>>(dy, y, dx, n, c, spatial);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_cross_entropy(const float *scores, const int32_t *labels, const float *weights, int n, int c, int spatial, float *grad,
                     float *error, void *workspace, size_t ws_bytes, pz_stream_t stream) {
	if (int rc = sm_check(n, c, spatial)) return rc;
	PZ_REQUIRE(scores && labels && grad && error, "pz_cross_entropy: null tensor");
	const long npos = (long)n * spatial;
	PZ_REQUIRE(workspace && ws_bytes >= (size_t)npos * sizeof(float), "pz_cross_entropy: workspace needs %ld bytes", npos * 4);

	hipStream_t st = pz::as_stream(stream);
	cross_entropy_kernel<<<pz::ceil_div(npos, 4), 256, 0, st>>>(scores, labels, weights, n, c, spatial, grad, (float *)workspace);
	PZ_LAUNCH_CHECK();
	sum_to_scalar_kernel<<<1, 256, 0, st>>>((const float *)workspace, npos, error);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

}  // extern "C"
