// Shared host/device helpers for libpuzzle_mi355 (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include <cstdint>
#include <cstring>

#include "../../include/puzzle_mi355.h"

namespace pz {

void set_error(const char *fmt, ...);

inline hipStream_t as_stream(pz_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kWave = 64;        // CDNA wavefront
constexpr int kNumXCD = 8;       // MI355X: 8 XCDs, block b is observed to land on XCD b % 8
constexpr int kNumCU = 256;

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// grid for bandwidth kernels: one workgroup per `per_block` items, dispatched in address order — the chip then sweeps
// the operands as one moving window (6.3 TB/s in tools/probes/stream_bw.hip; the same accesses as a grid-stride loop of
// a few thousand workgroups reach 4.6-5.2). The kernels keep their grid-stride loops for the (huge) remainder beyond the cap.
inline int stream_grid(size_t work_items, int per_block) {
	size_t blocks = (work_items + per_block - 1) / per_block;
	size_t cap = (size_t)1 << 22;
	return (int)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

// pz_conv2d_bwd_data_bnstats: the BatchNorm in FRONT of the layer whose backward-data launch is asked to sum that BatchNorm's backward
// statistics over the gated gradient in its epilogue — y = relu(gab[c].x * gx + gab[c].y) was the layer's input; per produced channel
// and strip (64 pixels in the implicit GEMM, 32 tiles in the Winograd kernel) {sum q, sum q (gx - gmean[c])}, q = dx * (y > 0), to gst
struct BnStatsOut {
	const float *gx, *gab, *gmean;
	float *gst;          // float2 [channel][strips]
};

// Winograd F(2x2, 3x3) forward / backward-data (wino.hip); `which` is PZ_CONV_FWD or PZ_CONV_BWD_DATA
bool wino_eligible(const pz_conv_desc *d, int which, int P, int Q);
size_t wino_workspace_bytes(const pz_conv_desc *d, int which, int P, int Q);
int wino_conv(const pz_conv_desc *d, int which, int P, int Q, const float *in, const float *w, const float *bias, float *out,
              void *workspace, hipStream_t st, float *stats = nullptr, bool filters_ready = false, void *vscratch = nullptr,
              const BnStatsOut *bst = nullptr);
int wino_bnstats_strips(const pz_conv_desc *d, int P, int Q);     // strips per channel a backward-data launch leaves for `bst` (0: it cannot)
// scratch for the transformed input of a forward / backward-data launch (0: the launch transforms its patches itself); passed
// as `vscratch`
size_t wino_input_bytes(const pz_conv_desc *d, int which, int P, int Q);
// transformed filters of up to kWinoBatch (layer, pass) pairs in one launch; u[i] = what wino_conv expects at `workspace`
constexpr int kWinoBatch = 40;
int wino_filter_batch(const pz_conv_desc *const *descs, const int *which, const float *const *w, float *const *u, int n, hipStream_t st);
int wino_stats_strips(const pz_conv_desc *d, int P, int Q);      // statistics blocks per channel of a forward launch (0: none)
bool wino_wgrad_eligible(const pz_conv_desc *d, int P, int Q);
size_t wino_wgrad_workspace_bytes(const pz_conv_desc *d, int P, int Q);
int wino_wgrad(const pz_conv_desc *d, int P, int Q, const float *x, const float *dy, float *dw, float alpha, float beta,
               void *workspace, hipStream_t st);

// Winograd F(4x4, 3x3) forward / backward-data (wino4.hip), taken by the wino_* entry points above where wino4_pick says so
bool wino4_pick(const pz_conv_desc *d, int which, int P, int Q);
size_t wino4_workspace_bytes(const pz_conv_desc *d, int which);
int wino4_stats_strips(const pz_conv_desc *d, int P, int Q);
int wino4_filter_batch(const pz_conv_desc *const *descs, const int *which, const float *const *w, float *const *u, int n, hipStream_t st);
int wino4_conv(const pz_conv_desc *d, int which, int P, int Q, const float *in, const float *w, const float *bias, float *out,
               void *workspace, hipStream_t st, float *stats, bool filters_ready, void *vscratch, const BnStatsOut *bst = nullptr);
size_t wino4_input_bytes(const pz_conv_desc *d, int which, int P, int Q);

// direct backward-data for stride-2 convolutions with <= 4 input maps (thin.hip): the stem layer
bool thin_dgrad_eligible(const pz_conv_desc *d, int P, int Q);
size_t thin_dgrad_workspace_bytes(const pz_conv_desc *d);
int thin_dgrad(const pz_conv_desc *d, int P, int Q, const float *dy, const float *w, float *dx, void *workspace, hipStream_t st);

}  // namespace pz

#define PZ_REQUIRE(cond, ...)                                   \
	do {                                                        \
		if (!(cond)) {                                          \
			pz::set_error(__VA_ARGS__);                         \
			return PZ_ERR_INVALID;                              \
		}                                                       \
	} while (0)

#define PZ_HIP(call)                                                                         \
	do {                                                                                     \
		hipError_t e_ = (call);                                                              \
		if (e_ != hipSuccess) {                                                              \
			pz::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
			return e_ == hipErrorOutOfMemory ? PZ_ERR_NOMEM : PZ_ERR_HIP;                    \
		}                                                                                    \
	} while (0)

#define PZ_LAUNCH_CHECK()                                                                    \
	do {                                                                                     \
		hipError_t e_ = hipGetLastError();                                                   \
		if (e_ != hipSuccess) {                                                              \
			pz::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
			return PZ_ERR_HIP;                                                               \
		}                                                                                    \
	} while (0)

// ---- device-side wavefront helpers (64 lanes) ----------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
	for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
	return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
	for (int m = 32; m > 0; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
	return v;
}

// block-wide sum for blocks of up to 1024 threads; every thread gets the result. `smem` >= 16 floats.
__device__ __forceinline__ float block_sum(float v, float *smem) {
	v = wave_sum(v);
	const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
	__syncthreads();
	if (lane == 0) smem[wid] = v;
	__syncthreads();
	float r = 0.f;
	for (int i = 0; i < nw; ++i) r += smem[i];
	return r;
}
