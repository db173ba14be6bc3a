// Spatial batch normalisation for NCHW fp32 — HBM-bound kernels (forward-train 12 B/elem, backward 20 B/elem).
// Replaces DnnContext.batchNormNd / batchNormNdBackward — Hip/Wrappers/MIOpen.py:634-688; formulas pinned by
// Cuda/Wrappers/CuDnnNorm.py:23-77 (biased variance for normalisation, EMA running stats with `factor`).
//
// A channel's data are N slabs of hw contiguous floats (stride c*hw). A workgroup owns (channel, split); T = 16..256
// threads (the power of two covering hw/4) walk one slab with 16-byte accesses, 256/T slabs side by side. The accesses are
// only 4-byte aligned (hw = 55*55 or 7*7 is odd, so slab bases are not 16-B aligned) — gfx950 takes unaligned dwordx4
// — and every thread keeps U of them in flight before touching the data: an HBM-bound stream needs ~10 MB in flight across
// the chip, which one dependent load per thread does not give. The hw % 4 trailing floats of each slab go scalar.
// Statistics are shifted sums (sum(x-K), sum((x-K)^2), K = first element of the channel) reduced per workgroup in
// fp32 and merged across workgroups in fp64 in a fixed order -> deterministic, no atomics.
#include "common.h"

namespace {

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));     // 16-byte access, 4-byte alignment

struct BnGeom {
	int n, c, hw, splits, nt;      // nt: threads cooperating on one slab (16..256)
};

__host__ __device__ inline size_t bn_parts_offset_floats(int c) { return (size_t)c * 4; }      // partials sit behind 2c doubles

// Every workgroup takes its 256 / nt slabs once (splits = ceil(n / (256 / nt))). Blocks are dispatched channel-fastest,
// i.e. in memory order, so the chip sweeps the tensor as one moving window instead of a few thousand independent slab
// streams: tools/probes/stream_bw.hip reaches 6.3 TB/s with a linear sweep where grid-stride patterns stay at 4.6-5.2,
// and the read-one-write-one inference kernel went from 4.2 to 5.6 TB/s on 55x55 maps with this change alone.
inline BnGeom bn_geom(int n, int c, int hw) {
	BnGeom g;
	g.n = n, g.c = c, g.hw = hw;
	const int n4 = hw >> 2;
	g.nt = 16;
	while (g.nt < 256 && g.nt < n4) g.nt <<= 1;
	const int rpp = 256 / g.nt;
	int s = (n + rpp - 1) / rpp;
	if (s > 65535) s = 65535;
	if (s < 1) s = 1;
	g.splits = s;
	return g;
}

inline BnGeom bn_geom_coarse(int n, int c, int hw) {       // >= 8 workgroups per CU across the launch, at most 64 per channel
	BnGeom g = bn_geom(n, c, hw);
	int s = (8 * pz::kNumCU + c - 1) / c;
	if (s > g.splits) s = g.splits;
	if (s > 64) s = 64;
	g.splits = s < 1 ? 1 : s;
	return g;
}

// Visits this workgroup's share of channel `ch`: vec(u, off) is called for U independent 4-float groups at element
// offsets `off` (loads only), then use(u, off) for the same groups (arithmetic / stores), then one(off) for the
// trailing scalars.
template <int U, typename Vec, typename Use, typename One>
__device__ __forceinline__ void channel_foreach(const BnGeom &g, int ch, int split, Vec vec, Use use, One one) {
	const int t = threadIdx.x & (g.nt - 1), ty = threadIdx.x / g.nt, rpp = 256 / g.nt;
	const int n4 = g.hw >> 2, rem = g.hw & 3;
	const int first = split * rpp + ty, stride = g.splits * rpp;
	const size_t slab = (size_t)g.c * g.hw, chan = (size_t)ch * g.hw;

	if (t < n4) {
		int n = first, v = t;
		while (n < g.n) {
			size_t off[U];
			bool ok[U];
#pragma unroll
			for (int u = 0; u < U; ++u) {
				ok[u] = n < g.n;
				off[u] = (size_t)n * slab + chan + 4 * v;
				if (ok[u]) vec(u, off[u]);
				v += g.nt;
				if (v >= n4) v = t, n += stride;
			}
#pragma unroll
			for (int u = 0; u < U; ++u)
				if (ok[u]) use(u, off[u]);
		}
	}
	if (t < rem)
		for (int n = first; n < g.n; n += stride) one((size_t)n * slab + chan + 4 * n4 + t);
}

// Same walk, for loaders that need to know where a vector sits: vec(u, off, n, e) with e = first element's index inside
// the (n, ch) plane; one(off, n, e) likewise. Same thread -> element assignment and order as channel_foreach.
template <int U, typename Vec, typename Use, typename One>
__device__ __forceinline__ void channel_foreach_pos(const BnGeom &g, int ch, int split, Vec vec, Use use, One one) {
	const int t = threadIdx.x & (g.nt - 1), ty = threadIdx.x / g.nt, rpp = 256 / g.nt;
	const int n4 = g.hw >> 2, rem = g.hw & 3;
	const int first = split * rpp + ty, stride = g.splits * rpp;
	const size_t slab = (size_t)g.c * g.hw, chan = (size_t)ch * g.hw;

	if (t < n4) {
		int n = first, v = t;
		while (n < g.n) {
			size_t off[U];
			bool ok[U];
#pragma unroll
			for (int u = 0; u < U; ++u) {
				ok[u] = n < g.n;
				off[u] = (size_t)n * slab + chan + 4 * v;
				if (ok[u]) vec(u, off[u], n, 4 * v);
				v += g.nt;
				if (v >= n4) v = t, n += stride;
			}
#pragma unroll
			for (int u = 0; u < U; ++u)
				if (ok[u]) use(u, off[u]);
		}
	}
	if (t < rem)
		for (int n = first; n < g.n; n += stride) one((size_t)n * slab + chan + 4 * n4 + t, n, 4 * n4 + t);
}

// ---- forward statistics: ws[(ch*S + s)*2 + {0,1}] = {sum(x-K), sum((x-K)^2)}, wsK[ch] = K
__global__ void __launch_bounds__(256) bn_stats_kernel(const float *__restrict__ x, BnGeom g, float *__restrict__ part,
                                                        float *__restrict__ shift) {
	__shared__ float red[16];
	const int ch = blockIdx.x, s = blockIdx.y;
	const float K = x[(size_t)ch * g.hw];
	float s1 = 0.f, s2 = 0.f;
	f4u xv[4];

	channel_foreach<4>(
	    g, ch, s, [&](int u, size_t off) { xv[u] = *reinterpret_cast<const f4u *>(x + off); },
	    [&](int u, size_t) {
		    const float a = xv[u][0] - K, b = xv[u][1] - K, c = xv[u][2] - K, d = xv[u][3] - K;
		    s1 += (a + b) + (c + d);
		    s2 += (a * a + b * b) + (c * c + d * d);
	    },
	    [&](size_t off) {
		    const float a = x[off] - K;
		    s1 += a;
		    s2 += a * a;
	    });

	s1 = block_sum(s1, red);
	s2 = block_sum(s2, red);
	if (threadIdx.x == 0) {
		float *pp = part + bn_parts_offset_floats(g.c);
		pp[((size_t)ch * g.splits + s) * 2 + 0] = s1;
		pp[((size_t)ch * g.splits + s) * 2 + 1] = s2;
		if (s == 0) shift[ch] = K;
	}
}

// Partial sums live in one buffer: merged[ch*2 + {0,1}] doubles first — the per-channel totals, summed in fp64 in a fixed
// order by bn_reduce_parts_kernel right after the kernel that wrote the partials — then part[(ch*splits + s)*2 + {0,1}]
// floats, then c floats (the forward statistics' shift). Consumers read `merged` only, whatever geometry produced the
// partials (the sweep geometry leaves up to 256 per channel: merging them in every consumer workgroup would cost more
// than it saves).
__device__ __forceinline__ void bn_merge(const float *buf, int ch, double &a, double &b) {
	const double *merged = reinterpret_cast<const double *>(buf);
	a = merged[2 * ch], b = merged[2 * ch + 1];
}

// one wave per channel: lane l adds partials l, l+64, ... in order, then a fixed shuffle tree
__global__ void __launch_bounds__(256) bn_reduce_parts_kernel(float *__restrict__ buf, int splits, int c) {
	const int ch = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (ch >= c) return;
	const float *part = buf + bn_parts_offset_floats(c);
	double a = 0.0, b = 0.0;
	for (int i = lane; i < splits; i += 64) {
		a += (double)part[((size_t)ch * splits + i) * 2 + 0];
		b += (double)part[((size_t)ch * splits + i) * 2 + 1];
	}
#pragma unroll
	for (int m = 32; m > 0; m >>= 1) a += __shfl_xor(a, m, 64), b += __shfl_xor(b, m, 64);
	if (lane == 0) {
		double *merged = reinterpret_cast<double *>(buf);
		merged[2 * ch] = a, merged[2 * ch + 1] = b;
	}
}

inline void bn_reduce_parts(float *part, int splits, int c, hipStream_t st) {
	bn_reduce_parts_kernel<<<(c + 3) / 4, 256, 0, st>>>(part, splits, c);
}

// y = a*x + b with a = rstd*scale, b = bias - mean*a, both formed with explicit fma so that the backward kernels of the
// fused BN+ReLU pair re-create bit-identical y (and hence the same ReLU mask) from x
__device__ __forceinline__ void bn_affine(float rstd, float mean, float scale, float bias, float &a, float &b) {
	a = rstd * scale;
	b = __builtin_fmaf(-mean, a, bias);
}

template <bool RELU>
__device__ __forceinline__ float bn_act(float x, float a, float b) {
	const float y = __builtin_fmaf(x, a, b);
	return RELU ? (y > 0.f ? y : 0.f) : y;
}

// Deferred apply: what the BatchNorm forward derives per channel from {mean, var} without touching the tensor — saved
// statistics, running statistics and the affine coefficients coef[ch] = {a, b} of y = a*x + b, which the consumer
// (bn_apply_add_kernel) applies while it reads x anyway. Called from the kernels that produce {mean, var}, so that it
// costs no launch of its own. `coef == nullptr`: nothing to do.
struct BnFinal {
	const float *scale, *bias;
	float *run_mean, *run_var, *save_mean, *save_invvar, *coef;
	float eps, factor;
	double cnt;
};

__device__ __forceinline__ void bn_finalize_one(const BnFinal &f, int ch, float mean, float var_f) {
	const double var = (double)var_f;
	const float rstd = (float)(1.0 / sqrt(var + (double)f.eps));
	f.save_mean[ch] = mean;
	f.save_invvar[ch] = rstd;
	const double unbiased = f.cnt > 1.0 ? var * f.cnt / (f.cnt - 1.0) : var;
	f.run_mean[ch] = (1.f - f.factor) * f.run_mean[ch] + f.factor * mean;
	f.run_var[ch] = (1.f - f.factor) * f.run_var[ch] + f.factor * (float)unbiased;
	float a, b;
	bn_affine(rstd, mean, f.scale[ch], f.bias[ch], a, b);
	f.coef[2 * ch] = a, f.coef[2 * ch + 1] = b;
}

// ---- statistics from the producing convolution's strip sums: stats[ch*strips + strip] = {shift, s1, s2, -} over
// `strip_px` consecutive pixels of the flattened (n, hw) axis. One workgroup per channel folds them, in fp64 and about the
// common reference R = the channel's first shift, into S = sum(v - R) and Q = sum((v - R)^2):
//   sum_strip (v - R)   = s1 + n_s*(shift - R)          sum_strip (v - R)^2 = s2 + 2*(shift - R)*s1 + n_s*(shift - R)^2
// (no divisions in the loop; R is a typical value of the channel, so mean - R is small and Q/N - ((S/N))^2 is benign),
// threads own strips t, t+256, ..., then a fixed-order tree -> pre[ch] = {mean, var}.
// With few channels one workgroup per channel leaves most of the chip idle (64 channels x 50 176 strips after the
// stem): the strips of a channel are then cut into `parts` segments (blockIdx.y) whose {S, Q} go to `partial`, and
// bn_merge_finish_kernel adds them in order.
__global__ void __launch_bounds__(256) bn_merge_strips_kernel(const float4 *__restrict__ stats, int strips, int strip_px, long total_px,
                                                               int c, float *__restrict__ pre, int parts, double *__restrict__ partial,
                                                               BnFinal fin) {
	__shared__ double sh_s[256], sh_q[256];
	const int ch = blockIdx.x, t = threadIdx.x;
	const float4 *mine = stats + (size_t)ch * strips;
	const double R = (double)mine[0].x;
	const int per = (strips + parts - 1) / parts, s_lo = blockIdx.y * per, s_hi = min(s_lo + per, strips);

	double S = 0.0, Q = 0.0;
	for (int s = s_lo + t; s < s_hi; s += 256) {
		const float4 e = mine[s];
		const long left = total_px - (long)s * strip_px;
		// (entries that carry their own count — the Winograd epilogue's tile blocks — say so in .w)
		const double nb = e.w > 0.f ? (double)e.w : (double)(left < strip_px ? left : strip_px), d = (double)e.x - R;
		S += (double)e.y + nb * d;
		Q += (double)e.z + d * (2.0 * (double)e.y + nb * d);
	}

	sh_s[t] = S, sh_q[t] = Q;
	__syncthreads();
	for (int w = 128; w > 0; w >>= 1) {
		if (t < w) sh_s[t] += sh_s[t + w], sh_q[t] += sh_q[t + w];
		__syncthreads();
	}
	if (t == 0) {
		if (parts > 1) {
			partial[((size_t)ch * parts + blockIdx.y) * 2 + 0] = sh_s[0];
			partial[((size_t)ch * parts + blockIdx.y) * 2 + 1] = sh_q[0];
			return;
		}
		const double n = (double)total_px, m = sh_s[0] / n;
		const double var = sh_q[0] / n - m * m;
		const float mean_f = (float)(R + m), var_f = (float)(var > 0.0 ? var : 0.0);
		pre[2 * ch + 0] = mean_f;
		pre[2 * ch + 1] = var_f;
		if (fin.coef) bn_finalize_one(fin, ch, mean_f, var_f);
	}
}

__global__ void __launch_bounds__(256) bn_merge_finish_kernel(const float4 *__restrict__ stats, int strips, long total_px, int c,
                                                               float *__restrict__ pre, int parts, const double *__restrict__ partial,
                                                               BnFinal fin) {
	const int ch = blockIdx.x * blockDim.x + threadIdx.x;
	if (ch >= c) return;
	double S = 0.0, Q = 0.0;
	for (int p = 0; p < parts; ++p) S += partial[((size_t)ch * parts + p) * 2 + 0], Q += partial[((size_t)ch * parts + p) * 2 + 1];
	const double R = (double)stats[(size_t)ch * strips].x, n = (double)total_px, m = S / n;
	const double var = Q / n - m * m;
	const float mean_f = (float)(R + m), var_f = (float)(var > 0.0 ? var : 0.0);
	pre[2 * ch + 0] = mean_f;
	pre[2 * ch + 1] = var_f;
	if (fin.coef) bn_finalize_one(fin, ch, mean_f, var_f);
}

// launches the strip merge; `scratch` (after the 2*c floats of `pre`) must hold c * parts * 2 doubles
inline int bn_merge_parts(int c) {
	int parts = (2 * pz::kNumCU + c - 1) / c;
	return parts < 1 ? 1 : (parts > 16 ? 16 : parts);
}

inline void bn_merge_strips(const float *stats, int strips, long total_px, int c, float *pre, double *scratch, hipStream_t st,
                            const BnFinal &fin = BnFinal{}) {
	int parts = bn_merge_parts(c);
	if (strips < 4 * 256 * parts) parts = 1;          // not worth a second launch
	bn_merge_strips_kernel<<<dim3(c, parts), 256, 0, st>>>(reinterpret_cast<const float4 *>(stats), strips, PZ_CONV_STATS_STRIP,
	                                                       total_px, c, pre, parts, scratch, parts > 1 ? BnFinal{} : fin);
	if (parts > 1)
		bn_merge_finish_kernel<<<(c + 255) / 256, 256, 0, st>>>(reinterpret_cast<const float4 *>(stats), strips, total_px, c, pre,
		                                                        parts, scratch, fin);
}

// out = act( (a1*x1 + b1) + (AFF2 ? a2*x2 + b2 : x2) )   /   out = a1*x1 + b1 when x2 == nullptr
// — the residual sum of ResNet blocks (Modules/Add.py after two BatchNorm branches) with the normalisations applied on
// the fly: bit-identical to materialising both BN outputs first (same fma, same order), 8 B/elem less traffic per BN.
// Sign mask of a fused ReLU's output, for the backward kernels that only need (y > 0): one byte per 4 consecutive elements
// of a (n, ch) plane (bit e = element e of the vector), then one byte per trailing element; `relu_mask_stride` bytes per
// plane. 1/16 of the bytes of reading y back.
__host__ __device__ inline int relu_mask_stride(int hw) { return (hw >> 2) + 3; }

template <bool RELU, bool AFF2>
__global__ void __launch_bounds__(256) bn_apply_add_kernel(const float *__restrict__ x1, const float *__restrict__ coef1,
                                                            const float *__restrict__ x2, const float *__restrict__ coef2,
                                                            float *__restrict__ out, BnGeom g, unsigned char *__restrict__ mask = nullptr) {
	const int ch = blockIdx.x, s = blockIdx.y;
	const float a1 = coef1[2 * ch], b1 = coef1[2 * ch + 1];
	const float a2 = AFF2 ? coef2[2 * ch] : 1.f, b2 = AFF2 ? coef2[2 * ch + 1] : 0.f;
	const bool has2 = x2 != nullptr;

	auto one = [&](float u, float v) {
		const float y1 = __builtin_fmaf(u, a1, b1);
		if (!has2) return RELU ? (y1 > 0.f ? y1 : 0.f) : y1;      // bn_act<RELU>: what bn_apply_train_kernel writes
		const float y2 = AFF2 ? __builtin_fmaf(v, a2, b2) : v;
		const float t = y1 + y2;
		return RELU ? t * (t > 0.f ? 1.f : 0.f) : t;            // reluKer's x * (x > 0)
	};

	f4u xv[4], yv[4];
	size_t mi[4];
	const int ms = relu_mask_stride(g.hw);
	channel_foreach_pos<4>(
	    g, ch, s,
	    [&](int u, size_t off, int n, int e0) {
		    xv[u] = *reinterpret_cast<const f4u *>(x1 + off);
		    yv[u] = has2 ? *reinterpret_cast<const f4u *>(x2 + off) : f4u{0.f, 0.f, 0.f, 0.f};
		    mi[u] = (size_t)(n * g.c + ch) * ms + (e0 >> 2);
	    },
	    [&](int u, size_t off) {
		    const f4u r = f4u{one(xv[u][0], yv[u][0]), one(xv[u][1], yv[u][1]), one(xv[u][2], yv[u][2]), one(xv[u][3], yv[u][3])};
		    *reinterpret_cast<f4u *>(out + off) = r;
		    if (RELU && mask)
			    mask[mi[u]] = (unsigned char)((r[0] > 0.f ? 1 : 0) | (r[1] > 0.f ? 2 : 0) | (r[2] > 0.f ? 4 : 0) | (r[3] > 0.f ? 8 : 0));
	    },
	    [&](size_t off, int n, int e0) {
		    const float r = one(x1[off], has2 ? x2[off] : 0.f);
		    out[off] = r;
		    if (RELU && mask) mask[(size_t)(n * g.c + ch) * ms + (g.hw >> 2) + (e0 & 3)] = r > 0.f ? 1 : 0;
	    });
}

template <bool RELU, bool PRE = false>
__global__ void __launch_bounds__(256) bn_apply_train_kernel(const float *x, float *y, BnGeom g,
                                                              const float *__restrict__ part, const float *__restrict__ shift,
                                                              const float *__restrict__ scale, const float *__restrict__ bias,
                                                              float *__restrict__ run_mean, float *__restrict__ run_var,
                                                              float *__restrict__ save_mean, float *__restrict__ save_invvar,
                                                              float eps, float factor) {
	const int ch = blockIdx.x, s = blockIdx.y;

	const double cnt = (double)g.n * g.hw;
	double var;
	float mean;
	if (PRE) {                  // `part` holds {mean, var} per channel (bn_merge_strips_kernel)
		mean = part[2 * ch], var = (double)part[2 * ch + 1];
	} else {
		double S1, S2;
		bn_merge(part, ch, S1, S2);
		const double m1 = S1 / cnt;
		var = S2 / cnt - m1 * m1;
		var = var > 0.0 ? var : 0.0;
		mean = (float)((double)shift[ch] + m1);
	}
	const float rstd = (float)(1.0 / sqrt(var + (double)eps));

	if (s == 0 && threadIdx.x == 0) {
		save_mean[ch] = mean;
		save_invvar[ch] = rstd;
		const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
		run_mean[ch] = (1.f - factor) * run_mean[ch] + factor * mean;
		run_var[ch] = (1.f - factor) * run_var[ch] + factor * (float)unbiased;
	}

	float a, b;
	bn_affine(rstd, mean, scale[ch], bias[ch], a, b);
	f4u xv[4];

	channel_foreach<4>(
	    g, ch, s, [&](int u, size_t off) { xv[u] = *reinterpret_cast<const f4u *>(x + off); },
	    [&](int u, size_t off) {
		    *reinterpret_cast<f4u *>(y + off) = f4u{bn_act<RELU>(xv[u][0], a, b), bn_act<RELU>(xv[u][1], a, b),
		                                            bn_act<RELU>(xv[u][2], a, b), bn_act<RELU>(xv[u][3], a, b)};
	    },
	    [&](size_t off) { y[off] = bn_act<RELU>(x[off], a, b); });
}

__global__ void __launch_bounds__(256) bn_infer_kernel(const float *x, float *y, BnGeom g,
                                                        const float *__restrict__ scale, const float *__restrict__ bias,
                                                        const float *__restrict__ mean, const float *__restrict__ var, float eps) {
	const int ch = blockIdx.x, s = blockIdx.y;

	// NumpyDnn.batchNorm2d: scale / sqrt(var + eps) * (x - mean) + bias
	const float a = scale[ch] / sqrtf(var[ch] + eps), mu = mean[ch], b = bias[ch];
	f4u xv[4];

	channel_foreach<4>(
	    g, ch, s, [&](int u, size_t off) { xv[u] = *reinterpret_cast<const f4u *>(x + off); },
	    [&](int u, size_t off) {
		    *reinterpret_cast<f4u *>(y + off) =
		        f4u{a * (xv[u][0] - mu) + b, a * (xv[u][1] - mu) + b, a * (xv[u][2] - mu) + b, a * (xv[u][3] - mu) + b};
	    },
	    [&](size_t off) { y[off] = a * (x[off] - mu) + b; });
}

// ---- backward: partials {sum dy, sum dy*(x-mean)} then dx. RELU: the layer's output went through a fused ReLU, so dy
// is first masked with (y > 0), y re-created from x (Cuda/Kernels/ElementWise.py:119-172 reluDer uses the output sign)
template <bool RELU>
__device__ __forceinline__ float bn_gate(float dy, float x, float a, float b) {
	return RELU ? (__builtin_fmaf(x, a, b) > 0.f ? dy : 0.f) : dy;
}

// The two running sums of the backward statistics, written with explicit fma so that every kernel that produces them
// (bn_bwd_stats_kernel, bn_gate_stats_kernel) rounds identically — the fused and unfused paths stay bit-identical.
__device__ __forceinline__ void bn_bwd_acc4(const f4u &q, const f4u &x, float mu, float &s1, float &s2) {
#pragma clang fp contract(off)
	s1 += (q[0] + q[1]) + (q[2] + q[3]);
	const float d0 = x[0] - mu, d1 = x[1] - mu, d2 = x[2] - mu, d3 = x[3] - mu;
	const float p01 = __builtin_fmaf(q[0], d0, q[1] * d1), p23 = __builtin_fmaf(q[2], d2, q[3] * d3);
	s2 += p01 + p23;
}

__device__ __forceinline__ void bn_bwd_acc1(float q, float x, float mu, float &s1, float &s2) {
#pragma clang fp contract(off)
	s1 += q;
	s2 = __builtin_fmaf(q, x - mu, s2);
}

template <bool RELU>
__global__ void __launch_bounds__(256) bn_bwd_stats_kernel(const float *__restrict__ x, const float *__restrict__ dy, BnGeom g,
                                                            const float *__restrict__ save_mean,
                                                            const float *__restrict__ save_invvar, const float *__restrict__ scale,
                                                            const float *__restrict__ bias, float *__restrict__ part,
                                                            const float *__restrict__ gate_coef = nullptr) {
	__shared__ float red[16];
	const int ch = blockIdx.x, s = blockIdx.y;

	const float mu = save_mean[ch];
	float a = 0.f, b = 0.f;
	if (RELU) {          // the forward's own {a, b} when the caller kept them (pz_bn_bwd_gate), else re-derived the same way
		if (gate_coef) a = gate_coef[2 * ch], b = gate_coef[2 * ch + 1];
		else bn_affine(save_invvar[ch], mu, scale[ch], bias[ch], a, b);
	}
	float s1 = 0.f, s2 = 0.f;
	f4u xv[4], gv[4];

	channel_foreach<4>(
	    g, ch, s,
	    [&](int u, size_t off) {
		    xv[u] = *reinterpret_cast<const f4u *>(x + off);
		    gv[u] = *reinterpret_cast<const f4u *>(dy + off);
	    },
	    [&](int u, size_t) {
		    const f4u v = xv[u];
		    f4u q = gv[u];
#pragma unroll
		    for (int e = 0; e < 4; ++e) q[e] = bn_gate<RELU>(q[e], v[e], a, b);
		    bn_bwd_acc4(q, v, mu, s1, s2);
	    },
	    [&](size_t off) { bn_bwd_acc1(bn_gate<RELU>(dy[off], x[off], a, b), x[off], mu, s1, s2); });

	s1 = block_sum(s1, red);
	s2 = block_sum(s2, red);
	if (threadIdx.x == 0) {
		float *pp = part + bn_parts_offset_floats(g.c);
		pp[((size_t)ch * g.splits + s) * 2 + 0] = s1;
		pp[((size_t)ch * g.splits + s) * 2 + 1] = s2;
	}
}

template <bool RELU>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const float *x, const float *dy,
                                                            float *dx, BnGeom g, const float *__restrict__ part,
                                                            const float *__restrict__ scale, const float *__restrict__ bias,
                                                            const float *__restrict__ save_mean,
                                                            const float *__restrict__ save_invvar, float *__restrict__ dscale,
                                                            float *__restrict__ dbias, float *__restrict__ dscale_acc,
                                                            float *__restrict__ dbias_acc, float alpha, float beta,
                                                            const float *__restrict__ gate_coef = nullptr) {
	const int ch = blockIdx.x, s = blockIdx.y;

	double S1, S2;
	bn_merge(part, ch, S1, S2);
	const float mu = save_mean[ch], rstd = save_invvar[ch], sc = scale[ch];
	const float db = (float)S1, ds = (float)(S2 * (double)rstd);      // dscale = rstd * sum dy*(x-mu)

	if (s == 0 && threadIdx.x == 0) {
		dscale[ch] = ds;
		dbias[ch] = db;
		// optional parameter-gradient accumulate (BatchNormND.accGradParams: grad = scale*fresh + momentum*grad) — saves the
		// two 5 us vector kernels per layer
		if (dscale_acc) dscale_acc[ch] = alpha * ds + (beta == 0.f ? 0.f : beta * dscale_acc[ch]);
		if (dbias_acc) dbias_acc[ch] = alpha * db + (beta == 0.f ? 0.f : beta * dbias_acc[ch]);
	}

	// dx = sc*rstd*(dy - db/m - xhat*ds/m),  xhat = (x-mu)*rstd
	const float inv_m = 1.f / ((float)g.n * (float)g.hw);
	const float k0 = sc * rstd, k1 = db * inv_m, k2 = ds * inv_m * rstd;
	float a = 0.f, b = 0.f;
	if (RELU) {
		if (gate_coef) a = gate_coef[2 * ch], b = gate_coef[2 * ch + 1];
		else bn_affine(rstd, mu, sc, bias[ch], a, b);
	}
	f4u xv[4], gv[4];

	channel_foreach<4>(
	    g, ch, s,
	    [&](int u, size_t off) {
		    xv[u] = *reinterpret_cast<const f4u *>(x + off);
		    gv[u] = *reinterpret_cast<const f4u *>(dy + off);
	    },
	    [&](int u, size_t off) {
		    const f4u v = xv[u];
		    f4u q = gv[u];
#pragma unroll
		    for (int e = 0; e < 4; ++e) q[e] = k0 * (bn_gate<RELU>(q[e], v[e], a, b) - k1 - (v[e] - mu) * k2);
		    *reinterpret_cast<f4u *>(dx + off) = q;
	    },
	    [&](size_t off) { dx[off] = k0 * (bn_gate<RELU>(dy[off], x[off], a, b) - k1 - (x[off] - mu) * k2); });
}

// dx = A*dy + (B*x + C) per channel from the coefficient form of the backward (pz_bn_bwd_coef) — the expression the
// convolution kernels evaluate while gathering (conv.hip, BNX), written out for consumers that cannot fold it.
__global__ void __launch_bounds__(256) bn_bwd_apply_coef_kernel(const float *x, const float *dy, float *dx, BnGeom g,
                                                                 const float4 *__restrict__ coef) {
	const int ch = blockIdx.x, s = blockIdx.y;
	const float4 k = coef[ch];
	f4u xv[4], gv[4];
	channel_foreach<4>(
	    g, ch, s,
	    [&](int u, size_t off) {
		    xv[u] = *reinterpret_cast<const f4u *>(x + off);
		    gv[u] = *reinterpret_cast<const f4u *>(dy + off);
	    },
	    [&](int u, size_t off) {
		    f4u q;
#pragma unroll
		    for (int e = 0; e < 4; ++e) q[e] = __builtin_fmaf(k.x, gv[u][e], __builtin_fmaf(k.y, xv[u][e], k.z));
		    *reinterpret_cast<f4u *>(dx + off) = q;
	    },
	    [&](size_t off) { dx[off] = __builtin_fmaf(k.x, dy[off], __builtin_fmaf(k.y, x[off], k.z)); });
}

// Gradient fan-in of a residual block fused with the statistics pass of the batch-norm backward(s) that consume it:
//   g = (g0 + g1) * (y > 0)                       (OpAdd3Gate: Replicate fan-in + reluDer of the block's output ReLU)
//   partA[ch][s] = {sum g, sum g*(xa - mean_a)}    (what bn_bwd_stats_kernel<false> computes for the main branch's last BN)
//   partB likewise for the projection-shortcut BN when the block has one
// One pass writes g and leaves both BNs only their apply pass: same loop structure and accumulation order as
// bn_bwd_stats_kernel, hence bit-identical partial sums.
// UP2: g0 and g1 are the gradients of two stride-2 pointwise convolutions kept compact — (n, c, ceil(h/2), ceil(w/2)),
// the value of pixel (2i, 2j); every other pixel of the full-size gradient is zero (`up` = {w, compact w, compact
// plane size, magic for / w}). The sums see exactly the terms of the dense form (x + 0 = x), so results are bit-identical
// to materialising the zero-filled tensors first; the two 4x larger tensors are neither written nor read.
struct Up2Geom {
	int w, qc, plane_c;
	unsigned magic_w;        // ceil(2^32 / w): e / w == umulhi(e, magic_w) for e < 2^20
	unsigned bytes_c;        // compact tensor bytes (buffer range)
};

__device__ __forceinline__ float up2_fetch(__amdgpu_buffer_rsrc_t r, unsigned plane_off, const Up2Geom &up, int e) {
	const int yy = (int)__umulhi((unsigned)e, up.magic_w), xx = e - yy * up.w;
	const bool even = ((yy | xx) & 1) == 0;
	const unsigned off = even ? (plane_off + (unsigned)((yy >> 1) * up.qc + (xx >> 1))) * 4u : 0xfffffff0u;
	return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

template <bool TWO, bool UP2 = false>
__global__ void __launch_bounds__(256) bn_gate_stats_kernel(const float *__restrict__ g0, const float *__restrict__ g1,
                                                             const float *__restrict__ y, float *__restrict__ gout, BnGeom g,
                                                             const float *__restrict__ xa, const float *__restrict__ mean_a,
                                                             float *__restrict__ part_a, const float *__restrict__ xb,
                                                             const float *__restrict__ mean_b, float *__restrict__ part_b,
                                                             Up2Geom up = Up2Geom{}, const unsigned char *__restrict__ mask = nullptr) {
	__shared__ float red[16];
	const int ch = blockIdx.x, s = blockIdx.y;
	const float mua = mean_a[ch], mub = TWO ? mean_b[ch] : 0.f;
	float a1 = 0.f, a2 = 0.f, b1 = 0.f, b2 = 0.f;
	f4u v0[2], v1[2], vy[2], va[2], vb[2];
	const int ms = relu_mask_stride(g.hw);           // with `mask` the gate is a bit per element and y is not read
	const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void *)g0, 0, UP2 ? up.bytes_c : 0u, 0x00020000);
	const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void *)g1, 0, UP2 ? up.bytes_c : 0u, 0x00020000);

	channel_foreach_pos<2>(
	    g, ch, s,
	    [&](int u, size_t off, int n, int e0) {
		    if constexpr (UP2) {
			    const unsigned plane = (unsigned)(n * g.c + ch) * (unsigned)up.plane_c;
#pragma unroll
			    for (int e = 0; e < 4; ++e) v0[u][e] = up2_fetch(r0, plane, up, e0 + e), v1[u][e] = up2_fetch(r1, plane, up, e0 + e);
		    } else {
			    v0[u] = *reinterpret_cast<const f4u *>(g0 + off);
			    v1[u] = *reinterpret_cast<const f4u *>(g1 + off);
		    }
		    if (mask) {
			    const unsigned m = mask[(size_t)(n * g.c + ch) * ms + (e0 >> 2)];
#pragma unroll
			    for (int e = 0; e < 4; ++e) vy[u][e] = (m >> e) & 1u ? 1.f : 0.f;
		    } else {
			    vy[u] = *reinterpret_cast<const f4u *>(y + off);
		    }
		    va[u] = *reinterpret_cast<const f4u *>(xa + off);
		    if (TWO) vb[u] = *reinterpret_cast<const f4u *>(xb + off);
	    },
	    [&](int u, size_t off) {
		    f4u q;
#pragma unroll
		    for (int e = 0; e < 4; ++e) q[e] = (v0[u][e] + v1[u][e]) * (vy[u][e] > 0.f ? 1.f : 0.f);
		    *reinterpret_cast<f4u *>(gout + off) = q;
		    bn_bwd_acc4(q, va[u], mua, a1, a2);
		    if (TWO) bn_bwd_acc4(q, vb[u], mub, b1, b2);
	    },
	    [&](size_t off, int n, int e0) {
		    float s0, s1;
		    if constexpr (UP2) {
			    const unsigned plane = (unsigned)(n * g.c + ch) * (unsigned)up.plane_c;
			    s0 = up2_fetch(r0, plane, up, e0), s1 = up2_fetch(r1, plane, up, e0);
		    } else {
			    s0 = g0[off], s1 = g1[off];
		    }
		    const bool pos = mask ? mask[(size_t)(n * g.c + ch) * ms + (g.hw >> 2) + (e0 & 3)] != 0 : y[off] > 0.f;
		    const float q = (s0 + s1) * (pos ? 1.f : 0.f);
		    gout[off] = q;
		    bn_bwd_acc1(q, xa[off], mua, a1, a2);
		    if (TWO) bn_bwd_acc1(q, xb[off], mub, b1, b2);
	    });

	a1 = block_sum(a1, red);
	a2 = block_sum(a2, red);
	if (TWO) b1 = block_sum(b1, red), b2 = block_sum(b2, red);
	if (threadIdx.x == 0) {
		float *pa = part_a + bn_parts_offset_floats(g.c);
		pa[((size_t)ch * g.splits + s) * 2 + 0] = a1;
		pa[((size_t)ch * g.splits + s) * 2 + 1] = a2;
		if (TWO) {
			float *pb = part_b + bn_parts_offset_floats(g.c);
			pb[((size_t)ch * g.splits + s) * 2 + 0] = b1;
			pb[((size_t)ch * g.splits + s) * 2 + 1] = b2;
		}
	}
}

// {mean, var} per channel + the strip-merge scratch (c * parts * 2 doubles, 8-byte aligned after 2*c floats)
inline size_t bn_pre_ws_bytes(int c) { return ((size_t)2 * c + 2) * sizeof(float) + (size_t)c * 16 * 2 * sizeof(double); }

inline size_t bn_ws_bytes(const BnGeom &g) {
	const size_t stats = (bn_parts_offset_floats(g.c) + (size_t)g.c * g.splits * 2 + g.c) * sizeof(float);
	const size_t pre = bn_pre_ws_bytes(g.c);
	return stats > pre ? stats : pre;
}

int bn_check(int n, int c, int hw) {
	PZ_REQUIRE(n > 0 && c > 0 && hw > 0, "batchnorm: non-positive dimension (%d, %d, %d)", n, c, hw);
	PZ_REQUIRE(c <= 65535 * 32, "batchnorm: too many channels");
	return PZ_OK;
}

}  // namespace

extern "C" {

int pz_bn_workspace_bytes(int n, int c, int hw, size_t *nbytes) {
	if (int rc = bn_check(n, c, hw)) return rc;
	*nbytes = bn_ws_bytes(bn_geom(n, c, hw));
	return PZ_OK;
}

int pz_bn_fwd_train_act(const float *x, float *y, int n, int c, int hw, const float *scale, const float *bias,
                        float *run_mean, float *run_var, float *save_mean, float *save_invvar, float epsilon, float factor,
                        int act, void *workspace, size_t ws_bytes, pz_stream_t stream) {
	if (int rc = bn_check(n, c, hw)) return rc;
	PZ_REQUIRE(x && y && scale && bias && run_mean && run_var && save_mean && save_invvar, "pz_bn_fwd_train: null tensor");
	PZ_REQUIRE(act == PZ_BN_ACT_NONE || act == PZ_BN_ACT_RELU, "pz_bn_fwd_train: unknown fused activation %d", act);
	const BnGeom g = bn_geom(n, c, hw);
	PZ_REQUIRE(workspace && ws_bytes >= bn_ws_bytes(g), "pz_bn_fwd_train: workspace too small");

	// the forward statistics pass is short (small tensors: the big ones get their statistics from the convolution's
	// epilogue): a coarser geometry than the sweep, fewer and longer workgroups, measured 19 against 28 us
	const BnGeom gc = bn_geom_coarse(n, c, hw);
	float *part = (float *)workspace, *shift = part + bn_parts_offset_floats(c) + (size_t)c * g.splits * 2;
	hipStream_t st = pz::as_stream(stream);
	dim3 grid(c, g.splits);

	bn_stats_kernel<<<dim3(c, gc.splits), 256, 0, st>>>(x, gc, part, shift);
	PZ_LAUNCH_CHECK();
	bn_reduce_parts(part, gc.splits, c, st);
	PZ_LAUNCH_CHECK();
	if (act == PZ_BN_ACT_RELU)
		bn_apply_train_kernel<true><<<grid, 256, 0, st>>>(x, y, g, part, shift, scale, bias, run_mean, run_var, save_mean,
		                                                  save_invvar, epsilon, factor);
	else
		bn_apply_train_kernel<false><<<grid, 256, 0, st>>>(x, y, g, part, shift, scale, bias, run_mean, run_var, save_mean,
		                                                   save_invvar, epsilon, factor);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_bn_fwd_train_pre(const float *x, float *y, int n, int c, int hw, const float *scale, const float *bias,
                        float *run_mean, float *run_var, float *save_mean, float *save_invvar, float epsilon, float factor,
                        int act, const float *stats, int strips, void *workspace, size_t ws_bytes, pz_stream_t stream) {
	if (int rc = bn_check(n, c, hw)) return rc;
	PZ_REQUIRE(x && y && scale && bias && run_mean && run_var && save_mean && save_invvar && stats, "pz_bn_fwd_train_pre: null tensor");
	PZ_REQUIRE(act == PZ_BN_ACT_NONE || act == PZ_BN_ACT_RELU, "pz_bn_fwd_train_pre: unknown fused activation %d", act);
	const long total_px = (long)n * hw;
	// (strips of PZ_CONV_STATS_STRIP pixels from the implicit GEMM, or Winograd tile blocks that carry their own counts)
	PZ_REQUIRE(strips >= 1, "pz_bn_fwd_train_pre: no statistics entries");
	const BnGeom g = bn_geom(n, c, hw);
	PZ_REQUIRE(workspace && ws_bytes >= bn_ws_bytes(g), "pz_bn_fwd_train_pre: workspace too small");

	float *pre = (float *)workspace;             // 2*c floats, then the merge scratch
	PZ_REQUIRE(ws_bytes >= bn_pre_ws_bytes(c), "pz_bn_fwd_train_pre: workspace too small for the strip merge");
	hipStream_t st = pz::as_stream(stream);
	bn_merge_strips(stats, strips, total_px, c, pre, reinterpret_cast<double *>(pre + 2 * c + (2 * c) % 2), st);
	PZ_LAUNCH_CHECK();

	const BnGeom gs = g;
	dim3 grid(c, gs.splits);
	if (act == PZ_BN_ACT_RELU)
		bn_apply_train_kernel<true, true><<<grid, 256, 0, st>>>(x, y, gs, pre, nullptr, scale, bias, run_mean, run_var, save_mean,
		                                                        save_invvar, epsilon, factor);
	else
		bn_apply_train_kernel<false, true><<<grid, 256, 0, st>>>(x, y, gs, pre, nullptr, scale, bias, run_mean, run_var, save_mean,
		                                                         save_invvar, epsilon, factor);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_bn_fwd_train_defer(int n, int c, int hw, const float *scale, const float *bias, float *run_mean, float *run_var,
                          float *save_mean, float *save_invvar, float epsilon, float factor, const float *stats, int strips,
                          float *coef, void *workspace, size_t ws_bytes, pz_stream_t stream) {
	if (int rc = bn_check(n, c, hw)) return rc;
	PZ_REQUIRE(scale && bias && run_mean && run_var && save_mean && save_invvar && stats && coef, "pz_bn_fwd_train_defer: null tensor");
	const long total_px = (long)n * hw;
	// (strips of PZ_CONV_STATS_STRIP pixels from the implicit GEMM, or Winograd tile blocks that carry their own counts)
	PZ_REQUIRE(strips >= 1, "pz_bn_fwd_train_defer: no statistics entries");
	PZ_REQUIRE(workspace && ws_bytes >= bn_pre_ws_bytes(c), "pz_bn_fwd_train_defer: workspace too small");

	float *pre = (float *)workspace;
	hipStream_t st = pz::as_stream(stream);
	// {mean, var} -> saved / running statistics and coefficients in the merge's last kernel (no launch of its own)
	const BnFinal fin{scale, bias, run_mean, run_var, save_mean, save_invvar, coef, epsilon, factor, (double)total_px};
	bn_merge_strips(stats, strips, total_px, c, pre, reinterpret_cast<double *>(pre + 2 * c + (2 * c) % 2), st, fin);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

// Statistics pass's partial sums of one channel per wave -> merged totals (bn_reduce_parts_kernel's arithmetic) -> {mean,
// var} about the channel's shift -> saved / running statistics and coefficients (bn_finalize_one), all in one launch
__global__ void __launch_bounds__(256) bn_reduce_finalize_kernel(float *__restrict__ buf, int splits, int c, const float *__restrict__ shift,
                                                                  BnFinal fin) {
	const int ch = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (ch >= c) return;
	const float *part = buf + bn_parts_offset_floats(c);
	double a = 0.0, b = 0.0;
	for (int i = lane; i < splits; i += 64) {
		a += (double)part[((size_t)ch * splits + i) * 2 + 0];
		b += (double)part[((size_t)ch * splits + i) * 2 + 1];
	}
#pragma unroll
	for (int m = 32; m > 0; m >>= 1) a += __shfl_xor(a, m, 64), b += __shfl_xor(b, m, 64);
	if (lane == 0) {
		double *merged = reinterpret_cast<double *>(buf);
		merged[2 * ch] = a, merged[2 * ch + 1] = b;
		const double m1 = a / fin.cnt;
		double var = b / fin.cnt - m1 * m1;
		var = var > 0.0 ? var : 0.0;
		const float mean_f = (float)((double)shift[ch] + m1), var_f = (float)var;
		bn_finalize_one(fin, ch, mean_f, var_f);
	}
}

// Training-mode forward without the normalisation pass: statistics (from the producing convolution's strip sums when
// `stats` is given, else from a pass over x), saved / running statistics, and coef[ch] = {a, b} of y = a*x + b. Whoever
// reads the normalised tensor applies the pair on the fly (pz_bn_apply_add, pz_conv2d_fwd_bn); the lazy-buffer layer of
// the backend (puzzlelib_amd/lazy.py) decides whether the tensor is ever written.
int pz_bn_fwd_train_coef(const float *x, int n, int c, int hw, const float *scale, const float *bias, float *run_mean,
                         float *run_var, float *save_mean, float *save_invvar, float epsilon, float factor, const float *stats,
                         int strips, float *coef, void *workspace, size_t ws_bytes, pz_stream_t stream) {
	if (stats)
		return pz_bn_fwd_train_defer(n, c, hw, scale, bias, run_mean, run_var, save_mean, save_invvar, epsilon, factor, stats,
		                             strips, coef, workspace, ws_bytes, stream);
	if (int rc = bn_check(n, c, hw)) return rc;
	PZ_REQUIRE(x && scale && bias && run_mean && run_var && save_mean && save_invvar && coef, "pz_bn_fwd_train_coef: null tensor");
	const BnGeom g = bn_geom(n, c, hw);
	PZ_REQUIRE(workspace && ws_bytes >= bn_ws_bytes(g) && ws_bytes >= bn_pre_ws_bytes(c), "pz_bn_fwd_train_coef: workspace too small");

	const BnGeom gc = bn_geom_coarse(n, c, hw);
	float *part = (float *)workspace, *shift = part + bn_parts_offset_floats(c) + (size_t)c * g.splits * 2;
	hipStream_t st = pz::as_stream(stream);
	bn_stats_kernel<<<dim3(c, gc.splits), 256, 0, st>>>(x, gc, part, shift);
	PZ_LAUNCH_CHECK();
	// partial sums -> merged totals -> {mean, var} -> saved / running statistics and coefficients: one launch, one wave per channel
	const BnFinal fin{scale, bias, run_mean, run_var, save_mean, save_invvar, coef, epsilon, factor, (double)n * hw};
	bn_reduce_finalize_kernel<<<(c + 3) / 4, 256, 0, st>>>(part, gc.splits, c, shift, fin);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_relu_mask_bytes(int n, int c, int hw, size_t *nbytes) {
	if (int rc = bn_check(n, c, hw)) return rc;
	PZ_REQUIRE(nbytes != nullptr, "pz_relu_mask_bytes: null output");
	*nbytes = (size_t)n * c * relu_mask_stride(hw);
	return PZ_OK;
}

int pz_bn_apply_add(const float *x1, const float *coef1, const float *x2, const float *coef2, float *out, int n, int c, int hw,
                    int relu, pz_stream_t stream) {
	return pz_bn_apply_add_mask(x1, coef1, x2, coef2, out, nullptr, n, c, hw, relu, stream);
}

int pz_bn_apply_add_mask(const float *x1, const float *coef1, const float *x2, const float *coef2, float *out, unsigned char *mask,
                         int n, int c, int hw, int relu, pz_stream_t stream) {
	if (int rc = bn_check(n, c, hw)) return rc;
	PZ_REQUIRE(x1 && coef1 && out, "pz_bn_apply_add: null tensor");
	PZ_REQUIRE(x2 || !coef2, "pz_bn_apply_add: coefficients of a second operand without the operand");
	PZ_REQUIRE(mask == nullptr || relu, "pz_bn_apply_add_mask: the sign mask belongs to the fused ReLU");
	const BnGeom g = bn_geom(n, c, hw);
	const dim3 grid(c, g.splits);
	hipStream_t st = pz::as_stream(stream);

	if (relu && coef2) bn_apply_add_kernel<true, true><<<grid, 256, 0, st>>>(x1, coef1, x2, coef2, out, g, mask);
	else if (relu) bn_apply_add_kernel<true, false><<<grid, 256, 0, st>>>(x1, coef1, x2, coef2, out, g, mask);
	else if (coef2) bn_apply_add_kernel<false, true><<<grid, 256, 0, st>>>(x1, coef1, x2, coef2, out, g);
	else bn_apply_add_kernel<false, false><<<grid, 256, 0, st>>>(x1, coef1, x2, coef2, out, g);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_bn_fwd_train(const float *x, float *y, int n, int c, int hw, const float *scale, const float *bias, float *run_mean,
                    float *run_var, float *save_mean, float *save_invvar, float epsilon, float factor, void *workspace,
                    size_t ws_bytes, pz_stream_t stream) {
	return pz_bn_fwd_train_act(x, y, n, c, hw, scale, bias, run_mean, run_var, save_mean, save_invvar, epsilon, factor,
	                           PZ_BN_ACT_NONE, workspace, ws_bytes, stream);
}

int pz_bn_fwd_infer(const float *x, float *y, int n, int c, int hw, const float *scale, const float *bias, const float *mean,
                    const float *var, float epsilon, pz_stream_t stream) {
	if (int rc = bn_check(n, c, hw)) return rc;
	PZ_REQUIRE(x && y && scale && bias && mean && var, "pz_bn_fwd_infer: null tensor");
	const BnGeom g = bn_geom(n, c, hw);
	bn_infer_kernel<<<dim3(c, g.splits), 256, 0, pz::as_stream(stream)>>>(x, y, g, scale, bias, mean, var, epsilon);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_bn_bwd_act(const float *x, const float *dy, float *dx, int n, int c, int hw, const float *scale, const float *bias,
                  const float *save_mean, const float *save_invvar, float *dscale, float *dbias, int act, void *workspace,
                  size_t ws_bytes, pz_stream_t stream) {
	return pz_bn_bwd_acc(x, dy, dx, n, c, hw, scale, bias, save_mean, save_invvar, dscale, dbias, act, nullptr, nullptr, 1.f, 0.f,
	                     workspace, ws_bytes, stream);
}

int pz_bn_bwd_acc(const float *x, const float *dy, float *dx, int n, int c, int hw, const float *scale, const float *bias,
                  const float *save_mean, const float *save_invvar, float *dscale, float *dbias, int act, float *dscale_acc,
                  float *dbias_acc, float alpha, float beta, void *workspace, size_t ws_bytes, pz_stream_t stream) {
	if (int rc = bn_check(n, c, hw)) return rc;
	PZ_REQUIRE(x && dy && dx && scale && save_mean && save_invvar && dscale && dbias, "pz_bn_bwd: null tensor");
	PZ_REQUIRE(act == PZ_BN_ACT_NONE || (act == PZ_BN_ACT_RELU && bias), "pz_bn_bwd: fused activation %d needs the bias", act);
	const BnGeom g = bn_geom(n, c, hw);
	PZ_REQUIRE(workspace && ws_bytes >= bn_ws_bytes(g), "pz_bn_bwd: workspace too small");

	float *part = (float *)workspace;
	hipStream_t st = pz::as_stream(stream);
	dim3 grid(c, g.splits);

	if (act == PZ_BN_ACT_RELU) {
		bn_bwd_stats_kernel<true><<<grid, 256, 0, st>>>(x, dy, g, save_mean, save_invvar, scale, bias, part);
		PZ_LAUNCH_CHECK();
		bn_reduce_parts(part, g.splits, c, st);
		bn_bwd_apply_kernel<true><<<grid, 256, 0, st>>>(x, dy, dx, g, part, scale, bias, save_mean, save_invvar, dscale, dbias,
		                                                dscale_acc, dbias_acc, alpha, beta);
	} else {
		bn_bwd_stats_kernel<false><<<grid, 256, 0, st>>>(x, dy, g, save_mean, save_invvar, scale, bias, part);
		PZ_LAUNCH_CHECK();
		bn_reduce_parts(part, g.splits, c, st);
		bn_bwd_apply_kernel<false><<<grid, 256, 0, st>>>(x, dy, dx, g, part, scale, bias, save_mean, save_invvar, dscale, dbias,
		                                                 dscale_acc, dbias_acc, alpha, beta);
	}
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

// Backward of a BatchNorm whose output went through a ReLU, the gate (y > 0) re-created from x with the forward's own
// coefficient pairs `gate_coef` = {a, b} per channel (pz_bn_fwd_train_coef): dy is masked on load in both passes, the
// ReLU's output is not read and its derivative costs no pass of its own.
int pz_bn_bwd_gate(const float *x, const float *dy, float *dx, int n, int c, int hw, const float *scale, const float *save_mean,
                   const float *save_invvar, float *dscale, float *dbias, const float *gate_coef, void *workspace,
                   size_t ws_bytes, pz_stream_t stream) {
	if (int rc = bn_check(n, c, hw)) return rc;
	PZ_REQUIRE(x && dy && dx && scale && save_mean && save_invvar && dscale && dbias && gate_coef, "pz_bn_bwd_gate: null tensor");
	const BnGeom g = bn_geom(n, c, hw);
	PZ_REQUIRE(workspace && ws_bytes >= bn_ws_bytes(g), "pz_bn_bwd_gate: workspace too small");
	float *part = (float *)workspace;
	hipStream_t st = pz::as_stream(stream);
	dim3 grid(c, g.splits);
	bn_bwd_stats_kernel<true><<<grid, 256, 0, st>>>(x, dy, g, save_mean, save_invvar, scale, nullptr, part, gate_coef);
	PZ_LAUNCH_CHECK();
	bn_reduce_parts(part, g.splits, c, st);
	bn_bwd_apply_kernel<true><<<grid, 256, 0, st>>>(x, dy, dx, g, part, scale, nullptr, save_mean, save_invvar, dscale, dbias,
	                                                nullptr, nullptr, 1.f, 0.f, gate_coef);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

// Statistics pass of the backward alone: partial sums {sum dy, sum dy*(x - mean)} in the layout pz_bn_bwd_coef /
// pz_bn_bwd_from_partials read (what pz_bn_gate_stats leaves when the gradient comes out of a fan-in).
int pz_bn_bwd_stats(const float *x, const float *dy, int n, int c, int hw, const float *save_mean, float *partials,
                    pz_stream_t stream) {
	if (int rc = bn_check(n, c, hw)) return rc;
	PZ_REQUIRE(x && dy && save_mean && partials, "pz_bn_bwd_stats: null tensor");
	const BnGeom g = bn_geom(n, c, hw);
	hipStream_t st = pz::as_stream(stream);
	bn_bwd_stats_kernel<false><<<dim3(c, g.splits), 256, 0, st>>>(x, dy, g, save_mean, nullptr, nullptr, nullptr, partials);
	PZ_LAUNCH_CHECK();
	bn_reduce_parts(partials, g.splits, c, st);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

// Backward coefficients: dx = A*dy + B*x + C per channel (A = gamma*rstd, B = -A*k2, C = A*(mean*k2 - k1) with
// k1 = dbias/m, k2 = dgamma*rstd/m) plus the parameter gradients, from the partial sums — for consumers that apply the
// BatchNorm backward while they gather dy (pz_conv2d_bwd_data_bn / pz_conv2d_bwd_filter_bn): the 12 B/elem apply pass
// over the tensor disappears.
__global__ void __launch_bounds__(256) bn_bwd_coef_kernel(const float *__restrict__ part, int splits, int c, float inv_m,
                                                           const float *__restrict__ scale, const float *__restrict__ save_mean,
                                                           const float *__restrict__ save_invvar, float *__restrict__ dscale,
                                                           float *__restrict__ dbias, float *__restrict__ dscale_acc,
                                                           float *__restrict__ dbias_acc, float alpha, float beta,
                                                           float4 *__restrict__ coef) {
	const int ch = blockIdx.x * blockDim.x + threadIdx.x;
	if (ch >= c) return;
	double S1, S2;
	bn_merge(part, ch, S1, S2);
	const float mu = save_mean[ch], rstd = save_invvar[ch], sc = scale[ch];
	const float db = (float)S1, ds = (float)(S2 * (double)rstd);
	dscale[ch] = ds;
	dbias[ch] = db;
	if (dscale_acc) dscale_acc[ch] = alpha * ds + (beta == 0.f ? 0.f : beta * dscale_acc[ch]);
	if (dbias_acc) dbias_acc[ch] = alpha * db + (beta == 0.f ? 0.f : beta * dbias_acc[ch]);

	const float k0 = sc * rstd, k1 = db * inv_m, k2 = ds * inv_m * rstd;
	coef[ch] = make_float4(k0, -k0 * k2, k0 * (mu * k2 - k1), 0.f);
}

int pz_bn_bwd_coef(int n, int c, int hw, const float *scale, const float *save_mean, const float *save_invvar, float *dscale,
                   float *dbias, float *dscale_acc, float *dbias_acc, float alpha, float beta, const float *partials,
                   float *coef, pz_stream_t stream) {
	if (int rc = bn_check(n, c, hw)) return rc;
	PZ_REQUIRE(scale && save_mean && save_invvar && dscale && dbias && partials && coef, "pz_bn_bwd_coef: null tensor");
	const BnGeom g = bn_geom(n, c, hw);
	bn_bwd_coef_kernel<<<(c + 255) / 256, 256, 0, pz::as_stream(stream)>>>(
	    partials, g.splits, c, 1.f / ((float)n * (float)hw), scale, save_mean, save_invvar, dscale, dbias, dscale_acc, dbias_acc,
	    alpha, beta, reinterpret_cast<float4 *>(coef));
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_bn_bwd_apply_coef(const float *x, const float *dy, float *dx, int n, int c, int hw, const float *coef, pz_stream_t stream) {
	if (int rc = bn_check(n, c, hw)) return rc;
	PZ_REQUIRE(x && dy && dx && coef, "pz_bn_bwd_apply_coef: null tensor");
	const BnGeom g = bn_geom(n, c, hw);
	bn_bwd_apply_coef_kernel<<<dim3(c, g.splits), 256, 0, pz::as_stream(stream)>>>(x, dy, dx, g, reinterpret_cast<const float4 *>(coef));
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_bn_gate_stats(const float *g0, const float *g1, const float *y, const unsigned char *mask, float *gout, int n, int c, int hw,
                     const float *xa, const float *mean_a, float *part_a, const float *xb, const float *mean_b, float *part_b,
                     pz_stream_t stream) {
	if (int rc = bn_check(n, c, hw)) return rc;
	PZ_REQUIRE(g0 && g1 && (y || mask) && gout && xa && mean_a && part_a, "pz_bn_gate_stats: null tensor");
	PZ_REQUIRE((xb == nullptr) == (mean_b == nullptr) && (xb == nullptr) == (part_b == nullptr),
	           "pz_bn_gate_stats: the second batch-norm needs all of x, mean and partials");
	const BnGeom g = bn_geom(n, c, hw);
	const dim3 grid(c, g.splits);
	hipStream_t st = pz::as_stream(stream);
	if (xb) bn_gate_stats_kernel<true><<<grid, 256, 0, st>>>(g0, g1, y, gout, g, xa, mean_a, part_a, xb, mean_b, part_b, Up2Geom{}, mask);
	else bn_gate_stats_kernel<false><<<grid, 256, 0, st>>>(g0, g1, y, gout, g, xa, mean_a, part_a, nullptr, nullptr, nullptr, Up2Geom{}, mask);
	PZ_LAUNCH_CHECK();
	bn_reduce_parts(part_a, g.splits, c, st);
	if (xb) bn_reduce_parts(part_b, g.splits, c, st);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_bn_gate_stats_up2(const float *g0c, const float *g1c, const float *y, const unsigned char *mask, float *gout, int n, int c,
                         int h, int w, const float *xa, const float *mean_a, float *part_a, const float *xb, const float *mean_b,
                         float *part_b, pz_stream_t stream) {
	if (int rc = bn_check(n, c, h * w)) return rc;
	PZ_REQUIRE(g0c && g1c && (y || mask) && gout && xa && mean_a && part_a, "pz_bn_gate_stats_up2: null tensor");
	PZ_REQUIRE((xb == nullptr) == (mean_b == nullptr) && (xb == nullptr) == (part_b == nullptr),
	           "pz_bn_gate_stats_up2: the second batch-norm needs all of x, mean and partials");
	const int pc = (h + 1) / 2, qc = (w + 1) / 2;
	const size_t bytes_c = (size_t)n * c * pc * qc * sizeof(float);
	PZ_REQUIRE(h * w < (1 << 20) && bytes_c < 0xfffffff0u, "pz_bn_gate_stats_up2: map or tensor too large");
	const BnGeom g = bn_geom(n, c, h * w);
	Up2Geom up{w, qc, pc * qc, (unsigned)((((unsigned long long)1 << 32) + w - 1) / w), (unsigned)bytes_c};
	const dim3 grid(c, g.splits);
	hipStream_t st = pz::as_stream(stream);
	if (xb) bn_gate_stats_kernel<true, true><<<grid, 256, 0, st>>>(g0c, g1c, y, gout, g, xa, mean_a, part_a, xb, mean_b, part_b, up, mask);
	else bn_gate_stats_kernel<false, true><<<grid, 256, 0, st>>>(g0c, g1c, y, gout, g, xa, mean_a, part_a, nullptr, nullptr, nullptr, up, mask);
	PZ_LAUNCH_CHECK();
	bn_reduce_parts(part_a, g.splits, c, st);
	if (xb) bn_reduce_parts(part_b, g.splits, c, st);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_bn_bwd_from_partials(const float *x, const float *dy, float *dx, int n, int c, int hw, const float *scale,
                            const float *save_mean, const float *save_invvar, float *dscale, float *dbias,
                            float *dscale_acc, float *dbias_acc, float alpha, float beta, const float *partials,
                            pz_stream_t stream) {
	if (int rc = bn_check(n, c, hw)) return rc;
	PZ_REQUIRE(x && dy && dx && scale && save_mean && save_invvar && dscale && dbias && partials, "pz_bn_bwd_from_partials: null tensor");
	const BnGeom g = bn_geom(n, c, hw);
	bn_bwd_apply_kernel<false><<<dim3(c, g.splits), 256, 0, pz::as_stream(stream)>>>(
	    x, dy, dx, g, partials, scale, nullptr, save_mean, save_invvar, dscale, dbias, dscale_acc, dbias_acc, alpha, beta);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

// pz_bn_bwd_gate when the statistics pass is already done: `partials` carry {sum q, sum q*(x - mean)} of the GATED gradient
// q = dy * (relu(a x + b) > 0) — left by the backward-data launch that produced dy (pz_conv2d_bwd_data_bnstats). One pass.
int pz_bn_bwd_gate_from_partials(const float *x, const float *dy, float *dx, int n, int c, int hw, const float *scale,
                                 const float *save_mean, const float *save_invvar, float *dscale, float *dbias,
                                 const float *gate_coef, const float *partials, pz_stream_t stream) {
	if (int rc = bn_check(n, c, hw)) return rc;
	PZ_REQUIRE(x && dy && dx && scale && save_mean && save_invvar && dscale && dbias && gate_coef && partials,
	           "pz_bn_bwd_gate_from_partials: null tensor");
	const BnGeom g = bn_geom(n, c, hw);
	bn_bwd_apply_kernel<true><<<dim3(c, g.splits), 256, 0, pz::as_stream(stream)>>>(
	    x, dy, dx, g, partials, scale, nullptr, save_mean, save_invvar, dscale, dbias, nullptr, nullptr, 1.f, 0.f, gate_coef);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_bn_bwd(const float *x, const float *dy, float *dx, int n, int c, int hw, const float *scale, const float *save_mean,
              const float *save_invvar, float *dscale, float *dbias, void *workspace, size_t ws_bytes, pz_stream_t stream) {
	return pz_bn_bwd_act(x, dy, dx, n, c, hw, scale, nullptr, save_mean, save_invvar, dscale, dbias, PZ_BN_ACT_NONE, workspace,
	                     ws_bytes, stream);
}

}  // extern "C"
