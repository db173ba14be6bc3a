// The remaining run-time compiled kernels of the reference's Cuda/Kernels/ directory that other PuzzleLib modules reach
// through the same backend object (Unittester.py:114-122 lists their tests for the Hip backend):
//   point-wise cost kernels — Cuda/Kernels/Costs.py:8-72 (bceKer, hingeKer, smoothL1Ker, l1HingeKer): gradient per element
//                   plus one error term per element; the terms are summed afterwards in a fixed order (the reference
//                   atomicAdds them into the 0-d error array, so its last bits depend on the schedule) and ADDED to the
//                   error scalar, as the atomics would
//   PReLU           — Cuda/Kernels/PRelu.py:14-133: y = x * (x > 0 ? 1 : slope[c / div]); dx = dy * ((x > 0) + (x <= 0) * slope);
//                   dslope[c] = sum_{n, pixels} dy * x * (x <= 0)
//   reflection pad  — Cuda/Kernels/Pad.py:45-230: out[o] = in[reflect(o - lpad)] (edge element not repeated); the backward
//                   pass is a GATHER over the at most 3 (1-d) / 9 (2-d) output positions that map to an input element —
//                   deterministic, where the reference scatters with atomicAdd
//   up-sampling     — Cuda/Kernels/Upsample.py:8-298: nearest (integer scale, block copy / block sum) and linear
//                   ("align corners": r = (in - 1) / (out - 1), taps floor(r * o) and its neighbour) in 2-d and 3-d; the
//                   linear backward is a gather over the output range that can reach an input element, evaluated with the
//                   forward's own float expressions (reference: atomicAdd scatter). The reference's 3-d linear forward reads
//                   `d1 * inw * inw` for one of its eight taps (Upsample.py:221) — a typo invisible to its test (inh == inw);
//                   here every tap uses inh * inw.
//   CTC loss        — Cuda/Kernels/CTC.py:9-270: log-space forward / backward variables per sample (see below)
//   embedding       — Cuda/Kernels/Embedder.py:10-88: out[t, :] = vocabulary[word[t], :] (word -1: row of zeros);
//                   vocabulary[word[t], :] += scale * grad[t, :] with fp32 atomic adds (as the reference: rows repeat)
// None of these is on a timed path; they are HBM-bound one-output-per-thread kernels with lanes along the contiguous axis.
#include "common.h"

namespace {

constexpr int kT = 256;

inline int grid_for(size_t total) { return (int)((total + kT - 1) / kT); }

// ------------------------------------------------------------------------------------------------ point-wise costs
__global__ void __launch_bounds__(kT) bce_kernel(const float *__restrict__ scores, const int32_t *__restrict__ labels, float *__restrict__ grad,
                                                  float *__restrict__ terms, int numsamples, int spatial, size_t total) {
	const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
	if (i >= total) return;
	const float prob = 1.0f / (1.0f + expf(-scores[i]));
	const bool pos = labels[i] == 1;
	terms[i] = (pos ? -logf(prob) : -logf(1.0f - prob)) / spatial;
	grad[i] = ((pos ? 1.0f : 0.0f) - prob) / numsamples / spatial;
}

__global__ void __launch_bounds__(kT) hinge_kernel(const float *__restrict__ scores, const int32_t *__restrict__ labels, float *__restrict__ grad,
                                                    float *__restrict__ terms, int numsamples, int numcases, size_t total) {
	const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
	if (i >= total) return;
	const float score = scores[i];
	const int label = labels[i];
	terms[i] = fmaxf(0.0f, 1.0f - score * label) / numcases;
	grad[i] = score * label < 1.0f ? (float)label / numsamples / numcases : 0.0f;
}

__global__ void __launch_bounds__(kT) smooth_l1_kernel(const float *__restrict__ pred, const float *__restrict__ target, float *__restrict__ grad,
                                                        float *__restrict__ terms, float norm, float fullnorm, size_t total) {
	const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
	if (i >= total) return;
	const float diff = pred[i] - target[i];
	const float sign = diff > 0.0f ? 1.0f : -1.0f;
	const bool quad = diff * sign < 1.0f;
	terms[i] = quad ? diff * diff / 2.0f * norm : (sign * diff - 0.5f) * norm;
	grad[i] = quad ? diff * fullnorm : sign * fullnorm;
}

__global__ void __launch_bounds__(kT) l1_hinge_kernel(const float *__restrict__ x1, const float *__restrict__ x2, const int32_t *__restrict__ labels,
                                                       float *__restrict__ g1, float *__restrict__ g2, float *__restrict__ terms,
                                                       int numsamples, int numcases, size_t total) {
	const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
	if (i >= total) return;
	const float diff = x1[i] - x2[i];
	const float sign = diff > 0.0f ? 1.0f : -1.0f;
	const int label = labels[i / numcases];
	const float ad = fabsf(diff);
	terms[i] = (label == 0) ? fmaxf(0.0f, 1.0f - ad) / numcases : ad / numcases;
	const float inside = ad < 1.0f ? 1.0f : 0.0f;
	g1[i] = (label == 0 ? inside * -sign : sign) / numsamples / numcases;
	g2[i] = (label == 0 ? inside * sign : -sign) / numsamples / numcases;
}

// error += sum(terms), one workgroup, fixed order (element i goes to thread i % 1024; threads combined by block_sum)
__global__ void __launch_bounds__(1024) add_sum_kernel(const float *__restrict__ terms, size_t total, float *__restrict__ error) {
	__shared__ float smem[16];
	float acc = 0.f;
	for (size_t i = threadIdx.x; i < total; i += 1024) acc += terms[i];
	const float sum = block_sum(acc, smem);
	if (threadIdx.x == 0) *error += sum;
}

// ------------------------------------------------------------------------------------------------ PReLU
__global__ void __launch_bounds__(kT) prelu_fwd_kernel(const float *__restrict__ x, const float *__restrict__ slopes, float *__restrict__ y,
                                                        int div, int mapsize, int maps, size_t total) {
	const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
	if (i >= total) return;
	const int c = (int)((i / mapsize) % maps) / div;
	const float v = x[i];
	y[i] = v * (v > 0.0f ? 1.0f : slopes[c]);
}

__global__ void __launch_bounds__(kT) prelu_bwd_data_kernel(const float *__restrict__ dy, const float *__restrict__ slopes,
                                                             const float *__restrict__ x, float *__restrict__ dx, int div, int mapsize,
                                                             int maps, size_t total) {
	const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
	if (i >= total) return;
	const int c = (int)((i / mapsize) % maps) / div;
	const float v = x[i];
	dx[i] = dy[i] * ((v > 0.0f ? 1.0f : 0.0f) + (v <= 0.0f ? 1.0f : 0.0f) * slopes[c]);
}

// one workgroup per map: sum over (image, pixel) of dy * x * (x <= 0)
__global__ void __launch_bounds__(kT) prelu_bwd_params_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ out,
                                                               int n, int maps, int mapsize) {
	__shared__ float smem[16];
	const int c = blockIdx.x;
	float acc = 0.f;
	for (int b = 0; b < n; ++b) {
		const size_t base = ((size_t)b * maps + c) * mapsize;
		for (int p = threadIdx.x; p < mapsize; p += kT) {
			const float v = x[base + p];
			acc += dy[base + p] * v * (v <= 0.0f ? 1.0f : 0.0f);
		}
	}
	const float sum = block_sum(acc, smem);
	if (threadIdx.x == 0) out[c] = sum;
}

// ------------------------------------------------------------------------------------------------ reflection pad
__device__ __forceinline__ int reflect_src(int o, int in, int lpad) {
	// output position o -> input position (Cuda/Kernels/Pad.py:45-55 for non-negative pads)
	const int x = o - lpad;
	return x < 0 ? -x : (x >= in ? 2 * (in - 1) - x : x);
}

__global__ void __launch_bounds__(kT) reflectpad_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, int inh, int inw, int upad,
                                                             int lpad, int outh, int outw, size_t total) {
	const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
	if (i >= total) return;
	const int ow = (int)(i % outw), oh = (int)((i / outw) % outh);
	const size_t plane = i / ((size_t)outw * outh);
	y[i] = x[(plane * inh + reflect_src(oh, inh, upad)) * inw + reflect_src(ow, inw, lpad)];
}

// the output positions that read input position p of an axis: p + lpad always; the mirror images lpad - p (left border,
// 1 <= p <= lpad) and 2*(in - 1) - p + lpad (right border, in - 1 - rpad <= p <= in - 2)
__device__ __forceinline__ int reflect_dsts(int p, int in, int lpad, int rpad, int (&dst)[3]) {
	int n = 0;
	dst[n++] = p + lpad;
	if (p >= 1 && p <= lpad) dst[n++] = lpad - p;
	if (p >= in - 1 - rpad && p <= in - 2) dst[n++] = 2 * (in - 1) - p + lpad;
	return n;
}

__global__ void __launch_bounds__(kT) reflectpad_bwd_kernel(const float *__restrict__ dy, float *__restrict__ dx, int inh, int inw, int upad,
                                                             int bpad, int lpad, int rpad, size_t total) {
	const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
	if (i >= total) return;
	const int outh = inh + upad + bpad, outw = inw + lpad + rpad;
	const int w = (int)(i % inw), h = (int)((i / inw) % inh);
	const size_t plane = i / ((size_t)inw * inh);
	int hs[3], ws[3];
	const int nh = reflect_dsts(h, inh, upad, bpad, hs), nw = reflect_dsts(w, inw, lpad, rpad, ws);
	float acc = 0.f;
	for (int a = 0; a < nh; ++a)
		for (int b = 0; b < nw; ++b) acc += dy[(plane * outh + hs[a]) * outw + ws[b]];
	dx[i] = acc;
}

// ------------------------------------------------------------------------------------------------ up-sampling
struct UpGeom {
	int ind, inh, inw, outd, outh, outw, sd, sh, sw;
	float rd, rh, rw;
};

__global__ void __launch_bounds__(kT) upsample_nearest_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, UpGeom g, size_t total) {
	const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
	if (i >= total) return;
	const int ow = (int)(i % g.outw), oh = (int)((i / g.outw) % g.outh), od = (int)((i / ((size_t)g.outw * g.outh)) % g.outd);
	const size_t plane = i / ((size_t)g.outw * g.outh * g.outd);
	y[i] = x[((plane * g.ind + od / g.sd) * g.inh + oh / g.sh) * g.inw + ow / g.sw];
}

__global__ void __launch_bounds__(kT) upsample_nearest_bwd_kernel(const float *__restrict__ dy, float *__restrict__ dx, UpGeom g, size_t total) {
	const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
	if (i >= total) return;
	const int w = (int)(i % g.inw), h = (int)((i / g.inw) % g.inh), d = (int)((i / ((size_t)g.inw * g.inh)) % g.ind);
	const size_t plane = i / ((size_t)g.inw * g.inh * g.ind);
	float acc = 0.f;
	for (int a = 0; a < g.sd; ++a)
		for (int b = 0; b < g.sh; ++b)
			for (int c = 0; c < g.sw; ++c)
				acc += dy[((plane * g.outd + d * g.sd + a) * g.outh + h * g.sh + b) * g.outw + w * g.sw + c];
	dx[i] = acc;
}

// taps of one axis for output position o: i0 = (int)(r * o), i1 = i0 + (i0 < in - 1), weights 1 - f and f with f = r*o - i0
__device__ __forceinline__ void lin_taps(float r, int o, int in, int &i0, int &i1, float &w0, float &w1) {
	const float pos = r * o;
	i0 = (int)pos;
	i1 = i0 + (i0 < in - 1 ? 1 : 0);
	w1 = pos - i0;
	w0 = 1.0f - w1;
}

__global__ void __launch_bounds__(kT) upsample_linear_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, UpGeom g, size_t total) {
	const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
	if (i >= total) return;
	const int ow = (int)(i % g.outw), oh = (int)((i / g.outw) % g.outh), od = (int)((i / ((size_t)g.outw * g.outh)) % g.outd);
	const size_t plane = i / ((size_t)g.outw * g.outh * g.outd);
	int d0, d1, h0, h1, w0, w1;
	float ad0, ad1, ah0, ah1, aw0, aw1;
	lin_taps(g.rd, od, g.ind, d0, d1, ad0, ad1);
	lin_taps(g.rh, oh, g.inh, h0, h1, ah0, ah1);
	lin_taps(g.rw, ow, g.inw, w0, w1, aw0, aw1);
	const float *p = x + plane * g.ind * g.inh * g.inw;
	auto at = [&](int d, int h, int w) { return p[((size_t)d * g.inh + h) * g.inw + w]; };
	const float lo = ah0 * (aw0 * at(d0, h0, w0) + aw1 * at(d0, h0, w1)) + ah1 * (aw0 * at(d0, h1, w0) + aw1 * at(d0, h1, w1));
	if (g.ind == 1 && g.outd == 1) {        // 2-d: exactly the reference's expression (no depth factor)
		y[i] = lo;
		return;
	}
	const float hi = ah0 * (aw0 * at(d1, h0, w0) + aw1 * at(d1, h0, w1)) + ah1 * (aw0 * at(d1, h1, w0) + aw1 * at(d1, h1, w1));
	y[i] = ad0 * lo + ad1 * hi;
}

// weight with which output position o of an axis reads input position p (0 if it does not), from the forward's own taps
__device__ __forceinline__ float lin_weight(float r, int o, int in, int p) {
	int i0, i1;
	float w0, w1;
	lin_taps(r, o, in, i0, i1, w0, w1);
	return (i0 == p ? w0 : 0.0f) + (i1 == p ? w1 : 0.0f);      // (i0 == i1 at the far border: both weights land on p)
}

// the output range [lo, hi] that can read input position p: floor(r*o) in {p - 1, p}
__device__ __forceinline__ void lin_range(float r, int out, int p, int &lo, int &hi) {
	if (r <= 0.0f) {
		lo = 0, hi = out - 1;
		return;
	}
	lo = max(0, (int)floorf((p - 1) / r) - 1);
	hi = min(out - 1, (int)ceilf((p + 1) / r) + 1);
}

__global__ void __launch_bounds__(kT) upsample_linear_bwd_kernel(const float *__restrict__ dy, float *__restrict__ dx, UpGeom g, size_t total) {
	const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
	if (i >= total) return;
	const int w = (int)(i % g.inw), h = (int)((i / g.inw) % g.inh), d = (int)((i / ((size_t)g.inw * g.inh)) % g.ind);
	const size_t plane = i / ((size_t)g.inw * g.inh * g.ind);
	const bool flat = g.ind == 1 && g.outd == 1;
	int dlo = 0, dhi = 0, hlo, hhi, wlo, whi;
	if (!flat) lin_range(g.rd, g.outd, d, dlo, dhi);
	lin_range(g.rh, g.outh, h, hlo, hhi);
	lin_range(g.rw, g.outw, w, wlo, whi);
	const float *p = dy + plane * g.outd * g.outh * g.outw;
	float acc = 0.f;
	for (int od = dlo; od <= dhi; ++od) {
		const float ad = flat ? 1.0f : lin_weight(g.rd, od, g.ind, d);
		if (ad == 0.0f) continue;
		for (int oh = hlo; oh <= hhi; ++oh) {
			const float ah = lin_weight(g.rh, oh, g.inh, h);
			if (ah == 0.0f) continue;
			for (int ow = wlo; ow <= whi; ++ow) {
				const float aw = lin_weight(g.rw, ow, g.inw, w);
				if (aw != 0.0f) acc += (flat ? ah * aw : ad * ah * aw) * p[((size_t)od * g.outh + oh) * g.outw + ow];
			}
		}
	}
	dx[i] = acc;
}

// ------------------------------------------------------------------------------------------------ embedding
__global__ void __launch_bounds__(kT) embed_fwd_kernel(const int32_t *__restrict__ words, const float *__restrict__ vocab, float *__restrict__ out,
                                                        int embsize, size_t total) {
	const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
	if (i >= total) return;
	const int word = words[i / embsize];
	out[i] = word == -1 ? 0.0f : vocab[(size_t)word * embsize + i % embsize];
}

__global__ void __launch_bounds__(kT) embed_bwd_kernel(const int32_t *__restrict__ words, const float *__restrict__ grad, float *__restrict__ vocab,
                                                        float scale, int embsize, size_t total) {
	const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
	if (i >= total) return;
	const int word = words[i / embsize];
	if (word == -1) return;
	atomicAdd(&vocab[(size_t)word * embsize + i % embsize], scale * grad[i]);
}

// ------------------------------------------------------------------------------------------------ CTC loss
// Cuda/Kernels/CTC.py:9-192 (ctcmod.ctcLoss; caller Cost/CTC.py:23-30): one workgroup per sample. probs (T, batch, vocab) are
// softmax outputs; the extended label sequence of sample b is blank, l0, blank, l1, ..., blank (S = 2 L + 1 positions).
//   alphas : forward variables in log space, kept in global memory (T x S per sample at row offset T * (2 * off[b] + b))
//   betas  : backward variables, two rows in LDS; per time step the positions that carry the same label are summed
//            (logPlus, ascending position) and grad[t, b, v] = -p + exp(sum_v - log p + nll_b)
// The reference groups equal labels with an in-kernel radix sort; here the caller passes, per sample, the positions sorted
// by label (stable) and the segment bounds — the labels are host data in the reference's own API (`lengths` is), so the
// grouping is a numpy argsort.
constexpr int kCtcMaxS = 4096;

__device__ __forceinline__ float log_plus(float a, float b) {
	if (a <= -INFINITY) return b;
	if (b <= -INFINITY) return a;
	return log1pf(expf(-fabsf(a - b))) + fmaxf(a, b);
}

struct CtcArgs {
	const float *probs;
	const int32_t *datalen, *labels, *offsets;      // offsets[batch + 1]: prefix sums of the label lengths
	float *alphas, *nll, *grad;
	const int32_t *order, *seg_start, *seg_label, *seg_off;     // per sample: positions by label, segment bounds (local), labels
	int T, batch, vocab, blank;
};

__global__ void __launch_bounds__(256) ctc_alphas_kernel(CtcArgs a) {
	__shared__ int32_t ext[kCtcMaxS];
	const int b = blockIdx.x;
	const int off = a.offsets[b], S = 2 * (a.offsets[b + 1] - off) + 1;
	const float *p = a.probs + (size_t)b * a.vocab;
	const size_t tstride = (size_t)a.batch * a.vocab;
	float *alpha = a.alphas + (size_t)a.T * (2 * off + b);

	for (int i = threadIdx.x; i < S; i += 256) {
		const int label = (i & 1) == 0 ? a.blank : a.labels[off + i / 2];
		ext[i] = label;
		alpha[i] = i < 2 ? logf(p[label]) : -INFINITY;
	}
	__syncthreads();
	const int T = a.datalen[b];
	for (int t = 1; t < T; ++t) {
		for (int i = threadIdx.x; i < S; i += 256) {
			float prev = alpha[(size_t)(t - 1) * S + i];
			if (i > 0) {
				prev = log_plus(prev, alpha[(size_t)(t - 1) * S + i - 1]);
				if (i > 1 && ext[i] != a.blank && ext[i] != ext[i - 2]) prev = log_plus(prev, alpha[(size_t)(t - 1) * S + i - 2]);
			}
			alpha[(size_t)t * S + i] = prev + logf(p[(size_t)t * tstride + ext[i]]);
		}
		__syncthreads();
	}
	if (threadIdx.x == 0)
		a.nll[b] = -(S > 1 ? log_plus(alpha[(size_t)(T - 1) * S + S - 2], alpha[(size_t)(T - 1) * S + S - 1]) : alpha[(size_t)(T - 1) * S]);
}

__global__ void __launch_bounds__(256) ctc_betas_kernel(CtcArgs a) {
	__shared__ int32_t ext[kCtcMaxS];
	__shared__ float beta[2][kCtcMaxS];
	const int b = blockIdx.x;
	const int off = a.offsets[b], S = 2 * (a.offsets[b + 1] - off) + 1;
	const float *p = a.probs + (size_t)b * a.vocab;
	float *g = a.grad + (size_t)b * a.vocab;
	const size_t tstride = (size_t)a.batch * a.vocab;
	const float *alpha = a.alphas + (size_t)a.T * (2 * off + b);
	const float loglike = a.nll[b];
	if (loglike >= INFINITY) return;                       // no valid alignment: the gradient stays zero

	for (int i = threadIdx.x; i < S; i += 256) ext[i] = (i & 1) == 0 ? a.blank : a.labels[off + i / 2];
	const int32_t *order = a.order + (2 * off + b);
	const int seg0 = a.seg_off[b], nseg = a.seg_off[b + 1] - seg0;
	const int32_t *seg_start = a.seg_start + seg0 + b;     // nseg + 1 entries per sample
	const int32_t *seg_label = a.seg_label + seg0;
	const int T = a.datalen[b];
	__syncthreads();

	int src = 0;
	for (int t = T - 1; t >= 0; --t) {
		if (t < T - 1) {
			const int dst = src ^ 1;
			for (int i = threadIdx.x; i < S; i += 256) {
				float next = beta[src][i];
				if (i < S - 1) {
					next = log_plus(next, beta[src][i + 1]);
					if (i < S - 2 && ext[i] != a.blank && ext[i] != ext[i + 2]) next = log_plus(next, beta[src][i + 2]);
				}
				beta[dst][i] = next + logf(p[(size_t)t * tstride + ext[i]]);
			}
			src = dst;
		} else {
			for (int i = threadIdx.x; i < S; i += 256)
				beta[0][i] = i >= S - 2 ? logf(p[(size_t)(T - 1) * tstride + ext[i]]) : -INFINITY;
		}
		for (int i = threadIdx.x; i < a.vocab; i += 256) g[(size_t)t * tstride + i] = -p[(size_t)t * tstride + i];
		__syncthreads();
		for (int j = threadIdx.x; j < nseg; j += 256) {
			float gr = -INFINITY;
			for (int k = seg_start[j]; k < seg_start[j + 1]; ++k) {
				const int i = order[k];
				gr = log_plus(gr, alpha[(size_t)t * S + i] + beta[src][i]);
			}
			const size_t at = (size_t)t * tstride + seg_label[j];
			const float data = p[at];
			if (data > 0.0f) g[at] += expf(gr - logf(data) + loglike);
		}
		__syncthreads();
	}
}

}  // namespace

extern "C" {

int pz_ctc_loss(const float *probs, const int32_t *datalen, const int32_t *labels, const int32_t *offsets, const int32_t *order,
                const int32_t *seg_start, const int32_t *seg_label, const int32_t *seg_off, int T, int batch, int vocab, int blank,
                int max_positions, float *alphas, float *nll, float *grad, float *error, pz_stream_t stream) {
	PZ_REQUIRE(probs && datalen && labels && offsets && order && seg_start && seg_label && seg_off && alphas && nll && grad && error,
	           "pz_ctc_loss: null argument");
	PZ_REQUIRE(T > 0 && batch > 0 && vocab > 0 && blank >= 0 && blank < vocab, "pz_ctc_loss: bad geometry");
	PZ_REQUIRE(max_positions <= kCtcMaxS, "pz_ctc_loss: %d extended label positions (the kernel holds %d)", max_positions, kCtcMaxS);
	hipStream_t st = pz::as_stream(stream);
	CtcArgs a{probs, datalen, labels, offsets, alphas, nll, grad, order, seg_start, seg_label, seg_off, T, batch, vocab, blank};
	ctc_alphas_kernel<<<batch, 256, 0, st>>>(a);
	PZ_LAUNCH_CHECK();
	ctc_betas_kernel<<<batch, 256, 0, st>>>(a);
	PZ_LAUNCH_CHECK();
	add_sum_kernel<<<1, 1024, 0, st>>>(nll, (size_t)batch, error);      // error += sum_b nll[b], fixed order
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_cost_pointwise(int kind, const float *a, const void *b, const int32_t *labels, float *error, float *grad, float *grad2, float *terms,
                      size_t total, int numsamples, int numcases, float norm, float fullnorm, pz_stream_t stream) {
	PZ_REQUIRE(a && error && grad && terms && total > 0, "pz_cost_pointwise: null argument or empty tensor");
	hipStream_t st = pz::as_stream(stream);
	switch (kind) {
	case PZ_COST_BCE:
		PZ_REQUIRE(labels, "pz_cost_pointwise(bce): labels missing");
		bce_kernel<<<grid_for(total), kT, 0, st>>>(a, labels, grad, terms, numsamples, numcases, total);
		break;
	case PZ_COST_HINGE:
		PZ_REQUIRE(labels, "pz_cost_pointwise(hinge): labels missing");
		hinge_kernel<<<grid_for(total), kT, 0, st>>>(a, labels, grad, terms, numsamples, numcases, total);
		break;
	case PZ_COST_SMOOTH_L1:
		PZ_REQUIRE(b, "pz_cost_pointwise(smoothL1): target missing");
		smooth_l1_kernel<<<grid_for(total), kT, 0, st>>>(a, (const float *)b, grad, terms, norm, fullnorm, total);
		break;
	case PZ_COST_L1_HINGE:
		PZ_REQUIRE(b && labels && grad2 && numcases > 0, "pz_cost_pointwise(l1Hinge): second operand, labels or second gradient missing");
		l1_hinge_kernel<<<grid_for(total), kT, 0, st>>>(a, (const float *)b, labels, grad, grad2, terms, numsamples, numcases, total);
		break;
	default:
		PZ_REQUIRE(false, "pz_cost_pointwise: unknown cost %d", kind);
	}
	PZ_LAUNCH_CHECK();
	add_sum_kernel<<<1, 1024, 0, st>>>(terms, total, error);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_prelu_fwd(const float *x, const float *slopes, float *y, int n, int maps, int mapsize, int shared, pz_stream_t stream) {
	PZ_REQUIRE(x && slopes && y && n > 0 && maps > 0 && mapsize > 0, "pz_prelu_fwd: bad arguments");
	const size_t total = (size_t)n * maps * mapsize;
	prelu_fwd_kernel<<<grid_for(total), kT, 0, pz::as_stream(stream)>>>(x, slopes, y, shared ? maps : 1, mapsize, maps, total);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_prelu_bwd_data(const float *dy, const float *slopes, const float *x, float *dx, int n, int maps, int mapsize, int shared,
                      pz_stream_t stream) {
	PZ_REQUIRE(dy && slopes && x && dx && n > 0 && maps > 0 && mapsize > 0, "pz_prelu_bwd_data: bad arguments");
	const size_t total = (size_t)n * maps * mapsize;
	prelu_bwd_data_kernel<<<grid_for(total), kT, 0, pz::as_stream(stream)>>>(dy, slopes, x, dx, shared ? maps : 1, mapsize, maps, total);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_prelu_bwd_params(const float *x, const float *dy, float *per_map, int n, int maps, int mapsize, pz_stream_t stream) {
	PZ_REQUIRE(x && dy && per_map && n > 0 && maps > 0 && mapsize > 0, "pz_prelu_bwd_params: bad arguments");
	prelu_bwd_params_kernel<<<maps, kT, 0, pz::as_stream(stream)>>>(x, dy, per_map, n, maps, mapsize);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_reflectpad2d_fwd(const float *x, float *y, size_t planes, int inh, int inw, int upad, int bpad, int lpad, int rpad, pz_stream_t stream) {
	PZ_REQUIRE(x && y && planes > 0 && inh > 0 && inw > 0, "pz_reflectpad2d_fwd: bad arguments");
	PZ_REQUIRE(upad >= 0 && bpad >= 0 && lpad >= 0 && rpad >= 0 && inh >= (upad > bpad ? upad : bpad) + 1 &&
	               inw >= (lpad > rpad ? lpad : rpad) + 1,
	           "pz_reflectpad2d_fwd: pads (%d, %d, %d, %d) do not fit a %d x %d map", upad, bpad, lpad, rpad, inh, inw);
	const int outh = inh + upad + bpad, outw = inw + lpad + rpad;
	const size_t total = planes * outh * outw;
	reflectpad_fwd_kernel<<<grid_for(total), kT, 0, pz::as_stream(stream)>>>(x, y, inh, inw, upad, lpad, outh, outw, total);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_reflectpad2d_bwd(const float *dy, float *dx, size_t planes, int inh, int inw, int upad, int bpad, int lpad, int rpad,
                        pz_stream_t stream) {
	PZ_REQUIRE(dy && dx && planes > 0 && inh > 0 && inw > 0, "pz_reflectpad2d_bwd: bad arguments");
	PZ_REQUIRE(upad >= 0 && bpad >= 0 && lpad >= 0 && rpad >= 0 && inh >= (upad > bpad ? upad : bpad) + 1 &&
	               inw >= (lpad > rpad ? lpad : rpad) + 1,
	           "pz_reflectpad2d_bwd: pads (%d, %d, %d, %d) do not fit a %d x %d map", upad, bpad, lpad, rpad, inh, inw);
	const size_t total = planes * inh * inw;
	reflectpad_bwd_kernel<<<grid_for(total), kT, 0, pz::as_stream(stream)>>>(dy, dx, inh, inw, upad, bpad, lpad, rpad, total);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

static int up_geom(UpGeom &g, int ind, int inh, int inw, int sd, int sh, int sw, int linear) {
	PZ_REQUIRE(ind > 0 && inh > 0 && inw > 0 && sd > 0 && sh > 0 && sw > 0, "pz_upsample: bad geometry");
	g = UpGeom{ind, inh, inw, ind * sd, inh * sh, inw * sw, sd, sh, sw, 0.f, 0.f, 0.f};
	if (linear) {
		// r = (in - 1) / (out - 1) computed in double and rounded once, as Python's float division + np.float32(...) does
		// (Upsample.py:336,408); a one-element output axis has a single position 0: r is irrelevant there
		g.rd = g.outd > 1 ? (float)((double)(ind - 1) / (double)(g.outd - 1)) : 0.f;
		g.rh = g.outh > 1 ? (float)((double)(inh - 1) / (double)(g.outh - 1)) : 0.f;
		g.rw = g.outw > 1 ? (float)((double)(inw - 1) / (double)(g.outw - 1)) : 0.f;
	}
	return PZ_OK;
}

int pz_upsample_fwd(const float *x, float *y, size_t planes, int ind, int inh, int inw, int sd, int sh, int sw, int linear, pz_stream_t stream) {
	PZ_REQUIRE(x && y && planes > 0, "pz_upsample_fwd: bad arguments");
	UpGeom g;
	if (int rc = up_geom(g, ind, inh, inw, sd, sh, sw, linear)) return rc;
	const size_t total = planes * g.outd * g.outh * g.outw;
	if (linear)
		upsample_linear_fwd_kernel<<<grid_for(total), kT, 0, pz::as_stream(stream)>>>(x, y, g, total);
	else
		upsample_nearest_fwd_kernel<<<grid_for(total), kT, 0, pz::as_stream(stream)>>>(x, y, g, total);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_upsample_bwd(const float *dy, float *dx, size_t planes, int ind, int inh, int inw, int sd, int sh, int sw, int linear,
                    pz_stream_t stream) {
	PZ_REQUIRE(dy && dx && planes > 0, "pz_upsample_bwd: bad arguments");
	UpGeom g;
	if (int rc = up_geom(g, ind, inh, inw, sd, sh, sw, linear)) return rc;
	const size_t total = planes * ind * inh * inw;
	if (linear)
		upsample_linear_bwd_kernel<<<grid_for(total), kT, 0, pz::as_stream(stream)>>>(dy, dx, g, total);
	else
		upsample_nearest_bwd_kernel<<<grid_for(total), kT, 0, pz::as_stream(stream)>>>(dy, dx, g, total);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_embed_fwd(const int32_t *words, const float *vocab, float *out, size_t tokens, int embsize, pz_stream_t stream) {
	PZ_REQUIRE(words && vocab && out && tokens > 0 && embsize > 0, "pz_embed_fwd: bad arguments");
	const size_t total = tokens * embsize;
	embed_fwd_kernel<<<grid_for(total), kT, 0, pz::as_stream(stream)>>>(words, vocab, out, embsize, total);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_embed_bwd_params(const int32_t *words, const float *grad, float *vocab, float scale, size_t tokens, int embsize, pz_stream_t stream) {
	PZ_REQUIRE(words && grad && vocab && tokens > 0 && embsize > 0, "pz_embed_bwd_params: bad arguments");
	const size_t total = tokens * embsize;
	embed_bwd_kernel<<<grid_for(total), kT, 0, pz::as_stream(stream)>>>(words, grad, vocab, scale, embsize, total);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

}  // extern "C"
