// Element-wise family: activations fwd/bwd, dropout, axpy/add/mul/linear, weight decay, the optimizer updates.
// One generic streaming kernel, specialised per op by a functor: 16 B per lane per operand, grid-stride, with a
// scalar strided variant for the reference's `slice=` launches. All of these are pure HBM-bound.
// Replaces the ElementwiseKernel objects of Cuda/Kernels/ElementWise.py (launcher Cuda/SourceModule.py:176-226);
// each functor restates the reference kernel body it names (CPU twin: CPU/Kernels/ElementWise.py).
#include "common.h"

namespace {

constexpr int kMaxPtrs = 5, kMaxScalars = 5;

struct EltArgs {
	float *p[kMaxPtrs];
	float s[kMaxScalars];
	size_t count;
	long start, stop, step;
};

__device__ __forceinline__ float gt0(float x) { return x > 0.f ? 1.f : 0.f; }
__device__ __forceinline__ float le0(float x) { return x <= 0.f ? 1.f : 0.f; }

// Functor contract: NP operands, RD/WR = bitmask of operands read / written, f() updates v[] in place.
#define PZ_ELT(NAME, NP_, RD_, WR_, BODY)                                                        \
	struct NAME {                                                                                \
		static constexpr int NP = NP_;                                                           \
		static constexpr unsigned RD = RD_, WR = WR_;                                            \
		__device__ static __forceinline__ void f(float (&v)[NP_], const float *s) { BODY }       \
	};

PZ_ELT(OpSigmoid, 2, 0b10, 0b01, v[0] = 1.f / (1.f + expf(-v[1]));)                                   // ElementWise.py:9-33
PZ_ELT(OpSigmoidDer, 3, 0b110, 0b001, v[0] = v[1] * v[2] * (1.f - v[2]);)                             // :36-61
PZ_ELT(OpTanh, 2, 0b10, 0b01, v[0] = tanhf(v[1]);)                                                   // :64-88
PZ_ELT(OpTanhDer, 3, 0b110, 0b001, v[0] = v[1] * (1.f - v[2] * v[2]);)                                // :91-116
PZ_ELT(OpRelu, 2, 0b10, 0b01, v[0] = v[1] * gt0(v[1]);)                                              // :119-144
PZ_ELT(OpReluDer, 3, 0b110, 0b001, v[0] = v[1] * gt0(v[2]);)                                         // :147-172
PZ_ELT(OpLeakyRelu, 2, 0b10, 0b01, v[0] = v[1] * (gt0(v[1]) + s[0] * le0(v[1]));)                    // :175-203
PZ_ELT(OpLeakyReluDer, 3, 0b110, 0b001, v[0] = v[1] * (gt0(v[2]) + s[0] * le0(v[2]));)               // :206-237
PZ_ELT(OpElu, 2, 0b10, 0b01, v[0] = v[1] * gt0(v[1]) + s[0] * (expf(v[1]) - 1.f) * le0(v[1]);)       // :240-269
PZ_ELT(OpEluDer, 3, 0b110, 0b001, v[0] = v[1] * (gt0(v[2]) + (v[2] + s[0]) * le0(v[2]));)            // :272-301
PZ_ELT(OpSoftPlus, 2, 0b10, 0b01, v[0] = logf(1.f + expf(v[1]));)                                    // :304-329
PZ_ELT(OpSoftPlusDer, 3, 0b110, 0b001, v[0] = v[1] * (1.f - expf(-v[2]));)                           // :332-357
PZ_ELT(OpClip, 2, 0b10, 0b01,                                                                         // :360-388
       const float x = v[1]; const float a = s[0]; const float b = s[1];
       v[0] = x * ((x > a && x < b) ? 1.f : 0.f) + a * (x <= a ? 1.f : 0.f) + b * (x >= b ? 1.f : 0.f);)
PZ_ELT(OpClipDer, 3, 0b110, 0b001, v[0] = v[1] * ((v[2] > s[0] && v[2] < s[1]) ? 1.f : 0.f);)        // :391-424
PZ_ELT(OpGelu, 2, 0b10, 0b01, v[0] = 0.5f * v[1] * (1.f + erff(v[1] / 1.4142135623730951f));)        // :427-456
PZ_ELT(OpGeluDer, 3, 0b110, 0b001,                                                                    // :459-492
       const float d = v[2];
       v[0] = v[1] * (0.5f * (1.f + erff(d / 1.4142135623730951f)) + d / 1.7724538509055159f * expf(-0.5f * d * d));)
PZ_ELT(OpDropout, 3, 0b110, 0b001,                                                                    // :495-536
       v[0] = v[1] * (__float_as_uint(v[2]) < __float_as_uint(s[0]) ? 1.f : 0.f) / s[1];)
PZ_ELT(OpAxpy, 2, 0b11, 0b01, v[0] = v[0] + v[1] * s[0];)                                            // :582-606
PZ_ELT(OpAdd, 3, 0b110, 0b001, v[0] = v[1] * s[0] + v[2] * s[1];)                                    // :1017-1045
PZ_ELT(OpMul, 3, 0b110, 0b001, v[0] = v[1] * v[2];)                                                  // :1048-1072
PZ_ELT(OpLinear, 2, 0b10, 0b01, v[0] = s[0] * v[1] + s[1];)                                          // :1075-1099
PZ_ELT(OpAbs, 2, 0b10, 0b01, v[0] = fabsf(v[1]);)                                                    // :1119-1124
PZ_ELT(OpWeightDecay, 2, 0b11, 0b01, v[0] -= s[0] * v[1];)                                           // :1111-1116
PZ_ELT(OpL1Penalty, 3, 0b110, 0b001,                                                                  // :1127-1132
       v[0] = v[1] - s[0] * ((0.f <= v[2] ? 1.f : 0.f) - (v[2] < 0.f ? 1.f : 0.f));)
PZ_ELT(OpL1Grad, 3, 0b110, 0b001, v[0] = (v[1] - v[2] > 0.f ? -s[0] : s[0]);)                        // :1135-1140
PZ_ELT(OpRbm, 3, 0b110, 0b001, const float p = 1.f / (1.f + expf(-v[1])); v[0] = v[2] < p ? 1.f : 0.f;)   // :1102-1108

// The update rules read the gradient as v[1] * (last scalar): 1 normally (exact), 1/N when the data-parallel exchange left the
// arena holding the SUM over ranks (Grid.py:126-133 divides inside its reduce; here the division rides in the update kernel
// instead of a pass of its own over the arena — same rounding: one fp32 product per element)
PZ_ELT(OpAdam, 4, 0b1111, 0b1101,                                                                     // :710-757
       float g = v[1] * s[4];
       asm volatile("" : "+v"(g));                   // the product is rounded on its own (never contracted into the sums below)
       v[2] += s[1] * (g - v[2]);
       v[3] += s[2] * (g * g - v[3]);
       v[0] += s[0] * v[2] / (sqrtf(v[3]) + s[3]);)
PZ_ELT(OpClassicMomSGD, 3, 0b111, 0b101,                                                              // :760-806
       float g = v[1] * s[2];
       asm volatile("" : "+v"(g));
       v[2] = s[1] * v[2] + s[0] * g; v[0] += v[2];)
PZ_ELT(OpNesterovMomSGD, 3, 0b111, 0b101,                                                             // :809-857
       const float m = v[2];
       float g = v[1] * s[2];
       asm volatile("" : "+v"(g));
       v[2] = s[1] * m + s[0] * g;
       v[0] += s[1] * s[1] * m + (1.f + s[1]) * s[0] * g;)
PZ_ELT(OpRmsprop, 3, 0b111, 0b101,                                                                    // :860-905
       v[2] = s[1] * v[2] + (1.f - s[1]) * v[1] * v[1];
       v[0] += s[0] * v[1] / (sqrtf(v[2]) + s[2]);)
PZ_ELT(OpAdagrad, 3, 0b111, 0b101, v[2] += v[1] * v[1]; v[0] += s[0] * v[1] / (sqrtf(v[2]) + s[1]);) // :662-707
PZ_ELT(OpAdadelta, 4, 0b1111, 0b1101,                                                                 // :609-659
       const float g = v[1];
       v[2] += (1.f - s[0]) * (g * g - v[2]);
       const float dx = sqrtf((v[3] + s[1]) / (v[2] + s[1])) * g;
       v[3] += (1.f - s[0]) * (dx * dx - v[3]);
       v[0] += dx;)
PZ_ELT(OpRmspropGraves, 5, 0b11111, 0b11101,                                                          // :908-960
       const float g = v[1];
       v[3] = s[1] * v[3] + (1.f - s[1]) * g * g;
       v[2] = s[1] * v[2] + (1.f - s[1]) * g;
       v[4] = s[2] * v[4] + s[0] * g / sqrtf(v[3] - v[2] * v[2] + s[3]);
       v[0] += v[4];)
PZ_ELT(OpSmorms3, 5, 0b11111, 0b11101,                                                                // :963-1014
       const float g = v[1];
       const float r = 1.f / (v[2] + 1.f);
       const float mgi = (1.f - r) * v[3] + r * g;
       const float msi = (1.f - r) * v[4] + r * g * g;
       const float x = mgi * mgi / (msi + s[1]);
       v[2] = 1.f + v[2] * (1.f - x);
       v[3] = mgi;
       v[4] = msi;
       v[0] += g * fminf(s[0], x) / (sqrtf(msi) + s[1]);)

PZ_ELT(OpAdd3, 3, 0b110, 0b001, v[0] = v[1] + v[2];)      // residual sum / gradient fan-in in one 12 B/elem pass
// residual sum followed by an in-place ReLU, and gradient fan-in gated by that ReLU's output sign (reluKer / reluDerKer
// rules, ElementWise.py:119-172) — one pass each instead of two
PZ_ELT(OpAdd3Relu, 3, 0b110, 0b001, const float t = v[1] + v[2]; v[0] = t * gt0(t);)
PZ_ELT(OpAdd3Gate, 4, 0b1110, 0b0001, v[0] = (v[1] + v[2]) * gt0(v[3]);)
PZ_ELT(OpIadd, 2, 0b11, 0b01, v[0] += v[1];)              // Cuda/GPUArray.py:127-136 inplaceArithmKer
PZ_ELT(OpImul, 2, 0b11, 0b01, v[0] *= v[1];)

template <typename OP>
__global__ void __launch_bounds__(256) elt_dense_kernel(EltArgs a) {
	const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t nthreads = (size_t)gridDim.x * blockDim.x;
	const size_t n4 = a.count >> 2;

	for (size_t i = tid; i < n4; i += nthreads) {
		float4 r[OP::NP];
#pragma unroll
		for (int k = 0; k < OP::NP; ++k)
			if (OP::RD >> k & 1u) r[k] = reinterpret_cast<const float4 *>(a.p[k])[i];

		float v[4][OP::NP];
#pragma unroll
		for (int k = 0; k < OP::NP; ++k) {
			v[0][k] = r[k].x, v[1][k] = r[k].y, v[2][k] = r[k].z, v[3][k] = r[k].w;
		}
#pragma unroll
		for (int e = 0; e < 4; ++e) OP::f(v[e], a.s);

#pragma unroll
		for (int k = 0; k < OP::NP; ++k)
			if (OP::WR >> k & 1u) reinterpret_cast<float4 *>(a.p[k])[i] = make_float4(v[0][k], v[1][k], v[2][k], v[3][k]);
	}

	const size_t tail = n4 << 2;
	if (tid < a.count - tail) {
		const size_t i = tail + tid;
		float v[OP::NP];
#pragma unroll
		for (int k = 0; k < OP::NP; ++k)
			if (OP::RD >> k & 1u) v[k] = a.p[k][i];
		OP::f(v, a.s);
#pragma unroll
		for (int k = 0; k < OP::NP; ++k)
			if (OP::WR >> k & 1u) a.p[k][i] = v[k];
	}
}

// i = start + t*step < stop  (Cuda/SourceModule.py:216-226 `${name}_strided`); also the unaligned dense fallback
template <typename OP>
__global__ void __launch_bounds__(256) elt_strided_kernel(EltArgs a, size_t nwork) {
	for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < nwork; t += (size_t)gridDim.x * blockDim.x) {
		const long i = a.start + (long)t * a.step;
		if (i >= a.stop) break;
		float v[OP::NP];
#pragma unroll
		for (int k = 0; k < OP::NP; ++k)
			if (OP::RD >> k & 1u) v[k] = a.p[k][i];
		OP::f(v, a.s);
#pragma unroll
		for (int k = 0; k < OP::NP; ++k)
			if (OP::WR >> k & 1u) a.p[k][i] = v[k];
	}
}

// dropout2d: one mask word per feature map — Cuda/Kernels/ElementWise.py:539-579
__global__ void __launch_bounds__(256) dropout2d_kernel(float *out, const float *in, const uint32_t *bits, uint32_t v, float p,
                                                         int mapsize, size_t count) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
		out[i] = in[i] * (bits[i / mapsize] < v ? 1.f : 0.f) / p;
}

__global__ void __launch_bounds__(256) cast_i32_f32_kernel(float *out, const int32_t *in, size_t n) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (float)in[i];
}

__global__ void __launch_bounds__(256) cast_f32_i32_kernel(int32_t *out, const float *in, size_t n) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (int32_t)in[i];
}

// storage casts between fp32 and fp16 (GPUArray.astype, castFP32toFP16 / castFP16toFP32 of Cuda/Kernels/ElementWise.py:1143-1156):
// round-to-nearest-even like numpy; arithmetic stays fp32 everywhere
__global__ void __launch_bounds__(256) cast_f32_f16_kernel(_Float16 *out, const float *in, size_t n) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (_Float16)in[i];
}

__global__ void __launch_bounds__(256) cast_f16_f32_kernel(float *out, const _Float16 *in, size_t n) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (float)in[i];
}

template <typename OP>
int launch(const EltArgs &a, int nptrs, int nscalars, int need_scalars, bool dense, hipStream_t st) {
	PZ_REQUIRE(nptrs == OP::NP, "pz_eltwise: op expects %d operands, got %d", OP::NP, nptrs);
	PZ_REQUIRE(nscalars >= need_scalars, "pz_eltwise: op expects %d scalars, got %d", need_scalars, nscalars);

	bool aligned = true;
	for (int k = 0; k < OP::NP; ++k) aligned = aligned && (((uintptr_t)a.p[k] & 15) == 0);

	if (dense && aligned) {
		elt_dense_kernel<OP><<<pz::stream_grid((a.count >> 2) + 1, 256), 256, 0, st>>>(a);
	} else {
		const size_t nwork = a.stop > a.start ? (size_t)((a.stop - a.start + a.step - 1) / a.step) : 0;
		if (nwork == 0) return PZ_OK;
		elt_strided_kernel<OP><<<pz::stream_grid(nwork, 256), 256, 0, st>>>(a, nwork);
	}
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

}  // namespace

// Many small `out = alpha*x + beta*y` in one launch (one workgroup per job): the 106 BatchNorm parameter-gradient
// accumulates of a ResNet-50 step (Modules/BatchNormND.py:86-92: two addVectorToVector per layer) as two launches instead
// of 106. Same expression, same file, same contraction as OpAdd.
struct MultiAddJob {
	float *out;
	const float *x, *y;
	float alpha, beta;
	unsigned n, pad;
};
struct MultiAddArgs {
	MultiAddJob job[PZ_MULTI_ADD_MAX];
};

__global__ void __launch_bounds__(256) multi_add_kernel(MultiAddArgs a) {
	const MultiAddJob j = a.job[blockIdx.x];
	for (unsigned i = threadIdx.x; i < j.n; i += 256) {
		float v[3] = {0.f, j.x[i], j.y[i]};
		const float s[2] = {j.alpha, j.beta};
		v[0] = v[1] * s[0] + v[2] * s[1];
		j.out[i] = v[0];
	}
}

extern "C" {

int pz_multi_add(int njobs, float *const *out, const float *const *x, const float *const *y, const float *alpha,
                 const float *beta, const unsigned *n, pz_stream_t stream) {
	PZ_REQUIRE(njobs >= 1 && njobs <= PZ_MULTI_ADD_MAX && out && x && y && alpha && beta && n, "pz_multi_add: bad arguments");
	MultiAddArgs a{};
	for (int i = 0; i < njobs; ++i) {
		PZ_REQUIRE(out[i] && x[i] && y[i], "pz_multi_add: job %d has a null operand", i);
		a.job[i] = MultiAddJob{out[i], x[i], y[i], alpha[i], beta[i], n[i], 0u};
	}
	multi_add_kernel<<<njobs, 256, 0, pz::as_stream(stream)>>>(a);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_eltwise(int op, size_t count, void *const *ptrs, int nptrs, const float *scalars, int nscalars, int64_t start,
               int64_t stop, int64_t step, pz_stream_t stream) {
	PZ_REQUIRE(op >= 0 && op < PZ_OP_COUNT, "pz_eltwise: unknown op %d", op);
	PZ_REQUIRE(nptrs >= 1 && nptrs <= kMaxPtrs && nscalars >= 0 && nscalars <= kMaxScalars, "pz_eltwise: bad operand counts");
	PZ_REQUIRE(step >= 1 && start >= 0 && stop <= (int64_t)count, "pz_eltwise: bad slice (%ld, %ld, %ld) for %zu elements",
	           (long)start, (long)stop, (long)step, count);
	if (count == 0) return PZ_OK;

	EltArgs a{};
	for (int k = 0; k < nptrs; ++k) {
		PZ_REQUIRE(ptrs[k] != nullptr, "pz_eltwise: operand %d is null", k);
		a.p[k] = (float *)ptrs[k];
	}
	for (int k = 0; k < nscalars; ++k) a.s[k] = scalars[k];
	a.count = count, a.start = start, a.stop = stop, a.step = step;

	const bool dense = start == 0 && step == 1 && stop == (int64_t)count;
	hipStream_t st = pz::as_stream(stream);

#define PZ_CASE(ID, OP, NS) \
	case ID:                \
		return launch<OP>(a, nptrs, nscalars, NS, dense, st);

	switch (op) {
		PZ_CASE(PZ_OP_SIGMOID, OpSigmoid, 0)
		PZ_CASE(PZ_OP_SIGMOID_DER, OpSigmoidDer, 0)
		PZ_CASE(PZ_OP_TANH, OpTanh, 0)
		PZ_CASE(PZ_OP_TANH_DER, OpTanhDer, 0)
		PZ_CASE(PZ_OP_RELU, OpRelu, 0)
		PZ_CASE(PZ_OP_RELU_DER, OpReluDer, 0)
		PZ_CASE(PZ_OP_LEAKY_RELU, OpLeakyRelu, 1)
		PZ_CASE(PZ_OP_LEAKY_RELU_DER, OpLeakyReluDer, 1)
		PZ_CASE(PZ_OP_ELU, OpElu, 1)
		PZ_CASE(PZ_OP_ELU_DER, OpEluDer, 1)
		PZ_CASE(PZ_OP_SOFTPLUS, OpSoftPlus, 0)
		PZ_CASE(PZ_OP_SOFTPLUS_DER, OpSoftPlusDer, 0)
		PZ_CASE(PZ_OP_CLIP, OpClip, 2)
		PZ_CASE(PZ_OP_CLIP_DER, OpClipDer, 2)
		PZ_CASE(PZ_OP_GELU, OpGelu, 0)
		PZ_CASE(PZ_OP_GELU_DER, OpGeluDer, 0)
		PZ_CASE(PZ_OP_DROPOUT, OpDropout, 2)
		PZ_CASE(PZ_OP_AXPY, OpAxpy, 1)
		PZ_CASE(PZ_OP_ADD, OpAdd, 2)
		PZ_CASE(PZ_OP_MUL, OpMul, 0)
		PZ_CASE(PZ_OP_LINEAR, OpLinear, 2)
		PZ_CASE(PZ_OP_ABS, OpAbs, 0)
		PZ_CASE(PZ_OP_WEIGHT_DECAY, OpWeightDecay, 1)
		PZ_CASE(PZ_OP_L1_PENALTY, OpL1Penalty, 1)
		PZ_CASE(PZ_OP_L1_GRAD, OpL1Grad, 1)
		PZ_CASE(PZ_OP_RBM, OpRbm, 0)
		PZ_CASE(PZ_OP_ADAM, OpAdam, 5)
		PZ_CASE(PZ_OP_CLASSIC_MOM_SGD, OpClassicMomSGD, 3)
		PZ_CASE(PZ_OP_NESTEROV_MOM_SGD, OpNesterovMomSGD, 3)
		PZ_CASE(PZ_OP_RMSPROP, OpRmsprop, 3)
		PZ_CASE(PZ_OP_ADAGRAD, OpAdagrad, 2)
		PZ_CASE(PZ_OP_ADADELTA, OpAdadelta, 2)
		PZ_CASE(PZ_OP_RMSPROP_GRAVES, OpRmspropGraves, 4)
		PZ_CASE(PZ_OP_SMORMS3, OpSmorms3, 2)
		PZ_CASE(PZ_OP_ADD3, OpAdd3, 0)
		PZ_CASE(PZ_OP_IADD, OpIadd, 0)
		PZ_CASE(PZ_OP_IMUL, OpImul, 0)
		PZ_CASE(PZ_OP_ADD3_RELU, OpAdd3Relu, 0)
		PZ_CASE(PZ_OP_ADD3_GATE, OpAdd3Gate, 0)

		case PZ_OP_DROPOUT2D: {
			PZ_REQUIRE(nptrs == 3 && nscalars >= 3 && dense, "pz_eltwise: dropout2d expects 3 operands, 3 scalars, no slice");
			uint32_t v, mapsize;
			memcpy(&v, &scalars[0], 4);
			memcpy(&mapsize, &scalars[2], 4);
			PZ_REQUIRE(mapsize > 0, "pz_eltwise: dropout2d mapsize must be positive");
			dropout2d_kernel<<<pz::stream_grid(count, 256), 256, 0, st>>>(a.p[0], a.p[1], (const uint32_t *)a.p[2], v,
			                                                              scalars[1], (int)mapsize, count);
			PZ_LAUNCH_CHECK();
			return PZ_OK;
		}
	}
#undef PZ_CASE
	PZ_REQUIRE(false, "pz_eltwise: unhandled op %d", op);
}

int pz_cast_i32_f32(float *out, const int32_t *in, size_t count, pz_stream_t stream) {
	if (count == 0) return PZ_OK;
	cast_i32_f32_kernel<<<pz::stream_grid(count, 256), 256, 0, pz::as_stream(stream)>>>(out, in, count);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_cast_f32_i32(int32_t *out, const float *in, size_t count, pz_stream_t stream) {
	if (count == 0) return PZ_OK;
	cast_f32_i32_kernel<<<pz::stream_grid(count, 256), 256, 0, pz::as_stream(stream)>>>(out, in, count);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_cast_f32_f16(uint16_t *out, const float *in, size_t count, pz_stream_t stream) {
	if (count == 0) return PZ_OK;
	cast_f32_f16_kernel<<<pz::stream_grid(count, 256), 256, 0, pz::as_stream(stream)>>>((_Float16 *)out, in, count);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

int pz_cast_f16_f32(float *out, const uint16_t *in, size_t count, pz_stream_t stream) {
	if (count == 0) return PZ_OK;
	cast_f16_f32_kernel<<<pz::stream_grid(count, 256), 256, 0, pz::as_stream(stream)>>>(out, (const _Float16 *)in, count);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

}  // extern "C"
