// Probe for DESIGN.md 3.1e: which packed-fp32 instruction form goes wrong while another wave on the SIMD runs
// v_mfma_f32_32x32x16_bf16? One victim kernel per form: every thread applies the packed instruction to fresh pseudo-random
// operands in a loop and compares both lanes with the scalar instructions bit for bit; the neighbour is a pure MFMA loop
// on a second stream. Build: hipcc --offload-arch=gfx950 -O3 -o pk_forms_probe pk_forms_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float rnd(unsigned &h) {
	h = h * 1664525u + 1013904223u;
	return __builtin_bit_cast(float, 0x3f800000u | (h >> 9)) - 1.5f;
}
__device__ __forceinline__ bool ne(float a, float b) { return __builtin_bit_cast(unsigned, a) != __builtin_bit_cast(unsigned, b); }

template <int FORM>
__global__ void __launch_bounds__(256) victim(unsigned *bad, float *out, int iters) {
	unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 99u;
	unsigned mism = 0;
	float sink = 0.f;
	for (int it = 0; it < iters; ++it) {
		f32x2 a = {rnd(h), rnd(h)}, b = {rnd(h), rnd(h)}, c = {rnd(h), rnd(h)}, r;
		float e0, e1;
		if (FORM == 0) {          // v_pk_add_f32 a, b
			asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
			asm volatile("v_add_f32 %0, %1, %2" : "=v"(e0) : "v"(a[0]), "v"(b[0]));
			asm volatile("v_add_f32 %0, %1, %2" : "=v"(e1) : "v"(a[1]), "v"(b[1]));
		} else if (FORM == 1) {   // a - b via neg modifiers
			asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
			asm volatile("v_sub_f32 %0, %1, %2" : "=v"(e0) : "v"(a[0]), "v"(b[0]));
			asm volatile("v_sub_f32 %0, %1, %2" : "=v"(e1) : "v"(a[1]), "v"(b[1]));
		} else if (FORM == 2) {   // v_pk_mul_f32
			asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
			asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e0) : "v"(a[0]), "v"(b[0]));
			asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e1) : "v"(a[1]), "v"(b[1]));
		} else if (FORM == 3) {   // v_pk_mul_f32 op_sel:[0,1]: lo lane = a.lo * b.hi, hi lane = a.hi * b.hi
			asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "v"(b));
			asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e0) : "v"(a[0]), "v"(b[1]));
			asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e1) : "v"(a[1]), "v"(b[1]));
		} else if (FORM == 4) {   // v_pk_fma_f32 op_sel_hi:[0,1,1]: src0 lo broadcast
			asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
			asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e0) : "v"(a[0]), "v"(b[0]), "v"(c[0]));
			asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e1) : "v"(a[0]), "v"(b[1]), "v"(c[1]));
		} else if (FORM == 5) {   // v_pk_fma_f32 plain
			asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
			asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e0) : "v"(a[0]), "v"(b[0]), "v"(c[0]));
			asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e1) : "v"(a[1]), "v"(b[1]), "v"(c[1]));
		} else {                  // v_pk_add_f32 op_sel:[0,1] op_sel_hi:[0,1]: horizontal add a.lo + b.hi | a.lo... (as in the BN kernel)
			asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
			asm volatile("v_add_f32 %0, %1, %2" : "=v"(e0) : "v"(a[0]), "v"(b[1]));
			asm volatile("v_add_f32 %0, %1, %2" : "=v"(e1) : "v"(a[0]), "v"(b[1]));
		}
		mism += ne(r[0], e0) ? 1u : 0u;
		mism += ne(r[1], e1) ? 0x10000u : 0u;
		sink += r[0] + r[1];
	}
	if (mism) atomicAdd(bad, 1u), atomicAdd(bad + 1, mism & 0xffffu), atomicAdd(bad + 2, mism >> 16);
	out[blockIdx.x * 256 + threadIdx.x] = sink;
}

__global__ void __launch_bounds__(256) mfma_busy(float *out, int iters, int bf16) {
	f32x16 acc[4];
	for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
	u16x8 a, b;
	for (int e = 0; e < 8; ++e) a[e] = (unsigned short)(0x3c00 + threadIdx.x + e), b[e] = (unsigned short)(0x3c10 + e);
	float fa = 1.0f + threadIdx.x * 1e-3f, fb = 0.5f;
	for (int it = 0; it < iters; ++it)
		for (int i = 0; i < 4; ++i) {
			if (bf16) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
			else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i], 0, 0, 0);
		}
	float s = 0.f;
	for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int FORM>
void run(const char *what, hipStream_t s2, unsigned *bad, float *out, float *mout) {
	for (int mode = 0; mode < 3; ++mode) {
		hipMemset(bad, 0, 12);
		for (int it = 0; it < 5; ++it) {
			if (mode) mfma_busy<<<768, 256, 0, s2>>>(mout, 30000, mode == 2);
			victim<FORM><<<2048, 256>>>(bad, out, 2000);
			hipDeviceSynchronize();
		}
		unsigned hb[3];
		hipMemcpy(hb, bad, 12, hipMemcpyDeviceToHost);
		printf("%-52s %-10s threads %7u  lo-lane mismatches %9u  hi-lane %9u\n", what, mode == 0 ? "idle" : mode == 1 ? "fp32 MFMA" : "bf16 MFMA", hb[0], hb[1], hb[2]);
	}
}

int main() {
	unsigned *bad;
	float *out, *mout;
	hipMalloc(&bad, 12), hipMalloc(&out, 2048 * 256 * 4), hipMalloc(&mout, 1024 * 256 * 4);
	hipStream_t s2;
	hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
	run<0>("v_pk_add_f32", s2, bad, out, mout);
	run<1>("v_pk_add_f32 neg_lo:[0,1] neg_hi:[0,1]", s2, bad, out, mout);
	run<2>("v_pk_mul_f32", s2, bad, out, mout);
	run<3>("v_pk_mul_f32 op_sel:[0,1]", s2, bad, out, mout);
	run<4>("v_pk_fma_f32 op_sel_hi:[0,1,1]", s2, bad, out, mout);
	run<5>("v_pk_fma_f32", s2, bad, out, mout);
	run<6>("v_pk_add_f32 op_sel:[0,1] op_sel_hi:[0,1]", s2, bad, out, mout);
	return 0;
}
