// Probe: sustained fp32 MFMA rate and shader clock of the device under a pure v_mfma_f32_32x32x2_f32 loop
// (no memory traffic) — the practical ceiling the convolution kernels are compared with in DESIGN.md.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip ; run: ./mfma_peak [waves_per_simd] [ilp] [iters] [random_data]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ILP>
__global__ void __launch_bounds__(256) mfma_loop(float *out, int iters, long long *cycles, int random_data) {
	f32x16 acc[ILP];
	for (int i = 0; i < ILP; ++i)
		for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
	// 8 operand pairs of pseudo-random values per lane, cycled without any VALU work (register toggling like real data)
	float a[8], b[8];
	unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
	for (int k = 0; k < 8; ++k) {
		h = h * 1664525u + 1013904223u;
		a[k] = ((int)(h >> 8) - (1 << 23)) * (random_data ? 1.19e-7f : 0.f) + (random_data ? 0.f : 1e-3f);
		h = h * 1664525u + 1013904223u;
		b[k] = ((int)(h >> 8) - (1 << 23)) * (random_data ? 1.19e-7f : 0.f) + (random_data ? 0.f : 1e-3f);
	}
	const long long t0 = clock64();
	for (int it = 0; it < iters / 8; ++it) {
#pragma unroll
		for (int k = 0; k < 8; ++k)
#pragma unroll
			for (int i = 0; i < ILP; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], acc[i], 0, 0, 0);
	}
	const long long t1 = clock64();
	float s = 0.f;
	for (int i = 0; i < ILP; ++i)
		for (int r = 0; r < 16; ++r) s += acc[i][r];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

int main(int argc, char **argv) {
	const int rnd = argc > 4 ? atoi(argv[4]) : 1;
	const int wps = argc > 1 ? atoi(argv[1]) : 2, ilp = argc > 2 ? atoi(argv[2]) : 4;
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount, blocks = cus * wps;      // 256 threads = 4 waves = 1 wave per SIMD per block
	float *out;
	long long *cyc, hcyc = 0;
	hipMalloc(&out, (size_t)blocks * 256 * 4);
	hipMalloc(&cyc, 8);
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	const int iters = argc > 3 ? atoi(argv[3]) : 200000;
	for (int rep = 0; rep < 3; ++rep) {
		hipEventRecord(e0);
		if (ilp == 1) mfma_loop<1><<<blocks, 256>>>(out, iters, cyc, rnd);
		else if (ilp == 2) mfma_loop<2><<<blocks, 256>>>(out, iters, cyc, rnd);
		else mfma_loop<4><<<blocks, 256>>>(out, iters, cyc, rnd);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms = 0;
		hipEventElapsedTime(&ms, e0, e1);
		hipMemcpy(&hcyc, cyc, 8, hipMemcpyDeviceToHost);
		const double flop = (double)blocks * 4 * iters * (ilp == 1 ? 1 : ilp == 2 ? 2 : 4) * 32.0 * 32 * 2 * 2;
		// clock64 counts at a fixed 100 MHz reference on gfx9: derive the shader clock from MFMA issue instead
		const double mfma_per_simd = (double)wps * iters * (ilp == 1 ? 1 : ilp == 2 ? 2 : 4);
		printf("%d CUs, %d waves/SIMD, ilp %d: %.2f ms, %.1f TFLOP/s, implied shader clock %.3f GHz (64 cycles per MFMA per SIMD), s_memtime ticks %lld\n",
		       cus, wps, ilp, ms, flop / ms / 1e9, mfma_per_simd * 64 / (ms * 1e-3) / 1e9, hcyc);
	}
	return 0;
}
