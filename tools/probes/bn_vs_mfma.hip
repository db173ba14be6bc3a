// Probe for DESIGN.md 3.1e: the library's bn_gate_stats_kernel<TWO> (through pz_bn_gate_stats, main stream) next to a
// PURE MFMA loop on a second stream. Modes of the neighbour: none, fp32 MFMA, bf16 MFMA with random operands, bf16 MFMA
// with all-zero operands (same instruction stream, little switching activity).
// Build: hipcc --offload-arch=gfx950 -O3 -o bn_vs_mfma bn_vs_mfma.hip -ldl ; run from the repo root: tools/probes/bn_vs_mfma
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(256) mfma_busy(float *out, int iters, int mode) {
	f32x16 acc[4];
	for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
	u16x8 a, b;
	unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1u;
	for (int e = 0; e < 8; ++e) {
		h = h * 1664525u + 1013904223u;
		a[e] = mode == 3 ? 0 : (unsigned short)(0x3c00u + ((h >> 9) & 0x3ffu) + ((h >> 31) << 15));
		h = h * 1664525u + 1013904223u;
		b[e] = mode == 3 ? 0 : (unsigned short)(0x3c00u + ((h >> 9) & 0x3ffu) + ((h >> 31) << 15));
	}
	float fa = 1.0f + threadIdx.x * 1e-3f, fb = 0.5f;
	for (int it = 0; it < iters; ++it)
		for (int i = 0; i < 4; ++i) {
			if (mode >= 2) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
			else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i], 0, 0, 0);
		}
	float s = 0.f;
	for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}

typedef int (*gate_stats_t)(const float *, const float *, const float *, const unsigned char *, float *, int, int, int, const float *,
                            const float *, float *, const float *, const float *, float *, void *);
typedef int (*ws_bytes_t)(int, int, int, size_t *);

int main() {
	void *lib = dlopen("puzzlelib_amd/libpuzzle_mi355.so", RTLD_NOW);
	if (!lib) { printf("dlopen: %s\n", dlerror()); return 1; }
	gate_stats_t gate_stats = (gate_stats_t)dlsym(lib, "pz_bn_gate_stats");
	ws_bytes_t ws_bytes = (ws_bytes_t)dlsym(lib, "pz_bn_workspace_bytes");
	const int n = 64, c = 1024, hw = 196;
	const size_t N = (size_t)n * c * hw;
	size_t wsb = 0;
	ws_bytes(n, c, hw, &wsb);
	float *g0, *g1, *y, *xa, *xb, *ma, *mb, *pa, *pb, *gout, *mout;
	hipMalloc(&g0, N * 4), hipMalloc(&g1, N * 4), hipMalloc(&y, N * 4), hipMalloc(&xa, N * 4), hipMalloc(&xb, N * 4), hipMalloc(&gout, N * 4);
	hipMalloc(&ma, c * 4), hipMalloc(&mb, c * 4), hipMalloc(&pa, wsb), hipMalloc(&pb, wsb), hipMalloc(&mout, 1024 * 256 * 4);
	std::vector<float> h(N);
	float *bufs[5] = {g0, g1, y, xa, xb};
	for (int k = 0; k < 5; ++k) {
		for (size_t i = 0; i < N; ++i) h[i] = (float)((((i + 77 * k) * 2654435761u) >> 7) & 0xffff) / 32768.f - 1.f;
		hipMemcpy(bufs[k], h.data(), N * 4, hipMemcpyHostToDevice);
	}
	hipMemcpy(ma, h.data(), c * 4, hipMemcpyHostToDevice), hipMemcpy(mb, h.data() + c, c * 4, hipMemcpyHostToDevice);
	hipMemset(pa, 0, wsb), hipMemset(pb, 0, wsb);
	hipStream_t s2;
	hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
	const size_t nf = wsb / 4;
	std::vector<float> ref(nf), got(nf);
	gate_stats(g0, g1, y, nullptr, gout, n, c, hw, xa, ma, pa, xb, mb, pb, nullptr);
	hipDeviceSynchronize();
	hipMemcpy(ref.data(), pa, wsb, hipMemcpyDeviceToHost);
	const char *names[] = {"idle chip", "fp32 MFMA 32x32x2 loop", "bf16 MFMA 32x32x16 loop, random operands", "bf16 MFMA 32x32x16 loop, zero operands"};
	for (int mode = 0; mode < 4; ++mode) {
		int bad_launches = 0, bad_floats = 0;
		for (int it = 0; it < 20; ++it) {
			if (mode) mfma_busy<<<768, 256, 0, s2>>>(mout, 30000, mode);
			gate_stats(g0, g1, y, nullptr, gout, n, c, hw, xa, ma, pa, xb, mb, pb, nullptr);
			hipDeviceSynchronize();
			hipMemcpy(got.data(), pa, wsb, hipMemcpyDeviceToHost);
			int d = 0;
			for (size_t i = 0; i < (size_t)c * 4 + (size_t)c * 16 * 2; ++i) d += memcmp(&got[i], &ref[i], 4) != 0;
			bad_launches += d != 0, bad_floats += d;
		}
		printf("%-46s launches with wrong partial sums: %2d of 20 (%d floats)\n", names[mode], bad_launches, bad_floats);
	}
	return 0;
}
