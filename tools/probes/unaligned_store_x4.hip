// Probe: do 16-byte buffer/global STORES work at 4-byte alignment on gfx950? Lane t writes 4 floats at element 5*t + 1.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

__global__ void probe(float *buf_dst, float *glb_dst, unsigned bytes) {
	const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)buf_dst, 0, bytes, 0x00020000);
	const int t = threadIdx.x;
	const unsigned off = (5u * t + 1u) * 4u;
	f32x4 v = {1000.f + 4 * t, 1001.f + 4 * t, 1002.f + 4 * t, 1003.f + 4 * t};
	__builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, 0);
	*reinterpret_cast<f4u *>(reinterpret_cast<char *>(glb_dst) + off) = v;
}

int main() {
	const int n = 512;
	float h1[n], h2[n], *d1, *d2;
	hipMalloc(&d1, sizeof(h1)); hipMalloc(&d2, sizeof(h2));
	hipMemset(d1, 0, sizeof(h1)); hipMemset(d2, 0, sizeof(h2));
	probe<<<1, 64>>>(d1, d2, sizeof(h1));
	hipMemcpy(h1, d1, sizeof(h1), hipMemcpyDeviceToHost);
	hipMemcpy(h2, d2, sizeof(h2), hipMemcpyDeviceToHost);
	int bad1 = 0, bad2 = 0;
	for (int t = 0; t < 64; ++t)
		for (int e = 0; e < 4; ++e) {
			bad1 += h1[5 * t + 1 + e] != 1000.f + 4 * t + e;
			bad2 += h2[5 * t + 1 + e] != 1000.f + 4 * t + e;
		}
	printf("unaligned dwordx4 stores: buffer_store mismatches %d, global_store mismatches %d (h1[0..5] = %g %g %g %g %g %g)\n", bad1, bad2,
	       h1[0], h1[1], h1[2], h1[3], h1[4], h1[5]);
	return 0;
}
