// Probe: do 16-byte buffer/global loads work at 4-byte alignment on gfx950? (needed for pixel-run loads whose rows are
// odd-sized, e.g. 55x55 or 7x7 feature maps). Prints mismatches; 0 means the hardware handles dword-aligned dwordx4.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float *src, float *out_buf, float *out_glb, unsigned bytes) {
	const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, bytes, 0x00020000);
	const int t = threadIdx.x;                 // lane t reads 4 floats starting at element 5*t + 1 (never 16-B aligned in general)
	const unsigned off = (5u * t + 1u) * 4u;
	f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
	f32x4 w = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(src) + off);
	for (int e = 0; e < 4; ++e) {
		out_buf[4 * t + e] = v[e];
		out_glb[4 * t + e] = w[e];
	}
}

int main() {
	const int n = 1024;
	float h[n], *d, *o1, *o2, r1[256], r2[256];
	for (int i = 0; i < n; ++i) h[i] = (float)i;
	hipMalloc(&d, sizeof(h)); hipMalloc(&o1, sizeof(r1)); hipMalloc(&o2, sizeof(r2));
	hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
	probe<<<1, 64>>>(d, o1, o2, sizeof(h));
	hipMemcpy(r1, o1, sizeof(r1), hipMemcpyDeviceToHost);
	hipMemcpy(r2, o2, sizeof(r2), hipMemcpyDeviceToHost);
	int bad1 = 0, bad2 = 0;
	for (int t = 0; t < 64; ++t)
		for (int e = 0; e < 4; ++e) {
			bad1 += r1[4 * t + e] != (float)(5 * t + 1 + e);
			bad2 += r2[4 * t + e] != (float)(5 * t + 1 + e);
		}
	printf("unaligned dwordx4: buffer_load mismatches %d, global_load mismatches %d (first: %g %g)\n", bad1, bad2, r1[0], r2[0]);
	return 0;
}
