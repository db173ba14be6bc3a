// Does hipExtAnyOrderLaunch let the NEXT kernel of a stream start while the previous one drains (gfx950)?
// Two independent kernels A, B: every workgroup spins for `us` microseconds; 40 KB of LDS pin 4 workgroups per CU, the grid
// is 1.1 rounds of the chip (a full round + a 10 % tail). In-order: 2 x 2 workgroup-times. If B's workgroups are dispatched
// as soon as A has no more to dispatch: ~3 (A's tail and B's full round share the chip).
//   hipcc --offload-arch=gfx950 -O2 any_order.hip -o any_order && ./any_order
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

__global__ void __launch_bounds__(256) spin_kernel(float *out, long cycles) {
	__shared__ float pad[10 * 1024];
	pad[threadIdx.x] = (float)threadIdx.x;
	__syncthreads();
	const long t0 = wall_clock64();
	float acc = pad[(threadIdx.x * 7) & 1023];
	while (wall_clock64() - t0 < cycles) acc = acc * 1.0001f + 0.5f;
	if (acc == 123.456f) out[blockIdx.x] = acc;
}

int main() {
	float *out;
	hipMalloc(&out, 1 << 20);
	hipStream_t st;
	hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	const int grid = 256 * 4 + 100;
	const long cycles = 100 * 200;      // wall_clock64 runs at 100 MHz: 200 us per workgroup
	for (int mode = 0; mode < 3; ++mode) {
		for (int rep = 0; rep < 3; ++rep) {
			hipEventRecord(e0, st);
			for (int pair = 0; pair < 4; ++pair) {
				spin_kernel<<<grid, 256, 0, st>>>(out, cycles);
				if (mode == 0)
					spin_kernel<<<grid, 256, 0, st>>>(out, cycles);
				else
					hipExtLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), 0, st, nullptr, nullptr, mode == 1 ? hipExtAnyOrderLaunch : 0, out, cycles);
			}
			hipEventRecord(e1, st);
			hipEventSynchronize(e1);
			float ms;
			hipEventElapsedTime(&ms, e0, e1);
			printf("%s: 4 pairs %.3f ms (in-order expectation %.3f, tail-filled %.3f)\n",
			       mode == 0 ? "plain <<<>>>         " : (mode == 1 ? "hipExtAnyOrderLaunch" : "hipExtLaunch flags=0"), ms, 4 * 4 * 0.2, 4 * 3 * 0.2);
		}
	}
	return 0;
}
