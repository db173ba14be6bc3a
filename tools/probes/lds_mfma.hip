// Probe: MFMA rate of a wave tile fed from LDS (no global traffic): which fragment-read shape / barrier cadence /
// wave-tile size keeps v_mfma_f32_32x32x2_f32 closest to its 155 TFLOP/s pure-register rate.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_mfma lds_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS tile of a 32-deep k-step: [group][half][row][PER] floats; lane (row, h) reads the PER floats at [g][h][row] with one
// ds_read_b32 / b64 / b128 and uses them for PER consecutive k2-steps (PER = 1 is the [k][row] layout of the igemm kernel,
// PER = 2 the half-cell layout of the backward-filter kernel)
template <int TM, int TN, int PER, bool BAR, int SB, int MEM = 0, int BK = 32, int VALU = 0>
__global__ void __launch_bounds__(256) loop(float *out, int steps, const float *src = nullptr, unsigned plane = 0) {
	constexpr int ROWS_A = 2 * 32 * TM, ROWS_B = 2 * 32 * TN, G = BK / (2 * PER);
	constexpr int PAD = PER == 1 ? 0 : PER == 2 ? 2 : 1;
	typedef float vec __attribute__((ext_vector_type(PER)));
	__shared__ __attribute__((aligned(16))) float As[2][G][2][ROWS_A + PAD][PER];
	__shared__ __attribute__((aligned(16))) float Bs[2][G][2][ROWS_B + PAD][PER];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
	const int l31 = lane & 31, lhi = lane >> 5;
	for (int i = tid; i < (int)(sizeof(As) / 4); i += 256) ((float *)As)[i] = (i * 2654435761u >> 8) * 1e-9f;
	for (int i = tid; i < (int)(sizeof(Bs) / 4); i += 256) ((float *)Bs)[i] = (i * 40503u >> 8) * 1e-9f;
	__syncthreads();

	f32x16 acc[TM][TN];
	for (int i = 0; i < TM; ++i)
		for (int j = 0; j < TN; ++j)
			for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

	// MEM & 2: every thread fetches 8 x 16 B per k-step the way the backward-filter kernel does (thread = (run, row0): 8 lanes
	// cover 128 contiguous bytes of one channel plane, 4 + 4 planes 32 apart), MEM & 1: parks them in the other LDS buffer
	constexpr int NLD = (ROWS_A + ROWS_B) / 32;
	f32x4 ld[NLD];
	for (int i = 0; i < NLD; ++i) ld[i] = f32x4{1.f, 2.f, 3.f, 4.f};
	const int run = tid & 7, row0 = tid >> 3;
	// tensor = 1024 rows (channels) x 1 MB; block = (128-row tile, split of the long axis); a step advances 8 runs = 128 B
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 1u << 30, 0x00020000);
	const unsigned rowbase = ((blockIdx.x % 4) * 128u + row0) * plane;
	unsigned col = (blockIdx.x / 4) * 8192u + run * 16u;
	unsigned base = rowbase + col;

	int junk[4] = {lane, lane + 1, lane + 2, lane + 3};
	for (int s = 0; s < steps; ++s) {
		const int buf = s & 1;
		float av[2][TM][PER], bv[2][TN][PER];
		auto rd = [&](int g, int slot) {
#pragma unroll
			for (int i = 0; i < TM; ++i) {
				const vec v = *(const vec *)&As[buf][g][lhi][wm * 32 * TM + i * 32 + l31][0];
#pragma unroll
				for (int p = 0; p < PER; ++p) av[slot][i][p] = PER == 1 ? ((const float *)&v)[0] : ((const float *)&v)[p];
			}
#pragma unroll
			for (int j = 0; j < TN; ++j) {
				const vec v = *(const vec *)&Bs[buf][g][lhi][wn * 32 * TN + j * 32 + l31][0];
#pragma unroll
				for (int p = 0; p < PER; ++p) bv[slot][j][p] = ((const float *)&v)[p];
			}
		};
		rd(0, 0);
#pragma unroll
		for (int g = 0; g < G; ++g) {
			if (g + 1 < G) rd(g + 1, (g + 1) & 1);
			if ((MEM & 2) && g * NLD / G < NLD) {
#pragma unroll
				for (int q = g * NLD / G; q < (g + 1) * NLD / G; ++q)
					ld[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, base, (unsigned)(32 * q) * plane, 0));
			}
#pragma unroll
			for (int q = 0; q < VALU; ++q) asm volatile("v_add_u32 %0, %0, %1" : "+v"(junk[q & 3]) : "v"(lane));      // VALU per group
			if (SB) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int p = 0; p < PER; ++p)
#pragma unroll
				for (int i = 0; i < TM; ++i)
#pragma unroll
					for (int j = 0; j < TN; ++j)
						acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][i][p], bv[g & 1][j][p], acc[i][j], 0, 0, 0);
			if (SB) __builtin_amdgcn_sched_barrier(0);
		}
		if (MEM & 2) col = (col + 128u) & (plane - 1u), base = rowbase + col;       // next 8 runs of the same planes
		if (MEM & 1) {
#pragma unroll
			for (int q = 0; q < NLD; ++q) {
				float *dst = q < ROWS_A / 32 ? &As[buf ^ 1][0][0][0][0] : &Bs[buf ^ 1][0][0][0][0];
				const int r = row0 + 32 * (q < ROWS_A / 32 ? q : q - ROWS_A / 32);
				const int rows = (q < ROWS_A / 32 ? ROWS_A : ROWS_B) + PAD;
				// run -> group/half placement for PER == 2 ([run][half][row][2]); other PER values park the same bytes linearly
				*(f32x2 *)&dst[((run * 2 + 0) * rows + r) * 2] = f32x2{ld[q][0], ld[q][1]};
				*(f32x2 *)&dst[((run * 2 + 1) * rows + r) * 2] = f32x2{ld[q][2], ld[q][3]};
			}
		}
		if (BAR) __syncthreads();
	}
	if (MEM & 2) for (int q = 0; q < NLD; ++q) acc[0][0][q % 16] += ld[q][0];
	if (VALU) acc[0][0][0] += (float)(junk[0] + junk[1] + junk[2] + junk[3]);
	float sum = 0.f;
	for (int i = 0; i < TM; ++i)
		for (int j = 0; j < TN; ++j)
			for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
	out[blockIdx.x * 256 + tid] = sum;
}

template <int TM, int TN, int W, bool BAR, int SB, int MEM = 0, int BK = 32, int VALU = 0>
void run(const char *name, int per_cu) {
	const int blocks = 256 * per_cu, steps = 4000;
	float *out, *src = nullptr;
	hipMalloc(&out, (size_t)blocks * 256 * 4);
	const unsigned plane = 1u << 20;                         // 1 MB rows, 1024 rows = 1 GB (rows 32*q apart: 4 + 4 of them <= 128+3*32 rows)
	if (MEM & 2) { hipMalloc(&src, (size_t)1 << 30); hipMemset(src, 0, (size_t)1 << 30); }
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	float best = 1e9;
	for (int rep = 0; rep < 3; ++rep) {
		hipEventRecord(e0);
		loop<TM, TN, W, BAR, SB, MEM, BK, VALU><<<blocks, 256>>>(out, steps, src, plane);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		best = ms < best ? ms : best;
	}
	const double flop = (double)blocks * steps * 2.0 * (64 * TM) * (64 * TN) * BK;
	printf("%-44s %d blocks/CU: %7.2f ms  %6.1f TFLOP/s  (%s)\n", name, per_cu, best, flop / best / 1e9, hipGetErrorString(hipGetLastError()));
	hipFree(out);
	if (src) hipFree(src);
}

int main() {
	run<2, 2, 1, true, 1>("64x64 wave tile, b32 reads, barrier/step", 2);
	run<2, 2, 2, true, 1>("64x64 wave tile, b64 reads, barrier/step", 2);
	run<2, 2, 4, true, 1>("64x64 wave tile, b128 reads, barrier/step", 2);
	run<2, 2, 2, false, 1>("64x64, b64, no barrier", 2);
	run<2, 2, 2, true, 0>("64x64, b64, barrier, compiler-scheduled", 2);
	run<2, 2, 2, true, 1>("64x64, b64, barrier", 1);
	run<2, 2, 2, true, 1>("64x64, b64, barrier", 3);
	run<2, 4, 2, true, 1>("64x128 wave tile (128x256 block), b64", 1);
	run<4, 2, 2, true, 1>("128x64 wave tile (256x128 block), b64", 1);
	run<2, 4, 2, false, 1>("64x128 wave tile, b64, no barrier", 1);
	run<2, 4, 4, true, 1>("64x128 wave tile, b128", 1);
	run<2, 4, 1, true, 1>("64x128 wave tile, b32", 1);
	run<2, 2, 1, true, 1, 0, 16>("64x64 b32, BK=16 barrier/step", 2);
	run<2, 2, 1, true, 1, 0, 16>("64x64 b32, BK=16 barrier/step", 4);
	run<2, 2, 1, true, 1, 0, 32>("64x64 b32, BK=32 barrier/step", 4);
	run<2, 2, 2, true, 1, 0, 16>("64x64 b64, BK=16 barrier/step", 4);
	run<2, 2, 1, true, 1, 0, 24>("64x64 b32, BK=24 barrier/step", 3);
	run<2, 2, 1, true, 1, 0, 16>("64x64 b32, BK=16 barrier/step", 3);
	run<2, 2, 1, true, 1, 0, 24>("64x64 b32, BK=24 barrier/step", 4);
	run<2, 2, 1, false, 1, 0, 16>("64x64 b32, BK=16 NO barrier", 4);
	run<2, 2, 2, true, 1, 0, 32, 4>("64x64 b64 + 4 VALU per 8 MFMA", 2);
	run<2, 2, 2, true, 1, 0, 32, 8>("64x64 b64 + 8 VALU per 8 MFMA", 2);
	run<2, 2, 2, true, 1, 0, 32, 16>("64x64 b64 + 16 VALU per 8 MFMA", 2);
	run<2, 2, 2, true, 1, 0, 32, 32>("64x64 b64 + 32 VALU per 8 MFMA", 2);
	run<2, 2, 2, true, 1, 1>("64x64 b64 + 16 ds_write_b64 per step", 2);
	run<2, 2, 2, true, 1, 2>("64x64 b64 + 8 global b128 loads per step", 2);
	run<2, 2, 2, true, 1, 3>("64x64 b64 + loads + LDS stores", 2);
	return 0;
}
