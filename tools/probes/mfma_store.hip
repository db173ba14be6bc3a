// Probe (round 4): does the epilogue's HBM write stream hide behind other workgroups' MFMAs? A workgroup of 4 waves issues
// `nmfma` v_mfma_f32_32x32x2_f32 per wave out of registers (no loads, no LDS: the k-loop of a 128x128 tile of reduction
// length nmfma/2) and then stores its 64 KB of accumulators with the production epilogue's instruction stream (16 stores of
// 16 B per lane and wave: 8 channel rows x 128 B each). Patterns:
//   3  PQ = 3028 (rows 16-byte aligned, 128-byte runs straddle lines) — 4  PQ = 3025 with every lane's address rounded down to 16 B
//   5 / 6  PQ = 3025 / 3028 with 4 rows x 256 B per store instruction — 7 / 8  PQ = 3056 / 3032: rows 64- / 32-byte aligned
//   0  the implicit GEMM's: 128 channel rows x 512 B at the plane stride of a (256, OC, PQ) tensor, PQ = 3025 (55x55: rows
//      4-byte aligned) — 1  the same with PQ = 3072 (rows 16-byte... 128-byte aligned) — 2  64 KB contiguous per workgroup
// Printed: ms for MFMAs alone, stores alone, both; the sum and the max are what no overlap / full overlap would give.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_store mfma_store.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)

struct Args {
	float *y;
	unsigned y_bytes;
	int nmfma, store, pat, PQ, OC, npix, mtiles, load, remap, blocks, prio;
};

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) probe(Args a) {
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	const int wm = wave >> 1, wn = wave & 1;
	f32x16 acc[2][2];
#pragma unroll
	for (int i = 0; i < 2; ++i)
#pragma unroll
		for (int j = 0; j < 2; ++j)
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[i][j][r] = (float)(lane + r);
	float fa[2] = {1.f + lane * 1e-3f, 2.f - lane * 1e-3f}, fb[2] = {0.5f + lane * 1e-3f, 0.25f};

	const int n_it = a.nmfma / 4;
	// prio 2: the first generation of workgroups (4 per CU) starts a quarter of a workgroup's life apart: slot = blockIdx / 256
	if (a.prio == 2 && blockIdx.x < 1024) {
		const long long until = __builtin_readcyclecounter() + (long long)(blockIdx.x >> 8) * a.nmfma * 64;      // 4 waves x nmfma x 64 cycles / 4
		while (__builtin_readcyclecounter() < until) __builtin_amdgcn_s_sleep(8);
	}
	for (int it = 0; it < n_it; ++it) {
		// prio 1: a wave raises its own priority as it advances (quarters of the loop): whoever is ahead gets the matrix pipe
		// first, the resident workgroups stop finishing — and draining their stores — all at the same time
		if (a.prio == 1) {
			if (it == n_it / 4) __builtin_amdgcn_s_setprio(1);
			if (it == n_it / 2) __builtin_amdgcn_s_setprio(2);
			if (it == 3 * n_it / 4) __builtin_amdgcn_s_setprio(3);
		}
#pragma unroll
		for (int i = 0; i < 2; ++i)
#pragma unroll
			for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
	}

	if (!a.store) {
		if (acc[0][0][0] + acc[1][1][15] + acc[0][1][3] + acc[1][0][7] == 123.456f) a.y[tid] = 1.f;      // keeps the MFMAs alive
		return;
	}
	const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void *)a.y, 0, a.y_bytes, 0x00020000);
	const int rr = lane >> 3, c4 = lane & 7;
	float lsum = 0.f;
	// remap: the 8 XCDs take consecutive workgroup ids in turn; give each XCD a contiguous eighth of the tiles instead
	const int bid = a.remap ? (int)(blockIdx.x & 7) * (a.blocks >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
	const int tm = bid % a.mtiles, tn = bid / a.mtiles;
#pragma unroll
	for (int i = 0; i < 2; ++i)
#pragma unroll
		for (int j = 0; j < 2; ++j)
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				f32x4 v = {acc[i][j][4 * k], acc[i][j][4 * k + 1], acc[i][j][4 * k + 2], acc[i][j][4 * k + 3]};
				unsigned off;
				if (a.pat == 2) {
					off = (unsigned)bid * 65536u + (unsigned)wave * 16384u + (unsigned)((i * 2 + j) * 4 + k) * 1024u + lane * 16u;
				} else {
					const bool wide = a.pat == 5 || a.pat == 6;      // 4 rows x 256 B per instruction instead of 8 rows x 128 B
					const int ch = tm * 128 + wm * 64 + i * 32 + (wide ? (j * 4 + k) * 4 + (lane >> 4) : rr + 8 * k);
					const int opix = tn * 128 + wn * 64 + (wide ? (lane & 15) * 4 : j * 32 + c4 * 4);
					const int n_img = opix / a.PQ, pq = opix - n_img * a.PQ;
					off = opix < a.npix ? (((unsigned)n_img * a.OC + ch) * (unsigned)a.PQ + pq) * 4u : 0xffffffffu;
					if (a.pat == 4 && opix < a.npix) off &= ~15u;      // the traffic of 16-byte groups cut at aligned addresses
				}
				if (a.load) {
					const f32x4 w = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yr, off, 0, 0));
					lsum += w[0] + w[1] + w[2] + w[3];
				} else {
					__builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yr, off, 0, 0);
				}
			}
	if (a.load && lsum == 123.456f) a.y[tid] = 1.f;
}

static float run(Args a, int blocks, int reps) {
	hipEvent_t s, e;
	CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
	for (int i = 0; i < 3; ++i) probe<<<blocks, 256>>>(a);
	CK(hipEventRecord(s));
	for (int i = 0; i < reps; ++i) probe<<<blocks, 256>>>(a);
	CK(hipEventRecord(e)); CK(hipEventSynchronize(e));
	float ms; CK(hipEventElapsedTime(&ms, s, e));
	return ms / reps;
}

int main() {
	const int N = 256, OC = 256;
	const size_t cap = (size_t)N * OC * 3072 * 4;      // 805 MB
	float *y; CK(hipMalloc(&y, cap)); CK(hipMemset(y, 0, cap));
	printf("%-10s %6s | %8s %8s %8s | %8s %8s  (ms; %d x %d x PQ tensor, 128x128 tiles)\n", "pattern", "nmfma", "mfma", "store", "both", "sum", "max", N, OC);
	const char *names[9] = {"pq3025", "pq3072", "linear", "pq3028", "pq3025cut", "pq3025wide", "pq3028wide", "pq3056", "pq3032"};
	const int pqs[9] = {3025, 3072, 3072, 3028, 3025, 3025, 3028, 3056, 3032};
	for (int remap = 1; remap < 2; ++remap)
	for (int pat = 0; pat < 9; ++pat) {
		const int PQ = pqs[pat];
		const int npix = N * PQ;
		const int ntiles = (npix + 127) / 128, mtiles = OC / 128;
		const int blocks = ntiles * mtiles;
		for (int nm : {128, 512}) {
			Args a = {y, (unsigned)((size_t)N * OC * PQ * 4), nm, 0, pat, PQ, OC, npix, mtiles, 0, remap, blocks & ~7, 0};
			const float tm = run(a, blocks, 20);
			a.store = 1; a.nmfma = 0;
			const float ts = run(a, blocks, 20);
			a.nmfma = nm;
			const float tb = run(a, blocks, 20);
			a.prio = 1;
			const float tp = run(a, blocks, 20);
			a.prio = 2;
			const float tq = run(a, blocks, 20);
			a.prio = 0;
			a.nmfma = 0; a.load = 1;
			const float tl = run(a, blocks, 20);
			printf("%-10s%s %6d | %8.3f %8.3f %8.3f | %8.3f %8.3f   %.0f MB   loads alone %.3f   both with progress priority %.3f, staggered start %.3f\n", names[pat], remap ? "/x" : "  ", nm, tm, ts, tb, tm + ts, tm > ts ? tm : ts, blocks * 65536e-6, tl, tp, tq);
		}
	}
	return 0;
}
