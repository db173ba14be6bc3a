// Probe: the implicit GEMM's k-loop (128x128 tile, BK = 16, 4 waves, LDS double buffer, one barrier per k-tile, 4
// workgroups per CU) on synthetic operands, to separate what the loop costs from what a finite launch costs.
//   MODE 0  no staging at all (MFMAs + fragment reads + barrier): the loop's ceiling
//   MODE 1  the production scheme: 2 x 16-byte filter loads + 8 x 4-byte pixel gathers into registers, parked in the
//           other LDS buffer with ds_write_b128 / ds_write_b32 at the end of the k-tile
//   MODE 2  the same loads as LDS-direct buffer loads (buffer_load_dword[x4] ... lds: the data never passes the vector
//           registers, M0 carries the wave's LDS base, lane l lands at base + l * size)
//   MODE 3  LDS-direct into a three-stage ring, loads issued two k-tiles ahead (vmcnt(10) at the end of a k-tile)
//   MODE 4  MODE 1 with a pseudo-random start delay per workgroup (co-resident workgroups out of phase)
//   MODE 5  MODE 1 with tile indices drawn from per-XCD counters, stolen from the neighbours when the own range is
//           empty, and 1/8 surplus workgroups that exit at once (dynamic balance across XCDs without a persistent loop)
// It checks that MODES 2 / 3 compute what MODE 1 computes (layout of the direct loads, zero fill of out-of-range lanes),
// prints a per-workgroup timeline (start / end stamps, XCC) of single launches, and times 10 back-to-back launches per
// grid size. Findings (MI355X, profiles/r02_lds_dma_probe.txt): DESIGN.md section 3.1f.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_dma lds_dma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))

constexpr int BM = 128, BN = 128, BK = 16;
constexpr unsigned kOOB = 0x80000000u;
__device__ unsigned long long *g_stamps = nullptr;
__device__ int g_ctr[8];          // MODE 5: tiles handed out per XCD (range x = tiles [x*per, (x+1)*per)), stolen from the neighbours when the own range is empty
__device__ int g_ntiles;       // optional [block][2] start / end timestamps (100 MHz counter) + hardware id

// src: `planes` channel planes of `plane` bytes each (pixels contiguous); filt: [k][128] floats, k = 0..kfilt-1 (wraps)
template <int MODE, bool OOBTEST>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8)))
loop(float *out, int ktiles, const float *src, unsigned plane, unsigned planes, const float *filt, unsigned kfilt) {
	constexpr int NS = MODE == 3 ? 3 : 2;             // LDS stages
	__shared__ __attribute__((aligned(16))) float smem[NS * BK * (BM + BN)];
	float(*As)[BK][BM] = reinterpret_cast<float(*)[BK][BM]>(smem);
	float(*Bs)[BK][BN] = reinterpret_cast<float(*)[BK][BN]>(smem + NS * BK * BM);
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lhi = lane >> 5;
	const unsigned long long t_start = __builtin_readcyclecounter();
	const unsigned long long rt_start = wall_clock64();
	unsigned tile = blockIdx.x;
	if (MODE == 5) {
		__shared__ int sh_tile;
		if (tid == 0) {
			unsigned xcc;
			asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
			xcc &= 7;
			const int per = (g_ntiles + 7) / 8;
			int got = -1;
			for (int i = 0; i < 8 && got < 0; ++i) {
				const int r = (xcc + i) & 7;
				const int cnt = min(per, g_ntiles - r * per);
				if (cnt <= 0) continue;
				if (__hip_atomic_load(&g_ctr[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= cnt) continue;
				const int t = atomicAdd(&g_ctr[r], 1);
				if (t < cnt) got = r * per + t;
			}
			sh_tile = got;
		}
		__syncthreads();
		if (sh_tile < 0) return;
		tile = (unsigned)sh_tile;
	}

	const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, plane * planes, 0x00020000);
	const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *)filt, 0, kfilt * BM * 4, 0x00020000);

	// B loader: thread = pixel column jb, k rows kb0*8 .. kb0*8+7 (a wave = 64 consecutive pixels of one row)
	const int jb = tid % BN, kb0 = __builtin_amdgcn_readfirstlane(tid / BN);
	unsigned voffB = ((tile % 2048u) * BN + jb) * 4u;
	if (OOBTEST && (jb % 7) == 3) voffB = kOOB;                     // some lanes out of range: must read as 0
	// A loader: 16 B per thread, f = tid + i*256 -> k row f/32, m4 = (f%32)*4: linear in f
	unsigned voffA[2];
	for (int i = 0; i < 2; ++i) voffA[i] = (unsigned)(tid + i * 256) * 16u;

	f32x16 acc[2][2];
	for (int i = 0; i < 2; ++i)
		for (int j = 0; j < 2; ++j)
			for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

	f32x4 ra[2];
	float rb[8];

	auto load_part = [&](int kt, int buf, int j) {
		const unsigned soffA = ((unsigned)(kt * BK) % kfilt) * BM * 4u;
		const unsigned soffB = (((unsigned)(kt * BK) % planes) + kb0 * 8 + j) * plane;
		if (MODE == 1 || MODE == 4 || MODE == 5) {
			if (j < 2) ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, voffA[j], soffA, 0));
			rb[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, voffB, soffB, 0));
		} else if (MODE >= 2) {
			if (j < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, LDS_PTR(&As[buf][0][0] + (wave * 64 + j * 256) * 4), 16, voffA[j], soffA, 0, 0);
			__builtin_amdgcn_raw_ptr_buffer_load_lds(xr, LDS_PTR(&Bs[buf][kb0 * 8 + j][(wave & 1) * 64]), 4, voffB, soffB, 0, 0);
		}
	};
	auto store_tile = [&](int buf) {
		if (MODE == 1 || MODE == 4 || MODE == 5) {
			for (int i = 0; i < 2; ++i) {
				const int f = tid + i * 256;
				*reinterpret_cast<f32x4 *>(&As[buf][f / 32][(f % 32) * 4]) = ra[i];
			}
			for (int i = 0; i < 8; ++i) Bs[buf][kb0 * 8 + i][jb] = rb[i];
		} else if (MODE == 2) {
			__builtin_amdgcn_s_waitcnt(0x0f70);       // vmcnt(0): the direct loads have landed in LDS
		}
	};
	auto read_frag = [&](int buf, int ks, float (&av)[2], float (&bv)[2]) {
		for (int i = 0; i < 2; ++i) av[i] = As[buf][ks + lhi][wm * 64 + i * 32 + l31];
		for (int j = 0; j < 2; ++j) bv[j] = Bs[buf][ks + lhi][wn * 64 + j * 32 + l31];
	};
	auto compute_tile = [&](int buf, int kt_next, bool has_next) {
		float av[2][2], bv[2][2];
		read_frag(buf, 0, av[0], bv[0]);
#pragma unroll
		for (int j = 0; j < BK / 2; ++j) {
			if (j + 1 < BK / 2) read_frag(buf, 2 * (j + 1), av[(j + 1) & 1], bv[(j + 1) & 1]);
			if (has_next && MODE != 0) load_part(kt_next, MODE == 3 ? (buf == 0 ? 2 : buf - 1) : buf ^ 1, j);      // MODE 3: stage of tile kt+2 = (buf + 2) % 3
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int i = 0; i < 2; ++i)
#pragma unroll
				for (int jj = 0; jj < 2; ++jj)
					acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j & 1][i], bv[j & 1][jj], acc[i][jj], 0, 0, 0);
			__builtin_amdgcn_sched_barrier(0);
		}
	};

	if (MODE == 4) {           // staggered start: a pseudo-random delay of up to ~one k-tile so that co-resident workgroups run out of phase
		const int n = (int)((blockIdx.x * 2654435761u) >> 28);
		for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(6);
	}
	if (MODE == 3) {
		// three-stage ring, loads two k-tiles ahead: at the end of tile kt only the loads of tile kt+1 (issued a whole
		// tile earlier) have to have landed -> vmcnt(10) leaves the 10 loads of tile kt+2 in flight
#pragma unroll
		for (int j = 0; j < BK / 2; ++j) load_part(0, 0, j);
		if (ktiles > 1) {
#pragma unroll
			for (int j = 0; j < BK / 2; ++j) load_part(1, 1, j);
			__builtin_amdgcn_s_waitcnt(0x0f7a);      // vmcnt(10)
		} else {
			__builtin_amdgcn_s_waitcnt(0x0f70);
		}
		__syncthreads();
		int stage = 0;
		for (int kt = 0; kt + 1 < ktiles; ++kt) {
			const bool more = kt + 2 < ktiles;
			compute_tile(stage, kt + 2, more);
			if (more) __builtin_amdgcn_s_waitcnt(0x0f7a); else __builtin_amdgcn_s_waitcnt(0x0f70);
			__syncthreads();
			stage = stage == 2 ? 0 : stage + 1;
		}
		compute_tile(stage, 0, false);
	} else {
	if (MODE == 0) {
		for (int i = tid; i < 2 * BK * (BM + BN); i += 256) smem[i] = (i * 2654435761u >> 8) * 1e-9f;
	} else {
#pragma unroll
		for (int j = 0; j < BK / 2; ++j) load_part(0, 0, j);
		store_tile(0);
	}
	__syncthreads();
	for (int kt = 0; kt + 1 < ktiles; ++kt) {
		const int buf = kt & 1;
		compute_tile(buf, kt + 1, true);
		store_tile(buf ^ 1);
		__syncthreads();
	}
	compute_tile((ktiles - 1) & 1, 0, false);
	}

	if (g_stamps && tid == 0) {
		unsigned hwid;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
		unsigned xcc;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
		g_stamps[blockIdx.x * 4 + 0] = rt_start;
		g_stamps[blockIdx.x * 4 + 1] = wall_clock64();
		g_stamps[blockIdx.x * 4 + 2] = ((unsigned long long)xcc << 32) | hwid;
		g_stamps[blockIdx.x * 4 + 3] = __builtin_readcyclecounter() - t_start;
	}
	float *o = out + (size_t)tile * BM * BN;
	for (int i = 0; i < 2; ++i)
		for (int j = 0; j < 2; ++j)
			for (int r = 0; r < 16; ++r) o[((i * 2 + j) * 16 + r) * 256 + tid] = acc[i][j][r];
}

__global__ void fill(float *p, size_t n, unsigned seed) {
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
		p[i] = (float)(((unsigned)i * 2654435761u + seed) >> 20) * (1.f / 4096.f) - 0.5f;
}

template <int MODE, bool OOBTEST>
float run(const char *name, int blocks, int ktiles, float *out, const float *src, const float *filt, bool print = true, int extra_lds = 0) {
	const unsigned plane = 1u << 20, planes = 1024, kfilt = 4096;
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	float best = 1e9;
	for (int rep = 0; rep < 3; ++rep) {
		hipEventRecord(e0);
		loop<MODE, OOBTEST><<<blocks, 256, extra_lds>>>(out, ktiles, src, plane, planes, filt, kfilt);      // extra LDS = fewer resident workgroups
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		best = ms < best ? ms : best;
	}
	const double flop = (double)blocks * ktiles * 2.0 * BM * BN * BK;
	if (print) printf("%-58s %5d blocks x %4d k-tiles: %8.3f ms  %6.1f TFLOP/s  (%s)\n", name, blocks, ktiles, best, flop / best / 1e9,
	                  hipGetErrorString(hipGetLastError()));
	return best;
}

int main() {
	float *src, *filt, *out;
	hipMalloc(&src, (size_t)1 << 30);
	hipMalloc(&filt, (size_t)4096 * BM * 4);
	const int blocks = 12288;
	hipMalloc(&out, (size_t)blocks * BM * BN * 4);
	fill<<<4096, 256>>>(src, (size_t)1 << 28, 1u);
	fill<<<256, 256>>>(filt, (size_t)4096 * BM, 7u);
	hipDeviceSynchronize();

	// correctness: MODE 2 == MODE 1, with and without out-of-range lanes
	for (int oob = 0; oob < 2; ++oob) {
		std::vector<float> a((size_t)64 * BM * BN), b(a.size());
		if (oob) run<1, true>("", 64, 8, out, src, filt, false); else run<1, false>("", 64, 8, out, src, filt, false);
		hipMemcpy(a.data(), out, a.size() * 4, hipMemcpyDeviceToHost);
		hipMemset(out, 0, a.size() * 4);
		if (oob) run<2, true>("", 64, 8, out, src, filt, false); else run<2, false>("", 64, 8, out, src, filt, false);
		hipMemcpy(b.data(), out, b.size() * 4, hipMemcpyDeviceToHost);
		size_t bad = 0;
		double norm = 0;
		for (size_t i = 0; i < a.size(); ++i) bad += a[i] != b[i], norm += std::fabs(a[i]);
		printf("LDS-direct vs register-staged (%s): %zu of %zu values differ, mean |v| = %g\n", oob ? "with out-of-range lanes" : "all lanes in range",
		       bad, a.size(), norm / a.size());
		hipMemset(out, 0, a.size() * 4);
		if (oob) run<3, true>("", 64, 8, out, src, filt, false); else run<3, false>("", 64, 8, out, src, filt, false);
		hipMemcpy(b.data(), out, b.size() * 4, hipMemcpyDeviceToHost);
		bad = 0;
		for (size_t i = 0; i < a.size(); ++i) bad += a[i] != b[i];
		printf("3-stage LDS-direct ring vs register-staged: %zu of %zu values differ\n", bad, a.size());
	}

	{
		unsigned long long *stamps;
		hipMalloc(&stamps, 8192 * 4 * 8);
		hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &stamps, sizeof(stamps));
		for (int nb : {1024, 784, 3072}) {
			const unsigned plane = 1u << 20, planes = 1024, kfilt = 4096;
			for (int r = 0; r < 3; ++r) loop<1, false><<<nb, 256>>>(out, 64, src, plane, planes, filt, kfilt);
			hipDeviceSynchronize();
			std::vector<unsigned long long> h(nb * 4);
			hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
			unsigned long long t0 = ~0ull, t1 = 0;
			for (int b = 0; b < nb; ++b) t0 = std::min(t0, h[b * 4]), t1 = std::max(t1, h[b * 4 + 1]);
			printf("timeline of %d blocks x 64 k-tiles (100 MHz ticks; kernel spans %llu): start offset / end offset / duration, by deciles of blocks sorted by end time\n", nb, t1 - t0);
			std::vector<int> order(nb);
			for (int b = 0; b < nb; ++b) order[b] = b;
			std::sort(order.begin(), order.end(), [&](int a, int b) { return h[a * 4 + 1] < h[b * 4 + 1]; });
			for (int q = 0; q <= 10; ++q) {
				const int b = order[std::min(nb - 1, q * nb / 10)];
				printf("  %3d%%: block %5d start %6llu end %6llu dur %6llu  xcc %llu cu/se id %#llx\n", q * 10, b, h[b * 4] - t0, h[b * 4 + 1] - t0,
				       h[b * 4 + 1] - h[b * 4], h[b * 4 + 2] >> 32, h[b * 4 + 2] & 0xffffffffull);
			}
			// per-XCC last end
			unsigned long long xend[8] = {}, xstart[8];
			for (int x = 0; x < 8; ++x) xstart[x] = ~0ull;
			int xcount[8] = {};
			for (int b = 0; b < nb; ++b) {
				const int x = (int)(h[b * 4 + 2] >> 32) & 7;
				xend[x] = std::max(xend[x], h[b * 4 + 1] - t0), xstart[x] = std::min(xstart[x], h[b * 4] - t0), xcount[x]++;
			}
			for (int x = 0; x < 8; ++x) printf("  xcc %d: %4d blocks, first start %5llu, last end %6llu\n", x, xcount[x], xstart[x], xend[x]);
		}
		stamps = nullptr;
		hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &stamps, sizeof(stamps));
	}
	// grid-size dependence, launches back to back (10 per measurement): is a launch with ~3 tiles per CU slower per tile?
	for (int nb : {512, 768, 784, 1024, 1568, 3072, 3136, 6144, 12288}) {
		hipEvent_t e0, e1;
		hipEventCreate(&e0), hipEventCreate(&e1);
		for (int kt : {16, 64}) {
			const unsigned plane = 1u << 20, planes = 1024, kfilt = 4096;
			auto timeit = [&](auto launch) {
				launch();
				hipEventRecord(e0);
				for (int r = 0; r < 10; ++r) launch();
				hipEventRecord(e1);
				hipEventSynchronize(e1);
				float ms;
				hipEventElapsedTime(&ms, e0, e1);
				return ms / 10;
			};
			const float m1 = timeit([&] { loop<1, false><<<nb, 256>>>(out, kt, src, plane, planes, filt, kfilt); });
			const float m2 = timeit([&] { loop<2, false><<<nb, 256>>>(out, kt, src, plane, planes, filt, kfilt); });
			const float m3 = timeit([&] {
				static const int zeros[8] = {};
				hipMemcpyToSymbolAsync(HIP_SYMBOL(g_ctr), zeros, sizeof(zeros), 0, hipMemcpyHostToDevice, 0);
				hipMemcpyToSymbolAsync(HIP_SYMBOL(g_ntiles), &nb, sizeof(int), 0, hipMemcpyHostToDevice, 0);
				loop<5, false><<<nb + nb / 8 + 8, 256>>>(out, kt, src, plane, planes, filt, kfilt);
			});
			const double gf = (double)nb * kt * 2.0 * BM * BN * BK / 1e9;
			printf("%5d blocks (%5.2f per CU) x %3d k-tiles: production %7.3f ms %6.1f TF | LDS-direct %7.3f ms %6.1f TF | dynamic tile ids %7.3f ms %6.1f TF\n",
			       nb, nb / 256.0, kt, m1, gf / m1, m2, gf / m2, m3, gf / m3);
		}
	}
	return 0;
}
