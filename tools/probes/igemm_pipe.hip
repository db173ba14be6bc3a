// Probe (round 4): how much of the matrix pipe ONE workgroup per CU (one wave per SIMD) of the implicit GEMM's k-loop can use,
// and what the k-tile boundary costs. The production loop (128x128 tile, BK = 16, 4 waves, LDS double buffer) ends every
// k-tile with  [wait for the tile's last global load] -> park in LDS -> barrier -> first fragment read -> MFMA : a lone wave
// idles through all of it, and profiles/r02_lds_dma_probe.txt fits a lone-wave utilisation of ~0.43 (1-(1-p)^n for n
// resident waves per SIMD). Variants:
//   V 0  production order (loads spread over the 8 k2-steps, park + barrier + first read at the tile's end)
//   V 1  all of the next tile's loads issued in k2-steps 0-1, rest as V 0
//   V 2  V 1 + the next tile parked during k2-step 6, the barrier BETWEEN the MFMAs of k2-step 7 and the next tile's first
//        fragments read behind it: the boundary sits in the shadow of k2-step 7's MFMAs
//   V 3  V 2 with two register sets: loads run two k-tiles ahead (a whole tile of latency cover)
//   V 4  V 2 without sched_barrier clumps around the MFMAs of a k2-step (compiler's own interleave)
//   V 5  no staging at all (fragment reads + MFMAs + barrier): ceiling of the structure
// Every staged variant must produce V 0's values bit for bit.
// Build: hipcc --offload-arch=gfx950 -O3 -o igemm_pipe igemm_pipe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 16;

template <int V, int WPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, 8)))
loop(float *out, int ktiles, const float *src, unsigned plane, unsigned planes, const float *filt, unsigned kfilt) {
	__shared__ __attribute__((aligned(16))) float smem[2 * BK * (BM + BN)];
	float(*As)[BK][BM] = reinterpret_cast<float(*)[BK][BM]>(smem);
	float(*Bs)[BK][BN] = reinterpret_cast<float(*)[BK][BN]>(smem + 2 * BK * BM);
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lhi = lane >> 5;
	const unsigned tile = blockIdx.x;

	const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, plane * planes, 0x00020000);
	const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *)filt, 0, kfilt * BM * 4, 0x00020000);

	const int jb = tid % BN, kb0 = __builtin_amdgcn_readfirstlane(tid / BN);
	const unsigned voffB = ((tile % 2048u) * BN + jb) * 4u;
	unsigned voffA[2];
	for (int i = 0; i < 2; ++i) voffA[i] = (unsigned)(tid + i * 256) * 16u;

	f32x16 acc[2][2];
	for (int i = 0; i < 2; ++i)
		for (int j = 0; j < 2; ++j)
			for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

	f32x4 ra[2][2];          // [register set][i]
	float rb[2][8];
	float av[2][2], bv[2][2];

	auto load_one = [&](auto set, int kt, int j) {       // the j-th of a tile's 8 gathers (+ the filter loads for j < 2)
		constexpr int S = decltype(set)::value;
		const unsigned soffA = ((unsigned)(kt * BK) % kfilt) * BM * 4u;
		const unsigned soffB = (((unsigned)(kt * BK) % planes) + kb0 * 8 + j) * plane;
		if (j < 2) ra[S][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, voffA[j], soffA, 0));
		rb[S][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, voffB, soffB, 0));
	};
	auto park = [&](auto set, int buf) {
		constexpr int S = decltype(set)::value;
#pragma unroll
		for (int i = 0; i < 2; ++i) {
			const int f = tid + i * 256;
			*reinterpret_cast<f32x4 *>(&As[buf][f / 32][(f % 32) * 4]) = ra[S][i];
		}
#pragma unroll
		for (int i = 0; i < 8; ++i) Bs[buf][kb0 * 8 + i][jb] = rb[S][i];
	};
	auto read_frag = [&](int buf, int ks, int slot) {
#pragma unroll
		for (int i = 0; i < 2; ++i) av[slot][i] = As[buf][ks + lhi][wm * 64 + i * 32 + l31];
#pragma unroll
		for (int j = 0; j < 2; ++j) bv[slot][j] = Bs[buf][ks + lhi][wn * 64 + j * 32 + l31];
	};
	auto mfma = [&](int slot, int i, int jj) {
		acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[slot][i], bv[slot][jj], acc[i][jj], 0, 0, 0);
	};
	using S0 = std::integral_constant<int, 0>;
	using S1 = std::integral_constant<int, 1>;

	if constexpr (V == 0 || V == 1 || V == 5) {
		// ---- production structure
		auto compute_tile = [&](int buf, int kt_next, bool has_next) {
			read_frag(buf, 0, 0);
#pragma unroll
			for (int j = 0; j < BK / 2; ++j) {
				if (j + 1 < BK / 2) read_frag(buf, 2 * (j + 1), (j + 1) & 1);
				if (V != 5 && has_next) {
					if (V == 0) load_one(S0{}, kt_next, j);
					else if (j < 2) {
#pragma unroll
						for (int q = 0; q < 4; ++q) load_one(S0{}, kt_next, j * 4 + q);
					}
				}
				__builtin_amdgcn_sched_barrier(0);
				mfma(j & 1, 0, 0), mfma(j & 1, 0, 1), mfma(j & 1, 1, 0), mfma(j & 1, 1, 1);
				__builtin_amdgcn_sched_barrier(0);
			}
		};
		if (V == 5) {
			for (int i = tid; i < 2 * BK * (BM + BN); i += 256) smem[i] = (i * 2654435761u >> 8) * 1e-9f;
		} else {
#pragma unroll
			for (int j = 0; j < 8; ++j) load_one(S0{}, 0, j);
			park(S0{}, 0);
		}
		__syncthreads();
		for (int kt = 0; kt + 1 < ktiles; ++kt) {
			const int buf = kt & 1;
			compute_tile(buf, kt + 1, true);
			if (V != 5) park(S0{}, buf ^ 1);
			__syncthreads();
		}
		compute_tile((ktiles - 1) & 1, 0, false);
	} else {
		// ---- boundary in the shadow of k2-step 7. On entry the fragments of k2-step 0 are in slot 0.
		// lset: register set the loads of this tile go to; pset: the set parked during this tile (V 3: loads two tiles ahead)
		auto tile_body = [&](int buf, auto lset, auto pset, int kt_load, bool do_load, bool do_park) {
#pragma unroll
			for (int j = 0; j < BK / 2; ++j) {
				if (j + 1 < BK / 2) read_frag(buf, 2 * (j + 1), (j + 1) & 1);
				if (do_load && j < 2) {
#pragma unroll
					for (int q = 0; q < 4; ++q) load_one(lset, kt_load, j * 4 + q);
				}
				if (do_park && j == 6) park(pset, buf ^ 1);
				if (V != 4) __builtin_amdgcn_sched_barrier(0);
				if (j == 7 && do_park) {
					mfma(1, 0, 0), mfma(1, 0, 1);
					__builtin_amdgcn_sched_barrier(0);
					__syncthreads();                       // everyone's park is visible, everyone is done reading `buf`
					read_frag(buf ^ 1, 0, 0);
					__builtin_amdgcn_sched_barrier(0);
					mfma(1, 1, 0), mfma(1, 1, 1);
				} else {
					mfma(j & 1, 0, 0), mfma(j & 1, 0, 1), mfma(j & 1, 1, 0), mfma(j & 1, 1, 1);
				}
				if (V != 4) __builtin_amdgcn_sched_barrier(0);
			}
		};
#pragma unroll
		for (int j = 0; j < 8; ++j) load_one(S0{}, 0, j);
		park(S0{}, 0);
		if (V == 3 && ktiles > 1) {
#pragma unroll
			for (int j = 0; j < 8; ++j) load_one(S1{}, 1, j);
		}
		__syncthreads();
		read_frag(0, 0, 0);
		if constexpr (V == 3) {
			// tile t: loads of tile t+2 into set t&1, parks set (t+1)&1; unrolled by two so that the sets are static
			int kt = 0;
			for (; kt + 2 < ktiles; kt += 2) {
				tile_body(0, S0{}, S1{}, kt + 2, true, true);
				tile_body(1, S1{}, S0{}, kt + 3, kt + 3 < ktiles, true);
			}
			// 1 or 2 tiles left (ktiles - kt)
			if (ktiles - kt == 2) {
				tile_body(0, S0{}, S1{}, 0, false, true);
				tile_body(1, S1{}, S0{}, 0, false, false);
			} else {
				tile_body(0, S0{}, S1{}, 0, false, false);
			}
		} else {
			for (int kt = 0; kt + 1 < ktiles; ++kt) tile_body(kt & 1, S0{}, S0{}, kt + 1, true, true);
			tile_body((ktiles - 1) & 1, S0{}, S0{}, 0, false, false);
		}
	}

	float *o = out + (size_t)tile * BM * BN;
	for (int i = 0; i < 2; ++i)
		for (int j = 0; j < 2; ++j)
			for (int r = 0; r < 16; ++r) o[((i * 2 + j) * 16 + r) * 256 + tid] = acc[i][j][r];
}

// ---- b64 fragment reads: LDS cells [group of 4 k][half][row]{2 floats}; lane (row, h) reads k = {4g+2h, 4g+2h+1} with one
// ds_read_b64 and feeds two consecutive k2-steps. KT = k-tile depth (16: 16 KB per workgroup as production, 32: 64 KB, two
// workgroups per CU). Production order otherwise (gathers spread over the tile, park + barrier + first read at its end).
template <int KT, int WPE, bool STAGE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, 8)))
loop64(float *out, int ksteps16, const float *src, unsigned plane, unsigned planes, const float *filt, unsigned kfilt) {
	constexpr int G = KT / 4;                       // groups of 4 k per tile
	__shared__ __attribute__((aligned(16))) float smem[2 * KT * (BM + BN)];
	typedef float f32x2 __attribute__((ext_vector_type(2)));
	auto cellA = [&](int buf, int g, int h, int row) { return reinterpret_cast<f32x2 *>(smem + ((size_t)(buf * G + g) * 2 + h) * BM * 2) + row; };
	auto cellB = [&](int buf, int g, int h, int row) { return reinterpret_cast<f32x2 *>(smem + 2 * KT * BM + ((size_t)(buf * G + g) * 2 + h) * BN * 2) + row; };
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lhi = lane >> 5;
	const unsigned tile = blockIdx.x;
	const int ktiles = ksteps16 * 16 / KT;

	const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, plane * planes, 0x00020000);
	const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *)filt, 0, kfilt * BM * 4, 0x00020000);
	constexpr int NB = KT / 2;                      // gathers per thread and tile (2 threads per pixel column)
	constexpr int NA = KT * BM / 4 / 256;           // 16-byte filter loads per thread and tile
	const int jb = tid % BN, kb0 = __builtin_amdgcn_readfirstlane(tid / BN);
	const unsigned voffB = ((tile % 2048u) * BN + jb) * 4u;

	f32x16 acc[2][2];
	for (int i = 0; i < 2; ++i)
		for (int j = 0; j < 2; ++j)
			for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
	f32x4 ra[NA];
	float rb[NB];

	auto load_one = [&](int kt, int j) {
		const unsigned soffA = ((unsigned)(kt * KT) % kfilt) * BM * 4u;
		const unsigned soffB = (((unsigned)(kt * KT) % planes) + kb0 * NB + j) * plane;
		if (j < NA) ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, (unsigned)(tid + j * 256) * 16u, soffA, 0));
		rb[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, voffB, soffB, 0));
	};
	auto park = [&](int buf) {
#pragma unroll
		for (int i = 0; i < NA; ++i) reinterpret_cast<f32x4 *>(smem + (size_t)buf * KT * BM)[tid + i * 256] = ra[i];      // tile image = global order
#pragma unroll
		for (int i = 0; i < NB; i += 2) {
			const int k = kb0 * NB + i;              // even: pair (k, k+1) = element pair of cell [k/4][(k/2)&1]
			*cellB(buf, k / 4, (k / 2) & 1, jb) = f32x2{rb[i], rb[i + 1]};
		}
	};
	f32x2 av[2][2], bv[2][2];
	auto read_frag = [&](int buf, int g, int slot) {
#pragma unroll
		for (int i = 0; i < 2; ++i) av[slot][i] = *cellA(buf, g, lhi, wm * 64 + i * 32 + l31);
#pragma unroll
		for (int j = 0; j < 2; ++j) bv[slot][j] = *cellB(buf, g, lhi, wn * 64 + j * 32 + l31);
	};
	auto compute_tile = [&](int buf, int kt_next, bool has_next) {
		read_frag(buf, 0, 0);
#pragma unroll
		for (int g = 0; g < G; ++g) {
			if (g + 1 < G) read_frag(buf, g + 1, (g + 1) & 1);
			if (STAGE && has_next) {
#pragma unroll
				for (int q = 0; q < NB / G; ++q) load_one(kt_next, g * (NB / G) + q);
			}
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int e = 0; e < 2; ++e)
#pragma unroll
				for (int i = 0; i < 2; ++i)
#pragma unroll
					for (int jj = 0; jj < 2; ++jj)
						acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][i][e], bv[g & 1][jj][e], acc[i][jj], 0, 0, 0);
			__builtin_amdgcn_sched_barrier(0);
		}
	};
	if (STAGE) {
#pragma unroll
		for (int j = 0; j < NB; ++j) load_one(0, j);
		park(0);
	} else {
		for (int i = tid; i < 2 * KT * (BM + BN); i += 256) smem[i] = (i * 2654435761u >> 8) * 1e-9f;
	}
	__syncthreads();
	for (int kt = 0; kt + 1 < ktiles; ++kt) {
		const int buf = kt & 1;
		compute_tile(buf, kt + 1, true);
		if (STAGE) park(buf ^ 1);
		__syncthreads();
	}
	compute_tile((ktiles - 1) & 1, 0, false);

	float *o = out + (size_t)tile * BM * BN;
	for (int i = 0; i < 2; ++i)
		for (int j = 0; j < 2; ++j)
			for (int r = 0; r < 16; ++r) o[((i * 2 + j) * 16 + r) * 256 + tid] = acc[i][j][r];
}

template <int KT, int WPE, bool STAGE>
float timeit64(int blocks, int ksteps16, float *out, const float *src, const float *filt, int reps = 10) {
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	loop64<KT, WPE, STAGE><<<blocks, 256>>>(out, ksteps16, src, 1u << 20, 1024, filt, 4096);
	hipEventRecord(e0);
	for (int r = 0; r < reps; ++r) loop64<KT, WPE, STAGE><<<blocks, 256>>>(out, ksteps16, src, 1u << 20, 1024, filt, 4096);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	hipEventDestroy(e0), hipEventDestroy(e1);
	return ms / reps;
}

__global__ void fill(float *p, size_t n, unsigned seed) {
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
		p[i] = (float)(((unsigned)i * 2654435761u + seed) >> 20) * (1.f / 4096.f) - 0.5f;
}

static const unsigned kPlane = 1u << 20, kPlanes = 1024, kFilt = 4096;

template <int V, int WPE>
float timeit(int blocks, int ktiles, float *out, const float *src, const float *filt, int reps = 10) {
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	loop<V, WPE><<<blocks, 256>>>(out, ktiles, src, kPlane, kPlanes, filt, kFilt);
	hipEventRecord(e0);
	for (int r = 0; r < reps; ++r) loop<V, WPE><<<blocks, 256>>>(out, ktiles, src, kPlane, kPlanes, filt, kFilt);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	hipEventDestroy(e0), hipEventDestroy(e1);
	return ms / reps;
}

template <int V, int WPE>
bool same_as_v0(const std::vector<float> &ref, int blocks, int ktiles, float *out, const float *src, const float *filt) {
	hipMemset(out, 0, ref.size() * 4);
	loop<V, WPE><<<blocks, 256>>>(out, ktiles, src, kPlane, kPlanes, filt, kFilt);
	std::vector<float> b(ref.size());
	hipMemcpy(b.data(), out, b.size() * 4, hipMemcpyDeviceToHost);
	size_t bad = 0;
	for (size_t i = 0; i < ref.size(); ++i) bad += ref[i] != b[i];
	printf("V%d (waves_per_eu %d), %d k-tiles: %zu of %zu values differ from V0  (%s)\n", V, WPE, ktiles, bad, ref.size(),
	       hipGetErrorString(hipGetLastError()));
	return bad == 0;
}

int main() {
	float *src, *filt, *out;
	hipMalloc(&src, (size_t)1 << 30);
	hipMalloc(&filt, (size_t)kFilt * BM * 4);
	const int maxblocks = 6144;
	hipMalloc(&out, (size_t)maxblocks * BM * BN * 4);
	fill<<<4096, 256>>>(src, (size_t)1 << 28, 1u);
	fill<<<256, 256>>>(filt, (size_t)kFilt * BM, 7u);
	hipDeviceSynchronize();

	for (int kt : {1, 2, 3, 8, 9}) {
		std::vector<float> ref((size_t)64 * BM * BN);
		hipMemset(out, 0, ref.size() * 4);
		loop<0, 4><<<64, 256>>>(out, kt, src, kPlane, kPlanes, filt, kFilt);
		hipMemcpy(ref.data(), out, ref.size() * 4, hipMemcpyDeviceToHost);
		double norm = 0;
		for (float v : ref) norm += std::fabs(v);
		printf("V0 mean |v| = %g\n", norm / ref.size());
		same_as_v0<1, 4>(ref, 64, kt, out, src, filt);
		same_as_v0<2, 4>(ref, 64, kt, out, src, filt);
		same_as_v0<3, 4>(ref, 64, kt, out, src, filt);
		same_as_v0<3, 3>(ref, 64, kt, out, src, filt);
		same_as_v0<4, 4>(ref, 64, kt, out, src, filt);
	}

	printf("\nTFLOP/s by variant (10 launches back to back); blocks per CU = resident waves per SIMD up to 4\n");
	printf("%-28s %8s %8s %8s %8s %8s %8s %8s\n", "blocks x k-tiles", "V0", "V1", "V2", "V3", "V3/3wpe", "V4", "V5");
	for (int nb : {256, 512, 768, 784, 1024, 1568, 3072, 6144}) {
		for (int kt : {16, 64}) {
			const double gf = (double)nb * kt * 2.0 * BM * BN * BK / 1e9;
			const float t0 = timeit<0, 4>(nb, kt, out, src, filt);
			const float t1 = timeit<1, 4>(nb, kt, out, src, filt);
			const float t2 = timeit<2, 4>(nb, kt, out, src, filt);
			const float t3 = timeit<3, 4>(nb, kt, out, src, filt);
			const float t33 = timeit<3, 3>(nb, kt, out, src, filt);
			const float t4 = timeit<4, 4>(nb, kt, out, src, filt);
			const float t5 = timeit<5, 4>(nb, kt, out, src, filt);
			printf("%5d (%5.2f/CU) x %3d        %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f\n", nb, nb / 256.0, kt, gf / t0, gf / t1, gf / t2,
			       gf / t3, gf / t33, gf / t4, gf / t5);
		}
	}

	printf("\nb64 fragment reads (k pairs per lane), TFLOP/s: k-tile 16 / 32, staged (S) or fragment reads only (N); [2wpe] = two waves per SIMD allowed\n");
	printf("%-28s %9s %9s %9s %9s %9s\n", "blocks x k-tiles(16)", "KT16 S", "KT16 N", "KT32 S", "KT32 N", "KT32 S 2wpe");
	for (int nb : {256, 512, 768, 1024, 1536, 3072, 6144}) {
		for (int kt : {16, 64}) {
			const double gf = (double)nb * kt * 2.0 * BM * BN * BK / 1e9;
			const float a = timeit64<16, 4, true>(nb, kt, out, src, filt);
			const float b = timeit64<16, 4, false>(nb, kt, out, src, filt);
			const float c = timeit64<32, 4, true>(nb, kt, out, src, filt);
			const float d = timeit64<32, 4, false>(nb, kt, out, src, filt);
			const float e = timeit64<32, 2, true>(nb, kt, out, src, filt);
			printf("%5d (%5.2f/CU) x %3d        %9.1f %9.1f %9.1f %9.1f %9.1f\n", nb, nb / 256.0, kt, gf / a, gf / b, gf / c, gf / d, gf / e);
		}
	}
	return 0;
}
