// Winograd F(2x2, 3x3) convolution for gfx950: the 3x3 / stride-1 / undilated / ungrouped forward and backward-data
// passes (ConvFwdAlgo.winograd of Hip/Wrappers/MIOpen.py:28, miopenConvolutionFwdAlgoWinograd). 2.25x fewer
// multiply-accumulates than the implicit GEMM, all of them still on v_mfma_f32_32x32x2_f32:
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      per 2x2 output tile, 4x4 input patch d, 3x3 filter g
//
// One launch, nothing transformed goes through HBM except the filters (16 values per 3x3 filter, written by a tiny
// pre-pass like the implicit GEMM's filter packing):
//   * a workgroup (4 waves) owns 32 tiles x 64 produced channels x all 16 transform positions: 16 independent
//     [64 x C] . [C x 32] products, wave w accumulating positions 4w..4w+3 (8 MFMA tiles, 128 accumulator registers);
//   * the reduction runs over chunks of 4 channels: waves 0-1 gather the 4x4 patches (buffer loads, zero padding by
//     out-of-range offsets), transform them in registers and leave V[pos][tile][c] in LDS; waves 2-3 copy the matching
//     block of transformed filters U[pos][k][c]; LDS is double-buffered, one barrier per chunk;
//   * the epilogue passes the accumulators through LDS 16 channels at a time so that each thread holds the 16 positions
//     of one (tile, channel), applies A^T . A, adds the bias and stores the 2x2 outputs (lanes along the tile row).
#include "common.h"

#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));


#ifndef WN_WAVES
#define WN_WAVES 8      // forward / backward-data workgroup: 8 waves (2 positions each, 4 waves per SIMD) or 4 (4 positions, 2 per SIMD)
#endif
#ifndef WN_SELFWAVE
#define WN_SELFWAVE 0   // forward / backward-data: 0 = shared loaders + one barrier per chunk; 1 = barrier-free self-sufficient
                        // waves (wino_conv_kernel_sw) — measured equal on all four 3x3 layer shapes, kept for the comparison
#endif

namespace {

constexpr unsigned kOOB = 0xfffffff0u;      // buffer byte offset beyond every tensor: loads return 0, stores are dropped

constexpr int TB = 32;       // tiles per workgroup
constexpr int KB = 64;       // produced channels per workgroup
constexpr int BC = 4;        // reduction channels per chunk
constexpr int kVFloats = 16 * TB * BC, kUFloats = 16 * KB * BC;

struct WinoFilterArgs {
	const float *w;          // (K, C, 3, 3)
	float *u;                // [kblocks][chunks][16][2][KB][2]
	int mode;                // 0: forward (produced = K, reduction = C); 1: backward-data (produced = C, reduction = K, taps flipped)
	int K, C;                // dims of w
	int prod, red;           // produced / reduction channel counts
	int kblocks, chunks;
};

__device__ __forceinline__ void wino_filter_body(const WinoFilterArgs &a, long first, long step) {
	const long total = (long)a.kblocks * a.chunks * KB * BC;
	for (long i = first; i < total; i += step) {
		const int ci = (int)(i % 2), kk = (int)((i / 2) % KB), h = (int)((i / (2 * KB)) % 2);
		const long blk = i / (2 * KB * 2);
		const int chunk = (int)(blk % a.chunks), kb = (int)(blk / a.chunks);
		const int k = kb * KB + kk, c = chunk * BC + h * 2 + ci;

		float g[3][3];
#pragma unroll
		for (int r = 0; r < 3; ++r)
#pragma unroll
			for (int s = 0; s < 3; ++s) {
				float v = 0.f;
				if (k < a.prod && c < a.red)
					v = a.mode == 0 ? a.w[((long)k * a.C + c) * 9 + r * 3 + s] : a.w[((long)c * a.C + k) * 9 + (2 - r) * 3 + (2 - s)];
				g[r][s] = v;
			}

		float t[4][3];
#pragma unroll
		for (int s = 0; s < 3; ++s) {
			t[0][s] = g[0][s];
			t[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]);
			t[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]);
			t[3][s] = g[2][s];
		}
		float *dst = a.u + (blk * 16 * 2 + h) * (KB * 2) + kk * 2 + ci;
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const float u0 = t[r][0], u1 = 0.5f * (t[r][0] + t[r][1] + t[r][2]), u2 = 0.5f * (t[r][0] - t[r][1] + t[r][2]), u3 = t[r][2];
			dst[(r * 4 + 0) * (2 * KB * 2)] = u0;
			dst[(r * 4 + 1) * (2 * KB * 2)] = u1;
			dst[(r * 4 + 2) * (2 * KB * 2)] = u2;
			dst[(r * 4 + 3) * (2 * KB * 2)] = u3;
		}
	}
}

__global__ void __launch_bounds__(256) wino_filter_kernel(WinoFilterArgs a) {
	wino_filter_body(a, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

// the filter transforms of several layers / passes in one launch (pz_conv2d_prepack); job j owns blocks [start[j], start[j + 1])
struct WinoFilterBatch {
	int n, start[pz::kWinoBatch + 1];
	WinoFilterArgs job[pz::kWinoBatch];
};
static_assert(sizeof(WinoFilterBatch) <= 4000, "kernel arguments");

__global__ void __launch_bounds__(256) wino_filter_batch_kernel(WinoFilterBatch b) {
	int j = 0;
	while (j + 1 < b.n && (int)blockIdx.x >= b.start[j + 1]) ++j;
	const int nb = b.start[j + 1] - b.start[j];
	wino_filter_body(b.job[j], (long)(blockIdx.x - b.start[j]) * 256 + threadIdx.x, (long)nb * 256);
}

struct WinoArgs {
	const float *x;          // gathered tensor (N, C, H, W)
	const float *u;          // transformed filters
	const float *bias;       // per produced channel or NULL
	float *y;                // (N, K, P, Q)
	int N, C, H, W, K, P, Q;
	int pad_h, pad_w;
	int TY, TX, tiles;       // 2x2 tiles per image column / row, N*TY*TX
	int chunks, tblocks;
	unsigned x_bytes, y_bytes;
	// optional [K][tblocks] {shift, sum(v - shift), sum((v - shift)^2), count} over the workgroup's 32 tiles of a channel,
	// for a following batch normalisation (the strip format of the implicit GEMM's epilogue, with an explicit count:
	// border tiles hold fewer than 4 pixels) — 8-wave kernel only
	float4 *stats;
};

// D[m = channel][n = tile] of the 16 positions -> output tensor. Register i of a lane is row 8 (i / 4) + 4 (lane / 32) + i % 4,
// column lane % 32. The accumulators pass through LDS 16 channels at a time so that one thread holds the 16 positions
// of a (tile, channel), applies A^T . A, adds the bias and stores the 2x2 outputs (lanes along the tile row).
__device__ __forceinline__ void wino_epilogue(const WinoArgs &a, f32x16 (&acc)[4][2], float *smem, int kb, int tb, int tid, int wave,
                                              int lane) {
	const int l31 = lane & 31, lhi = lane >> 5;
	const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void *)a.y, 0, a.y_bytes, 0x00020000);
	float *Ms = smem;                               // [16 positions][16 channels][32 tiles]

	const int t = tb * TB + l31;
	const bool tv = t < a.tiles;
	const int n = t / (a.TY * a.TX), rr = t - n * (a.TY * a.TX);
	const int ty = rr / a.TX, tx = rr - ty * a.TX;
	const bool row1 = 2 * ty + 1 < a.P, col1 = 2 * tx + 1 < a.Q;
	const unsigned pq4 = (unsigned)(a.P * a.Q) * 4u;
	const unsigned obase = (unsigned)((((long)n * a.K) * a.P + 2 * ty) * a.Q + 2 * tx) * 4u;

#pragma unroll
	for (int q = 0; q < 4; ++q) {
#pragma unroll
		for (int p = 0; p < 4; ++p)
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				const int kk = 8 * (i >> 2) + 4 * lhi + (i & 3);
				Ms[((4 * wave + p) * 16 + kk) * 32 + l31] = acc[p][q >> 1][8 * (q & 1) + i];
			}
		__syncthreads();

#pragma unroll
		for (int j = 0; j < 2; ++j) {
			const int kk = (tid >> 5) + 8 * j;
			const int k = kb * KB + q * 16 + kk;
			float m[4][4];
#pragma unroll
			for (int pos = 0; pos < 16; ++pos) m[pos >> 2][pos & 3] = Ms[(pos * 16 + kk) * 32 + l31];

			float r0[4], r1[4];
#pragma unroll
			for (int v = 0; v < 4; ++v) {
				r0[v] = m[0][v] + m[1][v] + m[2][v];
				r1[v] = m[1][v] - m[2][v] - m[3][v];
			}
			const float b = (a.bias != nullptr && k < a.K) ? a.bias[k] : 0.f;
			const float y00 = r0[0] + r0[1] + r0[2] + b, y01 = r0[1] - r0[2] - r0[3] + b;
			const float y10 = r1[0] + r1[1] + r1[2] + b, y11 = r1[1] - r1[2] - r1[3] + b;

			const bool kv = tv && k < a.K;
			const unsigned o = obase + (unsigned)k * pq4;
			const unsigned q4 = (unsigned)a.Q * 4u;
			__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y00), yr, kv ? o : kOOB, 0, 0);
			__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y01), yr, kv && col1 ? o + 4u : kOOB, 0, 0);
			__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y10), yr, kv && row1 ? o + q4 : kOOB, 0, 0);
			__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y11), yr, kv && row1 && col1 ? o + q4 + 4u : kOOB, 0, 0);
		}
		if (q < 3) __syncthreads();
	}
}

__global__ void __launch_bounds__(256, 2) wino_conv_kernel(WinoArgs a) {
	constexpr int kStage = kVFloats + kUFloats;                                      // one chunk of both operands: 24 KB
	__shared__ __attribute__((aligned(16))) float smem[3 * kStage];                    // 72 KB, two workgroups per CU

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int l31 = lane & 31, lhi = lane >> 5;

	const int kb = blockIdx.x / a.tblocks, tb = blockIdx.x - kb * a.tblocks;

	const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.x_bytes, 0x00020000);
	const unsigned hw4 = (unsigned)(a.H * a.W) * 4u;

	// ---- loader work of a chunk, the same amount on every wave:
	//   * patches: a thread owns (tile 16 (wave / 2) + lane % 16, channel lane / 16) and one half of the transform
	//     (wave % 2 = 0: rows 0-1 of V from patch rows 0-2; 1: rows 2-3 from patch rows 1-3) — three 16-byte row loads
	//     at any 4-byte alignment; rows outside the image get the out-of-range offset, columns outside are cleared;
	//   * transformed filters: 4 x 16 B per thread, a linear copy.
	const int hf = wave & 1;
	const int lt = 16 * (wave >> 1) + (lane & 15), lc = lane >> 4;
	unsigned voff[3];
	bool colok[4], fix[3];
	bool anyfix = false;
	{
		const int t = tb * TB + lt;
		const bool tv = t < a.tiles;
		const int n = t / (a.TY * a.TX), r = t - n * (a.TY * a.TX);
		const int ty = r / a.TX, tx = r - ty * a.TX;
		const int row0 = 2 * ty - a.pad_h + hf, col0 = 2 * tx - a.pad_w;
		const long base = (((long)n * a.C + lc) * a.H + row0) * a.W + col0;
#pragma unroll
		for (int e = 0; e < 3; ++e) {
			const bool ok = tv && (unsigned)(row0 + e) < (unsigned)a.H;
			const long off = base + (long)e * a.W;
			// a row that starts in front of the tensor (image 0, channel 0, row 0, left padding) would be out of range as a
			// whole: it is loaded from the tensor's first element instead and shifted by one column when it is consumed
			fix[e] = ok && off < 0;
			voff[e] = ok ? (off < 0 ? 0u : (unsigned)off * 4u) : kOOB;
			anyfix = anyfix || fix[e];
		}
#pragma unroll
		for (int j = 0; j < 4; ++j) colok[j] = (unsigned)(col0 + j) < (unsigned)a.W;
	}
	anyfix = __builtin_amdgcn_readfirstlane((int)(__builtin_amdgcn_ballot_w64(anyfix) != 0ull)) != 0;
	const unsigned vdst = (unsigned)(((lc >> 1) * TB + lt) * 2 + (lc & 1)) + (unsigned)(hf * 8) * (TB * BC);
	const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(
	    (void *)(a.u + (size_t)kb * a.chunks * kUFloats), 0, (unsigned)a.chunks * (kUFloats * 4u), 0x00020000);

	f32x4 sp[3], su[4];                         // staged chunk: 3 patch rows, 4 x 16 B of filters

	auto issue_loads = [&](int chunk) {
		const unsigned soff = (unsigned)chunk * (BC * hw4);
#pragma unroll
		for (int e = 0; e < 3; ++e) sp[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, voff[e], soff, 0));
#pragma unroll
		for (int i = 0; i < 4; ++i)
			su[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ur, (unsigned)(tid + i * 256) * 16u,
			                                                                          (unsigned)chunk * (kUFloats * 4u), 0));
	};

	// the staged chunk -> LDS stage `stg` in 8 slices, each of which sits behind one MFMA of the chunk being multiplied.
	// VALU instructions take matrix-pipe time (tools/probes/lds_mfma.hip), so the transform is packed-fp32 math on column
	// pairs: slices 0-3 clear the columns outside the image (column 1 never is, padding <= 1) and form the two rows of
	// B^T d this half needs; 4-5 apply (.) B to them and store; 4-7 also store 16 B of filters each
	f32x2 tA[2], tB[2];
	auto store_slice = [&](auto half, auto fixed, float *stg, int slice) {
		constexpr int HF = decltype(half)::value;
		if (slice < 4) {
			if constexpr (decltype(fixed)::value)
				if (slice == 0) {
#pragma unroll
					for (int e = 0; e < 3; ++e)          // a row loaded from the tensor's first element: move it one column right
						if (fix[e]) sp[e] = f32x4{0.f, sp[e][0], sp[e][1], sp[e][2]};
				}
			if (slice != 1) {
				const int j = slice;
#pragma unroll
				for (int e = 0; e < 3; ++e) sp[e][j] = colok[j] ? sp[e][j] : 0.f;
			}
			if (slice == 1 || slice == 3) {
				const int q = slice >> 1;
				const f32x2 e0 = {sp[0][2 * q], sp[0][2 * q + 1]}, e1 = {sp[1][2 * q], sp[1][2 * q + 1]};
				const f32x2 e2 = {sp[2][2 * q], sp[2][2 * q + 1]};
				if constexpr (HF == 0) {
					tA[q] = e0 - e2, tB[q] = e1 + e2;         // rows 0, 1 of B^T d from patch rows (0, 1, 2)
				} else {
					tA[q] = e1 - e0, tB[q] = e0 - e2;         // rows 2, 3 from patch rows (1, 2, 3)
				}
			}
		} else if (slice < 6) {
			const f32x2 t01 = slice == 4 ? tA[0] : tB[0], t23 = slice == 4 ? tA[1] : tB[1];
			f32x2 v01, v23;                               // (t0 - t2, t1 + t2), (t2 - t1, t1 - t3)
			asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(v01) : "v"(t01), "v"(t23));
			v23[0] = t23[0] - t01[1], v23[1] = t01[1] - t23[1];      // (scalar: a packed op whose LOW lane reads a HIGH half is not safe)
			float *dst = stg + vdst + (slice - 4) * 4 * (TB * BC);
			dst[0 * (TB * BC)] = v01[0];
			dst[1 * (TB * BC)] = v01[1];
			dst[2 * (TB * BC)] = v23[0];
			dst[3 * (TB * BC)] = v23[1];
		}
		if (slice >= 4) reinterpret_cast<f32x4 *>(stg + kVFloats)[tid + (slice - 4) * 256] = su[slice - 4];
	};

	f32x16 acc[4][2];
#pragma unroll
	for (int p = 0; p < 4; ++p)
#pragma unroll
		for (int m = 0; m < 2; ++m)
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[p][m][r] = 0.f;

	// fragment addresses: lane half h reads channels {2h, 2h+1} of the chunk as 8 bytes (k2-step s takes component s)
	const int vfrag = ((4 * wave * 2 + lhi) * TB + l31) * 2;                  // + p * (2 * TB * 2)
	const int ufrag = kVFloats + ((4 * wave * 2 + lhi) * KB + l31) * 2;       // + p * (2 * KB * 2) + mt * 64

	// fragments of positions 0-1 of the next chunk are prefetched across the barrier (double-buffered registers); those of
	// positions 2-3 are read right after it, behind the first MFMAs
	struct Frag {
		f32x2 bv[2], av[2][2];
	};
	auto read_frags = [&](const float *stg, Frag &f, int p0) {
#pragma unroll
		for (int p = 0; p < 2; ++p) {
			f.bv[p] = *reinterpret_cast<const f32x2 *>(stg + vfrag + (p0 + p) * (2 * TB * 2));
			f.av[p][0] = *reinterpret_cast<const f32x2 *>(stg + ufrag + (p0 + p) * (2 * KB * 2));
			f.av[p][1] = *reinterpret_cast<const f32x2 *>(stg + ufrag + (p0 + p) * (2 * KB * 2) + 64);
		}
	};

	// Steady state of chunk ch (LDS stage ch % 3, fragments already in registers): the fragment reads of chunk ch+1 are
	// issued; the 8 MFMAs of k2-step 0; the 8 MFMAs of k2-step 1, each carrying one slice of chunk ch+2's transform / LDS
	// store (its global loads were issued at the end of chunk ch-1: a whole chunk of latency cover); the loads of chunk
	// ch+3 go into the freed staging registers; the only barrier of the chunk. Three stages: the one being read, the one prefetched from, the one being written.
	auto run = [&](auto half, auto fixed) {
		Frag f0, f1;
		issue_loads(0);
#pragma unroll
		for (int sl = 0; sl < 8; ++sl) store_slice(half, fixed, smem, sl);
		if (a.chunks > 1) {
			issue_loads(1);
#pragma unroll
			for (int sl = 0; sl < 8; ++sl) store_slice(half, fixed, smem + kStage, sl);
		}
		if (a.chunks > 2) issue_loads(2);
		__syncthreads();
		read_frags(smem, f0, 0);

		int s_cur = 0;                              // stage of chunk ch
		auto body = [&](int ch, Frag &cur, Frag &nxt) {
			const int s_nxt = s_cur == 2 ? 0 : s_cur + 1, s_wr = s_nxt == 2 ? 0 : s_nxt + 1;
			Frag late;
			read_frags(smem + s_cur * kStage, late, 2);
			read_frags(smem + s_nxt * kStage, nxt, 0);       // past the last chunk: a stale stage, never used
			// past the last chunk the slices store stale registers into a stage nobody reads any more, and the loads fetch
			// the last chunk again: no branches in the steady state
			float *wr = smem + s_wr * kStage;
#pragma unroll
			for (int p = 0; p < 4; ++p)
#pragma unroll
				for (int m = 0; m < 2; ++m) {
					const Frag &f = p < 2 ? cur : late;
					acc[p][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.av[p & 1][m][0], f.bv[p & 1][0], acc[p][m], 0, 0, 0);
				}
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int p = 0; p < 4; ++p)
#pragma unroll
				for (int m = 0; m < 2; ++m) {
					const Frag &f = p < 2 ? cur : late;
					acc[p][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.av[p & 1][m][1], f.bv[p & 1][1], acc[p][m], 0, 0, 0);
					store_slice(half, fixed, wr, p * 2 + m);
					__builtin_amdgcn_sched_barrier(0);
				}
			issue_loads(min(ch + 3, a.chunks - 1));
			__builtin_amdgcn_sched_barrier(0);
			__syncthreads();
			s_cur = s_nxt;
		};

		int ch = 0;
		for (; ch + 1 < a.chunks; ch += 2) {
			body(ch, f0, f1);
			body(ch + 1, f1, f0);
		}
		if (ch < a.chunks) body(ch, f0, f1);
	};
	if (anyfix) {                                   // one wave of the whole launch
		if (hf == 0)
			run(std::integral_constant<int, 0>{}, std::true_type{});
		else
			run(std::integral_constant<int, 1>{}, std::true_type{});
	} else if (hf == 0) {
		run(std::integral_constant<int, 0>{}, std::false_type{});
	} else {
		run(std::integral_constant<int, 1>{}, std::false_type{});
	}


	wino_epilogue(a, acc, smem, kb, tb, tid, wave, lane);
}


#if WN_WAVES == 8
// ------------------------------------------------------------------------------------------------
// The same workgroup block (32 tiles x 64 channels x 16 positions, 3-stage ring) on 8 waves: wave w accumulates positions
// 2w, 2w+1 (64 accumulator registers), so two workgroups per CU put 4 waves on every SIMD instead of 2 — per SIMD the same
// MFMAs and the same loader work, spread over twice as many instruction streams. Waves 0-3 gather and transform the
// patches exactly as wino_conv_kernel's four waves do, waves 4-7 copy the transformed filters (4 x 16 B per thread).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wino_epilogue8(const WinoArgs &a, f32x16 (&acc)[2][2], float *smem, int kb, int tb, int tid, int wave,
                                               int lane) {
	const int l31 = lane & 31, lhi = lane >> 5;
	const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void *)a.y, 0, a.y_bytes, 0x00020000);
	float *Ms = smem;                               // [16 positions][16 channels][32 tiles]

	const int t = tb * TB + l31;
	const bool tv = t < a.tiles;
	const int n = t / (a.TY * a.TX), rr = t - n * (a.TY * a.TX);
	const int ty = rr / a.TX, tx = rr - ty * a.TX;
	const bool row1 = 2 * ty + 1 < a.P, col1 = 2 * tx + 1 < a.Q;
	const unsigned pq4 = (unsigned)(a.P * a.Q) * 4u;
	const unsigned obase = (unsigned)((((long)n * a.K) * a.P + 2 * ty) * a.Q + 2 * tx) * 4u;

#pragma unroll
	for (int q = 0; q < 4; ++q) {
#pragma unroll
		for (int p = 0; p < 2; ++p)
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				const int kk = 8 * (i >> 2) + 4 * lhi + (i & 3);
				Ms[((2 * wave + p) * 16 + kk) * 32 + l31] = acc[p][q >> 1][8 * (q & 1) + i];
			}
		__syncthreads();

		{
			const int kk = tid >> 5;                    // 512 threads = 32 tiles x 16 channels
			const int k = kb * KB + q * 16 + kk;
			float m[4][4];
#pragma unroll
			for (int pos = 0; pos < 16; ++pos) m[pos >> 2][pos & 3] = Ms[(pos * 16 + kk) * 32 + l31];

			float r0[4], r1[4];
#pragma unroll
			for (int v = 0; v < 4; ++v) {
				r0[v] = m[0][v] + m[1][v] + m[2][v];
				r1[v] = m[1][v] - m[2][v] - m[3][v];
			}
			const float b = (a.bias != nullptr && k < a.K) ? a.bias[k] : 0.f;
			const float y00 = r0[0] + r0[1] + r0[2] + b, y01 = r0[1] - r0[2] - r0[3] + b;
			const float y10 = r1[0] + r1[1] + r1[2] + b, y11 = r1[1] - r1[2] - r1[3] + b;

			const bool kv = tv && k < a.K;
			const unsigned o = obase + (unsigned)k * pq4;
			const unsigned q4 = (unsigned)a.Q * 4u;
			__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y00), yr, kv ? o : kOOB, 0, 0);
			__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y01), yr, kv && col1 ? o + 4u : kOOB, 0, 0);
			__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y10), yr, kv && row1 ? o + q4 : kOOB, 0, 0);
			__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y11), yr, kv && row1 && col1 ? o + q4 + 4u : kOOB, 0, 0);

			if (a.stats) {
				// the 32 lanes of a half-wave hold this channel's 32 tiles: shifted sums about the block's first output
				// (tile tb*TB always exists), fixed shuffle tree -> deterministic
				const float shift = __shfl(y00, lane & 32);
				float s1 = 0.f, s2 = 0.f, cnt = 0.f;
				auto take = [&](float v, bool ok) {
					const float dlt = ok ? v - shift : 0.f;
					s1 += dlt, s2 = __builtin_fmaf(dlt, dlt, s2), cnt += ok ? 1.f : 0.f;
				};
				take(y00, kv), take(y01, kv && col1), take(y10, kv && row1), take(y11, kv && row1 && col1);
#pragma unroll
				for (int m = 16; m > 0; m >>= 1) s1 += __shfl_xor(s1, m), s2 += __shfl_xor(s2, m), cnt += __shfl_xor(cnt, m);
				if (l31 == 0 && k < a.K) a.stats[(size_t)k * a.tblocks + tb] = make_float4(shift, s1, s2, cnt);
			}
		}
		if (q < 3) __syncthreads();
	}
}

__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) wino_conv_kernel8(WinoArgs a) {
	constexpr int kStage = kVFloats + kUFloats;
	__shared__ __attribute__((aligned(16))) float smem[3 * kStage];

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int l31 = lane & 31, lhi = lane >> 5;
	const int kb = blockIdx.x / a.tblocks, tb = blockIdx.x - kb * a.tblocks;

	const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.x_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(
	    (void *)(a.u + (size_t)kb * a.chunks * kUFloats), 0, (unsigned)a.chunks * (kUFloats * 4u), 0x00020000);
	const unsigned hw4 = (unsigned)(a.H * a.W) * 4u;

	// patch role (waves 0-3), as in wino_conv_kernel
	const int pw = wave & 3, hf = pw & 1;
	const int lt = 16 * (pw >> 1) + (lane & 15), lc = lane >> 4;
	unsigned voff[3];
	bool colok[4], fix[3];
	bool anyfix = false;
	{
		const int t = tb * TB + lt;
		const bool tv = t < a.tiles && wave < 4;
		const int n = t / (a.TY * a.TX), r = t - n * (a.TY * a.TX);
		const int ty = r / a.TX, tx = r - ty * a.TX;
		const int row0 = 2 * ty - a.pad_h + hf, col0 = 2 * tx - a.pad_w;
		const long base = (((long)n * a.C + lc) * a.H + row0) * a.W + col0;
#pragma unroll
		for (int e = 0; e < 3; ++e) {
			const bool ok = tv && (unsigned)(row0 + e) < (unsigned)a.H;
			const long off = base + (long)e * a.W;
			fix[e] = ok && off < 0;
			voff[e] = ok ? (off < 0 ? 0u : (unsigned)off * 4u) : kOOB;
			anyfix = anyfix || fix[e];
		}
#pragma unroll
		for (int j = 0; j < 4; ++j) colok[j] = (unsigned)(col0 + j) < (unsigned)a.W;
	}
	anyfix = __builtin_amdgcn_readfirstlane((int)(__builtin_amdgcn_ballot_w64(anyfix) != 0ull)) != 0;
	const unsigned vdst = (unsigned)(((lc >> 1) * TB + lt) * 2 + (lc & 1)) + (unsigned)(hf * 8) * (TB * BC);
	const unsigned uidx = (unsigned)(tid & 255);          // filter role (waves 4-7): 16-byte element of the chunk's block

	f32x4 st[4];                                // staged chunk: 3 patch rows or 4 x 16 B of filters

	auto issue_loads = [&](auto role, int chunk) {
		if constexpr (decltype(role)::value == 0) {
			const unsigned soff = (unsigned)chunk * (BC * hw4);
#pragma unroll
			for (int e = 0; e < 3; ++e) st[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, voff[e], soff, 0));
		} else {
#pragma unroll
			for (int i = 0; i < 4; ++i)
				st[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ur, (uidx + i * 256u) * 16u,
				                                                                          (unsigned)chunk * (kUFloats * 4u), 0));
		}
	};

	f32x2 tA[2], tB[2];
	auto store_slice = [&](auto role, auto half, auto fixed, float *stg, int slice) {
		constexpr int HF = decltype(half)::value;
		if constexpr (decltype(role)::value == 1) {
			if (slice >= 4) reinterpret_cast<f32x4 *>(stg + kVFloats)[uidx + (slice - 4) * 256] = st[slice - 4];
			return;
		} else {
			if (slice < 2) return;                    // the staged rows are needed from the third MFMA on
			const int sl = slice - 2;                 // 0-3: columns, 4-5: rows
			if (sl < 4) {
				if constexpr (decltype(fixed)::value)
					if (sl == 0) {
#pragma unroll
						for (int e = 0; e < 3; ++e)
							if (fix[e]) st[e] = f32x4{0.f, st[e][0], st[e][1], st[e][2]};
					}
				if (sl != 1) {
#pragma unroll
					for (int e = 0; e < 3; ++e) st[e][sl] = colok[sl] ? st[e][sl] : 0.f;
				}
				if (sl == 1 || sl == 3) {
					const int q = sl >> 1;
					const f32x2 e0 = {st[0][2 * q], st[0][2 * q + 1]}, e1 = {st[1][2 * q], st[1][2 * q + 1]};
					const f32x2 e2 = {st[2][2 * q], st[2][2 * q + 1]};
					if constexpr (HF == 0) {
						tA[q] = e0 - e2, tB[q] = e1 + e2;
					} else {
						tA[q] = e1 - e0, tB[q] = e0 - e2;
					}
				}
			} else {
				const f32x2 t01 = sl == 4 ? tA[0] : tB[0], t23 = sl == 4 ? tA[1] : tB[1];
				f32x2 v01, v23;
				asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(v01) : "v"(t01), "v"(t23));
				v23[0] = t23[0] - t01[1], v23[1] = t01[1] - t23[1];      // (scalar: a packed op whose LOW lane reads a HIGH half is not safe)
				float *dst = stg + vdst + (sl - 4) * 4 * (TB * BC);
				dst[0 * (TB * BC)] = v01[0];
				dst[1 * (TB * BC)] = v01[1];
				dst[2 * (TB * BC)] = v23[0];
				dst[3 * (TB * BC)] = v23[1];
			}
		}
	};

	f32x16 acc[2][2];
#pragma unroll
	for (int p = 0; p < 2; ++p)
#pragma unroll
		for (int m = 0; m < 2; ++m)
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[p][m][r] = 0.f;

	const int vfrag = ((2 * wave * 2 + lhi) * TB + l31) * 2;
	const int ufrag = kVFloats + ((2 * wave * 2 + lhi) * KB + l31) * 2;

	struct Frag {
		f32x2 bv[2], av[2][2];
	};
	auto read_frags = [&](const float *stg, Frag &f) {
#pragma unroll
		for (int p = 0; p < 2; ++p) {
			f.bv[p] = *reinterpret_cast<const f32x2 *>(stg + vfrag + p * (2 * TB * 2));
			f.av[p][0] = *reinterpret_cast<const f32x2 *>(stg + ufrag + p * (2 * KB * 2));
			f.av[p][1] = *reinterpret_cast<const f32x2 *>(stg + ufrag + p * (2 * KB * 2) + 64);
		}
	};

	auto run = [&](auto role, auto half, auto fixed) {
		Frag f0, f1;
		issue_loads(role, 0);
#pragma unroll
		for (int sl = 0; sl < 8; ++sl) store_slice(role, half, fixed, smem, sl);
		if (a.chunks > 1) {
			issue_loads(role, 1);
#pragma unroll
			for (int sl = 0; sl < 8; ++sl) store_slice(role, half, fixed, smem + kStage, sl);
		}
		if (a.chunks > 2) issue_loads(role, 2);
		__syncthreads();
		read_frags(smem, f0);

		int s_cur = 0;
		auto body = [&](int ch, Frag &cur, Frag &nxt) {
			const int s_nxt = s_cur == 2 ? 0 : s_cur + 1, s_wr = s_nxt == 2 ? 0 : s_nxt + 1;
			read_frags(smem + s_nxt * kStage, nxt);
			float *wr = smem + s_wr * kStage;
#pragma unroll
			for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
				for (int p = 0; p < 2; ++p)
#pragma unroll
					for (int m = 0; m < 2; ++m) {
						acc[p][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.av[p][m][s2], cur.bv[p][s2], acc[p][m], 0, 0, 0);
						store_slice(role, half, fixed, wr, s2 * 4 + p * 2 + m);
						// the patch rows are consumed by slice 5, the filter registers by the last slice
						if (s2 * 4 + p * 2 + m == (decltype(role)::value == 0 ? 5 : 7)) issue_loads(role, min(ch + 3, a.chunks - 1));
						__builtin_amdgcn_sched_barrier(0);
					}
			__syncthreads();
			s_cur = s_nxt;
		};

		int ch = 0;
		for (; ch + 1 < a.chunks; ch += 2) {
			body(ch, f0, f1);
			body(ch + 1, f1, f0);
		}
		if (ch < a.chunks) body(ch, f0, f1);
	};

	using R0 = std::integral_constant<int, 0>;
	using R1 = std::integral_constant<int, 1>;
	if (wave >= 4) {
		run(R1{}, R0{}, std::false_type{});
	} else if (anyfix) {
		if (hf == 0)
			run(R0{}, R0{}, std::true_type{});
		else
			run(R0{}, R1{}, std::true_type{});
	} else if (hf == 0) {
		run(R0{}, R0{}, std::false_type{});
	} else {
		run(R0{}, R1{}, std::false_type{});
	}

	wino_epilogue8(a, acc, smem, kb, tb, tid, wave, lane);
}
#endif  // WN_WAVES == 8

#if WN_SELFWAVE
// ------------------------------------------------------------------------------------------------
// The same product with self-sufficient waves (no barrier in the reduction loop). Wave w accumulates positions 4w..4w+3
// = row w of the transformed patch, and that row needs only two rows of the patch (0: d0-d2, 1: d1+d2, 2: d2-d1,
// 3: d1-d3): the wave loads those two rows of all 128 patches of a chunk itself (two patches per lane), applies (.) B,
// and copies its own quarter of the transformed filters — into a private two-stage LDS ring. Nothing a wave reads was
// written by another wave, LDS executes a wave's accesses in order, so the only barrier is the one in front of the
// epilogue; the four waves drift apart and fill each other's load / transform phases with MFMAs.
// Cost: the patch rows are loaded by two waves each (16 instead of 12 KB per chunk per workgroup, L1 hits).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2) wino_conv_kernel_sw(WinoArgs a) {
	constexpr int kWaveStage = 4 * TB * BC + 4 * KB * BC;        // one wave's V (512) + U (1024) floats of a chunk
	__shared__ __attribute__((aligned(16))) float smem[4 * 2 * kWaveStage];       // 48 KB

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int l31 = lane & 31, lhi = lane >> 5;
	const int kb = blockIdx.x / a.tblocks, tb = blockIdx.x - kb * a.tblocks;

	const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.x_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(
	    (void *)(a.u + (size_t)kb * a.chunks * kUFloats), 0, (unsigned)a.chunks * (kUFloats * 4u), 0x00020000);
	const unsigned hw4 = (unsigned)(a.H * a.W) * 4u;
	float *mine = smem + wave * (2 * kWaveStage);

	// patch rows of this wave's transform row; lane -> tile lane % 32, channels lane / 32 and lane / 32 + 2 of the chunk
	const int rowA = wave == 0 ? 0 : wave == 2 ? 2 : 1, rowB = wave == 0 ? 2 : wave == 2 ? 1 : wave == 1 ? 2 : 3;
	unsigned voff[2][2];
	bool colok[4], fix[2][2];
	bool anyfix = false;
	{
		const int t = tb * TB + l31;
		const bool tv = t < a.tiles;
		const int n = t / (a.TY * a.TX), r = t - n * (a.TY * a.TX);
		const int ty = r / a.TX, tx = r - ty * a.TX;
		const int row0 = 2 * ty - a.pad_h, col0 = 2 * tx - a.pad_w;
#pragma unroll
		for (int i = 0; i < 2; ++i)
#pragma unroll
			for (int e = 0; e < 2; ++e) {
				const int row = row0 + (e == 0 ? rowA : rowB);
				const bool ok = tv && (unsigned)row < (unsigned)a.H;
				const long off = (((long)n * a.C + lhi + 2 * i) * a.H + row) * a.W + col0;
				fix[i][e] = ok && off < 0;            // see wino_conv_kernel: the tensor's first row behind the left padding
				voff[i][e] = ok ? (off < 0 ? 0u : (unsigned)off * 4u) : kOOB;
				anyfix = anyfix || fix[i][e];
			}
#pragma unroll
		for (int j = 0; j < 4; ++j) colok[j] = (unsigned)(col0 + j) < (unsigned)a.W;
	}
	anyfix = __builtin_amdgcn_readfirstlane((int)(__builtin_amdgcn_ballot_w64(anyfix) != 0ull)) != 0;
	const unsigned vdst = (unsigned)(l31 * 2 + lhi);             // + (nu * 2 + i) * 64

	f32x4 sp[2][2], su[4];

	auto issue_loads = [&](int chunk) {
		const unsigned soff = (unsigned)chunk * (BC * hw4);
#pragma unroll
		for (int i = 0; i < 2; ++i)
#pragma unroll
			for (int e = 0; e < 2; ++e)
				sp[i][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, voff[i][e], soff, 0));
#pragma unroll
		for (int i = 0; i < 4; ++i)
			su[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
			    ur, (unsigned)(wave * (4 * KB * BC) + (lane + i * 64) * 4) * 4u, (unsigned)chunk * (kUFloats * 4u), 0));
	};

	// slices 0/2: clear the columns outside the image of patch 0/1; 1/3: its row of B^T d, (.) B, four LDS stores;
	// 4-7: 16 B of filters each
	auto store_slice = [&](auto row, auto fixed, float *stg, int slice) {
		constexpr int ROW = decltype(row)::value;
		if (slice < 4) {
			const int i = slice >> 1;
			if ((slice & 1) == 0) {
#pragma unroll
				for (int e = 0; e < 2; ++e) {
					if constexpr (decltype(fixed)::value)
						if (fix[i][e]) sp[i][e] = f32x4{0.f, sp[i][e][0], sp[i][e][1], sp[i][e][2]};
#pragma unroll
					for (int j = 0; j < 4; ++j)
						if (j != 1) sp[i][e][j] = colok[j] ? sp[i][e][j] : 0.f;
				}
			} else {
				f32x2 t01, t23;
				const f32x2 a01 = {sp[i][0][0], sp[i][0][1]}, a23 = {sp[i][0][2], sp[i][0][3]};
				const f32x2 b01 = {sp[i][1][0], sp[i][1][1]}, b23 = {sp[i][1][2], sp[i][1][3]};
				if constexpr (ROW == 1) {
					t01 = a01 + b01, t23 = a23 + b23;
				} else {
					t01 = a01 - b01, t23 = a23 - b23;
				}
				f32x2 v01, v23;                               // (t0 - t2, t1 + t2), (t2 - t1, t1 - t3)
				asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(v01) : "v"(t01), "v"(t23));
				v23[0] = t23[0] - t01[1], v23[1] = t01[1] - t23[1];      // (scalar: a packed op whose LOW lane reads a HIGH half is not safe)
				float *dst = stg + vdst + i * 64;
				dst[0 * 128] = v01[0];
				dst[1 * 128] = v01[1];
				dst[2 * 128] = v23[0];
				dst[3 * 128] = v23[1];
			}
		} else {
			reinterpret_cast<f32x4 *>(stg + 4 * TB * BC)[lane + (slice - 4) * 64] = su[slice - 4];
		}
	};

	f32x16 acc[4][2];
#pragma unroll
	for (int p = 0; p < 4; ++p)
#pragma unroll
		for (int m = 0; m < 2; ++m)
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[p][m][r] = 0.f;

	const int vfrag = (lhi * TB + l31) * 2;                          // + p * 128
	const int ufrag = 4 * TB * BC + (lhi * KB + l31) * 2;            // + p * 256 + mt * 64

	struct Frag {
		f32x2 bv[2], av[2][2];
	};
	auto read_frags = [&](const float *stg, Frag &f, int p0) {
#pragma unroll
		for (int p = 0; p < 2; ++p) {
			f.bv[p] = *reinterpret_cast<const f32x2 *>(stg + vfrag + (p0 + p) * 128);
			f.av[p][0] = *reinterpret_cast<const f32x2 *>(stg + ufrag + (p0 + p) * 256);
			f.av[p][1] = *reinterpret_cast<const f32x2 *>(stg + ufrag + (p0 + p) * 256 + 64);
		}
	};

	auto run = [&](auto row, auto fixed) {
		issue_loads(0);
#pragma unroll
		for (int sl = 0; sl < 8; ++sl) store_slice(row, fixed, mine, sl);
		issue_loads(min(1, a.chunks - 1));

		// chunk ch lives in stage ch % 2; its successor's staged registers are parked into the other stage while the first
		// half of its MFMAs run, the chunk after that is loaded into the freed registers
		auto body = [&](int ch, float *cur_stage, float *oth_stage) {
			Frag lo, hi;
			read_frags(cur_stage, lo, 0);
			read_frags(cur_stage, hi, 2);
#pragma unroll
			for (int p = 0; p < 4; ++p)
#pragma unroll
				for (int m = 0; m < 2; ++m) {
					const Frag &f = p < 2 ? lo : hi;
					acc[p][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.av[p & 1][m][0], f.bv[p & 1][0], acc[p][m], 0, 0, 0);
					store_slice(row, fixed, oth_stage, p * 2 + m);
					__builtin_amdgcn_sched_barrier(0);
				}
			issue_loads(min(ch + 2, a.chunks - 1));
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int p = 0; p < 4; ++p)
#pragma unroll
				for (int m = 0; m < 2; ++m) {
					const Frag &f = p < 2 ? lo : hi;
					acc[p][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.av[p & 1][m][1], f.bv[p & 1][1], acc[p][m], 0, 0, 0);
				}
			__builtin_amdgcn_sched_barrier(0);
		};

		int ch = 0;
		for (; ch + 1 < a.chunks; ch += 2) {
			body(ch, mine, mine + kWaveStage);
			body(ch + 1, mine + kWaveStage, mine);
		}
		if (ch < a.chunks) body(ch, mine, mine + kWaveStage);
	};

	auto dispatch = [&](auto fixed) {
		switch (wave) {
		case 0: run(std::integral_constant<int, 0>{}, fixed); break;
		case 1: run(std::integral_constant<int, 1>{}, fixed); break;
		case 2: run(std::integral_constant<int, 2>{}, fixed); break;
		default: run(std::integral_constant<int, 3>{}, fixed); break;
		}
	};
	if (anyfix)
		dispatch(std::true_type{});
	else
		dispatch(std::false_type{});

	__syncthreads();              // every wave is done with its ring before the epilogue reuses the memory
	wino_epilogue(a, acc, smem, kb, tb, tid, wave, lane);
}
#endif  // WN_SELFWAVE

// ------------------------------------------------------------------------------------------------
// backward-filter through the same transform:  dU[pos][k][c] = sum over tiles (A dY A^T)[pos] . (B^T d B)[pos],
// dg = G^T dU G. 16 products [64 k x tiles] . [tiles x 32 c] per workgroup, the reduction running over chunks of 4 tiles
// that lie side by side in one tile row — so a chunk's geometry (image, tile row, first/last chunk of the row) is
// wave-uniform: row validity picks the tensor's or a zero-length buffer descriptor, the chunk's place in the tensor is
// the scalar offset of the loads, and only the transforms and the border-column selects are vector work. The tile range
// is cut into `splits` slices (blockIdx); each leaves its 16 x 64 x 32 partial sums in a slab, summed and transformed
// back to 3x3 by the two small kernels below. Signs of row / column 3 of A dY A^T are folded into that last step.
// ------------------------------------------------------------------------------------------------
constexpr int CB = 32;                       // reduction-side (input) channels per workgroup
constexpr int kSlab = 16 * KB * CB;          // floats per (k block, c block) of transformed filter gradients

struct WinoWgradArgs {
	const float *x, *dy;     // (N, C, H, W), (N, K, P, Q)
	float *slabs;            // [splits][kblocks][cblocks][16][KB][CB]
	int N, C, H, W, K, P, Q, pad;
	int TY, TX4;             // tile rows per image, chunks (4 tiles) per tile row
	int chunks, splits, kblocks, cblocks;
	unsigned x_bytes, dy_bytes;
};

__device__ __forceinline__ float sel_mask(unsigned long long m, float v) {
	float r;
	asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(m));
	return r;
}

__global__ void __launch_bounds__(256, 2) wino_wgrad_kernel(WinoWgradArgs a) {
	constexpr int kStage = kVFloats + kUFloats;
	__shared__ __attribute__((aligned(16))) float smem[3 * kStage];

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int l31 = lane & 31, lhi = lane >> 5;

	const int nblk = a.kblocks * a.cblocks;
	const int split = blockIdx.x / nblk, blk = blockIdx.x - split * nblk;
	const int kb = blk / a.cblocks, cb = blk - kb * a.cblocks;
	const int g0 = (int)((long)a.chunks * split / a.splits), g1 = (int)((long)a.chunks * (split + 1) / a.splits);
	const int nch = g1 - g0;

	const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.x_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc((void *)a.dy, 0, a.dy_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t nullr = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, 0, 0x00020000);

	// ---- per-lane constants. Patches: tile lane % 4 of the chunk, channel (lane / 4) + 16 (wave / 2) of the block, half
	// wave % 2 of the transform. Gradient tiles: tile lane % 4, produced channel (lane / 4) + 16 wave.
	const int hf = wave & 1, t4 = lane & 3;
	const int cl = (lane >> 2) + 16 * (wave >> 1), c = cb * CB + cl;
	const int kl = (lane >> 2) + 16 * wave, k = kb * KB + kl;
	const unsigned voffx = c < a.C ? (unsigned)(c * a.H * a.W + 2 * t4) * 4u : kOOB;
	const unsigned voffx_first = c < a.C ? (voffx == 0 ? 0u : voffx - 4u) : kOOB;      // first tensor row: see issue_loads
	const unsigned voffz = k < a.K ? (unsigned)(k * a.P * a.Q + 2 * t4) * 4u : kOOB;

	// border columns, as lane masks (SGPR pairs): the first chunk of a tile row loses column 0 of tile 0 to the padding,
	// the last one the columns beyond the map
	const int jl = a.TX4 - 1;
	unsigned long long mlast[4], mz[2];
#pragma unroll
	for (int j = 0; j < 4; ++j) mlast[j] = __builtin_amdgcn_ballot_w64((unsigned)(8 * jl + 2 * t4 - a.pad + j) < (unsigned)a.W);
	const unsigned long long mfirst0 = __builtin_amdgcn_ballot_w64(2 * t4 - a.pad >= 0);
	mz[0] = __builtin_amdgcn_ballot_w64(8 * jl + 2 * t4 < a.Q), mz[1] = __builtin_amdgcn_ballot_w64(8 * jl + 2 * t4 + 1 < a.Q);

	const unsigned vdst = (unsigned)(((t4 >> 1) * CB + cl) * 2 + (t4 & 1)) + (unsigned)(hf * 8) * (2 * CB * 2);
	const unsigned zdst = (unsigned)kVFloats + (unsigned)(((t4 >> 1) * KB + kl) * 2 + (t4 & 1));

	// chunk g -> (image, tile row, chunk in row), advanced incrementally (scalar)
	struct Geo {
		int n, ty, j;
	};
	auto geo_of = [&](int g) {
		Geo q;
		q.j = g % a.TX4;
		const int r = g / a.TX4;
		q.ty = r % a.TY, q.n = r / a.TY;
		return q;
	};
	auto advance = [&](Geo &q) {
		if (++q.j == a.TX4) {
			q.j = 0;
			if (++q.ty == a.TY) q.ty = 0, ++q.n;
		}
	};

	f32x4 sp[3];
	f32x2 sz[2];
	struct Pend {                                // what the transform of a staged chunk needs to know about it
		bool first, last, fixrow;
	};

	auto issue_loads = [&](const Geo &q, Pend &pd) {
		pd.first = q.j == 0, pd.last = q.j == jl, pd.fixrow = false;
#pragma unroll
		for (int e = 0; e < 3; ++e) {
			const int row = 2 * q.ty - a.pad + hf + e;
			const bool ok = (unsigned)row < (unsigned)a.H;
			const long off = (((long)q.n * a.C) * a.H + row) * a.W + 8 * q.j - a.pad;
			if (ok && off < 0) {
				// the first row of the tensor with left padding: the scalar offset would be negative. Load one column further
				// right (the lane at the tensor's first element from that element) and shift that lane when it is consumed.
				pd.fixrow = true;
				sp[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, voffx_first, 0, 0));
			} else {
				sp[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ok ? xr : nullr, voffx, ok ? (unsigned)off * 4u : 0u, 0));
			}
		}
#pragma unroll
		for (int r = 0; r < 2; ++r) {
			const int row = 2 * q.ty + r;
			const bool ok = row < a.P;
			const long off = (((long)q.n * a.K) * a.P + row) * a.Q + 8 * q.j;
			sz[r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(ok ? zr : nullr, voffz, ok ? (unsigned)off * 4u : 0u, 0));
		}
	};

	f32x2 tA[2], tB[2], zr1, zr2;
	auto store_slice = [&](auto half, float *stg, const Pend &pd, int slice) {
		constexpr int HF = decltype(half)::value;
		const unsigned long long all = ~0ull;
		if (slice == 0) {
			if (pd.fixrow) {                         // wave-uniform, one chunk of the launch
				const int e = a.pad - HF;
				if (voffx == 0 && e >= 0 && e < 3) sp[e] = f32x4{0.f, sp[e][0], sp[e][1], sp[e][2]};
			}
			const unsigned long long m = (pd.last ? mlast[0] : all) & (pd.first ? mfirst0 : all);
#pragma unroll
			for (int e = 0; e < 3; ++e) sp[e][0] = sel_mask(m, sp[e][0]);
			const unsigned long long m0 = pd.last ? mz[0] : all, m1 = pd.last ? mz[1] : all;
#pragma unroll
			for (int r = 0; r < 2; ++r) sz[r][0] = sel_mask(m0, sz[r][0]), sz[r][1] = sel_mask(m1, sz[r][1]);
		} else if (slice == 2 || slice == 3) {
			const unsigned long long m = pd.last ? mlast[slice] : all;
#pragma unroll
			for (int e = 0; e < 3; ++e) sp[e][slice] = sel_mask(m, sp[e][slice]);
		}
		if (slice == 1 || slice == 3) {
			const int q = slice >> 1;
			const f32x2 e0 = {sp[0][2 * q], sp[0][2 * q + 1]}, e1 = {sp[1][2 * q], sp[1][2 * q + 1]};
			const f32x2 e2 = {sp[2][2 * q], sp[2][2 * q + 1]};
			if constexpr (HF == 0) {
				tA[q] = e0 - e2, tB[q] = e1 + e2;
			} else {
				tA[q] = e1 - e0, tB[q] = e0 - e2;
			}
		}
		if (slice == 1) zr1 = sz[0] + sz[1], zr2 = sz[0] - sz[1];       // rows of A dY: (d0, d0 + d1, d0 - d1, [-]d1)
		if (slice == 4 || slice == 5) {
			const f32x2 t01 = slice == 4 ? tA[0] : tB[0], t23 = slice == 4 ? tA[1] : tB[1];
			f32x2 v01, v23;
			asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(v01) : "v"(t01), "v"(t23));
			v23[0] = t23[0] - t01[1], v23[1] = t01[1] - t23[1];      // (scalar: a packed op whose LOW lane reads a HIGH half is not safe)
			float *dst = stg + vdst + (slice - 4) * 4 * (2 * CB * 2);
			dst[0 * (2 * CB * 2)] = v01[0];
			dst[1 * (2 * CB * 2)] = v01[1];
			dst[2 * (2 * CB * 2)] = v23[0];
			dst[3 * (2 * CB * 2)] = v23[1];
		}
		if (slice >= 4) {                            // row (a, b) of A dY -> (a, a + b, a - b, [-]b), one row per slice
			const int i = slice - 4;
			const f32x2 row = i == 0 ? sz[0] : i == 1 ? zr1 : i == 2 ? zr2 : sz[1];
			f32x2 mid;
			mid[0] = row[0] + row[1], mid[1] = row[0] - row[1];             // (scalar, see above)
			float *dst = stg + zdst + i * 4 * (2 * KB * 2);
			dst[0 * (2 * KB * 2)] = row[0];
			dst[1 * (2 * KB * 2)] = mid[0];
			dst[2 * (2 * KB * 2)] = mid[1];
			dst[3 * (2 * KB * 2)] = row[1];
		}
	};

	f32x16 acc[4][2];
#pragma unroll
	for (int p = 0; p < 4; ++p)
#pragma unroll
		for (int m = 0; m < 2; ++m)
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[p][m][r] = 0.f;

	const int vfrag = ((4 * wave * 2 + lhi) * CB + l31) * 2;
	const int ufrag = kVFloats + ((4 * wave * 2 + lhi) * KB + l31) * 2;

	struct Frag {
		f32x2 bv[2], av[2][2];
	};
	auto read_frags = [&](const float *stg, Frag &f, int p0) {
#pragma unroll
		for (int p = 0; p < 2; ++p) {
			f.bv[p] = *reinterpret_cast<const f32x2 *>(stg + vfrag + (p0 + p) * (2 * CB * 2));
			f.av[p][0] = *reinterpret_cast<const f32x2 *>(stg + ufrag + (p0 + p) * (2 * KB * 2));
			f.av[p][1] = *reinterpret_cast<const f32x2 *>(stg + ufrag + (p0 + p) * (2 * KB * 2) + 64);
		}
	};

	auto run = [&](auto half) {
		Frag f0, f1;
		Geo q = geo_of(g0);
		Pend pd;
		issue_loads(q, pd);
#pragma unroll
		for (int sl = 0; sl < 8; ++sl) store_slice(half, smem, pd, sl);
		if (nch > 1) {
			advance(q);
			issue_loads(q, pd);
#pragma unroll
			for (int sl = 0; sl < 8; ++sl) store_slice(half, smem + kStage, pd, sl);
		}
		int issued = nch > 1 ? 2 : 1;                // chunks whose loads have been issued
		if (nch > 2) advance(q), issue_loads(q, pd), ++issued;
		__syncthreads();
		read_frags(smem, f0, 0);

		int s_cur = 0;
		auto body = [&](Frag &cur, Frag &nxt) {
			const int s_nxt = s_cur == 2 ? 0 : s_cur + 1, s_wr = s_nxt == 2 ? 0 : s_nxt + 1;
			Frag late;
			read_frags(smem + s_cur * kStage, late, 2);
			read_frags(smem + s_nxt * kStage, nxt, 0);
			float *wr = smem + s_wr * kStage;
#pragma unroll
			for (int p = 0; p < 4; ++p)
#pragma unroll
				for (int m = 0; m < 2; ++m) {
					const Frag &f = p < 2 ? cur : late;
					acc[p][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.av[p & 1][m][0], f.bv[p & 1][0], acc[p][m], 0, 0, 0);
				}
			__builtin_amdgcn_sched_barrier(0);
			const Pend pc = pd;
#pragma unroll
			for (int p = 0; p < 4; ++p)
#pragma unroll
				for (int m = 0; m < 2; ++m) {
					const Frag &f = p < 2 ? cur : late;
					acc[p][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.av[p & 1][m][1], f.bv[p & 1][1], acc[p][m], 0, 0, 0);
					store_slice(half, wr, pc, p * 2 + m);
					__builtin_amdgcn_sched_barrier(0);
				}
			// past the last chunk: the same chunk again (its stores go to a stage nobody reads any more)
			if (issued < nch) advance(q), ++issued;
			issue_loads(q, pd);
			__builtin_amdgcn_sched_barrier(0);
			__syncthreads();
			s_cur = s_nxt;
		};

		int ch = 0;
		for (; ch + 1 < nch; ch += 2) {
			body(f0, f1);
			body(f1, f0);
		}
		if (ch < nch) body(f0, f1);
	};
	if (hf == 0)
		run(std::integral_constant<int, 0>{});
	else
		run(std::integral_constant<int, 1>{});

	// ---- partial sums -> slab[pos][k][c]: register i of a lane is row (k) 8 (i / 4) + 4 (lane / 32) + i % 4, column (c) lane % 32
	float *slab = a.slabs + (size_t)blockIdx.x * kSlab;
#pragma unroll
	for (int p = 0; p < 4; ++p)
#pragma unroll
		for (int m = 0; m < 2; ++m)
#pragma unroll
			for (int i = 0; i < 16; ++i) {
				const int kk = m * 32 + 8 * (i >> 2) + 4 * lhi + (i & 3);
				slab[((4 * wave + p) * KB + kk) * CB + l31] = acc[p][m][i];
			}
}

#if WN_WAVES == 8
// The same workgroup block on 8 waves (see wino_conv_kernel8): wave w accumulates positions 2w, 2w+1; waves 0-3 load and
// transform the patches, waves 4-7 the gradient tiles — half the loader work per thread, four waves per SIMD.
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) wino_wgrad_kernel8(WinoWgradArgs a) {
	constexpr int kStage = kVFloats + kUFloats;
	__shared__ __attribute__((aligned(16))) float smem[3 * kStage];

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int l31 = lane & 31, lhi = lane >> 5;

	const int nblk = a.kblocks * a.cblocks;
	const int split = blockIdx.x / nblk, blk = blockIdx.x - split * nblk;
	const int kb = blk / a.cblocks, cb = blk - kb * a.cblocks;
	const int g0 = (int)((long)a.chunks * split / a.splits), g1 = (int)((long)a.chunks * (split + 1) / a.splits);
	const int nch = g1 - g0;

	const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.x_bytes, 0x00020000);

	const int pw = wave & 3, hf = pw & 1, t4 = lane & 3;
	const int cl = (lane >> 2) + 16 * (pw >> 1), c = cb * CB + cl;
	const int kl = (lane >> 2) + 16 * pw, k = kb * KB + kl;
	const unsigned voffx = c < a.C ? (unsigned)(c * a.H * a.W + 2 * t4) * 4u : kOOB;
	const unsigned voffx_first = c < a.C ? (voffx == 0 ? 0u : voffx - 4u) : kOOB;
	const unsigned voffz = k < a.K ? (unsigned)(k * a.P * a.Q + 2 * t4) * 4u : kOOB;

	const int jl = a.TX4 - 1;
	unsigned long long mlast[4], mz[2];
#pragma unroll
	for (int j = 0; j < 4; ++j) mlast[j] = __builtin_amdgcn_ballot_w64((unsigned)(8 * jl + 2 * t4 - a.pad + j) < (unsigned)a.W);
	const unsigned long long mfirst0 = __builtin_amdgcn_ballot_w64(2 * t4 - a.pad >= 0);
	mz[0] = __builtin_amdgcn_ballot_w64(8 * jl + 2 * t4 < a.Q), mz[1] = __builtin_amdgcn_ballot_w64(8 * jl + 2 * t4 + 1 < a.Q);

	const unsigned vdst = (unsigned)(((t4 >> 1) * CB + cl) * 2 + (t4 & 1)) + (unsigned)(hf * 8) * (2 * CB * 2);
	const unsigned zdst = (unsigned)kVFloats + (unsigned)(((t4 >> 1) * KB + kl) * 2 + (t4 & 1));

	// chunk g -> (image, tile row, chunk in row) and the element offsets of its first patch row / gradient row, all kept
	// in scalar registers and advanced by additions (the scalar unit is shared by the CU's 16 waves: recomputing the
	// offsets with 64-bit multiplies per chunk saturated it)
	struct Geo {
		int n, ty, j;
		int row0;        // first patch row this wave loads: 2 ty - pad + hf
		int xo, zo;      // element offset of x(n, 0, row0, 8 j - pad) (negative only in front of the tensor) / dy(n, 0, 2 ty, 8 j)
		unsigned xsz[3], zsz[2];     // descriptor size word of each row load: the tensor's bytes, or 0 for a row outside the map
	};
	const int x_img = a.C * a.H * a.W, z_img = a.K * a.P * a.Q;
	auto set_rows = [&](Geo &q) {              // per tile row, not per chunk
#pragma unroll
		for (int e = 0; e < 3; ++e) q.xsz[e] = (unsigned)(q.row0 + e) < (unsigned)a.H ? a.x_bytes : 0u;
#pragma unroll
		for (int r = 0; r < 2; ++r) q.zsz[r] = 2 * q.ty + r < a.P ? a.dy_bytes : 0u;
	};
	auto geo_of = [&](int g) {
		Geo q;
		q.j = g % a.TX4;
		const int r = g / a.TX4;
		q.ty = r % a.TY, q.n = r / a.TY;
		q.row0 = 2 * q.ty - a.pad + hf;
		q.xo = q.n * x_img + q.row0 * a.W + 8 * q.j - a.pad;
		q.zo = q.n * z_img + 2 * q.ty * a.Q + 8 * q.j;
		set_rows(q);
		return q;
	};
	auto advance = [&](Geo &q) {
		if (++q.j == a.TX4) {
			q.j = 0;
			if (++q.ty == a.TY) {
				q.ty = 0, ++q.n;
				q.row0 = hf - a.pad;
				q.xo = q.n * x_img + q.row0 * a.W - a.pad, q.zo = q.n * z_img;
			} else {
				q.row0 += 2;
				q.xo += 2 * a.W - 8 * (a.TX4 - 1), q.zo += 2 * a.Q - 8 * (a.TX4 - 1);
			}
			set_rows(q);
		} else {
			q.xo += 8, q.zo += 8;
		}
	};

	f32x4 sp[3];
	f32x2 sz[2];
	struct Pend {
		bool first, last, fixrow;
	};

	// `head`: the launch's very first chunk (image 0, tile row 0, chunk 0) can start in front of the tensor; only the
	// prologue ever loads it, so the steady-state loop carries neither that test nor the shift of the affected lane
	auto issue_loads = [&](auto role, auto head, const Geo &q, Pend &pd) {
		pd.first = q.j == 0, pd.last = q.j == jl, pd.fixrow = false;
		if constexpr (decltype(role)::value == 0) {
#pragma unroll
			for (int e = 0; e < 3; ++e) {
				const bool ok = q.xsz[e] != 0u;
				const int off = q.xo + e * a.W;
				if (decltype(head)::value && ok && off < 0) {
					pd.fixrow = true;
					sp[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, voffx_first, 0, 0));
				} else {
					// a row outside the image reads through a zero-length descriptor (one scalar select of its size word)
					// (the scalar offset of a row outside the image is never used: zero records fail the range check first)
					const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, q.xsz[e], 0x00020000);
					sp[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voffx, (unsigned)off * 4u, 0));
				}
			}
		} else {
#pragma unroll
			for (int r = 0; r < 2; ++r) {
				const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.dy, 0, q.zsz[r], 0x00020000);
				sz[r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, voffz, (unsigned)(q.zo + r * a.Q) * 4u, 0));
			}
		}
	};

	// slice s rides behind MFMA s of the chunk; the staged registers are first needed behind the third MFMA
	f32x2 tA[2], tB[2], zr1, zr2;
	auto store_slice = [&](auto role, auto half, auto head, float *stg, const Pend &pd, int slice) {
		constexpr int HF = decltype(half)::value;
		const unsigned long long all = ~0ull;
		const int w = slice - 2;
		if (w < 0) return;
		if constexpr (decltype(role)::value == 0) {
			if (w == 0) {
				if (decltype(head)::value && pd.fixrow) {
					const int e = a.pad - HF;
					if (voffx == 0 && e >= 0 && e < 3) sp[e] = f32x4{0.f, sp[e][0], sp[e][1], sp[e][2]};
				}
				const unsigned long long m = (pd.last ? mlast[0] : all) & (pd.first ? mfirst0 : all);
#pragma unroll
				for (int e = 0; e < 3; ++e) sp[e][0] = sel_mask(m, sp[e][0]);
			} else if (w == 2 || w == 3) {
				const unsigned long long m = pd.last ? mlast[w] : all;
#pragma unroll
				for (int e = 0; e < 3; ++e) sp[e][w] = sel_mask(m, sp[e][w]);
			}
			if (w == 1 || w == 3) {
				const int q = w >> 1;
				const f32x2 e0 = {sp[0][2 * q], sp[0][2 * q + 1]}, e1 = {sp[1][2 * q], sp[1][2 * q + 1]};
				const f32x2 e2 = {sp[2][2 * q], sp[2][2 * q + 1]};
				if constexpr (HF == 0) {
					tA[q] = e0 - e2, tB[q] = e1 + e2;
				} else {
					tA[q] = e1 - e0, tB[q] = e0 - e2;
				}
			}
			if (w == 4 || w == 5) {
				const f32x2 t01 = w == 4 ? tA[0] : tB[0], t23 = w == 4 ? tA[1] : tB[1];
				f32x2 v01, v23;
				asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(v01) : "v"(t01), "v"(t23));
				v23[0] = t23[0] - t01[1], v23[1] = t01[1] - t23[1];      // (scalar: a packed op whose LOW lane reads a HIGH half is not safe)
				float *dst = stg + vdst + (w - 4) * 4 * (2 * CB * 2);
				dst[0 * (2 * CB * 2)] = v01[0];
				dst[1 * (2 * CB * 2)] = v01[1];
				dst[2 * (2 * CB * 2)] = v23[0];
				dst[3 * (2 * CB * 2)] = v23[1];
			}
		} else {
			if (w == 0) {
				const unsigned long long m0 = pd.last ? mz[0] : all, m1 = pd.last ? mz[1] : all;
#pragma unroll
				for (int r = 0; r < 2; ++r) sz[r][0] = sel_mask(m0, sz[r][0]), sz[r][1] = sel_mask(m1, sz[r][1]);
			} else if (w == 1) {
				zr1 = sz[0] + sz[1], zr2 = sz[0] - sz[1];
			} else {
				const int i = w - 2;                     // row (a, b) of A dY -> (a, a + b, a - b, [-]b)
				const f32x2 row = i == 0 ? sz[0] : i == 1 ? zr1 : i == 2 ? zr2 : sz[1];
				f32x2 mid;
				mid[0] = row[0] + row[1], mid[1] = row[0] - row[1];             // (scalar, see above)
				float *dst = stg + zdst + i * 4 * (2 * KB * 2);
				dst[0 * (2 * KB * 2)] = row[0];
				dst[1 * (2 * KB * 2)] = mid[0];
				dst[2 * (2 * KB * 2)] = mid[1];
				dst[3 * (2 * KB * 2)] = row[1];
			}
		}
	};

	f32x16 acc[2][2];
#pragma unroll
	for (int p = 0; p < 2; ++p)
#pragma unroll
		for (int m = 0; m < 2; ++m)
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[p][m][r] = 0.f;

	const int vfrag = ((2 * wave * 2 + lhi) * CB + l31) * 2;
	const int ufrag = kVFloats + ((2 * wave * 2 + lhi) * KB + l31) * 2;

	struct Frag {
		f32x2 bv[2], av[2][2];
	};
	auto read_frags = [&](const float *stg, Frag &f) {
#pragma unroll
		for (int p = 0; p < 2; ++p) {
			f.bv[p] = *reinterpret_cast<const f32x2 *>(stg + vfrag + p * (2 * CB * 2));
			f.av[p][0] = *reinterpret_cast<const f32x2 *>(stg + ufrag + p * (2 * KB * 2));
			f.av[p][1] = *reinterpret_cast<const f32x2 *>(stg + ufrag + p * (2 * KB * 2) + 64);
		}
	};

	auto run = [&](auto role, auto half) {
		Frag f0, f1;
		Geo q = geo_of(g0);
		Pend pd;
		issue_loads(role, std::true_type{}, q, pd);
#pragma unroll
		for (int sl = 0; sl < 8; ++sl) store_slice(role, half, std::true_type{}, smem, pd, sl);
		if (nch > 1) {
			advance(q);
			issue_loads(role, std::false_type{}, q, pd);
#pragma unroll
			for (int sl = 0; sl < 8; ++sl) store_slice(role, half, std::false_type{}, smem + kStage, pd, sl);
		}
		int issued = nch > 1 ? 2 : 1;
		if (nch > 2) advance(q), issue_loads(role, std::false_type{}, q, pd), ++issued;
		__syncthreads();
		read_frags(smem, f0);

		int o_cur = 0, o_nxt = kStage, o_wr = 2 * kStage;      // float offsets of the three stages, rotated by renaming
		auto body = [&](Frag &cur, Frag &nxt) {
			read_frags(smem + o_nxt, nxt);
			float *wr = smem + o_wr;
			const Pend pc = pd;
#pragma unroll
			for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
				for (int p = 0; p < 2; ++p)
#pragma unroll
					for (int m = 0; m < 2; ++m) {
						acc[p][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.av[p][m][s2], cur.bv[p][s2], acc[p][m], 0, 0, 0);
						store_slice(role, half, std::false_type{}, wr, pc, s2 * 4 + p * 2 + m);
						// the patch rows are consumed by slice 5 (the gradient rows by the last one): their next loads go out
						// two MFMAs earlier. Past the last chunk: the same chunk again, parked in a stage nobody reads any
						// more (were it the launch's first chunk, its row in front of the tensor would read as zeros — unused)
						if (s2 * 4 + p * 2 + m == (decltype(role)::value == 0 ? 5 : 7)) {
							if (issued < nch) advance(q), ++issued;
							issue_loads(role, std::false_type{}, q, pd);
						}
						__builtin_amdgcn_sched_barrier(0);
					}
			__syncthreads();
			const int t = o_cur;
			o_cur = o_nxt, o_nxt = o_wr, o_wr = t;
		};

		int ch = 0;
		for (; ch + 1 < nch; ch += 2) {
			body(f0, f1);
			body(f1, f0);
		}
		if (ch < nch) body(f0, f1);
	};
	using R0 = std::integral_constant<int, 0>;
	using R1 = std::integral_constant<int, 1>;
	if (wave >= 4)
		run(R1{}, R0{});
	else if (hf == 0)
		run(R0{}, R0{});
	else
		run(R0{}, R1{});

	float *slab = a.slabs + (size_t)blockIdx.x * kSlab;
#pragma unroll
	for (int p = 0; p < 2; ++p)
#pragma unroll
		for (int m = 0; m < 2; ++m)
#pragma unroll
			for (int i = 0; i < 16; ++i) {
				const int kk = m * 32 + 8 * (i >> 2) + 4 * lhi + (i & 3);
				slab[((2 * wave + p) * KB + kk) * CB + l31] = acc[p][m][i];
			}
}
#endif  // WN_WAVES == 8

// sums the slabs of `per` consecutive splits (blockIdx.y = group) into one: out[group][block][...]
__global__ void __launch_bounds__(256) wino_wgrad_sum_kernel(const float *__restrict__ slabs, float *__restrict__ out, size_t block_elems,
                                                             int splits, int groups) {
	const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (e >= block_elems) return;
	const int g = blockIdx.y;
	const int s0 = (int)((long)splits * g / groups), s1 = (int)((long)splits * (g + 1) / groups);
	float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
	int s = s0;
	for (; s + 4 <= s1; s += 4) {
		v0 += slabs[(size_t)s * block_elems + e], v1 += slabs[(size_t)(s + 1) * block_elems + e];
		v2 += slabs[(size_t)(s + 2) * block_elems + e], v3 += slabs[(size_t)(s + 3) * block_elems + e];
	}
	for (; s < s1; ++s) v0 += slabs[(size_t)s * block_elems + e];
	out[(size_t)g * block_elems + e] = (v0 + v1) + (v2 + v3);
}

// dg[k][c] = G^T dU G over the (at most a few) remaining partial slabs, dw = alpha dg + beta dw
__global__ void __launch_bounds__(256) wino_wgrad_finish_kernel(const float *__restrict__ part, int nparts, size_t block_elems, float *dw,
                                                                int K, int C, int kblocks, int cblocks, float alpha, float beta) {
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= K * C) return;
	const int k = idx / C, c = idx - k * C;
	const int kb = k / KB, kl = k - kb * KB, cb = c / CB, cl = c - cb * CB;
	const float *src = part + (size_t)(kb * cblocks + cb) * kSlab + kl * CB + cl;

	float u[4][4];
#pragma unroll
	for (int pos = 0; pos < 16; ++pos) u[pos >> 2][pos & 3] = 0.f;
	for (int s = 0; s < nparts; ++s)
#pragma unroll
		for (int pos = 0; pos < 16; ++pos) u[pos >> 2][pos & 3] += src[(size_t)s * block_elems + pos * (KB * CB)];
	// row / column 3 of A dY A^T were accumulated with the opposite sign
#pragma unroll
	for (int i = 0; i < 4; ++i) u[i][3] = -u[i][3], u[3][i] = -u[3][i];

	float t[3][4];
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		t[0][j] = u[0][j] + 0.5f * (u[1][j] + u[2][j]);
		t[1][j] = 0.5f * (u[1][j] - u[2][j]);
		t[2][j] = 0.5f * (u[1][j] + u[2][j]) + u[3][j];
	}
	float *dst = dw + (size_t)idx * 9;
#pragma unroll
	for (int r = 0; r < 3; ++r) {
		const float g0 = t[r][0] + 0.5f * (t[r][1] + t[r][2]), g1 = 0.5f * (t[r][1] - t[r][2]), g2 = 0.5f * (t[r][1] + t[r][2]) + t[r][3];
		if (beta == 0.f) {
			dst[r * 3 + 0] = alpha * g0, dst[r * 3 + 1] = alpha * g1, dst[r * 3 + 2] = alpha * g2;
		} else {
			dst[r * 3 + 0] = alpha * g0 + beta * dst[r * 3 + 0];
			dst[r * 3 + 1] = alpha * g1 + beta * dst[r * 3 + 1];
			dst[r * 3 + 2] = alpha * g2 + beta * dst[r * 3 + 2];
		}
	}
}

}  // namespace

namespace pz {

static void wino_dims(const pz_conv_desc *d, int which, int P, int Q, int *prod, int *red) {
	*prod = which == PZ_CONV_FWD ? d->k : d->c;
	*red = which == PZ_CONV_FWD ? d->c : d->k;
	(void)P, (void)Q;
}

bool wino_eligible(const pz_conv_desc *d, int which, int P, int Q) {
	if (which != PZ_CONV_FWD && which != PZ_CONV_BWD_DATA) return false;
	// 5x5 filters with pad 2 (NiN's 96 -> 192 layer, TestLib/CnnCifar10NIN.py:13-49): F(2x2, 5x5) on the F(4x4, 3x3) kernel of wino4.hip —
	// the same 6x6 patches and 36 positions, tile step 2, its own filter and output transforms. Written in round 6 without a
	// device to run it on: opt-in (PUZZLE_MI355_WINO5=1) until it has been verified and timed.
	const bool five = d->r == 5 && d->s == 5;
	if (five) {
		static const bool on = [] { const char *e = getenv("PUZZLE_MI355_WINO5"); return e && atoi(e) != 0; }();
		if (!on || d->pad_h != 2 || d->pad_w != 2) return false;
	} else if (d->r != 3 || d->s != 3 || d->pad_h != d->pad_w || d->pad_h > 1) {
		return false;
	}
	if (d->stride_h != 1 || d->stride_w != 1 || d->dil_h != 1 || d->dil_w != 1 || d->groups != 1) return false;
	int prod, red;
	wino_dims(d, which, P, Q, &prod, &red);
	if (red % BC != 0) return false;
	// backward-data of an unpadded layer pads by 2: the F(2x2) kernel's patch code assumes at most one padding column (its
	// second column is never masked, the first tensor row is shifted by one) — only the F(4x4) kernel serves that case
	if (which == PZ_CONV_BWD_DATA && d->pad_h == 0 && !wino4_pick(d, which, P, Q)) return false;
	const size_t lim = 0xfffffff0u;
	return (size_t)d->n * d->c * d->h * d->w * 4 < lim && (size_t)d->n * d->k * P * Q * 4 < lim;
}

size_t wino_workspace_bytes(const pz_conv_desc *d, int which, int P, int Q) {
	if (wino4_pick(d, which, P, Q)) return wino4_workspace_bytes(d, which);
	int prod, red;
	wino_dims(d, which, P, Q, &prod, &red);
	return (size_t)ceil_div(prod, KB) * (red / BC) * kUFloats * sizeof(float);
}

// which = PZ_CONV_FWD: out(N,K,P,Q) = conv(in(N,C,H,W), w) + bias;  PZ_CONV_BWD_DATA: out(N,C,H,W) = conv^T(in(N,K,P,Q), w)
int wino_stats_strips(const pz_conv_desc *d, int P, int Q) {
	if (wino4_pick(d, PZ_CONV_FWD, P, Q)) return wino4_stats_strips(d, P, Q);
#if WN_WAVES == 8
	return ceil_div((long)d->n * ((P + 1) / 2) * ((Q + 1) / 2), TB);
#else
	return 0;
#endif
}

static WinoFilterArgs wino_filter_args(const pz_conv_desc *d, int which, int P, int Q, const float *w, float *u) {
	int prod, red;
	wino_dims(d, which, P, Q, &prod, &red);
	WinoFilterArgs fa{};
	fa.w = w, fa.u = u, fa.mode = which == PZ_CONV_FWD ? 0 : 1;
	fa.K = d->k, fa.C = d->c, fa.prod = prod, fa.red = red;
	fa.kblocks = ceil_div(prod, KB), fa.chunks = red / BC;
	return fa;
}

int wino_filter_batch(const pz_conv_desc *const *descs, const int *which, const float *const *w, float *const *u, int n_all, hipStream_t st) {
	WinoFilterBatch b{};
	const pz_conv_desc *d4[kWinoBatch];
	int which4[kWinoBatch], n4 = 0, n = 0;
	const float *w4[kWinoBatch];
	float *u4[kWinoBatch];
	for (int i = 0; i < n_all; ++i) {
		int P, Q;
		P = (descs[i]->h + 2 * descs[i]->pad_h - 3) + 1, Q = (descs[i]->w + 2 * descs[i]->pad_w - 3) + 1;      // 3x3, stride 1, undilated
		if (wino4_pick(descs[i], which[i], P, Q)) {
			d4[n4] = descs[i], which4[n4] = which[i], w4[n4] = w[i], u4[n4] = u[i], ++n4;
			continue;
		}
		b.job[n] = wino_filter_args(descs[i], which[i], P, Q, w[i], u[i]);
		const long ftotal = (long)b.job[n].kblocks * b.job[n].chunks * KB * BC;
		b.start[n + 1] = b.start[n] + stream_grid(ftotal, 256);
		++n;
	}
	b.n = n;
	if (int rc = wino4_filter_batch(d4, which4, w4, u4, n4, st)) return rc;
	if (n == 0) return PZ_OK;
	wino_filter_batch_kernel<<<b.start[n], 256, 0, st>>>(b);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

size_t wino_input_bytes(const pz_conv_desc *d, int which, int P, int Q) { return wino4_input_bytes(d, which, P, Q); }

int wino_bnstats_strips(const pz_conv_desc *d, int P, int Q) {
	// only the F(4x4) kernel's backward-data epilogue sums them: one strip per block of 32 output tiles of the INPUT map
	if (!wino_eligible(d, PZ_CONV_BWD_DATA, P, Q) || !wino4_pick(d, PZ_CONV_BWD_DATA, P, Q)) return 0;
	const int m = d->r == 5 ? 2 : 4;
	return ceil_div((long)d->n * ((d->h + m - 1) / m) * ((d->w + m - 1) / m), 32);
}

int wino_conv(const pz_conv_desc *d, int which, int P, int Q, const float *in, const float *w, const float *bias, float *out,
              void *workspace, hipStream_t st, float *stats, bool filters_ready, void *vscratch, const BnStatsOut *bst) {
	if (wino4_pick(d, which, P, Q)) return wino4_conv(d, which, P, Q, in, w, bias, out, workspace, st, stats, filters_ready, vscratch, bst);
	PZ_REQUIRE(bst == nullptr, "wino_conv: the F(2x2) kernel has no statistics epilogue for a BatchNorm in front of the layer");
	int prod, red;
	wino_dims(d, which, P, Q, &prod, &red);

	WinoFilterArgs fa = wino_filter_args(d, which, P, Q, w, (float *)workspace);
	if (!filters_ready) {
		const long ftotal = (long)fa.kblocks * fa.chunks * KB * BC;
		wino_filter_kernel<<<stream_grid(ftotal, 256), 256, 0, st>>>(fa);
		PZ_LAUNCH_CHECK();
	}

	WinoArgs a{};
	a.x = in, a.u = (const float *)workspace, a.bias = bias, a.y = out;
	a.N = d->n, a.C = red, a.K = prod;
	if (which == PZ_CONV_FWD) {
		a.H = d->h, a.W = d->w, a.P = P, a.Q = Q, a.pad_h = d->pad_h, a.pad_w = d->pad_w;
	} else {
		a.H = P, a.W = Q, a.P = d->h, a.Q = d->w, a.pad_h = 2 - d->pad_h, a.pad_w = 2 - d->pad_w;
	}
	a.TY = (a.P + 1) / 2, a.TX = (a.Q + 1) / 2, a.tiles = a.N * a.TY * a.TX;
	a.chunks = fa.chunks, a.tblocks = ceil_div(a.tiles, TB);
	a.x_bytes = (unsigned)((size_t)a.N * a.C * a.H * a.W * 4);
	a.y_bytes = (unsigned)((size_t)a.N * a.K * a.P * a.Q * 4);
	a.stats = reinterpret_cast<float4 *>(stats);
#if WN_WAVES == 8
	wino_conv_kernel8<<<a.tblocks * fa.kblocks, 512, 0, st>>>(a);
#elif WN_SELFWAVE
	wino_conv_kernel_sw<<<a.tblocks * fa.kblocks, 256, 0, st>>>(a);
#else
	wino_conv_kernel<<<a.tblocks * fa.kblocks, 256, 0, st>>>(a);
#endif
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

// ---- backward-filter
struct WgradPlanW {
	int TY, TX4, chunks, kblocks, cblocks, splits, groups;
	size_t slab_bytes, part_bytes;
};

static WgradPlanW wino_wgrad_plan(const pz_conv_desc *d, int P, int Q) {
	WgradPlanW p{};
	p.TY = (P + 1) / 2, p.TX4 = ((Q + 1) / 2 + 3) / 4;
	p.chunks = d->n * p.TY * p.TX4;
	p.kblocks = ceil_div(d->k, KB), p.cblocks = ceil_div(d->c, CB);
	const int nblk = p.kblocks * p.cblocks;
	int splits = ceil_div(2 * kNumCU, nblk);                     // two workgroups per CU
	const int most = p.chunks / 16 > 0 ? p.chunks / 16 : 1;      // ... of at least 16 chunks each
	p.splits = splits < 1 ? 1 : (splits > most ? most : splits);
	p.groups = p.splits > 8 ? 8 : p.splits;
	p.slab_bytes = (size_t)p.splits * nblk * kSlab * sizeof(float);
	p.part_bytes = p.splits > p.groups ? (size_t)p.groups * nblk * kSlab * sizeof(float) : 0;
	return p;
}

bool wino_wgrad_eligible(const pz_conv_desc *d, int P, int Q) {
	if (d->r != 3 || d->s != 3 || d->stride_h != 1 || d->stride_w != 1 || d->dil_h != 1 || d->dil_w != 1 || d->groups != 1) return false;
	if (d->pad_h != d->pad_w || d->pad_h > 1) return false;
	const size_t lim = 0xfffffff0u;
	return (size_t)d->n * d->c * d->h * d->w * 4 < lim && (size_t)d->n * d->k * P * Q * 4 < lim;
}

size_t wino_wgrad_workspace_bytes(const pz_conv_desc *d, int P, int Q) {
	const WgradPlanW p = wino_wgrad_plan(d, P, Q);
	return p.slab_bytes + p.part_bytes;
}

int wino_wgrad(const pz_conv_desc *d, int P, int Q, const float *x, const float *dy, float *dw, float alpha, float beta,
               void *workspace, hipStream_t st) {
	const WgradPlanW p = wino_wgrad_plan(d, P, Q);
	const int nblk = p.kblocks * p.cblocks;

	WinoWgradArgs a{};
	a.x = x, a.dy = dy, a.slabs = (float *)workspace;
	a.N = d->n, a.C = d->c, a.H = d->h, a.W = d->w, a.K = d->k, a.P = P, a.Q = Q, a.pad = d->pad_h;
	a.TY = p.TY, a.TX4 = p.TX4, a.chunks = p.chunks, a.splits = p.splits, a.kblocks = p.kblocks, a.cblocks = p.cblocks;
	a.x_bytes = (unsigned)((size_t)d->n * d->c * d->h * d->w * 4);
	a.dy_bytes = (unsigned)((size_t)d->n * d->k * P * Q * 4);
#if WN_WAVES == 8
	wino_wgrad_kernel8<<<p.splits * nblk, 512, 0, st>>>(a);
#else
	wino_wgrad_kernel<<<p.splits * nblk, 256, 0, st>>>(a);
#endif
	PZ_LAUNCH_CHECK();

	const size_t block_elems = (size_t)nblk * kSlab;
	const float *part = a.slabs;
	int nparts = p.splits;
	if (p.splits > p.groups) {
		float *out = (float *)((char *)workspace + p.slab_bytes);
		wino_wgrad_sum_kernel<<<dim3((unsigned)((block_elems + 255) / 256), p.groups), 256, 0, st>>>(a.slabs, out, block_elems, p.splits, p.groups);
		PZ_LAUNCH_CHECK();
		part = out, nparts = p.groups;
	}
	wino_wgrad_finish_kernel<<<ceil_div((long)d->k * d->c, 256), 256, 0, st>>>(part, nparts, block_elems, dw, d->k, d->c, p.kblocks,
	                                                                          p.cblocks, alpha, beta);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

}  // namespace pz
