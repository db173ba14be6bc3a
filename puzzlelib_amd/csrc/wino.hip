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

#ifndef WN_ABL
#define WN_ABL 0        // timing-only ablations (wrong results): 1 no global loads, 2 no LDS stores, 4 no MFMAs, 8 no epilogue
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

__global__ void __launch_bounds__(256) wino_filter_kernel(WinoFilterArgs a) {
	const long total = (long)a.kblocks * a.chunks * KB * BC;
	for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
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
};

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
			asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(v23) : "v"(t23), "v"(t01));
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
#if !(WN_ABL & 4)
#pragma unroll
			for (int p = 0; p < 4; ++p)
#pragma unroll
				for (int m = 0; m < 2; ++m) {
					const Frag &f = p < 2 ? cur : late;
					acc[p][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.av[p & 1][m][0], f.bv[p & 1][0], acc[p][m], 0, 0, 0);
				}
#endif
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int p = 0; p < 4; ++p)
#pragma unroll
				for (int m = 0; m < 2; ++m) {
					const Frag &f = p < 2 ? cur : late;
#if !(WN_ABL & 4)
					acc[p][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.av[p & 1][m][1], f.bv[p & 1][1], acc[p][m], 0, 0, 0);
#endif
#if !(WN_ABL & 2)
					store_slice(half, fixed, wr, p * 2 + m);
#endif
					__builtin_amdgcn_sched_barrier(0);
				}
#if !(WN_ABL & 1)
			issue_loads(min(ch + 3, a.chunks - 1));
#endif
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

#if WN_ABL & 8
	{
		float sum = 0.f;
#pragma unroll
		for (int p = 0; p < 4; ++p) sum += acc[p][0][p] + acc[p][1][15 - p];
		a.y[(size_t)blockIdx.x * 256 + tid] = sum;
		return;
	}
#endif

	// ---- epilogue: D[m = channel][n = tile]; register i of a lane is row 8 (i / 4) + 4 (lane / 32) + i % 4, column lane % 32
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

}  // namespace

namespace pz {

static void wino_dims(const pz_conv_desc *d, int which, int P, int Q, int *prod, int *red) {
	*prod = which == PZ_CONV_FWD ? d->k : d->c;
	*red = which == PZ_CONV_FWD ? d->c : d->k;
	(void)P, (void)Q;
}

bool wino_eligible(const pz_conv_desc *d, int which, int P, int Q) {
	if (which != PZ_CONV_FWD && which != PZ_CONV_BWD_DATA) return false;
	if (d->r != 3 || d->s != 3 || d->stride_h != 1 || d->stride_w != 1 || d->dil_h != 1 || d->dil_w != 1 || d->groups != 1) return false;
	if (d->pad_h != d->pad_w || d->pad_h > 1) return false;      // the shifted-row fix-up of the first tensor row assumes one padding column
	int prod, red;
	wino_dims(d, which, P, Q, &prod, &red);
	if (red % BC != 0) return false;
	const size_t lim = 0xfffffff0u;
	return (size_t)d->n * d->c * d->h * d->w * 4 < lim && (size_t)d->n * d->k * P * Q * 4 < lim;
}

size_t wino_workspace_bytes(const pz_conv_desc *d, int which, int P, int Q) {
	int prod, red;
	wino_dims(d, which, P, Q, &prod, &red);
	return (size_t)ceil_div(prod, KB) * (red / BC) * kUFloats * sizeof(float);
}

// which = PZ_CONV_FWD: out(N,K,P,Q) = conv(in(N,C,H,W), w) + bias;  PZ_CONV_BWD_DATA: out(N,C,H,W) = conv^T(in(N,K,P,Q), w)
int wino_conv(const pz_conv_desc *d, int which, int P, int Q, const float *in, const float *w, const float *bias, float *out,
              void *workspace, hipStream_t st) {
	int prod, red;
	wino_dims(d, which, P, Q, &prod, &red);

	WinoFilterArgs fa{};
	fa.w = w, fa.u = (float *)workspace, fa.mode = which == PZ_CONV_FWD ? 0 : 1;
	fa.K = d->k, fa.C = d->c, fa.prod = prod, fa.red = red;
	fa.kblocks = ceil_div(prod, KB), fa.chunks = red / BC;
	const long ftotal = (long)fa.kblocks * fa.chunks * KB * BC;
	wino_filter_kernel<<<stream_grid(ftotal, 256), 256, 0, st>>>(fa);
	PZ_LAUNCH_CHECK();

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
	wino_conv_kernel<<<a.tblocks * fa.kblocks, 256, 0, st>>>(a);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

}  // namespace pz
