// fp32 convolution for gfx950 on the f32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32).
//
//   forward      y[n,k,p,q]  = sum_{c,r,s} x[n,c,p*st+r*d-pad, ...] * w[k,c,r,s]        implicit GEMM  M=K  N=N*P*Q  Kred=C*R*S
//   backward-data            = the same kernel run on dy with re-packed (flipped / stride-class) filters,
//                              one launch per stride residue class, each writing its own output pixels
//   backward-filter dw[k,crs]= sum_{n,p,q} dy[n,k,p,q] * x[n,c,...]                      implicit GEMM  M=K  N=C*R*S  Kred=N*P*Q
//                              split over Kred across workgroups, deterministic slab reduce with the
//                              beta*dw + alpha*sum epilogue (the reference's extra addKer pass, MIOpen.py:432-433, fused)
//
// Tiling (64-lane waves): a workgroup is 4 waves; each wave owns a (32*TM)x(32*TN) accumulator made of
// 32x32 MFMA tiles. Operands are staged through LDS in layouts whose MFMA fragment reads (lane l reads
// row/col l&31 at k = l>>5) are bank-conflict free: [k][m] (forward / backward-data, 4-byte reads) and
// [run][half][row]{2 pixels} (backward-filter, 8-byte reads = two k2-steps). im2col never exists in memory:
// in forward / backward-data each lane owns one pixel column of the B tile and the reduction runs tap-major
// (k = tap*C + c: one mask test and one per-lane offset per k-tile, the channel offset travels as a scalar);
// backward-filter gathers 16-byte runs of 4 output pixels for both operands. Epilogues of contiguous outputs
// go through LDS to store 16 bytes per lane and can leave per-channel strip sums for a following BatchNorm.
//
// Replaces DnnContext.convNd / convNdBackwardData / convNdBackwardParams — Hip/Wrappers/MIOpen.py:333-462.
#include "common.h"

#include <array>
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

// structural knobs with measured-equal alternatives. The timing-only ablation rig (PZ_ABL & co.) is
// not part of the shipped sources: tools/dev/measurement_rig.patch puts it back onto a scratch copy (tools/ablate.sh).
#ifndef PZ_IG_LOAD_STEPS
#define PZ_IG_LOAD_STEPS 8        // k2-steps of a k-tile over which the next tile's global loads are spread
#endif
#ifndef PZ_WG_SETS
#define PZ_WG_SETS 1              // 2 = gathers issued two k-steps ahead (needs ~32 more registers; +1 % before the 16-byte LDS cells)
#endif
#ifndef PZ_WG_RUNS
#define PZ_WG_RUNS 8              // 4-pixel runs per backward-filter k-step: 8 (32 pixels, 2 workgroups/CU) or 4 (16 pixels, 4/CU)
#endif
#ifndef PZ_IG_TALL
#define PZ_IG_TALL 0              // 1: 256x128 implicit-GEMM tiles on 8 waves for layers with a multiple of 256 output rows
                                  // (half the pixel gathers per MFMA; measured 3-12 % slower on every such 1x1 layer)
#endif
#ifndef PZ_WG_WAVES
#define PZ_WG_WAVES 4             // waves per backward-filter workgroup: 4, or 8 for tiles of >= 8 MFMA tiles (measured equal, +-2 %)
#endif
#ifndef PZ_TAIL_MIN_GAIN
#define PZ_TAIL_MIN_GAIN 24
#endif
#ifndef PZ_LB
#define PZ_LB 4
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace pz {
// How the matrix pipe multiplies fp32 operands (pz_conv_math_set): 0 = v_mfma_f32_32x32x2_f32; 6 / 9 = every fp32 operand
// is split EXACTLY into three bf16 terms (24 significand bits = 8 + 8 + 8) when it is staged in LDS and the product runs as
// 6 / 9 v_mfma_f32_32x32x16_bf16 partial products with fp32 accumulation (gfx950 issues bf16 MFMA FLOP at 16x the fp32
// rate). Every partial product is exact in fp32; 9 terms reproduce a*b exactly, 6 leave out the three terms below
// 2^-23 |a||b| — see DESIGN.md section 3.1e and tools/probes/split_probe.hip for the measured error next to the f32 MFMA.
int g_conv_math = 0;
int conv_math() { return g_conv_math; }
}  // namespace pz

namespace {

// ---- optional launch-level profiling (pz_conv_profile_*): event pairs around the MFMA launches only
struct ProfRec {
	hipEvent_t a, b;
	int family;
	double flops;
};
bool g_prof_on = false;
std::vector<ProfRec> g_prof;

struct ProfScope {
	hipStream_t st;
	ProfRec rec;
	bool on;
	ProfScope(hipStream_t s, int family, double flops) : st(s), on(g_prof_on) {
		if (!on) return;
		rec.family = family, rec.flops = flops;
		(void)hipEventCreate(&rec.a);
		(void)hipEventCreate(&rec.b);
		(void)hipEventRecord(rec.a, st);
	}
	~ProfScope() {
		if (!on) return;
		(void)hipEventRecord(rec.b, st);
		g_prof.push_back(rec);
	}
};

constexpr int kPadTap = 63;                 // table sentinel of padded k rows: tap bit 63 is never set -> operand reads as 0
constexpr unsigned kOOB = 0xfffffff0u;      // buffer-load byte offset beyond every tensor: the hardware returns 0

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float buf_load_f32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
	return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// bijective XCD-aware remap (block b runs on XCD b % 8): every XCD walks a contiguous range of tiles, so the
// m-tiles that share one pixel panel are consumed back-to-back out of the same 4 MiB L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
	const int q = nblk / pz::kNumXCD, r = nblk % pz::kNumXCD;
	const int xcd = bid % pz::kNumXCD, idx = bid / pz::kNumXCD;
	return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ---- exact 3-way bf16 split of fp32 values: hi = RN_bf16(v), mid = RN_bf16(v - hi), lo = v - hi - mid. The two
// subtractions are exact and lo fits 8 significand bits, so v == hi + mid + lo exactly, with |mid| <= 2^-8 |v| and
// |lo| <= 2^-16 |v| (round-to-nearest keeps the terms small and their signs unbiased). Eight values -> three 16-byte
// cells of 8 bf16 each, the operand fragment of one lane of v_mfma_f32_32x32x16_bf16. v_cvt_pk_bf16_f32 converts a pair.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split3_cells(const float (&v)[8], u32x4 &hi, u32x4 &mid, u32x4 &lo) {
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		const float a0 = v[2 * q], a1 = v[2 * q + 1];
		const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a0, a1}, bf16x2));
		const float r0 = a0 - __builtin_bit_cast(float, h << 16), r1 = a1 - __builtin_bit_cast(float, h & 0xffff0000u);
		const unsigned m = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){r0, r1}, bf16x2));
		const float s0 = r0 - __builtin_bit_cast(float, m << 16), s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
		const unsigned l = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){s0, s1}, bf16x2));
		hi[q] = h, mid[q] = m, lo[q] = l;
	}
}

// ------------------------------------------------------------------------------------------------
// filter packing + k-table (one tiny launch in front of every implicit-GEMM launch)
// ------------------------------------------------------------------------------------------------
struct PackArgs {
	const float *w;      // (K, Cg, R, S)
	float *wp;           // [groups][kred_pad][mpad]  (m contiguous)
	int2 *tab;           // [kred_pad] {byte offset of the tap inside one image, tap index r*S+s (63 = padding)}
	int Kg, Cg, R, S, groups;
	int mode;            // 0: forward (m = out channel, kred = (c, r, s));  1: backward-data class (m = in channel, kred = (k, r'', s''))
	int M, mpad, kred, kred_pad;
	int a_h, a_w, st_h, st_w, Rc, Sc;      // backward-data residue class: taps r = a + st*(Rc-1-r'')
	int dil_h, dil_w;
	int in_h, in_w;                         // spatial dims of the tensor the GEMM gathers from
	// tap-major reduction order (reduction channels % 16 == 0): kred = tap * chans + channel, so the 16 rows of a k-tile
	// share one filter tap; `tab` then holds one int4 per k-tile {tap byte offset, tap index, first channel's byte offset}
	int tapmajor, chans;
};

__device__ __forceinline__ void pack_filter_body(const PackArgs &a, long first, long step) {
	const long total = (long)a.groups * a.kred_pad * a.mpad;
	const int RS = a.mode == 0 ? a.R * a.S : a.Rc * a.Sc;
	const int Sx = a.mode == 0 ? a.S : a.Sc;

	for (long i = first; i < total; i += step) {
		const int m = (int)(i % a.mpad);
		const long t = i / a.mpad;
		const int kr = (int)(t % a.kred_pad);
		const int g = (int)(t / a.kred_pad);

		float v = 0.f;
		if (m < a.M && kr < a.kred) {
			const int ch = a.tapmajor ? kr % a.chans : kr / RS, rs = a.tapmajor ? kr / a.chans : kr - ch * RS;
			const int rr = rs / Sx, ss = rs - rr * Sx;

			if (a.mode == 0) {
				v = a.w[(((long)(g * a.Kg + m) * a.Cg + ch) * a.R + rr) * a.S + ss];
			} else {
				const int r = a.a_h + a.st_h * (a.Rc - 1 - rr), s = a.a_w + a.st_w * (a.Sc - 1 - ss);
				v = a.w[(((long)(g * a.Kg + ch) * a.Cg + m) * a.R + r) * a.S + s];
			}
		}
		a.wp[i] = v;

		if (g == 0 && m == 0 && a.tapmajor) {
			if (kr % 16 == 0) {
				const int ch = kr % a.chans, rs = kr / a.chans;
				const int rr = rs / Sx, ss = rs - rr * Sx;
				reinterpret_cast<int4 *>(a.tab)[kr / 16] =
				    make_int4((rr * a.dil_h * a.in_w + ss * a.dil_w) * 4, rs, ch * a.in_h * a.in_w * 4, 0);
			}
		} else if (g == 0 && m == 0) {
			int2 e = make_int2(0, kPadTap);
			if (kr < a.kred) {
				const int ch = kr / RS, rs = kr - ch * RS;
				const int rr = rs / Sx, ss = rs - rr * Sx;
				e = make_int2((ch * a.in_h * a.in_w + rr * a.dil_h * a.in_w + ss * a.dil_w) * 4, rs);
			}
			a.tab[kr] = e;
		}
	}
}

__global__ void __launch_bounds__(256) pack_filter_kernel(PackArgs a) {
	pack_filter_body(a, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

// Several layers' filter operands in ONE launch (pz_conv2d_prepack): job j owns the blocks [start[j], start[j + 1])
constexpr int kPackBatch = 24;
struct PackBatch {
	int n, start[kPackBatch + 1];
	PackArgs job[kPackBatch];
};
static_assert(sizeof(PackBatch) <= 4000, "kernel arguments");

__global__ void __launch_bounds__(256) pack_filter_batch_kernel(PackBatch b) {
	int j = 0;
	while (j + 1 < b.n && (int)blockIdx.x >= b.start[j + 1]) ++j;
	const int nb = b.start[j + 1] - b.start[j];
	pack_filter_body(b.job[j], (long)(blockIdx.x - b.start[j]) * 256 + threadIdx.x, (long)nb * 256);
}

// The same filters for the split kernels (tap-major orders only): pre-split into bf16 terms and laid out as the 16-byte
// operand cells the kernel copies straight into LDS — wp16[g][k-tile][term 0..2][k half 0..1][m] = 8 bf16 of row m at
// k = 16 kt + 8 half + 0..7 (6 bytes per filter element instead of 4).
__global__ void __launch_bounds__(256) pack_filter_split_kernel(PackArgs a) {
	const int nkt = a.kred_pad / 16;
	const long total = (long)a.groups * nkt * 2 * a.mpad;
	const int RS = a.mode == 0 ? a.R * a.S : a.Rc * a.Sc;
	const int Sx = a.mode == 0 ? a.S : a.Sc;
	u32x4 *wp16 = reinterpret_cast<u32x4 *>(a.wp);

	for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
		const int m = (int)(i % a.mpad);
		long t = i / a.mpad;
		const int half = (int)(t & 1);
		t >>= 1;
		const int kt = (int)(t % nkt), g = (int)(t / nkt);

		float v[8];
#pragma unroll
		for (int e = 0; e < 8; ++e) {
			const int kr = kt * 16 + half * 8 + e;
			v[e] = 0.f;
			if (m < a.M && kr < a.kred) {
				const int ch = kr % a.chans, rs = kr / a.chans;
				const int rr = rs / Sx, ss = rs - rr * Sx;
				if (a.mode == 0) {
					v[e] = a.w[(((long)(g * a.Kg + m) * a.Cg + ch) * a.R + rr) * a.S + ss];
				} else {
					const int r = a.a_h + a.st_h * (a.Rc - 1 - rr), s = a.a_w + a.st_w * (a.Sc - 1 - ss);
					v[e] = a.w[(((long)(g * a.Kg + ch) * a.Cg + m) * a.R + r) * a.S + s];
				}
			}
		}
		u32x4 hi, mid, lo;
		split3_cells(v, hi, mid, lo);
		const long cell = (((long)g * nkt + kt) * 6 + half) * a.mpad + m;
		wp16[cell] = hi, wp16[cell + 2 * a.mpad] = mid, wp16[cell + 4 * a.mpad] = lo;

		if (g == 0 && m == 0 && half == 0) {
			const int kr = kt * 16;
			const int ch = kr % a.chans, rs = kr / a.chans;
			const int rr = rs / Sx, ss = rs - rr * Sx;
			reinterpret_cast<int4 *>(a.tab)[kt] = make_int4((rr * a.dil_h * a.in_w + ss * a.dil_w) * 4, rs, ch * a.in_h * a.in_w * 4, 0);
		}
	}
}

// table only (backward-filter: j = (c, r, s) column of the GEMM)
__global__ void __launch_bounds__(256) build_tab_kernel(int2 *tab, int n, int npad, int R, int S, int dil_h, int dil_w,
                                                        int in_h, int in_w) {
	const int j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= npad) return;
	int2 e = make_int2(0, (31 << 8) | 31);       // padding column: row/col bit 31 is never set
	if (j < n) {
		const int ch = j / (R * S), rs = j - ch * R * S;
		const int rr = rs / S, ss = rs - rr * S;
		e = make_int2((ch * in_h * in_w + rr * dil_h * in_w + ss * dil_w) * 4, (rr << 8) | ss);
	}
	tab[j] = e;
}

// host side: one table per (device, geometry), kept for the life of the process (a few KB each). The first request
// builds it on the caller's stream and waits for it, so that later launches on any stream find it complete.
static int2 *cached_wgrad_tab(int n, int npad, int R, int S, int dil_h, int dil_w, int in_h, int in_w, hipStream_t st) {
	static std::mutex mu;
	static std::map<std::array<int, 9>, int2 *> cache;
	int dev = 0;
	(void)hipGetDevice(&dev);
	const std::array<int, 9> key{dev, n, npad, R, S, dil_h, dil_w, in_h, in_w};
	std::lock_guard<std::mutex> lock(mu);
	auto it = cache.find(key);
	if (it != cache.end()) return it->second;
	int2 *tab = nullptr;
	if (hipMalloc(&tab, (size_t)npad * sizeof(int2)) != hipSuccess) return nullptr;
	build_tab_kernel<<<pz::ceil_div(npad, 256), 256, 0, st>>>(tab, n, npad, R, S, dil_h, dil_w, in_h, in_w);
	if (hipStreamSynchronize(st) != hipSuccess) return nullptr;
	cache.emplace(key, tab);
	return tab;
}

// ------------------------------------------------------------------------------------------------
// implicit-GEMM forward / backward-data kernel
// ------------------------------------------------------------------------------------------------
struct IgemmArgs {
	const float *x;       // gathered tensor (N, C_total, H, W)
	const float *wp;      // packed filters [groups][kred_pad][mpad]
	const int2 *tab;      // [kred_pad]
	const float *bias;    // per output channel or NULL
	float *y;             // (N, OC_total, OH, OW)
	int C_total, H, W, Cg;
	int M, mpad, kred_pad;
	int Pv, Qv, npix;                      // virtual output grid, npix = N*Pv*Qv
	int vs_h, vs_w, pad_h, pad_w;          // gather coordinate = p*vs - pad + r*dil
	int R, S, dil_h, dil_w;                // taps of the gathered problem (R*S <= 63)
	unsigned x_bytes, wp_bytes, y_bytes;   // extents for the buffer descriptors
	int OC_total, OH, OW, os_h, os_w, oo_h, oo_w;   // output coordinate = p*os + oo
	int tiles_m, tiles_n;
	// tail balancing: the first `full_tiles` tiles are whole workgroups; each of the remaining tiles is cut into
	// `tail_splits` k-slices whose partial accumulators go to `slabs` and are summed by igemm_tail_reduce_kernel
	int full_tiles, tail_splits;
	float *slabs;
	int tapmajor;                          // `tab` is the per-k-tile int4 table of the tap-major order
	int contig;                            // output pixel index == position inside the image (stride-1 output grid):
	                                       // the epilogue goes through LDS and stores 4 pixels (16 B) per lane
	// BNX kernels: the gathered tensor is not materialised — element (n, c, ...) is xcoef[c].x * x + xcoef[c].y * x2 +
	// xcoef[c].z (a BatchNorm backward applied while gathering its incoming gradient x, with x2 = the BN's input)
	const float *x2;
	const float4 *xcoef;
	float4 *stats;                         // optional [OC_total][stat_strips] {shift, sum(v-shift), sum((v-shift)^2), -} per
	int stat_strips;                       // (channel, 32*TN-pixel strip) for a following batch normalisation
	// epilogue of contiguous outputs (pz_conv2d_fwd_relu / pz_conv2d_bwd_data_gate): y = max(y, 0) after the bias; y = 0 where
	// gate <= 0 (gate: a tensor of y's shape — the output of the ReLU whose backward follows this backward-data pass)
	int relu;
	const float *gate;
	// XBN kernels: the gathered tensor is relu?(xbn[c].x * x + xbn[c].y) per channel c — a BatchNorm (+ in-place ReLU) whose
	// normalised output was never written (Conv2D 1x1 behind BatchNorm2D + Activation(relu), Models/Nets/ResNet.py:27-33)
	const float2 *xbn;
	int xbn_relu;
	// backward-data launches whose output is the gradient w.r.t. y = relu(gab[c].x * gx + gab[c].y) — the ReLU output of the
	// BatchNorm in FRONT of this layer, never written (pz_conv2d_bwd_data_bnstats): the epilogue, which holds the gradient
	// tile, reads the same tile of gx and leaves per channel row and 64-pixel strip {sum q, sum q * (gx - gmean[c])} with
	// q = dx * (y > 0) in gst[channel * stat_strips + strip] — the statistics pass of that BatchNorm's backward
	// (bn_bwd_stats_kernel<true>: two more reads of tensors of this size) disappears
	const float *gx;
	const float2 *gab;
	const float *gmean;
	float2 *gst;
};

// D[row][col] of one workgroup tile -> output tensor. col = lane&31 (pixel), row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// Buffer stores: the per-lane part of the address (pixel, lane half) is a 32-bit offset computed once per column block,
// the channel row is a scalar offset -> no per-store address arithmetic; rows beyond M and pixels outside the output
// get the out-of-range offset and are dropped by the hardware.
template <int BM, int BN, int WM, int WN, int TM, int TN>
__device__ __forceinline__ void igemm_store_tile(const IgemmArgs &a, int tm, int tn, int g, int wm, int wn, int lane,
                                                 f32x16 (&acc)[TM][TN], int only_i = -1) {
	const int l31 = lane & 31, lhi = lane >> 5;
	const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void *)a.y, 0, a.y_bytes, 0x00020000);
	const unsigned plane_bytes = (unsigned)(a.OH * a.OW) * 4u;
	const bool full_m = tm * BM + BM <= a.M;

#pragma unroll
	for (int j = 0; j < TN; ++j) {
		const int opix = tn * BN + wn * (32 * TN) + j * 32 + l31;
		unsigned voff = kOOB;
		if (opix < a.npix) {
			const int pq_sz = a.Pv * a.Qv;
			const int on = opix / pq_sz;
			const int opq = opix - on * pq_sz;
			const int op = opq / a.Qv, oq = opq - op * a.Qv;
			const int oh = op * a.os_h + a.oo_h, ow = oq * a.os_w + a.oo_w;
			if ((unsigned)oh < (unsigned)a.OH && (unsigned)ow < (unsigned)a.OW)
				voff = (unsigned)((((long)on * a.OC_total + (long)g * a.M + 4 * lhi) * a.OH + oh) * a.OW + ow) * 4u;
		}

#pragma unroll
		for (int i = 0; i < TM; ++i) {
			if (only_i >= 0 && i != only_i) continue;      // (the slab reduce gives each sub-tile row a workgroup of its own)
#pragma unroll
			for (int r = 0; r < 16; ++r) {
				const int ms = tm * BM + wm * (32 * TM) + i * 32 + (r & 3) + 8 * (r >> 2);     // wave-uniform row (lane half adds 4)
				float v = acc[i][j][r];
				if (a.bias) v += a.bias[g * a.M + min(ms + 4 * lhi, a.M - 1)];
				const unsigned vo = (full_m || ms + 4 * lhi < a.M) ? voff : kOOB;
				__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, vo, (unsigned)ms * plane_bytes, 0);
			}
			__builtin_amdgcn_sched_barrier(0);      // one 32x32 tile (16 accumulator registers) in flight at a time
		}
	}
}

// Epilogue through LDS for outputs whose pixel index is contiguous inside an image (forward, stride-1 backward-data).
// The MFMA result layout gives a lane one pixel of 16 scattered channel rows: stored directly that is 64 four-byte stores
// per lane and the texture-address unit, not HBM, bounds short-K layers. Each wave parks HALF a 32-row band of its
// 32*TN = 64 pixel strip in its own 4.25 KB of LDS ([16 rows][68]) and reads it back as [4 rows][16 x 4 pixels]: 16 B per
// lane, 256 contiguous bytes per channel row and store instruction. The length of that run is what the write path is
// sensitive to: planes of 55x55 / 14x14 / 7x7 floats start at 4- or 16-byte phases, so NO run is made of whole 128-byte
// lines, and tools/probes/mfma_store.hip (profiles/r04_store_alignment_probe.txt) times the 793 MB of the 64 -> 256 layer
// at 0.147 ms for line-aligned rows, 0.315 ms for 128-byte runs at 4-byte phase (every run is two partial lines) and
// 0.265 ms for 256-byte runs (head, one whole line, tail) — with 128 MFMAs per wave in front 0.242 / 0.369 / 0.290 ms.
// Groups that straddle two images or the end of the tensor fall back to 4-byte stores. With a.stats the same pass
// accumulates, per channel row and 64-pixel strip, shifted sums for the batch normalisation that follows (saves its
// statistics pass over y).
// sum over the 16 lanes of a DPP row, left in every lane: four v_add_f32 with lane-permuting operands (the vector pipe alone —
// __shfl_xor goes through the LDS pipe); fixed tree, deterministic
__device__ __forceinline__ float row16_sum(float v) {
	auto dpp = [](float x, auto ctrl) {
		return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
	};
	v += dpp(v, std::integral_constant<int, 0xB1>{});       // quad_perm [1, 0, 3, 2]
	v += dpp(v, std::integral_constant<int, 0x4E>{});       // quad_perm [2, 3, 0, 1]
	v += dpp(v, std::integral_constant<int, 0x141>{});      // row_half_mirror: the other quad of the half
	v += dpp(v, std::integral_constant<int, 0x140>{});      // row_mirror: the other half
	return v;
}

constexpr int kEpiStride = 68;
constexpr int kEpiFloatsPerWave = 16 * kEpiStride;

template <int BM, int BN, int WM, int WN, int TM, int TN>
__device__ __forceinline__ void igemm_store_tile_lds(const IgemmArgs &a, int tm, int tn, int g, int wm, int wn, int wave, int lane,
                                                     f32x16 (&acc)[TM][TN], float *smem, int only_i = -1) {
	static_assert(TN == 2, "a wave's strip is two 32-pixel MFMA tiles wide");
	const int l31 = lane & 31, lhi = lane >> 5;
	const int r4 = lane >> 4, c16 = lane & 15;
	float *scr = smem + wave * kEpiFloatsPerWave;
	const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void *)a.y, 0, a.y_bytes, 0x00020000);
	const int PQ = a.OH * a.OW;
	const int strip0 = tn * BN + wn * 64;                         // first pixel of this wave's strip

	const int opix = strip0 + c16 * 4;                            // this lane's 4 pixels
	const int n_img = opix / PQ, pq = opix - n_img * PQ;
	const int nvalid = min(4, a.npix - opix);                     // <= 0: beyond the tensor
	const bool whole = nvalid == 4 && pq + 3 < PQ;

#pragma unroll
	for (int i = 0; i < TM; ++i) {
		if (only_i >= 0 && i != only_i) continue;
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const int row_base = tm * BM + wm * (32 * TM) + i * 32 + 16 * h;      // first channel row of this half band
#pragma unroll
			for (int j = 0; j < 2; ++j)
#pragma unroll
				for (int r = 8 * h; r < 8 * h + 8; ++r) scr[((r & 3) + 8 * ((r >> 2) & 1) + 4 * lhi) * kEpiStride + j * 32 + l31] = acc[i][j][r];

			float st_shift[4], st_s1[4], st_s2[4];
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const int ch = row_base + 4 * k + r4;
				typedef float f32x4_alias __attribute__((ext_vector_type(4), may_alias));      // written as scalars, read as vectors
				f32x4 v = *reinterpret_cast<const f32x4_alias *>(&scr[(4 * k + r4) * kEpiStride + c16 * 4]);
				const bool row_ok = ch < a.M;
				if (a.bias) {
					const float bv = a.bias[g * a.M + min(ch, a.M - 1)];
					v[0] += bv, v[1] += bv, v[2] += bv, v[3] += bv;
				}

				if (a.relu) {                    // x * (x > 0), the element-wise kernel's own form (csrc/eltwise.hip OpRelu): same bits, signed zeros included
#pragma unroll
					for (int e = 0; e < 4; ++e) v[e] = v[e] * (v[e] > 0.f ? 1.f : 0.f);
				}

				if (a.gst) {                     // (wave-uniform) q = v * (relu(bn(gx)) > 0): sums for that BatchNorm's backward
					const int cg = g * a.M + min(ch, a.M - 1);
					const float2 ab = a.gab[cg];
					const float mu = a.gmean[cg];
					const __amdgpu_buffer_rsrc_t gxr = __builtin_amdgcn_make_buffer_rsrc((void *)a.gx, 0, a.y_bytes, 0x00020000);
					const unsigned gchan = (unsigned)(g * a.M + ch) * (unsigned)PQ;
					float s1 = 0.f, s2 = 0.f;
					if (whole) {
						const unsigned off = row_ok ? (((unsigned)n_img * a.OC_total) * (unsigned)PQ + gchan + pq) * 4u : kOOB;
						const f32x4 xv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(gxr, off, 0, 0));
#pragma unroll
						for (int e = 0; e < 4; ++e) {
							const float q = __builtin_fmaf(xv[e], ab.x, ab.y) > 0.f ? v[e] : 0.f;      // bn_gate<true>'s form
							s1 += q;
							s2 = __builtin_fmaf(q, xv[e] - mu, s2);
						}
					} else {
#pragma unroll
						for (int e = 0; e < 4; ++e) {
							const int o = opix + e;
							const int n2 = o / PQ, pq2 = o - n2 * PQ;
							const bool ok = row_ok && e < nvalid;
							const float xe = buf_load_f32(gxr, ok ? (((unsigned)n2 * a.OC_total) * (unsigned)PQ + gchan + pq2) * 4u : kOOB, 0);
							const float q = (ok && __builtin_fmaf(xe, ab.x, ab.y) > 0.f) ? v[e] : 0.f;
							s1 += q;
							s2 = __builtin_fmaf(q, xe - mu, s2);
						}
					}
					s1 = row16_sum(s1), s2 = row16_sum(s2);             // (every lane of the row holds them; its first lane stores)
					if (c16 == 0 && row_ok && strip0 < a.npix) a.gst[(size_t)(g * a.M + ch) * a.stat_strips + strip0 / 64] = make_float2(s1, s2);
				}

				if (a.stats) {                   // (wave-uniform)
					const float shift = __shfl(v[0], lane & ~15);      // first pixel of the strip, same for the row's 16 lanes
					float s1 = 0.f, s2 = 0.f;
#pragma unroll
					for (int e = 0; e < 4; ++e) {
						const float dlt = e < nvalid ? v[e] - shift : 0.f;
						s1 += dlt;
						s2 = __builtin_fmaf(dlt, dlt, s2);
					}
					st_shift[k] = shift, st_s1[k] = row16_sum(s1), st_s2[k] = row16_sum(s2);      // (every lane of the row holds them)
				}

				const unsigned chan_off = (unsigned)(g * a.M + ch) * (unsigned)PQ;
				if (whole) {
					const unsigned off = row_ok ? (((unsigned)n_img * a.OC_total) * (unsigned)PQ + chan_off + pq) * 4u : kOOB;
					if (a.gate) {                // (wave-uniform; out-of-range rows read zeros and are not stored)
						const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc((void *)a.gate, 0, a.y_bytes, 0x00020000);
						const f32x4 gt = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(gr, off, 0, 0));
#pragma unroll
						for (int e = 0; e < 4; ++e) v[e] = v[e] * (gt[e] > 0.f ? 1.f : 0.f);      // OpReluDer's form
					}
					__builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yr, off, 0, 0);
				} else {
					auto store_one = [&](int e, float val) {
						const int o = opix + e;
						const int n2 = o / PQ, pq2 = o - n2 * PQ;
						const unsigned off = (row_ok && e < nvalid) ? (((unsigned)n2 * a.OC_total) * (unsigned)PQ + chan_off + pq2) * 4u : kOOB;
						if (a.gate) {
							const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc((void *)a.gate, 0, a.y_bytes, 0x00020000);
							val = val * (buf_load_f32(gr, off, 0) > 0.f ? 1.f : 0.f);
						}
						__builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), yr, off, 0, 0);
					};
					store_one(0, v[0]), store_one(1, v[1]), store_one(2, v[2]), store_one(3, v[3]);
				}
			}

			if (a.stats) {                   // one store for the half band's 16 rows: lane (r4, c16 = k) writes row 4 k + r4
				float sh = st_shift[0], q1 = st_s1[0], q2 = st_s2[0];
#pragma unroll
				for (int k = 1; k < 4; ++k)
					if (c16 == k) sh = st_shift[k], q1 = st_s1[k], q2 = st_s2[k];
				const int ch = row_base + 4 * c16 + r4;
				if (c16 < 4 && ch < a.M && strip0 < a.npix)
					a.stats[(size_t)(g * a.M + ch) * a.stat_strips + strip0 / 64] = make_float4(sh, q1, q2, 0.f);
			}
		}
	}
}

// 4 waves (128x128, 64x256 tiles; 4 workgroups per CU) or 8 waves (256x128 tiles for >= 256 output rows: the gathered pixel
// panel serves twice as many rows, half the gathers per MFMA; 2 workgroups per CU = the same 4 waves per SIMD)
// PF2: the next-but-one k-tile's global loads are in flight while a k-tile is multiplied (two register sets, the loop is
// unrolled by two so that the sets stay static), the next tile is parked in LDS during k2-step 6, and the barrier sits
// BETWEEN the MFMAs of k2-step 7 with the next tile's first fragments read behind it: no wave waits for a load it issued a
// quarter of a tile ago, and the k-tile boundary (park, barrier, first fragment read) is in the shadow of MFMAs. Costs 16
// registers (3 waves per SIMD instead of 4); tools/probes/igemm_pipe.hip variant V3 measured it at +4..6 % from 3 to 24
// tiles per CU (profiles/r04_igemm_pipe_probe.txt). Same products in the same order: bit-identical results.
template <int BM, int BN, int WM, int WN, bool TAPMAJOR, bool BNX = false, bool PF2 = false, bool XBN = false>
__global__ void __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu((BNX && WM * WN == 4 && BM != 64) || PF2 ? 3 : 4, 8)))
igemm_conv_kernel(IgemmArgs a) {
	static_assert(!BNX || TAPMAJOR, "the BatchNorm-backward gather rides on the tap-major order");
	static_assert(!XBN || (TAPMAJOR && !BNX && !PF2), "the BatchNorm-forward gather: tap-major, plain loop");
	static_assert(!PF2 || (TAPMAJOR && !BNX && WM * WN == 4), "two-tiles-ahead loads: tap-major, plain gathers, 4 waves");
	constexpr int BK = 16, NT = 64 * WM * WN;
	constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
	static_assert((WM * WN == 4 || WM * WN == 8) && TM >= 1 && TN >= 1, "4 or 8 waves per workgroup");

	// one LDS block: operand tiles during the k loop, per-wave transposition scratch in the epilogue
	static_assert(2 * BK * (BM + BN) >= WM * WN * kEpiFloatsPerWave, "epilogue scratch does not fit the operand tiles");
	__shared__ __attribute__((aligned(16))) float smem[2 * BK * (BM + BN)];
	float(*As)[BK][BM] = reinterpret_cast<float(*)[BK][BM]>(smem);
	float(*Bs)[BK][BN] = reinterpret_cast<float(*)[BK][BN]>(smem + 2 * BK * BM);

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = wave / WN, wn = wave % WN;
	const int g = blockIdx.z;

	// whole tiles first (XCD-aware order), then the k-slices of the tail tiles
	int L, kslice = -1;
	if ((int)blockIdx.x < a.full_tiles) {
		L = xcd_remap(blockIdx.x, a.full_tiles);
	} else {
		const int t = blockIdx.x - a.full_tiles;
		L = a.full_tiles + t / a.tail_splits;
		kslice = t % a.tail_splits;
	}
	const int tm = L % a.tiles_m, tn = L / a.tiles_m;

	// ---- B (gathered pixels) loader: this thread owns one pixel column of the tile for the whole kernel.
	// Gathers are buffer loads: 32-bit byte offsets, and a tap that falls outside the image (or a pixel outside the
	// tensor) gets an out-of-range offset for which the hardware returns 0 — zero padding costs no select, no branch.
	constexpr int NB = BK / (NT / BN);        // k rows per thread; a wave covers NB consecutive rows
	const int jb = tid % BN;
	const int kb0 = __builtin_amdgcn_readfirstlane(tid / BN);

	const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.x_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *)a.wp, 0, a.wp_bytes, 0x00020000);

	const int pix = tn * BN + jb;
	unsigned long long tapmask = 0;           // bit r*S+s: tap (r, s) of this pixel lies inside the image
	unsigned base_bytes = 0;
	if (pix < a.npix) {
		const int pq_sz = a.Pv * a.Qv;
		const int n_img = pix / pq_sz;
		const int pq = pix - n_img * pq_sz;
		const int pp = pq / a.Qv, qq = pq - pp * a.Qv;
		const int h0 = pp * a.vs_h - a.pad_h, w0 = qq * a.vs_w - a.pad_w;

		for (int r = 0; r < a.R; ++r)
			for (int t = 0; t < a.S; ++t) {
				const bool ok = (unsigned)(h0 + r * a.dil_h) < (unsigned)a.H && (unsigned)(w0 + t * a.dil_w) < (unsigned)a.W;
				tapmask |= (unsigned long long)ok << (r * a.S + t);
			}
		base_bytes = (unsigned)((((long)n_img * a.C_total + (long)g * a.Cg) * a.H + h0) * a.W + w0) * 4u;
	}

	// ---- A (packed filters) loader: 16 B per thread, m contiguous; the k-tile advance is a scalar offset
	constexpr int NA = (BK * BM / 4) / NT;
	static_assert(NA >= 1, "A tile too small");
	unsigned voffA[NA];
#pragma unroll
	for (int i = 0; i < NA; ++i) {
		const int f = tid + i * NT;
		const int kk = f / (BM / 4), m4 = (f % (BM / 4)) * 4;
		voffA[i] = (unsigned)(((long)g * a.kred_pad + kk) * a.mpad + tm * BM + m4) * 4u;
	}

	f32x16 acc[TM][TN];
#pragma unroll
	for (int i = 0; i < TM; ++i)
#pragma unroll
		for (int j = 0; j < TN; ++j)
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

	constexpr int NSETS = PF2 ? 2 : 1;
	f32x4 ra[NSETS][NA];
	float rb[NSETS][NB];
	float rb2[BNX ? NB : 1];                  // BNX: the BatchNorm input elements that go with the gradient elements
	float4 bnc[BNX ? NB : 1];                 // ... and the coefficients of their channels (scalar loads)
	float2 xc[XBN ? NB : 1];                  // XBN: {a, b} of the gathered channels (scalar loads)
	const __amdgpu_buffer_rsrc_t x2r = __builtin_amdgcn_make_buffer_rsrc((void *)(BNX ? a.x2 : a.x), 0, a.x_bytes, 0x00020000);

	const int l31 = lane & 31, lhi = lane >> 5;
	int2 e[NB];

	// tap-major order: the NB gathers of a k-tile share the tap (one mask test, one per-lane offset) and differ by a
	// scalar channel offset, which travels in the buffer load's soffset — VALU instructions serialise with MFMAs
	// (tools/probes/lds_mfma.hip), so the loop keeps them to a handful per k-tile
	const unsigned mask_lo = (unsigned)tapmask, mask_hi = (unsigned)(tapmask >> 32);
	const unsigned hw4 = (unsigned)(a.H * a.W) * 4u;
	const unsigned row_off = (unsigned)(kb0 * NB) * hw4;      // this wave's first channel inside the k-tile
	unsigned voff_tile = kOOB, soff_tile = 0;

	// global -> register loads of k-tile `kt`, cut into BK/2 parts so that each part's address arithmetic and its
	// load issue can sit in the shadow of one k2-step's MFMAs (the matrix pipe is busy 4 x 64 cycles per k2-step)
	auto load_tab = [&](int kt) {
		if constexpr (TAPMAJOR) {
			// no table: a pointwise filter (one tap at offset 0, k-tile kt starts at channel 16 kt)
			const int4 t = a.tab ? reinterpret_cast<const int4 *>(a.tab)[kt] : make_int4(0, 0, (int)((unsigned)(kt * BK) * hw4), 0);
			const unsigned word = t.y < 32 ? mask_lo : mask_hi;
			voff_tile = (word >> (t.y & 31)) & 1u ? base_bytes + (unsigned)t.x : kOOB;
			soff_tile = (unsigned)t.z + row_off;
			if constexpr (BNX) {                  // channels of this k-tile: (16 kt) mod C onwards (C = reduction channels, % 16 == 0)
				const int ch0 = g * a.Cg + (kt * BK) % a.Cg + kb0 * NB;
#pragma unroll
				for (int i = 0; i < NB; ++i) bnc[i] = a.xcoef[ch0 + i];
			}
			if constexpr (XBN) {
				const int ch0 = g * a.Cg + (kt * BK) % a.Cg + kb0 * NB;
#pragma unroll
				for (int i = 0; i < NB; ++i) xc[i] = a.xbn[ch0 + i];
			}
		} else {
#pragma unroll
			for (int i = 0; i < NB; ++i) e[i] = a.tab[kt * BK + kb0 * NB + i];   // contiguous: one wide scalar load
		}
	};

	using Set0 = std::integral_constant<int, 0>;
	using Set1 = std::integral_constant<int, NSETS - 1>;
	auto load_part = [&](int kt, int j, auto set) {
		constexpr int SET = decltype(set)::value;
		constexpr int PER = NB >= BK / 2 ? NB / (BK / 2) : 1;          // gathers per k2-step (NB < 8: one every (BK/2)/NB steps)
		constexpr int EVERY = NB >= BK / 2 ? 1 : (BK / 2) / NB;
		if (j < NA)
			ra[SET][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, voffA[j], (unsigned)(kt * BK * a.mpad) * 4u, 0));
		if (j % EVERY != 0) return;
#pragma unroll
		for (int t = 0; t < PER; ++t) {
			const int i = (j / EVERY) * PER + t;
			if constexpr (TAPMAJOR) {
				rb[SET][i] = buf_load_f32(xr, voff_tile, soff_tile + (unsigned)i * hw4);
				if constexpr (BNX) rb2[i] = buf_load_f32(x2r, voff_tile, soff_tile + (unsigned)i * hw4);
			} else {
				const bool ok = (tapmask >> e[i].y) & 1ull;
				rb[SET][i] = buf_load_f32(xr, ok ? base_bytes + (unsigned)e[i].x : kOOB, 0);
			}
		}
	};

	auto store_tile = [&](int buf, auto set) {
		constexpr int SET = decltype(set)::value;
#pragma unroll
		for (int i = 0; i < NA; ++i) {
			const int f = tid + i * NT;
			const int kk = f / (BM / 4), m4 = (f % (BM / 4)) * 4;
			*reinterpret_cast<f32x4 *>(&As[buf][kk][m4]) = ra[SET][i];
		}
#pragma unroll
		for (int i = 0; i < NB; ++i) {
			float v = rb[SET][i];
			if constexpr (BNX) v = __builtin_fmaf(bnc[i].x, rb[SET][i], __builtin_fmaf(bnc[i].y, rb2[i], bnc[i].z));
			if constexpr (XBN) {                  // bn_apply_add_kernel's own expression (csrc/bn.hip): same bits as the written tensor
				v = __builtin_fmaf(rb[SET][i], xc[i].x, xc[i].y);
				if (a.xbn_relu) v = v > 0.f ? v : 0.f;
			}
			Bs[buf][kb0 * NB + i][jb] = v;
		}
	};

	auto read_frag = [&](int buf, int ks, float (&av)[TM], float (&bv)[TN]) {
#pragma unroll
		for (int i = 0; i < TM; ++i) av[i] = As[buf][ks + lhi][wm * (32 * TM) + i * 32 + l31];
#pragma unroll
		for (int j = 0; j < TN; ++j) bv[j] = Bs[buf][ks + lhi][wn * (32 * TN) + j * 32 + l31];
	};

	// one k-tile of MFMAs; fragments are double-buffered in registers (the LDS read of k2-step j+1 is issued before the
	// MFMAs of step j), the next tile's global loads are interleaved one part per k2-step
	auto compute_tile = [&](int buf, int kt_next, bool has_next) {
		float av[2][TM], bv[2][TN];
		read_frag(buf, 0, av[0], bv[0]);
		if (has_next) load_tab(kt_next);

#pragma unroll
		for (int j = 0; j < BK / 2; ++j) {
			if (j + 1 < BK / 2) read_frag(buf, 2 * (j + 1), av[(j + 1) & 1], bv[(j + 1) & 1]);
			if (has_next && j < PZ_IG_LOAD_STEPS) {
#pragma unroll
				for (int q = 0; q < (BK / 2) / PZ_IG_LOAD_STEPS; ++q) load_part(kt_next, j * ((BK / 2) / PZ_IG_LOAD_STEPS) + q, Set0{});
			}
			__builtin_amdgcn_sched_barrier(0);        // keep this step's LDS reads / gather ahead of its MFMAs ...
#pragma unroll
			for (int i = 0; i < TM; ++i)
#pragma unroll
				for (int jj = 0; jj < TN; ++jj)
					acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j & 1][i], bv[j & 1][jj], acc[i][jj], 0, 0, 0);
			__builtin_amdgcn_sched_barrier(0);        // ... and the next step's behind them (they run in the MFMA shadow)
		}
	};

	const int nk_all = a.kred_pad / BK;
	int kt0 = 0, kt1 = nk_all;
	if (kslice >= 0) {
		kt0 = (int)((long)nk_all * kslice / a.tail_splits);
		kt1 = (int)((long)nk_all * (kslice + 1) / a.tail_splits);
	}

	if constexpr (PF2) {
		float av[2][TM], bv[2][TN];
		// one k-tile out of LDS buffer `buf`; on entry the fragments of its k2-step 0 are in slot 0. lset: the register set
		// this tile's loads (of tile kt_load, two ahead) go to; pset: the set parked into the other buffer during k2-step 6
		auto tile_body = [&](int buf, auto lset, auto pset, int kt_load, bool do_load, bool do_park) {
#pragma unroll
			for (int j = 0; j < BK / 2; ++j) {
				if (j + 1 < BK / 2) read_frag(buf, 2 * (j + 1), av[(j + 1) & 1], bv[(j + 1) & 1]);
				if (do_load && j < 2) {
					if (j == 0) load_tab(kt_load);
#pragma unroll
					for (int q = 0; q < BK / 4; ++q) load_part(kt_load, j * (BK / 4) + q, lset);
				}
				if (do_park && j == BK / 2 - 2) store_tile(buf ^ 1, pset);
				__builtin_amdgcn_sched_barrier(0);
				if (j == BK / 2 - 1 && do_park) {
#pragma unroll
					for (int jj = 0; jj < TN; ++jj) acc[0][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j & 1][0], bv[j & 1][jj], acc[0][jj], 0, 0, 0);
					__builtin_amdgcn_sched_barrier(0);
					__syncthreads();                     // everyone's park is visible, everyone is done reading `buf`
					read_frag(buf ^ 1, 0, av[0], bv[0]);         // (slot 0; this k2-step's fragments are in slot 1)
					__builtin_amdgcn_sched_barrier(0);
#pragma unroll
					for (int i = 1; i < TM; ++i)
#pragma unroll
						for (int jj = 0; jj < TN; ++jj)
							acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j & 1][i], bv[j & 1][jj], acc[i][jj], 0, 0, 0);
				} else {
#pragma unroll
					for (int i = 0; i < TM; ++i)
#pragma unroll
						for (int jj = 0; jj < TN; ++jj)
							acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j & 1][i], bv[j & 1][jj], acc[i][jj], 0, 0, 0);
				}
				__builtin_amdgcn_sched_barrier(0);
			}
		};
		static_assert((BK / 2) % 2 == 0, "k2-step BK/2 - 1 is odd: its fragments sit in slot 1, the next tile's first in slot 0");

		const int nk = kt1 - kt0;
		load_tab(kt0);
#pragma unroll
		for (int j = 0; j < BK / 2; ++j) load_part(kt0, j, Set0{});
		store_tile(0, Set0{});
		if (nk > 1) {
			load_tab(kt0 + 1);
#pragma unroll
			for (int j = 0; j < BK / 2; ++j) load_part(kt0 + 1, j, Set1{});
		}
		__syncthreads();
		read_frag(0, 0, av[0], bv[0]);

		int t = 0;                                       // tile t: loads of tile t+2 into set t & 1, parks set (t+1) & 1
		for (; t + 2 < nk; t += 2) {
			tile_body(0, Set0{}, Set1{}, kt0 + t + 2, true, true);
			tile_body(1, Set1{}, Set0{}, kt0 + t + 3, t + 3 < nk, true);
		}
		if (nk - t == 2) {
			tile_body(0, Set0{}, Set1{}, 0, false, true);
			tile_body(1, Set1{}, Set0{}, 0, false, false);
		} else {
			tile_body(0, Set0{}, Set1{}, 0, false, false);
		}
	} else {
	load_tab(kt0);
#pragma unroll
	for (int j = 0; j < BK / 2; ++j) load_part(kt0, j, Set0{});
	store_tile(0, Set0{});
	__syncthreads();

	for (int kt = kt0; kt + 1 < kt1; ++kt) {
		const int buf = (kt - kt0) & 1;
		compute_tile(buf, kt + 1, true);
		store_tile(buf ^ 1, Set0{});
		__syncthreads();
	}
	compute_tile((kt1 - 1 - kt0) & 1, 0, false);
	}

	if (kslice < 0) {
		if (a.contig) {
			__syncthreads();          // every wave is done reading the operand tiles
			igemm_store_tile_lds<BM, BN, WM, WN, TM, TN>(a, tm, tn, g, wm, wn, wave, lane, acc, smem);
		} else {
			igemm_store_tile<BM, BN, WM, WN, TM, TN>(a, tm, tn, g, wm, wn, lane, acc);
		}
	} else {
		// partial accumulators of a tail slice: slab[(tail tile, slice)][register][thread] — coalesced 256-B rows
		float *slab = a.slabs + ((size_t)(blockIdx.x - a.full_tiles) + (size_t)g * (gridDim.x - a.full_tiles)) * (BM * BN);
#pragma unroll
		for (int i = 0; i < TM; ++i)
#pragma unroll
			for (int j = 0; j < TN; ++j)
#pragma unroll
				for (int r = 0; r < 16; ++r) slab[((i * TN + j) * 16 + r) * NT + tid] = acc[i][j][r];
	}
}

// ------------------------------------------------------------------------------------------------
// the same implicit GEMM on the bf16 matrix pipe: fp32 operands split exactly into three bf16 terms
// ------------------------------------------------------------------------------------------------
// Tap-major problems only (reduction channels in whole k-tiles). A k-tile of 16 is ONE v_mfma_f32_32x32x16_bf16 per
// (partial product, 32x32 tile): LDS holds 16-byte cells [term][k half][row] = the 8 k-values one lane feeds, read back
// with ds_read_b128 (12 reads for 6 * TM * TN MFMAs per wave). The filters arrive pre-split (pack_filter_split_kernel) and
// are copied; the thread that gathers pixel column jb for k half kb0 holds exactly one cell's 8 values, splits them in
// registers (split3_cells) and writes three cells — no transposition anywhere.
// Pipeline: a k-tile's MFMAs last a third of the fp32 kernel's, so the gathers run TWO k-tiles ahead (two register
// sets, the loop is unrolled by two so that they stay static): step t issues the gathers of tile t+2 and the filter
// cells of tile t+1, runs the MFMAs of tile t out of LDS buffer t&1 and — in their shadow — splits tile t+1 into
// buffer (t+1)&1. One barrier per k-tile.
template <int BM, int BN, int WM, int WN, bool BNX, int NPROD>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BM == 128 ? 3 : 2, 8)))
igemm_split_kernel(IgemmArgs a) {
	constexpr int BK = 16, NT = 256;
	constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
	static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "4 waves per workgroup");
	static_assert(NPROD == 6 || NPROD == 9, "6 or 9 partial products");

	constexpr int kCells = 2 * 6 * (BM + BN);
	static_assert(kCells * 4 >= WM * WN * kEpiFloatsPerWave, "epilogue scratch does not fit the operand tiles");
	__shared__ u32x4 smem16[kCells];
	u32x4(*As16)[6][BM] = reinterpret_cast<u32x4(*)[6][BM]>(smem16);                 // [buf][term * 2 + k half][row]
	u32x4(*Bs16)[6][BN] = reinterpret_cast<u32x4(*)[6][BN]>(smem16 + 2 * 6 * BM);

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = wave / WN, wn = wave % WN;
	const int g = blockIdx.z;

	int L, kslice = -1;
	if ((int)blockIdx.x < a.full_tiles) {
		L = xcd_remap(blockIdx.x, a.full_tiles);
	} else {
		const int t = blockIdx.x - a.full_tiles;
		L = a.full_tiles + t / a.tail_splits;
		kslice = t % a.tail_splits;
	}
	const int tm = L % a.tiles_m, tn = L / a.tiles_m;

	// ---- B (gathered pixels): this thread owns pixel column jb and NB consecutive reduction channels of every k-tile
	constexpr int NB = BK * BN / NT, CB = NB / 8;           // 8 (one cell) or 16 (two cells)
	static_assert(NB % 8 == 0, "a thread gathers whole operand cells");
	const int jb = tid % BN;
	const int kb0 = __builtin_amdgcn_readfirstlane(tid / BN);

	const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.x_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t x2r = __builtin_amdgcn_make_buffer_rsrc((void *)(BNX ? a.x2 : a.x), 0, a.x_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *)a.wp, 0, a.wp_bytes, 0x00020000);

	const int pix = tn * BN + jb;
	unsigned long long tapmask = 0;
	unsigned base_bytes = 0;
	if (pix < a.npix) {
		const int pq_sz = a.Pv * a.Qv;
		const int n_img = pix / pq_sz;
		const int pq = pix - n_img * pq_sz;
		const int pp = pq / a.Qv, qq = pq - pp * a.Qv;
		const int h0 = pp * a.vs_h - a.pad_h, w0 = qq * a.vs_w - a.pad_w;
		for (int r = 0; r < a.R; ++r)
			for (int t = 0; t < a.S; ++t) {
				const bool ok = (unsigned)(h0 + r * a.dil_h) < (unsigned)a.H && (unsigned)(w0 + t * a.dil_w) < (unsigned)a.W;
				tapmask |= (unsigned long long)ok << (r * a.S + t);
			}
		base_bytes = (unsigned)((((long)n_img * a.C_total + (long)g * a.Cg) * a.H + h0) * a.W + w0) * 4u;
	}
	const unsigned mask_lo = (unsigned)tapmask, mask_hi = (unsigned)(tapmask >> 32);
	const unsigned hw4 = (unsigned)(a.H * a.W) * 4u;
	const unsigned row_off = (unsigned)(kb0 * NB) * hw4;

	// ---- A (pre-split filters): the 6 * BM cells of a k-tile in LDS order, cell f = (term * 2 + half) * BM + row
	constexpr int NA = (6 * BM + NT - 1) / NT;
	unsigned voffA[NA];
#pragma unroll
	for (int i = 0; i < NA; ++i) {
		const int f = tid + i * NT;
		voffA[i] = f < 6 * BM ? (unsigned)(((long)g * (a.kred_pad / BK) * 6 + f / BM) * a.mpad + tm * BM + f % BM) * 16u : kOOB;
	}

	f32x16 acc[TM][TN];
#pragma unroll
	for (int i = 0; i < TM; ++i)
#pragma unroll
		for (int j = 0; j < TN; ++j)
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

	u32x4 ra[NA];
	float rb[2][NB];
	float rb2[2][BNX ? NB : 1];
	const int l31 = lane & 31, lhi = lane >> 5;

	const int nk_all = a.kred_pad / BK;
	int kt0 = 0, kt1 = nk_all;
	if (kslice >= 0) {
		kt0 = (int)((long)nk_all * kslice / a.tail_splits);
		kt1 = (int)((long)nk_all * (kslice + 1) / a.tail_splits);
	}
	const int kt_last = kt1 - 1;

	auto issue_a = [&](int kt) {
#pragma unroll
		for (int i = 0; i < NA; ++i)
			ra[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, voffA[i], (unsigned)(kt * 6 * a.mpad) * 16u, 0));
	};
	auto issue_b = [&](int kt, float (&dst)[NB], float (&dst2)[BNX ? NB : 1]) {
		const int4 t = reinterpret_cast<const int4 *>(a.tab)[kt];
		const unsigned word = t.y < 32 ? mask_lo : mask_hi;
		const unsigned voff = (word >> (t.y & 31)) & 1u ? base_bytes + (unsigned)t.x : kOOB;
		const unsigned soff = (unsigned)t.z + row_off;
#pragma unroll
		for (int i = 0; i < NB; ++i) {
			dst[i] = buf_load_f32(xr, voff, soff + (unsigned)i * hw4);
			if constexpr (BNX) dst2[i] = buf_load_f32(x2r, voff, soff + (unsigned)i * hw4);
		}
	};
	auto store_a = [&](int buf) {
#pragma unroll
		for (int i = 0; i < NA; ++i) {
			const int f = tid + i * NT;
			if (6 * BM % NT == 0 || f < 6 * BM) (&As16[buf][0][0])[f] = ra[i];
		}
	};
	// tile kt's gathers (BNX: through the BatchNorm backward of their channels) -> three terms -> LDS buffer buf
	auto split_b = [&](int kt, const float (&src)[NB], const float (&src2)[BNX ? NB : 1], int buf) {
		const int ch0 = g * a.Cg + (kt * BK) % a.Cg + kb0 * NB;       // BNX: channels of this thread's cells (wave-uniform)
#pragma unroll
		for (int c = 0; c < CB; ++c) {
			float v[8];
#pragma unroll
			for (int e = 0; e < 8; ++e) {
				const int i = c * 8 + e;
				if constexpr (BNX) {
					const float4 co = a.xcoef[ch0 + i];
					v[e] = __builtin_fmaf(co.x, src[i], __builtin_fmaf(co.y, src2[i], co.z));
				} else {
					v[e] = src[i];
				}
			}
			u32x4 hi, mid, lo;
			split3_cells(v, hi, mid, lo);
			const int half = kb0 * CB + c;
			Bs16[buf][half][jb] = hi, Bs16[buf][2 + half][jb] = mid, Bs16[buf][4 + half][jb] = lo;
		}
	};
	auto mma_tile = [&](int buf) {
		bf16x8 fa[TM][3], fb[TN][3];
#pragma unroll
		for (int t = 0; t < 3; ++t) {
#pragma unroll
			for (int i = 0; i < TM; ++i) fa[i][t] = __builtin_bit_cast(bf16x8, As16[buf][2 * t + lhi][wm * (32 * TM) + i * 32 + l31]);
#pragma unroll
			for (int j = 0; j < TN; ++j) fb[j][t] = __builtin_bit_cast(bf16x8, Bs16[buf][2 * t + lhi][wn * (32 * TN) + j * 32 + l31]);
		}
		// term orders 0 (hi*hi), 1, 2 (, 3, 4): the large terms' fragments are the first to arrive
#pragma unroll
		for (int order = 0; order <= (NPROD == 9 ? 4 : 2); ++order)
#pragma unroll
			for (int ta = 0; ta < 3; ++ta) {
				const int tb = order - ta;
				if (tb < 0 || tb > 2) continue;
#pragma unroll
				for (int i = 0; i < TM; ++i)
#pragma unroll
					for (int jj = 0; jj < TN; ++jj)
						acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][ta], fb[jj][tb], acc[i][jj], 0, 0, 0);
			}
	};
	// one k-tile; SET = (kt - kt0) & 1 names the LDS buffer it computes from and the register set it refills
	// (BNX gathers two tensors: with two sets in flight the kernel would not fit three waves per SIMD — it keeps one,
	// filled at the start of a step and split at its end)
	constexpr int AHEAD = BNX ? 1 : 2;
	auto step = [&](auto set_tag, int kt) {
		constexpr int SET = decltype(set_tag)::value;
		constexpr int FILL = AHEAD == 2 ? SET : 0, DRAIN = AHEAD == 2 ? SET ^ 1 : 0;
		const int kt_a = min(kt + 1, kt_last), kt_b = min(kt + AHEAD, kt_last);
		issue_a(kt_a);
		issue_b(kt_b, rb[FILL], rb2[FILL]);
		mma_tile(SET);
		split_b(kt_a, rb[DRAIN], rb2[DRAIN], SET ^ 1);
		store_a(SET ^ 1);
		__syncthreads();
	};

	issue_a(kt0);
	issue_b(kt0, rb[0], rb2[0]);
	if constexpr (AHEAD == 2) issue_b(min(kt0 + 1, kt_last), rb[1], rb2[1]);
	split_b(kt0, rb[0], rb2[0], 0);
	store_a(0);
	__syncthreads();

	int kt = kt0;
	for (; kt + 1 < kt1; kt += 2) {
		step(std::integral_constant<int, 0>{}, kt);
		step(std::integral_constant<int, 1>{}, kt + 1);
	}
	if (kt < kt1) step(std::integral_constant<int, 0>{}, kt);

	float *scratch = reinterpret_cast<float *>(smem16);
	if (kslice < 0) {
		if (a.contig)         // (the last step's barrier: every wave is done reading the operand tiles)
			igemm_store_tile_lds<BM, BN, WM, WN, TM, TN>(a, tm, tn, g, wm, wn, wave, lane, acc, scratch);
		else
			igemm_store_tile<BM, BN, WM, WN, TM, TN>(a, tm, tn, g, wm, wn, lane, acc);
	} else {
		float *slab = a.slabs + ((size_t)(blockIdx.x - a.full_tiles) + (size_t)g * (gridDim.x - a.full_tiles)) * (BM * BN);
#pragma unroll
		for (int i = 0; i < TM; ++i)
#pragma unroll
			for (int j = 0; j < TN; ++j)
#pragma unroll
				for (int r = 0; r < 16; ++r) slab[((i * TN + j) * 16 + r) * NT + tid] = acc[i][j][r];
	}
}

// sums the k-slices of one tail tile (fixed order) and writes it out like a whole tile. One single-wave workgroup per
// (tail tile, wave of the producing workgroup, 32-row sub-tile row): the slab rows of a wave are 256 contiguous bytes per
// accumulator register. The slices are summed in order, but their loads are issued kGroup slices at a time (one workgroup
// per tile walking the slices one by one was a chain of `splits` dependent memory latencies: 28 us per launch where the
// data is 16 MB).
template <int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(64) igemm_tail_reduce_kernel(IgemmArgs a) {
	constexpr int NT = 64 * WM * WN, TM = BM / WM / 32, TN = BN / WN / 32;
	constexpr int kGroup = 256 / (TN * 16);             // slices in flight (<= 256 registers)
	const int lane = threadIdx.x;
	const int isel = blockIdx.x % TM, wave = (blockIdx.x / TM) % (WM * WN), tile = blockIdx.x / (TM * WM * WN);
	const int wm = wave / WN, wn = wave % WN;
	const int g = blockIdx.z;
	const int tid = wave * 64 + lane;          // the thread of the producing workgroup whose accumulators this lane sums

	const int ntail = a.tiles_m * a.tiles_n - a.full_tiles;
	const int L = a.full_tiles + tile;
	const int tm = L % a.tiles_m, tn = L / a.tiles_m;
	const float *slab0 = a.slabs + ((size_t)tile * a.tail_splits + (size_t)g * ntail * a.tail_splits) * (BM * BN);

	f32x16 acc[TM][TN];
#pragma unroll
	for (int i = 0; i < TM; ++i) {
		if (i != isel) continue;
#pragma unroll
		for (int j = 0; j < TN; ++j)
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

		for (int sl0 = 0; sl0 < a.tail_splits; sl0 += kGroup) {
			float part[kGroup][TN][16];
#pragma unroll
			for (int q = 0; q < kGroup; ++q) {
				const int sl = min(sl0 + q, a.tail_splits - 1);              // clamped: the surplus loads are not added
				const float *slab = slab0 + (size_t)sl * (BM * BN);
#pragma unroll
				for (int j = 0; j < TN; ++j)
#pragma unroll
					for (int r = 0; r < 16; ++r) part[q][j][r] = slab[((i * TN + j) * 16 + r) * NT + tid];
			}
#pragma unroll
			for (int q = 0; q < kGroup; ++q)
				if (sl0 + q < a.tail_splits) {
#pragma unroll
					for (int j = 0; j < TN; ++j)
#pragma unroll
						for (int r = 0; r < 16; ++r) acc[i][j][r] += part[q][j][r];
				}
		}
	}

	if (a.contig) {
		__shared__ __attribute__((aligned(16))) float smem[kEpiFloatsPerWave];
		igemm_store_tile_lds<BM, BN, WM, WN, TM, TN>(a, tm, tn, g, wm, wn, 0, lane, acc, smem, isel);
	} else {
		igemm_store_tile<BM, BN, WM, WN, TM, TN>(a, tm, tn, g, wm, wn, lane, acc, isel);
	}
}

// ------------------------------------------------------------------------------------------------
// backward-filter kernel: acc[m = out channel][n = (c,r,s)] over kred = pixels, split along kred
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
	const float *x;       // (N, C_total, H, W)
	const float *dy;      // (N, K_total, P, Q)
	const int2 *tab;      // [ncrs_pad]
	float *out;           // dw (splits == 1) or partial slabs [split][groups*Kg*ncrs]
	int C_total, H, W, Cg;
	int K_total, P, Q, Kg;
	int ncrs;
	int st_h, st_w, pad_h, pad_w;
	int R, S, dil_h, dil_w;
	unsigned x_bytes, dy_bytes;
	int npix, steps_total, steps_per_split;     // npix counts RUNS of 4 pixels: N * P * Q4
	unsigned Q4, magic_q4, magic_p;             // Q4 = ceil(Q/4); magic = ceil(2^32 / d) for exact u / d by __umulhi
	int tiles_m, tiles_n;
	float alpha, beta;
	int direct;           // 1: out = beta*out + alpha*acc ; 0: out[split] = acc
	size_t slab;          // groups*Kg*ncrs
	// BNX kernels: dy is not materialised — it is coef[k].x * dy + coef[k].y * bnx + coef[k].z per output channel k, i.e.
	// the backward of the BatchNorm that follows this convolution applied while gathering (dy = the BN's incoming
	// gradient, bnx = this convolution's output / the BN's input, same shape)
	const float *bnx;
	const float4 *bncoef;
	// BIAS kernels: the bias gradient db[k] = sum over (image, pixel) of dy rides on the operand-A runs the workgroups of
	// tile column 0 park anyway (no pass of its own over dy): db (direct) or per-split partials [split][K_total]
	float *db_out;
	// XBN kernels (pointwise gathers): operand B is relu?(xbn[c].x * x + xbn[c].y) per input channel c — the normalised, activated
	// input of this convolution, never written (see IgemmArgs)
	const float2 *xbn;
	int xbn_relu;
};

// The reduction axis is enumerated in RUNS of 4 consecutive output pixels of one output row (rows padded to a multiple
// of 4): a run of dy is 16 contiguous bytes, and for stride-1 convolutions so is the matching run of x for any tap, so
// both operands arrive as 16-byte loads (4-byte aligned is enough on gfx950). A k-step is 8 runs = 32 (padded) pixels.
// LDS keeps a run as two 8-byte half-cells [run][half][row]{2 pixels}: the MFMA fragment of lane (row, half h) is
// {pixel 2h, 2h+1} of its row, i.e. the k order inside a run is (0,2 | 1,3) for both operands.
// GATHER: 0 = any stride (element-wise x gathers), 1 = unit stride along w (16-byte x runs with edge masks),
//         2 = pointwise 1x1 / stride 1 / pad 0 (16-byte runs, only the row tail is masked: no per-tap address or
//             mask arithmetic — VALU instructions serialise with the MFMAs, see tools/probes/lds_mfma.hip)
// WM x WN = 4 or 8 waves. With 8 (PZ_WG_WAVES=8, tiles of at least 8 MFMA tiles) a wave owns half as many accumulators
// and gathers half as many runs, so two workgroups per CU (the LDS limit) put 4 waves on every SIMD instead of 2 — what
// gave the Winograd kernels 6-11 % changes nothing here (every census layer within +-2 %), so 4 stays the default.
template <int BM, int BN, int WM, int WN, int GATHER, int RUNS, bool BNX = false, bool BIAS = false, bool XBN = false>
__global__ void __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(RUNS == 8 || BM > 128 ? WM * WN / 2 : 4, 8)))
wgrad_conv_kernel(WgradArgs a) {
	static_assert(!(BNX && BIAS), "a convolution in front of a BatchNorm has no bias gradient of its own to fold");
	static_assert(!XBN || (GATHER == 2 && !BIAS), "the BatchNorm-forward gather: pointwise layers without a bias");
	constexpr bool UNIT_W = GATHER >= 1, POINTWISE = GATHER == 2;
	constexpr int NT = 64 * WM * WN;
	constexpr int RP = NT / RUNS;                // tile rows loaded per pass (one 16-byte run per thread)
	static_assert(RUNS % 2 == 0, "runs are consumed in pairs (one per lane half)");
	constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
	static_assert((WM * WN == 4 || WM * WN == 8) && TM >= 1 && TN >= 1, "4 or 8 waves per workgroup");

	// double buffered (one barrier per k-step). A run is one 16-byte cell [run pair][run of the pair][row]: parked with one
	// ds_write_b128, and the fragment of lane (row, half h) is the whole cell of run 2g+h = four k2-steps per ds_read_b128.
	// The odd row stride (BM + 1 cells) spreads the 8 runs a group of 8 lanes writes over all banks.
	__shared__ __attribute__((aligned(16))) f32x4 As[2][RUNS / 2][2][BM + 1];
	__shared__ __attribute__((aligned(16))) f32x4 Bs[2][RUNS / 2][2][BN + 1];
	__shared__ int2 tabs[BN];

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = wave / WN, wn = wave % WN;
	const int g = blockIdx.z;

	// XCD-aware order: one XCD walks consecutive (split, tile) pairs, so the tiles that re-read one split's pixel range
	// of dy and x find it in that XCD's L2
	const int ntiles = a.tiles_m * a.tiles_n;
	const int B = xcd_remap(blockIdx.x, gridDim.x);
	const int split = B / ntiles, L = B - split * ntiles;
	const int tm = L % a.tiles_m, tn = L / a.tiles_m;

	if (tid < BN) tabs[tid] = a.tab[tn * BN + tid];

	const int run = tid % RUNS, row0 = tid / RUNS;    // this thread's run of the k-step, first tile row it loads (32 rows per pass)
	constexpr int NA = BM / RP, NB = BN / RP;
	static_assert(NA >= 1 && NB >= 1, "tile narrower than one load pass");

	f32x16 acc[TM][TN];
#pragma unroll
	for (int i = 0; i < TM; ++i)
#pragma unroll
		for (int j = 0; j < TN; ++j)
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

	// two register sets: the gathers of k-step s+2 are issued while step s computes and are parked in LDS at the end of
	// step s+1, a full step after their issue — with 2 workgroups per CU there is too little other work to hide an
	// exposed HBM latency behind
	constexpr int SETS = PZ_WG_SETS;              // register sets of gathers in flight (2: issued two k-steps ahead)
	f32x4 ra[SETS][NA], rb[SETS][NB];
	f32x4 ra2[SETS][BNX ? NA : 1];           // BNX: the BatchNorm input runs that go with the dy runs
	unsigned mb[SETS][NB];         // valid-pixel masks of the operand-B runs in flight
	int x_img_of[SETS] = {};       // x_img of the step held by each set (the rare far-left fix-up in store_step needs it)
	int nq_of[SETS] = {};          // BIAS: real pixels of the dy run held by each set
	float bsum[BIAS ? NA : 1] = {};        // BIAS: this thread's share of the bias gradient of its NA rows
	const bool bias_here = BIAS && tn == 0;
	const int PQ = a.P * a.Q;

	__syncthreads();   // tabs visible

	// this thread gathers the same NB filter taps (rows of operand B) in every k-step: keep them decoded in registers,
	// so the loop has no LDS read whose wait would also drain the fragment reads queued ahead of it
	int tap_h[NB], tap_w[NB], tap_off[NB];
#pragma unroll
	for (int i = 0; i < NB; ++i) {
		const int2 e = tabs[row0 + RP * i];
		tap_h[i] = (e.y >> 8) * a.dil_h, tap_w[i] = (e.y & 0xff) * a.dil_w, tap_off[i] = e.x >> 2;
	}

	const int l31 = lane & 31, lhi = lane >> 5;

	const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.x_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t dyr = __builtin_amdgcn_make_buffer_rsrc((void *)a.dy, 0, a.dy_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t bnr = __builtin_amdgcn_make_buffer_rsrc((void *)(BNX ? a.bnx : a.dy), 0, a.dy_bytes, 0x00020000);

	float4 bnc[BNX ? NA : 1];                // this thread's operand-A rows are the same output channels in every k-step
	if constexpr (BNX) {
#pragma unroll
		for (int i = 0; i < NA; ++i) bnc[i] = a.bncoef[g * a.Kg + min(tm * BM + row0 + RP * i, a.Kg - 1)];
	}
	const bool full_m = tm * BM + BM <= a.Kg;
	float2 xc[XBN ? NB : 1];                 // ... and its operand-B rows the same input channels (pointwise: column = channel)
	if constexpr (XBN) {
#pragma unroll
		for (int i = 0; i < NB; ++i) xc[i] = a.xbn[g * a.Cg + min(tn * BN + row0 + RP * i, a.Cg - 1)];
	}

	// ---- per-step state of this thread's run: image / row / first column, number of real pixels in the run
	unsigned dy_off = kOOB;        // byte offset of dy[n, g*Kg + tm*BM + row0, p, q0]
	int x_img = 0;                 // element offset of x[n, g*Cg, hb, wb] (may point left of / above the image)
	int hb = 0, wb = 0, nq = 0;    // input row/col of tap (0,0) for pixel q0; real pixels in the run (0..4)

	auto load_head = [&](int step) {
		const unsigned u = (unsigned)(step * RUNS + run);
		nq = 0, dy_off = kOOB;
		if (u < (unsigned)a.npix) {            // a.npix = N * P * Q4 runs
			// multiply-high division, exact while u * divisor < 2^32 (igemm_eligible); divisor 1 has no 32-bit magic
			const unsigned np = a.Q4 == 1u ? u : __umulhi(u, a.magic_q4), qb = u - np * a.Q4;
			const unsigned n_img = a.P == 1 ? np : __umulhi(np, a.magic_p), p = np - n_img * a.P;
			const int q0 = (int)qb * 4;
			nq = min(4, a.Q - q0);
			hb = (int)p * a.st_h - a.pad_h, wb = q0 * a.st_w - a.pad_w;
			dy_off = ((((unsigned)n_img * a.K_total + (unsigned)(g * a.Kg + tm * BM + row0)) * a.P + p) * a.Q + q0) * 4u;
			x_img = (int)(((unsigned)n_img * a.C_total + (unsigned)(g * a.Cg)) * (unsigned)(a.H * a.W)) + hb * a.W + wb;
		}
	};

	// one part = one operand-A row pass or one operand-B row pass (each a 16-byte load per thread). Pixels beyond the
	// row end (nq < 4) are zeroed in operand B only: operand A then holds finite data of the next row, times zero.
	auto load_part = [&](int set, int j) {
		if (j < NA) {
			const int i = j;
			const bool ok = dy_off != kOOB && (full_m || tm * BM + row0 + RP * i < a.Kg);
			ra[set][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
			    dyr, ok ? dy_off : kOOB, (unsigned)(RP * i) * (unsigned)PQ * 4u, 0));
			if constexpr (BNX)
				ra2[set][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
				    bnr, ok ? dy_off : kOOB, (unsigned)(RP * i) * (unsigned)PQ * 4u, 0));
		} else if (j < NA + NB) {
			const int i = j - NA;
			const int w0 = wb + tap_w[i];
			const bool row_ok = nq > 0 && (unsigned)(hb + tap_h[i]) < (unsigned)a.H;
			const int first = x_img + tap_off[i];        // element offset of the run's first input column inside the tensor

			if constexpr (POINTWISE) {
				const unsigned m = (1u << nq) - 1u;          // nq = 0 for runs beyond the tensor
				rb[set][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
				    xr, m != 0u ? (unsigned)first * 4u : kOOB, 0, 0));
				mb[set][i] = m;
			} else if constexpr (UNIT_W) {
				// valid pixels of the run: q in [lo, hi)
				const int lo = max(0, -w0), cnt = row_ok ? min(nq, a.W - w0) - lo : 0;
				const unsigned m = cnt > 0 ? ((1u << cnt) - 1u) << lo : 0u;          // v_bfm_b32

				// A run that starts left of the tensor's first byte (first row of the first image, left padding) would wrap
				// the 32-bit offset: such a lane loads nothing here and is flagged (bit 4) for store_step to gather it.
				const bool far_left = first < 0 && m != 0u;
				rb[set][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
				    xr, (m != 0u && first >= 0) ? (unsigned)first * 4u : kOOB, 0, 0));
				mb[set][i] = far_left ? (m | 16u) : m;      // the mask is applied when the run is parked in LDS: no wait on the load here
			} else {
				f32x4 v;
#pragma unroll
				for (int q = 0; q < 4; ++q) {
					const bool ok = row_ok && q < nq && (unsigned)(w0 + q * a.st_w) < (unsigned)a.W;
					v[q] = buf_load_f32(xr, ok ? (unsigned)(first + q * a.st_w) * 4u : kOOB, 0);
				}
				rb[set][i] = v, mb[set][i] = 0xfu;
			}
		}
		if (j == NA + NB - 1) x_img_of[set] = x_img;
		if (BIAS && j == 0) nq_of[set] = nq;
	};

	auto store_step = [&](int set, int buf) {
#pragma unroll
		for (int i = 0; i < NA; ++i) {
			f32x4 v = ra[set][i];
			if constexpr (BNX) {
#pragma unroll
				for (int q = 0; q < 4; ++q)
					v[q] = __builtin_fmaf(bnc[i].x, (float)ra[set][i][q], __builtin_fmaf(bnc[i].y, (float)ra2[set][i][q], bnc[i].z));
			}
			As[buf][run >> 1][run & 1][row0 + RP * i] = v;
			if constexpr (BIAS) {
				if (bias_here) {              // pixels past the row end hold the next row's data (they meet zeros in operand B)
					const int m = (1 << nq_of[set]) - 1;
					float t[4];
#pragma unroll
					for (int q = 0; q < 4; ++q) t[q] = __builtin_bit_cast(float, __builtin_bit_cast(int, (float)v[q]) & __builtin_amdgcn_sbfe(m, q, 1));
					bsum[i] += (t[0] + t[1]) + (t[2] + t[3]);
				}
			}
		}
#pragma unroll
		for (int i = 0; i < NB; ++i) {
			const unsigned m = mb[set][i];
			if (GATHER == 1 && (m & 16u)) {  // rare: run starting left of the tensor base, gathered element-wise
				const int first = x_img_of[set] + tap_off[i];
#pragma unroll
				for (int q = 0; q < 4; ++q) rb[set][i][q] = buf_load_f32(xr, (m >> q) & 1u ? (unsigned)(first + q) * 4u : kOOB, 0);
			}
			// bit q of m -> all-ones / zero word (v_bfe_i32), then one AND per element: 2 VALU instead of test + compare + select
			f32x4 v;
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				float t = (float)rb[set][i][q];
				if constexpr (XBN) {              // bn_apply_add_kernel's expression, THEN the mask (relu(b) of a pixel beyond the row is not 0)
					t = __builtin_fmaf(t, xc[i].x, xc[i].y);
					if (a.xbn_relu) t = t > 0.f ? t : 0.f;
				}
				v[q] = __builtin_bit_cast(float, __builtin_bit_cast(int, t) & __builtin_amdgcn_sbfe((int)m, q, 1));
			}
			Bs[buf][run >> 1][run & 1][row0 + RP * i] = v;
		}
	};

	// fragments of one run pair (4 k2-steps): lane (row, h) takes the 4 pixels of run 2g+h of its row
	auto read_frag = [&](int buf, int g2, f32x4 (&av)[TM], f32x4 (&bv)[TN]) {
#pragma unroll
		for (int i = 0; i < TM; ++i)
			av[i] = As[buf][g2][lhi][wm * (32 * TM) + i * 32 + l31];
#pragma unroll
		for (int j = 0; j < TN; ++j)
			bv[j] = Bs[buf][g2][lhi][wn * (32 * TN) + j * 32 + l31];
	};

	// MFMAs of the k-step parked in LDS buffer `buf`; when has_next, the gathers of step `next` go into register set `set`
	auto compute_step = [&](int buf, int set, int next, bool has_next) {
		constexpr int G2 = RUNS / 2;
		f32x4 av[2][TM], bv[2][TN];
		read_frag(buf, 0, av[0], bv[0]);
		if (has_next) load_head(next);

#pragma unroll
		for (int g2 = 0; g2 < G2; ++g2) {
			if (g2 + 1 < G2) read_frag(buf, g2 + 1, av[(g2 + 1) & 1], bv[(g2 + 1) & 1]);
#pragma unroll
			for (int sub = 0; sub < 4; ++sub) {
				{                                         // the next step's gathers, spread evenly over this step's k2-steps
					constexpr int SLOTS = G2 * 4, PARTS = NA + NB;
					constexpr int STRIDE = SLOTS >= PARTS ? SLOTS / PARTS : 1;
					constexpr int PER = SLOTS >= PARTS ? 1 : (PARTS + SLOTS - 1) / SLOTS;
					const int slot = g2 * 4 + sub;
					if (has_next && slot % STRIDE == 0) {
#pragma unroll
						for (int j = 0; j < PER; ++j)
							if ((slot / STRIDE) * PER + j < PARTS) load_part(set, (slot / STRIDE) * PER + j);
					}
				}
				__builtin_amdgcn_sched_barrier(0);
#pragma unroll
				for (int i = 0; i < TM; ++i)
#pragma unroll
					for (int jj = 0; jj < TN; ++jj)
						acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g2 & 1][i][sub], bv[g2 & 1][jj][sub], acc[i][jj], 0, 0, 0);
				__builtin_amdgcn_sched_barrier(0);
			}
		}
	};

	const int s_begin = split * a.steps_per_split;
	const int s_end = min(s_begin + a.steps_per_split, a.steps_total);

	if (s_begin < s_end && SETS == 2) {
		// prologue: step s_begin -> set 0 -> LDS buffer 0; step s_begin+1 -> set 1 (left in flight)
		constexpr int S1 = SETS - 1;
		load_head(s_begin);
#pragma unroll
		for (int j = 0; j < NA + NB; ++j) load_part(0, j);
		store_step(0, 0);
		if (s_begin + 1 < s_end) {
			load_head(s_begin + 1);
#pragma unroll
			for (int j = 0; j < NA + NB; ++j) load_part(S1, j);
		}
		__syncthreads();

		// invariant at an even position: LDS buffer 0 holds `step`, set 1 holds step+1 in flight, set 0 is free
		int step = s_begin;
		for (; step + 3 < s_end; step += 2) {
			compute_step(0, 0, step + 2, true);
			store_step(S1, 1);
			__syncthreads();
			compute_step(1, S1, step + 3, true);
			store_step(0, 0);
			__syncthreads();
		}

		const int left = s_end - step;           // 1, 2 or 3 steps remain
		if (left == 3) {
			compute_step(0, 0, step + 2, true);
			store_step(S1, 1);
			__syncthreads();
			compute_step(1, S1, 0, false);
			store_step(0, 0);
			__syncthreads();
			compute_step(0, 0, 0, false);
		} else if (left == 2) {
			compute_step(0, 0, 0, false);
			store_step(S1, 1);
			__syncthreads();
			compute_step(1, S1, 0, false);
		} else {
			compute_step(0, 0, 0, false);
		}
	} else if (s_begin < s_end) {
		// one register set: the gathers of step s+1 are issued during step s and parked at its end
		load_head(s_begin);
#pragma unroll
		for (int j = 0; j < NA + NB; ++j) load_part(0, j);
		store_step(0, 0);
		__syncthreads();

		for (int step = s_begin; step + 1 < s_end; ++step) {
			const int buf = (step - s_begin) & 1;
			compute_step(buf, 0, step + 1, true);
			store_step(0, buf ^ 1);
			__syncthreads();
		}
		compute_step((s_end - 1 - s_begin) & 1, 0, 0, false);
	}

	if constexpr (BIAS) {
		if (bias_here) {
			// the RUNS threads of a row are consecutive lanes: butterfly over them (fixed order), lane `run == 0` stores
			static_assert(RUNS == 8 || RUNS == 4, "bias gradient: a row's lanes are a power of two");
#pragma unroll
			for (int i = 0; i < NA; ++i) {
				float v = bsum[i];
#pragma unroll
				for (int o = 1; o < RUNS; o <<= 1) v += __shfl_xor(v, o);
				const int m = tm * BM + row0 + RP * i;
				if (run == 0 && m < a.Kg) {
					if (a.direct) {
						float *o = a.db_out + g * a.Kg + m;
						*o = (a.beta == 0.f ? 0.f : a.beta * *o) + a.alpha * v;
					} else {
						a.db_out[(size_t)split * a.K_total + g * a.Kg + m] = v;
					}
				}
			}
		}
	}

	float *outb = a.out + (a.direct ? 0 : (size_t)split * a.slab) + (size_t)g * a.Kg * a.ncrs;
#pragma unroll
	for (int j = 0; j < TN; ++j) {
		const int col = tn * BN + wn * (32 * TN) + j * 32 + l31;
		if (col >= a.ncrs) continue;
#pragma unroll
		for (int i = 0; i < TM; ++i) {
#pragma unroll
			for (int r = 0; r < 16; ++r) {
				const int m = tm * BM + wm * (32 * TM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
				if (m < a.Kg) {
					float *o = outb + (size_t)m * a.ncrs + col;
					const float v = acc[i][j][r];
					if (a.direct)
						*o = (a.beta == 0.f ? 0.f : a.beta * *o) + a.alpha * v;
					else
						*o = v;
				}
			}
		}
	}
}

// ------------------------------------------------------------------------------------------------
// backward-filter of pointwise (1x1, stride 1, pad 0) convolutions on the bf16 matrix pipe (split operands)
// ------------------------------------------------------------------------------------------------
// dw[k][c] = sum over (image, pixel) dy[n][k][pix] * x[n][c][pix]: both operands are rows of contiguous pixels, and the
// reduction index is the pixel — exactly the k axis of a v_mfma_f32_32x32x16_bf16 fragment (8 consecutive k per lane). An
// image plane is walked in CELLS of 8 consecutive pixels (the last cell of a plane is masked in operand B); a thread loads
// one cell (2 x 16 bytes) of one row per operand, splits it into three bf16 cells (split3_cells) and parks them in LDS as
// [term][cell of the k-step][row]; a k-step is 2 cells = 16 pixels = one MFMA per (partial product, 32x32 tile).
// Split along the reduction like wgrad_conv_kernel (slabs + wgrad_reduce_kernel); a.steps_* count k-steps of 2 cells,
// a.npix = cells in the tensor, a.Q4 = cells per plane (magic_q4 its multiply-high reciprocal).
#ifndef PZ_WGS_BNX_AHEAD
#define PZ_WGS_BNX_AHEAD 2
#endif
template <int BM, int BN, int WM, int WN, bool BNX, int NPROD>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BNX && PZ_WGS_BNX_AHEAD == 2 ? 2 : 3, 8)))
wgrad_split_kernel(WgradArgs a) {
	constexpr int NT = 256;
	constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
	static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "4 waves per workgroup");
	constexpr int NA = (BM + 127) / 128, NB = (BN + 127) / 128;        // cells per thread and operand

	// [buf][term][cell][row]; the row pitch of BM + 2 cells spreads the (2 rows x 2 cells) x 2 a group of 8 lanes writes
	__shared__ u32x4 As16[2][3][2][BM + 2];
	__shared__ u32x4 Bs16[2][3][2][BN + 2];

	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = wave / WN, wn = wave % WN;
	const int g = blockIdx.z;

	const int ntiles = a.tiles_m * a.tiles_n;
	const int B = xcd_remap(blockIdx.x, gridDim.x);
	const int split = B / ntiles, L = B - split * ntiles;
	const int tm = L % a.tiles_m, tn = L / a.tiles_m;

	const int cell = tid & 1, row0 = tid >> 1;
	const int PQ = a.P * a.Q;

	const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.x_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t dyr = __builtin_amdgcn_make_buffer_rsrc((void *)a.dy, 0, a.dy_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t bnr = __builtin_amdgcn_make_buffer_rsrc((void *)(BNX ? a.bnx : a.dy), 0, a.dy_bytes, 0x00020000);

	// rows of this thread (fixed for the whole kernel): byte offsets of (row, pixel 0) inside image 0, or out of range
	unsigned rowA[NA], rowB[NB];
	float4 bnc[BNX ? NA : 1];
#pragma unroll
	for (int i = 0; i < NA; ++i) {
		const int r = row0 + 128 * i, m = tm * BM + r;
		rowA[i] = (r < BM && m < a.Kg) ? (unsigned)(g * a.Kg + m) * (unsigned)PQ * 4u : kOOB;
		if constexpr (BNX) bnc[i] = a.bncoef[g * a.Kg + min(m, a.Kg - 1)];
	}
#pragma unroll
	for (int i = 0; i < NB; ++i) {
		const int r = row0 + 128 * i, c = tn * BN + r;
		rowB[i] = (r < BN && c < a.Cg) ? (unsigned)(g * a.Cg + c) * (unsigned)PQ * 4u : kOOB;
	}

	f32x16 acc[TM][TN];
#pragma unroll
	for (int i = 0; i < TM; ++i)
#pragma unroll
		for (int j = 0; j < TN; ++j)
#pragma unroll
			for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

	constexpr int AHEAD = BNX ? PZ_WGS_BNX_AHEAD : 2;
	f32x4 ra[AHEAD][NA][2], rb[AHEAD][NB][2], ra2[AHEAD][BNX ? NA : 1][2];
	int nvalid[AHEAD];            // real pixels in this thread's cell of the step a set holds (0..8)
	const int l31 = lane & 31, lhi = lane >> 5;

	// global loads of k-step `step` into register set `set`
	auto issue = [&](int step, int set) {
		const unsigned u = (unsigned)(step * 2 + cell);
		unsigned img_dy = kOOB, img_x = kOOB;
		nvalid[set] = 0;
		if (u < (unsigned)a.npix) {
			const unsigned n_img = a.Q4 == 1u ? u : __umulhi(u, a.magic_q4), c = u - n_img * a.Q4;
			const unsigned pix = c * 8u;
			nvalid[set] = min(8, PQ - (int)pix);
			img_dy = (n_img * (unsigned)a.K_total * (unsigned)PQ + pix) * 4u;
			img_x = (n_img * (unsigned)a.C_total * (unsigned)PQ + pix) * 4u;
		}
#pragma unroll
		for (int i = 0; i < NA; ++i) {
			const unsigned off = (img_dy != kOOB && rowA[i] != kOOB) ? img_dy + rowA[i] : kOOB;
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				ra[set][i][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dyr, off, 16 * h, 0));
				if constexpr (BNX) ra2[set][i][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bnr, off, 16 * h, 0));
			}
		}
#pragma unroll
		for (int i = 0; i < NB; ++i) {
			const unsigned off = (img_x != kOOB && rowB[i] != kOOB) ? img_x + rowB[i] : kOOB;
#pragma unroll
			for (int h = 0; h < 2; ++h) rb[set][i][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 16 * h, 0));
		}
	};

	// register set -> three bf16 terms -> LDS buffer
	auto park = [&](int set, int buf) {
#pragma unroll
		for (int i = 0; i < NA; ++i) {
			float v[8];
#pragma unroll
			for (int e = 0; e < 8; ++e) {
				v[e] = ra[set][i][e >> 2][e & 3];
				if constexpr (BNX) v[e] = __builtin_fmaf(bnc[i].x, v[e], __builtin_fmaf(bnc[i].y, (float)ra2[set][i][e >> 2][e & 3], bnc[i].z));
			}
			u32x4 hi, mid, lo;
			split3_cells(v, hi, mid, lo);
			const int r = row0 + 128 * i;
			if (BM % 128 == 0 || r < BM) As16[buf][0][cell][r] = hi, As16[buf][1][cell][r] = mid, As16[buf][2][cell][r] = lo;
		}
		// the last cell of an image plane is shorter than 8 pixels: its tail reads the next plane (or nothing) and is
		// zeroed in operand B. Rare (one k-step in PQ/16), so the masking sits behind a wave-uniform test.
		const bool ragged = __builtin_amdgcn_ballot_w64(nvalid[set] < 8) != 0ull;
#pragma unroll
		for (int i = 0; i < NB; ++i) {
			float v[8];
#pragma unroll
			for (int e = 0; e < 8; ++e) v[e] = rb[set][i][e >> 2][e & 3];
			if (ragged) {
#pragma unroll
				for (int e = 0; e < 8; ++e) v[e] = e < nvalid[set] ? v[e] : 0.f;
			}
			u32x4 hi, mid, lo;
			split3_cells(v, hi, mid, lo);
			const int r = row0 + 128 * i;
			if (BN % 128 == 0 || r < BN) Bs16[buf][0][cell][r] = hi, Bs16[buf][1][cell][r] = mid, Bs16[buf][2][cell][r] = lo;
		}
	};

	auto mma_step = [&](int buf) {
		bf16x8 fa[TM][3], fb[TN][3];
#pragma unroll
		for (int t = 0; t < 3; ++t) {
#pragma unroll
			for (int i = 0; i < TM; ++i) fa[i][t] = __builtin_bit_cast(bf16x8, As16[buf][t][lhi][wm * (32 * TM) + i * 32 + l31]);
#pragma unroll
			for (int j = 0; j < TN; ++j) fb[j][t] = __builtin_bit_cast(bf16x8, Bs16[buf][t][lhi][wn * (32 * TN) + j * 32 + l31]);
		}
#pragma unroll
		for (int order = 0; order <= (NPROD == 9 ? 4 : 2); ++order)
#pragma unroll
			for (int ta = 0; ta < 3; ++ta) {
				const int tb = order - ta;
				if (tb < 0 || tb > 2) continue;
#pragma unroll
				for (int i = 0; i < TM; ++i)
#pragma unroll
					for (int jj = 0; jj < TN; ++jj)
						acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][ta], fb[jj][tb], acc[i][jj], 0, 0, 0);
			}
	};

	const int s_begin = split * a.steps_per_split;
	const int s_end = min(s_begin + a.steps_per_split, a.steps_total);

	// steps past s_end address cells beyond this split's range: they are loaded (harmless, the next split's data or
	// nothing) but never multiplied
	auto step_fn = [&](auto par_tag, int step) {
		constexpr int PAR = decltype(par_tag)::value;
		constexpr int FILL = AHEAD == 2 ? PAR : 0, DRAIN = AHEAD == 2 ? PAR ^ 1 : 0;
		issue(step + AHEAD, FILL);
		mma_step(PAR);
		park(DRAIN, PAR ^ 1);
		__syncthreads();
	};

	if (s_begin < s_end) {
		issue(s_begin, 0);
		if constexpr (AHEAD == 2) issue(s_begin + 1, 1);
		park(0, 0);
		__syncthreads();
		int step = s_begin;
		for (; step + 1 < s_end; step += 2) {
			step_fn(std::integral_constant<int, 0>{}, step);
			step_fn(std::integral_constant<int, 1>{}, step + 1);
		}
		if (step < s_end) step_fn(std::integral_constant<int, 0>{}, step);
	}

	float *outb = a.out + (a.direct ? 0 : (size_t)split * a.slab) + (size_t)g * a.Kg * a.ncrs;
#pragma unroll
	for (int j = 0; j < TN; ++j) {
		const int col = tn * BN + wn * (32 * TN) + j * 32 + l31;
		if (col >= a.ncrs) continue;
#pragma unroll
		for (int i = 0; i < TM; ++i) {
#pragma unroll
			for (int r = 0; r < 16; ++r) {
				const int m = tm * BM + wm * (32 * TM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
				if (m < a.Kg) {
					float *o = outb + (size_t)m * a.ncrs + col;
					const float v = acc[i][j][r];
					if (a.direct)
						*o = (a.beta == 0.f ? 0.f : a.beta * *o) + a.alpha * v;
					else
						*o = v;
				}
			}
		}
	}
}

// dw = beta*dw + alpha * (slab 0 + slab 1 + ...), deterministic. A workgroup owns 64 consecutive 16-byte elements; its W
// waves each sum a contiguous range of the slabs (in order, eight loads in flight) and wave 0 adds the W range sums in
// order. W follows the split count (small filters have up to 512 slabs: one thread walking them was a chain of 128
// dependent memory round trips), W = 1 is the plain in-order sum.
// Workgroups from `dw_blocks` on (only launched with a folded bias gradient) add the per-split bias partials
// bpart[split][K] the same way, one thread per channel.
template <int W>
__global__ void __launch_bounds__(64 * W) wgrad_reduce_kernel(float *__restrict__ dw, const float *__restrict__ slabs, size_t n,
                                                              int splits, float alpha, float beta, int dw_blocks,
                                                              const float *__restrict__ bpart, float *__restrict__ db, int K) {
	typedef float f4 __attribute__((ext_vector_type(4), aligned(4)));
	__shared__ f4 sh[W > 1 ? W : 1][64];
	if ((int)blockIdx.x >= dw_blocks) {
		const int k = ((int)blockIdx.x - dw_blocks) * (64 * W) + (int)threadIdx.x;
		if (k < K) {
			float r = 0.f;
			for (int sp = 0; sp < splits; ++sp) r += bpart[(size_t)sp * K + k];
			db[k] = (beta == 0.f ? 0.f : beta * db[k]) + alpha * r;
		}
		return;
	}
	const size_t n4 = n >> 2;
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const size_t i = (size_t)blockIdx.x * 64 + lane;
	const int per = (splits + W - 1) / W, k0 = w * per, k1 = min(k0 + per, splits);

	f4 s = {0.f, 0.f, 0.f, 0.f};
	if (i < n4) {
		for (int k = k0; k < k1; k += 8) {
			f4 v[8];
#pragma unroll
			for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const f4 *>(slabs + (size_t)min(k + q, k1 - 1) * n + 4 * i);
#pragma unroll
			for (int q = 0; q < 8; ++q)
				if (k + q < k1) s += v[q];
		}
	}
	if (W > 1) {
		sh[w][lane] = s;
		__syncthreads();
		if (w != 0) return;
		s = sh[0][lane];
#pragma unroll
		for (int q = 1; q < W; ++q) s += sh[q][lane];
	}
	if (i < n4) {
		f4 *o = reinterpret_cast<f4 *>(dw + 4 * i);
		const f4 old = beta == 0.f ? f4{0.f, 0.f, 0.f, 0.f} : *o;
		*o = (beta == 0.f ? f4{0.f, 0.f, 0.f, 0.f} : beta * old) + alpha * s;
	}
	// the n % 4 trailing elements
	if (blockIdx.x == 0 && w == 0 && (n4 << 2) + lane < n) {
		const size_t t = (n4 << 2) + lane;
		float r = 0.f;
		for (int k = 0; k < splits; ++k) r += slabs[(size_t)k * n + t];
		dw[t] = (beta == 0.f ? 0.f : beta * dw[t]) + alpha * r;
	}
}

inline void launch_wgrad_reduce(float *dw, const float *slabs, size_t n, int splits, float alpha, float beta, hipStream_t st,
                                const float *bpart = nullptr, float *db = nullptr, int K = 0) {
	const int blocks = (int)(((n >> 2) + 63) / 64) + ((n >> 2) == 0 ? 1 : 0);
	const int W = splits >= 128 ? 16 : splits >= 32 ? 4 : 1;
	const int extra = db ? pz::ceil_div(K, 64 * W) : 0;
	if (W == 16)
		wgrad_reduce_kernel<16><<<blocks + extra, 1024, 0, st>>>(dw, slabs, n, splits, alpha, beta, blocks, bpart, db, K);
	else if (W == 4)
		wgrad_reduce_kernel<4><<<blocks + extra, 256, 0, st>>>(dw, slabs, n, splits, alpha, beta, blocks, bpart, db, K);
	else
		wgrad_reduce_kernel<1><<<blocks + extra, 64, 0, st>>>(dw, slabs, n, splits, alpha, beta, blocks, bpart, db, K);
}

// db[k] = beta*db[k] + alpha * sum_{n,pq} dy[n,k,pq], two deterministic stages: workgroup (k, s) sums the images
// n = s (mod S) of channel k with 16-byte (4-byte aligned) loads, four in flight; the finish kernel adds the S partials
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

__global__ void __launch_bounds__(256) bias_grad_partial_kernel(const float *__restrict__ dy, float *__restrict__ part, int N, int K,
                                                                 int PQ, int S) {
	__shared__ float red[16];
	const int k = blockIdx.x, s = blockIdx.y;
	const int n4 = PQ >> 2, rem = PQ & 3;
	float acc = 0.f;

	// flat index over (image of this split, 4-float group)
	const int nimg = (N - s + S - 1) / S;
	const int items = nimg * n4;
	for (int i0 = threadIdx.x; i0 < items; i0 += 4 * 256) {
		f4u v[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const int i = i0 + u * 256;
			v[u] = f4u{0.f, 0.f, 0.f, 0.f};
			if (i < items) {
				const int img = i / n4, g = i - img * n4;
				v[u] = *reinterpret_cast<const f4u *>(dy + ((size_t)(s + img * S) * K + k) * PQ + 4 * g);
			}
		}
#pragma unroll
		for (int u = 0; u < 4; ++u) acc += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
	}
	for (int i = threadIdx.x; i < nimg * rem; i += 256) {
		const int img = i / rem, e = i - img * rem;
		acc += dy[((size_t)(s + img * S) * K + k) * PQ + 4 * n4 + e];
	}

	acc = block_sum(acc, red);
	if (threadIdx.x == 0) part[(size_t)k * S + s] = acc;
}

__global__ void __launch_bounds__(256) bias_grad_finish_kernel(const float *__restrict__ part, float *__restrict__ db, int K, int S,
                                                                float alpha, float beta) {
	const int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= K) return;
	float s = 0.f;
	for (int i = 0; i < S; ++i) s += part[(size_t)k * S + i];
	db[k] = (beta == 0.f ? 0.f : beta * db[k]) + alpha * s;
}

inline int bias_grad_splits(int n, int k) {
	int s = pz::ceil_div(8 * pz::kNumCU, k);
	s = s > n ? n : s;
	return s > 64 ? 64 : (s < 1 ? 1 : s);
}

// last block of a backward-filter workspace: the bias-gradient partials, of the two-stage kernels above or (one row per
// split of the filter-gradient launch) of the form folded into wgrad_conv_kernel
inline size_t bias_part_bytes(const pz_conv_desc *d, int wgrad_splits) {
	const int s = bias_grad_splits(d->n, d->k);
	return (((size_t)d->k * (s > wgrad_splits ? s : wgrad_splits) * sizeof(float)) + 255) & ~(size_t)255;
}

// single-stage variant (one workgroup per channel) for callers without workspace
__global__ void __launch_bounds__(256) bias_grad_kernel(const float *__restrict__ dy, float *__restrict__ db, int N, int K, int PQ,
                                                         float alpha, float beta) {
	__shared__ float red[16];
	const int k = blockIdx.x;
	float s = 0.f;
	for (int n = 0; n < N; ++n) {
		const float *row = dy + ((size_t)n * K + k) * PQ;
		for (int i = threadIdx.x; i < PQ; i += blockDim.x) s += row[i];
	}
	s = block_sum(s, red);
	if (threadIdx.x == 0) db[k] = (beta == 0.f ? 0.f : beta * db[k]) + alpha * s;
}

// ------------------------------------------------------------------------------------------------
// direct (one thread per output) kernels: ConvFwdAlgo.direct and the fallback for configurations the
// implicit-GEMM path does not take (stride > 1 together with dilation > 1 in backward-data)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) direct_fwd_kernel(pz_conv_desc d, int P, int Q, const float *__restrict__ x,
                                                          const float *__restrict__ w, const float *__restrict__ bias,
                                                          float *__restrict__ y) {
	const int Cg = d.c / d.groups, Kg = d.k / d.groups;
	const size_t total = (size_t)d.n * d.k * P * Q;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
		const int q = i % Q, p = (i / Q) % P, k = (i / ((size_t)Q * P)) % d.k, n = i / ((size_t)Q * P * d.k);
		const int g = k / Kg;
		float s = bias ? bias[k] : 0.f;
		for (int c = 0; c < Cg; ++c)
			for (int r = 0; r < d.r; ++r) {
				const int hh = p * d.stride_h - d.pad_h + r * d.dil_h;
				if ((unsigned)hh >= (unsigned)d.h) continue;
				for (int ss = 0; ss < d.s; ++ss) {
					const int ww = q * d.stride_w - d.pad_w + ss * d.dil_w;
					if ((unsigned)ww >= (unsigned)d.w) continue;
					s += x[(((size_t)n * d.c + g * Cg + c) * d.h + hh) * d.w + ww] * w[(((size_t)k * Cg + c) * d.r + r) * d.s + ss];
				}
			}
		y[i] = s;
	}
}

__global__ void __launch_bounds__(256) direct_bwd_data_kernel(pz_conv_desc d, int P, int Q, const float *__restrict__ dy,
                                                               const float *__restrict__ w, float *__restrict__ dx) {
	const int Cg = d.c / d.groups, Kg = d.k / d.groups;
	const size_t total = (size_t)d.n * d.c * d.h * d.w;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
		const int ww = i % d.w, hh = (i / d.w) % d.h, c = (i / ((size_t)d.w * d.h)) % d.c, n = i / ((size_t)d.w * d.h * d.c);
		const int g = c / Cg, cl = c - g * Cg;
		float s = 0.f;
		for (int r = 0; r < d.r; ++r) {
			const int ph = hh + d.pad_h - r * d.dil_h;
			if (ph < 0 || ph % d.stride_h) continue;
			const int p = ph / d.stride_h;
			if (p >= P) continue;
			for (int ss = 0; ss < d.s; ++ss) {
				const int qw = ww + d.pad_w - ss * d.dil_w;
				if (qw < 0 || qw % d.stride_w) continue;
				const int q = qw / d.stride_w;
				if (q >= Q) continue;
				for (int kl = 0; kl < Kg; ++kl) {
					const int k = g * Kg + kl;
					s += dy[(((size_t)n * d.k + k) * P + p) * Q + q] * w[(((size_t)k * Cg + cl) * d.r + r) * d.s + ss];
				}
			}
		}
		dx[i] = s;
	}
}

// one workgroup per filter element
__global__ void __launch_bounds__(256) direct_bwd_filter_kernel(pz_conv_desc d, int P, int Q, const float *__restrict__ x,
                                                                 const float *__restrict__ dy, float *__restrict__ dw,
                                                                 float alpha, float beta) {
	__shared__ float red[16];
	const int Cg = d.c / d.groups, Kg = d.k / d.groups;
	const int e = blockIdx.x;
	const int ss = e % d.s, r = (e / d.s) % d.r, cl = (e / (d.s * d.r)) % Cg, k = e / (d.s * d.r * Cg);
	const int g = k / Kg;

	float s = 0.f;
	const int npq = d.n * P * Q;
	for (int i = threadIdx.x; i < npq; i += blockDim.x) {
		const int q = i % Q, p = (i / Q) % P, n = i / (Q * P);
		const int hh = p * d.stride_h - d.pad_h + r * d.dil_h, ww = q * d.stride_w - d.pad_w + ss * d.dil_w;
		if ((unsigned)hh < (unsigned)d.h && (unsigned)ww < (unsigned)d.w)
			s += x[(((size_t)n * d.c + g * Cg + cl) * d.h + hh) * d.w + ww] * dy[(((size_t)n * d.k + k) * P + p) * Q + q];
	}
	s = block_sum(s, red);
	if (threadIdx.x == 0) dw[e] = (beta == 0.f ? 0.f : beta * dw[e]) + alpha * s;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
inline size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

struct FwdPlan {
	int bm, bn;                 // tile
	int tiles_m, tiles_n, mpad, kred, kred_pad;
	int full_tiles, tail_splits, blocks;     // tail balancing (see IgemmArgs)
	int split;                               // 0, or the number of bf16 partial products (pz::g_conv_math) of a tap-major problem
	size_t wp_bytes, tab_bytes, slab_bytes;
};

// chans = reduction channels per tap (the tap-major order needs them in whole k-tiles)
FwdPlan plan_igemm(int M, int kred, long npix, int groups, int chans) {
	FwdPlan p;
	p.split = chans % 16 == 0 ? pz::g_conv_math : 0;
	// 64 x 256 tiles when 64-row tiles pad the channel axis less than 128-row ones (M <= 64, 160, 192, 320, ...)
	const bool narrow = M <= 64 || pz::ceil_div(M, 64) * 64 < pz::ceil_div(M, 128) * 128;
	p.bm = narrow ? 64 : 128;
	p.bn = narrow ? 256 : 128;
#if PZ_IG_TALL
	// 256 x 128 tiles (8 waves) when the row axis fills them: half the pixel gathers per MFMA
	if (!narrow && M >= 256 && M % 256 == 0) p.bm = 256;
#endif
	p.tiles_m = pz::ceil_div(M, p.bm);
	p.tiles_n = pz::ceil_div(npix, p.bn);
	p.mpad = p.tiles_m * p.bm;
	p.kred = kred;
	p.kred_pad = pz::ceil_div(kred, 16) * 16;
	p.wp_bytes = align256((size_t)groups * p.kred_pad * p.mpad * (p.split ? 6 : 4));
	p.tab_bytes = align256((size_t)p.kred_pad * sizeof(int2));

	// The matrix pipes bound the kernel, so a launch takes ceil(tiles / #CU) tile-times: a last round that fills only
	// part of the chip is cut along k so that every CU gets an equal share of it (deterministic slab reduce).
	const int tiles = p.tiles_m * p.tiles_n, nk = p.kred_pad / 16;
	const int rem = (int)((long)tiles * groups % pz::kNumCU);
	p.full_tiles = tiles, p.tail_splits = 1;
	if (groups == 1 && rem != 0) {
		// cost of the last round in tile-times: unsplit = 1; split s ways = ceil(rem*s / #CU) / s. Slices keep >= 4
		// k-tiles and the slab stays <= 1024 tiles (64 MB).
		int best = 1;
		double best_cost = 0.92;                   // only split for a >= 8 % shorter last round (0.6 / 0.75 / never measured worse)
		for (int sp = 2; sp <= nk / 4 && rem * sp <= 1024; ++sp) {
			const double cost = (double)pz::ceil_div((long)rem * sp, pz::kNumCU) / sp;
			if (cost < best_cost - 1e-9) best_cost = cost, best = sp;
		}
		// ... and the slab reduce is a launch of its own (~12 us): a k-tile takes ~1 us of a CU, so the round has to get
		// PZ_TAIL_MIN_GAIN k-tile-times shorter to pay for it (thresholds 8 ... 40 measured within 0.1 ms per ResNet-50 step)
		if ((1.0 - best_cost) * nk < PZ_TAIL_MIN_GAIN) best = 1;
		if (best > 1) p.full_tiles = tiles - rem, p.tail_splits = best;
	}
	// A launch of few, long tiles (a 5x5 layer on 16x16 maps at batch 128: 256 tiles of 300 k-tiles) leaves a CU one or two
	// workgroups, whose four waves cannot keep the matrix pipes busy on their own: every tile is cut along k until the chip
	// holds ~4 workgroups per CU (NiN's 96 <- 192 5x5 backward-data: 0.41 -> 0.30 ms); slices keep >= 8 k-tiles, slabs <= 64 MB.
	if (groups == 1 && p.tail_splits == 1 && tiles <= 2 * pz::kNumCU && nk >= 32) {
		int sp = pz::ceil_div(4 * pz::kNumCU, tiles);
		if (sp > nk / 8) sp = nk / 8;
		if (sp > 1024 / tiles) sp = 1024 / tiles;
		if (sp >= 2) p.full_tiles = 0, p.tail_splits = sp;
	}
	const int ntail = tiles - p.full_tiles;
	p.blocks = p.full_tiles + ntail * p.tail_splits;
	p.slab_bytes = p.tail_splits > 1 ? align256((size_t)ntail * p.tail_splits * p.bm * p.bn * sizeof(float)) : 0;
	return p;
}

int check_desc(const pz_conv_desc *d, int *P, int *Q) {
	PZ_REQUIRE(d != nullptr, "conv: null descriptor");
	PZ_REQUIRE(d->n > 0 && d->c > 0 && d->h > 0 && d->w > 0 && d->k > 0 && d->r > 0 && d->s > 0, "conv: non-positive dimension");
	PZ_REQUIRE(d->groups >= 1 && d->c % d->groups == 0 && d->k % d->groups == 0,
	           "conv: %d input / %d output maps not divisible by %d groups", d->c, d->k, d->groups);
	PZ_REQUIRE(d->stride_h >= 1 && d->stride_w >= 1 && d->dil_h >= 1 && d->dil_w >= 1 && d->pad_h >= 0 && d->pad_w >= 0,
	           "conv: invalid stride/dilation/pad");
	const int eh = d->h + 2 * d->pad_h - d->dil_h * (d->r - 1) - 1, ew = d->w + 2 * d->pad_w - d->dil_w * (d->s - 1) - 1;
	PZ_REQUIRE(eh >= 0 && ew >= 0, "conv: filter larger than padded input");
	*P = eh / d->stride_h + 1;
	*Q = ew / d->stride_w + 1;
	// (filters of more than 63 taps or more than 31 rows / columns — a sentence-wide 3 x 128 filter, Models/Nets/SentiNet.py:23 — are
	// served by the one-thread-per-output kernels: igemm_eligible keeps them off the tap tables, whose sentinel is kPadTap)
	PZ_REQUIRE((long)d->r * d->dil_h < (1 << 16) && (long)d->s * d->dil_w < (1 << 16), "conv: filter extent too large");
	PZ_REQUIRE((size_t)d->c * d->h * d->w < ((size_t)1 << 31) && (size_t)d->k * *P * *Q < ((size_t)1 << 31),
	           "conv: one image exceeds 2^31 elements");
	PZ_REQUIRE((size_t)d->n * *P * *Q < ((size_t)1 << 31) && (size_t)d->n * d->h * d->w < ((size_t)1 << 31),
	           "conv: pixel count exceeds 2^31");
	return PZ_OK;
}

// 128 x 128 tap-major launches with at least this many k-tiles take the two-tiles-ahead form of the kernel (PF2): measured on
// the ResNet-50 census (profiles/r04_igemm_pf2_census.txt) -3..-8 % from 32 k-tiles up (the default), even at 16, +3..7 % at 8 (the longer
// prologue and 3 instead of 4 waves per SIMD are not paid back by 8 tiles). PUZZLE_MI355_IG_PF2 = minimum (0: never) — A/B aid.
static int ig_prefetch2_min_ktiles() {
	static const int n = [] {
		const char *e = getenv("PUZZLE_MI355_IG_PF2");
		const int v = e ? atoi(e) : 32;
		return v <= 0 ? (1 << 24) : v;
	}();
	return n;
}

template <int BM, int BN, int WM, int WN>
void launch_igemm(const FwdPlan &p, const IgemmArgs &a, int groups, hipStream_t st, double flops) {
	constexpr int lds_pad = 0;
	// the instantiation the two-tiles-ahead form exists for; every other tile falls through to the plain kernels
	constexpr bool kHasPF2 = BM == 128 && BN == 128 && WM * WN == 4;
	{
		// profile bracket = the MFMA kernel alone (what rocprofv3 lists under its name); all of the launch's algorithmic
		// FLOP are its work — the slab reduce of a k-sliced last round only adds
		ProfScope prof(st, BM == 64 ? 1 : 0, flops);
		if constexpr (WM * WN == 4) {
			if (p.split) {
				if (p.split == 6 && a.x2)
					igemm_split_kernel<BM, BN, WM, WN, true, 6><<<dim3(p.blocks, 1, groups), 64 * WM * WN, lds_pad, st>>>(a);
				else if (p.split == 6)
					igemm_split_kernel<BM, BN, WM, WN, false, 6><<<dim3(p.blocks, 1, groups), 64 * WM * WN, lds_pad, st>>>(a);
				else if (a.x2)
					igemm_split_kernel<BM, BN, WM, WN, true, 9><<<dim3(p.blocks, 1, groups), 64 * WM * WN, lds_pad, st>>>(a);
				else
					igemm_split_kernel<BM, BN, WM, WN, false, 9><<<dim3(p.blocks, 1, groups), 64 * WM * WN, lds_pad, st>>>(a);
				goto launched;
			}
		}
		if (a.xbn) {                             // (conv2d_fwd_impl admits it for this instantiation only: xbn_fwd_eligible)
			if constexpr (kHasPF2)
				igemm_conv_kernel<BM, BN, WM, WN, true, false, false, true><<<dim3(p.blocks, 1, groups), 64 * WM * WN, lds_pad, st>>>(a);
		} else if (a.x2)
			igemm_conv_kernel<BM, BN, WM, WN, true, true><<<dim3(p.blocks, 1, groups), 64 * WM * WN, lds_pad, st>>>(a);
		else if (kHasPF2 && a.tapmajor && a.kred_pad >= 16 * ig_prefetch2_min_ktiles()) {
			if constexpr (kHasPF2)
				igemm_conv_kernel<BM, BN, WM, WN, true, false, true><<<dim3(p.blocks, 1, groups), 64 * WM * WN, lds_pad, st>>>(a);
		} else if (a.tapmajor)
			igemm_conv_kernel<BM, BN, WM, WN, true><<<dim3(p.blocks, 1, groups), 64 * WM * WN, lds_pad, st>>>(a);
		else
			igemm_conv_kernel<BM, BN, WM, WN, false><<<dim3(p.blocks, 1, groups), 64 * WM * WN, lds_pad, st>>>(a);
	launched:;
	}
	if (p.tail_splits > 1)
		igemm_tail_reduce_kernel<BM, BN, WM, WN><<<dim3((a.tiles_m * a.tiles_n - p.full_tiles) * (WM * WN) * (BM / WM / 32), 1, groups), 64, 0, st>>>(a);
}

void run_igemm(const FwdPlan &p, IgemmArgs a, float *slabs, int groups, hipStream_t st, double flops) {
	a.tiles_m = p.tiles_m, a.tiles_n = p.tiles_n;
	a.full_tiles = p.full_tiles, a.tail_splits = p.tail_splits, a.slabs = slabs;
	if (p.bm == 256)
		launch_igemm<256, 128, 4, 2>(p, a, groups, st, flops);
	else if (p.bm == 64)
		launch_igemm<64, 256, 1, 4>(p, a, groups, st, flops);
	else
		launch_igemm<128, 128, 2, 2>(p, a, groups, st, flops);
}

// backward-data residue classes
struct DgradClass {
	int a_h, a_w, Rc, Sc, oo_h, oo_w, pad_h, pad_w, Pv, Qv;
};

int dgrad_classes(const pz_conv_desc *d, DgradClass *cls, bool *needs_zero) {
	int n = 0;
	*needs_zero = false;
	for (int ah = 0; ah < d->stride_h; ++ah)
		for (int aw = 0; aw < d->stride_w; ++aw) {
			DgradClass c;
			c.a_h = ah, c.a_w = aw;
			c.Rc = ah < d->r ? (d->r - ah + d->stride_h - 1) / d->stride_h : 0;
			c.Sc = aw < d->s ? (d->s - aw + d->stride_w - 1) / d->stride_w : 0;
			c.oo_h = ((ah - d->pad_h) % d->stride_h + d->stride_h) % d->stride_h;
			c.oo_w = ((aw - d->pad_w) % d->stride_w + d->stride_w) % d->stride_w;
			c.Pv = c.oo_h < d->h ? (d->h - c.oo_h + d->stride_h - 1) / d->stride_h : 0;
			c.Qv = c.oo_w < d->w ? (d->w - c.oo_w + d->stride_w - 1) / d->stride_w : 0;
			if (c.Pv == 0 || c.Qv == 0) continue;               // no output pixel in this class
			if (c.Rc == 0 || c.Sc == 0) {
				*needs_zero = true;                              // pixels exist but no tap reaches them
				continue;
			}
			const int delta_h = (c.oo_h + d->pad_h - ah) / d->stride_h, delta_w = (c.oo_w + d->pad_w - aw) / d->stride_w;
			c.pad_h = (c.Rc - 1) * d->dil_h - delta_h;
			c.pad_w = (c.Sc - 1) * d->dil_w - delta_w;
			cls[n++] = c;
		}
	return n;
}

// the MFMA kernels address tensors through buffer descriptors (32-bit byte offsets) and keep per-pixel tap masks in
// registers: tensors must stay below 4 GiB and the filter below 64 taps (31 per side for backward-filter)
bool igemm_eligible(const pz_conv_desc *d, int P, int Q) {
	const size_t xb = (size_t)d->n * d->c * d->h * d->w * 4, yb = (size_t)d->n * d->k * P * Q * 4;
	// the backward-filter kernel splits run indices by multiply-high division, exact while runs * divisor < 2^32
	const unsigned long long runs = (unsigned long long)d->n * P * ((Q + 3) / 4);
	if (runs * (unsigned long long)std::max(P, (Q + 3) / 4) >= (1ull << 32)) return false;
	return xb < kOOB && yb < kOOB && d->r * d->s <= 63 && d->r <= 31 && d->s <= 31;
}

bool dgrad_uses_igemm(const pz_conv_desc *d) {
	const bool strided = d->stride_h > 1 || d->stride_w > 1, dilated = d->dil_h > 1 || d->dil_w > 1;
	return !(strided && dilated) && d->stride_h <= 4 && d->stride_w <= 4;
}

struct WgradPlan {
	int bm, bn, tiles_m, tiles_n, ncrs, ncrs_pad, steps_total, steps_per_split, splits;
	int runs;             // 4-pixel runs per k-step of the kernel instantiation the plan is for
	size_t tab_bytes, slab_elems;
};

// Backward-filter walks the pixels in runs of 4 inside a row, rows padded to whole runs: 14-wide maps execute 16/14 of
// their MFMAs and loads, 7-wide ones 8/7. For a pointwise, unit-stride, unpadded layer x and dy are the same plane layout
// and the sum over pixels does not care how the plane is cut into rows, so it is read as ONE row of P*Q pixels: only the
// plane's last run is partial (7 x 7 -> 13 runs instead of 14, 14 x 14 -> 49 instead of 56).
static void wgrad_plane(const pz_conv_desc *d, int P, int Q, int *rows, int *cols) {
	*rows = P, *cols = Q;
	if (d->r == 1 && d->s == 1 && d->pad_h == 0 && d->pad_w == 0 && d->stride_h == 1 && d->stride_w == 1) *rows = 1, *cols = P * Q;
}

WgradPlan plan_wgrad(const pz_conv_desc *d, int P, int Q) {
	WgradPlan p;
	wgrad_plane(d, P, Q, &P, &Q);
	const int Kg = d->k / d->groups, Cg = d->c / d->groups;
	p.ncrs = Cg * d->r * d->s;
	p.bm = (Kg <= 64 || pz::ceil_div(Kg, 64) * 64 < pz::ceil_div(Kg, 128) * 128) ? 64 : 128;
	// the narrower column tile when it pads less (conv1: 147 columns = 3 x 64 rather than 2 x 128)
	p.bn = (p.ncrs <= 64 || pz::ceil_div(p.ncrs, 64) * 64 < pz::ceil_div(p.ncrs, 128) * 128) ? 64 : 128;
	// the stem (64 x 147 filter gradient, strided gather): one 64 x 192 tile per workgroup — the dy runs are fetched once
	// instead of once per 64-column tile, and a wave's 32 x 96 share does 3 MFMAs per 4 fragment reads instead of 1 per 2
	if (p.bm == 64 && Kg <= 64 && p.ncrs > 128 && p.ncrs <= 192 && d->stride_w > 1 && d->r * d->s > 1 && PZ_WG_RUNS == 8 && PZ_WG_WAVES == 4)
		p.bn = 192;      // (not a pointwise filter: those may arrive with a BatchNorm fold, which this instantiation does not carry)
	p.runs = PZ_WG_RUNS;
	// 129..192 output maps under a filter with taps (NiN's 5x5 layers: 192 maps, 2 400 gathered columns): ONE 192-row tile
	// instead of three 64-row ones — the x runs, which every tap gathers again, are fetched once instead of three times and a
	// wave's 96 x 64 share does 6 MFMAs per 5 fragment reads instead of 2 per 3; 16-pixel k-steps keep two such workgroups
	// on a CU (82 KB of LDS each otherwise). Only where the gathered side is wide (>= 4 column tiles): a 192 x 75 gradient would
	// be ONE tile cut into 512 slabs, whose reduce costs more than the tile saves.
	if (Kg > 128 && Kg <= 192 && p.bn == 128 && p.ncrs >= 512 && d->r * d->s > 1 && d->stride_w == 1 && PZ_WG_WAVES == 4) p.bm = 192, p.runs = 4;
	p.tiles_m = pz::ceil_div(Kg, p.bm);
	p.tiles_n = pz::ceil_div(p.ncrs, p.bn);
	p.ncrs_pad = p.tiles_n * p.bn;
	const long nruns = (long)d->n * P * ((Q + 3) / 4);         // reduction axis in runs of 4 pixels (rows padded to 4)
	p.steps_total = pz::ceil_div(nruns, p.runs);

	const int tiles = p.tiles_m * p.tiles_n * d->groups;
	int splits = (PZ_WG_RUNS == 8 || p.bm == 192 ? 2 : 4) * pz::kNumCU / tiles;       // workgroups that fit a CU (LDS): one balanced round
	const int min_steps = 8 * 8 / p.runs;                                             // >= 256 pixels per split
	const int max_by_work = p.steps_total / min_steps > 0 ? p.steps_total / min_steps : 1;
	if (splits > max_by_work) splits = max_by_work;
	if (splits < 1) splits = 1;
	p.steps_per_split = pz::ceil_div(p.steps_total, splits);
	p.splits = pz::ceil_div(p.steps_total, p.steps_per_split);
	p.tab_bytes = align256((size_t)p.ncrs_pad * sizeof(int2));
	p.slab_elems = (size_t)d->groups * Kg * p.ncrs;
	return p;
}

// Which backward-filter problems take wgrad_split_kernel in a split math mode: pointwise, unit stride, and at least 128
// channels on both sides — with 64 the layer is bound by HBM (55x55 maps), and the fp32 kernel's 128-byte row pieces move
// the bytes faster than the split kernel's 64-byte ones (measured 0.30 vs 0.36-0.39 ms on the three stage-2 shapes).
bool wgrad_split_eligible(const pz_conv_desc *d) {
	return pz::g_conv_math != 0 && d->r == 1 && d->s == 1 && d->pad_h == 0 && d->pad_w == 0 && d->stride_h == 1 && d->stride_w == 1 &&
	       d->k / d->groups >= 128 && d->c / d->groups >= 128;
}

// k-steps of 2 cells = 16 pixels of an image plane
WgradPlan plan_wgrad_split(const pz_conv_desc *d, int P, int Q) {
	WgradPlan p;
	const int Kg = d->k / d->groups, Cg = d->c / d->groups;
	p.ncrs = Cg;
	p.bm = p.bn = 128;
	p.tiles_m = pz::ceil_div(Kg, p.bm);
	p.tiles_n = pz::ceil_div(p.ncrs, p.bn);
	p.ncrs_pad = p.tiles_n * p.bn;
	const long cells = (long)d->n * pz::ceil_div(P * Q, 8);
	p.steps_total = (int)pz::ceil_div(cells, 2);

	const int tiles = p.tiles_m * p.tiles_n * d->groups;
	int splits = 3 * pz::kNumCU / tiles;                           // three workgroups fit a CU (LDS, registers): one balanced round
	const int max_by_work = p.steps_total / 16 > 0 ? p.steps_total / 16 : 1;   // >= 16 k-steps (256 pixels) per split
	if (splits > max_by_work) splits = max_by_work;
	if (splits < 1) splits = 1;
	p.steps_per_split = pz::ceil_div(p.steps_total, splits);
	p.splits = pz::ceil_div(p.steps_total, p.steps_per_split);
	p.tab_bytes = 0;
	p.runs = 0;
	p.slab_elems = (size_t)d->groups * Kg * p.ncrs;
	return p;
}

// the Winograd F(2x2, 3x3) path (wino.hip) serves the 3x3 / stride-1 forward and backward-data passes: always when asked
// for by name, and by default from 32 channels on either side (1.6-1.8x the implicit GEMM on every 3x3 layer of the
// ResNet-50 census, tools/wino_check.py; below that the 64-channel workgroup block is mostly padding)
bool uses_winograd(const pz_conv_desc *d, int which, int P, int Q, int algo) {
	if (algo != PZ_CONV_ALGO_WINOGRAD && algo != PZ_CONV_ALGO_AUTO) return false;
	if (!(which == PZ_CONV_BWD_FILTER ? pz::wino_wgrad_eligible(d, P, Q) : pz::wino_eligible(d, which, P, Q))) return false;
	return algo == PZ_CONV_ALGO_WINOGRAD || (d->c >= 32 && d->k >= 32);
}

// One place decides which kernel family serves a request — execution, workspace sizes and pz_conv2d_algo_used all ask
// here: Winograd first, then the thin backward-data kernels (reported as `direct`: vector-ALU kernels), then the implicit
// GEMM, else the one-thread-per-output kernels.
enum ConvPath { PATH_DIRECT, PATH_WINOGRAD, PATH_THIN, PATH_IGEMM };

ConvPath conv_path(const pz_conv_desc *d, int which, int P, int Q, int algo) {
	if (uses_winograd(d, which, P, Q, algo)) return PATH_WINOGRAD;
	if (which == PZ_CONV_BWD_DATA && algo == PZ_CONV_ALGO_AUTO && pz::thin_dgrad_eligible(d, P, Q)) return PATH_THIN;
	if (algo == PZ_CONV_ALGO_DIRECT || !igemm_eligible(d, P, Q) || (which == PZ_CONV_BWD_DATA && !dgrad_uses_igemm(d))) return PATH_DIRECT;
	return PATH_IGEMM;
}

}  // namespace

extern "C" {

int pz_conv_math_set(int products) {
	PZ_REQUIRE(products == 0 || products == 6 || products == 9, "pz_conv_math_set: %d is not one of 0 (f32 MFMA), 6, 9 (bf16 partial products)", products);
	pz::g_conv_math = products;
	return PZ_OK;
}

int pz_conv_math_get(int *products) {
	PZ_REQUIRE(products != nullptr, "pz_conv_math_get: null output");
	*products = pz::g_conv_math;
	return PZ_OK;
}

int pz_conv_profile_enable(int on) {
	g_prof_on = on != 0;
	return PZ_OK;
}

int pz_conv_profile_collect(double total_ms[PZ_CONV_PROFILE_FAMILIES], double total_flops[PZ_CONV_PROFILE_FAMILIES],
                            long long launches[PZ_CONV_PROFILE_FAMILIES]) {
	PZ_HIP(hipDeviceSynchronize());
	for (int f = 0; f < PZ_CONV_PROFILE_FAMILIES; ++f) total_ms[f] = 0.0, total_flops[f] = 0.0, launches[f] = 0;
	for (ProfRec &r : g_prof) {
		float ms = 0.f;
		(void)hipEventElapsedTime(&ms, r.a, r.b);
		total_ms[r.family] += ms;
		total_flops[r.family] += r.flops;
		launches[r.family] += 1;
		(void)hipEventDestroy(r.a);
		(void)hipEventDestroy(r.b);
	}
	g_prof.clear();
	return PZ_OK;
}

int pz_conv2d_out_shape(const pz_conv_desc *d, int *p, int *q) { return check_desc(d, p, q); }

int pz_conv2d_algo_used(const pz_conv_desc *d, int which, int algo, int *used) {
	int P, Q;
	if (int rc = check_desc(d, &P, &Q)) return rc;
	PZ_REQUIRE(used != nullptr && which >= PZ_CONV_FWD && which <= PZ_CONV_BWD_FILTER, "pz_conv2d_algo_used: bad arguments");
	const ConvPath path = conv_path(d, which, P, Q, algo);
	*used = path == PATH_WINOGRAD ? PZ_CONV_ALGO_WINOGRAD : path == PATH_IGEMM ? PZ_CONV_ALGO_IMPLICIT_GEMM : PZ_CONV_ALGO_DIRECT;
	return PZ_OK;
}

// `prepared`: the pass is going to be handed a prepared filter operand (pz_conv2d_{fwd,bwd_data}_pre): the workspace then
// only holds what the launch itself needs (the slabs of k-sliced tiles), not a second copy of the packed filters
static int conv_workspace_bytes(const pz_conv_desc *d, int which, int algo, bool prepared, size_t *nbytes) {
	int P, Q;
	if (int rc = check_desc(d, &P, &Q)) return rc;
	PZ_REQUIRE(nbytes != nullptr, "pz_conv2d_workspace_bytes: null output");
	PZ_REQUIRE(which >= PZ_CONV_FWD && which <= PZ_CONV_BWD_FILTER, "pz_conv2d_workspace_bytes: unknown pass %d", which);
	*nbytes = 0;
	const ConvPath path = conv_path(d, which, P, Q, algo);
	if (path == PATH_WINOGRAD) {
		// forward / backward-data: [transformed filters unless prepared | transformed input where the launch takes it from a pass of its own]
		*nbytes = which == PZ_CONV_BWD_FILTER
		              ? align256(pz::wino_wgrad_workspace_bytes(d, P, Q)) + align256((size_t)d->k * bias_grad_splits(d->n, d->k) * sizeof(float))
		              : (prepared ? 0 : align256(pz::wino_workspace_bytes(d, which, P, Q))) + pz::wino_input_bytes(d, which, P, Q);
		return PZ_OK;
	}
	if (path == PATH_THIN) {
		*nbytes = align256(pz::thin_dgrad_workspace_bytes(d));
		return PZ_OK;
	}
	if (path == PATH_DIRECT) return PZ_OK;

	const int Kg = d->k / d->groups, Cg = d->c / d->groups;

	if (which == PZ_CONV_FWD) {
		FwdPlan p = plan_igemm(Kg, Cg * d->r * d->s, (long)d->n * P * Q, d->groups, Cg);
		*nbytes = prepared && !p.split ? p.slab_bytes : p.wp_bytes + p.tab_bytes + p.slab_bytes;

	} else if (which == PZ_CONV_BWD_DATA) {
		DgradClass cls[16];
		bool nz;
		const int nc = dgrad_classes(d, cls, &nz);
		size_t total = 0;
		for (int i = 0; i < nc; ++i) {
			FwdPlan p = plan_igemm(Cg, Kg * cls[i].Rc * cls[i].Sc, (long)d->n * cls[i].Pv * cls[i].Qv, d->groups, Kg);
			total += p.wp_bytes + p.tab_bytes + p.slab_bytes;
		}
		*nbytes = total;

	} else {
		WgradPlan p = wgrad_split_eligible(d) ? plan_wgrad_split(d, P, Q) : plan_wgrad(d, P, Q);
		*nbytes = p.tab_bytes + (p.splits > 1 ? align256(p.slab_elems * p.splits * sizeof(float)) : 0) +
		          bias_part_bytes(d, p.splits);                                                 // bias-gradient partials
	}
	return PZ_OK;
}

int pz_conv2d_workspace_bytes(const pz_conv_desc *d, int which, int algo, size_t *nbytes) {
	return conv_workspace_bytes(d, which, algo, false, nbytes);
}

int pz_conv2d_workspace_bytes_pre(const pz_conv_desc *d, int which, int algo, size_t *nbytes) {
	return conv_workspace_bytes(d, which, algo, true, nbytes);
}

int pz_conv2d_fwd_stats_strips(const pz_conv_desc *d, int algo, int *strips) {
	int P, Q;
	if (int rc = check_desc(d, &P, &Q)) return rc;
	PZ_REQUIRE(strips != nullptr, "pz_conv2d_fwd_stats_strips: null output");
	if (algo != PZ_CONV_ALGO_DIRECT && igemm_eligible(d, P, Q) && uses_winograd(d, PZ_CONV_FWD, P, Q, algo))
		*strips = pz::wino_stats_strips(d, P, Q);      // one entry per (channel, block of 32 tiles), with explicit counts
	else
		*strips = (algo == PZ_CONV_ALGO_DIRECT || !igemm_eligible(d, P, Q)) ? 0 : pz::ceil_div((long)d->n * P * Q, PZ_CONV_STATS_STRIP);
	return PZ_OK;
}

// The forward pass's filter operand of the implicit GEMM: [wp | tab] at `base`
static FwdPlan fwd_pack_args(const pz_conv_desc *d, int P, int Q, const float *w, char *base, PackArgs *out) {
	const int Kg = d->k / d->groups, Cg = d->c / d->groups;
	FwdPlan p = plan_igemm(Kg, Cg * d->r * d->s, (long)d->n * P * Q, d->groups, Cg);
	PackArgs pa{};
	pa.w = w, pa.wp = (float *)base, pa.tab = (int2 *)(base + p.wp_bytes);
	pa.Kg = Kg, pa.Cg = Cg, pa.R = d->r, pa.S = d->s, pa.groups = d->groups, pa.mode = 0;
	pa.M = Kg, pa.mpad = p.mpad, pa.kred = p.kred, pa.kred_pad = p.kred_pad;
	pa.dil_h = d->dil_h, pa.dil_w = d->dil_w, pa.in_h = d->h, pa.in_w = d->w;
	pa.tapmajor = Cg % 16 == 0, pa.chans = Cg;
	*out = pa;
	return p;
}

static int conv2d_fwd_impl(const pz_conv_desc *d, const float *x, const float *w, const void *packed, const float *bias, float *y,
                           float *stats, int algo, void *workspace, size_t ws_bytes, pz_stream_t stream, int relu = 0,
                           const float *xbn = nullptr, int xbn_relu = 0);

// Convolutions whose gathers can apply a PRECEDING BatchNorm (+ ReLU) on the fly (pz_conv2d_fwd_xbn / pz_conv2d_bwd_filter_xbn):
// pointwise, unit stride, unpadded (a padded tap would have to read 0, not relu(b)), ungrouped, reduction channels in whole
// k-tiles, fp32 math. Forward: the 128 x 128 implicit GEMM below the two-tiles-ahead threshold (that form parks a tile's gathers
// a tile after it read their coefficients). Backward-filter: the pointwise form of wgrad_conv_kernel.
static bool xbn_eligible(const pz_conv_desc *d, int which, int P, int Q, int algo) {
	if (!(d->r == 1 && d->s == 1 && d->pad_h == 0 && d->pad_w == 0 && d->stride_h == 1 && d->stride_w == 1 && d->groups == 1 &&
	      d->c % 16 == 0 && algo != PZ_CONV_ALGO_DIRECT && igemm_eligible(d, P, Q) && pz::g_conv_math == 0))
		return false;
	if (which == PZ_CONV_FWD) {
		const FwdPlan p = plan_igemm(d->k, d->c, (long)d->n * P * Q, 1, d->c);
		return p.bm == 128 && p.kred_pad < 16 * ig_prefetch2_min_ktiles();
	}
	return which == PZ_CONV_BWD_FILTER;
}

int pz_conv2d_xbn_supported(const pz_conv_desc *d, int which, int algo, int *supported) {
	int P, Q;
	if (int rc = check_desc(d, &P, &Q)) return rc;
	PZ_REQUIRE(supported != nullptr, "pz_conv2d_xbn_supported: null output");
	*supported = xbn_eligible(d, which, P, Q, algo) ? 1 : 0;
	return PZ_OK;
}

int pz_conv2d_fwd_xbn(const pz_conv_desc *d, const float *x, const float *xcoef, int xrelu, const float *w, const void *packed,
                      const float *bias, float *y, float *stats, int algo, void *workspace, size_t ws_bytes, pz_stream_t stream) {
	int P, Q;
	if (int rc = check_desc(d, &P, &Q)) return rc;
	PZ_REQUIRE(xcoef != nullptr, "pz_conv2d_fwd_xbn: null coefficients");
	PZ_REQUIRE(xbn_eligible(d, PZ_CONV_FWD, P, Q, algo), "pz_conv2d_fwd_xbn: this convolution cannot normalise its input on the fly (pz_conv2d_xbn_supported)");
	return conv2d_fwd_impl(d, x, packed ? nullptr : w, packed, bias, y, stats, algo, workspace, ws_bytes, stream, 0, xcoef, xrelu);
}

// Which passes can apply an activation in their epilogue (pz_conv2d_fwd_relu / pz_conv2d_bwd_data_gate): implicit-GEMM
// launches whose output pixels are contiguous per image — every forward pass on that path, backward-data at unit stride.
int pz_conv2d_epilogue_supported(const pz_conv_desc *d, int which, int algo, int *supported) {
	int P, Q;
	if (int rc = check_desc(d, &P, &Q)) return rc;
	PZ_REQUIRE(supported != nullptr && (which == PZ_CONV_FWD || which == PZ_CONV_BWD_DATA), "pz_conv2d_epilogue_supported: bad arguments");
	*supported = conv_path(d, which, P, Q, algo) == PATH_IGEMM && (which == PZ_CONV_FWD || (d->stride_h == 1 && d->stride_w == 1));
	return PZ_OK;
}

int pz_conv2d_fwd_relu(const pz_conv_desc *d, const float *x, const float *w, const void *packed, const float *bias, float *y, int algo,
                       void *workspace, size_t ws_bytes, pz_stream_t stream) {
	int ok = 0;
	if (int rc = pz_conv2d_epilogue_supported(d, PZ_CONV_FWD, algo, &ok)) return rc;
	PZ_REQUIRE(ok, "pz_conv2d_fwd_relu: this configuration has no activation epilogue (pz_conv2d_epilogue_supported)");
	return conv2d_fwd_impl(d, x, packed ? nullptr : w, packed, bias, y, nullptr, algo, workspace, ws_bytes, stream, 1);
}

int pz_conv2d_fwd(const pz_conv_desc *d, const float *x, const float *w, const float *bias, float *y, int algo,
                  void *workspace, size_t ws_bytes, pz_stream_t stream) {
	return conv2d_fwd_impl(d, x, w, nullptr, bias, y, nullptr, algo, workspace, ws_bytes, stream);
}

int pz_conv2d_fwd_stats(const pz_conv_desc *d, const float *x, const float *w, const float *bias, float *y, float *stats,
                        int algo, void *workspace, size_t ws_bytes, pz_stream_t stream) {
	return conv2d_fwd_impl(d, x, w, nullptr, bias, y, stats, algo, workspace, ws_bytes, stream);
}

int pz_conv2d_fwd_pre(const pz_conv_desc *d, const float *x, const void *packed, const float *bias, float *y, float *stats,
                      int algo, void *workspace, size_t ws_bytes, pz_stream_t stream) {
	PZ_REQUIRE(packed != nullptr, "pz_conv2d_fwd_pre: no prepared filter operand");
	return conv2d_fwd_impl(d, x, nullptr, packed, bias, y, stats, algo, workspace, ws_bytes, stream);
}

static int conv2d_fwd_impl(const pz_conv_desc *d, const float *x, const float *w, const void *packed, const float *bias, float *y,
                           float *stats, int algo, void *workspace, size_t ws_bytes, pz_stream_t stream, int relu, const float *xbn,
                           int xbn_relu) {
	int P, Q;
	if (int rc = check_desc(d, &P, &Q)) return rc;
	PZ_REQUIRE(x && (w || packed) && y, "pz_conv2d_fwd: null tensor");
	hipStream_t st = pz::as_stream(stream);

	if (uses_winograd(d, PZ_CONV_FWD, P, Q, algo)) {
		PZ_REQUIRE(stats == nullptr || pz::wino_stats_strips(d, P, Q) > 0, "pz_conv2d_fwd_stats: this Winograd build does not produce statistics");
		size_t need;
		conv_workspace_bytes(d, PZ_CONV_FWD, algo, packed != nullptr, &need);
		PZ_REQUIRE(need == 0 || (workspace != nullptr && ws_bytes >= need), "pz_conv2d_fwd: workspace %zu < required %zu bytes", ws_bytes, need);
		const size_t fbytes = packed ? 0 : align256(pz::wino_workspace_bytes(d, PZ_CONV_FWD, P, Q));
		void *vscratch = pz::wino_input_bytes(d, PZ_CONV_FWD, P, Q) > 0 ? (char *)workspace + fbytes : nullptr;
		ProfScope prof(st, 3, 2.0 * d->n * P * Q * (double)d->k * d->c * d->r * d->s);
		return pz::wino_conv(d, PZ_CONV_FWD, P, Q, x, w, bias, y, packed ? const_cast<void *>(packed) : workspace, st, stats, packed != nullptr,
		                     vscratch);
	}
	PZ_REQUIRE(packed == nullptr || (algo != PZ_CONV_ALGO_DIRECT && igemm_eligible(d, P, Q)), "pz_conv2d_fwd_pre: this configuration takes no prepared operand");

	if (algo == PZ_CONV_ALGO_DIRECT || !igemm_eligible(d, P, Q)) {
		PZ_REQUIRE(stats == nullptr, "pz_conv2d_fwd_stats: this configuration cannot produce strip statistics");
		const size_t total = (size_t)d->n * d->k * P * Q;
		direct_fwd_kernel<<<pz::stream_grid(total, 256), 256, 0, st>>>(*d, P, Q, x, w, bias, y);
		PZ_LAUNCH_CHECK();
		return PZ_OK;
	}

	size_t need;
	conv_workspace_bytes(d, PZ_CONV_FWD, algo, packed != nullptr, &need);
	PZ_REQUIRE(need == 0 || (workspace != nullptr && ws_bytes >= need), "pz_conv2d_fwd: workspace %zu < required %zu bytes", ws_bytes, need);

	const int Kg = d->k / d->groups, Cg = d->c / d->groups;
	PackArgs pa{};
	FwdPlan p = fwd_pack_args(d, P, Q, w, packed ? (char *)const_cast<void *>(packed) : (char *)workspace, &pa);
	PZ_REQUIRE(packed == nullptr || !p.split, "pz_conv2d_fwd_pre: the split math modes prepare their operands per call");

	float *wp = pa.wp;
	int2 *tab = pa.tab;
	float *slabs = packed ? (float *)workspace : (float *)((char *)workspace + p.wp_bytes + p.tab_bytes);

	if (!packed) {
		const long ptotal = (long)d->groups * p.kred_pad * p.mpad;
		if (p.split)
			pack_filter_split_kernel<<<pz::stream_grid(ptotal / 8, 256), 256, 0, st>>>(pa);
		else
			pack_filter_kernel<<<pz::stream_grid(ptotal, 256), 256, 0, st>>>(pa);
		PZ_LAUNCH_CHECK();
	}

	IgemmArgs a{};
	a.x = x, a.wp = wp, a.tab = tab, a.bias = bias, a.y = y;
	a.C_total = d->c, a.H = d->h, a.W = d->w, a.Cg = Cg;
	a.M = Kg, a.mpad = p.mpad, a.kred_pad = p.kred_pad;
	a.Pv = P, a.Qv = Q, a.npix = d->n * P * Q;
	a.vs_h = d->stride_h, a.vs_w = d->stride_w, a.pad_h = d->pad_h, a.pad_w = d->pad_w;
	a.R = d->r, a.S = d->s, a.dil_h = d->dil_h, a.dil_w = d->dil_w;
	a.x_bytes = (unsigned)((size_t)d->n * d->c * d->h * d->w * 4);
	a.wp_bytes = (unsigned)((size_t)d->groups * p.kred_pad * p.mpad * (p.split ? 6 : 4));
	a.y_bytes = (unsigned)((size_t)d->n * d->k * P * Q * 4);
	a.OC_total = d->k, a.OH = P, a.OW = Q, a.os_h = 1, a.os_w = 1, a.oo_h = 0, a.oo_w = 0;
	a.tapmajor = pa.tapmajor;
	a.contig = 1, a.stats = reinterpret_cast<float4 *>(stats);
	a.relu = relu;
	a.xbn = reinterpret_cast<const float2 *>(xbn), a.xbn_relu = xbn_relu;
	// launch_igemm has the described-operand gather in ONE instantiation (128x128, four waves, plain math): any other plan
	// would launch nothing and leave y unwritten — refuse here, whatever xbn_fwd_eligible and plan_igemm say today
	PZ_REQUIRE(xbn == nullptr || (p.bm == 128 && !p.split),
			   "pz_conv2d_fwd_xbn: the plan of this layer (tile %d, split %d) has no form that reads a described operand", p.bm, p.split);
	a.stat_strips = pz::ceil_div((long)d->n * P * Q, PZ_CONV_STATS_STRIP);
	static_assert(PZ_CONV_STATS_STRIP == 64, "strip = 32 * TN pixels of both tile configurations");
	run_igemm(p, a, slabs, d->groups, st, 2.0 * d->n * P * Q * (double)d->k * Cg * d->r * d->s);
	PZ_LAUNCH_CHECK();
	return PZ_OK;
}

// ---- filter operands prepared ahead of the pass, many layers per launch ------------------------------------------------
// What a pass derives from the filter tensor alone — the implicit GEMM's packed forward operand + gather table, the
// Winograd kernels' transformed filters (forward: G g G^T; backward-data: the same of the flipped, transposed filters) —
// depends on the parameters only. A training step used to launch one ~5 us kernel per layer and pass for it, each on the
// critical path of its stream; pz_conv2d_prepack prepares any number of them in a few launches (the caller keeps them
// until the parameters change) and pz_conv2d_{fwd,bwd_data}_pre consume them.
int pz_conv2d_prepack_bytes(const pz_conv_desc *d, int which, int algo, size_t *nbytes) {
	int P, Q;
	if (int rc = check_desc(d, &P, &Q)) return rc;
	PZ_REQUIRE(nbytes != nullptr && (which == PZ_CONV_FWD || which == PZ_CONV_BWD_DATA), "pz_conv2d_prepack_bytes: bad arguments");
	*nbytes = 0;
	if (uses_winograd(d, which, P, Q, algo)) {
		*nbytes = align256(pz::wino_workspace_bytes(d, which, P, Q));
		return PZ_OK;
	}
	if (which != PZ_CONV_FWD || algo == PZ_CONV_ALGO_DIRECT || !igemm_eligible(d, P, Q)) return PZ_OK;
	PackArgs pa;
	FwdPlan p = fwd_pack_args(d, P, Q, nullptr, nullptr, &pa);
	if (!p.split) *nbytes = p.wp_bytes + p.tab_bytes;
	return PZ_OK;
}

int pz_conv2d_prepack(const pz_prepack_job *jobs, int njobs, pz_stream_t stream) {
	PZ_REQUIRE(jobs != nullptr || njobs == 0, "pz_conv2d_prepack: null job list");
	hipStream_t st = pz::as_stream(stream);
	PackBatch batch{};
	auto flush = [&]() -> int {
		if (batch.n == 0) return PZ_OK;
		pack_filter_batch_kernel<<<batch.start[batch.n], 256, 0, st>>>(batch);
		PZ_LAUNCH_CHECK();
		batch.n = 0;
		return PZ_OK;
	};
	const pz_conv_desc *wd[pz::kWinoBatch];
	int wwhich[pz::kWinoBatch], nw = 0;
	const float *ww[pz::kWinoBatch];
	float *wu[pz::kWinoBatch];
	auto flush_wino = [&]() -> int {
		if (nw == 0) return PZ_OK;
		int rc = pz::wino_filter_batch(wd, wwhich, ww, wu, nw, st);
		nw = 0;
		return rc;
	};

	for (int i = 0; i < njobs; ++i) {
		const pz_prepack_job &job = jobs[i];
		int P, Q;
		if (int rc = check_desc(&job.desc, &P, &Q)) return rc;
		size_t nbytes;
		if (int rc = pz_conv2d_prepack_bytes(&job.desc, job.which, job.algo, &nbytes)) return rc;
		PZ_REQUIRE(nbytes > 0 && job.w && job.packed, "pz_conv2d_prepack: job %d has nothing to prepare (or null pointers)", i);
		if (uses_winograd(&job.desc, job.which, P, Q, job.algo)) {
			wd[nw] = &job.desc, wwhich[nw] = job.which, ww[nw] = job.w, wu[nw] = (float *)job.packed;
			if (++nw == pz::kWinoBatch)
				if (int rc = flush_wino()) return rc;
			continue;
		}
		PackArgs pa;
		FwdPlan p = fwd_pack_args(&job.desc, P, Q, job.w, (char *)job.packed, &pa);
		const long ptotal = (long)job.desc.groups * p.kred_pad * p.mpad;
		batch.job[batch.n] = pa;
		batch.start[batch.n + 1] = batch.start[batch.n] + pz::stream_grid(ptotal, 1024);      // 4 elements per thread
		if (++batch.n == kPackBatch)
			if (int rc = flush()) return rc;
	}
	if (int rc = flush()) return rc;
	return flush_wino();
}

// convolutions whose gathers can apply a following BatchNorm's backward on the fly (pz_conv2d_bwd_*_bn): pointwise
// filter without padding (no padded tap would turn the affine constant into a contribution), ungrouped, MFMA path,
// reduction channels in whole k-tiles
bool bn_fold_eligible(const pz_conv_desc *d, int P, int Q, int algo) {
	return d->r == 1 && d->s == 1 && d->pad_h == 0 && d->pad_w == 0 && d->groups == 1 && d->k % 16 == 0 &&
	       algo != PZ_CONV_ALGO_DIRECT && igemm_eligible(d, P, Q) && dgrad_uses_igemm(d);
}

int pz_conv2d_bn_fold_supported(const pz_conv_desc *d, int algo, int *supported) {
	int P, Q;
	if (int rc = check_desc(d, &P, &Q)) return rc;
	PZ_REQUIRE(supported != nullptr, "pz_conv2d_bn_fold_supported: null output");
	*supported = bn_fold_eligible(d, P, Q, algo) ? 1 : 0;
	return PZ_OK;
}

struct DgradBnStats {       // pz_conv2d_bwd_data_bnstats: the BatchNorm in front of the layer (IgemmArgs::gx ...) and where its sums go
	const float *gx, *gab, *gmean;
	float *partials;
};

static int conv2d_bwd_data_impl(const pz_conv_desc *d, const float *dy, const float *bnx, const float *bncoef, const float *w,
                                float *dx, int algo, void *workspace, size_t ws_bytes, pz_stream_t stream, const void *packed = nullptr,
                                const float *gate = nullptr, const DgradBnStats *bst = nullptr);

// strip sums of a backward-data epilogue -> the merged pair per channel, in fp64 and a fixed order, where the BatchNorm
// backward kernels read them (bn.hip: bn_merge — 2 c doubles at the head of the partials buffer)
__global__ void __launch_bounds__(256) dgrad_bnstats_merge_kernel(const float2 *__restrict__ gst, int strips, double *__restrict__ merged) {
	__shared__ double ra[256], rb[256];
	const int ch = blockIdx.x, tid = threadIdx.x;
	double sa = 0.0, sb = 0.0;
	for (int i = tid; i < strips; i += 256) {
		const float2 v = gst[(size_t)ch * strips + i];
		sa += (double)v.x, sb += (double)v.y;
	}
	ra[tid] = sa, rb[tid] = sb;
	__syncthreads();
	for (int h = 128; h > 0; h >>= 1) {
		if (tid < h) ra[tid] += ra[tid + h], rb[tid] += rb[tid + h];
		__syncthreads();
	}
	if (tid == 0) merged[2 * ch] = ra[0], merged[2 * ch + 1] = rb[0];
}

static size_t dgrad_bnstats_head_bytes(int c) { return align256((size_t)c * 2 * sizeof(double)); }

// strips per channel the backward-data launch of this layer leaves: 64-pixel strips of the implicit GEMM's contiguous-output
// epilogue, blocks of 32 output tiles of the F(4x4) Winograd kernel; 0 = the launch cannot sum them
static long dgrad_bnstats_strips(const pz_conv_desc *d, int P, int Q, int algo) {
	const ConvPath path = conv_path(d, PZ_CONV_BWD_DATA, P, Q, algo);
	if (path == PATH_WINOGRAD) return pz::wino_bnstats_strips(d, P, Q);
	if (path == PATH_IGEMM && d->stride_h == 1 && d->stride_w == 1) return pz::ceil_div((long)d->n * d->h * d->w, PZ_CONV_STATS_STRIP);
	return 0;
}

int pz_conv2d_bwd_data_bnstats_bytes(const pz_conv_desc *d, int algo, size_t *nbytes) {
	int P, Q;
	if (int rc = check_desc(d, &P, &Q)) return rc;
	PZ_REQUIRE(nbytes != nullptr, "pz_conv2d_bwd_data_bnstats_bytes: null output");
	const long strips = dgrad_bnstats_strips(d, P, Q, algo);
	// 0: this configuration has no such epilogue (F(2x2) Winograd / strided / direct forms)
	*nbytes = strips > 0 ? dgrad_bnstats_head_bytes(d->c) + (size_t)d->c * strips * sizeof(float2) : 0;
	return PZ_OK;
}

int pz_conv2d_bwd_data_bnstats(const pz_conv_desc *d, const float *dy, const float *bnx, const float *bncoef, const float *w,
                               float *dx, const float *gx, const float *gab, const float *gmean, float *partials, int algo,
                               void *workspace, size_t ws_bytes, pz_stream_t stream) {
	int P, Q;
	if (int rc = check_desc(d, &P, &Q)) return rc;
	PZ_REQUIRE(dgrad_bnstats_strips(d, P, Q, algo) > 0,
	           "pz_conv2d_bwd_data_bnstats: this configuration has no statistics epilogue (pz_conv2d_bwd_data_bnstats_bytes)");
	PZ_REQUIRE(gx && gab && gmean && partials, "pz_conv2d_bwd_data_bnstats: null BatchNorm operand");
	PZ_REQUIRE((bnx == nullptr) == (bncoef == nullptr), "pz_conv2d_bwd_data_bnstats: the gradient-side fold needs both its operands");
	PZ_REQUIRE(bnx == nullptr || bn_fold_eligible(d, P, Q, algo), "pz_conv2d_bwd_data_bnstats: this convolution cannot fold a BatchNorm backward");
	const DgradBnStats bst{gx, gab, gmean, partials};
	return conv2d_bwd_data_impl(d, dy, bnx, bncoef, w, dx, algo, workspace, ws_bytes, stream, nullptr, nullptr, &bst);
}

int pz_conv2d_bwd_data_gate(const pz_conv_desc *d, const float *dy, const float *w, const float *gate, float *dx, int algo,
                            void *workspace, size_t ws_bytes, pz_stream_t stream) {
	int ok = 0;
	if (int rc = pz_conv2d_epilogue_supported(d, PZ_CONV_BWD_DATA, algo, &ok)) return rc;
	PZ_REQUIRE(ok && gate, "pz_conv2d_bwd_data_gate: this configuration has no gated epilogue (pz_conv2d_epilogue_supported)");
	return conv2d_bwd_data_impl(d, dy, nullptr, nullptr, w, dx, algo, workspace, ws_bytes, stream, nullptr, gate);
}

int pz_conv2d_bwd_data_pre(const pz_conv_desc *d, const float *dy, const void *packed, float *dx, int algo, void *workspace,
                           size_t ws_bytes, pz_stream_t stream) {
	PZ_REQUIRE(packed != nullptr, "pz_conv2d_bwd_data_pre: no prepared filter operand");
	return conv2d_bwd_data_impl(d, dy, nullptr, nullptr, nullptr, dx, algo, workspace, ws_bytes, stream, packed);
}

int pz_conv2d_bwd_data(const pz_conv_desc *d, const float *dy, const float *w, float *dx, int algo, void *workspace,
                       size_t ws_bytes, pz_stream_t stream) {
	return conv2d_bwd_data_impl(d, dy, nullptr, nullptr, w, dx, algo, workspace, ws_bytes, stream);
}

int pz_conv2d_bwd_data_bn(const pz_conv_desc *d, const float *dy, const float *bnx, const float *bncoef, const float *w,
                          float *dx, int algo, void *workspace, size_t ws_bytes, pz_stream_t stream) {
	int P, Q;
	if (int rc = check_desc(d, &P, &Q)) return rc;
	PZ_REQUIRE(bnx && bncoef, "pz_conv2d_bwd_data_bn: null BatchNorm operand");
	PZ_REQUIRE(bn_fold_eligible(d, P, Q, algo), "pz_conv2d_bwd_data_bn: this convolution cannot fold a BatchNorm backward");
	return conv2d_bwd_data_impl(d, dy, bnx, bncoef, w, dx, algo, workspace, ws_bytes, stream);
}

static int conv2d_bwd_data_impl(const pz_conv_desc *d, const float *dy, const float *bnx, const float *bncoef, const float *w,
                                float *dx, int algo, void *workspace, size_t ws_bytes, pz_stream_t stream, const void *packed,
                                const float *gate, const DgradBnStats *bst) {
	int P, Q;
	if (int rc = check_desc(d, &P, &Q)) return rc;
	PZ_REQUIRE(dy && (w || packed) && dx, "pz_conv2d_bwd_data: null tensor");
	hipStream_t st = pz::as_stream(stream);

	if (bnx == nullptr && uses_winograd(d, PZ_CONV_BWD_DATA, P, Q, algo)) {
		size_t need;
		conv_workspace_bytes(d, PZ_CONV_BWD_DATA, algo, packed != nullptr, &need);
		PZ_REQUIRE(need == 0 || (workspace != nullptr && ws_bytes >= need), "pz_conv2d_bwd_data: workspace %zu < required %zu bytes", ws_bytes, need);
		const size_t fbytes = packed ? 0 : align256(pz::wino_workspace_bytes(d, PZ_CONV_BWD_DATA, P, Q));
		void *vscratch = pz::wino_input_bytes(d, PZ_CONV_BWD_DATA, P, Q) > 0 ? (char *)workspace + fbytes : nullptr;
		if (bst) {
			const int strips = pz::wino_bnstats_strips(d, P, Q);
			PZ_REQUIRE(strips > 0, "pz_conv2d_bwd_data_bnstats: this Winograd form has no statistics epilogue");
			const pz::BnStatsOut out{bst->gx, bst->gab, bst->gmean, (float *)((char *)bst->partials + dgrad_bnstats_head_bytes(d->c))};
			{
				ProfScope prof(st, 3, 2.0 * d->n * P * Q * (double)d->k * d->c * d->r * d->s);
				if (int rc = pz::wino_conv(d, PZ_CONV_BWD_DATA, P, Q, dy, w, nullptr, dx, packed ? const_cast<void *>(packed) : workspace, st,
				                           nullptr, packed != nullptr, vscratch, &out))
					return rc;
			}
			dgrad_bnstats_merge_kernel<<<d->c, 256, 0, st>>>(reinterpret_cast<const float2 *>(out.gst), strips, reinterpret_cast<double *>(bst->partials));
			PZ_LAUNCH_CHECK();
			return PZ_OK;
		}
		ProfScope prof(st, 3, 2.0 * d->n * P * Q * (double)d->k * d->c * d->r * d->s);
		return pz::wino_conv(d, PZ_CONV_BWD_DATA, P, Q, dy, w, nullptr, dx, packed ? const_cast<void *>(packed) : workspace, st, nullptr,
		                     packed != nullptr, vscratch);
	}
	// (only the Winograd form of backward-data takes a prepared operand: the pointwise layers read the filter tensor itself)
	PZ_REQUIRE(packed == nullptr, "pz_conv2d_bwd_data_pre: this configuration takes no prepared operand");

	// a handful of input maps behind a stride-2 filter (the network's first layer): the dedicated vector-ALU kernel
	if (algo == PZ_CONV_ALGO_AUTO && bnx == nullptr && pz::thin_dgrad_eligible(d, P, Q)) {
		const size_t need = pz::thin_dgrad_workspace_bytes(d);
		PZ_REQUIRE(workspace != nullptr && ws_bytes >= need, "pz_conv2d_bwd_data: workspace %zu < required %zu bytes", ws_bytes, need);
		return pz::thin_dgrad(d, P, Q, dy, w, dx, workspace, st);
	}

	if (algo == PZ_CONV_ALGO_DIRECT || !dgrad_uses_igemm(d) || !igemm_eligible(d, P, Q)) {
		const size_t total = (size_t)d->n * d->c * d->h * d->w;
		direct_bwd_data_kernel<<<pz::stream_grid(total, 256), 256, 0, st>>>(*d, P, Q, dy, w, dx);
		PZ_LAUNCH_CHECK();
		return PZ_OK;
	}

	size_t need;
	pz_conv2d_workspace_bytes(d, PZ_CONV_BWD_DATA, algo, &need);
	PZ_REQUIRE(workspace != nullptr && ws_bytes >= need, "pz_conv2d_bwd_data: workspace %zu < required %zu bytes", ws_bytes, need);

	const int Kg = d->k / d->groups, Cg = d->c / d->groups;
	DgradClass cls[16];
	bool needs_zero;
	const int nc = dgrad_classes(d, cls, &needs_zero);

	if (needs_zero) PZ_HIP(hipMemsetAsync(dx, 0, (size_t)d->n * d->c * d->h * d->w * sizeof(float), st));

	// algorithmic work of backward-data = that of forward; shared out over the class launches by GEMM size
	const double flops_total = 2.0 * d->n * P * Q * (double)d->k * Cg * d->r * d->s;
	double gemm_total = 0.0;
	for (int i = 0; i < nc; ++i) gemm_total += (double)cls[i].Pv * cls[i].Qv * cls[i].Rc * cls[i].Sc;

	char *wsp = (char *)workspace;
	for (int i = 0; i < nc; ++i) {
		const DgradClass &c = cls[i];
		FwdPlan p = plan_igemm(Cg, Kg * c.Rc * c.Sc, (long)d->n * c.Pv * c.Qv, d->groups, Kg);

		float *wp = (float *)wsp;
		wsp += p.wp_bytes;
		int2 *tab = (int2 *)wsp;
		wsp += p.tab_bytes;
		float *slabs = (float *)wsp;
		wsp += p.slab_bytes;

		PackArgs pa{};
		pa.w = w, pa.wp = wp, pa.tab = tab;
		pa.Kg = Kg, pa.Cg = Cg, pa.R = d->r, pa.S = d->s, pa.groups = d->groups, pa.mode = 1;
		pa.M = Cg, pa.mpad = p.mpad, pa.kred = p.kred, pa.kred_pad = p.kred_pad;
		pa.a_h = c.a_h, pa.a_w = c.a_w, pa.st_h = d->stride_h, pa.st_w = d->stride_w, pa.Rc = c.Rc, pa.Sc = c.Sc;
		pa.dil_h = d->dil_h, pa.dil_w = d->dil_w, pa.in_h = P, pa.in_w = Q;
		pa.tapmajor = Kg % 16 == 0, pa.chans = Kg;
		// A pointwise filter (K, C, 1, 1) already IS the packed operand [kred = k][m = c] of its backward-data GEMM when
		// neither axis needs padding: the kernel reads the filter tensor itself and derives the k-table (no pack launch)
		const bool as_is = d->r == 1 && d->s == 1 && d->groups == 1 && pa.tapmajor && !p.split && p.mpad == Cg && p.kred_pad == p.kred;
		const long ptotal = (long)d->groups * p.kred_pad * p.mpad;
		if (as_is)
			wp = const_cast<float *>(w), tab = nullptr;
		else if (p.split)
			pack_filter_split_kernel<<<pz::stream_grid(ptotal / 8, 256), 256, 0, st>>>(pa);
		else
			pack_filter_kernel<<<pz::stream_grid(ptotal, 256), 256, 0, st>>>(pa);
		PZ_LAUNCH_CHECK();

		IgemmArgs a{};
		a.x = dy, a.wp = wp, a.tab = tab, a.bias = nullptr, a.y = dx;
		a.C_total = d->k, a.H = P, a.W = Q, a.Cg = Kg;
		a.M = Cg, a.mpad = p.mpad, a.kred_pad = p.kred_pad;
		a.Pv = c.Pv, a.Qv = c.Qv, a.npix = d->n * c.Pv * c.Qv;
		a.vs_h = 1, a.vs_w = 1, a.pad_h = c.pad_h, a.pad_w = c.pad_w;
		a.R = c.Rc, a.S = c.Sc, a.dil_h = d->dil_h, a.dil_w = d->dil_w;
		a.x_bytes = (unsigned)((size_t)d->n * d->k * P * Q * 4);
		a.wp_bytes = (unsigned)((size_t)d->groups * p.kred_pad * p.mpad * (p.split ? 6 : 4));
		a.y_bytes = (unsigned)((size_t)d->n * d->c * d->h * d->w * 4);
		a.OC_total = d->c, a.OH = d->h, a.OW = d->w;
		a.os_h = d->stride_h, a.os_w = d->stride_w, a.oo_h = c.oo_h, a.oo_w = c.oo_w;
		a.tapmajor = pa.tapmajor;
		a.contig = d->stride_h == 1 && d->stride_w == 1 && c.Pv == d->h && c.Qv == d->w && c.oo_h == 0 && c.oo_w == 0;
		a.stats = nullptr;
		a.gate = gate;
		PZ_REQUIRE(gate == nullptr || a.contig, "pz_conv2d_bwd_data_gate: output pixels are not contiguous");
		a.x2 = bnx, a.xcoef = reinterpret_cast<const float4 *>(bncoef);
		if (bst) {
			PZ_REQUIRE(a.contig && nc == 1, "pz_conv2d_bwd_data_bnstats: output pixels are not contiguous");
			a.gx = bst->gx, a.gab = reinterpret_cast<const float2 *>(bst->gab), a.gmean = bst->gmean;
			a.gst = reinterpret_cast<float2 *>((char *)bst->partials + dgrad_bnstats_head_bytes(d->c));
			a.stat_strips = pz::ceil_div((long)a.npix, PZ_CONV_STATS_STRIP);
		}
		run_igemm(p, a, slabs, d->groups, st, flops_total * ((double)c.Pv * c.Qv * c.Rc * c.Sc) / gemm_total);
		PZ_LAUNCH_CHECK();
		if (bst) {
			dgrad_bnstats_merge_kernel<<<d->c, 256, 0, st>>>(a.gst, a.stat_strips, reinterpret_cast<double *>(bst->partials));
			PZ_LAUNCH_CHECK();
		}
	}
	PZ_REQUIRE(bst == nullptr || nc == 1, "pz_conv2d_bwd_data_bnstats: no implicit-GEMM launch took the statistics");
	return PZ_OK;
}

static int conv2d_bwd_filter_impl(const pz_conv_desc *d, const float *x, const float *dy, const float *bnx, const float *bncoef,
                                  float *dw, float *db, float alpha, float beta, int algo, void *workspace, size_t ws_bytes,
                                  pz_stream_t stream, const float *xbn = nullptr, int xbn_relu = 0);

int pz_conv2d_bwd_filter_xbn(const pz_conv_desc *d, const float *x, const float *xcoef, int xrelu, const float *dy, const float *bnx,
                             const float *bncoef, float *dw, float alpha, float beta, int algo, void *workspace, size_t ws_bytes,
                             pz_stream_t stream) {
	int P, Q;
	if (int rc = check_desc(d, &P, &Q)) return rc;
	PZ_REQUIRE(xcoef != nullptr, "pz_conv2d_bwd_filter_xbn: null coefficients");
	PZ_REQUIRE((bnx == nullptr) == (bncoef == nullptr), "pz_conv2d_bwd_filter_xbn: the gradient-side fold needs both its operands");
	PZ_REQUIRE(xbn_eligible(d, PZ_CONV_BWD_FILTER, P, Q, algo) && (bnx == nullptr || bn_fold_eligible(d, P, Q, algo)),
	           "pz_conv2d_bwd_filter_xbn: this convolution cannot normalise its input on the fly (pz_conv2d_xbn_supported)");
	return conv2d_bwd_filter_impl(d, x, dy, bnx, bncoef, dw, nullptr, alpha, beta, algo, workspace, ws_bytes, stream, xcoef, xrelu);
}

int pz_conv2d_bwd_filter(const pz_conv_desc *d, const float *x, const float *dy, float *dw, float *db, float alpha,
                         float beta, int algo, void *workspace, size_t ws_bytes, pz_stream_t stream) {
	return conv2d_bwd_filter_impl(d, x, dy, nullptr, nullptr, dw, db, alpha, beta, algo, workspace, ws_bytes, stream);
}

int pz_conv2d_bwd_filter_bn(const pz_conv_desc *d, const float *x, const float *dy, const float *bnx, const float *bncoef,
                            float *dw, float alpha, float beta, int algo, void *workspace, size_t ws_bytes, pz_stream_t stream) {
	int P, Q;
	if (int rc = check_desc(d, &P, &Q)) return rc;
	PZ_REQUIRE(bnx && bncoef, "pz_conv2d_bwd_filter_bn: null BatchNorm operand");
	PZ_REQUIRE(bn_fold_eligible(d, P, Q, algo), "pz_conv2d_bwd_filter_bn: this convolution cannot fold a BatchNorm backward");
	return conv2d_bwd_filter_impl(d, x, dy, bnx, bncoef, dw, nullptr, alpha, beta, algo, workspace, ws_bytes, stream);
}

static int conv2d_bwd_filter_impl(const pz_conv_desc *d, const float *x, const float *dy, const float *bnx, const float *bncoef,
                                  float *dw, float *db, float alpha, float beta, int algo, void *workspace, size_t ws_bytes,
                                  pz_stream_t stream, const float *xbn, int xbn_relu) {
	int P, Q;
	if (int rc = check_desc(d, &P, &Q)) return rc;
	PZ_REQUIRE(x && dy && dw, "pz_conv2d_bwd_filter: null tensor");
	hipStream_t st = pz::as_stream(stream);

	const int Kg = d->k / d->groups, Cg = d->c / d->groups;
	const bool gemm_path = !(algo == PZ_CONV_ALGO_DIRECT || !igemm_eligible(d, P, Q));

	// The bias gradient rides in the filter-gradient kernel (wgrad_conv_kernel<..., BIAS>) when that kernel runs on fp32
	// operands and the workspace has the partials' block (PUZZLE_MI355_BIAS_FOLD=0: the two-stage kernels, A/B aid)
	size_t need_all = 0;
	if (db && gemm_path) pz_conv2d_workspace_bytes(d, PZ_CONV_BWD_FILTER, algo, &need_all);
	static const bool fold_allowed = [] { const char *e = getenv("PUZZLE_MI355_BIAS_FOLD"); return !e || atoi(e) != 0; }();
	const bool fold_bias = db && gemm_path && fold_allowed && bnx == nullptr && !uses_winograd(d, PZ_CONV_BWD_FILTER, P, Q, algo) &&
	                       !wgrad_split_eligible(d) && workspace && ws_bytes >= need_all;

	if (db && !fold_bias) {
		const int S = bias_grad_splits(d->n, d->k);
		const size_t part_bytes = align256((size_t)d->k * S * sizeof(float));

		if (gemm_path && workspace && ws_bytes >= need_all && S > 1) {
			float *part = (float *)((char *)workspace + need_all - part_bytes);      // last block of the workspace
			bias_grad_partial_kernel<<<dim3(d->k, S), 256, 0, st>>>(dy, part, d->n, d->k, P * Q, S);
			PZ_LAUNCH_CHECK();
			bias_grad_finish_kernel<<<pz::ceil_div(d->k, 256), 256, 0, st>>>(part, db, d->k, S, alpha, beta);
		} else {
			bias_grad_kernel<<<d->k, 256, 0, st>>>(dy, db, d->n, d->k, P * Q, alpha, beta);
		}
		PZ_LAUNCH_CHECK();
	}

	if (algo == PZ_CONV_ALGO_DIRECT || !igemm_eligible(d, P, Q)) {
		direct_bwd_filter_kernel<<<d->k * Cg * d->r * d->s, 256, 0, st>>>(*d, P, Q, x, dy, dw, alpha, beta);
		PZ_LAUNCH_CHECK();
		return PZ_OK;
	}

	if (bnx == nullptr && uses_winograd(d, PZ_CONV_BWD_FILTER, P, Q, algo)) {
		const size_t need = pz::wino_wgrad_workspace_bytes(d, P, Q);
		PZ_REQUIRE(workspace != nullptr && ws_bytes >= need, "pz_conv2d_bwd_filter: workspace %zu < required %zu bytes", ws_bytes, need);
		ProfScope prof(st, 3, 2.0 * d->n * P * Q * (double)d->k * d->c * 9);
		return pz::wino_wgrad(d, P, Q, x, dy, dw, alpha, beta, workspace, st);
	}

	const bool split_math = wgrad_split_eligible(d);
	WgradPlan p = split_math ? plan_wgrad_split(d, P, Q) : plan_wgrad(d, P, Q);
	const size_t need = p.tab_bytes + (p.splits > 1 ? align256(p.slab_elems * p.splits * sizeof(float)) : 0);   // (+ bias partials, optional)
	PZ_REQUIRE((workspace != nullptr || need == 0) && ws_bytes >= need, "pz_conv2d_bwd_filter: workspace %zu < required %zu bytes", ws_bytes, need);

	int2 *tab = (int2 *)workspace;
	float *slabs = (float *)((char *)workspace + p.tab_bytes);

	// the plane as the kernel walks it (wgrad_plane: another rectangle of the same area for pointwise layers)
	pz_conv_desc dd = *d;
	const int P0 = P, Q0 = Q;
	if (!split_math) {
		wgrad_plane(d, P0, Q0, &P, &Q);
		if (P != P0 || Q != Q0) dd.h = P, dd.w = Q;
	}
	d = &dd;

	if (!split_math) {
		// the gather table depends on the geometry alone: built once per (device, geometry) into memory the library keeps,
		// not once per call into the workspace (37 launches of ~4 us per ResNet-50 step in front of the main kernels)
		tab = cached_wgrad_tab(p.ncrs, p.ncrs_pad, d->r, d->s, d->dil_h, d->dil_w, d->h, d->w, st);
		PZ_REQUIRE(tab != nullptr, "pz_conv2d_bwd_filter: gather table allocation failed");
	}

	WgradArgs a{};
	a.x = x, a.dy = dy, a.tab = tab;
	a.C_total = d->c, a.H = d->h, a.W = d->w, a.Cg = Cg;
	a.K_total = d->k, a.P = P, a.Q = Q, a.Kg = Kg;
	a.ncrs = p.ncrs;
	a.st_h = d->stride_h, a.st_w = d->stride_w, a.pad_h = d->pad_h, a.pad_w = d->pad_w;
	a.R = d->r, a.S = d->s, a.dil_h = d->dil_h, a.dil_w = d->dil_w;
	a.x_bytes = (unsigned)((size_t)d->n * d->c * d->h * d->w * 4);
	a.dy_bytes = (unsigned)((size_t)d->n * d->k * P * Q * 4);
	a.npix = d->n * P * ((Q + 3) / 4), a.steps_total = p.steps_total, a.steps_per_split = p.steps_per_split;
	a.Q4 = (unsigned)((Q + 3) / 4);
	a.magic_q4 = (unsigned)((((unsigned long long)1 << 32) + a.Q4 - 1) / a.Q4);
	a.magic_p = (unsigned)((((unsigned long long)1 << 32) + P - 1) / P);
	a.tiles_m = p.tiles_m, a.tiles_n = p.tiles_n;
	a.alpha = alpha, a.beta = beta;
	a.direct = p.splits == 1;
	a.out = a.direct ? dw : slabs;
	a.slab = p.slab_elems;
	a.bnx = bnx, a.bncoef = reinterpret_cast<const float4 *>(bncoef);
	a.xbn = reinterpret_cast<const float2 *>(xbn), a.xbn_relu = xbn_relu;
	float *bpart = fold_bias ? (float *)((char *)workspace + need_all - bias_part_bytes(d, p.splits)) : nullptr;
	a.db_out = fold_bias ? (a.direct ? db : bpart) : nullptr;

	dim3 grid(p.tiles_m * p.tiles_n * p.splits, 1, d->groups);
	if (split_math) {
		// cells of 8 pixels of an image plane instead of runs of 4 pixels of a row
		const unsigned cpp = (unsigned)pz::ceil_div(P * Q, 8);
		a.npix = (int)(d->n * cpp), a.Q4 = cpp;
		a.magic_q4 = (unsigned)((((unsigned long long)1 << 32) + cpp - 1) / cpp);
		ProfScope prof(st, 2, 2.0 * d->n * P * Q * (double)d->k * Cg);
#define PZ_WGRAD_SPLIT(BM_, BN_, WM_, WN_) \
		(pz::g_conv_math == 6 ? (a.bnx ? wgrad_split_kernel<BM_, BN_, WM_, WN_, true, 6><<<grid, 256, 0, st>>>(a)   \
		                               : wgrad_split_kernel<BM_, BN_, WM_, WN_, false, 6><<<grid, 256, 0, st>>>(a)) \
		                      : (a.bnx ? wgrad_split_kernel<BM_, BN_, WM_, WN_, true, 9><<<grid, 256, 0, st>>>(a)   \
		                               : wgrad_split_kernel<BM_, BN_, WM_, WN_, false, 9><<<grid, 256, 0, st>>>(a)))
		PZ_WGRAD_SPLIT(128, 128, 2, 2);          // wgrad_split_eligible: both sides fill 128-row tiles
#undef PZ_WGRAD_SPLIT
	} else {
	ProfScope prof(st, 2, 2.0 * d->n * P * Q * (double)d->k * Cg * d->r * d->s);
	const bool unit_w = d->stride_w == 1;
	const bool pointwise = d->r == 1 && d->s == 1 && d->pad_h == 0 && d->pad_w == 0 && d->stride_h == 1 && unit_w;
#define PZ_WGRAD_LAUNCH(BM_, BN_, WM_, WN_) \
	(a.xbn     ? (a.bnx ? wgrad_conv_kernel<BM_, BN_, WM_, WN_, 2, PZ_WG_RUNS, true, false, true><<<grid, 64 * WM_ * WN_, 0, st>>>(a) \
	                    : wgrad_conv_kernel<BM_, BN_, WM_, WN_, 2, PZ_WG_RUNS, false, false, true><<<grid, 64 * WM_ * WN_, 0, st>>>(a)) \
	 : a.bnx   ? (pointwise ? wgrad_conv_kernel<BM_, BN_, WM_, WN_, 2, PZ_WG_RUNS, true><<<grid, 64 * WM_ * WN_, 0, st>>>(a) \
	                        : wgrad_conv_kernel<BM_, BN_, WM_, WN_, 0, PZ_WG_RUNS, true><<<grid, 64 * WM_ * WN_, 0, st>>>(a)) \
	 : a.db_out ? (pointwise ? wgrad_conv_kernel<BM_, BN_, WM_, WN_, 2, PZ_WG_RUNS, false, true><<<grid, 64 * WM_ * WN_, 0, st>>>(a) \
	               : unit_w  ? wgrad_conv_kernel<BM_, BN_, WM_, WN_, 1, PZ_WG_RUNS, false, true><<<grid, 64 * WM_ * WN_, 0, st>>>(a) \
	                         : wgrad_conv_kernel<BM_, BN_, WM_, WN_, 0, PZ_WG_RUNS, false, true><<<grid, 64 * WM_ * WN_, 0, st>>>(a)) \
	 : pointwise ? wgrad_conv_kernel<BM_, BN_, WM_, WN_, 2, PZ_WG_RUNS><<<grid, 64 * WM_ * WN_, 0, st>>>(a) \
	 : unit_w  ? wgrad_conv_kernel<BM_, BN_, WM_, WN_, 1, PZ_WG_RUNS><<<grid, 64 * WM_ * WN_, 0, st>>>(a) \
	           : wgrad_conv_kernel<BM_, BN_, WM_, WN_, 0, PZ_WG_RUNS><<<grid, 64 * WM_ * WN_, 0, st>>>(a))
#if PZ_WG_WAVES == 8
	if (p.bm == 128 && p.bn == 128) PZ_WGRAD_LAUNCH(128, 128, 2, 4);
	else if (p.bm == 128 && p.bn == 64) PZ_WGRAD_LAUNCH(128, 64, 4, 2);
	else if (p.bm == 64 && p.bn == 128) PZ_WGRAD_LAUNCH(64, 128, 2, 4);
#else
	if (p.bm == 128 && p.bn == 128) PZ_WGRAD_LAUNCH(128, 128, 2, 2);
	else if (p.bm == 128 && p.bn == 64) PZ_WGRAD_LAUNCH(128, 64, 2, 2);
	else if (p.bm == 64 && p.bn == 128) PZ_WGRAD_LAUNCH(64, 128, 2, 2);
	else if (p.bm == 192) {                    // plan_wgrad: unit-stride gathers under a filter with taps, 16-pixel k-steps
		if (a.db_out) wgrad_conv_kernel<192, 128, 2, 2, 1, 4, false, true><<<grid, 256, 0, st>>>(a);
		else wgrad_conv_kernel<192, 128, 2, 2, 1, 4><<<grid, 256, 0, st>>>(a);
	} else if (p.bm == 64 && p.bn == 192) {      // plan_wgrad: strided gathers only
		if (a.db_out) wgrad_conv_kernel<64, 192, 2, 2, 0, PZ_WG_RUNS, false, true><<<grid, 256, 0, st>>>(a);
		else wgrad_conv_kernel<64, 192, 2, 2, 0, PZ_WG_RUNS><<<grid, 256, 0, st>>>(a);
	}
#endif
	else PZ_WGRAD_LAUNCH(64, 64, 2, 2);
#undef PZ_WGRAD_LAUNCH
	}
	PZ_LAUNCH_CHECK();

	if (!a.direct) {
		launch_wgrad_reduce(dw, slabs, p.slab_elems, p.splits, alpha, beta, st, bpart, fold_bias ? db : nullptr, d->k);
		PZ_LAUNCH_CHECK();
	}
	return PZ_OK;
}

}  // extern "C"
