/*
 * puzzle_mi355.h — C ABI of libpuzzle_mi355.so, the MI355X (gfx950 / CDNA4) operator backend that sits
 * behind PuzzleLib's Backend/{gpuarray,Blas,Dnn,Kernels}.py dispatch surface.
 *
 * Conventions
 *   - every entry returns an int status: 0 = OK, non-zero = failure; pz_last_error() returns a
 *     thread-local, human readable message for the last failure on the calling thread;
 *   - all tensors are fp32 (labels/indices int32, RNG words uint32), dense, C-contiguous, NCHW / KCRS;
 *   - pointers are raw device pointers unless named h_*; no entry retains a pointer after it returns;
 *   - every compute entry takes a stream (NULL = the device's default stream) and is asynchronous
 *     with respect to the host; nothing here synchronises unless its name says so;
 *   - no torch / Python types appear in any signature.
 *
 * Each group cites the reference interface it replaces (paths relative to puzzlelib/PuzzleLib).
 *
 * Stability. Entries declared plainly are the STABLE boundary: one per reference interface (SURVEY.md section 8b's list — device /
 * memory / streams / events, pz_conv2d_{fwd,bwd_data,bwd_filter} + workspace and shape queries, pz_gemm, pz_bn_{fwd_train,fwd_infer,
 * bwd}, pz_pool2d_*, pz_softmax_*, pz_cross_entropy, the reductions, pz_eltwise, casts, pz_rng_*, pz_comm_*, and the operators beside
 * the hot path). Entries marked PZ_FUSED are PRIVATE to this build's own Python shim (puzzlelib_amd/dnn.py, fusion.py, kernels.py):
 * fused or partial forms of the stable entries — an epilogue, a described operand, a statistics hand-over between two launches —
 * whose buffers have layouts private to the library and whose signatures may change from build to build together with the shim.
 * A third-party binding needs the stable entries only; every PZ_FUSED entry computes what a sequence of
 * stable entries computes (tests/test_gpu_3_fusion.py holds them to that sequence, bit for bit where the summation order is kept).
 */
#ifndef PUZZLE_MI355_H
#define PUZZLE_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PZ_FUSED             /* marks entries private to the build's own shim (see "Stability" above) */

typedef void *pz_stream_t;   /* hipStream_t */
typedef void *pz_event_t;    /* hipEvent_t  */
typedef struct pz_pool *pz_pool_t;
typedef struct pz_rng *pz_rng_t;
typedef struct pz_comm *pz_comm_t;
typedef struct pz_module *pz_module_t;

#define PZ_OK 0
#define PZ_ERR_INVALID 1      /* bad argument / unsupported configuration (ValueError on the Python side) */
#define PZ_ERR_HIP 2          /* a HIP runtime call failed                                              */
#define PZ_ERR_NOMEM 3        /* device allocation failed                                               */
#define PZ_ERR_COMM 4         /* RCCL failure / librccl not loadable                                    */

/* ---- library / device: replaces Driver.Device (Cuda/Source/Core/Device.c:160-173) -------------- */
int pz_version(void);
const char *pz_build_id(void);                            /* first 16 hex digits of sha256 over the sources AND compiler flags this .so was built from */
const char *pz_build_flags(void);                         /* the compiler flags themselves (csrc/Makefile: FLAGS + EXTRA); anything but the default = a variant build */
const char *pz_last_error(void);
int pz_init(int device);                                  /* hipSetDevice + arch check (gfx950)  */
int pz_device_count(int *count);
int pz_device_name(int device, char *buf, int buflen);
int pz_device_arch(int device, char *buf, int buflen);
int pz_device_sync(void);
int pz_device_mem_info(size_t *free_bytes, size_t *total_bytes);
int pz_device_num_cus(int device, int *cus);

/* ---- memory: replaces Driver.Buffer / MemoryPool (Cuda/Source/Core/Buffer.c, Allocator.c:29-75,359-362)
 * The pool is a size-class free list: a released block is held and handed out again for the next request
 * of the same class (stream-ordered reuse on the compute stream).                                     */
int pz_malloc(void **ptr, size_t nbytes);
int pz_free(void *ptr);
int pz_pool_create(pz_pool_t *pool);
int pz_pool_destroy(pz_pool_t pool);
int pz_pool_alloc(pz_pool_t pool, size_t nbytes, void **ptr);
int pz_pool_release(pz_pool_t pool, void *ptr);
int pz_pool_free_held(pz_pool_t pool);
int pz_pool_stats(pz_pool_t pool, size_t *held_bytes, size_t *live_bytes, size_t *n_held, size_t *n_live);
int pz_pool_driver_allocs(long *count, double *seconds);   /* pool misses served by hipMalloc in this process, and the host time they took */
int pz_pool_oom_events(long *count);     /* how often an allocation hit "out of memory" (and waited / trimmed) in this process */
int pz_host_alloc_pinned(void **h_ptr, size_t nbytes);
int pz_host_free_pinned(void *h_ptr);

int pz_memcpy_h2d(void *dst, const void *h_src, size_t nbytes, pz_stream_t stream);   /* Buffer.set  Driver.h:80-106 */
int pz_memcpy_d2h(void *h_dst, const void *src, size_t nbytes, pz_stream_t stream);   /* Buffer.get               */
int pz_memcpy_d2d(void *dst, const void *src, size_t nbytes, pz_stream_t stream);     /* Buffer.copy              */
int pz_memcpy_2d(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width_bytes, size_t height,
                 pz_stream_t stream);                                                  /* Driver.memcpy2D Cuda/GPUBackend.py:275-329 */
int pz_memset_d32(void *dst, uint32_t value, size_t count, pz_stream_t stream);       /* Buffer.fillD32 Cuda/GPUArray.py:171-176 */
/* generic strided gather/scatter of a <=6-d view (element size 4): GPUArray.get/set/copy of a non-contiguous view
 * (Cuda/Source/Core/Array.c)                                                                           */
int pz_strided_copy(void *dst, const int64_t *dst_strides, const void *src, const int64_t *src_strides,
                    const int64_t *shape, int ndim, pz_stream_t stream);

/* ---- streams / events: replaces Driver.Stream / Driver.Event (Cuda/Source/Core/Stream.c:101-239) ---- */
int pz_stream_create(pz_stream_t *stream);
/* level < 0 / 0 / > 0: lowest / middle / highest queue priority of the device (hipStreamCreateWithPriority) */
int pz_stream_create_priority(pz_stream_t *stream, int level);
int pz_stream_destroy(pz_stream_t stream);
int pz_stream_sync(pz_stream_t stream);
int pz_stream_wait_event(pz_stream_t stream, pz_event_t event);
int pz_event_create(pz_event_t *event);
int pz_event_destroy(pz_event_t event);
int pz_event_record(pz_event_t event, pz_stream_t stream);
int pz_event_sync(pz_event_t event);
int pz_event_query(pz_event_t event, int *done);       /* Stream.c:219-239 (event.query()): 1 once everything recorded before it has run */
int pz_event_elapsed_ms(pz_event_t start, pz_event_t end, float *ms);

/* ---- convolution: replaces DnnContext.convNd / convNdBackwardData / convNdBackwardParams
 *      (Hip/Wrappers/MIOpen.py:333-462 over miopenConvolution{Forward,BackwardData,BackwardWeights,BackwardBias})
 * Cross-correlation, NCHW x KCRS -> NKPQ, P = (H + 2*pad - dil*(R-1) - 1)/stride + 1.                  */
typedef struct pz_conv_desc {
	int n, c, h, w;                 /* input  (N, C, H, W)                      */
	int k, r, s;                    /* filter (K, C/groups, R, S)               */
	int stride_h, stride_w, pad_h, pad_w, dil_h, dil_w;
	int groups;
} pz_conv_desc;

enum { PZ_CONV_ALGO_AUTO = -1, PZ_CONV_ALGO_DIRECT = 1, PZ_CONV_ALGO_WINOGRAD = 3, PZ_CONV_ALGO_IMPLICIT_GEMM = 5 };
enum { PZ_CONV_FWD = 0, PZ_CONV_BWD_DATA = 1, PZ_CONV_BWD_FILTER = 2 };

int pz_conv2d_out_shape(const pz_conv_desc *d, int *p, int *q);
/* The kernel family a request resolves to: *used = PZ_CONV_ALGO_DIRECT, _WINOGRAD or _IMPLICIT_GEMM for pass `which`
 * under the requested `algo` (what convNdbenchmark, Hip/Wrappers/MIOpen.py:465-519, enumerates and times). */
int pz_conv2d_algo_used(const pz_conv_desc *d, int which, int algo, int *used);
int pz_conv2d_workspace_bytes(const pz_conv_desc *d, int which, int algo, size_t *nbytes);
/* ... the same for a pass that is handed a prepared filter operand (pz_conv2d_fwd_pre / pz_conv2d_bwd_data_pre): only what
 * the launch itself needs (slabs of k-sliced tiles), not a second copy of the packed filters. */
PZ_FUSED int pz_conv2d_workspace_bytes_pre(const pz_conv_desc *d, int which, int algo, size_t *nbytes);
/* y = conv(x, w) (+ bias[k] when bias != NULL) */
/* Activation epilogues (backend-internal fusion behind Conv2D -> Activation(relu), Modules/Activation.py:52-70): is the pass
 * served by a kernel that can apply them (implicit GEMM, output pixels contiguous per image)? */
PZ_FUSED int pz_conv2d_epilogue_supported(const pz_conv_desc *d, int which, int algo, int *supported);
/* y = max(conv(x, w) + bias, 0); w, or packed = a prepared operand (pz_conv2d_prepack) with w ignored */
PZ_FUSED int pz_conv2d_fwd_relu(const pz_conv_desc *d, const float *x, const float *w, const void *packed, const float *bias, float *y, int algo,
                       void *workspace, size_t ws_bytes, pz_stream_t stream);
/* dx = bwd_data(dy, w) where gate > 0, else 0 (gate: dx's shape — the output of the ReLU in front of this convolution, whose
 * reluDer follows; Modules/Activation.py:58-60) */
PZ_FUSED int pz_conv2d_bwd_data_gate(const pz_conv_desc *d, const float *dy, const float *w, const float *gate, float *dx, int algo,
                            void *workspace, size_t ws_bytes, pz_stream_t stream);
int pz_conv2d_fwd(const pz_conv_desc *d, const float *x, const float *w, const float *bias, float *y,
                  int algo, void *workspace, size_t ws_bytes, pz_stream_t stream);
/* Forward convolution that also leaves, for a batch normalisation reading y next (Conv -> BatchNorm pairs of
 * Models/Nets/ResNet.py:27-30), per-channel shifted sums over strips of PZ_CONV_STATS_STRIP consecutive
 * output pixels (flattened n*P*Q axis): stats[(k*strips + strip)*4 + {0,1,2}] = {shift, sum(y-shift), sum((y-shift)^2)},
 * shift = the strip's first value of channel k. pz_conv2d_fwd_stats_strips reports the number of strips (0: this
 * configuration runs on a path that cannot produce them — call pz_conv2d_fwd and let the BN compute its own).  */
#define PZ_CONV_STATS_STRIP 64
PZ_FUSED int pz_conv2d_fwd_stats_strips(const pz_conv_desc *d, int algo, int *strips);
PZ_FUSED int pz_conv2d_fwd_stats(const pz_conv_desc *d, const float *x, const float *w, const float *bias, float *y,
                        float *stats, int algo, void *workspace, size_t ws_bytes, pz_stream_t stream);
/* dx = conv^T(dy, w); dx has the (n,c,h,w) of the descriptor */
int pz_conv2d_bwd_data(const pz_conv_desc *d, const float *dy, const float *w, float *dx,
                       int algo, void *workspace, size_t ws_bytes, pz_stream_t stream);
/* Filter operands prepared ahead of the pass (they depend on the parameters only): pz_conv2d_prepack_bytes = size of what
 * pass `which` (PZ_CONV_FWD / PZ_CONV_BWD_DATA) derives from the filter tensor — 0 when it reads the tensor itself;
 * pz_conv2d_prepack fills any number of them in a few launches; pz_conv2d_{fwd,bwd_data}_pre are pz_conv2d_fwd_stats /
 * pz_conv2d_bwd_data taking the prepared operand instead of the filter tensor. The caller owns the buffers and re-prepares
 * them when the parameters change (puzzlelib_amd/backend.py: once per training step, behind the optimizer's update).    */
typedef struct {
	pz_conv_desc desc;
	int which, algo;
	const float *w;
	void *packed;
} pz_prepack_job;
PZ_FUSED int pz_conv2d_prepack_bytes(const pz_conv_desc *d, int which, int algo, size_t *nbytes);
PZ_FUSED int pz_conv2d_prepack(const pz_prepack_job *jobs, int njobs, pz_stream_t stream);
PZ_FUSED int pz_conv2d_fwd_pre(const pz_conv_desc *d, const float *x, const void *packed, const float *bias, float *y, float *stats,
                      int algo, void *workspace, size_t ws_bytes, pz_stream_t stream);
PZ_FUSED int pz_conv2d_bwd_data_pre(const pz_conv_desc *d, const float *dy, const void *packed, float *dx, int algo, void *workspace,
                           size_t ws_bytes, pz_stream_t stream);
/* dw <- beta*dw + alpha*sum(x (x) dy); db (optional) <- beta*db + alpha*sum(dy): the accumulate contract of
 * MIOpen.py:414-433,441-455 (scale = alpha, momentum = beta) fused into the reduction epilogue. With the workspace of
 * pz_conv2d_workspace_bytes the bias gradient is summed inside the filter-gradient kernel (from the dy runs it stages
 * anyway) and reduced with the filter gradient's slabs; without a workspace a one-stage kernel sums it.            */
int pz_conv2d_bwd_filter(const pz_conv_desc *d, const float *x, const float *dy, float *dw, float *db,
                         float alpha, float beta, int algo, void *workspace, size_t ws_bytes, pz_stream_t stream);

/* How the matrix pipe multiplies the fp32 operands of the MFMA convolution / GEMM kernels (the counterpart of the
 * reference's per-context math switches — cudnnSetConvolutionMathType in Cuda/Source/Libs/CuDnn.c, MIOpen picks its fp32
 * solver itself): 0 = v_mfma_f32_32x32x2_f32; 6 or 9 = each fp32 operand is split exactly into three bf16 terms
 * (8 + 8 + 8 significand bits) and the product is the sum of 6 / 9 exact bf16 partial products accumulated in fp32 by
 * v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate on gfx950). 9 terms: a*b exactly; 6 terms: the three terms below
 * 2^-23 |a||b| are left out. Inputs, outputs and accumulation stay fp32 in every mode. Process-wide; set it before
 * querying workspace sizes (the packed-filter part of a workspace is 1.5x larger in the split modes).                */
int pz_conv_math_set(int products);
int pz_conv_math_get(int *products);

/* Output tile of the Winograd 3x3 forward / backward-data kernels (ConvFwdAlgo.winograd, Hip/Wrappers/MIOpen.py:28 — MIOpen
 * picks its Winograd variant itself): 0 = by multiplication count per layer (F(4x4,3x3) where the map fills 4x4 tiles, else
 * F(2x2,3x3)), 2 = F(2x2,3x3) only, 4 = F(4x4,3x3) on every eligible layer. F(4x4) multiplies 4x less than the direct sum
 * (F(2x2): 2.25x) and rounds ~8x coarser (about 4e-6 relative L2 in fp32). Process-wide; set it before querying workspace
 * sizes or preparing filter operands.                                                                                  */
int pz_conv_winograd_tile_set(int tile);
int pz_conv_winograd_tile_get(int *tile);

/* Launch-level profiling of the convolution kernels (bench.py's roofline leg; the analogue of the reference's
 * Driver timing hooks, Cuda/GPUBackend.py:332-368): while enabled, every MFMA convolution launch is bracketed by
 * HIP events on the launch stream. collect() synchronises, sums per kernel family and resets.
 * family: 0 = igemm 128x128 tile, 1 = igemm 64x256 tile, 2 = backward-filter (all tiles), 3 = Winograd
 * (forward, backward-data and backward-filter of 3x3 stride-1 layers; F(4x4,3x3) / F(2x2,3x3) tiles, see pz_conv_winograd_tile_set). `total_flops` is the algorithmic (direct-convolution) count in every family; the Winograd
 * kernel executes 1/2.25 of it on the matrix pipe (times the padding of odd maps to whole 2x2 tiles).            */
#define PZ_CONV_PROFILE_FAMILIES 4
PZ_FUSED int pz_conv_profile_enable(int on);
PZ_FUSED int pz_conv_profile_collect(double total_ms[PZ_CONV_PROFILE_FAMILIES], double total_flops[PZ_CONV_PROFILE_FAMILIES],
                            long long launches[PZ_CONV_PROFILE_FAMILIES]);

/* ---- GEMM: replaces BlasContext.gemm (Cuda/Source/Libs/CuBlas.c:327-402); row-major,
 *      C[M,N] = alpha*op(A)*op(B) + beta*C, lda/ldb/ldc = row pitches in elements.                    */
int pz_gemm(int trans_a, int trans_b, int m, int n, int k, float alpha, const float *a, int lda,
            const float *b, int ldb, float beta, float *c, int ldc, pz_stream_t stream);
/* the same with scratch for a split along K (small outputs with long reductions — the 256 x 1000 x 2048 classifier —
 * would otherwise run on a handful of the 256 CUs); partial tiles are added in a fixed order, no atomics.
 * Batched GEMM (BlasContext.gemmBatched, CuBlas.c:308-312, both group formats) is a loop of these calls over the
 * groups: lda / ldb / ldc express the "bgp" layout directly.                                                      */
PZ_FUSED int pz_gemm_workspace_bytes(int m, int n, int k, size_t *nbytes);
PZ_FUSED int pz_gemm_ws(int trans_a, int trans_b, int m, int n, int k, float alpha, const float *a, int lda, const float *b, int ldb,
               float beta, float *c, int ldc, void *workspace, size_t ws_bytes, pz_stream_t stream);

/* ---- batch normalisation (spatial): replaces DnnContext.batchNormNd / batchNormNdBackward
 *      (Hip/Wrappers/MIOpen.py:634-688). x viewed as (n, c, hw). running stats updated in place:
 *      run <- (1-factor)*run + factor*batch (variance: unbiased). ws: pz_bn_workspace_bytes.            */
int pz_bn_workspace_bytes(int n, int c, int hw, size_t *nbytes);
int pz_bn_fwd_train(const float *x, float *y, int n, int c, int hw, const float *scale, const float *bias,
                    float *run_mean, float *run_var, float *save_mean, float *save_invvar,
                    float epsilon, float factor, void *workspace, size_t ws_bytes, pz_stream_t stream);
int pz_bn_fwd_infer(const float *x, float *y, int n, int c, int hw, const float *scale, const float *bias,
                    const float *mean, const float *var, float epsilon, pz_stream_t stream);
int pz_bn_bwd(const float *x, const float *dy, float *dx, int n, int c, int hw, const float *scale,
              const float *save_mean, const float *save_invvar, float *dscale, float *dbias,
              void *workspace, size_t ws_bytes, pz_stream_t stream);

/* Backend-internal fusion of BatchNorm with a following in-place ReLU (Models/Nets/ResNet.py:30-33 builds exactly that
 * pair): act = PZ_BN_ACT_RELU makes the forward write relu(bn(x)) and the backward gate dy with
 * (bn(x) > 0), re-created from x, scale, bias and the saved statistics — reluDer's rule (Cuda/Kernels/ElementWise.py
 * :119-172) without the two extra tensor passes. act = PZ_BN_ACT_NONE is pz_bn_fwd_train / pz_bn_bwd.            */
enum pz_bn_act { PZ_BN_ACT_NONE = 0, PZ_BN_ACT_RELU = 1 };
PZ_FUSED int pz_bn_fwd_train_act(const float *x, float *y, int n, int c, int hw, const float *scale, const float *bias,
                        float *run_mean, float *run_var, float *save_mean, float *save_invvar,
                        float epsilon, float factor, int act, void *workspace, size_t ws_bytes, pz_stream_t stream);
/* pz_bn_fwd_train_act with the statistics pass replaced by the producer's strip sums (pz_conv2d_fwd_stats): the
 * strips are merged per channel in fp64 with the pairwise (count, mean, M2) update, in a fixed order.            */
PZ_FUSED int pz_bn_fwd_train_pre(const float *x, float *y, int n, int c, int hw, const float *scale, const float *bias,
                        float *run_mean, float *run_var, float *save_mean, float *save_invvar,
                        float epsilon, float factor, int act, const float *stats, int strips,
                        void *workspace, size_t ws_bytes, pz_stream_t stream);
/* pz_bn_bwd_act that also accumulates the parameter gradients where the optimiser reads them:
 * dscale_acc <- alpha*dscale + beta*dscale_acc (same for dbias_acc; either may be NULL) — BatchNormND.accGradParams
 * (Modules/BatchNormND.py:74-83, Blas.addVectorToVector with alpha = scale, beta = momentum) without its two launches. */
PZ_FUSED int pz_bn_bwd_acc(const float *x, const float *dy, float *dx, int n, int c, int hw, const float *scale,
                  const float *bias, const float *save_mean, const float *save_invvar, float *dscale, float *dbias,
                  int act, float *dscale_acc, float *dbias_acc, float alpha, float beta,
                  void *workspace, size_t ws_bytes, pz_stream_t stream);
/* Residual fan-in fused with the statistics pass of the BatchNorm backward(s) it feeds (Replicate.updateGrad + reluDer +
 * the first half of batchNormNdBackward for bn*_branch2c and the projection-shortcut BN of Models/Nets/ResNet.py:36-58):
 * gout = (g0 + g1) * (y > 0); part_a[(k*splits + s)*2 + {0,1}] = {sum gout, sum gout*(xa - mean_a[k])} in the layout and
 * accumulation order pz_bn_bwd uses internally (bit-identical), likewise part_b for the optional second BN (xb == NULL:
 * none). Each part needs pz_bn_workspace_bytes. pz_bn_bwd_from_partials is then pz_bn_bwd_acc without its statistics
 * pass.                                                                                                          */
/* `mask` (optional, from pz_bn_apply_add_mask over the tensor y): the gate is read from one bit per element and y is not
 * touched (may be NULL) — 1/16 of the bytes; the same predicate, so the same results. */
PZ_FUSED int pz_bn_gate_stats(const float *g0, const float *g1, const float *y, const unsigned char *mask, float *gout, int n, int c,
                     int hw, const float *xa, const float *mean_a, float *part_a,
                     const float *xb, const float *mean_b, float *part_b, pz_stream_t stream);
/* pz_bn_gate_stats whose two incoming gradients come from stride-2 pointwise convolutions (the first convolutions of a
 * down-sampling block's two branches, Models/Nets/ResNet.py:36-46) and stay compact: g0c, g1c are (n, c, ceil(h/2),
 * ceil(w/2)) = the values at pixels (2i, 2j), every other pixel of the (n, c, h, w) gradients being zero. Same results,
 * bit for bit, as zero-filling them first; their backward-data passes write a quarter of the tensor (as a stride-1
 * problem on the compact grid) and nothing is memset. */
PZ_FUSED int pz_bn_gate_stats_up2(const float *g0c, const float *g1c, const float *y, const unsigned char *mask, float *gout, int n,
                         int c, int h, int w, const float *xa, const float *mean_a, float *part_a,
                         const float *xb, const float *mean_b, float *part_b, pz_stream_t stream);
PZ_FUSED int pz_bn_bwd_from_partials(const float *x, const float *dy, float *dx, int n, int c, int hw, const float *scale,
                            const float *save_mean, const float *save_invvar, float *dscale, float *dbias,
                            float *dscale_acc, float *dbias_acc, float alpha, float beta, const float *partials,
                            pz_stream_t stream);
/* Training-mode forward without the normalisation pass (what miopenBatchNormalizationForwardTraining computes per channel,
 * Hip/Wrappers/MIOpen.py:634-664, minus writing y): statistics from the producing convolution's strip sums (`stats`,
 * pz_conv2d_fwd_stats) or, with stats == NULL, from a pass over x; saved / running statistics; coef[2k..2k+1] = {a, b} of
 * y = a*x + b for whoever reads the normalised tensor (pz_bn_apply_add). Workspace: pz_bn_workspace_bytes.            */
PZ_FUSED int pz_bn_fwd_train_coef(const float *x, int n, int c, int hw, const float *scale, const float *bias, float *run_mean,
                         float *run_var, float *save_mean, float *save_invvar, float epsilon, float factor,
                         const float *stats, int strips, float *coef, void *workspace, size_t ws_bytes, pz_stream_t stream);
/* pz_bn_bwd whose incoming gradient is first gated with (y > 0), y = a*x + b re-created from x with the forward's own
 * pairs `gate_coef` (pz_bn_fwd_train_coef): BatchNormND.backward + Activation(relu, inplace).backward of the reference
 * (Modules/BatchNormND.py:75-88, Modules/Activation.py:62-70) in one pair of passes.                                */
PZ_FUSED int pz_bn_bwd_gate(const float *x, const float *dy, float *dx, int n, int c, int hw, const float *scale,
                   const float *save_mean, const float *save_invvar, float *dscale, float *dbias, const float *gate_coef,
                   void *workspace, size_t ws_bytes, pz_stream_t stream);
/* the statistics pass of pz_bn_bwd alone: `partials` (pz_bn_workspace_bytes) in the layout pz_bn_bwd_coef /
 * pz_bn_bwd_from_partials consume                                                                                  */
PZ_FUSED int pz_bn_bwd_stats(const float *x, const float *dy, int n, int c, int hw, const float *save_mean, float *partials,
                    pz_stream_t stream);
/* BatchNorm backward folded into the gathers of the convolution in front of it (Conv2D -> BatchNorm2D, both backward):
 * pz_bn_bwd_coef turns the partial sums (pz_bn_gate_stats) into the parameter gradients and coef[4k..4k+2] = {A, B, C}
 * with dx_bn = A*dy + B*x + C per channel; pz_conv2d_bwd_data_bn / pz_conv2d_bwd_filter_bn are pz_conv2d_bwd_data /
 * pz_conv2d_bwd_filter whose `dy` operand is that expression evaluated on the fly from dy (the BN's incoming gradient)
 * and bnx (the BN's input = this convolution's forward output). The BN's 12 B/elem apply pass and its output tensor
 * disappear. Only for convolutions pz_conv2d_bn_fold_supported accepts (1x1, no padding, ungrouped, MFMA path).   */
PZ_FUSED int pz_bn_bwd_coef(int n, int c, int hw, const float *scale, const float *save_mean, const float *save_invvar,
                   float *dscale, float *dbias, float *dscale_acc, float *dbias_acc, float alpha, float beta,
                   const float *partials, float *coef, pz_stream_t stream);
/* the same expression written out, dx = A*dy + (B*x + C), for a consumer that cannot fold it */
PZ_FUSED int pz_bn_bwd_apply_coef(const float *x, const float *dy, float *dx, int n, int c, int hw, const float *coef,
                         pz_stream_t stream);
PZ_FUSED int pz_conv2d_bn_fold_supported(const pz_conv_desc *d, int algo, int *supported);
PZ_FUSED int pz_conv2d_bwd_data_bn(const pz_conv_desc *d, const float *dy, const float *bnx, const float *bncoef,
                          const float *w, float *dx, int algo, void *workspace, size_t ws_bytes, pz_stream_t stream);
PZ_FUSED int pz_conv2d_bwd_filter_bn(const pz_conv_desc *d, const float *x, const float *dy, const float *bnx,
                            const float *bncoef, float *dw, float alpha, float beta, int algo,
                            void *workspace, size_t ws_bytes, pz_stream_t stream);
/* The backward twin of pz_conv2d_fwd_stats (round 6): the layer's input was y = relu(gab[c].x * gx + gab[c].y) — the ReLU output of
 * the BatchNorm in front of it (conv -> bn -> relu -> THIS conv, Models/Nets/ResNet.py:27-33), so the gradient dx this launch
 * produces goes through that ReLU's derivative and into that BatchNorm's backward next (Modules/Activation.py:62-70,
 * Modules/BatchNormND.py:65-72; formulas Cuda/Wrappers/CuDnnNorm.py:55-63). The epilogue, holding the dx tile, reads the same tile
 * of gx and leaves {sum q, sum q * (gx - gmean[c])}, q = dx * (y > 0), per channel in `partials` (pz_conv2d_bwd_data_bnstats_bytes
 * bytes; the merged pair per channel where pz_bn_bwd_gate_from_partials / pz_bn_bwd_coef read it): the statistics pass of that
 * BatchNorm's backward — two reads of tensors of dx's size — does not run. dx itself is stored ungated, as pz_conv2d_bwd_data
 * stores it. bnx / bncoef: optional fold of the BatchNorm BEHIND the layer on the gathered side (pz_conv2d_bwd_data_bn), or NULL.
 * Only for configurations with the contiguous-output epilogue (pz_conv2d_epilogue_supported; _bytes returns 0 otherwise). */
PZ_FUSED int pz_conv2d_bwd_data_bnstats_bytes(const pz_conv_desc *d, int algo, size_t *nbytes);
PZ_FUSED int pz_conv2d_bwd_data_bnstats(const pz_conv_desc *d, const float *dy, const float *bnx, const float *bncoef, const float *w,
                               float *dx, const float *gx, const float *gab, const float *gmean, float *partials, int algo,
                               void *workspace, size_t ws_bytes, pz_stream_t stream);
/* pz_bn_bwd_gate with the statistics pass already done (partials from pz_conv2d_bwd_data_bnstats): the apply pass alone */
PZ_FUSED int pz_bn_bwd_gate_from_partials(const float *x, const float *dy, float *dx, int n, int c, int hw, const float *scale,
                                 const float *save_mean, const float *save_invvar, float *dscale, float *dbias,
                                 const float *gate_coef, const float *partials, pz_stream_t stream);
/* The mirror image on the forward side: a BatchNorm (+ in-place ReLU) whose only readers are the pointwise
 * convolution behind it — its forward and its filter gradient (conv -> bn -> relu -> conv 1x1 inside every bottleneck block of
 * Models/Nets/ResNet.py:27-33) — is never written: both passes read the BatchNorm's INPUT x and evaluate
 * relu?(xcoef[2c] * x + xcoef[2c + 1]) per input channel c while gathering ({a, b} pairs of pz_bn_fwd_train_coef; the expression
 * of pz_bn_apply_add, so the results are bit-identical to convolving the written tensor). w or packed (pz_conv2d_prepack) as in
 * pz_conv2d_fwd_relu; stats as in pz_conv2d_fwd_stats (NULL: none); bnx / bncoef of pz_conv2d_bwd_filter_xbn are the optional
 * gradient-side fold of pz_conv2d_bwd_filter_bn (both NULL: dy is read as it is).                                          */
PZ_FUSED int pz_conv2d_xbn_supported(const pz_conv_desc *d, int which, int algo, int *supported);
PZ_FUSED int pz_conv2d_fwd_xbn(const pz_conv_desc *d, const float *x, const float *xcoef, int xrelu, const float *w, const void *packed,
                      const float *bias, float *y, float *stats, int algo, void *workspace, size_t ws_bytes, pz_stream_t stream);
PZ_FUSED int pz_conv2d_bwd_filter_xbn(const pz_conv_desc *d, const float *x, const float *xcoef, int xrelu, const float *dy, const float *bnx,
                             const float *bncoef, float *dw, float alpha, float beta, int algo, void *workspace, size_t ws_bytes,
                             pz_stream_t stream);
/* Deferred apply: for a BatchNorm whose only consumer is a residual Add (bn*_branch2c and the
 * projection shortcut of Models/Nets/ResNet.py:36-58) the normalised tensor is never written. pz_bn_fwd_train_defer does
 * everything pz_bn_fwd_train_pre does except the pass over x and returns coef[2k..2k+1] = {a, b} of y = a*x + b;
 * pz_bn_apply_add computes out = act((a1*x1 + b1) + (coef2 ? a2*x2 + b2 : x2)) — or out = a1*x1 + b1 when x2 == NULL —
 * with the same fma and summation order as the unfused kernels (bit-identical), saving 8 B/elem per deferred BN. */
PZ_FUSED int pz_bn_fwd_train_defer(int n, int c, int hw, const float *scale, const float *bias, float *run_mean, float *run_var,
                          float *save_mean, float *save_invvar, float epsilon, float factor, const float *stats,
                          int strips, float *coef, void *workspace, size_t ws_bytes, pz_stream_t stream);
PZ_FUSED int pz_bn_apply_add(const float *x1, const float *coef1, const float *x2, const float *coef2, float *out,
                    int n, int c, int hw, int relu, pz_stream_t stream);
/* ... also leaving the sign mask of the fused ReLU's output (pz_relu_mask_bytes bytes: one bit per element, a byte per 4
 * consecutive elements of a (n, channel) plane) for pz_bn_gate_stats, which then does not read `out` back. */
PZ_FUSED int pz_relu_mask_bytes(int n, int c, int hw, size_t *nbytes);
PZ_FUSED int pz_bn_apply_add_mask(const float *x1, const float *coef1, const float *x2, const float *coef2, float *out,
                         unsigned char *mask, int n, int c, int hw, int relu, pz_stream_t stream);
PZ_FUSED int pz_bn_bwd_act(const float *x, const float *dy, float *dx, int n, int c, int hw, const float *scale,
                  const float *bias, const float *save_mean, const float *save_invvar, float *dscale, float *dbias,
                  int act, void *workspace, size_t ws_bytes, pz_stream_t stream);

/* ---- pooling: replaces DnnContext.poolNd / poolNdBackward (Hip/Wrappers/MIOpen.py:549-598).
 *      index workspace (uint8 per output element, window-local arg-max) is optional for forward (NULL in
 *      test mode) and required for max backward.                                                        */
typedef struct pz_pool_desc {
	int n, c, h, w;
	int size_h, size_w, stride_h, stride_w, pad_h, pad_w;
	int mode;                       /* 0 max, 1 average incl. padding, 2 average excl. padding */
} pz_pool_desc;

int pz_pool2d_out_shape(const pz_pool_desc *d, int *p, int *q);
int pz_pool2d_fwd(const pz_pool_desc *d, const float *x, float *y, uint8_t *index_ws, pz_stream_t stream);
/* Max pooling over a batch normalisation that was only described (the lazy-buffer layer, puzzlelib_amd/lazy.py): reads the
 * BN's INPUT x and applies y = a*x + b per channel (coef = (c, 2) pairs from pz_bn_fwd_train_coef), optionally through
 * ReLU, while it stages the rows — the normalised tensor between Modules/BatchNorm2D.py, Modules/Activation.py (in place)
 * and Modules/MaxPool2D.py (the ResNet stem, Models/Nets/ResNet.py:88-96) is never written. Bit-identical to pz_bn_apply_add
 * followed by pz_pool2d_fwd. Only for the geometries pz_pool2d_fwd_bn_supported reports (max pooling, band kernel). */
PZ_FUSED int pz_pool2d_fwd_bn_supported(const pz_pool_desc *d, int *supported);
PZ_FUSED int pz_pool2d_fwd_bn(const pz_pool_desc *d, const float *x, const float *coef, int relu, float *y, uint8_t *index_ws,
                     pz_stream_t stream);
/* x/y are only read for max pooling when index_ws == NULL (arg-max recomputed, first maximum wins) */
int pz_pool2d_bwd(const pz_pool_desc *d, const float *dy, const float *x, const float *y, const uint8_t *index_ws,
                  float *dx, pz_stream_t stream);

/* ---- softmax / cross-entropy: replaces DnnContext.softmaxNd(+Backward) (MIOpen.py:601-631, channel mode,
 *      "accurate") and CostModule.crossEntropy (Cuda/Kernels/Costs.py:79-106,213-247). Tensors (n, c, spatial). */
int pz_softmax_fwd(const float *x, float *y, int n, int c, int spatial, pz_stream_t stream);
int pz_softmax_bwd(const float *dy, const float *y, float *dx, int n, int c, int spatial, pz_stream_t stream);
/* grad = w[c]*((c==label) - softmax(scores))/n ; *error = sum(-w*log p[label])/spatial (deterministic).
 * workspace: n*spatial floats. weights may be NULL.                                                    */
int pz_cross_entropy(const float *scores, const int32_t *labels, const float *weights, int n, int c, int spatial,
                     float *grad, float *error, void *workspace, size_t ws_bytes, pz_stream_t stream);

/* ---- reductions / matrix-vector: replaces MatModule (Cuda/Kernels/MatVec.py:231-374), ReductionKernel users
 *      (Cuda/GPUArray.py:80-103, Cuda/Kernels/Costs.py:178-182) and BlasContext.dot/l1norm (CuBlas.c:486-499). */
int pz_reduce_sum_rows(const float *t, int rows, int cols, float *out, float alpha, float beta, pz_stream_t stream);
int pz_reduce_sum_cols(const float *t, int z, int h, int w, float *out, float alpha, float beta, pz_stream_t stream);
int pz_argmax_rows(const float *t, int rows, int cols, int32_t *out, pz_stream_t stream);
int pz_argmax_cols(const float *t, int z, int h, int w, int32_t *out, pz_stream_t stream);
/* out[b] = mat[b] + vec[b] broadcast: axis 0 -> row i gets vec[i % veclen] (veclen | n: a (batch*maps, pixels) view
 * takes one value per map), axis 1 -> column j gets vec[j % veclen] (veclen | m). z matrices, z vectors of veclen. */
int pz_bias_add(float *out, const float *mat, const float *vec, int z, int n, int m, int veclen, int axis,
                pz_stream_t stream);
int pz_count_neq_i32(const int32_t *x, const int32_t *y, size_t count, float *out, pz_stream_t stream);
/* the accuracy / divergence reductions of CostModule.getAccuracyKernel (Cuda/Kernels/Costs.py:184-203; callers Cost/BCE.py:32-33,
 * Cost/L1Hinge.py:40-41, Cost/KLDivergence.py:33-34,55-56): kind 0 = calcBCEAccuracy (count of labels == 1 ? x <= 0 : x > 0),
 * kind 1 = l1HingeAccuracy (count of (d <= 1) != label); pz_kl_divergence writes grad = (y - x) * gradnorm and
 * *out = sum of y (log y - log x) over y > 0. Two-stage sums in a fixed order.                                              */
int pz_cost_accuracy(int kind, const float *x, const int32_t *labels, size_t count, float *out, pz_stream_t stream);
int pz_kl_divergence(const float *x, const float *y, float *grad, float gradnorm, size_t count, float *out, pz_stream_t stream);
int pz_reduce_minmax_f32(const float *x, size_t count, int is_max, float *out, pz_stream_t stream);
int pz_reduce_minmax_i32(const int32_t *x, size_t count, int is_max, int32_t *out, pz_stream_t stream);
int pz_dot(const float *x, const float *y, size_t count, float *out, pz_stream_t stream);
int pz_asum(const float *x, size_t count, float *out, pz_stream_t stream);

/* ---- element-wise family: replaces the ElementwiseKernel objects of Cuda/Kernels/ElementWise.py (launched
 *      through Cuda/SourceModule.py:176-226, incl. the `slice=` strided variant: start/stop/step; pass
 *      start=0, stop=count, step=1 for the dense case). ptrs[0] is the output (or the in-place operand).
 *      Operand/scalar order per op is the reference kernel's argument order and is listed next to each id. */
enum pz_eltwise_op {
	PZ_OP_SIGMOID = 0,        /* out, in                               */
	PZ_OP_SIGMOID_DER,        /* ingrad, outgrad, outdata              */
	PZ_OP_TANH, PZ_OP_TANH_DER,
	PZ_OP_RELU, PZ_OP_RELU_DER,
	PZ_OP_LEAKY_RELU, PZ_OP_LEAKY_RELU_DER,   /* + scalar a            */
	PZ_OP_ELU, PZ_OP_ELU_DER,                 /* + scalar a            */
	PZ_OP_SOFTPLUS, PZ_OP_SOFTPLUS_DER,
	PZ_OP_CLIP, PZ_OP_CLIP_DER,               /* + scalars a, b        */
	PZ_OP_GELU, PZ_OP_GELU_DER,               /* der: ingrad, outgrad, indata */
	PZ_OP_DROPOUT,            /* out, in, bits(u32); scalars v(as float bits), p       */
	PZ_OP_DROPOUT2D,          /* out, in, bits(u32); scalars v, p, mapsize(int bits)   */
	PZ_OP_AXPY,               /* y, x; alpha          : y += alpha*x  (toVectorAddVectorKer) */
	PZ_OP_ADD,                /* out, x, y; alpha, beta: out = alpha*x + beta*y (addKer)     */
	PZ_OP_MUL,                /* out, a, b                                                     */
	PZ_OP_LINEAR,             /* out, in; a, b        : out = a*in + b                         */
	PZ_OP_ABS,                /* out, in                                                       */
	PZ_OP_WEIGHT_DECAY,       /* grad, param; rate    : grad -= rate*param                     */
	PZ_OP_L1_PENALTY,         /* outgrad, ingrad, data; a                                      */
	PZ_OP_L1_GRAD,            /* grad, pred, target; norm                                      */
	PZ_OP_RBM,                /* out, in, uni                                                  */
	PZ_OP_ADAM,               /* param, grad, mg, ms; learnRate, fix1, fix2, epsilon, gradScale (the gradient is read as
	                           * grad * gradScale: 1, or 1/N behind a data-parallel sum — Grid.py:126-133) */
	PZ_OP_CLASSIC_MOM_SGD,    /* param, grad, mom; learnRate, momRate, gradScale               */
	PZ_OP_NESTEROV_MOM_SGD,   /* param, grad, mom; learnRate, momRate, gradScale               */
	PZ_OP_RMSPROP,            /* param, grad, ms; learnRate, factor, epsilon                   */
	PZ_OP_ADAGRAD,            /* param, grad, h; learnRate, epsilon                            */
	PZ_OP_ADADELTA,           /* param, grad, msg, msdx; rho, epsilon                          */
	PZ_OP_RMSPROP_GRAVES,     /* param, grad, mg, ms, delta; learnRate, alpha, momRate, epsilon*/
	PZ_OP_SMORMS3,            /* param, grad, mem, mg, ms; learnRate, epsilon                  */
	PZ_OP_ADD3,               /* out, a, b            : out = a + b (Add.updateData / Replicate.updateGrad fused) */
	PZ_OP_IADD,               /* out, in              : out += in  (GPUArray.__iadd__)         */
	PZ_OP_IMUL,               /* out, in              : out *= in                              */
	PZ_OP_ADD3_RELU,          /* out, a, b            : out = relu(a + b)  (Add followed by an in-place ReLU, fused) */
	PZ_OP_ADD3_GATE,          /* out, a, b, y         : out = (a + b) * (y > 0)  (Replicate fan-in + reluDer of the
	                             in-place ReLU that produced its input y, fused)                   */
	PZ_OP_COUNT
};

int pz_eltwise(int op, size_t count, void *const *ptrs, int nptrs, const float *scalars, int nscalars,
               int64_t start, int64_t stop, int64_t step, pz_stream_t stream);
/* up to PZ_MULTI_ADD_MAX independent small `out = alpha*x + beta*y` (addKer, Cuda/Kernels/ElementWise.py:1017-1045) in one
 * launch, one workgroup per job: the per-layer parameter-gradient accumulates of BatchNormND.accGradParams              */
#define PZ_MULTI_ADD_MAX 96
PZ_FUSED int pz_multi_add(int njobs, float *const *out, const float *const *x, const float *const *y, const float *alpha,
                 const float *beta, const unsigned *n, pz_stream_t stream);
int pz_cast_i32_f32(float *out, const int32_t *in, size_t count, pz_stream_t stream);
int pz_cast_f32_i32(int32_t *out, const float *in, size_t count, pz_stream_t stream);
/* fp16 as a STORAGE type only (GPUArray.astype, Cuda/GPUArray.py:188-199; castFP32toFP16 / castFP16toFP32,
 * Cuda/Kernels/ElementWise.py:1143-1156): IEEE half bit patterns, round to nearest even. No operator computes in fp16. */
int pz_cast_f32_f16(uint16_t *out, const float *in, size_t count, pz_stream_t stream);
int pz_cast_f16_f32(float *out, const uint16_t *in, size_t count, pz_stream_t stream);

/* ---- RNG: replaces RandomNumberGenerator.fillInteger/fillUniform/fillNormal (Cuda/Source/Libs/CuRand.c:231-234).
 *      Counter-based Philox4x32-10; statistical parity only (the reference's XORWOW stream is not reproduced). */
int pz_rng_create(uint64_t seed, pz_rng_t *rng);
int pz_rng_destroy(pz_rng_t rng);
int pz_rng_fill_u32(pz_rng_t rng, uint32_t *out, size_t count, pz_stream_t stream);
int pz_rng_fill_uniform(pz_rng_t rng, float *out, size_t count, pz_stream_t stream);           /* (0, 1] */
int pz_rng_fill_normal(pz_rng_t rng, float *out, size_t count, float mean, float stddev, pz_stream_t stream);

/* ---- beside the ResNet / NiN / LeNet path -------------------------------------------------------------
 * mask pooling: PoolModule.maxpool2d / maxpool2dBackward / maxunpool2d / maxunpool2dBackward (Cuda/Kernels/Pool.py:7-114);
 * mask[n, c, p, q] = flat index h*W + w of the maximum inside the (n, c) input plane, -1 for an empty window            */
int pz_maskpool2d_fwd(const pz_pool_desc *d, const float *x, float *y, int32_t *mask, pz_stream_t stream);
int pz_maskpool2d_bwd(const pz_pool_desc *d, const float *dy, const int32_t *mask, float *dx, pz_stream_t stream);
int pz_maxunpool2d_fwd(const float *x, const int32_t *mask, float *y, size_t planes, size_t in_plane, size_t out_plane,
                       pz_stream_t stream);
int pz_maxunpool2d_bwd(const float *dy, const int32_t *mask, float *dx, size_t planes, size_t in_plane, size_t out_plane,
                       pz_stream_t stream);
/* local response normalisation: DnnContext.lrn / lrnBackward (Hip/Wrappers/MIOpen.py:691-751); cross = 0: N x N window
 * inside a map (LRNMode.map), 1: N neighbouring maps (LRNMode.cross); `scale` (same shape as x) is the training-mode
 * workspace the backward reads, NULL in inference                                                                     */
int pz_lrn_fwd(const float *x, float *y, float *scale, int n, int c, int h, int w, int size, float alpha, float beta, float k,
               int cross, pz_stream_t stream);
int pz_lrn_bwd(const float *x, const float *dy, const float *scale, float *dx, int n, int c, int h, int w, int size,
               float alpha, float beta, float k, int cross, pz_stream_t stream);
/* SVM cost (CostModule.svm, Cuda/Kernels/Costs.py:109-130,250-276): gradient per score and the per-element error terms
 * (the caller sums them with pz_asum — they are non-negative — instead of the reference's atomicAdd)                */
int pz_svm_cost(const float *scores, const int32_t *labels, int samples, int cases, int spatial, int squared, float *grad,
                float *terms, pz_stream_t stream);
/* MatModule.matvec / argmin (Cuda/Kernels/MatVec.py:231-345) */
int pz_matvec(const float *mat, const float *vec, float *out, int z, int h, int w, int axis, float alpha, float beta,
              pz_stream_t stream);
int pz_argmin_rows(const float *t, int rows, int cols, int32_t *out, pz_stream_t stream);
int pz_argmin_cols(const float *t, int z, int h, int w, int32_t *out, pz_stream_t stream);
/* point-wise cost kernels bceKer / hingeKer / smoothL1Ker / l1HingeKer (Cuda/Kernels/Costs.py:8-72; callers Cost/BCE.py,
 * Hinge.py, SmoothL1.py, L1Hinge.py): `grad` (and `grad2` for l1Hinge's second operand) per element, the per-element
 * error terms to `terms` (scratch, `total` floats), and *error += sum(terms) in a fixed order — the reference atomicAdds
 * into the same 0-d array. a = scores / pred / x1; b = target / x2 (float, smoothL1 and l1Hinge only); labels int32;
 * numcases = the kernel's 6th scalar (spatialDim for bce, numcases otherwise); norm / fullnorm: smoothL1 only            */
#define PZ_COST_BCE 0
#define PZ_COST_HINGE 1
#define PZ_COST_SMOOTH_L1 2
#define PZ_COST_L1_HINGE 3
int pz_cost_pointwise(int kind, const float *a, const void *b, const int32_t *labels, float *error, float *grad, float *grad2,
                      float *terms, size_t total, int numsamples, int numcases, float norm, float fullnorm, pz_stream_t stream);
/* PReluModule.prelu / preluBackwardData / preluBackwardParams (Cuda/Kernels/PRelu.py:14-133); shared = one slope for
 * all maps; pz_prelu_bwd_params leaves sum_{n, pixels} dy * x * (x <= 0) per map (the caller adds the maps up when the
 * slope is shared, as the reference does with matsum)                                                                 */
int pz_prelu_fwd(const float *x, const float *slopes, float *y, int n, int maps, int mapsize, int shared, pz_stream_t stream);
int pz_prelu_bwd_data(const float *dy, const float *slopes, const float *x, float *dx, int n, int maps, int mapsize, int shared,
                      pz_stream_t stream);
int pz_prelu_bwd_params(const float *x, const float *dy, float *per_map, int n, int maps, int mapsize, pz_stream_t stream);
/* PadModule.reflectpad / reflectpadBackward (Cuda/Kernels/Pad.py:45-230); 1-d tensors are planes of height 1 with
 * upad = bpad = 0. Pads must be >= 0 and smaller than the map. The backward pass gathers (deterministic).              */
int pz_reflectpad2d_fwd(const float *x, float *y, size_t planes, int inh, int inw, int upad, int bpad, int lpad, int rpad,
                        pz_stream_t stream);
int pz_reflectpad2d_bwd(const float *dy, float *dx, size_t planes, int inh, int inw, int upad, int bpad, int lpad, int rpad,
                        pz_stream_t stream);
/* UpsampleModule.upsample2d / 3d (+Backward), modes "nearest" (linear = 0) and "linear" (1) with integer scales
 * (Cuda/Kernels/Upsample.py:8-455); 2-d tensors are volumes of depth 1 with sd = 1. planes = batch * maps.            */
int pz_upsample_fwd(const float *x, float *y, size_t planes, int ind, int inh, int inw, int sd, int sh, int sw, int linear,
                    pz_stream_t stream);
int pz_upsample_bwd(const float *dy, float *dx, size_t planes, int ind, int inh, int inw, int sd, int sh, int sw, int linear,
                    pz_stream_t stream);
/* CTCModule.ctcLoss (Cuda/Kernels/CTC.py:232-270; Cost/CTC.py:23-30): probs (T, batch, vocab) softmax outputs, datalen[batch],
 * labels concatenated, offsets[batch + 1] their prefix sums. Per sample the caller also passes the extended-label positions
 * sorted by label (`order`, at element offset 2 * offsets[b] + b like the alphas' rows), the bounds of the runs of equal
 * labels (`seg_start`: nseg_b + 1 local bounds at seg_off[b] + b) and their labels (`seg_label` at seg_off[b]). Writes the
 * forward variables (T * (2 * offsets[batch] + batch) floats), nll[batch], grad (zero-initialised by the caller beyond
 * datalen) and adds sum nll to *error (fixed order; the reference atomicAdds).                                          */
int pz_ctc_loss(const float *probs, const int32_t *datalen, const int32_t *labels, const int32_t *offsets, const int32_t *order,
                const int32_t *seg_start, const int32_t *seg_label, const int32_t *seg_off, int T, int batch, int vocab, int blank,
                int max_positions, float *alphas, float *nll, float *grad, float *error, pz_stream_t stream);
/* EmbedModule.embed / embedBackwardParams (Cuda/Kernels/Embedder.py:10-88): word index -1 = padding (zero row, no update) */
int pz_embed_fwd(const int32_t *words, const float *vocab, float *out, size_t tokens, int embsize, pz_stream_t stream);
int pz_embed_bwd_params(const int32_t *words, const float *grad, float *vocab, float scale, size_t tokens, int embsize,
                        pz_stream_t stream);

/* ---- run-time compiled kernels: replaces Driver.compile (Cuda/Source/Core/Driver.c:501-515; NVRTC there, `hipcc --genco` on the
 *      reference's HIP backend, Hip/SourceModule.py:61-99) and Driver.Module / Function (Cuda/Source/Core/Module.c:258-290). For
 *      USER kernels — backend.SourceModule / ElementwiseKernel / ReductionKernel (Cuda/SourceModule.py:31-393) — every operator
 *      of the library itself is precompiled. pz_rtc_compile needs no device (hiprtc, target gfx950): *code is a malloc'ed code
 *      object (pz_rtc_free_code), `log` gets the compiler's output; a compilation error is PZ_ERR_INVALID. pz_function_launch
 *      takes the kernel arguments packed as the kernel's parameter list lays them out (natural alignment).               */
int pz_rtc_compile(const char *source, const char *name, const char *const *options, int noptions, void **code, size_t *code_bytes,
                   char *log, size_t log_bytes);
int pz_rtc_free_code(void *code);
int pz_module_load(const void *code, pz_module_t *module);
int pz_module_unload(pz_module_t module);
int pz_module_function(pz_module_t module, const char *name, void **function);
int pz_function_launch(void *function, const unsigned *grid, const unsigned *block, unsigned shared_bytes, const void *args,
                       size_t args_bytes, pz_stream_t stream);

/* ---- data-parallel exchange: replaces NodeInfo.{sumTensor,broadcastBuffer} (Grid.py:54-63,103-157: IPC star)
 *      with RCCL collectives over xGMI. One communicator per process (one process per GPU).             */
#define PZ_COMM_ID_BYTES 128
int pz_comm_unique_id(char id[PZ_COMM_ID_BYTES]);
int pz_comm_init_rank(pz_comm_t *comm, int nranks, const char id[PZ_COMM_ID_BYTES], int rank);
int pz_comm_destroy(pz_comm_t comm);
/* health: pz_comm_probe loads librccl (0 = usable) without creating anything — ranks vote on it before entering the
 * collective pz_comm_init_rank; pz_comm_async_error polls ncclCommGetAsyncError; pz_comm_wait_event waits on the host
 * for an event recorded behind collectives, polling the communicator, and aborts it after timeout_s (<= 0: no limit)
 * — the reference's star has neither (a dead child blocks the parent's queue.get() forever, Grid.py:117-121).      */
int pz_comm_probe(void);
/* ranks / own rank as RCCL reports them for the live communicator (ncclCommCount, ncclCommUserRank) */
int pz_comm_info(pz_comm_t comm, int *nranks, int *rank);
int pz_comm_async_error(pz_comm_t comm);
int pz_comm_wait_event(pz_comm_t comm, pz_event_t event, double timeout_s);
int pz_comm_allreduce_sum_f32(pz_comm_t comm, const float *send, float *recv, size_t count, pz_stream_t stream);
/* in-place sum over several ranges of ONE tensor (element offsets / counts relative to base) as one RCCL group: a completion-set
 * bucket of a flat gradient arena whose blocks are laid out in sorted-name order (Optimizers/Optimizer.py:66-68) is not
 * contiguous — the blocks of the layers that finish together are scattered over the arena                                  */
int pz_comm_allreduce_sum_f32_ranges(pz_comm_t comm, float *base, const size_t *offsets, const size_t *counts, int nranges,
                                     pz_stream_t stream);
int pz_comm_broadcast(pz_comm_t comm, void *buf, size_t nbytes, int root, pz_stream_t stream);

#ifdef __cplusplus
}
#endif

#endif /* PUZZLE_MI355_H */
