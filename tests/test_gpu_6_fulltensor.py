"""
Whole-tensor parity at BASELINE.json's own sizes (configs 2 and 4), the way the reference's per-op tests compare
(Cuda/Wrappers/CuDnn.py:29-80 conv2dTest: np.allclose on the full outdata, ingrad and wgrad):

* every distinct convolution of the reference's ResNet-50 at batch 256 — the 7x7/2 stem, the fifteen 1x1 layer shapes, the
  four 3x3 layer shapes — and config 2's Conv2D(64 -> 128, 3x3, 56x56, batch 128): the launches the bench times (`auto`
  algorithms, the lazy layer on), compared ELEMENT BY ELEMENT with the fp64 oracle over the WHOLE batch: forward,
  backward-data (one host GEMM per chunk of 32 images) and the filter gradient — i.e. the b256 split-K / slab-reduce plan,
  the XCD-remapped interior tiles and the k-sliced tail rounds themselves, not a sub-batch launch with a different plan.
* one full-depth ResNet-50 training-mode step (53 convolutions, every fusion on) against the CPU network oracle: logits,
  loss and all 161 parameter gradients.

Tolerances (per element, stated where they are applied):
  implicit GEMM / stem kernels   |err| <= 1e-5 s + 1e-4 |ref|,  s = the tensor's scale: max(1, rms(ref)) for y and dx (unit-
                                 variance inputs, filters ~ 1/sqrt(fan-in)); sqrt(N P Q) for dw, the standard deviation of a
                                 sum of N P Q unit-variance products — SURVEY.md section 8(c)'s bound on O(1) data
  Winograd F(4x4) fwd / bwd-data |err| <= 6e-5 max|ref|      (test_winograd_convolution's stated bound)
  Winograd F(2x2) bwd-filter     |err| <= 2e-5 max|ref| + 1e-5 sqrt(N P Q)
"""
import os

import numpy as np
import pytest

import cpu_ref as R
import cpu_net as N

pytestmark = pytest.mark.gpu

CHUNK = 32          # images per host GEMM (bounds the fp64 im2col: 32 x 3025 x 576 x 8 B = 446 MB for the widest 3x3 layer)


def gpu(bnd, ary):
	return bnd.GPUArray.toGpu(np.ascontiguousarray(ary))


def dev_randn(bnd, shape, seed):
	out = bnd.GPUArray.empty(shape, dtype=np.float32)
	bnd.RandomNumberGenerator(seed=seed).fillNormal(out, mean=0.0, stddev=1.0)
	return out


def worst(got, ref, atol, rtol):
	"""max over elements of |err| / (atol + rtol |ref|), chunked so that no fp64 copy of a whole tensor is made"""
	assert got.shape == ref.shape, (got.shape, ref.shape)
	g, r = got.reshape(-1), ref.reshape(-1)
	top, step = 0.0, 1 << 24
	for i in range(0, g.size, step):
		gi, ri = g[i:i + step].astype(np.float64), r[i:i + step].astype(np.float64)
		assert np.isfinite(gi).all(), "non-finite values in the device result"
		top = max(top, float((np.abs(gi - ri) / (atol + rtol * np.abs(ri))).max()))
	return top


def rms(a):
	return float(np.sqrt(np.mean(np.square(a, dtype=np.float64))))


# (C, H, W) -> (K, size, stride, pad): Models/Nets/ResNet.py:23-121 on 224x224 inputs (55x55 stage-2 maps)
R50_CONVS = [
	((3, 224, 224), (64, 7, 2, 3)), ((64, 55, 55), (64, 1, 1, 0)), ((64, 55, 55), (64, 3, 1, 1)), ((64, 55, 55), (256, 1, 1, 0)),
	((256, 55, 55), (64, 1, 1, 0)), ((256, 55, 55), (128, 1, 2, 0)), ((128, 28, 28), (128, 3, 1, 1)), ((128, 28, 28), (512, 1, 1, 0)),
	((256, 55, 55), (512, 1, 2, 0)), ((512, 28, 28), (128, 1, 1, 0)), ((512, 28, 28), (256, 1, 2, 0)), ((256, 14, 14), (256, 3, 1, 1)),
	((256, 14, 14), (1024, 1, 1, 0)), ((512, 28, 28), (1024, 1, 2, 0)), ((1024, 14, 14), (256, 1, 1, 0)),
	((1024, 14, 14), (512, 1, 2, 0)), ((512, 7, 7), (512, 3, 1, 1)), ((512, 7, 7), (2048, 1, 1, 0)),
	((1024, 14, 14), (2048, 1, 2, 0)), ((2048, 7, 7), (512, 1, 1, 0)),
]
CASES = [(256, ) + l for l in R50_CONVS] + [(128, (64, 56, 56), (128, 3, 1, 1))]          # + config 2


@pytest.mark.parametrize("case", CASES, ids=lambda c: "b%d_%dx%dx%d_to_%d_k%d_s%d" % ((c[0], ) + c[1] + c[2][:3]))
def test_conv_whole_tensors_vs_fp64_oracle(bnd, case):
	from puzzlelib_amd import lazy, lib
	from puzzlelib_amd.surface import bound
	Dnn, G = bound().Dnn, bnd.GPUArray
	lazy.enabled, lazy.disabled = True, set()

	n, (c, h, w), (k, size, stride, pad) = case
	okw = dict(stride=(stride, stride), pad=(pad, pad), dilation=(1, 1), groups=1)
	algos = Dnn.ConvFwdAlgo.auto, Dnn.ConvBwdDataAlgo.auto, Dnn.ConvBwdFilterAlgo.auto
	desc = bnd.dnn.convDesc((n, c, h, w), (k, c, size, size), stride, pad, 1, 1)
	family = [bnd.dnn.convAlgoUsed(desc, which, -1) for which in (lib.CONV_FWD, lib.CONV_BWD_DATA, lib.CONV_BWD_FILTER)]
	if size == 3:
		assert family == [3, 3, 3], "3x3 layers run on the Winograd kernels under auto"

	x = dev_randn(bnd, (n, c, h, w), 31)
	wt = gpu(bnd, (np.random.RandomState(32).randn(k, c, size, size) / np.sqrt(c * size * size)).astype(np.float32))

	# the launches the training step makes, back to back, before anything is read back
	y = Dnn.convNd(x, wt, None, okw["stride"], okw["pad"], okw["dilation"], 1, algos[0])
	p, q = y.shape[2:]
	dy = dev_randn(bnd, (n, k, p, q), 33)
	dx = Dnn.convNdBackwardData(dy, wt, x, okw["stride"], okw["pad"], okw["dilation"], 1, algos[1])
	dw = G.zeros(wt.shape, dtype=np.float32)
	Dnn.convNdBackwardParams(x, dy, wt, None, okw["stride"], okw["pad"], okw["dilation"], 1, dw, None, 1.0, 1.0, algos[2])
	dw_again = G.zeros(wt.shape, dtype=np.float32)
	Dnn.convNdBackwardParams(x, dy, wt, None, okw["stride"], okw["pad"], okw["dilation"], 1, dw_again, None, 1.0, 1.0, algos[2])

	xh, wh, dyh = x.get(), wt.get(), dy.get()
	yh, dxh, dwh = y.get(), dx.get(), dw.get()
	assert np.array_equal(dwh, dw_again.get()), "the filter gradient's slab reduction is not repeatable"
	del x, dy, y, dx

	images = np.arange(n)                                  # y and dx are compared over the whole batch
	npix = n * p * q

	def bounds(which, ref):
		if family[which] == 3 and which < 2:
			return 6e-5 * max(1.0, float(np.abs(ref).max())), 0.0
		return 1e-5 * max(1.0, rms(ref)), 1e-4

	ratios = {}
	dw_ref = np.zeros(wh.shape, np.float64)
	for c0 in range(0, n, CHUNK):
		sel = slice(c0, min(n, c0 + CHUNK))
		dw_ref += R.conv2d_bwd_filter(xh[sel], dyh[sel], wh.shape, withbias=False, acc=np.float64, **okw)
		pick = images[(images >= sel.start) & (images < sel.stop)]
		if len(pick) == 0:
			continue
		ref = R.conv2d_fwd(xh[pick], wh, None, acc=np.float64, **okw)
		atol, rtol = bounds(0, ref)
		ratios["forward"] = max(ratios.get("forward", 0.0), worst(yh[pick], ref, atol, rtol))
		ref = R.conv2d_bwd_data(dyh[pick], wh, (len(pick), c, h, w), acc=np.float64, **okw)
		atol, rtol = bounds(1, ref)
		ratios["backward-data"] = max(ratios.get("backward-data", 0.0), worst(dxh[pick], ref, atol, rtol))

	if family[2] == 3:
		atol, rtol = 2e-5 * float(np.abs(dw_ref).max()) + 1e-5 * np.sqrt(npix), 0.0
	else:
		atol, rtol = 1e-5 * np.sqrt(npix), 1e-4
	ratios["backward-filter"] = worst(dwh, dw_ref, atol, rtol)

	print("whole-tensor error / bound:", {k_: round(v, 3) for k_, v in ratios.items()}, "kernel families", family)
	for name, ratio in ratios.items():
		assert ratio <= 1.0, "%s: an element is %.2f x its bound (families %s)" % (name, ratio, family)


# The launches of the training step that read a DESCRIBED operand (the lazy layer's folds) — at the same sizes, straight through the
# C ABI, element by element against the fp64 oracle applied to the host-evaluated operand (until round 5
# these were tied to the oracle at batch 256 only through fused == literal bit identity):
#   pz_conv2d_fwd_xbn            y  = conv(relu(a x + b), w)                      bottleneck blocks' last 1x1 layer, forward
#   pz_conv2d_bwd_filter_xbn     dw = relu(a x + b) (x) (A dy + B z + C)          ... its filter gradient, both operands described
#   pz_conv2d_bwd_data_bn        dx = conv^T(A dy + B z + C, w)                   ... and every 1x1 layer's backward-data behind a BatchNorm
#   pz_conv2d_bwd_filter_bn      dw = x (x) (A dy + B z + C)
# + the compact stride-2 backward-data (a stride-1 problem on the output grid).
FOLDED = [
	(256, (64, 55, 55), 256), (256, (128, 28, 28), 512), (256, (256, 14, 14), 1024),      # xbn + gradient-side fold (128x128 tiles)
	(256, (256, 55, 55), 64), (256, (1024, 14, 14), 256), (256, (2048, 7, 7), 512),       # 64x256 tiles / two-tiles-ahead reductions: no xbn
	(256, (256, 28, 28), 128),                                                            # = the compact grid of (256,55,55) -> 128 /2
]


@pytest.mark.parametrize("case", FOLDED, ids=lambda c: "b%d_%dx%dx%d_to_%d" % ((c[0], ) + c[1] + (c[2], )))
def test_folded_launches_whole_tensors_vs_fp64_oracle(bnd, case):
	import ctypes
	from puzzlelib_amd import lib
	G = bnd.GPUArray
	n, (c, h, w), k = case
	desc = bnd.dnn.convDesc((n, c, h, w), (k, c, 1, 1), 1, 0, 1, 1)
	algo = lib.CONV_ALGO_AUTO
	rng = np.random.RandomState(41)

	x, z, dy = dev_randn(bnd, (n, c, h, w), 42), dev_randn(bnd, (n, k, h, w), 43), dev_randn(bnd, (n, k, h, w), 44)
	wh = (rng.randn(k, c, 1, 1) / np.sqrt(c)).astype(np.float32)
	xco = np.stack([0.5 + rng.rand(c), 0.3 * rng.randn(c)], axis=1).astype(np.float32)                    # {a, b} per input channel
	gco = np.concatenate([np.stack([0.5 + rng.rand(k), 0.2 * rng.randn(k), 0.1 * rng.randn(k)], axis=1), np.zeros((k, 1))], axis=1).astype(np.float32)
	wt, gxco, ggco = gpu(bnd, wh), gpu(bnd, xco), gpu(bnd, gco)

	def ws(which):
		nbytes = bnd.dnn.convGeometry(desc, which, algo)[2]
		return (G.empty((max(nbytes, 4), ), dtype=np.uint8), nbytes)

	y = G.empty((n, k, h, w), dtype=np.float32)
	dx = G.empty((n, c, h, w), dtype=np.float32)
	dw_bn, dw_xbn = G.zeros(wh.shape, dtype=np.float32), G.zeros(wh.shape, dtype=np.float32)
	takes_xbn = bnd.dnn.xbnSupported(desc, lib.CONV_FWD, algo)
	assert takes_xbn == (k >= 128 and c <= 256), "128-row tiles below the two-tiles-ahead threshold take the forward gather"

	wsf, nf = ws(lib.CONV_FWD)
	if takes_xbn:
		lib.pz_conv2d_fwd_xbn(ctypes.byref(desc), x.rptr, gxco.rptr, 1, wt.rptr, None, None, y.optr, None, algo, wsf.optr, nf, None)
	wsd, nd = ws(lib.CONV_BWD_DATA)
	lib.pz_conv2d_bwd_data_bn(ctypes.byref(desc), dy.rptr, z.rptr, ggco.rptr, wt.rptr, dx.optr, algo, wsd.optr, nd, None)
	wsw, nw = ws(lib.CONV_BWD_FILTER)
	lib.pz_conv2d_bwd_filter_bn(ctypes.byref(desc), x.rptr, dy.rptr, z.rptr, ggco.rptr, dw_bn.optr, 1.0, 0.0, algo, wsw.optr, nw, None)
	if bnd.dnn.xbnSupported(desc, lib.CONV_BWD_FILTER, algo):
		lib.pz_conv2d_bwd_filter_xbn(ctypes.byref(desc), x.rptr, gxco.rptr, 1, dy.rptr, z.rptr, ggco.rptr, dw_xbn.optr, 1.0, 0.0, algo,
									 wsw.optr, nw, None)

	xh, zh, dyh = x.get(), z.get(), dy.get()
	yh, dxh, dwbh, dwxh = y.get(), dx.get(), dw_bn.get(), dw_xbn.get()
	del x, z, dy, y, dx

	okw = dict(stride=(1, 1), pad=(0, 0), dilation=(1, 1), groups=1)
	npix = n * h * w
	ratios = {}
	dw_bn_ref, dw_xbn_ref = np.zeros(wh.shape, np.float64), np.zeros(wh.shape, np.float64)
	for c0 in range(0, n, CHUNK):
		sel = slice(c0, min(n, c0 + CHUNK))
		# the described operands, evaluated on the host in fp64 from the device's own inputs
		act = np.maximum(xco[None, :, 0, None, None].astype(np.float64) * xh[sel] + xco[None, :, 1, None, None], 0.0)
		g = gco[None, :, 0, None, None].astype(np.float64) * dyh[sel] + gco[None, :, 1, None, None] * zh[sel] + gco[None, :, 2, None, None]
		if takes_xbn:
			ref = R.conv2d_fwd(act, wh, None, acc=np.float64, **okw)
			ratios["forward (xbn)"] = max(ratios.get("forward (xbn)", 0.0), worst(yh[sel], ref, 1e-5 * max(1.0, rms(ref)), 1e-4))
		ref = R.conv2d_bwd_data(g, wh, (sel.stop - sel.start, c, h, w), acc=np.float64, **okw)
		ratios["backward-data (bn)"] = max(ratios.get("backward-data (bn)", 0.0), worst(dxh[sel], ref, 1e-5 * max(1.0, rms(ref)), 1e-4))
		dw_bn_ref += R.conv2d_bwd_filter(xh[sel].astype(np.float64), g, wh.shape, withbias=False, acc=np.float64, **okw)
		dw_xbn_ref += R.conv2d_bwd_filter(act, g, wh.shape, withbias=False, acc=np.float64, **okw)
	ratios["backward-filter (bn)"] = worst(dwbh, dw_bn_ref, 1e-5 * np.sqrt(npix), 1e-4)
	if dwxh.any():
		ratios["backward-filter (xbn + bn)"] = worst(dwxh, dw_xbn_ref, 1e-5 * np.sqrt(npix), 1e-4)
	print("whole-tensor error / bound:", {k_: round(v, 3) for k_, v in ratios.items()})
	assert len(ratios) == (4 if takes_xbn else 3)
	for name, ratio in ratios.items():
		assert ratio <= 1.0, "%s: an element is %.2f x its bound" % (name, ratio)


# Round 6: the backward-data launch of a pointwise layer whose input was y = relu(a z + b) (the BatchNorm in FRONT of it, never
# written) also sums, in its epilogue, the statistics of that BatchNorm's backward over the GATED gradient q = dx * (y > 0):
# {sum q, sum q (z - mean)} per channel (pz_conv2d_bwd_data_bnstats; the reference's formulas: Cuda/Wrappers/CuDnnNorm.py:55-63,
# the ReLU's derivative by the output sign: Cuda/Kernels/ElementWise.py:119-172). (n, (k, h, w), c): dy (n, k, h, w) -> dx (n, c, h, w).
BNSTATS = [
	(3, (40, 13, 9), 24),            # ragged everything: 24 < one tile row band, 351 pixels = strips that straddle images, last strip short
	(5, (16, 7, 7), 130),            # two row tiles, the second nearly empty
	(256, (256, 55, 55), 64),        # the 64-row tile (HBM-bound launches of stage 2's tails)
	(256, (512, 28, 28), 128), (256, (1024, 14, 14), 256), (256, (2048, 7, 7), 512),       # 128-row tiles, k-sliced tail rounds
	# 3x3 layers (a fourth entry = the filter size): the F(4x4) Winograd kernel's backward-data epilogue — the BatchNorms in front of
	# the bottlenecks' 3x3 layers; 55 and 7 are no multiples of 4 (edge tiles, word-by-word rows), 13 x 9 leaves a ragged last tile block
	(256, (64, 55, 55), 64, 3), (256, (128, 28, 28), 128, 3), (256, (256, 14, 14), 256, 3), (256, (512, 7, 7), 512, 3), (40, (32, 13, 9), 64, 3),
]


# The epilogue behind this test was written in round 6, when no device was available to the build: it is
# compiled, its glue and this test's own arithmetic were run on the emulated C ABI, but the kernel code has never executed. It is
# opt-in in the product (PUZZLE_MI355_DGRAD_STATS=1) and so is its device test: tools/r06_validate.sh runs it first thing.
@pytest.mark.skipif(os.environ.get("PUZZLE_MI355_DGRAD_STATS", "0") != "1" and os.environ.get("PUZZLE_MI355_UNVERIFIED", "0") != "1",
					reason="opt-in feature (PUZZLE_MI355_DGRAD_STATS=1), device code not yet run on an MI355X: tools/r06_validate.sh")
@pytest.mark.parametrize("fold", [False, True], ids=["plain", "bn_fold"])
@pytest.mark.parametrize("case", BNSTATS, ids=lambda c: "b%d_%dx%dx%d_to_%d%s" % ((c[0], ) + c[1] + (c[2], "_3x3" if len(c) > 3 else "")))
def test_backward_data_epilogue_statistics_vs_fp64_oracle(bnd, case, fold):
	"""dx bit-identical to the launch without the epilogue sums; the merged sums against fp64 sums over the device's own dx
	(|err| <= 2e-6 * sum |terms|: fp32 strip sums of <= 64 terms each, merged in fp64); pz_bn_bwd_gate_from_partials against
	pz_bn_bwd_gate (the two-pass form) on the same operands: the input gradient to 2e-5 of its top, the parameter gradients to
	1e-5 of the sums of magnitudes behind them."""
	import ctypes
	from puzzlelib_amd import lib
	G = bnd.GPUArray
	n, (k, h, w), c = case[:3]
	r = case[3] if len(case) > 3 else 1
	desc = bnd.dnn.convDesc((n, c, h, w), (k, c, r, r), 1, r // 2, 1, 1)
	algo = lib.CONV_ALGO_AUTO
	rng = np.random.RandomState(51)
	hw = h * w

	dy, bz, z = dev_randn(bnd, (n, k, h, w), 52), dev_randn(bnd, (n, k, h, w), 53), dev_randn(bnd, (n, c, h, w), 54)
	wh = (rng.randn(k, c, r, r) / np.sqrt(k * r * r)).astype(np.float32)
	ab = np.stack([0.5 + rng.rand(c), 0.4 * rng.randn(c)], axis=1).astype(np.float32)       # the forward's {a, b} of the BatchNorm in front
	mean = (0.3 * rng.randn(c)).astype(np.float32)
	rstd = (0.5 + rng.rand(c)).astype(np.float32)
	scale = (ab[:, 0] / rstd).astype(np.float32)
	gco = np.concatenate([np.stack([0.5 + rng.rand(k), 0.2 * rng.randn(k), 0.1 * rng.randn(k)], axis=1), np.zeros((k, 1))], axis=1).astype(np.float32)
	wt, gab, gmean, grstd, gscale, ggco = (gpu(bnd, a) for a in (wh, ab, mean, rstd, scale, gco))

	size = ctypes.c_size_t(0)
	lib.pz_conv2d_bwd_data_bnstats_bytes(ctypes.byref(desc), algo, ctypes.byref(size))
	if r == 3 and size.value == 0:
		pytest.skip("this 3x3 layer runs on the F(2x2) Winograd kernel, which has no statistics epilogue")
	assert size.value >= 16 * c, "every pointwise stride-1 layer has the statistics epilogue"
	parts = G.empty((size.value, ), dtype=np.uint8)
	nbytes, foldable = bnd.dnn.convGeometry(desc, lib.CONV_BWD_DATA, algo)[2], bnd.dnn.convGeometry(desc, lib.CONV_BWD_DATA, algo)[4]
	if fold and not foldable:
		pytest.skip("this layer's gather cannot fold a BatchNorm backward (pz_conv2d_bn_fold_supported)")
	ws = G.empty((max(nbytes, 4), ), dtype=np.uint8)

	dx0, dx1 = G.empty((n, c, h, w), dtype=np.float32), G.empty((n, c, h, w), dtype=np.float32)
	if fold:
		lib.pz_conv2d_bwd_data_bn(ctypes.byref(desc), dy.rptr, bz.rptr, ggco.rptr, wt.rptr, dx0.optr, algo, ws.optr, nbytes, None)
	else:
		lib.pz_conv2d_bwd_data(ctypes.byref(desc), dy.rptr, wt.rptr, dx0.optr, algo, ws.optr, nbytes, None)
	lib.pz_conv2d_bwd_data_bnstats(
		ctypes.byref(desc), dy.rptr, bz.rptr if fold else None, ggco.rptr if fold else None, wt.rptr, dx1.optr, z.rptr, gab.rptr,
		gmean.rptr, parts.optr, algo, ws.optr, nbytes, None
	)
	dxh, zh = dx1.get(), z.get()
	assert np.array_equal(dx0.get(), dxh), "the epilogue sums must not change what is stored"

	# the merged sums: 2 c doubles at the head of the partials
	merged = parts.get()[:16 * c].view(np.float64).reshape(c, 2)
	gate = (ab[None, :, 0, None, None] * zh + ab[None, :, 1, None, None]) > 0                # fp32 fma on the device; ties are measure-zero here
	q = np.where(gate, dxh, 0).astype(np.float64)
	d = zh.astype(np.float64) - mean[None, :, None, None]
	s1, s2 = q.sum(axis=(0, 2, 3)), (q * d).sum(axis=(0, 2, 3))
	m1, m2 = np.abs(q).sum(axis=(0, 2, 3)), np.abs(q * d).sum(axis=(0, 2, 3))
	flips = int((np.abs(ab[None, :, 0, None, None].astype(np.float64) * zh + ab[None, :, 1, None, None]) < 1e-6).sum())
	tol = lambda m: 2e-6 * m + 1e-6 + (1e-2 * flips)
	assert (np.abs(merged[:, 0] - s1) <= tol(m1)).all(), "sum q: worst %.3e of %.3e" % (np.abs(merged[:, 0] - s1).max(), m1.max())
	assert (np.abs(merged[:, 1] - s2) <= tol(m2)).all(), "sum q (z - mean): worst %.3e of %.3e" % (np.abs(merged[:, 1] - s2).max(), m2.max())

	# the BatchNorm backward from these partials against the two-pass form
	size = ctypes.c_size_t(0)
	lib.pz_bn_workspace_bytes(n, c, hw, ctypes.byref(size))
	bws = G.empty((size.value, ), dtype=np.uint8)
	outs = []
	for one_pass in (False, True):
		gin, ds, db = G.empty((n, c, h, w), dtype=np.float32), G.empty((c, ), dtype=np.float32), G.empty((c, ), dtype=np.float32)
		if one_pass:
			lib.pz_bn_bwd_gate_from_partials(z.rptr, dx1.rptr, gin.optr, n, c, hw, gscale.rptr, gmean.rptr, grstd.rptr, ds.optr, db.optr,
											 gab.rptr, parts.rptr, None)
		else:
			lib.pz_bn_bwd_gate(z.rptr, dx1.rptr, gin.optr, n, c, hw, gscale.rptr, gmean.rptr, grstd.rptr, ds.optr, db.optr, gab.rptr,
							   bws.optr, size.value, None)
		outs.append((gin.get(), ds.get(), db.get()))
	(g2, ds2, db2), (g1, ds1, db1) = outs
	top = float(np.abs(g2).max())
	assert float(np.abs(g1 - g2).max()) <= 2e-5 * top, "input gradient differs by %.3e of its top" % (float(np.abs(g1 - g2).max()) / top)
	assert (np.abs(db1 - db2) <= 1e-5 * m1 + 1e-6).all() and (np.abs(ds1 - ds2) <= (1e-5 * m2 + 1e-6) * rstd).all()
	print("epilogue statistics: worst |sum err| / sum|terms| = %.2e / %.2e, one-pass BatchNorm backward within %.1e of the two-pass form" % (
		float((np.abs(merged[:, 0] - s1) / (m1 + 1e-30)).max()), float((np.abs(merged[:, 1] - s2) / (m2 + 1e-30)).max()),
		float(np.abs(g1 - g2).max()) / top))


def adoptDeviceGatesNested(cnet, layers, spec, prefix=""):
	"""test_gpu_5_nets.adoptDeviceGates for nested specs: the oracle's backward gates with the ReLU outputs / max-pool
	operands the DEVICE produced (cache keys as oracle/cpu_net.py builds them: "<index>", "<index>.b.<index>", ...). Returns
	(flipped gates, worst forward mismatch relative to the tensor's top)."""
	flips, mism = 0, 0.0
	for idx, layer in enumerate(layers):
		key = "%s%d" % (prefix, idx)
		if layer.kind == "act":
			y = layer.y.get()
			ref = cnet.cache[key]
			mism = max(mism, float(np.abs(y - ref).max()) / max(1.0, float(np.abs(ref).max())))
			flips += int(((y > 0) != (ref > 0)).sum())
			cnet.cache[key] = y
		elif layer.kind == "pool" and spec[idx][0] == "maxpool":
			cnet.cache[key] = (layer.x.get(), layer.y.get())
		elif layer.kind == "resid":
			for tag, branch, sub in (("b", layer.branches[0], spec[idx][1]), ("s", layer.branches[1], spec[idx][2])):
				if len(sub) > 0:
					f, m = adoptDeviceGatesNested(cnet, branch, sub, "%s.%s." % (key, tag))
					flips, mism = flips + f, max(mism, m)
	return flips, mism


@pytest.mark.parametrize("batch", [16])
def test_resnet50_full_depth_training_step_vs_oracle(bnd, batch):
	"""Config 4's network at full depth, training mode (batch statistics in all 53 batch norms), `auto` algorithms, every
	fusion of the lazy layer on and the convolution epilogues' statistics in use (second pass of the adaptive policy):
	logits, loss, cross-entropy gradient and ALL 161 parameter gradients against oracle/cpu_net.py on the same inputs and
	initial values. The oracle's backward gates with the device's ReLU / max-pool decisions (a pre-activation within rounding
	of zero may round to either side; see adoptDeviceGates in test_gpu_5_nets.py). Bound, per parameter gradient: relative L2
	error against the oracle with fp64 sums inside every operator <= max(5e-5, 1.3 x the distance of the fp32-summing oracle from
	that same yardstick) — 53 layers deep the order of fp32 sums alone moves the last block's gradients by ~1e-4 (two numpy
	runs differing only in accumulation type), so a fixed 5e-5 would test numpy's summation order, not the kernels; median over
	the 161 gradients <= 5e-5; no gradient above 1.6e-4 in absolute terms; every element within 1e-3 of its gradient's top."""
	from puzzlelib_amd import nets, optim, lazy
	from puzzlelib_amd.surface import bound
	gpuarray = bound().gpuarray
	lazy.enabled, lazy.disabled = True, set()

	np.random.seed(4321)
	net = nets.loadResNet(None, "50", actInplace=True, initscheme="he")
	net.layers.pop()                                       # the trailing SoftMax: training runs on raw scores
	spec = nets.resnet50_spec(softmax=False)
	optimizer = optim.Adam(alpha=1e-3)
	optimizer.setupOn(net, useGlobalState=True)
	cost = optim.CrossEntropy()
	rng = np.random.RandomState(99)
	data = rng.randn(batch, 3, 224, 224).astype(np.float32)
	labels = rng.randint(0, 1000, size=(batch, )).astype(np.int32)
	gdata, glabels = gpuarray.to_gpu(data), gpuarray.to_gpu(labels)
	params = {name: p.data.get() for name, p in net.namedParams().items()}
	assert len(params) == 161
	net.trainMode()

	def devicePass():
		for layer in net.walk():
			if layer.kind == "bn":
				layer.cfg["passes"] = 0
				layer.attrs["mean"].fill(0.0)
				layer.attrs["var"].fill(1.0)
		lazy.counters.clear()
		logits = net(gdata)
		grad = cost(logits, glabels, queryError=False)
		optimizer.zeroGradParams()
		net.backward(grad, updGrad=False)
		return logits, grad

	devicePass()                                           # the adaptive policy learns which convolutions feed a batch norm
	net.reset()
	logits, grad = devicePass()
	taken = dict(lazy.counters)
	assert taken.get("bn_apply_add", 0) == 16 and taken.get("dgrad_bn_fold", 0) >= 1, taken

	def oracle(acc):
		attrs = {name: (np.zeros(a.shape, np.float32) if name.endswith(".mean") else np.ones(a.shape, np.float32))
				 for name, a in net.namedAttrs().items()}      # running statistics as devicePass() resets them
		cnet = N.CpuNet(spec, params, attrs, acc=acc)
		cnet.train = True
		ref_logits = cnet.forward(data)
		err_ref, grad_ref = R.cross_entropy(ref_logits, labels)
		flips, mism = adoptDeviceGatesNested(cnet, net.layers, spec)
		cnet.zero_grads()
		cnet.backward(grad_ref)
		return cnet, ref_logits, err_ref, grad_ref, flips, mism

	# the yardstick: every operator correctly rounded from fp64 sums ...
	cnet, ref_logits, err_ref, grad_ref, flips, mism = oracle(np.float64)
	got_logits = logits.get()
	scale = float(np.abs(ref_logits).max())
	assert np.abs(got_logits - ref_logits).max() <= 2e-4 * scale, "logits: %.3e of their top" % (np.abs(got_logits - ref_logits).max() / scale)
	assert np.isclose(float(cost.devErr.get()), err_ref, rtol=1e-4), (float(cost.devErr.get()), err_ref)
	assert np.abs(grad.get() - grad_ref).max() <= 1e-6 + 1e-3 * np.abs(grad_ref).max()
	assert mism <= 2e-4, "a forward activation is %.3e of its top away from the oracle's" % mism
	# ... and what fp32 summation itself is worth 53 layers deep: the oracle with fp32 sums (numpy's order) against the same
	# yardstick. Two correctly working fp32 implementations are this far apart (measured: 1.15e-4 on the last block's filter
	# gradients at batch 16, median 2e-5 — the device, Winograd or implicit GEMM alike, sits at 1.1e-4 / 2e-5).
	c32 = oracle(np.float32)[0]

	def rel(a, b):
		return np.linalg.norm((a - b).astype(np.float64)) / (np.linalg.norm(b.astype(np.float64)) + 1e-30)

	# (the opt-in split math modes, PUZZLE_MI355_MATH=split6 / split9, sum 6 / 9 partial products per fp32 product: measured up to
	# 7.3e-5 on the first layer's BatchNorm bias gradient, whose value collects every layer's rounding — their floor is 1e-4)
	floor = 5e-5 if bnd.dnn.convMath == "f32" else 1e-4
	rels = []
	for name, p in net.namedParams().items():
		ref, got = cnet.grads[name], p.grad.get()
		assert np.isfinite(got).all(), name
		dev, own = rel(got, ref), rel(c32.grads[name], ref)
		rels.append((dev, own, name))
		# (round 5: the factor was 2 — a kernel twice as far from the yardstick as numpy's fp32 sums would have passed. Measured: 14 of
		# the 161 gradients lie above the floor, the device is at most 1.04 x the fp32-summing oracle's distance on them, worst 1.14e-4.)
		assert dev <= max(floor, 1.3 * own), "grad %s: relative L2 error %.3e (the fp32 oracle's own: %.3e)" % (name, dev, own)
		assert dev <= (1.6e-4 if bnd.dnn.convMath == "f32" else 2.5e-4), "grad %s: relative L2 error %.3e above the absolute cap" % (name, dev)
		assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max() + 1e-9, "element of grad %s" % name
	rels.sort(reverse=True)
	above = [(d / o, d, o, nm) for d, o, nm in rels if d > floor]
	print("gradients above the floor: %d; worst device / fp32-oracle ratio %s" % (len(above), max(above)[:3] if above else None))
	median = rels[len(rels) // 2][0]
	print("ResNet-50 b%d training step vs fp64-summing oracle: worst relative L2 gradient error %.3e (%s; fp32-summing oracle %.3e), "
		  "median %.3e, %d flipped gates adopted, worst forward mismatch %.2e" % (batch, rels[0][0], rels[0][2], rels[0][1], median, flips, mism))
	assert median <= 5e-5, "median relative L2 gradient error %.3e" % median
	net.reset()
