"""
Full-size and edge-case checks of the hot path on the GPU (⑶ of the task: BASELINE.json's full sizes through
size-independent properties; the reference's own edge cases).

* Config 2 of BASELINE.json — Conv2D 3x3, 64 -> 128 maps, 56x56, batch 128, fp32 — at full size: linearity of the
  forward pass, the adjoint identities <dy, conv(x; w)> = <bwd_data(dy; w), x> = <w, bwd_filter(x, dy)> (backward-data and
  backward-filter ARE the adjoints of forward in x and in w), and an oracle spot check on two images of the batch.
* ResNet-50 b256 stage shapes: batch-norm output statistics (per-channel mean = bias, variance = scale^2), residual
  kernels, pooling round trip (max-pool backward puts each gradient where the maximum was).
* Determinism: the same training step from the same state gives the same bits (no atomics anywhere on the path).
* Degenerate shapes the reference's tests touch: 1x1 maps, single image / channel, windows covering the whole plane,
  zero-size element-wise launches, batch-norm of a constant tensor.
"""
import numpy as np
import pytest

import cpu_ref as R
from conftest import assert_close

pytestmark = pytest.mark.gpu


def gpu(bnd, ary):
	return bnd.GPUArray.toGpu(np.ascontiguousarray(ary))


def dev_randn(bnd, shape, seed):
	"""Normal tensor generated on the device (full-size tensors would take seconds to create on the host)."""
	out = bnd.GPUArray.empty(shape, dtype=np.float32)
	rng = bnd.RandomNumberGenerator(seed=seed) if hasattr(bnd, "RandomNumberGenerator") else bnd.globalRng
	rng.fillNormal(out, mean=0.0, stddev=1.0)
	return out


@pytest.mark.parametrize("algo", [3, 5], ids=["winograd", "implicit-gemm"])        # `auto` takes Winograd for this layer
def test_config2_conv_full_size_properties(bnd, algo):
	n, c, k, h, w = 128, 64, 128, 56, 56
	okw = dict(stride=(1, 1), pad=(1, 1), dilation=(1, 1), groups=1)
	kw = dict(okw, algo=algo)
	dot = bnd.blas.dot
	desc = bnd.dnn.convDesc((n, c, h, w), (k, c, 3, 3), 1, 1, 1, 1)
	assert all(bnd.dnn.convAlgoUsed(desc, which, algo) == algo for which in (0, 1, 2))
	assert all(bnd.dnn.convAlgoUsed(desc, which, -1) == 3 for which in (0, 1, 2))

	x1, x2 = dev_randn(bnd, (n, c, h, w), 1), dev_randn(bnd, (n, c, h, w), 2)
	dy = dev_randn(bnd, (n, k, h, w), 3)
	wt = gpu(bnd, (np.random.RandomState(4).randn(k, c, 3, 3) / np.sqrt(c * 9)).astype(np.float32))

	y1, y2 = bnd.dnn.convNd(x1, wt, None, **kw), bnd.dnn.convNd(x2, wt, None, **kw)
	assert y1.shape == (n, k, h, w)

	# linearity in x: conv(2 x1 - 3 x2) == 2 conv(x1) - 3 conv(x2)
	xs = bnd.GPUArray.empty(x1.shape, dtype=np.float32)
	bnd.addKer(np.float32)(xs, x1, 2.0, x2, -3.0)
	ys = bnd.dnn.convNd(xs, wt, None, **kw)
	comb = bnd.GPUArray.empty(y1.shape, dtype=np.float32)
	bnd.addKer(np.float32)(comb, y1, 2.0, y2, -3.0)
	diff = bnd.GPUArray.empty(y1.shape, dtype=np.float32)
	bnd.addKer(np.float32)(diff, ys, 1.0, comb, -1.0)
	# three roundings of the same product meet here. Implicit GEMM: 2e-4 absolute on results of magnitude <= ~20; `auto` resolves
	# this layer to Winograd F(4x4, 3x3), whose stated tolerance is 6e-5 of the result's scale (test_winograd_convolution)
	lin_tol = 2e-4 if algo == 5 else 6e-5 * max(1.0, float(comb.max().get()), -float(comb.min().get()))
	assert float(diff.max().get()) < lin_tol and float(diff.min().get()) > -lin_tol, "forward is not linear in x"

	# adjoint identities (sums of 51 M products of O(1) terms: relative tolerance on the scalar)
	lhs = dot(dy.ravel(), y1.ravel())
	dx = bnd.dnn.convNdBackwardData(dy, wt, data=x1, **kw)
	mid = dot(dx.ravel(), x1.ravel())
	dw = bnd.dnn.convNdBackwardParams(x1, dy, wt, **kw)
	rhs = dot(dw.ravel(), wt.ravel())
	scale = abs(lhs) + np.sqrt(float(dy.size))
	assert abs(lhs - mid) < 2e-4 * scale, "backward-data is not the adjoint of forward: %r vs %r" % (lhs, mid)
	assert abs(lhs - rhs) < 2e-4 * scale, "backward-filter is not the adjoint of forward: %r vs %r" % (lhs, rhs)

	# oracle spot check: images 0 and 127 of the batch
	xh, wh, dyh = x1.get(), wt.get(), dy.get()
	for img in (0, n - 1):
		ref = R.conv2d_fwd(xh[img:img + 1], wh, None, acc=np.float64, **okw)
		assert_close(y1.get()[img:img + 1], ref, atol=1e-4, rtol=1e-4, what="forward, image %d" % img)
		ref = R.conv2d_bwd_data(dyh[img:img + 1], wh, (1, c, h, w), acc=np.float64, **okw)
		assert_close(dx.get()[img:img + 1], ref, atol=1e-4, rtol=1e-4, what="backward-data, image %d" % img)

	# backward-filter against float64 sums over a 4-image sub-batch
	dw4 = bnd.dnn.convNdBackwardParams(gpu(bnd, xh[:4]), gpu(bnd, dyh[:4]), wt, **kw)
	ref = R.conv2d_bwd_filter(xh[:4], dyh[:4], wh.shape, withbias=False, acc=np.float64, **okw)
	assert_close(dw4.get(), ref, atol=2e-6 * np.sqrt(4 * h * w) * 30, rtol=1e-4, what="backward-filter, 4 images")


@pytest.mark.parametrize("shape", [(256, 64, 55, 55), (256, 512, 28, 28), (256, 2048, 7, 7)])
def test_resnet_stage_batchnorm_normalises(bnd, shape):
	c = shape[1]
	x = dev_randn(bnd, shape, 7)
	bnd.linearKer(np.float32)(x, x, 3.0, -1.5)                          # mean -1.5, std 3
	rng = np.random.RandomState(1)
	scale, bias = (0.5 + rng.rand(c)).astype(np.float32), rng.randn(c).astype(np.float32)
	mean, var = gpu(bnd, np.zeros(c, np.float32)), gpu(bnd, np.ones(c, np.float32))

	y, sm, si = bnd.dnn.batchNormNd(x, mean, var, gpu(bnd, scale), gpu(bnd, bias), 1e-5, 1.0, False)
	n_red = shape[0] * shape[2] * shape[3]                              # samples per channel: 6-sigma sampling bounds
	assert_close(sm.get(), np.full(c, -1.5, np.float32), atol=6 * 3.0 / np.sqrt(n_red), what="batch mean")
	assert_close(1.0 / si.get() ** 2, np.full(c, 9.0, np.float32), rtol=6 * np.sqrt(2.0 / n_red), what="batch variance")

	# the output of every channel has mean = bias and variance = scale^2 (checked through the layer itself: a second
	# BN with unit scale / zero bias must report exactly those statistics)
	_, m2, i2 = bnd.dnn.batchNormNd(
		y, gpu(bnd, np.zeros(c, np.float32)), gpu(bnd, np.ones(c, np.float32)), gpu(bnd, np.ones(c, np.float32)),
		gpu(bnd, np.zeros(c, np.float32)), 0.0, 1.0, False
	)
	assert_close(m2.get(), bias, atol=2e-5, rtol=1e-4, what="output mean == bias")
	assert_close(1.0 / i2.get(), scale, atol=1e-5, rtol=1e-3, what="output std == scale")
	assert_close(mean.get(), sm.get(), atol=1e-6, what="factor 1: running mean == batch mean")

	# backward: gradients of a BN sum to zero over each channel, and dbias = sum(dy)
	dy = dev_randn(bnd, shape, 8)
	dx, dscale, dbias = bnd.dnn.batchNormNdBackward(dy, x, gpu(bnd, scale), sm, si, 1e-5)
	per_channel = bnd.matmod.matsum(dx.reshape(shape[0], c, shape[2] * shape[3]), axis=2)
	per_channel = bnd.matmod.matsum(per_channel, axis=0).get()
	assert np.abs(per_channel).max() < 1e-3 * np.sqrt(n_red), "dx does not sum to zero per channel"
	ref_db = bnd.matmod.matsum(bnd.matmod.matsum(dy.reshape(shape[0], c, shape[2] * shape[3]), axis=2), axis=0).get()
	assert_close(dbias.get(), ref_db, atol=1e-4 * np.sqrt(n_red), rtol=1e-4, what="dbias == sum(dy)")


def test_resnet_stem_maxpool_round_trip(bnd):
	shape = (256, 64, 112, 112)
	x = dev_randn(bnd, shape, 11)
	y, ws = bnd.dnn.poolNd(x, size=(3, 3), stride=(2, 2), pad=(0, 0), mode=bnd.PoolMode.max.value, test=False)
	assert y.shape == (256, 64, 55, 55)

	# every output is >= the window centre and equals some input; gradient routing: bwd(dy=1) has exactly one credit per
	# window (sum == number of windows) and x * (bwd > 0) recovers the maxima
	ones = bnd.GPUArray.empty(y.shape, dtype=np.float32).fill(1.0)
	dx = bnd.dnn.poolNdBackward(ones, x, y, ws, size=(3, 3), stride=(2, 2), pad=(0, 0), mode=bnd.PoolMode.max.value)
	assert abs(bnd.blas.l1norm(dx.ravel()) - y.size) < 1e-3 * y.size
	# <dx, x> = sum of the maxima = sum(y)
	lhs, rhs = bnd.blas.dot(dx.ravel(), x.ravel()), bnd.blas.dot(ones.ravel(), y.ravel())
	assert abs(lhs - rhs) < 1e-4 * (abs(rhs) + np.sqrt(y.size)), (lhs, rhs)

	sub = x.get()[:1, :2]
	ref = R.pool2d_fwd(sub, (3, 3), (2, 2), (0, 0), R.POOL_MAX)
	assert np.array_equal(y.get()[:1, :2], ref)


def test_training_step_is_deterministic(bnd, mini_golden):
	from puzzlelib_amd import nets, optim
	from puzzlelib_amd.surface import bound

	gpuarray = bound().gpuarray
	spec = nets.resnet_spec(stages=((8, 1), (16, 2)), classes=10, stem=8, softmax=False)
	spec = [l if l[0] != "avgpool" else ("avgpool", l[1], 8, 1, 0) for l in spec]
	data, labels = gpuarray.to_gpu(mini_golden["data"]), gpuarray.to_gpu(mini_golden["labels"])

	outs = []
	for _ in range(2):
		np.random.seed(7)
		net = nets.build(spec, name="mini", initscheme="he", actInplace=True)
		for name, var in net.namedParams().items():
			var.data.set(mini_golden["init_" + name])
		optimizer = optim.Adam(alpha=1e-3)
		optimizer.setupOn(net, useGlobalState=True)
		trainer = optim.Trainer(net, optim.CrossEntropy(), optimizer, batchsize=4)
		for _ in range(3):
			trainer.train(data, labels, random=False)
		outs.append({k: v.data.get() for k, v in net.namedParams().items()})

	for name in outs[0]:
		assert np.array_equal(outs[0][name], outs[1][name]), "non-deterministic parameter " + name


@pytest.mark.parametrize("cs", [
	dict(n=1, c=1, h=1, w=1, k=1, r=1, pad=0, stride=1),          # the smallest convolution
	dict(n=1, c=3, h=5, w=5, k=2, r=5, pad=0, stride=1),          # filter == image: 1x1 output
	dict(n=2, c=4, h=1, w=9, k=3, r=1, pad=0, stride=1),          # one-row maps
	dict(n=7, c=16, h=3, w=3, k=16, r=3, pad=1, stride=1),        # tap-major path, tiny maps, ragged pixel tile
	dict(n=1, c=2, h=8, w=8, k=5, r=3, pad=3, stride=3),          # padding larger than the filter reach
])
def test_conv_degenerate_shapes(bnd, cs):
	rng = np.random.RandomState(0)
	x = rng.randn(cs["n"], cs["c"], cs["h"], cs["w"]).astype(np.float32)
	w = rng.randn(cs["k"], cs["c"], cs["r"], cs["r"]).astype(np.float32)
	kw = dict(stride=(cs["stride"], ) * 2, pad=(cs["pad"], ) * 2, dilation=(1, 1), groups=1)
	y_ref = R.conv2d_fwd(x, w, None, acc=np.float64, **kw)
	dy = rng.randn(*y_ref.shape).astype(np.float32)

	gx, gw, gdy = gpu(bnd, x), gpu(bnd, w), gpu(bnd, dy)
	assert_close(bnd.dnn.convNd(gx, gw, None, **kw).get(), y_ref, atol=1e-4, rtol=1e-4, what="fwd")
	assert_close(bnd.dnn.convNdBackwardData(gdy, gw, data=gx, **kw).get(),
				 R.conv2d_bwd_data(dy, w, x.shape, acc=np.float64, **kw), atol=1e-4, rtol=1e-4, what="bwd data")
	dw_ref = R.conv2d_bwd_filter(x, dy, w.shape, withbias=False, acc=np.float64, **kw)
	assert_close(bnd.dnn.convNdBackwardParams(gx, gdy, gw, **kw).get(), dw_ref, atol=1e-4, rtol=1e-4, what="bwd filter")


def test_elementwise_and_norm_edge_cases(bnd):
	# zero-size launches are no-ops
	empty = bnd.GPUArray.empty((0, ), dtype=np.float32)
	bnd.reluKer(np.float32)(empty, empty)
	bnd.addKer(np.float32)(empty, empty, 1.0, empty, 1.0)
	assert empty.get().shape == (0, )

	# sizes around the 16-byte vector width, unaligned views
	for n in (1, 2, 3, 4, 5, 7, 8, 9, 1023):
		a = np.random.RandomState(n).randn(n + 3).astype(np.float32)
		ga = gpu(bnd, a)
		view = ga[1:n + 1]                                        # 4-byte aligned only
		out = bnd.GPUArray.empty((n, ), dtype=np.float32)
		bnd.reluKer(np.float32)(out, view)
		assert np.array_equal(out.get(), R.relu(a[1:n + 1]))

	# batch-norm of a constant tensor: variance 0 -> y = bias, invvar = 1/sqrt(eps)
	c = 3
	x = gpu(bnd, np.full((4, c, 5, 5), 2.5, np.float32))
	y, sm, si = bnd.dnn.batchNormNd(
		x, gpu(bnd, np.zeros(c, np.float32)), gpu(bnd, np.ones(c, np.float32)), gpu(bnd, np.full(c, 2.0, np.float32)),
		gpu(bnd, np.full(c, -1.0, np.float32)), 1e-5, 1.0, False
	)
	assert_close(sm.get(), np.full(c, 2.5, np.float32), atol=1e-6, what="mean of a constant")
	assert_close(si.get(), np.full(c, 1.0 / np.sqrt(1e-5), np.float32), rtol=1e-4, what="invvar of a constant")
	assert_close(y.get(), np.full((4, c, 5, 5), -1.0, np.float32), atol=1e-4, what="y == bias")

	# single-element reductions / N = 1 batch-norm with one pixel
	x1 = gpu(bnd, np.array([[[[4.0]]]], np.float32))
	y1, sm1, _ = bnd.dnn.batchNormNd(
		x1, gpu(bnd, np.zeros(1, np.float32)), gpu(bnd, np.ones(1, np.float32)), gpu(bnd, np.ones(1, np.float32)),
		gpu(bnd, np.zeros(1, np.float32)), 1e-5, 1.0, False
	)
	assert float(sm1.get()[0]) == 4.0 and abs(float(y1.get().ravel()[0])) < 1e-6

	# pooling window == plane (global average / global max), 1x1 window
	xp = np.random.RandomState(3).randn(2, 3, 7, 7).astype(np.float32)
	gp = gpu(bnd, xp)
	avg = bnd.dnn.poolNd(gp, size=(7, 7), stride=(1, 1), pad=(0, 0), mode=bnd.PoolMode.avgWithPad.value, test=True)
	assert_close(avg.get(), xp.mean(axis=(2, 3), keepdims=True), atol=1e-6, what="global average")
	mx = bnd.dnn.poolNd(gp, size=(7, 7), stride=(1, 1), pad=(0, 0), mode=bnd.PoolMode.max.value, test=True)
	assert np.array_equal(mx.get(), xp.max(axis=(2, 3), keepdims=True))
	one = bnd.dnn.poolNd(gp, size=(1, 1), stride=(1, 1), pad=(0, 0), mode=bnd.PoolMode.max.value, test=True)
	assert np.array_equal(one.get(), xp)

	# average pooling of 1x1 planes (a network whose last map is already 1x1): window == plane, forward and backward are copies
	# (round-3 advisor finding: the global-average fast path's 32-bit magic division does not cover hw == 1)
	for mode in (bnd.PoolMode.avgWithPad.value, bnd.PoolMode.avgNoPad.value):
		x1 = np.random.RandomState(5).randn(37, 300, 1, 1).astype(np.float32)          # > 1024 elements: several workgroups
		g1 = gpu(bnd, x1)
		y1, ws1 = bnd.dnn.poolNd(g1, size=(1, 1), stride=(1, 1), pad=(0, 0), mode=mode, test=False)
		assert np.array_equal(y1.get(), x1)
		dy1 = np.random.RandomState(6).randn(*x1.shape).astype(np.float32)
		dx1 = bnd.dnn.poolNdBackward(gpu(bnd, dy1), g1, y1, ws1, size=(1, 1), stride=(1, 1), pad=(0, 0), mode=mode)
		assert np.array_equal(dx1.get(), dy1), "average pooling backward on 1x1 planes"
	# ... and the fast path itself on the smallest plane it takes (2 elements) and on 7x7, backward against the oracle
	for hw in ((1, 2), (7, 7)):
		xa = np.random.RandomState(7).randn(9, 130, *hw).astype(np.float32)
		ga = gpu(bnd, xa)
		ya, wsa = bnd.dnn.poolNd(ga, size=hw, stride=(1, 1), pad=(0, 0), mode=bnd.PoolMode.avgWithPad.value, test=False)
		assert_close(ya.get(), xa.mean(axis=(2, 3), keepdims=True), atol=1e-6, what="global average %s" % (hw, ))
		dya = np.random.RandomState(8).randn(9, 130, 1, 1).astype(np.float32)
		dxa = bnd.dnn.poolNdBackward(gpu(bnd, dya), ga, ya, wsa, size=hw, stride=(1, 1), pad=(0, 0), mode=bnd.PoolMode.avgWithPad.value)
		assert_close(dxa.get(), np.broadcast_to(dya / (hw[0] * hw[1]), xa.shape), atol=1e-7, rtol=1e-6, what="global average backward %s" % (hw, ))


@pytest.mark.parametrize("tile", [2, 4])
@pytest.mark.parametrize("shape", [(256, 64, 55, 55), (256, 128, 28, 28), (256, 256, 14, 14), (256, 512, 7, 7)])
def test_resnet_3x3_layers_winograd_agrees_with_implicit_gemm(bnd, shape, tile):
	"""The four 3x3 layer shapes of ResNet-50 at batch 256, all three passes: the Winograd kernels (both output tiles) against
	the implicit GEMM (itself checked against the oracle at oracle-sized inputs) — two independent algorithms, agreement to
	3e-5 of the result's scale for F(2x2, 3x3), 1e-4 for F(4x4, 3x3) — and run-to-run determinism of the Winograd
	backward-filter's slab reduction."""
	bnd.dnn.setWinogradTile(tile)
	try:
		winograd_fullsize_case(bnd, shape, 3e-5 if tile == 2 else 1e-4)
	finally:
		bnd.dnn.setWinogradTile(bnd.dnn.winogradTileDefault)


def winograd_fullsize_case(bnd, shape, tol):
	n, c, h, w = shape
	kw = dict(stride=(1, 1), pad=(1, 1), dilation=(1, 1), groups=1)
	x, dy = dev_randn(bnd, shape, 11), dev_randn(bnd, shape, 12)
	wt = gpu(bnd, (np.random.RandomState(13).randn(c, c, 3, 3) / np.sqrt(c * 9)).astype(np.float32))
	diff = bnd.GPUArray.empty(shape, dtype=np.float32)

	def max_abs(a, b, out):
		bnd.addKer(np.float32)(out, a, 1.0, b, -1.0)
		return max(float(out.max().get()), -float(out.min().get()))

	y3, y5 = bnd.dnn.convNd(x, wt, None, algo=3, **kw), bnd.dnn.convNd(x, wt, None, algo=5, **kw)
	assert max_abs(y3, y5, diff) < tol * max(1.0, float(y5.max().get()))

	d3 = bnd.dnn.convNdBackwardData(dy, wt, data=x, algo=3, **kw)
	d5 = bnd.dnn.convNdBackwardData(dy, wt, data=x, algo=5, **kw)
	assert max_abs(d3, d5, diff) < tol * max(1.0, float(d5.max().get()))

	w3 = bnd.dnn.convNdBackwardParams(x, dy, wt, algo=3, **kw)
	w5 = bnd.dnn.convNdBackwardParams(x, dy, wt, algo=5, **kw)
	dwdiff = bnd.GPUArray.empty(wt.shape, dtype=np.float32)
	assert max_abs(w3, w5, dwdiff) < 3e-5 * max(1.0, float(w5.max().get()))
	assert np.array_equal(bnd.dnn.convNdBackwardParams(x, dy, wt, algo=3, **kw).get(), w3.get())


# ================================================================================================ config 4 at full size
# every 1x1 convolution shape of the reference's ResNet-50 (Models/Nets/ResNet.py:23-121, 55x55 stage-2 maps) and the
# 7x7/2 stem at batch 256: (C, H, W) -> (K, size, stride, pad)
R50_POINTWISE_AND_STEM = [
	((3, 224, 224), (64, 7, 2, 3)), ((64, 55, 55), (64, 1, 1, 0)), ((64, 55, 55), (256, 1, 1, 0)), ((256, 55, 55), (64, 1, 1, 0)),
	((256, 55, 55), (128, 1, 2, 0)), ((128, 28, 28), (512, 1, 1, 0)), ((256, 55, 55), (512, 1, 2, 0)), ((512, 28, 28), (128, 1, 1, 0)),
	((512, 28, 28), (256, 1, 2, 0)), ((256, 14, 14), (1024, 1, 1, 0)), ((512, 28, 28), (1024, 1, 2, 0)), ((1024, 14, 14), (256, 1, 1, 0)),
	((1024, 14, 14), (512, 1, 2, 0)), ((512, 7, 7), (2048, 1, 1, 0)), ((1024, 14, 14), (2048, 1, 2, 0)), ((2048, 7, 7), (512, 1, 1, 0)),
]


def max_abs_diff(bnd, a, b):
	d = bnd.GPUArray.empty(a.shape, dtype=np.float32)
	bnd.addKer(np.float32)(d, a, 1.0, b, -1.0)
	return max(float(d.max().get()), -float(d.min().get()))


@pytest.mark.parametrize("layer", R50_POINTWISE_AND_STEM, ids=lambda l: "%dx%dx%d_to_%d_k%d_s%d" % (l[0] + l[1][:3]))
def test_resnet50_pointwise_and_stem_layers_full_size(bnd, layer):
	"""All three passes of every 1x1 layer and of the stem at batch 256, as the training step runs them — `auto` kernels,
	the batch-norm backward folded into the backward gathers (pz_conv2d_bwd_{data,filter}_bn), compact stride-2 input
	gradients, the dedicated stem backward-data — through size-independent properties (the adjoint identities) and fp64
	oracle checks on single images."""
	from puzzlelib_amd import lazy, fusion
	from puzzlelib_amd.surface import bound
	surf = bound()
	Dnn, dot, G = surf.Dnn, bnd.blas.dot, bnd.GPUArray
	lazy.enabled, lazy.disabled = True, set()
	(c, h, w), (k, size, stride, pad) = layer
	n = 256
	okw = dict(stride=(stride, stride), pad=(pad, pad), dilation=(1, 1), groups=1)
	algos = Dnn.ConvFwdAlgo.auto, Dnn.ConvBwdDataAlgo.auto, Dnn.ConvBwdFilterAlgo.auto

	x = dev_randn(bnd, (n, c, h, w), 21)
	wt = gpu(bnd, (np.random.RandomState(22).randn(k, c, size, size) / np.sqrt(c * size * size)).astype(np.float32))
	y = Dnn.convNd(x, wt, None, okw["stride"], okw["pad"], okw["dilation"], 1, algos[0])
	p, q = y.shape[2:]
	dy = dev_randn(bnd, (n, k, p, q), 23)

	# adjoint identities at full size: <dy, conv(x; w)> == <bwd_data(dy; w), x> == <w, bwd_filter(x, dy)>
	lhs = dot(dy.ravel(), y.ravel())
	dx = Dnn.convNdBackwardData(dy, wt, x, okw["stride"], okw["pad"], okw["dilation"], 1, algos[1])
	if stride == 2 and size == 1:
		assert isinstance(lazy.pending(dx), fusion.Up2), "stride-2 pointwise layers keep their input gradient compact"
	mid = dot(dx.ravel(), x.ravel())
	dw = G.zeros(wt.shape, dtype=np.float32)
	Dnn.convNdBackwardParams(x, dy, wt, None, okw["stride"], okw["pad"], okw["dilation"], 1, dw, None, 1.0, 1.0, algos[2])
	rhs = dot(dw.ravel(), wt.ravel())
	scale = abs(lhs) + np.sqrt(float(dy.size))
	assert abs(lhs - mid) < 3e-4 * scale, "backward-data is not the adjoint of forward: %r vs %r" % (lhs, mid)
	assert abs(lhs - rhs) < 3e-4 * scale, "backward-filter is not the adjoint of forward: %r vs %r" % (lhs, rhs)

	# fp64 oracle on the first and the last image
	wh = wt.get()
	for img in (0, n - 1):
		xi, dyi = x[img:img + 1].get(), dy[img:img + 1].get()
		ref = R.conv2d_fwd(xi, wh, None, acc=np.float64, **okw)
		assert_close(y[img:img + 1].get(), ref, atol=2e-4, rtol=2e-4, what="forward, image %d" % img)
		ref = R.conv2d_bwd_data(dyi, wh, xi.shape, acc=np.float64, **okw)
		assert_close(dx[img:img + 1].get(), ref, atol=2e-4, rtol=2e-4, what="backward-data, image %d" % img)
	sub = 2
	dws = G.zeros(wt.shape, dtype=np.float32)
	Dnn.convNdBackwardParams(x[:sub], dy[:sub], wt, None, okw["stride"], okw["pad"], okw["dilation"], 1, dws, None, 1.0, 1.0, algos[2])
	ref = R.conv2d_bwd_filter(x[:sub].get(), dy[:sub].get(), wh.shape, withbias=False, acc=np.float64, **okw)
	assert_close(dws.get(), ref, atol=1e-4 * np.sqrt(sub * p * q), rtol=2e-4, what="backward-filter, %d images" % sub)

	if size != 1:
		return

	# the batch-norm behind this layer, backward: its input gradient is only described (A*dy + B*y + C per channel) and the
	# convolution's backward kernels evaluate it while gathering. Against the written-out form (same coefficients).
	rng = np.random.RandomState(24)
	coef = gpu(bnd, np.stack([1.0 + 0.1 * rng.randn(k), 0.05 * rng.randn(k), 0.01 * rng.randn(k), np.zeros(k)], axis=1).astype(np.float32))
	desc = bnd.dnn.convDesc((n, c, h, w), wt.shape, stride, pad, 1, 1)
	assert bnd.dnn.bnFoldSupported(desc, -1) == (k % 16 == 0)

	def described():
		g = G.empty(dy.shape, dtype=np.float32)
		lazy.attach(g, fusion.BnBwdApply(dy, y, coef))
		return g

	written = described()
	written.rptr                                                       # pz_bn_bwd_apply_coef
	lazy.counters.clear()
	dx_fold = Dnn.convNdBackwardData(described(), wt, x, okw["stride"], okw["pad"], okw["dilation"], 1, algos[1])
	dx_plain = Dnn.convNdBackwardData(written, wt, x, okw["stride"], okw["pad"], okw["dilation"], 1, algos[1])
	dw_fold, dw_plain = G.zeros(wt.shape, dtype=np.float32), G.zeros(wt.shape, dtype=np.float32)
	Dnn.convNdBackwardParams(x, described(), wt, None, okw["stride"], okw["pad"], okw["dilation"], 1, dw_fold, None, 1.0, 1.0, algos[2])
	Dnn.convNdBackwardParams(x, written, wt, None, okw["stride"], okw["pad"], okw["dilation"], 1, dw_plain, None, 1.0, 1.0, algos[2])
	assert lazy.counters.get("dgrad_bn_fold", 0) == 1 and lazy.counters.get("wgrad_bn_fold", 0) == 1
	top = max(float(dx_plain.max().get()), -float(dx_plain.min().get()))
	assert max_abs_diff(bnd, dx_fold, dx_plain) < 3e-5 * top + 1e-6, "backward-data with the batch-norm folded in"
	topw = max(float(dw_plain.max().get()), -float(dw_plain.min().get()))
	assert max_abs_diff(bnd, dw_fold, dw_plain) < 1e-4 * topw + 1e-5, "backward-filter with the batch-norm folded in"


def test_resnet50_b256_step_fused_equals_literal_and_matches_oracle(bnd):
	"""Config 4 itself: loadResNet("50", actInplace=True) at batch 256, forward + cross-entropy + backward, (1) under the
	lazy-buffer layer and (2) with it off (every reference call launches its own kernels), plus (3) with the batch-norm
	backward fold on. (1) and (2) must agree bit for bit in logits, loss and every parameter gradient (compared on the
	device); (3) to fp32 rounding; the logits of two images against the CPU oracle (evaluation mode, running statistics)."""
	import cpu_net as N
	from puzzlelib_amd import nets, optim, lazy, backend
	from puzzlelib_amd.surface import bound
	gpuarray = bound().gpuarray
	backend.DnnContext.convStatsPolicy = "never"          # (epilogue statistics differ in summation order from the BN's own pass)

	np.random.seed(1234)
	net = nets.loadResNet(None, "50", actInplace=True, initscheme="he")
	net.layers.pop()                                       # the trailing SoftMax: training runs on raw scores
	optimizer = optim.Adam(alpha=1e-3)
	optimizer.setupOn(net, useGlobalState=True)
	cost = optim.CrossEntropy()
	rng = np.random.RandomState(1234)
	data = gpuarray.to_gpu(rng.randn(256, 3, 224, 224).astype(np.float32))
	labels = gpuarray.to_gpu(rng.randint(0, 1000, size=(256, )).astype(np.int32))
	net.trainMode()

	def passOnce():
		for layer in net.walk():                           # same running statistics / momentum factor on every pass
			if layer.kind == "bn":
				layer.cfg["passes"] = 0
				layer.attrs["mean"].fill(0.0)
				layer.attrs["var"].fill(1.0)
		lazy.counters.clear()
		logits = net(data)
		grad = cost(logits, labels, queryError=False)
		optimizer.zeroGradParams()
		net.backward(grad, updGrad=False)
		out = (logits.copy(), float(cost.devErr.get()), optimizer.grads.ary.copy(), dict(lazy.counters))
		net.reset()
		return out

	try:
		lazy.disabled = {"bnbwdfold"}
		fused = passOnce()
		lazy.enabled = False
		literal = passOnce()
		lazy.enabled, lazy.disabled = True, set()
		folded = passOnce()
	finally:
		lazy.enabled, lazy.disabled = True, set()
		backend.DnnContext.convStatsPolicy = "adaptive"

	taken = fused[3]
	assert taken.get("bn_apply_add", 0) == 16 and taken.get("bn_bwd_gate", 0) == 33 and taken.get("gate_stats", 0) == 12
	assert taken.get("gate_stats_up2", 0) == 3 and taken.get("compact_dgrad", 0) == 6 and taken.get("bn_bwd_from_partials", 0) == 19
	assert folded[3].get("dgrad_bn_fold", 0) == 19 and folded[3].get("wgrad_bn_fold", 0) == 19

	assert fused[1] == literal[1], "loss: %r fused, %r literal" % (fused[1], literal[1])
	assert max_abs_diff(bnd, fused[0], literal[0]) == 0.0, "logits differ between the fused and the literal call sequence"
	assert max_abs_diff(bnd, fused[2], literal[2]) == 0.0, "parameter gradients differ between fused and literal"

	top = max(float(literal[2].max().get()), -float(literal[2].min().get()))
	assert folded[1] == literal[1]
	assert max_abs_diff(bnd, folded[2], literal[2]) < 2e-3 * top, "gradients with the batch-norm backward folded into the convolutions"
	assert np.isfinite(fused[1]) and 6.0 < fused[1] / 256 < 9.0, "cross-entropy of a random-init 1000-class net"

	# oracle spot check: two images, evaluation mode (running statistics as initialised), CPU restatement of the network
	net.evalMode()
	dev = net(data[:2]).get()
	spec = nets.resnet50_spec(softmax=False)
	params = {name: p.data.get() for name, p in net.namedParams().items()}
	attrs = {name: a.get() for name, a in net.namedAttrs().items()}
	cnet = N.CpuNet(spec, params, attrs)
	cnet.train = False
	ref = cnet.forward(data[:2].get())
	assert_close(dev, ref, atol=2e-3 * np.abs(ref).max() + 1e-4, rtol=2e-3, what="ResNet-50 logits vs oracle, 2 images")


# config 3 (TestLib/CnnCifar10NIN.py:13-49): the nine convolutions of the CIFAR-10 NiN at batch 128, all with a bias
NIN_LAYERS = [
	((3, 32, 32), (192, 5, 1, 2)), ((192, 32, 32), (160, 1, 1, 0)), ((160, 32, 32), (96, 1, 1, 0)),
	((96, 16, 16), (192, 5, 1, 2)), ((192, 16, 16), (192, 1, 1, 0)), ((192, 8, 8), (192, 3, 1, 1)),
	((192, 8, 8), (192, 1, 1, 0)), ((192, 8, 8), (10, 1, 1, 0)),
]


@pytest.mark.parametrize("layer", NIN_LAYERS, ids=lambda l: "%dx%dx%d_to_%d_k%d" % (l[0] + l[1][:2]))
def test_nin_layers_bias_gradient_rides_in_the_filter_gradient(bnd, layer):
	"""Backend/Dnn.py convNdBackwardParams(withbias=True) at config 3's sizes: the bias gradient is the sum of the output
	gradient over images and pixels (Hip/Wrappers/MIOpen.py:435-455), whichever kernel produced it — folded into the
	implicit-GEMM filter-gradient kernel (whole tensors against fp64 sums, |err| <= 1e-5 + 1e-4 |ref| scaled by the sum's
	length), and the accumulate form bgrad <- momentum * bgrad + scale * db on the same launch."""
	(c, h, w), (k, size, stride, pad) = layer
	n = 128
	rng = np.random.RandomState(k * 31 + c)
	x = dev_randn(bnd, (n, c, h, w), seed=11 + c)
	wt = gpu(bnd, (rng.randn(k, c, size, size) / np.sqrt(c * size * size)).astype(np.float32))
	oh, ow = (h + 2 * pad - size) // stride + 1, (w + 2 * pad - size) // stride + 1
	dy = dev_randn(bnd, (n, k, oh, ow), seed=23 + k)
	kw = dict(stride=stride, pad=pad)

	dw, db = bnd.dnn.convNdBackwardParams(x, dy, wt, withbias=True, **kw)
	dyh = dy.get().astype(np.float64)
	db_ref = dyh.sum(axis=(0, 2, 3))
	tol = np.sqrt(n * oh * ow)
	assert_close(db.get(), db_ref, atol=1e-5 * tol, rtol=1e-4, what="bias gradient")

	# the filter gradient next to it is the one the call without a bias gives, bit for bit
	dw_plain = bnd.dnn.convNdBackwardParams(x, dy, wt, withbias=False, **kw)
	assert np.array_equal(dw.get(), dw_plain.get()), "filter gradient changed by the folded bias gradient"

	# accumulate contract on both destinations
	wg0, bg0 = rng.randn(*wt.shape).astype(np.float32), rng.randn(k).astype(np.float32)
	wg, bg = gpu(bnd, wg0), gpu(bnd, bg0)
	bnd.dnn.convNdBackwardParams(x, dy, wt, withbias=True, wgrad=wg, bgrad=bg, scale=0.5, momentum=0.9, **kw)
	assert_close(bg.get(), 0.9 * bg0 + 0.5 * db_ref, atol=1e-5 * tol, rtol=1e-4, what="accumulated bias gradient")
	assert_close(wg.get(), 0.9 * wg0 + 0.5 * dw_plain.get().astype(np.float64), atol=1e-5 * tol, rtol=1e-4, what="accumulated filter gradient")
