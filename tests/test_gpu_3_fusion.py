"""
GPU tests of the lazy-buffer layer (puzzlelib_amd/lazy.py, fusion.py): sequences of the reference's own calls — nothing but
wrappers with the reference's signatures — must give the same numbers whether the backend defers and fuses them or
launches every call on the spot (`lazy.enabled = False`, the literal behaviour), bit for bit wherever the fused kernel
performs the same floating-point operations in the same order, and must match the fp64 oracle in both modes.
Hazards are tested with real data: a description must be evaluated with the values its inputs had when the call was made.
"""
import numpy as np
import pytest

import cpu_ref as R
from conftest import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture()
def surf(bnd):
	from puzzlelib_amd.surface import bound
	from puzzlelib_amd import lazy, backend
	lazy.enabled, lazy.disabled = True, set()
	backend.DnnContext.convStatsPolicy = "adaptive"
	yield bound()
	lazy.enabled, lazy.disabled = True, set()
	backend.DnnContext.convStatsPolicy = "adaptive"
	bnd.dnn.setWinogradTile(bnd.dnn.winogradTileDefault)


def both(fn):
	"""fn() under the lazy layer and with it off -> (lazy result, literal result)"""
	from puzzlelib_amd import lazy
	out = []
	for mode in (True, False):
		lazy.enabled = mode
		lazy.counters.clear()
		try:
			out.append((fn(), dict(lazy.counters)))
		finally:
			lazy.enabled = True
	return out


def bnParams(surf, rng, c):
	g = surf.gpuarray
	shape = (1, c, 1, 1)
	scale, bias = rng.randn(*shape).astype(np.float32), rng.randn(*shape).astype(np.float32)
	rm, rv = rng.randn(*shape).astype(np.float32), (0.5 + rng.rand(*shape)).astype(np.float32)
	return (scale, bias, rm, rv), lambda: (g.to_gpu(scale), g.to_gpu(bias), g.to_gpu(rm), g.to_gpu(rv))


@pytest.mark.parametrize("shape", [(4, 5, 2, 3), (3, 7, 55, 55), (5, 3, 7, 7), (8, 130, 14, 14)])
def test_batchnorm_relu_pair(surf, shape):
	"""BatchNorm2D -> Activation(relu, inplace=True) forward and backward as the reference issues them
	(Modules/BatchNormND.py:60-88, Activation.py:52-70). Lazy: one fused apply, and the ReLU derivative evaluated inside
	the batch-norm backward from x and the forward's own coefficients. Must equal the literal four kernels bit for bit."""
	g, Dnn, El = surf.gpuarray, surf.Dnn, surf.ElementWise
	rng = np.random.RandomState(11)
	x = (0.5 + 2.0 * rng.randn(*shape)).astype(np.float32)
	dy = rng.randn(*shape).astype(np.float32)
	(scale, bias, rm, rv), fresh = bnParams(surf, rng, shape[1])

	def run():
		gs, gb, grm, grv = fresh()
		gx, gdy = g.to_gpu(x), g.to_gpu(dy)
		y, sm, si = Dnn.batchNormNd(gx, gs, gb, grm, grv, 1e-5, 0.3, False)
		El.reluKer(np.float32)(y, y)
		El.reluDerKer(np.float32)(gdy, gdy, y)
		dx, ds, db = Dnn.batchNormNdBackward(gx, gdy, gs, sm, si, 1e-5)
		return [a.get() for a in (y, dx, ds, db, sm, si, grm, grv)]

	(lz, taken), (lit, none) = both(run)
	assert taken.get("bn_bwd_gate", 0) == 1 and taken.get("bn_apply_relu", 0) == 1
	assert none == {"bn_apply": 1}, "with the layer off the normalisation is written at once and nothing else is deferred"
	for a, b, what in zip(lz, lit, ("y", "dx", "dscale", "dbias", "savemean", "saveinvvar", "running mean", "running var")):
		assert np.array_equal(a, b), what

	rm_ref, rv_ref = rm.ravel().copy(), rv.ravel().copy()
	y_ref, sm_ref, si_ref = R.bn_fwd_train(x, scale.ravel(), bias.ravel(), rm_ref, rv_ref, 1e-5, 0.3, acc=np.float64)
	assert_close(lz[0], np.maximum(y_ref, 0), atol=2e-5, rtol=1e-4, what="relu(bn(x))")
	# the gate of the device (sign of its own fp32 y) is used for the oracle's backward: a pre-activation at rounding
	# distance from zero may gate either way, which is not what this test is about
	dx_ref, ds_ref, db_ref = R.bn_bwd(dy * (lz[0] > 0), x, scale.ravel(), sm_ref, si_ref, acc=np.float64)
	assert_close(lz[1], dx_ref, atol=1e-4 * np.abs(dx_ref).max() + 1e-6, rtol=1e-3, what="dx")
	assert_close(lz[2].ravel(), ds_ref, atol=1e-4 * np.abs(ds_ref).max() + 1e-5, rtol=1e-3, what="dscale")
	assert_close(lz[6].ravel(), rm_ref, atol=1e-5, rtol=1e-5, what="running mean")


def test_described_tensor_sees_the_inputs_of_call_time(surf):
	"""A batch-norm output that was only described must come out with the x of the moment batchNormNd was called, even if x
	is overwritten before anybody reads the output; same for a residual sum whose term changes."""
	g, Dnn, El, Blas = surf.gpuarray, surf.Dnn, surf.ElementWise, surf.Blas
	from puzzlelib_amd import lazy, fusion
	rng = np.random.RandomState(3)
	x = rng.randn(4, 6, 9, 9).astype(np.float32)
	(scale, bias, rm, rv), fresh = bnParams(surf, rng, 6)

	gs, gb, grm, grv = fresh()
	gx = g.to_gpu(x)
	y, sm, si = Dnn.batchNormNd(gx, gs, gb, grm, grv, 1e-5, 1.0, False)
	assert isinstance(lazy.pending(y), fusion.BnApply)
	gx.set(np.zeros_like(x))                                            # overwrites the description's input
	y_ref = R.bn_fwd_train(x, scale.ravel(), bias.ravel(), rm.ravel().copy(), rv.ravel().copy(), 1e-5, 1.0)[0]
	assert_close(y.get(), y_ref, atol=2e-5, rtol=1e-4, what="y after its input was overwritten")

	a, b = rng.randn(4, 6, 9, 9).astype(np.float32), rng.randn(4, 6, 9, 9).astype(np.float32)
	ga, gb2 = g.to_gpu(a), g.to_gpu(b)
	total = g.empty(a.shape, dtype=np.float32)
	total.fill(0)
	Blas.toVectorAddVector(total.ravel(), ga.ravel())
	Blas.toVectorAddVector(total.ravel(), gb2.ravel())
	assert isinstance(lazy.pending(total), fusion.Sum)
	El.reluKer(np.float32)(gb2, gb2)                                    # in-place change of a term, after the axpy
	assert np.array_equal(total.get(), a + b)

	# a gate described on a gradient applies the sign of y as it was at reluDerKer time
	gy, gg = g.to_gpu(a), g.to_gpu(b)
	El.reluDerKer(np.float32)(gg, gg, gy)
	assert isinstance(lazy.pending(gg), fusion.Gate)
	gy.fill(-1.0)
	assert np.array_equal(gg.get(), b * (a > 0))


@pytest.mark.parametrize("cfg", [dict(n=4, c=16, k=32, hw=(12, 12), r=1), dict(n=3, c=24, k=40, hw=(9, 11), r=1),
								 dict(n=4, c=64, k=64, hw=(20, 20), r=3), dict(n=2, c=3, k=16, hw=(32, 32), r=7, stride=2, pad=3),
								 # Winograd (auto): tile blocks with explicit counts — odd maps (border tiles hold 1 or 2 pixels), a last
								 # block of fewer than 32 tiles, output channels that do not fill a block of 64
								 dict(n=3, c=32, k=64, hw=(11, 13), r=3, algo="auto"), dict(n=5, c=64, k=96, hw=(7, 7), r=3, algo="auto"),
								 dict(n=2, c=128, k=128, hw=(28, 28), r=3, algo="auto"), dict(n=1, c=32, k=40, hw=(5, 6), r=3, pad=0, algo="auto"),
								 # the same through the F(4x4, 3x3) kernel's epilogue (strips of 32 tiles of 16 pixels, counted): ragged
								 # maps, fewer than 32 tiles, a channel block that is not full, several tile blocks
								 dict(n=3, c=32, k=64, hw=(11, 13), r=3, algo="auto", tile=4), dict(n=5, c=64, k=96, hw=(7, 7), r=3, algo="auto", tile=4),
								 dict(n=1, c=32, k=40, hw=(5, 6), r=3, pad=0, algo="auto", tile=4),
								 dict(n=6, c=64, k=64, hw=(28, 28), r=3, algo="auto", tile=4)])
def test_convolution_statistics_feed_the_batchnorm(surf, bnd, cfg):
	"""Conv2D -> BatchNorm2D: the convolution's epilogue leaves per-strip sums and the batch-norm skips its statistics
	pass (policy "always"; "adaptive" learns it after the first pass). Same mean / variance up to summation order."""
	from puzzlelib_amd import backend, lazy
	g, Dnn = surf.gpuarray, surf.Dnn
	bnd.dnn.setWinogradTile(cfg.get("tile", 0))          # (the fixture restores the default)
	rng = np.random.RandomState(5)
	n, c, k, (h, w), r = cfg["n"], cfg["c"], cfg["k"], cfg["hw"], cfg["r"]
	stride, pad = cfg.get("stride", 1), cfg.get("pad", r // 2)
	x = rng.randn(n, c, h, w).astype(np.float32)
	wt = (rng.randn(k, c, r, r) / np.sqrt(c * r * r)).astype(np.float32)
	(scale, bias, rm, rv), fresh = bnParams(surf, rng, k)
	algo = getattr(surf.Dnn.ConvFwdAlgo, cfg.get("algo", "implicitGemm"))

	results = {}
	for policy in ("never", "always"):
		backend.DnnContext.convStatsPolicy = policy
		lazy.counters.clear()
		gs, gb, grm, grv = fresh()
		y = Dnn.convNd(g.to_gpu(x), g.to_gpu(wt), None, (stride, stride), (pad, pad), (1, 1), 1, algo)
		out, sm, si = Dnn.batchNormNd(y, gs, gb, grm, grv, 1e-5, 0.3, False)
		results[policy] = [a.get() for a in (y, out, sm, si, grm, grv)] + [lazy.counters.get("conv_stats", 0)]
	assert results["never"][-1] == 0 and results["always"][-1] == 1
	assert np.array_equal(results["never"][0], results["always"][0]), "the convolution's output does not depend on the epilogue sums"
	for a, b, what in zip(results["never"][1:6], results["always"][1:6], ("y", "savemean", "saveinvvar", "rmean", "rvar")):
		assert_close(b, a, atol=2e-5, rtol=2e-5, what=what)

	# adaptive: the second pass of the same filter emits them
	backend.DnnContext.convStatsPolicy = "adaptive"
	gw = g.to_gpu(wt)
	for expect in (0, 1):
		lazy.counters.clear()
		gs, gb, grm, grv = fresh()
		y = Dnn.convNd(g.to_gpu(x), gw, None, (stride, stride), (pad, pad), (1, 1), 1, algo)
		Dnn.batchNormNd(y, gs, gb, grm, grv, 1e-5, 0.3, False)
		assert lazy.counters.get("conv_stats", 0) == expect


def residualBlock(surf, rng, n, c, hw, projection, stride):
	"""arrays of a bottleneck tail: main = conv1x1(a) -> BN, shortcut = conv1x1(x, stride) -> BN or x itself"""
	h, w = hw
	mid = max(c // 4, 4)
	p, q = (h + stride - 1) // stride, (w + stride - 1) // stride
	data = dict(
		a=rng.randn(n, mid, p, q).astype(np.float32), wa=(rng.randn(c, mid, 1, 1) / np.sqrt(mid)).astype(np.float32),
		x=rng.randn(n, c if not projection else mid * 2, h, w).astype(np.float32), dy=rng.randn(n, c, p, q).astype(np.float32)
	)
	if projection:
		data["ws"] = (rng.randn(c, mid * 2, 1, 1) / np.sqrt(mid * 2)).astype(np.float32)
	return data


@pytest.mark.parametrize("cfg", [dict(n=4, c=64, hw=(14, 14), projection=False, stride=1),
								 dict(n=3, c=128, hw=(13, 15), projection=True, stride=2),
								 dict(n=2, c=40, hw=(9, 9), projection=True, stride=1)])       # 40 maps: the fold declines
def test_residual_block_tail_and_its_backward(surf, cfg):
	"""The end of a ResNet block and the start of its backward exactly as the reference's modules issue them:
	  fwd  convNd, batchNormNd (x2 with a projection), zeros + 2 x toVectorAddVector (Add), reluKer in place
	  bwd  reluDerKer in place on the incoming gradient (itself a Replicate fan-in: zeros + 2 x toVectorAddVector),
	       batchNormNdBackward + addVectorToVector x2 per BN, convNdBackwardData / convNdBackwardParams
	Lazy: residual sum normalising on the fly (+ sign mask), fan-in + gate + both BNs' statistics in one pass, BN backward
	folded into the 1x1 convolutions, compact stride-2 input gradients, filter gradients on the side stream.
	Bit-identical to the literal sequence with the (rounding-different) BN-backward fold off; within tolerance with it on."""
	from puzzlelib_amd import lazy, backend
	g, Dnn, El, Blas = surf.gpuarray, surf.Dnn, surf.ElementWise, surf.Blas
	rng = np.random.RandomState(9)
	n, c, stride, projection = cfg["n"], cfg["c"], cfg["stride"], cfg["projection"]
	d = residualBlock(surf, rng, n, c, cfg["hw"], projection, stride)
	(_, _, _, _), freshA = bnParams(surf, rng, c)
	(_, _, _, _), freshS = bnParams(surf, rng, c)
	g0, g1 = rng.randn(*d["dy"].shape).astype(np.float32), rng.randn(*d["dy"].shape).astype(np.float32)
	auto = Dnn.ConvFwdAlgo.auto, Dnn.ConvBwdDataAlgo.auto, Dnn.ConvBwdFilterAlgo.auto
	backend.DnnContext.convStatsPolicy = "never"          # (epilogue statistics differ in summation order; tested above)

	def run():
		ga, gwa, gx = g.to_gpu(d["a"]), g.to_gpu(d["wa"]), g.to_gpu(d["x"])
		sA, bA, mA, vA = freshA()
		ca = Dnn.convNd(ga, gwa, None, (1, 1), (0, 0), (1, 1), 1, auto[0])
		ya, smA, siA = Dnn.batchNormNd(ca, sA, bA, mA, vA, 1e-5, 1.0, False)
		if projection:
			gws = g.to_gpu(d["ws"])
			sS, bS, mS, vS = freshS()
			cs = Dnn.convNd(gx, gws, None, (stride, stride), (0, 0), (1, 1), 1, auto[0])
			ys, smS, siS = Dnn.batchNormNd(cs, sS, bS, mS, vS, 1e-5, 1.0, False)
		else:
			ys = gx
		out = g.empty(ya.shape, dtype=np.float32, allocator=g.memoryPool)
		out.fill(0)
		for term in (ya, ys):
			Blas.toVectorAddVector(out.ravel(), term.ravel())
		El.reluKer(np.float32)(out, out)

		# backward: the gradient arrives as the next block's fan-in
		grad = g.empty(out.shape, dtype=np.float32, allocator=g.memoryPool)
		grad.fill(0)
		for term in (g.to_gpu(g0), g.to_gpu(g1)):
			Blas.toVectorAddVector(grad.ravel(), term.ravel())
		El.reluDerKer(np.float32)(grad, grad, out)

		res = {"out": out, "grad": grad}
		wgA = g.zeros(d["wa"].shape, dtype=np.float32)
		dA, dsA, dbA = Dnn.batchNormNdBackward(ca, grad, sA, smA, siA, 1e-5)
		res["dxa"] = Dnn.convNdBackwardData(dA, gwa, data=ga, stride=(1, 1), pad=(0, 0), dilation=(1, 1), groups=1, algo=auto[1])
		Dnn.convNdBackwardParams(ga, dA, gwa, None, (1, 1), (0, 0), (1, 1), 1, wgA, None, 1.0, 1.0, auto[2])
		res.update(wgA=wgA, dsA=dsA, dbA=dbA)
		if projection:
			wgS = g.zeros(d["ws"].shape, dtype=np.float32)
			dS, dsS, dbS = Dnn.batchNormNdBackward(cs, grad, sS, smS, siS, 1e-5)
			res["dxs"] = Dnn.convNdBackwardData(dS, gws, data=gx, stride=(stride, stride), pad=(0, 0), dilation=(1, 1), groups=1, algo=auto[1])
			Dnn.convNdBackwardParams(gx, dS, gws, None, (stride, stride), (0, 0), (1, 1), 1, wgS, None, 1.0, 1.0, auto[2])
			res.update(wgS=wgS, dsS=dsS, dbS=dbS)
		return {k: v.get() for k, v in res.items()}

	lazy.disabled = {"bnbwdfold"}
	(lz, taken), (lit, none) = both(run)
	assert taken.get("bn_apply_add", 0) == 1 and taken.get("gate_stats", 0) == 1 and taken.get("gate_by_mask", 0) == 1
	assert taken.get("bn_bwd_from_partials", 0) == (2 if projection else 1), "one fan-in pass serves both batch-norms"
	if projection and stride == 2:
		assert taken.get("compact_dgrad", 0) == 1
	for key in lit:
		assert np.array_equal(lz[key], lit[key]), "%s differs between the fused and the literal sequence" % key

	lazy.disabled = set()
	lazy.counters.clear()
	folded = run()
	expect = 0 if c % 16 else (2 if projection else 1)
	assert lazy.counters.get("dgrad_bn_fold", 0) == expect and lazy.counters.get("wgrad_bn_fold", 0) == expect
	for key in lit:
		scale = np.abs(lit[key]).max() + 1e-12
		assert_close(folded[key], lit[key], atol=3e-5 * scale, rtol=3e-4, what="BN backward folded into the convolution: " + key)

	# oracle: block output and the main branch's input gradient in fp64
	ca = R.conv2d_fwd(d["a"], d["wa"], None, stride=(1, 1), pad=(0, 0), dilation=(1, 1), groups=1, acc=np.float64)
	assert lit["out"].shape == ca.shape and (lit["out"] >= 0).all()
	assert np.array_equal(lit["grad"], (g0 + g1) * (lit["out"] > 0))


def test_gated_gradient_without_a_batchnorm_reader(surf):
	"""reluDerKer in place followed by a reader that is not the batch-norm backward (NiN: conv -> relu -> conv): the gate
	is applied by its own kernel when the convolution reads the gradient — same numbers as the literal sequence."""
	g, Dnn, El = surf.gpuarray, surf.Dnn, surf.ElementWise
	rng = np.random.RandomState(2)
	x, y = rng.randn(4, 8, 10, 10).astype(np.float32), rng.randn(4, 12, 10, 10).astype(np.float32)
	dy = rng.randn(4, 12, 10, 10).astype(np.float32)
	wt = rng.randn(12, 8, 3, 3).astype(np.float32)

	def run():
		gx, gy, gdy, gw = g.to_gpu(x), g.to_gpu(y), g.to_gpu(dy), g.to_gpu(wt)
		El.reluDerKer(np.float32)(gdy, gdy, gy)
		dx = Dnn.convNdBackwardData(gdy, gw, data=gx, stride=(1, 1), pad=(1, 1), dilation=(1, 1), groups=1, algo=Dnn.ConvBwdDataAlgo.auto)
		wg = g.zeros(wt.shape, dtype=np.float32)
		Dnn.convNdBackwardParams(gx, gdy, gw, None, (1, 1), (1, 1), (1, 1), 1, wg, None, 1.0, 1.0, Dnn.ConvBwdFilterAlgo.auto)
		return dx.get(), wg.get(), gdy.get()

	(lz, taken), (lit, _) = both(run)
	assert taken.get("gate", 0) == 1
	for a, b in zip(lz, lit):
		assert np.array_equal(a, b)
	assert np.array_equal(lz[2], dy * (y > 0))


def test_filter_gradients_on_the_side_stream(surf):
	"""Every convNdBackwardParams goes to the filter-gradient stream; whoever touches the gradient next (here: .get(), an
	axpy, the next accumulate call) is ordered behind it by the buffer's event. Same bits as on one stream, and the
	tensors the side stream reads may be overwritten or dropped right after the call."""
	from puzzlelib_amd import lazy
	g, Dnn, Blas = surf.gpuarray, surf.Dnn, surf.Blas
	rng = np.random.RandomState(4)
	x, dy = rng.randn(16, 64, 28, 28).astype(np.float32), rng.randn(16, 128, 28, 28).astype(np.float32)
	wt = rng.randn(128, 64, 3, 3).astype(np.float32)
	algo = Dnn.ConvBwdFilterAlgo.auto

	def run():
		gx, gdy, gw = g.to_gpu(x), g.to_gpu(dy), g.to_gpu(wt)
		arena = g.zeros((2 * wt.size, ), dtype=np.float32)            # two accumulators in one allocation, like the optimizer's
		first = arena[:wt.size].reshape(wt.shape)
		second = arena[wt.size:].reshape(wt.shape)
		Dnn.convNdBackwardParams(gx, gdy, gw, None, (1, 1), (1, 1), (1, 1), 1, first, None, 1.0, 1.0, algo)
		Dnn.convNdBackwardParams(gx, gdy, gw, None, (1, 1), (1, 1), (1, 1), 1, second, None, 0.5, 1.0, algo)
		gx.fill(0)                                                    # a main-stream write to what the side stream reads
		del gdy                                                       # ... and a free of the other operand
		Dnn.convNdBackwardParams(g.to_gpu(x), g.to_gpu(dy), gw, None, (1, 1), (1, 1), (1, 1), 1, first, None, 1.0, 1.0, algo)
		Blas.toVectorAddVector(arena, arena, alpha=1.0)               # main-stream read-modify-write of the arena
		return arena.get()

	surf.backend.dnn.sideWorkMean = 0.0           # (full-size tests before this one may have moved the policy to one stream)
	launches = surf.backend.dnn.sideLaunches
	with_side = run()
	# (the split math modes, and PUZZLE_MI355_SIDE_MAX_GFLOP=0, keep everything on one stream: backend.DnnContext.filterGradStream)
	expected = 3 if surf.backend.dnn.convMath == "f32" and surf.backend.dnn.sideStreamMaxGflop > 1.0 else 0
	assert surf.backend.dnn.sideLaunches == launches + expected
	lazy.disabled = {"sidestream"}
	one_stream = run()
	assert surf.backend.dnn.sideLaunches == launches + expected
	assert np.array_equal(with_side, one_stream)
	dw = R.conv2d_bwd_filter(x, dy, wt.shape, withbias=False, acc=np.float64, stride=(1, 1), pad=(1, 1), dilation=(1, 1), groups=1)
	scale = np.abs(dw).max()
	assert_close(with_side[:wt.size].reshape(wt.shape), 4.0 * dw, atol=2e-4 * scale, rtol=2e-4, what="2 x (dw + dw)")
	assert_close(with_side[wt.size:].reshape(wt.shape), 1.0 * dw, atol=2e-4 * scale, rtol=2e-4, what="2 x 0.5 dw")


def test_filter_gradient_follows_the_mark_in_front_of_backward_data_only_over_untouched_operands(surf):
	"""Modules/ConvND.py:84-95 calls backward-data, then backward-filter on the same gradient. On the filter-gradient stream the
	second has to wait for the producers of its operands, not for the first: it follows the main stream from a mark recorded
	in front of the backward-data launch (DnnContext.markBeforeBackwardData) — but only if neither operand was written since
	and its destination holds nothing pending. Values against the fp64 oracle in every case (the test allocator poisons the
	workspace with NaNs on whichever stream the launch runs on), bit-identical to the one-stream run."""
	from puzzlelib_amd import lazy
	g, Dnn = surf.gpuarray, surf.Dnn
	dnn = surf.backend.dnn
	if dnn.convMath != "f32" or dnn.sideStreamMaxGflop <= 1.0:
		pytest.skip("one stream by configuration")
	rng = np.random.RandomState(11)
	n, c, k, h = 32, 96, 160, 16
	x, dy = rng.randn(n, c, h, h).astype(np.float32), rng.randn(n, k, h, h).astype(np.float32)
	x2, dy2 = rng.randn(*x.shape).astype(np.float32), rng.randn(*dy.shape).astype(np.float32)
	wt = (rng.randn(k, c, 5, 5) / 50.0).astype(np.float32)
	kw = dict(stride=(1, 1), pad=(2, 2), dilation=(1, 1), groups=1)
	args = ((1, 1), (2, 2), (1, 1))

	def run(between):
		gx, gdy, gw = g.to_gpu(x), g.to_gpu(dy), g.to_gpu(wt)
		wg, bg = g.zeros(wt.shape, dtype=np.float32), g.zeros((k, ), dtype=np.float32)
		wg.get(), bg.get()                                             # (the zero fills are written, not pending)
		dx = Dnn.convNdBackwardData(gdy, gw, gx, *args, 1, Dnn.ConvBwdDataAlgo.auto)
		between(gx, gdy)
		Dnn.convNdBackwardParams(gx, gdy, gw, bg, *args, 1, wg, bg, 1.0, 0.0, Dnn.ConvBwdFilterAlgo.auto)
		return dx.get(), wg.get(), bg.get()

	cases = {
		"untouched": (lambda gx, gdy: None, x, dy, 1),
		"gradient rewritten": (lambda gx, gdy: gdy.set(dy2), x, dy2, 0),
		"input rewritten": (lambda gx, gdy: gx.set(x2), x2, dy, 0),
	}
	for name, (between, xr, dyr, marks) in cases.items():
		dnn.sideWorkMean = 0.0
		lazy.disabled = set()
		lazy.counters.clear()
		dx, dw, db = run(between)
		assert lazy.counters.get("wgrad_early_start", 0) == marks, name
		lazy.disabled = {"sidestream"}
		dx1, dw1, db1 = run(between)
		lazy.disabled = set()
		assert np.array_equal(dx, dx1) and np.array_equal(dw, dw1) and np.array_equal(db, db1), name
		dw_ref, db_ref = R.conv2d_bwd_filter(xr, dyr, wt.shape, withbias=True, acc=np.float64, **kw)
		scale = np.sqrt(n * h * h)
		assert_close(dw, dw_ref, atol=1e-5 * scale, rtol=1e-4, what=name + ": filter gradient")
		assert_close(db, db_ref, atol=1e-5 * scale, rtol=1e-4, what=name + ": bias gradient")
		assert_close(dx, R.conv2d_bwd_data(dy, wt, x.shape, acc=np.float64, **kw), atol=1e-4, rtol=1e-4, what=name + ": input gradient")


def test_borrowed_streams_are_ordered_by_the_buffers(surf):
	"""Optimizer.update(useStreams=True) (Optimizers/Optimizer.py:176-196): per-parameter updates on borrowed streams. The
	reference's streams are blocking ones; here ordering is carried by events on the buffers (updates raced
	with backward and the next forward). Three SGD steps on several parameters, useStreams on/off, same bits."""
	from puzzlelib_amd import nets, optim
	g = surf.gpuarray
	spec = [("conv", "c1", 3, 16, 3, 1, 1, True), ("relu", "r1"), ("conv", "c2", 16, 16, 3, 1, 1, True), ("relu", "r2"),
			("maxpool", "p", 2, 2, 0), ("flatten", "f"), ("linear", "fc", 16 * 8 * 8, 10)]
	rng = np.random.RandomState(6)
	data, labels = rng.randn(8, 3, 16, 16).astype(np.float32), rng.randint(0, 10, size=(8, )).astype(np.int32)

	results = []
	for streams in (False, True):
		np.random.seed(3)
		net = nets.build(spec, initscheme="he")
		opt = optim.MomentumSGD(learnRate=0.05, momRate=0.9)
		opt.setupOn(net, useGlobalState=False)
		cost = optim.CrossEntropy()
		net.trainMode()
		gd, gl = g.to_gpu(data), g.to_gpu(labels)
		for _ in range(3):
			grad = cost(net(gd), gl, queryError=False)
			opt.zeroGradParams()
			net.backward(grad, updGrad=False)
			opt.update(useStreams=streams, sync=False)
			net.reset()
		results.append({k: p.data.get() for k, p in net.namedParams().items()})
	for key in results[0]:
		assert np.array_equal(results[0][key], results[1][key]), key


@pytest.mark.parametrize("shape,size,stride,pad", [((5, 7, 21, 19), 3, 2, 0), ((3, 4, 16, 16), 2, 2, 0), ((2, 6, 13, 13), 3, 1, 1),
												   ((2, 64, 112, 112), 3, 2, 0)])
def test_max_pooling_normalises_a_described_batchnorm_on_the_fly(surf, shape, size, stride, pad):
	"""BatchNorm -> in-place ReLU -> MaxPool (the ResNet stem, Models/Nets/ResNet.py:88-96): the pooling reads the BN's
	input and applies the affine pair + ReLU while staging its rows; output and arg-max bytes equal the unfused sequence
	bit for bit, the normalised tensor is never written, and the backward pooling does not need it either."""
	from puzzlelib_amd import lazy
	g, Dnn, f32 = surf.gpuarray, surf.Dnn, np.float32
	rng = np.random.RandomState(11)
	x = rng.randn(*shape).astype(f32)
	c = shape[1]
	sc, bi = (1.0 + 0.2 * rng.randn(1, c, 1, 1)).astype(f32), (0.3 * rng.randn(1, c, 1, 1)).astype(f32)
	dy = None

	def run():
		nonlocal dy
		gx = g.to_gpu(x)
		mean, var = g.zeros((1, c, 1, 1), f32), g.to_gpu(np.ones((1, c, 1, 1), f32))
		y, _, _ = Dnn.batchNormNd(gx, g.to_gpu(sc), g.to_gpu(bi), mean, var, 1e-5, 1.0, False)
		surf.ElementWise.reluKer(f32)(y, y)
		before = dict(lazy.counters)
		pooled, ws = Dnn.poolNd(y, size, stride, pad, Dnn.PoolMode.max, False)
		fusedNow = lazy.counters.get("bn_pool", 0) - before.get("bn_pool", 0)
		if dy is None:
			dy = rng.randn(*pooled.shape).astype(f32)
		applied = lazy.counters.get("bn_apply_relu", 0)
		dx = Dnn.poolNdBackward(y, pooled, g.to_gpu(dy), ws, size, stride, pad, Dnn.PoolMode.max)
		wroteLater = lazy.counters.get("bn_apply_relu", 0) - applied
		return pooled.get(), np.array(ws.get()), dx.get(), y.get(), fusedNow, wroteLater

	fused = run()
	lazy.disabled = {"bnpool"}
	try:
		plain = run()
	finally:
		lazy.disabled = set()

	assert fused[4] == 1 and fused[5] == 0 and plain[4] == 0
	for a, b, what in zip(fused[:4], plain[:4], ("pooled", "arg-max bytes", "input gradient", "normalised tensor read afterwards")):
		assert np.array_equal(a, b), what
	ref = np.maximum(plain[3], 0)
	assert np.array_equal(ref, plain[3])


def test_described_tensors_captured_by_reference_keep_their_value(surf):
	"""(1) overwriting a described tensor entirely must first serve a tensor that refers to it —
	B = 0 + A while A is a pending zero fill, then A.set(ones): B is zeros, not unwritten memory; (2) an in-place ReLU (or
	a further term) joining a description changes the tensor: an accumulator that took it by reference keeps the old value."""
	g, Dnn, El, Blas = surf.gpuarray, surf.Dnn, surf.ElementWise, surf.Blas
	from puzzlelib_amd import lazy, fusion
	rng = np.random.RandomState(21)
	shape = (4, 6, 64)

	A = g.empty(shape, dtype=np.float32)               # (NaN-poisoned by the test allocator)
	A.fill(0)
	B = g.empty(shape, dtype=np.float32)
	B.fill(0)
	Blas.toVectorAddVector(B.ravel(), A.ravel())
	assert isinstance(lazy.pending(B), fusion.Sum)
	A.set(np.ones(shape, np.float32))
	assert np.array_equal(B.get(), np.zeros(shape, np.float32)) and np.array_equal(A.get(), np.ones(shape, np.float32))

	# 3-d BatchNorm output (its description is taken by reference into a sum), then reluKer(y, y)
	x = rng.randn(*shape).astype(np.float32)
	(scale, bias, rm, rv), fresh = bnParams(surf, rng, 6)
	gs, gb, grm, grv = fresh()
	y, sm, si = Dnn.batchNormNd(g.to_gpu(x), gs, gb, grm, grv, 1e-5, 1.0, False)
	acc = g.empty(shape, dtype=np.float32)
	acc.fill(0)
	Blas.toVectorAddVector(acc.ravel(), y.ravel())
	El.reluKer(np.float32)(y, y)
	y_ref = R.bn_fwd_train(x.reshape(4, 6, 64, 1), scale.ravel(), bias.ravel(), rm.ravel().copy(), rv.ravel().copy(), 1e-5, 1.0)[0]
	y_ref = y_ref.reshape(shape)
	assert_close(acc.get(), y_ref, atol=2e-5, rtol=1e-4, what="acc = bn(x), taken before the ReLU")
	assert (acc.get() < 0).any()
	assert_close(y.get(), np.maximum(y_ref, 0), atol=2e-5, rtol=1e-4, what="y = relu(bn(x))")

	# a sum that gets a further term after somebody referred to it
	a, b = rng.randn(*shape).astype(np.float32), rng.randn(*shape).astype(np.float32)
	ga, gb2 = g.to_gpu(a), g.to_gpu(b)
	s1 = g.empty(shape, dtype=np.float32)
	s1.fill(0)
	Blas.toVectorAddVector(s1.ravel(), ga.ravel())
	s2 = g.empty(shape, dtype=np.float32)
	s2.fill(0)
	Blas.toVectorAddVector(s2.ravel(), s1.ravel())
	Blas.toVectorAddVector(s1.ravel(), gb2.ravel())
	assert np.array_equal(s2.get(), a) and np.array_equal(s1.get(), a + b)

	# a gate joining a described fan-in that somebody referred to
	gy = g.to_gpu(rng.randn(*shape).astype(np.float32))
	fan = g.empty(shape, dtype=np.float32)
	fan.fill(0)
	Blas.toVectorAddVector(fan.ravel(), ga.ravel())
	Blas.toVectorAddVector(fan.ravel(), gb2.ravel())
	keep = g.empty(shape, dtype=np.float32)
	keep.fill(0)
	Blas.toVectorAddVector(keep.ravel(), fan.ravel())
	El.reluDerKer(np.float32)(fan, fan, gy)
	assert np.array_equal(keep.get(), a + b) and np.array_equal(fan.get(), (a + b) * (gy.get() > 0))


@pytest.mark.parametrize("policy", ["adaptive", "always"])
def test_conv_statistics_only_serve_a_batchnorm_over_the_very_tensor(surf, policy):
	"""the strip sums a convolution's epilogue leaves are per-channel sums of its (n, k, p, q) output.
	A BatchNorm over a reshape (Conv -> Flatten-like regrouping of channels) or over a slice of that output must compute
	its own statistics — two passes, so that the adaptive policy is armed in the second; all three against the oracle."""
	from puzzlelib_amd import backend, lazy
	g, Dnn = surf.gpuarray, surf.Dnn
	backend.DnnContext.convStatsPolicy = policy
	rng = np.random.RandomState(8)
	x = rng.randn(6, 8, 12, 12).astype(np.float32)
	w = (0.2 * rng.randn(16, 8, 3, 3)).astype(np.float32)
	gx, gw = g.to_gpu(x), g.to_gpu(w)
	conv_ref = R.conv2d_fwd(x, w, None, 1, 1)

	def bn(view, ref, c, what):
		(scale, bias, rm, rv), fresh = bnParams(surf, rng, c)
		gs, gb, grm, grv = fresh()
		y, sm, si = Dnn.batchNormNd(view, gs, gb, grm, grv, 1e-5, 0.5, False)
		rm_ref, rv_ref = rm.ravel().copy(), rv.ravel().copy()
		y_ref, sm_ref, si_ref = R.bn_fwd_train(ref, scale.ravel(), bias.ravel(), rm_ref, rv_ref, 1e-5, 0.5, acc=np.float64)
		assert_close(y.get(), y_ref, atol=1e-4, rtol=1e-4, what=what + ": y")
		assert_close(sm.get().ravel(), sm_ref.ravel(), atol=1e-5, rtol=1e-4, what=what + ": saved mean")
		assert_close(si.get().ravel(), si_ref.ravel(), atol=1e-5, rtol=1e-4, what=what + ": saved inverse deviation")
		assert_close(grv.get().ravel(), rv_ref, atol=1e-5, rtol=1e-4, what=what + ": running variance")

	for attempt in range(2):
		lazy.counters.clear()
		bn(Dnn.convNd(gx, gw, None, 1, 1, 1, 1, Dnn.ConvFwdAlgo.auto), conv_ref, 16, "pass %d conv -> bn" % attempt)
		bn(Dnn.convNd(gx, gw, None, 1, 1, 1, 1, Dnn.ConvFwdAlgo.auto).reshape(6, 32, 6, 12), conv_ref.reshape(6, 32, 6, 12), 32,
		   "pass %d conv -> reshape -> bn" % attempt)
		bn(Dnn.convNd(gx, gw, None, 1, 1, 1, 1, Dnn.ConvFwdAlgo.auto)[:3], conv_ref[:3], 16, "pass %d conv -> slice -> bn" % attempt)
		if attempt == 1 or policy == "always":
			assert lazy.counters.get("conv_stats", 0) == 3, "the convolutions did leave strip sums (only the first BN may use them)"


def test_prepared_filter_operands_follow_every_write_of_the_filter(surf):
	"""DnnContext.prepared: packed / Winograd-transformed filters are kept per (filter, pass, geometry) and must be
	re-derived whenever the filter's allocation was written — by .set(), through a slice view, in place by a kernel, by a
	kernel on a borrowed stream. Implicit-GEMM forward (1x1, 3-map stem-like), Winograd forward and backward-data."""
	from puzzlelib_amd import lazy
	g, Dnn, El = surf.gpuarray, surf.Dnn, surf.ElementWise
	rng = np.random.RandomState(12)
	cases = [((4, 32, 9, 9), (48, 32, 1, 1), 1, 0), ((4, 3, 20, 20), (16, 3, 5, 5), 2, 2), ((4, 32, 10, 10), (32, 32, 3, 3), 1, 1)]
	auto = (Dnn.ConvFwdAlgo.auto, Dnn.ConvBwdDataAlgo.auto)
	for xshape, wshape, stride, pad in cases:
		x = rng.randn(*xshape).astype(np.float32)
		gx = g.to_gpu(x)
		arena = g.to_gpu(rng.randn(int(np.prod(wshape)) + 64).astype(np.float32))            # the filter is a VIEW into an arena
		gw = arena[32:32 + int(np.prod(wshape))].reshape(wshape)

		def check(what):
			lazy.counters.clear()
			w = gw.get()
			y = Dnn.convNd(gx, gw, None, stride, pad, 1, 1, auto[0])
			ref = R.conv2d_fwd(x, w, None, stride, pad, acc=np.float64)
			assert_close(y.get(), ref, atol=2e-4 * np.abs(ref).max(), rtol=1e-4, what="%s %s: forward" % (wshape, what))
			dy = rng.randn(*ref.shape).astype(np.float32)
			dx = Dnn.convNdBackwardData(g.to_gpu(dy), gw, gx, stride, pad, 1, 1, auto[1])
			dref = R.conv2d_bwd_data(dy, w, x.shape, stride, pad, acc=np.float64)
			assert_close(dx.get(), dref, atol=2e-4 * np.abs(dref).max(), rtol=1e-4, what="%s %s: backward data" % (wshape, what))
			return lazy.counters.get("prepack_launch", 0)

		# (the split math modes prepare the implicit GEMM's operands inside every call: nothing to count there, results only)
		least = 1 if (surf.backend.dnn.convMath == "f32" or wshape[2:] == (3, 3)) else 0
		assert check("first use") >= least
		assert check("unchanged") == 0, "nothing is prepared again while the filter stands"
		gw.set(rng.randn(*wshape).astype(np.float32))
		assert check("after .set()") >= least
		gw[1:2].set(rng.randn(1, *wshape[1:]).astype(np.float32))
		assert check("after a write through a slice") >= least
		El.linearKer(np.float32)(gw, gw, 0.5, 0.25)
		assert check("after an in-place kernel") >= least
		stream = g.streamManager.borrow(1)[0]
		El.toVectorAddVectorKer(np.float32)(arena, g.to_gpu(rng.randn(arena.size).astype(np.float32)), 1.0, stream=stream)
		assert check("after a kernel on a borrowed stream wrote the arena") >= least
		g.streamManager.give([stream])


@pytest.mark.parametrize("inplace", [False, True])
def test_relu_and_its_gate_ride_in_the_convolution_epilogues(surf, inplace):
	"""Conv2D(bias) -> Activation(relu) (TestLib/CnnCifar10NIN.py:16-45; Modules/Activation.py:52-60, out of place by default):
	the forward launch takes the ReLU into its epilogue, the next layer's backward-data launch takes the ReLU's gate. Bit-identical
	to the literal sequence (same products, `x * (x > 0)` / `g * (y > 0)` as the element-wise kernels compute them); the
	convolution's own (pre-activation) output and the ungated gradient stay readable and keep the values of the call's time even
	after the parameters were updated."""
	from puzzlelib_amd import lazy
	g, Dnn, El = surf.gpuarray, surf.Dnn, surf.ElementWise
	rng = np.random.RandomState(41)
	x = rng.randn(5, 24, 9, 11).astype(np.float32)
	w1, b1 = (rng.randn(40, 24, 3, 3) / 15).astype(np.float32), rng.randn(1, 40, 1, 1).astype(np.float32)
	w2, b2 = (rng.randn(16, 40, 1, 1) / 6).astype(np.float32), rng.randn(1, 16, 1, 1).astype(np.float32)
	dy2 = rng.randn(5, 16, 9, 11).astype(np.float32)
	igemm = Dnn.ConvFwdAlgo.implicitGemm, Dnn.ConvBwdDataAlgo.implicitGemm

	def run(fused):
		lazy.enabled = fused
		lazy.counters.clear()
		arena = g.to_gpu(np.concatenate([w1.ravel(), b1.ravel(), w2.ravel(), b2.ravel()]))
		cuts = np.cumsum([0, w1.size, b1.size, w2.size, b2.size])
		gw1, gb1, gw2, gb2 = (arena[cuts[i]:cuts[i + 1]].reshape(t.shape) for i, t in enumerate((w1, b1, w2, b2)))
		gx = g.to_gpu(x)
		c1 = Dnn.convNd(gx, gw1, gb1, 1, 1, 1, 1, igemm[0])
		y1 = c1 if inplace else g.empty(c1.shape, dtype=np.float32)
		El.reluKer(np.float32)(y1, c1)
		c2 = Dnn.convNd(y1, gw2, gb2, 1, 0, 1, 1, igemm[0])
		# backward of the second convolution, then of the ReLU
		d1 = Dnn.convNdBackwardData(g.to_gpu(dy2), gw2, y1, 1, 0, 1, 1, igemm[1])
		g1 = d1 if inplace else g.empty(d1.shape, dtype=np.float32)
		El.reluDerKer(np.float32)(g1, d1, y1)
		counts = dict(lazy.counters)
		# the optimizer touches the parameters while the pre-activation tensors are still only described
		El.linearKer(np.float32)(arena, arena, 0.5, 0.125)
		out = {"y1": y1.get(), "c2": c2.get(), "g1": g1.get()}
		if not inplace:
			out["c1"], out["d1"] = c1.get(), d1.get()
		return out, counts, dict(lazy.counters)

	try:
		fused, taken, after = run(True)
		literal, _, _ = run(False)
	finally:
		lazy.enabled = True
	assert after.get("conv_relu", 0) == 1 and after.get("dgrad_gate", 0) == 1 and "relu" not in after and "gate" not in after, after
	if inplace:
		assert not taken, "in place everything up to the optimizer is only described: %s" % taken
	else:
		# out of place the activated launches run at once; ONE copy of the parameter arena serves both detached descriptions;
		# c1, c2 and d1 are materialised by .get() only (from the copy: the arena has been updated by then)
		assert taken.get("conv_relu", 0) == 1 and taken.get("dgrad_gate", 0) == 1 and taken.get("param_snapshot", 0) == 1, taken
		assert after.get("conv_deferred", 0) == 2 and after.get("dgrad_deferred", 0) == 1, after
	for key in literal:
		assert np.array_equal(fused[key], literal[key]), "%s differs from the literal call sequence" % key
	ref = R.conv2d_fwd(x, w1, b1.ravel(), 1, 1, acc=np.float64)
	assert_close(fused["y1"], R.relu(ref), atol=1e-4, rtol=1e-4, what="relu(conv + bias)")


def test_described_convolutions_survive_writes_to_what_they_read(surf):
	"""A convolution that waits for its ReLU (fusion.ConvFwd) or a backward-data launch that waits for its gate
	(fusion.ConvBwdData) is a tensor captured by reference: overwriting ANY of its operands — input, filter, bias, incoming
	gradient, gate — before it ran must leave it with the values of the call's time."""
	from puzzlelib_amd import lazy
	g, Dnn, El = surf.gpuarray, surf.Dnn, surf.ElementWise
	rng = np.random.RandomState(43)
	x = rng.randn(3, 16, 7, 9).astype(np.float32)
	w, b = (rng.randn(24, 16, 1, 1) / 4).astype(np.float32), rng.randn(1, 24, 1, 1).astype(np.float32)
	dy = rng.randn(3, 24, 7, 9).astype(np.float32)
	algo = Dnn.ConvFwdAlgo.implicitGemm, Dnn.ConvBwdDataAlgo.implicitGemm
	ref = R.conv2d_fwd(x, w, b.ravel(), 1, 0, acc=np.float64)
	other = lambda a: g.to_gpu(rng.randn(*a.shape).astype(np.float32))

	for victim in ("x", "w", "b"):
		for relu in (False, True):
			gx, gw, gb = g.to_gpu(x), g.to_gpu(w), g.to_gpu(b)
			c = Dnn.convNd(gx, gw, gb, 1, 0, 1, 1, algo[0])
			assert lazy.pending(c) is not None, "the launch waits for a ReLU that may follow"
			y = g.empty(c.shape, dtype=np.float32)
			if relu:
				El.reluKer(np.float32)(y, c)                         # runs the activated launch, detaches c onto a snapshot
			{"x": gx, "w": gw, "b": gb}[victim].set(other({"x": x, "w": w, "b": b}[victim]).get())
			assert_close(c.get(), ref, atol=1e-4, rtol=1e-4, what="pre-activation after %s was overwritten (relu taken: %s)" % (victim, relu))
			if relu:
				assert_close(y.get(), R.relu(ref), atol=1e-4, rtol=1e-4, what="activated output")

	# backward: the gate tensor and the incoming gradient are overwritten while the launch is still only described
	dref = R.conv2d_bwd_data(dy, w, x.shape, 1, 0, acc=np.float64)
	yact = R.relu(ref).astype(np.float32)[:, :16]             # any tensor of dx's shape serves as a gate
	for victim in ("dy", "gate", "w"):
		gw, gdy, gate = g.to_gpu(w), g.to_gpu(dy), g.to_gpu(np.ascontiguousarray(yact))
		lazy.setFact(gate, "convrelu", True)                      # as a fused forward launch leaves it
		d = Dnn.convNdBackwardData(gdy, gw, gate, 1, 0, 1, 1, algo[1])
		assert lazy.pending(d) is not None
		El.reluDerKer(np.float32)(d, d, gate)                     # in place: the gate joins the description
		assert lazy.pending(d) is not None
		{"dy": gdy, "gate": gate, "w": gw}[victim].set(other({"dy": dy, "gate": yact, "w": w}[victim]).get())
		assert_close(d.get(), dref * (yact > 0), atol=1e-4, rtol=1e-4, what="gated input gradient after %s was overwritten" % victim)


@pytest.mark.parametrize("rule", ["adam", "classicMomSGD", "nesterovMomSGD"])
def test_gradient_mean_rides_in_the_update_kernel(surf, rule):
	"""The data-parallel mean (Optimizers/Optimizer.py:166-167 -> Grid.py:126-133 sums and divides in one pass): the exchange
	leaves the arena described as `sum x 1/N` (fusion.Scaled) and the Adam / momentum-SGD kernels take the factor as a scalar.
	Bit-identical to scaling first; the description stays valid for whoever reads the gradients afterwards; a hook that
	touches the gradients before the update makes it run as the linear pass it stands for."""
	from puzzlelib_amd import lazy, fusion
	g, El = surf.gpuarray, surf.ElementWise
	rng = np.random.RandomState(31)
	n, scale = 100003, 0.125 + 1.0 / 3.0
	host = {k: rng.randn(n).astype(np.float32) for k in ("param", "grad", "mg")}
	host["ms"] = np.abs(rng.randn(n)).astype(np.float32)

	def update(param, grad, mg, ms):
		if rule == "adam":
			El.adamKer(np.float32)(param, grad, mg, ms, 1e-3, 0.1, 0.001, 1e-8)
		elif rule == "classicMomSGD":
			El.classicMomSGDKer(np.float32)(param, grad, mg, 0.01, 0.9)
		else:
			El.nesterovMomSGDKer(np.float32)(param, grad, mg, 0.01, 0.9)

	a = {k: g.to_gpu(v) for k, v in host.items()}
	El.linearKer(np.float32)(a["grad"], a["grad"], scale, 0.0)
	update(a["param"], a["grad"], a["mg"], a["ms"])

	b = {k: g.to_gpu(v) for k, v in host.items()}
	lazy.counters.clear()
	lazy.attach(b["grad"], fusion.Scaled(scale))
	update(b["param"], b["grad"], b["mg"], b["ms"])
	assert lazy.counters.get("grad_scale_folded", 0) == 1 and lazy.counters.get("grad_scale_pass", 0) == 0
	for key in ("param", "mg", "ms"):
		assert np.array_equal(a[key].get(), b[key].get()), "%s: %s differs from scaling first" % (rule, key)
	assert np.array_equal(b["grad"].get(), a["grad"].get()), "the described gradient reads as the scaled one"
	assert lazy.counters.get("grad_scale_pass", 0) == 1

	c = {k: g.to_gpu(v) for k, v in host.items()}
	lazy.counters.clear()
	lazy.attach(c["grad"], fusion.Scaled(scale))
	El.weightDecayKer(c["grad"], c["param"], 0.0)             # a hook touching the gradient first (rate 0: value unchanged)
	update(c["param"], c["grad"], c["mg"], c["ms"])
	assert lazy.counters.get("grad_scale_pass", 0) == 1 and lazy.counters.get("grad_scale_folded", 0) == 0
	assert np.array_equal(a["param"].get(), c["param"].get())


def test_prepared_filter_operands_follow_the_library_modes(surf):
	"""The layout of a prepared operand depends on process-wide modes of the library (Winograd output tile, math mode). Whoever
	changes them — here: a direct call of the C entry point, the way tools and other contexts do — must not leave this
	context consuming operands prepared under the old mode (round-3 advisor finding): lib.modeEpoch advances, the caches
	start over, results stay right."""
	from puzzlelib_amd import lazy, lib
	g, Dnn = surf.gpuarray, surf.Dnn
	rng = np.random.RandomState(21)
	x = rng.randn(4, 32, 12, 12).astype(np.float32)
	w = (rng.randn(32, 32, 3, 3) / 17).astype(np.float32)
	gx, gw = g.to_gpu(x), g.to_gpu(w)
	ref = R.conv2d_fwd(x, w, None, 1, 1, acc=np.float64)
	try:
		for tile in (4, 2, 4, 0):
			lib.pz_conv_winograd_tile_set(tile)                  # behind the context's back
			lazy.counters.clear()
			y = Dnn.convNd(gx, gw, None, 1, 1, 1, 1, Dnn.ConvFwdAlgo.auto)
			assert lazy.counters.get("prepack_launch", 0) == 1, "tile %d: operands of the previous mode were reused" % tile
			assert_close(y.get(), ref, atol=6e-5 * np.abs(ref).max(), rtol=0, what="forward after the tile changed to %d" % tile)
	finally:
		lib.pz_conv_winograd_tile_set(0)


def test_training_steps_with_prepared_operands_equal_per_call_packing(surf, mini_golden):
	"""Three Adam steps of the mini-ResNet: parameters bit-identical whether the filter operands are prepared once per
	step in batched launches or packed inside every convolution call (lazy.disabled = {"prepack"})."""
	from puzzlelib_amd import nets, optim, lazy
	g = surf.gpuarray
	spec = nets.resnet_spec(stages=((8, 1), (16, 2)), classes=10, stem=8, softmax=False)
	spec = [l if l[0] != "avgpool" else ("avgpool", l[1], 8, 1, 0) for l in spec]
	data, labels = g.to_gpu(mini_golden["data"]), g.to_gpu(mini_golden["labels"])

	def run(prepack):
		lazy.disabled = set() if prepack else {"prepack"}
		lazy.counters.clear()
		np.random.seed(77)
		net = nets.build(spec, name="mini", initscheme="he", actInplace=True)
		opt = optim.Adam(alpha=1e-3)
		opt.setupOn(net, useGlobalState=True)
		trainer = optim.Trainer(net, optim.CrossEntropy(), opt, batchsize=4)
		for _ in range(3):
			trainer.train(data, labels, random=False)
		return {k: p.data.get() for k, p in net.namedParams().items()}, lazy.counters.get("prepack_launch", 0)

	try:
		with_, launches = run(True)
		without, none = run(False)
	finally:
		lazy.disabled = set()
	assert none == 0 and launches >= 3
	for key in with_:
		assert np.array_equal(with_[key], without[key]), key


@pytest.mark.parametrize("cfg", [dict(n=8, c=64, hw=(55, 55), k=256), dict(n=6, c=128, hw=(28, 28), k=512), dict(n=5, c=256, hw=(14, 14), k=1024),
								 dict(n=3, c=48, hw=(9, 11), k=136), dict(n=4, c=512, hw=(7, 7), k=2048), dict(n=3, c=24, hw=(6, 6), k=32)])
def test_pointwise_convolution_reads_a_batchnorm_relu_that_was_never_written(surf, cfg):
	"""conv -> BatchNorm2D -> Activation(relu, inplace) -> Conv2D 1x1 -> BatchNorm2D and back, as the reference's bottleneck
	blocks issue it (Models/Nets/ResNet.py:27-33): the pointwise layer's forward and its filter gradient evaluate
	relu(a * x + b) while they gather (pz_conv2d_fwd_xbn / pz_conv2d_bwd_filter_xbn, the latter together with the fold of
	the FOLLOWING BatchNorm's backward) and the normalised tensor is never written — it still reads correctly afterwards.
	Bit-identical to the same calls with the fold off (the gathers use bn_apply's own expression). Layers the kernels do not
	take (512 channels: the two-tiles-ahead implicit GEMM; 24 channels: not whole k-tiles) write the tensor and agree too."""
	from puzzlelib_amd import lazy
	g, Dnn, El = surf.gpuarray, surf.Dnn, surf.ElementWise
	n, c, (h, w), k = cfg["n"], cfg["c"], cfg["hw"], cfg["k"]
	rng = np.random.RandomState(5)
	x = (0.3 + rng.randn(n, c, h, w)).astype(np.float32)
	W = (rng.randn(k, c, 1, 1) / np.sqrt(c)).astype(np.float32)
	dz = rng.randn(n, k, h, w).astype(np.float32)
	(_, _, _, _), fresh = bnParams(surf, rng, c)
	(_, _, _, _), fresh2 = bnParams(surf, rng, k)

	def run():
		gs, gb, grm, grv = fresh()
		gs2, gb2, grm2, grv2 = fresh2()
		gx, gW, gdz = g.to_gpu(x), g.to_gpu(W), g.to_gpu(dz)
		y, sm, si = Dnn.batchNormNd(gx, gs, gb, grm, grv, 1e-5, 0.3, False)
		El.reluKer(np.float32)(y, y)
		algo = surf.backend.ConvFwdAlgo.auto
		z = Dnn.convNd(y, gW, None, 1, 0, 1, 1, algo)
		u, sm2, si2 = Dnn.batchNormNd(z, gs2, gb2, grm2, grv2, 1e-5, 0.3, False)
		# backward: the following BatchNorm's input gradient (described), then this layer's two backward passes
		dzbn, ds2, db2 = Dnn.batchNormNdBackward(z, gdz, gs2, sm2, si2, 1e-5)
		dy = Dnn.convNdBackwardData(dzbn, gW, y, 1, 0, 1, 1, surf.backend.ConvBwdDataAlgo.auto)
		wgrad = g.zeros(W.shape, dtype=np.float32)
		Dnn.convNdBackwardParams(y, dzbn, gW, None, 1, 0, 1, 1, wgrad, None, 1.0, 0.0, surf.backend.ConvBwdFilterAlgo.auto)
		unwritten = lazy.pending(y) is not None
		return [a.get() for a in (z, u, dy, wgrad, y)], unwritten

	lazy.counters.clear()
	folded, unwritten = run()
	taken = dict(lazy.counters)
	lazy.disabled = {"xbn"}
	lazy.counters.clear()
	plain, _ = run()
	lazy.disabled = set()

	dnn = surf.backend.dnn
	desc = dnn.convDesc((n, c, h, w), W.shape, 1, 0, 1, 1)
	from puzzlelib_amd import lib
	takes = dnn.xbnSupported(desc, lib.CONV_FWD, lib.CONV_ALGO_AUTO)
	assert takes == (cfg["k"] in (256, 512, 1024)), "the three bottleneck shapes fold; 64-row tiles, the two-tiles-ahead form and 24 channels do not"
	assert (taken.get("conv_xbn", 0), taken.get("wgrad_xbn", 0)) == ((1, 1) if takes else (0, 0)), taken
	assert unwritten == takes, "the normalised tensor must stay a description through both passes"
	for a, b, what in zip(folded, plain, ("z", "bn(z)", "dy", "dW", "relu(bn(x)) read afterwards")):
		assert np.array_equal(a, b), what

	z_ref = R.conv2d_fwd(folded[4], W, None, acc=np.float64)
	assert_close(folded[0], z_ref, atol=1e-4 * np.abs(z_ref).max(), rtol=1e-4, what="z vs oracle on the device's own relu(bn(x))")
