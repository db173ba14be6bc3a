"""
The split math modes of the MFMA kernels (backend.DnnContext.setConvMath, include/puzzle_mi355.h pz_conv_math_set):
every fp32 operand is split exactly into three bf16 terms and multiplied as 6 / 9 partial products on
v_mfma_f32_32x32x16_bf16 with fp32 accumulation. Checked here: (1) the same float64 oracle and the same tolerances as the
fp32-MFMA kernels on shapes that reach every split kernel (tap-major implicit GEMM forward / backward-data, their
BatchNorm-folding variants, the pointwise backward-filter kernel incl. ragged image planes and split-K), (2) the error
against float64 is not larger than the fp32 MFMA's, (3) repeatability, (4) the switch itself.
The rest of the GPU suite can be run in a split mode as a whole: PUZZLE_MI355_MATH=split6 pytest -m gpu.
"""
import numpy as np
import pytest

import cpu_ref as R
from conftest import assert_close

pytestmark = pytest.mark.gpu

MODES = ("split6", "split9")

# (n, c, h, w, k, r, stride, pad): channels in whole k-tiles (tap-major) so that the split implicit GEMM takes them;
# >= 128 channels on both sides for the split backward-filter kernel; odd planes (ragged last cell); 64-row tiles;
# a 3x3 through the implicit GEMM; stride 2
SHAPES = [
	(8, 256, 14, 14, 128, 1, 1, 0),
	(4, 128, 7, 7, 512, 1, 1, 0),
	(3, 160, 5, 9, 144, 1, 1, 0),
	(6, 64, 13, 11, 64, 1, 1, 0),
	(4, 128, 15, 15, 256, 1, 2, 0),
	(2, 32, 12, 12, 48, 3, 1, 1),
	(2, 48, 9, 9, 32, 3, 2, 1),
]


@pytest.fixture
def dnn(bnd):
	yield bnd.dnn
	bnd.dnn.setConvMath(bnd.dnn.convMathDefault)


def passes(bnd, x, w, dy, stride, pad, algo):
	G = bnd.GPUArray
	gx, gw, gdy = G.toGpu(x), G.toGpu(w), G.toGpu(dy)
	y = bnd.dnn.convNd(gx, gw, None, stride, pad, algo=algo).get()
	dx = bnd.dnn.convNdBackwardData(gdy, gw, None, gx, stride, pad, algo=algo).get()
	dw = bnd.dnn.convNdBackwardParams(gx, gdy, gw, stride, pad, algo=algo).get()
	return y, dx, dw


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "n%dc%dh%dw%dk%dr%ds%dp%d" % s)
def test_split_modes_match_the_oracle_like_the_f32_mfma(bnd, dnn, shape):
	n, c, h, w_, k, r, stride, pad = shape
	rng = np.random.RandomState(7)
	x = rng.randn(n, c, h, w_).astype(np.float32)
	w = (rng.randn(k, c, r, r) / np.sqrt(c * r * r)).astype(np.float32)
	kw = dict(stride=stride, pad=pad, dilation=1, groups=1)
	y_ref = R.conv2d_fwd(x, w, None, acc=np.float64, **kw)
	dy = rng.randn(*y_ref.shape).astype(np.float32)
	dx_ref = R.conv2d_bwd_data(dy, w, x.shape, acc=np.float64, **kw)
	dw_ref = R.conv2d_bwd_filter(x, dy, w.shape, acc=np.float64, **kw)
	refs = (y_ref, dx_ref, dw_ref)
	algo = bnd.ConvFwdAlgo.implicitGemm.value            # (3x3 shapes: the implicit GEMM, not Winograd)
	scale = np.sqrt(dy.size / k)

	err = {}
	for mode in ("f32", ) + MODES:
		dnn.setConvMath(mode)
		got = passes(bnd, x, w, dy, stride, pad, algo)
		again = passes(bnd, x, w, dy, stride, pad, algo)
		for name, a, b, ref in zip(("fwd", "bwd data", "bwd filter"), got, again, refs):
			assert np.array_equal(a, b), "%s, %s: two runs differ" % (mode, name)
			atol = 1e-4 if name != "bwd filter" else 2e-6 * scale * 30
			assert_close(a, ref, atol=atol, rtol=1e-4, what="%s %s" % (mode, name))
		err[mode] = [float(np.abs(a - ref).max() / np.abs(ref).max()) for a, ref in zip(got, refs)]

	for mode in MODES:
		for name, e_split, e_f32 in zip(("fwd", "bwd data", "bwd filter"), err[mode], err["f32"]):
			assert e_split <= 2.0 * e_f32 + 1e-7, "%s %s: error %.3e against float64, fp32 MFMA %.3e" % (mode, name, e_split, e_f32)


def test_split_backward_filter_kernel_full_planes_and_accumulate(bnd, dnn):
	"""the pointwise backward-filter kernel: 55x55 planes (3025 = 378 cells + 1 pixel), many k-splits, the
	momentum * wgrad + scale * d contract (Hip/Wrappers/MIOpen.py:414-433)"""
	rng = np.random.RandomState(3)
	n, c, hw, k = 6, 128, 55, 192
	x, dy = rng.randn(n, c, hw, hw).astype(np.float32), rng.randn(n, k, hw, hw).astype(np.float32)
	w0 = rng.randn(k, c, 1, 1).astype(np.float32)
	ref = np.einsum("nkp,ncp->kc", dy.reshape(n, k, -1).astype(np.float64), x.reshape(n, c, -1).astype(np.float64)).reshape(k, c, 1, 1)
	G = bnd.GPUArray
	for mode in MODES:
		dnn.setConvMath(mode)
		wg = G.toGpu(w0)
		dnn.convNdBackwardParams(G.toGpu(x), G.toGpu(dy), G.toGpu(w0), 1, 0, wgrad=wg, scale=2.0, momentum=0.5)
		assert_close(wg.get(), 0.5 * w0 + 2.0 * ref, atol=2e-6 * np.sqrt(n * hw * hw) * 60, rtol=1e-4, what=mode)


def test_split_batchnorm_fold_matches_the_written_gradient(bnd, dnn):
	"""pz_conv2d_bwd_data_bn / pz_conv2d_bwd_filter_bn in a split mode: the BatchNorm backward applied while gathering
	gives what the kernels give on the materialised gradient (same arithmetic before the split)"""
	from puzzlelib_amd import lazy, fusion
	rng = np.random.RandomState(11)
	n, c, hw, k = 4, 128, 14, 256
	G = bnd.GPUArray
	x = G.toGpu(rng.randn(n, c, hw, hw).astype(np.float32))
	w = G.toGpu((rng.randn(k, c, 1, 1) / np.sqrt(c)).astype(np.float32))
	y = G.toGpu(rng.randn(n, k, hw, hw).astype(np.float32))          # the convolution's output = the BatchNorm's input
	dy = G.toGpu(rng.randn(n, k, hw, hw).astype(np.float32))
	coef = G.toGpu(np.concatenate([rng.rand(k, 1) + 0.5, rng.randn(k, 2) * 0.1, np.zeros((k, 1))], axis=1).astype(np.float32))

	def described():
		g = G.empty(dy.shape, dtype=np.float32)
		lazy.attach(g, fusion.BnBwdApply(dy, y, coef))
		return g

	for mode in MODES:
		dnn.setConvMath(mode)
		written = described()
		written.rptr
		lazy.counters.clear()
		dx_fold = dnn.convNdBackwardData(described(), w, None, x, 1, 0).get()
		dw_fold = dnn.convNdBackwardParams(x, described(), w, 1, 0).get()
		assert lazy.counters.get("dgrad_bn_fold", 0) == 1 and lazy.counters.get("wgrad_bn_fold", 0) == 1
		dx_plain = dnn.convNdBackwardData(written, w, None, x, 1, 0).get()
		dw_plain = dnn.convNdBackwardParams(x, written, w, 1, 0).get()
		assert np.array_equal(dx_fold, dx_plain), "%s: backward-data with the batch-norm folded in" % mode
		assert np.array_equal(dw_fold, dw_plain), "%s: backward-filter with the batch-norm folded in" % mode


def test_math_switch(bnd, dnn):
	import ctypes
	from puzzlelib_amd import lib
	got = ctypes.c_int(-1)
	for name, products in bnd.dnn.MATH.items():
		dnn.setConvMath(name)
		lib.pz_conv_math_get(ctypes.byref(got))
		assert got.value == products and dnn.convMath == name and not dnn.geometry
	with pytest.raises(ValueError):
		dnn.setConvMath("bf16")
	with pytest.raises(Exception):
		lib.pz_conv_math_set(7)
