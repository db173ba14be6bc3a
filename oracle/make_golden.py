"""
TEST INFRASTRUCTURE (build container only) — pins oracle/cpu_ref.py against the real reference and writes the
golden fixtures under tests/golden/.

    python oracle/make_golden.py            # check + (re)write fixtures
    python oracle/make_golden.py --check    # check only

What it does
  1. imports the Python reference with its numpy CPU backend (oracle/refimport.py) and compares every
     oracle function whose counterpart the reference CPU backend implements: conv forward, max/avg pool
     forward, batch-norm inference, GEMM, column sum, bias add, every activation fwd/bwd, dropout with a
     given mask, axpy/add/linear/weight-decay and all nine optimizer kernels (the gcc-JIT'd C loops);
  2b. does the same for the operators beside the hot path (f3: conv3d, deconvolutions, instance norm, both LRN modes, mask
     pooling / unpooling, batched matvec, SVM, PReLU, reflection pad, up-sampling, embedding) and binds the oracle's
     point-wise cost kernels into the reference's Cost modules (BCE, Hinge, SmoothL1, L1Hinge) to run THEIR tests;
  2. runs the reference's own `bnd`-parameterised unit tests (Cuda/Wrappers/CuDnn.py conv2dTest,
     convGroupTest, maxpool2dTest, softmax2dTest; CuDnnNorm.py batchNorm2dTest; CuBlas.py matrixTest;
     Cuda/Kernels/MatVec.py calcTest; Cuda/Kernels/Costs.py crossEntropyTest) against a stand-in `bnd`
     whose arrays/dnn/blas are the oracle — the reference's asserts are the judge of the backward /
     BN-train / softmax / cross-entropy restatements the CPU backend lacks;
  3. runs the reference LeNet forward (Models/Nets/LeNet.py) and the oracle's spec runner on the same
     seeded parameters;
  4. writes fixtures: tests/golden/ops.npz (per-op inputs and expected outputs; arrays tagged `ref_` were
     produced BY THE REFERENCE ITSELF, arrays tagged `orc_` by the oracle after the checks above),
     tests/golden/lenet.npz (seeded LeNet b64: reference forward logits + oracle full training step),
     tests/golden/miniresnet.npz (2-stage ResNet step from the oracle) and MANIFEST.json.
"""
import os, sys, json, types, argparse
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import refimport
import cpu_ref as R
import cpu_net as N
from puzzlelib_amd import nets

ATOL = 1e-5


def close(a, b, atol=ATOL, rtol=1e-5, what=""):
	a, b = np.asarray(a), np.asarray(b)
	assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
	ok = np.allclose(a, b, atol=atol, rtol=rtol)
	assert ok, "%s: max abs diff %g" % (what, np.max(np.abs(a.astype(np.float64) - b)))


# ----------------------------------------------------------------------------------------------
# 1. oracle vs imported reference CPU backend
# ----------------------------------------------------------------------------------------------

def check_against_reference(fx):
	from PuzzleLib.Backend import gpuarray, Blas, Dnn
	from PuzzleLib.Backend.Kernels import ElementWise as EW, MatVec
	from PuzzleLib.CPU.Wrappers import NumpyDnn

	rng = np.random.RandomState(1234)
	togpu = gpuarray.to_gpu

	# ---- conv forward through the dispatch surface (Backend/Dnn.py:344-346 -> NumpyDnn.conv2d)
	convcases = {
		"c0": dict(n=2, c=3, h=9, w=8, k=5, r=3, s=3, stride=1, pad=1, dil=1),
		"c1": dict(n=3, c=4, h=11, w=13, k=6, r=3, s=2, stride=2, pad=(1, 2), dil=1),
		"c2": dict(n=2, c=2, h=12, w=12, k=7, r=3, s=3, stride=1, pad=2, dil=2),
		"c3": dict(n=2, c=8, h=7, w=7, k=16, r=1, s=1, stride=2, pad=0, dil=1),
		"c4": dict(n=1, c=3, h=23, w=23, k=4, r=7, s=7, stride=2, pad=3, dil=1),
	}

	for name, cs in convcases.items():
		x = rng.randn(cs["n"], cs["c"], cs["h"], cs["w"]).astype(np.float32)
		w = rng.randn(cs["k"], cs["c"], cs["r"], cs["s"]).astype(np.float32)
		b = rng.randn(1, cs["k"], 1, 1).astype(np.float32)
		stride, pad, dil = R.pair(cs["stride"]), R.pair(cs["pad"]), R.pair(cs["dil"])

		ref = Dnn.convNd(togpu(x), togpu(w), togpu(b), stride, pad, dil, 1, None).get()
		orc = R.conv2d_fwd(x, w, b, stride, pad, dil)
		close(ref, orc, what="conv fwd " + name)

		dy = rng.randn(*ref.shape).astype(np.float32)
		wg0, bg0 = rng.randn(*w.shape).astype(np.float32), rng.randn(cs["k"]).astype(np.float32)

		fx["conv_%s_x" % name], fx["conv_%s_w" % name], fx["conv_%s_b" % name] = x, w, b.ravel()
		fx["conv_%s_cfg" % name] = np.array([*stride, *pad, *dil, 1], dtype=np.int32)
		fx["conv_%s_ref_y" % name] = ref
		fx["conv_%s_dy" % name] = dy
		fx["conv_%s_orc_dx" % name] = R.conv2d_bwd_data(dy, w, x.shape, stride, pad, dil)
		dw, db = R.conv2d_bwd_filter(x, dy, w.shape, stride, pad, dil, withbias=True)
		fx["conv_%s_orc_dw" % name], fx["conv_%s_orc_db" % name] = dw, db

		# accumulate mode: wgrad <- 0.5*wgrad + 2*dw   (Hip/Wrappers/MIOpen.py:414-433)
		wg, bg = R.conv2d_bwd_filter(
			x, dy, w.shape, stride, pad, dil, withbias=True, wgrad=wg0.copy(), bgrad=bg0.copy(), scale=2.0, momentum=0.5
		)
		fx["conv_%s_wg0" % name], fx["conv_%s_bg0" % name] = wg0, bg0
		fx["conv_%s_orc_wgacc" % name], fx["conv_%s_orc_bgacc" % name] = wg, bg

	# grouped conv: no reference CPU path; oracle only (checked by convGroupTest in step 2)
	x = rng.randn(3, 6, 5, 6).astype(np.float32)
	w = rng.randn(4, 3, 2, 2).astype(np.float32)
	dy_shape = R.conv2d_fwd(x, w, None, 1, 0, 1, 2).shape
	dy = rng.randn(*dy_shape).astype(np.float32)
	fx["conv_g2_x"], fx["conv_g2_w"], fx["conv_g2_dy"] = x, w, dy
	fx["conv_g2_cfg"] = np.array([1, 1, 0, 0, 1, 1, 2], dtype=np.int32)
	fx["conv_g2_orc_y"] = R.conv2d_fwd(x, w, None, 1, 0, 1, 2)
	fx["conv_g2_orc_dx"] = R.conv2d_bwd_data(dy, w, x.shape, 1, 0, 1, 2)
	fx["conv_g2_orc_dw"] = R.conv2d_bwd_filter(x, dy, w.shape, 1, 0, 1, 2)

	# ---- pooling forward (NumpyDnn.pool2d)
	x = rng.randn(3, 4, 9, 10).astype(np.float32)
	for name, (size, stride, pad) in {"p0": (3, 2, 1), "p1": (2, 2, 0), "p2": ((3, 2), (2, 1), (1, 0))}.items():
		ref, _ = Dnn.poolNd(togpu(x), R.pair(size), R.pair(stride), R.pair(pad), Dnn.PoolMode.max, False)
		ref = ref.get()
		close(ref, R.pool2d_fwd(x, size, stride, pad, R.POOL_MAX), what="maxpool " + name)

		dy = rng.randn(*ref.shape).astype(np.float32)
		fx["pool_%s_cfg" % name] = np.array([*R.pair(size), *R.pair(stride), *R.pair(pad)], dtype=np.int32)
		fx["pool_%s_ref_max" % name], fx["pool_%s_dy" % name] = ref, dy
		fx["pool_%s_orc_maxbwd" % name] = R.pool2d_bwd(dy, x, ref, size, stride, pad, R.POOL_MAX)

		for mode, tag in ((R.POOL_AVG_WITH_PAD, "avgp"), (R.POOL_AVG_NO_PAD, "avgn")):
			y = R.pool2d_fwd(x, size, stride, pad, mode)
			fx["pool_%s_orc_%s" % (name, tag)] = y
			fx["pool_%s_orc_%sbwd" % (name, tag)] = R.pool2d_bwd(dy, x, y, size, stride, pad, mode)

	fx["pool_x"] = x

	# avg pool, pad 0, is exact in the reference CPU backend (mean over the im2col row)
	ref, _ = Dnn.poolNd(togpu(x), (2, 2), (2, 2), (0, 0), Dnn.PoolMode.avgWithPad, False)
	close(ref.get(), R.pool2d_fwd(x, 2, 2, 0, R.POOL_AVG_WITH_PAD), what="avgpool pad0")

	# ---- batch-norm inference (NumpyDnn.batchNorm2d)
	x = rng.randn(4, 5, 3, 6).astype(np.float32)
	scale, bias = rng.randn(1, 5, 1, 1).astype(np.float32), rng.randn(1, 5, 1, 1).astype(np.float32)
	mean, var = rng.randn(1, 5, 1, 1).astype(np.float32), (1.0 + rng.randn(1, 5, 1, 1)**2).astype(np.float32)

	ref = Dnn.batchNormNd(togpu(x), togpu(scale), togpu(bias), togpu(mean), togpu(var), 1e-5, 0, True).get()
	close(ref, R.bn_fwd_infer(x, scale, bias, mean, var, 1e-5), what="bn inference")

	fx["bn_x"], fx["bn_scale"], fx["bn_bias"], fx["bn_mean"], fx["bn_var"] = x, scale.ravel(), bias.ravel(), \
		mean.ravel(), var.ravel()
	fx["bn_ref_infer"] = ref

	rmean, rvar = mean.ravel().copy(), var.ravel().copy()
	y, smean, sinv = R.bn_fwd_train(x, scale.ravel(), bias.ravel(), rmean, rvar, 1e-5, 0.25)
	dy = rng.randn(*x.shape).astype(np.float32)
	dx, dscale, dbias = R.bn_bwd(dy, x, scale.ravel(), smean, sinv)
	fx["bn_orc_train_y"], fx["bn_orc_savemean"], fx["bn_orc_saveinvvar"] = y, smean, sinv
	fx["bn_orc_runmean"], fx["bn_orc_runvar"] = rmean, rvar
	fx["bn_dy"], fx["bn_orc_dx"], fx["bn_orc_dscale"], fx["bn_orc_dbias"] = dy, dx, dscale, dbias

	# ---- GEMM / column sum / bias add / argmax
	A, B = rng.randn(7, 5).astype(np.float32), rng.randn(5, 9).astype(np.float32)
	close(Blas.mulMatrixOnMatrix(togpu(A), togpu(B)).get(), R.gemm(A, B), what="gemm nn")
	Bt = np.ascontiguousarray(B.T)
	close(Blas.mulMatrixOnMatrix(togpu(A), togpu(Bt), transpB=True).get(), R.gemm(A, Bt, transpB=True), what="gemm nt")
	At = np.ascontiguousarray(A.T)
	close(Blas.mulMatrixOnMatrix(togpu(At), togpu(B), transpA=True).get(), R.gemm(At, B, transpA=True), what="gemm tn")

	C0 = rng.randn(7, 9).astype(np.float32)
	fx["gemm_A"], fx["gemm_B"], fx["gemm_C0"] = A, B, C0
	fx["gemm_ref_nn"] = Blas.mulMatrixOnMatrix(togpu(A), togpu(B)).get()
	fx["gemm_orc_nn_ab"] = R.gemm(A, B, out=C0.copy(), alpha=0.5, beta=2.0)
	fx["gemm_orc_nt_ab"] = R.gemm(A, Bt, out=C0.copy(), transpB=True, alpha=-1.5, beta=1.0)
	fx["gemm_orc_tn_ab"] = R.gemm(At, B, out=C0.copy(), transpA=True, alpha=1.0, beta=1.0)

	M = rng.randn(33, 70).astype(np.float32)
	close(Blas.sumOnMatrix(togpu(M)).get(), R.matsum(M, 0), what="colsum")
	close(Blas.sumOnMatrix(togpu(M), cols=False).get(), R.matsum(M, 1), what="rowsum", atol=1e-4)
	v = rng.randn(70).astype(np.float32)
	biasout = gpuarray.empty(M.shape, dtype=np.float32)
	MatVec.addVecToMat(togpu(v), togpu(M), 1, biasout)
	close(biasout.get(), R.add_vec_to_mat(v, M, 1), what="bias add")
	assert np.array_equal(MatVec.argmax(togpu(M), axis=1).get(), R.argmax(M, 1))

	fx["mat_M"], fx["mat_v"] = M, v
	fx["mat_ref_colsum"] = Blas.sumOnMatrix(togpu(M)).get()
	fx["mat_ref_biasadd"] = biasout.get()
	fx["mat_ref_argmax"] = MatVec.argmax(togpu(M), axis=1).get().astype(np.int32)

	# ---- activations (CPU/Kernels/ElementWise.py through Backend/Kernels/ElementWise.py)
	x = (2.0 * rng.randn(1000)).astype(np.float32)
	g = rng.randn(1000).astype(np.float32)
	fx["act_x"], fx["act_g"] = x, g

	acts = {
		"sigmoid": (EW.sigmoidKer, EW.sigmoidDerKer, R.sigmoid, R.sigmoid_der, ()),
		"tanh": (EW.tanhKer, EW.tanhDerKer, R.tanh, R.tanh_der, ()),
		"relu": (EW.reluKer, EW.reluDerKer, R.relu, R.relu_der, ()),
		"leakyRelu": (EW.leakyReluKer, EW.leakyReluDerKer, R.leaky_relu, R.leaky_relu_der, (0.01, )),
		"elu": (EW.eluKer, EW.eluDerKer, R.elu, R.elu_der, (1.0, )),
		"softPlus": (EW.softPlusKer, EW.softPlusDerKer, R.softplus, R.softplus_der, ()),
		"clip": (EW.clipKer, EW.clipDerKer, R.clip, R.clip_der, (0.0, 6.0)),
	}

	for name, (ker, der, fn, dfn, args) in acts.items():
		out = gpuarray.empty(x.shape, dtype=np.float32)
		ker(np.dtype(np.float32))(out, togpu(x), *args)
		y = out.get()
		close(y, fn(x, *args), what=name)

		ing = gpuarray.empty(x.shape, dtype=np.float32)
		der(np.dtype(np.float32))(ing, togpu(g), togpu(y), *args)
		close(ing.get(), dfn(g, y, *args), what=name + " der")

		fx["act_ref_%s" % name], fx["act_ref_%s_der" % name] = y, ing.get()

	# strided variant (slice=...): the reference's CPU kernel object silently ignores `slice=` (CPU/SourceModule.py:197-205
	# always calls the dense entry), so the pin is the device semantics the Hip backend has — Cuda/SourceModule.py:216-226:
	# i = start + t*step while i < stop; untouched elements keep their old value.
	sl = slice(3, 900, 7)
	expected = x.copy()
	expected[sl] = R.relu(x[sl])
	fx["act_orc_relu_slice"] = expected

	# ---- dropout with a supplied mask
	bits = rng.randint(0, 2**32, size=1000, dtype=np.uint64).astype(np.uint32)
	v = int(0.5 * np.iinfo(np.uint32).max)
	out = gpuarray.empty(x.shape, dtype=np.float32)
	EW.dropoutKer(np.dtype(np.float32))(out, togpu(x), togpu(bits), v, np.float32(0.5))
	close(out.get(), R.dropout(x, bits, v, 0.5), what="dropout")
	fx["drop_bits"], fx["drop_v"], fx["drop_ref"] = bits, np.array([v], dtype=np.uint32), out.get()

	# ---- axpy / add / linear / weight decay
	y0 = rng.randn(1000).astype(np.float32)
	yy = togpu(y0.copy())
	Blas.toVectorAddVector(yy, togpu(x), alpha=0.3)
	close(yy.get(), R.axpy(y0.copy(), x, 0.3), what="axpy")
	fx["elt_y0"], fx["elt_ref_axpy"] = y0, yy.get()

	res = Blas.addVectorToVector(togpu(x), togpu(y0), alpha=0.7, beta=-1.1).get()
	close(res, R.add_scaled(x, 0.7, y0, -1.1), what="addVectorToVector")
	fx["elt_ref_add"] = res

	out = gpuarray.empty(x.shape, dtype=np.float32)
	EW.linearKer(np.dtype(np.float32))(out, togpu(x), 1.5, -0.25)
	close(out.get(), R.linear(x, 1.5, -0.25), what="linear")
	fx["elt_ref_linear"] = out.get()

	gr = togpu(g.copy())
	EW.weightDecayKer(gr, togpu(x), 1e-2)
	close(gr.get(), R.weight_decay(g.copy(), x, 1e-2), what="weightDecay")
	fx["elt_ref_wd"] = gr.get()

	# ---- optimizer kernels: three consecutive updates each
	def run_opt(refker, orcfn, nstates, scalars, tag, zero_state=False):
		p0 = rng.randn(500).astype(np.float32)
		grads = [rng.randn(500).astype(np.float32) for _ in range(3)]
		st0 = [np.abs(rng.randn(500)).astype(np.float32) * (0.0 if zero_state else 0.1) for _ in range(nstates)]

		p, st = togpu(p0.copy()), [togpu(s.copy()) for s in st0]
		po, sto = p0.copy(), [s.copy() for s in st0]

		for gg in grads:
			refker(np.dtype(np.float32))(p, togpu(gg), *st, *scalars)
			orcfn(po, gg, *sto, *scalars)

		close(p.get(), po, what=tag)
		for a, b in zip(st, sto):
			close(a.get(), b, what=tag + " state")

		fx["opt_%s_p0" % tag] = p0
		fx["opt_%s_grads" % tag] = np.stack(grads)
		fx["opt_%s_st0" % tag] = np.stack(st0)
		fx["opt_%s_scalars" % tag] = np.array(scalars, dtype=np.float64)
		fx["opt_%s_ref_p" % tag] = p.get()
		fx["opt_%s_ref_st" % tag] = np.stack([a.get() for a in st])

	run_opt(EW.adamKer, R.adam, 2, (1e-2, 0.1, 0.001, 1e-8), "adam")
	run_opt(EW.classicMomSGDKer, R.classic_mom_sgd, 1, (0.1, 0.9), "classicMomSGD")
	run_opt(EW.nesterovMomSGDKer, R.nesterov_mom_sgd, 1, (0.1, 0.9), "nesterovMomSGD")
	run_opt(EW.rmspropKer, R.rmsprop, 1, (1e-2, 0.9, 1e-6), "rmsprop")
	run_opt(EW.adagradKer, R.adagrad, 1, (1e-2, 1e-6), "adagrad")
	run_opt(EW.adadeltaKer, R.adadelta, 2, (0.95, 1e-6), "adadelta")
	run_opt(EW.rmspropGravesKer, R.rmsprop_graves, 3, (1e-3, 0.95, 0.9, 1e-2), "rmspropGraves", zero_state=True)
	run_opt(EW.smorms3Ker, R.smorms3, 3, (1e-2, 1e-6), "smorms3")

	# ---- softmax / cross-entropy / accuracy (no reference CPU path: oracle, judged in step 2)
	s = rng.randn(20, 10, 3).astype(np.float32)
	lab = rng.randint(0, 10, size=(20, 3)).astype(np.int32)
	err, grad = R.cross_entropy(s, lab)
	fx["ce_scores"], fx["ce_labels"], fx["ce_orc_err"], fx["ce_orc_grad"] = s, lab, np.array([err]), grad

	s2 = rng.randn(64, 10).astype(np.float32)
	lab2 = rng.randint(0, 10, size=(64, )).astype(np.int32)
	err2, grad2 = R.cross_entropy(s2, lab2)
	fx["ce2_scores"], fx["ce2_labels"], fx["ce2_orc_err"], fx["ce2_orc_grad"] = s2, lab2, np.array([err2]), grad2

	sm = rng.randn(5, 8, 2, 3).astype(np.float32)
	y = R.softmax_fwd(sm)
	gsm = rng.randn(*sm.shape).astype(np.float32)
	fx["sm_x"], fx["sm_orc_y"], fx["sm_g"], fx["sm_orc_dx"] = sm, y, gsm, R.softmax_bwd(gsm, y)

	print("[1] oracle == reference CPU backend on every op the reference implements: OK")


# ----------------------------------------------------------------------------------------------
# 2. the reference's own bnd-parameterised tests, judged on an oracle-backed stand-in backend
# ----------------------------------------------------------------------------------------------

class HostArray:
	"""Minimal GPUArray look-alike over numpy for the reference tests (shape/dtype/get/reshape/ravel)."""
	def __init__(self, a):
		self.a = np.ascontiguousarray(a)

	shape = property(lambda self: self.a.shape)
	dtype = property(lambda self: self.a.dtype)
	ndim = property(lambda self: self.a.ndim)
	size = property(lambda self: self.a.size)

	def get(self):
		return self.a.copy()

	def reshape(self, *shape):
		return HostArray(self.a.reshape(*shape))

	def ravel(self):
		return HostArray(self.a.ravel())

	def dimAt(self, i):
		return self.a.shape[i]

	@staticmethod
	def toGpu(a):
		return HostArray(a)

	@staticmethod
	def zeros(shape, dtype):
		return HostArray(np.zeros(shape, dtype=dtype))

	@staticmethod
	def empty(shape, dtype, allocator=None):
		return HostArray(np.empty(shape, dtype=dtype))


def make_oracle_bnd():
	from enum import Enum

	class PoolMode(Enum):
		max = R.POOL_MAX
		avgWithPad = R.POOL_AVG_WITH_PAD
		avgNoPad = R.POOL_AVG_NO_PAD

	class SoftMaxMode(Enum):
		perActivation = 0
		spatial = 1

	H = HostArray

	class Dnn:
		@staticmethod
		def convNd(data, W, bias=None, stride=1, pad=0, dilation=1, groups=1, **_):
			return H(R.conv2d_fwd(data.a, W.a, None if bias is None else bias.a, stride, pad, dilation, groups))

		@staticmethod
		def convNdBackwardData(grad, W, bias=None, data=None, stride=1, pad=0, dilation=1, postpad=0, groups=1, **_):
			if data is not None:
				shape = data.shape
			else:
				st, pd, dl = R.pair(stride), R.pair(pad), R.pair(dilation)
				n, _, oh, ow = grad.shape
				_, cg, fh, fw = W.shape
				shape = (n, cg * groups, (oh - 1) * st[0] + dl[0] * (fh - 1) - 2 * pd[0] + 1,
						 (ow - 1) * st[1] + dl[1] * (fw - 1) - 2 * pd[1] + 1)
			return H(R.conv2d_bwd_data(grad.a, W.a, shape, stride, pad, dilation, groups))

		@staticmethod
		def convNdBackwardParams(data, grad, W, stride=1, pad=0, dilation=1, groups=1, withbias=False, deconv=False,
								 wgrad=None, bgrad=None, scale=1.0, momentum=0.0, **_):
			res = R.conv2d_bwd_filter(data.a, grad.a, W.shape, stride, pad, dilation, groups, withbias,
									  None if wgrad is None else wgrad.a, None if bgrad is None else bgrad.a, scale, momentum)
			return (H(res[0]), H(res[1])) if withbias else H(res)

		@staticmethod
		def poolNd(data, size=2, stride=2, pad=0, mode=0, **_):
			return H(R.pool2d_fwd(data.a, size, stride, pad, mode))

		@staticmethod
		def poolNdBackward(grad, indata, outdata, size=2, stride=2, pad=0, mode=0, **_):
			return H(R.pool2d_bwd(grad.a, indata.a, outdata.a, size, stride, pad, mode))

		@staticmethod
		def softmaxNd(data, mode=1, allocator=None):
			return H(R.softmax_fwd(data.a))

		@staticmethod
		def softmaxNdBackward(grad, outdata, **_):
			return H(R.softmax_bwd(grad.a, outdata.a))

		@staticmethod
		def batchNormNd(data, mean, var, scale, bias, epsilon=1e-5, factor=1.0, test=False, mode=None, out=None, **_):
			if test:
				return H(R.bn_fwd_infer(data.a, scale.a, bias.a, mean.a, var.a, epsilon))
			y, sm, si = R.bn_fwd_train(data.a, scale.a, bias.a, mean.a, var.a, epsilon, factor)
			return H(y), H(sm), H(si)

		@staticmethod
		def batchNormNdBackward(grad, data, scale, savemean, saveinvvar, epsilon=1e-5, **_):
			dx, ds, db = R.bn_bwd(grad.a, data.a, scale.a, savemean.a, saveinvvar.a)
			return H(dx), H(ds), H(db)

	class BlasCtx:
		@staticmethod
		def gemm(A, B, out=None, transpA=False, transpB=False, alpha=1.0, beta=0.0, allocator=None):
			return H(R.gemm(A.a, B.a, None if out is None else out.a, transpA, transpB, alpha, beta))

	bnd = types.SimpleNamespace(
		GPUArray=HostArray, dnn=Dnn, blas=BlasCtx, PoolMode=PoolMode, SoftMaxMode=SoftMaxMode, nthreads=256
	)
	return bnd


def run_reference_tests_on_oracle():
	# these modules hold plain functions taking `bnd`; importing them does not need a device
	from PuzzleLib.Cuda.Wrappers import CuDnn, CuDnnNorm, CuBlas

	bnd = make_oracle_bnd()
	for seed in range(5):
		np.random.seed(100 + seed)
		CuDnn.conv2dTest(bnd, np.float32, ATOL)
		CuDnn.convGroupTest(bnd, np.float32, ATOL)
		CuDnn.maxpool2dTest(bnd, np.float32, ATOL)
		CuDnn.softmax2dTest(bnd, np.float32, ATOL)
		CuDnnNorm.batchNorm2dTest(bnd, np.float32, ATOL, np.float32)
		CuBlas.matrixTest(bnd, np.float32, ATOL)

	# Cuda/Kernels/MatVec.py calcTest and Costs.py crossEntropyTest take a *module* object
	from PuzzleLib.Cuda.Kernels import MatVec, Costs

	matmod = types.SimpleNamespace(
		GPUArray=HostArray,
		addVecToMat=lambda vec, mat, axis=0, out=None, allocator=None: HostArray(R.add_vec_to_mat(vec.a, mat.a, axis)),
		matsum=lambda t, axis=0, out=None, alpha=1.0, beta=0.0, allocator=None: HostArray(R.matsum(t.a, axis)),
		matvec=lambda mat, vec, axis=0, **_: HostArray(
			(mat.a @ vec.a if axis == 1 else mat.a.T @ vec.a).astype(np.float32)
		),
		argmax=lambda t, axis=0, allocator=None: HostArray(R.argmax(t.a, axis)),
	)

	def ce(scores, labels, weights=None, error=None, allocator=None):
		err, grad = R.cross_entropy(scores.a, labels.a, None if weights is None else weights.a)
		return HostArray(np.array(err, dtype=np.float32)), HostArray(grad)

	costmod = types.SimpleNamespace(GPUArray=HostArray, crossEntropy=ce)

	for seed in range(5):
		np.random.seed(200 + seed)
		MatVec.calcTest(matmod, np.float32, 1e-4)
		Costs.crossEntropyTest(costmod)

	print("[2] reference unit tests (conv2d/convGroup/maxpool2d/softmax2d/batchNorm2d/matrix/MatVec/crossEntropy) "
		  "pass on the oracle-backed bnd: OK")


# ----------------------------------------------------------------------------------------------
# 2b. operators beside the hot path: the reference's own tests for them, judged on the oracle
# ----------------------------------------------------------------------------------------------

def make_oracle_f3():
	"""Oracle-backed stand-ins for what the reference's f3 tests drive: a `bnd` with N-d convolution / deconvolution, both
	LRN modes and instance normalisation, and the kernel modules (pool, matvec, cost, PReLU, pad, upsample, embed)."""
	H = HostArray
	base = make_oracle_bnd()

	def triple(v):
		return (int(v), ) * 3 if isinstance(v, (int, np.integer)) else tuple(int(a) for a in v)

	def inshape(grad, W, stride, pad, dilation, groups, nd):
		st, pd, dl = (R.pair(stride), R.pair(pad), R.pair(dilation)) if nd == 2 else (triple(stride), triple(pad), triple(dilation))
		spatial = tuple((o - 1) * s + d * (f - 1) - 2 * p + 1 for o, s, d, f, p in zip(grad.shape[2:], st, dl, W.shape[2:], pd))
		return (grad.shape[0], W.shape[1] * groups) + spatial

	class Dnn(base.dnn):
		@staticmethod
		def convNd(data, W, bias=None, stride=1, pad=0, dilation=1, groups=1, **_):
			if data.ndim == 5:
				assert groups == 1
				y = R.conv3d_fwd(data.a, W.a, None if bias is None else bias.a, triple(stride), triple(pad), triple(dilation))
				return H(y.astype(np.float32))
			return base.dnn.convNd(data, W, bias, stride, pad, dilation, groups)

		@staticmethod
		def convNdBackwardData(grad, W, bias=None, data=None, stride=1, pad=0, dilation=1, postpad=0, groups=1, **_):
			nd = grad.ndim - 2
			shape = data.shape if data is not None else inshape(grad, W, stride, pad, dilation, groups, nd)
			if nd == 3:
				assert groups == 1
				dx = R.conv3d_bwd_data(grad.a, W.a, shape, triple(stride), triple(pad), triple(dilation))
			else:
				dx = R.conv2d_bwd_data(grad.a, W.a, shape, stride, pad, dilation, groups)
			if bias is not None:            # deconvolution forward (Hip/Wrappers/MIOpen.py:368-400 adds the bias over the maps)
				dx = dx + bias.a.reshape((1, -1) + (1, ) * nd)
			return H(dx.astype(np.float32))

		@staticmethod
		def convNdBackwardParams(data, grad, W, stride=1, pad=0, dilation=1, groups=1, withbias=False, deconv=False,
								 wgrad=None, bgrad=None, scale=1.0, momentum=0.0, **_):
			nd = data.ndim - 2
			if nd == 3:
				assert groups == 1 and wgrad is None and bgrad is None
				dw, db = R.conv3d_bwd_filter(data.a, grad.a, W.shape, triple(stride), triple(pad), triple(dilation))
			else:
				res = R.conv2d_bwd_filter(data.a, grad.a, W.shape, stride, pad, dilation, groups, True)
				dw, db = res
			if deconv:                      # the bias gradient sums over `data`'s maps (MIOpen.py:435-436)
				db = data.a.sum(axis=(0, ) + tuple(range(2, data.ndim)), dtype=np.float64)
			dw, db = dw.astype(np.float32), np.asarray(db, dtype=np.float32)
			return (H(dw), H(db)) if withbias else H(dw)

		@staticmethod
		def mapLRN(data, means=None, N=5, alpha=1e-4, beta=0.75, K=2.0, **_):
			return H(R.lrn_fwd(data.a, N, alpha, beta, K, cross=False))

		@staticmethod
		def mapLRNBackward(data, grad, means=None, N=5, alpha=1e-4, beta=0.75, K=2.0, **_):
			return H(R.lrn_bwd(data.a, grad.a, N, alpha, beta, K, cross=False))

		@staticmethod
		def crossMapLRN(data, N=5, alpha=1e-4, beta=0.75, K=2.0, **_):
			return H(R.lrn_fwd(data.a, N, alpha, beta, K, cross=True))

		@staticmethod
		def crossMapLRNBackward(data, outdata, grad, N=5, alpha=1e-4, beta=0.75, K=2.0, **_):
			return H(R.lrn_bwd(data.a, grad.a, N, alpha, beta, K, cross=True))

	def instanceNorm2d(data, scale, bias, epsilon=1e-5, **_):
		return tuple(H(a) for a in R.instance_norm_fwd(data.a, scale.a, bias.a, epsilon))

	def instanceNorm2dBackward(grad, data, extscale, savemean, saveinvvar, epsilon, affine=True, **_):
		res = R.instance_norm_bwd(grad.a, data.a, extscale.a, savemean.a, saveinvvar.a, affine)
		return tuple(H(a) for a in res) if affine else H(res)

	bnd = types.SimpleNamespace(
		GPUArray=H, dnn=Dnn, blas=base.blas, PoolMode=base.PoolMode, SoftMaxMode=base.SoftMaxMode, nthreads=256,
		instanceNorm2d=instanceNorm2d, instanceNorm2dBackward=instanceNorm2dBackward
	)

	poolmod = types.SimpleNamespace(
		GPUArray=H,
		maxpool2d=lambda data, size, stride, pad, allocator=None: tuple(H(a) for a in R.maskpool2d_fwd(data.a, size, stride, pad)),
		maxpool2dBackward=lambda grad, origshape, mask, size, stride, pad, allocator=None: H(R.maskpool2d_bwd(grad.a, mask.a, origshape)),
		maxunpool2d=lambda data, origshape, mask, allocator=None: H(R.maxunpool2d_fwd(data.a, mask.a, origshape)),
		maxunpool2dBackward=lambda grad, poolshape, mask, allocator=None: H(R.maxunpool2d_bwd(grad.a, mask.a)),
	)

	matmod = types.SimpleNamespace(
		GPUArray=H,
		addVecToMat=lambda vec, mat, axis=0, out=None, allocator=None: H(R.add_vec_to_mat(vec.a, mat.a, axis)),
		matsum=lambda t, axis=0, out=None, alpha=1.0, beta=0.0, allocator=None: H(R.matsum(t.a, axis)),
		matvec=lambda mat, vec, axis=0, out=None, alpha=1.0, beta=0.0, allocator=None: H(R.matvec(mat.a, vec.a, axis)),
		argmax=lambda t, axis=0, allocator=None: H(R.argmax(t.a, axis)),
	)

	def svm(scores, labels, mode, error=None, allocator=None):
		err, grad = R.svm_cost(scores.a, labels.a, mode)
		return H(np.array(err, dtype=np.float32)), H(grad)

	costmod = types.SimpleNamespace(GPUArray=H, svm=svm)

	prelumod = types.SimpleNamespace(
		GPUArray=H,
		prelu=lambda data, slopes, inplace=False, sharedMaps=False, allocator=None: H(R.prelu_fwd(data.a, slopes.a, sharedMaps)),
		preluBackwardData=lambda grad, slopes, indata, sharedMaps=False, allocator=None:
			H(R.prelu_bwd_data(grad.a, slopes.a, indata.a, sharedMaps)),
		preluBackwardParams=lambda indata, outgrad, sharedMaps=False, allocator=None: H(R.prelu_bwd_params(indata.a, outgrad.a, sharedMaps)),
	)
	padmod = types.SimpleNamespace(
		GPUArray=H,
		reflectpad=lambda data, pad, allocator=None: H(R.reflectpad_fwd(data.a, pad)),
		reflectpadBackward=lambda grad, pad, allocator=None: H(R.reflectpad_bwd(grad.a, pad)),
	)
	upsamplemod = types.SimpleNamespace(
		GPUArray=H,
		upsample2d=lambda data, scale, mode="nearest", allocator=None: H(R.upsample_fwd(data.a, scale, mode)),
		upsample2dBackward=lambda grad, scale, mode="nearest", allocator=None: H(R.upsample_bwd(grad.a, scale, mode)),
		upsample3d=lambda data, scale, mode="nearest", allocator=None: H(R.upsample_fwd(data.a, scale, mode)),
		upsample3dBackward=lambda grad, scale, mode="nearest", allocator=None: H(R.upsample_bwd(grad.a, scale, mode)),
	)
	embedmod = types.SimpleNamespace(
		GPUArray=H,
		embed=lambda data, W, allocator=None: H(R.embed_fwd(data.a, W.a)),
		embedBackwardParams=lambda indata, grad, W, scale: R.embed_bwd_params(indata.a, grad.a, W.a, scale),
	)
	def ctcLoss(data, datalen, labels, lengths, blank, error=None, normalized=False, returnAlphas=False, allocator=None):
		err, grad, alphas = R.ctc_loss(data.a, datalen.a, labels.a, lengths, blank, normalized)
		err = H(np.array(err, dtype=np.float32))
		return (err, H(grad), H(alphas)) if returnAlphas else (err, H(grad))

	ctcmod = types.SimpleNamespace(GPUArray=H, ctcLoss=ctcLoss)
	return bnd, dict(poolmod=poolmod, matmod=matmod, costmod=costmod, prelumod=prelumod, padmod=padmod, upsamplemod=upsamplemod,
					 embedmod=embedmod, ctcmod=ctcmod)


def run_reference_f3_tests_on_oracle():
	"""The tests the reference holds for the operators beside the hot path — they need a device in the reference
	(Unittester.py:114-122 runs them for the Hip backend) — executed here with the oracle as the backend under test."""
	from PuzzleLib.Cuda.Wrappers import CuDnn, CuDnnNorm
	from PuzzleLib.Cuda.Kernels import Pool, MatVec, Costs, PRelu, Pad, Upsample, Embedder, CTC
	import random

	bnd, mods = make_oracle_f3()
	for seed in range(3):
		np.random.seed(300 + seed)
		CuDnn.conv3dTest(bnd, np.float32, ATOL)
		CuDnn.deconv2dTest(bnd, np.float32, ATOL)
		CuDnn.deconv3dTest(bnd, np.float32, ATOL)
		CuDnn.deconvGroupTest(bnd, np.float32, ATOL)
		CuDnnNorm.instanceNorm2dTest(bnd, np.float32, ATOL, np.float32)
		CuDnnNorm.mapLRN2dTest(bnd, np.float32, ATOL)
		CuDnnNorm.crossMapLRN2dTest(bnd, np.float32, ATOL)
		Pool.poolTest(mods["poolmod"])
		Pool.unpoolTest(mods["poolmod"])
		MatVec.batchCalcTest(mods["matmod"], np.float32, 1e-4)
		Costs.svmTest(mods["costmod"])
		PRelu.preluTest(mods["prelumod"])
		Pad.reflectpad1dTest(mods["padmod"], np.float32)
		Pad.reflectpad2dTest(mods["padmod"], np.float32, ATOL)
		Upsample.upsample2dNearestTest(mods["upsamplemod"])
		Upsample.upsample2dLinearTest(mods["upsamplemod"])
		Upsample.upsample3dNearestTest(mods["upsamplemod"])
		Upsample.upsample3dLinearTest(mods["upsamplemod"])
		Embedder.embedTest(mods["embedmod"], np.float32, ATOL)
		random.seed(300 + seed)                 # (CTC.createData draws the label lengths from `random`)
		CTC.ctcLossTest(mods["ctcmod"])

	# point-wise cost kernels: the reference's Cost modules (Cost/BCE.py, Hinge.py, SmoothL1.py, L1Hinge.py) import their
	# kernel from Backend/Kernels/Costs.py, which binds nothing on the CPU backend; bound to the oracle's restatement of the
	# device kernels (Cuda/Kernels/Costs.py:8-72) the modules' own unit tests judge it.
	from PuzzleLib.Backend.Kernels import Costs as KCosts

	def onCpu(fn, ngrads):
		def ker(*args):
			arrays = [a.data if hasattr(a, "data") and isinstance(a.data, np.ndarray) else a for a in args]
			return fn(*arrays)
		return ker

	def bce(scores, labels, err, grad, numsamples, spatial):
		e, g = R.bce_cost(scores, labels, numsamples, spatial)
		err[...] += e
		grad[...] = g.reshape(grad.shape)

	def hinge(scores, labels, err, grad, numsamples, numcases):
		e, g = R.hinge_cost(scores, labels, numsamples, numcases)
		err[...] += e
		grad[...] = g

	def smoothl1(pred, target, err, grad, norm, fullnorm):
		e, g = R.smooth_l1_cost(pred, target, norm, fullnorm)
		err[...] += e
		grad[...] = g

	def l1hinge(x1, x2, labels, err, g1, g2, numsamples, numcases):
		e, a, b = R.l1_hinge_cost(x1, x2, labels, numsamples, numcases)
		err[...] += e
		g1[...], g2[...] = a, b

	KCosts.bceKer, KCosts.hingeKer = onCpu(bce, 1), onCpu(hinge, 1)
	KCosts.smoothL1Ker, KCosts.l1HingeKer = onCpu(smoothl1, 1), onCpu(l1hinge, 2)

	import importlib
	for seed in range(3):
		np.random.seed(400 + seed)
		importlib.import_module("PuzzleLib.Cost.BCE")
		sys.modules["PuzzleLib.Cost.BCE"].errorTest()
		importlib.import_module("PuzzleLib.Cost.Hinge")
		sys.modules["PuzzleLib.Cost.Hinge"].errorValTest()
		importlib.import_module("PuzzleLib.Cost.SmoothL1")
		sys.modules["PuzzleLib.Cost.SmoothL1"].errorTest()
		sys.modules["PuzzleLib.Cost.SmoothL1"].valTest()
		importlib.import_module("PuzzleLib.Cost.L1Hinge")
		sys.modules["PuzzleLib.Cost.L1Hinge"].errorTest()

	print("[2b] reference tests of the operators beside the hot path (conv3d / deconv2d / deconv3d / deconvGroup / instanceNorm2d / "
		  "mapLRN / crossMapLRN / maskpool / unpool / batched matvec / svm / prelu / reflectpad 1d+2d / upsample 2d+3d nearest+linear / "
		  "embed / ctcLoss; Cost modules BCE / Hinge / SmoothL1 / L1Hinge) pass on the oracle: OK")


def f3_fixtures(fx):
	"""Inputs and oracle outputs (orc_) for the GPU tests of the f3 operators, written after 2b pinned the oracle."""
	rng = np.random.RandomState(777)
	f32 = np.float32

	x = rng.randn(3, 4, 7, 6).astype(f32)
	y, mask = R.maskpool2d_fwd(x, (3, 2), (2, 2), (1, 0))
	dy = rng.randn(*y.shape).astype(f32)
	fx["f3_pool_x"], fx["f3_pool_cfg"], fx["f3_pool_dy"] = x, np.array([3, 2, 2, 2, 1, 0], np.int32), dy
	fx["f3_pool_orc_y"], fx["f3_pool_orc_mask"] = y, mask
	fx["f3_pool_orc_dx"] = R.maskpool2d_bwd(dy, mask, x.shape)
	up = R.maxunpool2d_fwd(y, mask, x.shape)
	gup = rng.randn(*up.shape).astype(f32)
	fx["f3_unpool_orc_y"], fx["f3_unpool_g"], fx["f3_unpool_orc_dx"] = up, gup, R.maxunpool2d_bwd(gup, mask)

	x = rng.randn(2, 7, 6, 5).astype(f32)
	dy = rng.randn(*x.shape).astype(f32)
	fx["f3_lrn_x"], fx["f3_lrn_dy"], fx["f3_lrn_cfg"] = x, dy, np.array([5, 1.0, 0.5, 2.0])
	for cross, tag in ((False, "map"), (True, "cross")):
		fx["f3_lrn_orc_%s_y" % tag] = R.lrn_fwd(x, 5, 1.0, 0.5, 2.0, cross)
		fx["f3_lrn_orc_%s_dx" % tag] = R.lrn_bwd(x, dy, 5, 1.0, 0.5, 2.0, cross)

	x = rng.randn(3, 4, 5, 6).astype(f32)
	scale, bias = rng.randn(4).astype(f32), rng.randn(4).astype(f32)
	y, sm, si, ext = R.instance_norm_fwd(x, scale, bias, 1e-5)
	dy = rng.randn(*x.shape).astype(f32)
	dx, ds, db = R.instance_norm_bwd(dy, x, ext, sm, si)
	fx["f3_in_x"], fx["f3_in_scale"], fx["f3_in_bias"], fx["f3_in_dy"] = x, scale, bias, dy
	fx["f3_in_orc_y"], fx["f3_in_orc_mean"], fx["f3_in_orc_invvar"] = y, sm, si
	fx["f3_in_orc_dx"], fx["f3_in_orc_dscale"], fx["f3_in_orc_dbias"] = dx, ds, db

	# 3-d convolution and 3-d deconvolution (stride 2, pad 1; filters 3 x 2 x 3)
	x = rng.randn(2, 3, 5, 6, 7).astype(f32)
	w = (0.3 * rng.randn(4, 3, 3, 2, 3)).astype(f32)
	b = rng.randn(4).astype(f32)
	st, pd, dl = (2, 1, 2), (1, 0, 1), (1, 1, 1)
	y = R.conv3d_fwd(x, w, b, st, pd, dl).astype(f32)
	dy = rng.randn(*y.shape).astype(f32)
	dw, db = R.conv3d_bwd_filter(x, dy, w.shape, st, pd, dl)
	fx["f3_c3_x"], fx["f3_c3_w"], fx["f3_c3_b"], fx["f3_c3_dy"] = x, w, b, dy
	fx["f3_c3_cfg"] = np.array([*st, *pd, *dl], np.int32)
	fx["f3_c3_orc_y"], fx["f3_c3_orc_dx"] = y, R.conv3d_bwd_data(dy, w, x.shape, st, pd, dl).astype(f32)
	fx["f3_c3_orc_dw"], fx["f3_c3_orc_db"] = dw.astype(f32), db.astype(f32)
	# deconvolution: data (2, 4, 2, 3, 3) with the same filter bank read as (inmaps=4, outmaps=3, ...)
	d = rng.randn(2, 4, 2, 3, 3).astype(f32)
	bd = rng.randn(3).astype(f32)
	oshape = (2, 3) + tuple((o - 1) * s + (f - 1) - 2 * p + 1 for o, s, f, p in zip(d.shape[2:], st, w.shape[2:], pd))
	out = (R.conv3d_bwd_data(d, w, oshape, st, pd, dl) + bd.reshape(1, -1, 1, 1, 1)).astype(f32)
	g = rng.randn(*out.shape).astype(f32)
	dwd, _ = R.conv3d_bwd_filter(g, d, w.shape, st, pd, dl)
	fx["f3_d3_x"], fx["f3_d3_b"], fx["f3_d3_g"] = d, bd, g
	fx["f3_d3_orc_y"] = out
	fx["f3_d3_orc_dx"] = R.conv3d_fwd(g, w, None, st, pd, dl).astype(f32)
	fx["f3_d3_orc_dw"], fx["f3_d3_orc_db"] = dwd.astype(f32), g.sum(axis=(0, 2, 3, 4), dtype=np.float64).astype(f32)

	# matvec / svm
	A = rng.randn(5, 12, 9).astype(f32)
	v, w_ = rng.randn(5, 9).astype(f32), rng.randn(5, 12).astype(f32)
	fx["f3_mv_A"], fx["f3_mv_v"], fx["f3_mv_w"] = A, v, w_
	fx["f3_mv_orc_rows"], fx["f3_mv_orc_cols"] = R.matvec(A, v, 1), R.matvec(A, w_, 0)
	s = rng.randn(10, 5, 3).astype(f32)
	lab = rng.randint(0, 5, size=(10, 3)).astype(np.int32)
	fx["f3_svm_scores"], fx["f3_svm_labels"] = s, lab
	for mode in ("l1", "l2"):
		err, grad = R.svm_cost(s, lab, mode)
		fx["f3_svm_orc_%s_err" % mode], fx["f3_svm_orc_%s_grad" % mode] = np.array([err], f32), grad

	# point-wise costs
	s = (1.5 * rng.randn(12, 1, 3, 4)).astype(f32)
	lab = rng.randint(0, 2, size=(12, 3, 4)).astype(np.int32)
	err, grad = R.bce_cost(s, lab, 12, 12)
	fx["f3_bce_scores"], fx["f3_bce_labels"], fx["f3_bce_orc_err"], fx["f3_bce_orc_grad"] = s, lab, np.array([err], f32), grad
	s = rng.randn(15, 6).astype(f32)
	lab = (rng.randint(0, 2, size=(15, 6)) * 2 - 1).astype(np.int32)
	err, grad = R.hinge_cost(s, lab, 15, 6)
	fx["f3_hinge_scores"], fx["f3_hinge_labels"], fx["f3_hinge_orc_err"], fx["f3_hinge_orc_grad"] = s, lab, np.array([err], f32), grad
	p, t = (2 * rng.randn(9, 11)).astype(f32), rng.randn(9, 11).astype(f32)
	err, grad = R.smooth_l1_cost(p, t, 1.0 / 11, 1.0 / 99)
	fx["f3_sl1_pred"], fx["f3_sl1_target"], fx["f3_sl1_orc_err"], fx["f3_sl1_orc_grad"] = p, t, np.array([err], f32), grad
	x1, x2 = rng.randn(14, 5).astype(f32), rng.randn(14, 5).astype(f32)
	lab = rng.randint(0, 2, size=(14, )).astype(np.int32)
	err, g1, g2 = R.l1_hinge_cost(x1, x2, lab, 14, 5)
	fx["f3_l1h_x1"], fx["f3_l1h_x2"], fx["f3_l1h_labels"] = x1, x2, lab
	fx["f3_l1h_orc_err"], fx["f3_l1h_orc_g1"], fx["f3_l1h_orc_g2"] = np.array([err], f32), g1, g2

	# PReLU (per map and shared), reflection pad, up-sampling, embedding
	x = rng.randn(4, 5, 3, 7).astype(f32)
	dy = rng.randn(*x.shape).astype(f32)
	slopes, one = rng.randn(5).astype(f32), rng.randn(1).astype(f32)
	fx["f3_prelu_x"], fx["f3_prelu_dy"], fx["f3_prelu_slopes"], fx["f3_prelu_shared"] = x, dy, slopes, one
	for shared, sl, tag in ((False, slopes, "map"), (True, one, "shared")):
		fx["f3_prelu_orc_%s_y" % tag] = R.prelu_fwd(x, sl, shared)
		fx["f3_prelu_orc_%s_dx" % tag] = R.prelu_bwd_data(dy, sl, x, shared)
		fx["f3_prelu_orc_%s_ds" % tag] = R.prelu_bwd_params(x, dy, shared)

	x = rng.randn(2, 3, 6, 9).astype(f32)
	pad2 = (2, 3, 4, 1)
	y = R.reflectpad_fwd(x, pad2)
	g = rng.randn(*y.shape).astype(f32)
	fx["f3_pad2_x"], fx["f3_pad2_pad"], fx["f3_pad2_g"] = x, np.array(pad2, np.int32), g
	fx["f3_pad2_orc_y"], fx["f3_pad2_orc_dx"] = y, R.reflectpad_bwd(g, pad2)
	x = rng.randn(3, 2, 11).astype(f32)
	y = R.reflectpad_fwd(x, (3, 5))
	g = rng.randn(*y.shape).astype(f32)
	fx["f3_pad1_x"], fx["f3_pad1_g"], fx["f3_pad1_orc_y"], fx["f3_pad1_orc_dx"] = x, g, y, R.reflectpad_bwd(g, (3, 5))

	x2, x3 = rng.randn(2, 3, 5, 4).astype(f32), rng.randn(2, 2, 3, 4, 5).astype(f32)
	for tag, x, scale in (("2d", x2, (2, 3)), ("3d", x3, (2, 1, 3))):
		fx["f3_up%s_x" % tag], fx["f3_up%s_scale" % tag] = x, np.array(scale, np.int32)
		for mode in ("nearest", "linear"):
			y = R.upsample_fwd(x, scale, mode)
			g = rng.randn(*y.shape).astype(f32)
			fx["f3_up%s_%s_g" % (tag, mode)] = g
			fx["f3_up%s_orc_%s_y" % (tag, mode)], fx["f3_up%s_orc_%s_dx" % (tag, mode)] = y, R.upsample_bwd(g, scale, mode)

	words = rng.randint(-1, 40, size=(6, 7)).astype(np.int32)
	vocab = rng.randn(40, 9).astype(f32)
	g = rng.randn(6, 7, 9).astype(f32)
	fx["f3_emb_words"], fx["f3_emb_vocab"], fx["f3_emb_g"] = words, vocab, g
	fx["f3_emb_orc_y"] = R.embed_fwd(words, vocab)
	fx["f3_emb_orc_vocab_after"] = R.embed_bwd_params(words, g, vocab.copy(), 0.25)

	# CTC: three samples, the second shorter in time, repeated labels, blank 0 and (second set) blank 3
	T, batch, vocab = 12, 3, 7
	scores = rng.randn(T, batch, vocab).astype(f32)
	datalen = np.array([12, 9, 12], np.int32)
	lengths = np.array([4, 3, 5], np.int32)
	fx["f3_ctc_scores"], fx["f3_ctc_datalen"], fx["f3_ctc_lengths"] = scores, datalen, lengths
	for tag, blank in (("b0", 0), ("b3", 3)):
		pool = [v for v in range(vocab) if v != blank]
		lab = np.array([pool[i] for i in rng.randint(0, len(pool), size=int(lengths.sum()))], np.int32)
		lab[1] = lab[0]                                        # a repeated label needs the blank between its two copies
		err, grad, alphas = R.ctc_loss(scores, datalen, lab, lengths, blank)
		fx["f3_ctc_%s_labels" % tag] = lab
		fx["f3_ctc_%s_orc_err" % tag], fx["f3_ctc_%s_orc_grad" % tag] = np.array([err], f32), grad


# ----------------------------------------------------------------------------------------------
# 3. LeNet: reference forward vs oracle runner; oracle full step
# ----------------------------------------------------------------------------------------------

def lenet_fixture():
	from PuzzleLib.Backend import gpuarray
	from PuzzleLib.Models.Nets.LeNet import loadLeNet

	np.random.seed(1234)                                   # TestLib/CnnMnistLenet.py:18
	net = loadLeNet(None, initscheme=None)
	data = np.random.randn(64, 1, 28, 28).astype(np.float32)
	labels = np.random.randint(0, 10, size=(64, )).astype(np.int32)

	net.evalMode()
	ref_logits = net(gpuarray.to_gpu(data)).get()

	params = {}
	for var, names in net.getVarTable().items():
		params[names[0]] = var.data.get()

	spec = nets.lenet_spec()
	cnet = N.CpuNet(spec, params)
	cnet.train = False
	close(ref_logits, cnet.forward(data), atol=1e-4, what="LeNet forward")

	cnet = N.CpuNet(spec, params)
	opt = N.CpuMomentumSGD(cnet, learnRate=0.1, momRate=0.9)     # TestLib/CnnMnistLenet.py optimizer
	pred, err = N.train_step(cnet, opt, data, labels)

	fx = {"labels": labels, "ref_logits": ref_logits, "orc_err": np.array([err], dtype=np.float32)}
	for k, v in cnet.params.items():
		flat = v.ravel()
		fx["orc_after_head_" + k] = flat[:256].copy()
		fx["orc_after_sum_" + k] = np.array([np.sum(flat, dtype=np.float64), np.sum(np.abs(flat), dtype=np.float64)])
		g = cnet.grads[k].ravel()
		fx["orc_grad_head_" + k] = g[:256].copy()
		fx["orc_grad_sum_" + k] = np.array([np.sum(g, dtype=np.float64), np.sum(np.abs(g), dtype=np.float64)])
		p0 = params[k].ravel()
		fx["ref_init_head_" + k] = p0[:64].copy()
		fx["ref_init_sum_" + k] = np.array([np.sum(p0, dtype=np.float64)])

	print("[3] LeNet b64: reference forward == oracle runner: OK")
	return fx


def miniresnet_fixture():
	rng = np.random.RandomState(4321)
	spec = nets.resnet_spec(stages=((8, 1), (16, 2)), classes=10, stem=8, softmax=False)
	pshapes, ashapes = nets.spec_param_shapes(spec)

	params = {}
	for k, shp in pshapes.items():
		if k.endswith(".W"):
			fan = int(np.prod(shp[1:])) if len(shp) == 4 else shp[0]
			params[k] = (rng.randn(*shp) * np.sqrt(2.0 / fan)).astype(np.float32)
		elif k.endswith(".scale"):
			params[k] = (1.0 + 0.1 * rng.randn(*shp)).astype(np.float32)
		else:
			params[k] = (0.1 * rng.randn(*shp)).astype(np.float32)

	attrs = {k: (np.zeros(shp, np.float32) if k.endswith(".mean") else np.ones(shp, np.float32)) for k, shp in
			 ashapes.items()}

	data = rng.randn(4, 3, 64, 64).astype(np.float32)
	labels = rng.randint(0, 10, size=(4, )).astype(np.int32)

	# avgpool 7x7 at the end expects 7x7 maps... 64 -> conv s2 32 -> pool 15 -> stage3 s2 8 ; use a 3-entry override
	spec = [l if l[0] != "avgpool" else ("avgpool", l[1], 8, 1, 0) for l in spec]

	cnet = N.CpuNet(spec, params, attrs)
	opt = N.CpuAdam(cnet, alpha=1e-3)
	pred, err = N.train_step(cnet, opt, data, labels)

	fx = {"data": data, "labels": labels, "orc_logits": pred, "orc_err": np.array([err], dtype=np.float32)}
	for k, v in params.items():
		fx["init_" + k] = v
	for k, v in cnet.params.items():
		fx["orc_after_" + k] = v
		fx["orc_grad_" + k] = cnet.grads[k]
	for k, v in cnet.attrs.items():
		fx["orc_attr_" + k] = v

	print("[4] mini-ResNet (2 stages, b4, 64x64) oracle step: err=%.6f" % err)
	return fx, spec


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--check", action="store_true")
	args = ap.parse_args()

	refimport.setup()

	ops = {}
	check_against_reference(ops)
	run_reference_tests_on_oracle()
	run_reference_f3_tests_on_oracle()
	f3_fixtures(ops)
	lenet = lenet_fixture()
	mini, minispec = miniresnet_fixture()

	if args.check:
		return

	out = os.path.join(ROOT, "tests", "golden")
	os.makedirs(out, exist_ok=True)

	np.savez_compressed(os.path.join(out, "ops.npz"), **ops)
	np.savez_compressed(os.path.join(out, "lenet.npz"), **lenet)
	np.savez_compressed(os.path.join(out, "miniresnet.npz"), **mini)

	manifest = {
		"generator": "oracle/make_golden.py",
		"reference": "puzzlelib/PuzzleLib v1.0.2 imported from /root/reference with Config.backend=cpu",
		"numpy": np.__version__,
		"key_prefixes": {
			"ref_": "computed by the reference itself (numpy CPU backend / gcc-JIT element-wise kernels)",
			"orc_": "computed by oracle/cpu_ref.py after it passed the reference comparison and the reference's own "
					"bnd-parameterised unit tests",
		},
		"miniresnet_spec": minispec,
		"files": {f: os.path.getsize(os.path.join(out, f)) for f in ("ops.npz", "lenet.npz", "miniresnet.npz")},
	}
	with open(os.path.join(out, "MANIFEST.json"), "w") as f:
		json.dump(manifest, f, indent=1)

	print("fixtures written to", out, manifest["files"])


if __name__ == "__main__":
	main()
