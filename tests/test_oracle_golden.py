"""
CPU tests: the oracle (oracle/cpu_ref.py, oracle/cpu_net.py) against the committed golden vectors.

`ref_*` arrays in tests/golden/ops.npz and lenet.npz were computed by the reference's own CPU backend (numpy
im2col/sgemm, gcc-JIT element-wise kernels) when oracle/make_golden.py ran in the build container; `orc_*` arrays are
oracle outputs stored after the oracle passed the reference comparison and the reference's bnd-parameterised unit tests
(regression pins). Nothing here needs /root/reference or a GPU.
"""
import json, os

import numpy as np
import pytest

import cpu_ref as R
import cpu_net as N
from conftest import assert_close, GOLDEN

CONV_CASES = ["c0", "c1", "c2", "c3", "c4"]


def test_manifest_describes_fixtures():
	manifest = json.load(open(os.path.join(GOLDEN, "MANIFEST.json")))
	assert manifest["generator"] == "oracle/make_golden.py"
	for name, size in manifest["files"].items():
		assert os.path.getsize(os.path.join(GOLDEN, name)) == size
	assert sum(manifest["files"].values()) < 2 * 1024 * 1024


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_against_reference_outputs(ops, case):
	x, w, b, dy = (ops["conv_%s_%s" % (case, k)] for k in ("x", "w", "b", "dy"))
	sh, sw, ph, pw, dh, dw, groups = (int(v) for v in ops["conv_%s_cfg" % case])
	kw = dict(stride=(sh, sw), pad=(ph, pw), dilation=(dh, dw), groups=groups)

	assert_close(R.conv2d_fwd(x, w, b, **kw), ops["conv_%s_ref_y" % case], what="forward == reference NumpyDnn.conv2d")

	# float64 accumulation agrees with the float32 fixtures to rounding
	assert_close(R.conv2d_bwd_data(dy, w, x.shape, acc=np.float64, **kw), ops["conv_%s_orc_dx" % case], atol=1e-4)
	dw_, db_ = R.conv2d_bwd_filter(x, dy, w.shape, withbias=True, acc=np.float64, **kw)
	assert_close(dw_, ops["conv_%s_orc_dw" % case], atol=1e-4)
	assert_close(db_, ops["conv_%s_orc_db" % case], atol=1e-4)

	# <dy, conv(x)> == <conv^T(dy), x> == <dw, w> (adjointness ties the three passes together)
	y0 = R.conv2d_fwd(x, w, None, acc=np.float64, **kw).astype(np.float64)
	lhs = np.sum(y0 * dy)
	assert np.isclose(lhs, np.sum(ops["conv_%s_orc_dx" % case].astype(np.float64) * x), rtol=1e-4)
	assert np.isclose(lhs, np.sum(ops["conv_%s_orc_dw" % case].astype(np.float64) * w), rtol=1e-4)


def test_conv_brute_force_small():
	"""Host loops of Cuda/Wrappers/CuDnn.py:29-80 (conv2dTest), written out once more independently of im2col."""
	import itertools
	rng = np.random.RandomState(0)
	n, c, h, w_, k, f, s = 1, 2, 6, 6, 4, 2, 2
	x = rng.randn(n, c, h, w_).astype(np.float32)
	w = rng.randn(k, c, f, f).astype(np.float32)
	y = R.conv2d_fwd(x, w, None, stride=s)

	ref = np.zeros(y.shape, dtype=np.float32)
	for b, oc, ic, yy, xx, dy, dx in itertools.product(range(n), range(k), range(c), range(y.shape[2]), range(y.shape[3]),
													 range(f), range(f)):
		ref[b, oc, yy, xx] += x[b, ic, yy * s + dy, xx * s + dx] * w[oc, ic, dy, dx]
	assert_close(y, ref)

	g = rng.randn(*y.shape).astype(np.float32)
	dxr, dwr = np.zeros(x.shape, np.float32), np.zeros(w.shape, np.float32)
	for b, ic, oc, yy, xx, dy, dx in itertools.product(range(n), range(c), range(k), range(g.shape[2]), range(g.shape[3]),
													 range(f), range(f)):
		dxr[b, ic, yy * s + dy, xx * s + dx] += w[oc, ic, dy, dx] * g[b, oc, yy, xx]
		dwr[oc, ic, dy, dx] += x[b, ic, yy * s + dy, xx * s + dx] * g[b, oc, yy, xx]
	assert_close(R.conv2d_bwd_data(g, w, x.shape, stride=s), dxr)
	assert_close(R.conv2d_bwd_filter(x, g, w.shape, stride=s), dwr)


def test_pool_bn_gemm_against_reference_outputs(ops):
	x = ops["pool_x"]
	for name in ("p0", "p1", "p2"):
		fh, fw, sh, sw, ph, pw = (int(v) for v in ops["pool_%s_cfg" % name])
		y = R.pool2d_fwd(x, (fh, fw), (sh, sw), (ph, pw), R.POOL_MAX)
		assert np.array_equal(y, ops["pool_%s_ref_max" % name])
		dx = R.pool2d_bwd(ops["pool_%s_dy" % name], x, y, (fh, fw), (sh, sw), (ph, pw), R.POOL_MAX)
		assert_close(dx, ops["pool_%s_orc_maxbwd" % name])
		assert np.isclose(dx.sum(), ops["pool_%s_dy" % name].sum(), rtol=1e-4)      # max-pool backward conserves mass

	bn = {k: ops["bn_" + k] for k in ("x", "scale", "bias", "mean", "var")}
	assert_close(R.bn_fwd_infer(bn["x"], bn["scale"], bn["bias"], bn["mean"], bn["var"]), ops["bn_ref_infer"])

	rm, rv = bn["mean"].copy(), bn["var"].copy()
	y, sm, si = R.bn_fwd_train(bn["x"], bn["scale"], bn["bias"], rm, rv, 1e-5, 0.25)
	assert_close(y, ops["bn_orc_train_y"])
	xh = (y - bn["bias"][None, :, None, None]) / bn["scale"][None, :, None, None]
	assert np.allclose(xh.mean(axis=(0, 2, 3)), 0, atol=1e-5) and np.allclose(xh.var(axis=(0, 2, 3)), 1, atol=1e-3)
	dx, ds, db = R.bn_bwd(ops["bn_dy"], bn["x"], bn["scale"], sm, si)
	assert_close(dx, ops["bn_orc_dx"])
	assert np.allclose(dx.sum(axis=(0, 2, 3)), 0, atol=1e-4)                         # BN backward removes the mean

	A, B = ops["gemm_A"], ops["gemm_B"]
	assert_close(R.gemm(A, B), ops["gemm_ref_nn"])
	assert_close(R.gemm(A, B, out=ops["gemm_C0"].copy(), alpha=0.5, beta=2.0), ops["gemm_orc_nn_ab"])
	assert_close(R.matsum(ops["mat_M"], 0), ops["mat_ref_colsum"])
	assert_close(R.add_vec_to_mat(ops["mat_v"], ops["mat_M"], 1), ops["mat_ref_biasadd"])
	assert np.array_equal(R.argmax(ops["mat_M"], 1), ops["mat_ref_argmax"])


def test_elementwise_against_reference_outputs(ops):
	x, g = ops["act_x"], ops["act_g"]
	table = {
		"sigmoid": (R.sigmoid, R.sigmoid_der, ()), "tanh": (R.tanh, R.tanh_der, ()), "relu": (R.relu, R.relu_der, ()),
		"leakyRelu": (R.leaky_relu, R.leaky_relu_der, (0.01, )), "elu": (R.elu, R.elu_der, (1.0, )),
		"softPlus": (R.softplus, R.softplus_der, ()), "clip": (R.clip, R.clip_der, (0.0, 6.0)),
	}
	for name, (fn, dfn, args) in table.items():
		y = ops["act_ref_%s" % name]
		assert_close(fn(x, *args), y, what=name)
		assert_close(dfn(g, y, *args), ops["act_ref_%s_der" % name], what=name + " der")

	assert_close(R.dropout(x, ops["drop_bits"], int(ops["drop_v"][0]), 0.5), ops["drop_ref"])
	assert_close(R.axpy(ops["elt_y0"].copy(), x, 0.3), ops["elt_ref_axpy"])
	assert_close(R.add_scaled(x, 0.7, ops["elt_y0"], -1.1), ops["elt_ref_add"])
	assert_close(R.linear(x, 1.5, -0.25), ops["elt_ref_linear"])
	assert_close(R.weight_decay(g.copy(), x, 1e-2), ops["elt_ref_wd"])


@pytest.mark.parametrize("tag,fn", [
	("adam", R.adam), ("classicMomSGD", R.classic_mom_sgd), ("nesterovMomSGD", R.nesterov_mom_sgd), ("rmsprop", R.rmsprop),
	("adagrad", R.adagrad), ("adadelta", R.adadelta), ("rmspropGraves", R.rmsprop_graves), ("smorms3", R.smorms3)
])
def test_optimizer_kernels_against_reference_outputs(ops, tag, fn):
	p = ops["opt_%s_p0" % tag].copy()
	st = [s.copy() for s in ops["opt_%s_st0" % tag]]
	scalars = [float(v) for v in ops["opt_%s_scalars" % tag]]
	for g in ops["opt_%s_grads" % tag]:
		fn(p, g, *st, *scalars)
	assert_close(p, ops["opt_%s_ref_p" % tag], what=tag)
	for a, ref in zip(st, ops["opt_%s_ref_st" % tag]):
		assert_close(a, ref, what=tag + " state")


def test_softmax_cross_entropy_properties(ops):
	y = R.softmax_fwd(ops["sm_x"])
	assert_close(y, ops["sm_orc_y"])
	assert np.allclose(y.sum(axis=1), 1.0, atol=1e-6)

	s, lab = ops["ce2_scores"], ops["ce2_labels"]
	err, grad = R.cross_entropy(s, lab)
	assert np.isclose(err, ops["ce2_orc_err"][0])
	assert np.allclose(grad.sum(axis=1), 0, atol=1e-6)

	# finite-difference check of the gradient direction (grads are descent directions: d(mean loss) = -<grad, ds>)
	eps, i, j = 1e-2, 3, 5
	sp = s.copy()
	sp[i, j] += eps
	errp, _ = R.cross_entropy(sp, lab)
	assert np.isclose((errp - err) / s.shape[0] / eps, -grad[i, j], rtol=5e-2, atol=1e-4)


def test_lenet_runner_reproduces_reference_logits(lenet_golden):
	from puzzlelib_amd import nets

	np.random.seed(1234)
	params = {}
	# same RNG call sequence as Models/Nets/LeNet.py with initscheme=None (xavier uniform, fan-in)
	for name, shape, fan in (("0.W", (16, 1, 3, 3), 9), ("3.W", (32, 16, 4, 4), 256), ("7.W", (800, 1024), 800),
							 ("9.W", (1024, 10), 1024)):
		bound = np.sqrt(3.0 / fan)
		params[name] = np.random.uniform(-bound, bound, shape).astype(np.float32)
	params.update({"0.b": np.zeros((1, 16, 1, 1), np.float32), "3.b": np.zeros((1, 32, 1, 1), np.float32),
				   "7.b": np.zeros(1024, np.float32), "9.b": np.zeros(10, np.float32)})

	for name in ("0.W", "3.W", "7.W", "9.W"):
		assert np.array_equal(params[name].ravel()[:64], lenet_golden["ref_init_head_" + name])

	data = np.random.randn(64, 1, 28, 28).astype(np.float32)
	labels = np.random.randint(0, 10, size=(64, )).astype(np.int32)

	net = N.CpuNet(nets.lenet_spec(), params)
	net.train = False
	assert_close(net.forward(data), lenet_golden["ref_logits"], atol=1e-4, what="LeNet forward == reference")

	net = N.CpuNet(nets.lenet_spec(), params)
	_, err = N.train_step(net, N.CpuMomentumSGD(net, 0.1, 0.9), data, labels)
	assert np.isclose(err, lenet_golden["orc_err"][0], rtol=1e-5)
	for k, v in net.params.items():
		assert_close(v.ravel()[:256], lenet_golden["orc_after_head_" + k], atol=1e-5, what=k)


def test_miniresnet_runner_regression(mini_golden):
	from puzzlelib_amd import nets
	spec = nets.resnet_spec(stages=((8, 1), (16, 2)), classes=10, stem=8, softmax=False)
	spec = [l if l[0] != "avgpool" else ("avgpool", l[1], 8, 1, 0) for l in spec]

	params = {k[5:]: mini_golden[k] for k in mini_golden.keys() if k.startswith("init_")}
	_, ashapes = nets.spec_param_shapes(spec)
	attrs = {k: (np.zeros(s, np.float32) if k.endswith(".mean") else np.ones(s, np.float32)) for k, s in ashapes.items()}

	net = N.CpuNet(spec, params, attrs)
	pred, err = N.train_step(net, N.CpuAdam(net, alpha=1e-3), mini_golden["data"], mini_golden["labels"])
	assert_close(pred, mini_golden["orc_logits"], atol=1e-4)
	assert np.isclose(err, mini_golden["orc_err"][0], rtol=1e-5)


def test_allreduce_mean_restatement():
	rng = np.random.RandomState(1)
	grads = [rng.randn(1000).astype(np.float32) for _ in range(4)]
	assert_close(R.grad_mean_allreduce(grads), np.mean(np.stack(grads).astype(np.float64), axis=0), atol=1e-6)


def test_f3_oracle_reproduces_its_fixtures_and_independent_formulas(ops):
	"""The oracle of the operators beside the hot path against the committed `f3_*` vectors (written after the reference's
	own tests for them passed on this oracle, oracle/make_golden.py step 2b) and against formulas written independently
	here: numpy's reflect padding, adjoint identities for the backward passes, brute-force loops on small cases."""
	# reflection pad == np.pad(mode="reflect"); backward is its adjoint
	x, g = ops["f3_pad2_x"], ops["f3_pad2_g"]
	pad = tuple(int(v) for v in ops["f3_pad2_pad"])
	y = R.reflectpad_fwd(x, pad)
	assert np.array_equal(y, np.pad(x, ((0, 0), (0, 0), pad[:2], pad[2:]), mode="reflect")) and np.array_equal(y, ops["f3_pad2_orc_y"])
	assert np.isclose(np.sum(y.astype(np.float64) * g), np.sum(R.reflectpad_bwd(g, pad).astype(np.float64) * x), rtol=1e-6)

	# up-sampling: nearest == np.repeat; linear reproduces the corners and is exact on affine ramps; backward = adjoint
	for tag in ("2d", "3d"):
		x, scale = ops["f3_up%s_x" % tag], tuple(int(v) for v in ops["f3_up%s_scale" % tag])
		near = x
		for axis, s in enumerate(scale):
			near = np.repeat(near, s, axis=2 + axis)
		assert np.array_equal(R.upsample_fwd(x, scale, "nearest"), near)
		lin = R.upsample_fwd(x, scale, "linear")
		assert_close(lin, ops["f3_up%s_orc_linear_y" % tag], atol=1e-6)
		corner = (slice(None), slice(None)) + (slice(None, None, None), ) * 0
		assert np.allclose(lin[(Ellipsis, ) + (0, ) * (x.ndim - 2)], x[(Ellipsis, ) + (0, ) * (x.ndim - 2)], atol=1e-6)
		assert np.allclose(lin[(Ellipsis, ) + (-1, ) * (x.ndim - 2)], x[(Ellipsis, ) + (-1, ) * (x.ndim - 2)], atol=1e-6)
		for mode in ("nearest", "linear"):
			g = ops["f3_up%s_%s_g" % (tag, mode)]
			lhs = np.sum(R.upsample_fwd(x, scale, mode).astype(np.float64) * g)
			assert np.isclose(lhs, np.sum(R.upsample_bwd(g, scale, mode).astype(np.float64) * x), rtol=1e-5)
			assert_close(R.upsample_bwd(g, scale, mode), ops["f3_up%s_orc_%s_dx" % (tag, mode)], atol=1e-6)
	ramp = np.arange(5, dtype=np.float32).reshape(1, 1, 1, 5) * np.ones((1, 1, 3, 1), np.float32)
	assert np.allclose(R.upsample_fwd(ramp, (1, 3), "linear")[0, 0, 0], np.linspace(0, 4, 15), atol=1e-5)

	# 3-d convolution: forward vs brute force on one output element, backward passes as adjoints
	cfg = [int(v) for v in ops["f3_c3_cfg"]]
	st, pd, dl = tuple(cfg[0:3]), tuple(cfg[3:6]), tuple(cfg[6:9])
	x, w, dy = ops["f3_c3_x"], ops["f3_c3_w"], ops["f3_c3_dy"]
	y = R.conv3d_fwd(x, w, None, st, pd, dl)
	xp = np.pad(x.astype(np.float64), ((0, 0), (0, 0)) + tuple((p, p) for p in pd))
	n, k, d, h, ww = 1, 2, 1, 2, 1
	acc = 0.0
	for c in range(x.shape[1]):
		for t in range(w.shape[2]):
			for r in range(w.shape[3]):
				for s in range(w.shape[4]):
					acc += xp[n, c, d * st[0] + t, h * st[1] + r, ww * st[2] + s] * w[k, c, t, r, s]
	assert np.isclose(y[n, k, d, h, ww], acc, rtol=1e-9)
	lhs = np.sum(y * dy)
	assert np.isclose(lhs, np.sum(R.conv3d_bwd_data(dy, w, x.shape, st, pd, dl) * x), rtol=1e-9)
	assert np.isclose(lhs, np.sum(R.conv3d_bwd_filter(x, dy, w.shape, st, pd, dl)[0] * w), rtol=1e-9)

	# instance norm: per (image, map) zero mean / unit variance before the affine pair; backward via a finite difference
	x, scale, bias = ops["f3_in_x"], ops["f3_in_scale"], ops["f3_in_bias"]
	y, sm, si, ext = R.instance_norm_fwd(x, scale, bias)
	norm = (y - bias.reshape(1, -1, 1, 1)) / scale.reshape(1, -1, 1, 1)
	assert np.allclose(norm.mean(axis=(2, 3)), 0, atol=1e-5) and np.allclose(norm.var(axis=(2, 3)), 1, atol=1e-3)
	assert_close(y, ops["f3_in_orc_y"], atol=1e-6)
	dy = ops["f3_in_dy"]
	dx, ds, db = R.instance_norm_bwd(dy, x, ext, sm, si)
	probe = np.zeros_like(x)
	probe[1, 2, 3, 4] = 1e-2
	num = (np.sum(R.instance_norm_fwd(x + probe, scale, bias)[0].astype(np.float64) * dy) -
		   np.sum(R.instance_norm_fwd(x - probe, scale, bias)[0].astype(np.float64) * dy)) / 2e-2
	assert np.isclose(dx[1, 2, 3, 4], num, rtol=2e-2, atol=1e-3)
	assert_close(db, dy.sum(axis=(0, 2, 3)), atol=1e-4)

	# cost kernels: gradients are the derivatives of the summed error (finite differences where the kernel is smooth)
	s, lab = ops["f3_bce_scores"].astype(np.float64), ops["f3_bce_labels"]
	prob = 1 / (1 + np.exp(-s.reshape(lab.shape)))
	assert np.isclose(float(ops["f3_bce_orc_err"][0]), -np.sum(lab * np.log(prob) + (1 - lab) * np.log(1 - prob)) / 12, rtol=1e-5)
	assert_close(ops["f3_bce_orc_grad"].reshape(lab.shape), (lab - prob) / 12 / 12, atol=1e-7)
	p, t = ops["f3_sl1_pred"].astype(np.float64), ops["f3_sl1_target"].astype(np.float64)
	d = np.abs(p - t)
	assert np.isclose(float(ops["f3_sl1_orc_err"][0]), np.sum(np.where(d < 1, d * d / 2, d - 0.5)) / 11, rtol=1e-5)

	# embedding: rows gathered, padding rows zero; the update adds scale * grad once per occurrence
	words, vocab, g = ops["f3_emb_words"], ops["f3_emb_vocab"], ops["f3_emb_g"]
	after = vocab.astype(np.float64).copy()
	for b in range(words.shape[0]):
		for t_ in range(words.shape[1]):
			if words[b, t_] != -1:
				after[words[b, t_]] += 0.25 * g[b, t_]
	assert_close(ops["f3_emb_orc_vocab_after"], after, atol=1e-5)

	# PReLU slope gradient (per map and shared) by direct summation
	x, dy = ops["f3_prelu_x"].astype(np.float64), ops["f3_prelu_dy"].astype(np.float64)
	per = (dy * x * (x <= 0)).sum(axis=(0, 2, 3))
	assert_close(ops["f3_prelu_orc_map_ds"], per, atol=1e-4)
	assert_close(ops["f3_prelu_orc_shared_ds"], [per.sum()], atol=1e-4)


def test_batchnorm_running_variance_convention_known_answer():
	"""closed-form case for the convention no reference test pins: running variance <- UNBIASED batch variance (what the
	reference's MIOpen / cuDNN calls write, Hip/Wrappers/MIOpen.py:656-660), saveinvvar from the biased one. Same case as the
	device test (tests/test_gpu_0_ops.py::running_variance_known_answer)."""
	c = 3
	x = np.empty((2, c, 2, 2), np.float32)
	for ch in range(c):
		x[:, ch] = ((ch + np.arange(8, dtype=np.float32)) * (ch + 1)).reshape(2, 2, 2)
	k = np.arange(1, c + 1, dtype=np.float64)
	rm, rv = np.full(c, 2.0, np.float32), np.full(c, 10.0, np.float32)
	_, sm, si = R.bn_fwd_train(x, np.ones(c, np.float32), np.zeros(c, np.float32), rm, rv, 1e-5, 0.25)
	assert np.allclose(sm, (np.arange(c) + 3.5) * k, atol=1e-5)
	assert np.allclose(si, 1.0 / np.sqrt(5.25 * k * k + 1e-5), rtol=1e-5)
	assert np.allclose(rm, 0.75 * 2.0 + 0.25 * (np.arange(c) + 3.5) * k, atol=1e-5)
	assert np.allclose(rv, 0.75 * 10.0 + 0.25 * 6.0 * k * k, rtol=1e-6), "running variance must use m/(m-1) * biased variance"
