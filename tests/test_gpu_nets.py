"""
GPU parity tests on whole networks through the module layer: LeNet b64 (config 1 — the forward logits in the fixture
were computed by the REFERENCE's own CPU backend on the same seed), a two-stage mini-ResNet training step (conv, BN,
ReLU, pooling, residual add, linear, cross-entropy, Adam; oracle fixture) and the Trainer/Validator loop.
Tolerances: forward logits atol 1e-4; parameters after one step atol 2e-5 (updates are O(lr)); gradients rtol 1e-3.
"""
import numpy as np
import pytest

import cpu_ref as R
import cpu_net as N
from conftest import assert_close

pytestmark = pytest.mark.gpu


def test_lenet_reference_forward_and_step(bnd, lenet_golden):
	from puzzlelib_amd import nets, train
	from puzzlelib_amd.surface import bound

	gpuarray = bound().gpuarray

	np.random.seed(1234)                               # TestLib/CnnMnistLenet.py:18
	net = nets.loadLeNet(None, initscheme=None)
	data = np.random.randn(64, 1, 28, 28).astype(np.float32)
	labels = np.random.randint(0, 10, size=(64, )).astype(np.int32)
	assert np.array_equal(labels, lenet_golden["labels"])

	variables = nets.namedVariables(net)
	for name, var in variables.items():
		head = var.data.get().ravel()[:64]
		assert np.array_equal(head, lenet_golden["ref_init_head_" + name]), "same seed must give the reference's init: " + name

	net.evalMode()
	logits = net(gpuarray.to_gpu(data)).get()
	assert_close(logits, lenet_golden["ref_logits"], atol=1e-4, rtol=1e-4, what="LeNet forward vs reference CPU backend")

	optimizer = train.MomentumSGD(learnRate=0.1, momRate=0.9)
	optimizer.setupOn(net, useGlobalState=True)
	cost = train.CrossEntropy()
	trainer = train.Trainer(net, cost, optimizer, batchsize=64)
	trainer.train(gpuarray.to_gpu(data), gpuarray.to_gpu(labels), random=False)

	assert np.isclose(cost.getMeanError() * 64, lenet_golden["orc_err"][0], rtol=1e-4)

	for name, var in nets.namedVariables(net).items():
		p, g = var.data.get().ravel(), var.grad.get().ravel()
		assert_close(g[:256], lenet_golden["orc_grad_head_" + name], atol=1e-5, rtol=1e-3, what="grad " + name)
		assert_close(p[:256], lenet_golden["orc_after_head_" + name], atol=2e-5, rtol=1e-4, what="param " + name)
		ref_sum, ref_abs = lenet_golden["orc_after_sum_" + name]
		assert abs(p.sum(dtype=np.float64) - ref_sum) <= 1e-3 + 1e-4 * ref_abs, "param checksum " + name


def build_mini(mini_golden):
	from puzzlelib_amd import nets
	from puzzlelib_amd.surface import bound

	spec = nets.resnet_spec(stages=((8, 1), (16, 2)), classes=10, stem=8, softmax=False)
	spec = [l if l[0] != "avgpool" else ("avgpool", l[1], 8, 1, 0) for l in spec]

	np.random.seed(7)
	net = nets.build(spec, name="mini", initscheme="he")

	for name, var in nets.namedVariables(net).items():
		var.data.set(mini_golden["init_" + name])
	return net, spec, bound().gpuarray


def test_mini_resnet_training_step(bnd, mini_golden):
	from puzzlelib_amd import nets, train

	net, spec, gpuarray = build_mini(mini_golden)
	data, labels = mini_golden["data"], mini_golden["labels"]

	optimizer = train.Adam(alpha=1e-3)
	optimizer.setupOn(net, useGlobalState=True)
	cost = train.CrossEntropy()

	net.trainMode()
	pred = net(gpuarray.to_gpu(data))
	assert_close(pred.get(), mini_golden["orc_logits"], atol=2e-4, rtol=1e-3, what="logits")

	grad = cost(pred, gpuarray.to_gpu(labels), queryError=False)
	assert np.isclose(cost.devErr.get(), mini_golden["orc_err"][0], rtol=1e-4)

	optimizer.zeroGradParams()
	net.backward(grad, updGrad=False)

	for name, var in nets.namedVariables(net).items():
		ref = mini_golden["orc_grad_" + name]
		scale = np.abs(ref).max() + 1e-6
		assert_close(var.grad.get(), ref, atol=2e-3 * scale, rtol=2e-3, what="grad " + name)

	optimizer.update()

	for name, var in nets.namedVariables(net).items():
		assert_close(var.data.get(), mini_golden["orc_after_" + name], atol=3e-4, rtol=1e-4, what="param " + name)
	for name, attr in nets.namedAttrs(net).items():
		assert_close(attr.get(), mini_golden["orc_attr_" + name], atol=1e-5, rtol=1e-4, what="running stat " + name)


def test_mini_resnet_matches_oracle_for_several_steps(bnd, mini_golden):
	"""3 Adam steps on device vs 3 oracle steps (loss trajectory; Adam's sign-like first steps amplify tiny gradient
	differences, hence the loose parameter tolerance — the loss is the invariant that is checked tightly)."""
	from puzzlelib_amd import nets, train

	net, spec, gpuarray = build_mini(mini_golden)
	data, labels = mini_golden["data"], mini_golden["labels"]

	params = {k[5:]: mini_golden[k] for k in mini_golden.keys() if k.startswith("init_")}
	pshapes, ashapes = nets.spec_param_shapes(spec)
	attrs = {k: (np.zeros(s, np.float32) if k.endswith(".mean") else np.ones(s, np.float32)) for k, s in ashapes.items()}
	cnet = N.CpuNet(spec, params, attrs)
	copt = N.CpuAdam(cnet, alpha=1e-3)

	optimizer = train.Adam(alpha=1e-3)
	optimizer.setupOn(net, useGlobalState=True)
	cost = train.CrossEntropy()
	trainer = train.Trainer(net, cost, optimizer, batchsize=4)

	gdata, glabels = gpuarray.to_gpu(data), gpuarray.to_gpu(labels)
	for step in range(3):
		_, err = N.train_step(cnet, copt, data, labels)
		trainer.train(gdata, glabels, random=False)
		assert np.isclose(cost.getMeanError() * 4, err, rtol=2e-3), "step %d: device %s vs oracle %s" % (
			step, cost.getMeanError() * 4, err
		)


def test_inplace_relu_fusion_is_bit_identical_and_matches_oracle(bnd, mini_golden):
	"""actInplace=True (Models/Nets/ResNet.py:33,58) lets Sequential fold every ReLU into its neighbours (BN+ReLU,
	Add+ReLU, ReLU-derivative into the Replicate fan-in or the BN backward). One Adam step must give the same bits as
	the unfused module sequence, and the oracle's loss / parameters within the usual tolerances."""
	from puzzlelib_amd import nets, train, nn
	from puzzlelib_amd.surface import bound

	gpuarray = bound().gpuarray
	spec = nets.resnet_spec(stages=((8, 1), (16, 2)), classes=10, stem=8, softmax=False)
	spec = [l if l[0] != "avgpool" else ("avgpool", l[1], 8, 1, 0) for l in spec]
	data, labels = mini_golden["data"], mini_golden["labels"]

	results = {}
	from puzzlelib_amd import backend
	dnn = bnd.dnn
	for fused in (False, True, "no-bn-add", "no-gate-stats", "no-strided-grad", "no-relu-mask", "no-overlap"):
		backend.DnnContext.overlapFilterGrad = fused != "no-overlap"    # else filter gradients stay on the main stream
		launched = getattr(dnn, "sideLaunches", 0)
		nn.Sequential.fuseBnBackward = False           # covered by its own test below (same values up to fp32 rounding, not bit-identical)
		nn.Sequential.fuseInplaceRelu = bool(fused)
		nn.Sequential.fuseBnAdd = fused in (True, "no-gate-stats", "no-strided-grad", "no-relu-mask", "no-overlap")      # else the residual Add reads materialised BN outputs
		nn.Sequential.fuseGateStats = fused in (True, "no-bn-add", "no-strided-grad", "no-relu-mask", "no-overlap")      # else BN backward sums its own statistics
		nn.Sequential.fuseStridedGrad = fused != "no-strided-grad"      # else the stride-2 1x1 convolutions zero-fill their input gradients
		nn.Sequential.fuseReluMask = fused != "no-relu-mask"            # else the fan-in reads the block output back for its sign
		try:
			np.random.seed(7)
			net = nets.build(spec, name="mini", initscheme="he", actInplace=True)
			for name, var in nets.namedVariables(net).items():
				var.data.set(mini_golden["init_" + name])

			optimizer = train.Adam(alpha=1e-3)
			optimizer.setupOn(net, useGlobalState=True)
			cost = train.CrossEntropy()
			net.trainMode()

			pred = net(gpuarray.to_gpu(data))
			logits = pred.get()
			grad = cost(pred, gpuarray.to_gpu(labels), queryError=False)
			optimizer.zeroGradParams()
			net.backward(grad, updGrad=False)
			grads = {name: var.grad.get() for name, var in nets.namedVariables(net).items()}
			optimizer.update()
			params = {name: var.data.get() for name, var in nets.namedVariables(net).items()}

			convs = [m for m in allModules(net) if isinstance(m, nn.Conv2D)]
			adds = [m for m in allModules(net) if isinstance(m, nn.Add)]
			assert getattr(dnn, "sideLaunches", 0) - launched == (0 if fused == "no-overlap" else len(convs)), \
				"every convolution's filter gradient went to the side stream (and none with the switch off)"
			assert sum(m.reluMask is not None for m in adds) == (2 if fused in (True, "no-strided-grad", "no-overlap") else 0), \
				"the Adds of blocks 1 and 2 leave the sign mask their fan-in gates with"
			assert sum(m.compactGrad for m in convs) == (2 if fused in (True, "no-bn-add", "no-relu-mask", "no-overlap") else 0), \
				"the down-sampling block's two stride-2 1x1 convolutions keep their input gradients compact"
			if fused is True:
				reps = [m for m in allModules(net) if isinstance(m, nn.Replicate)]
				assert sum(len(m.statsFor) for m in reps) == 4, "blocks 1 and 2 (each with a projection BN) hand their BN statistics to the next fan-in"
				bns = [m for m in allModules(net) if isinstance(m, nn.BatchNorm2D)]
				assert sum(m.deferApply for m in bns) == 5, "3 blocks' branch-tail BNs + 2 projection-shortcut BNs feed an Add"
			if fused:
				acts = [m for m in allModules(net) if isinstance(m, nn.Activation)]
				assert acts and all(m.dataFused and m.gradFused for m in acts[:-1]), "every inner ReLU must have been absorbed"
				assert acts[-1].dataFused          # last block's ReLU: forward fused into Add, derivative computed by itself

			results[fused] = (logits, float(cost.devErr.get()), grads, params)
		finally:
			nn.Sequential.fuseInplaceRelu = nn.Sequential.fuseBnAdd = nn.Sequential.fuseGateStats = True
			nn.Sequential.fuseBnBackward = nn.Sequential.fuseStridedGrad = nn.Sequential.fuseReluMask = True
			backend.DnnContext.overlapFilterGrad = True

	(l1, e1, g1, p1) = results[True]
	for other in (False, "no-bn-add", "no-gate-stats", "no-strided-grad", "no-relu-mask", "no-overlap"):
		l0, e0, g0, p0 = results[other]
		assert np.array_equal(l0, l1) and e0 == e1
		for name in g0:
			assert np.array_equal(g0[name], g1[name]), "grad %s (%s)" % (name, other)
			assert np.array_equal(p0[name], p1[name]), "param %s (%s)" % (name, other)

	assert_close(l1, mini_golden["orc_logits"], atol=2e-4, rtol=1e-3, what="logits")
	assert np.isclose(e1, mini_golden["orc_err"][0], rtol=1e-4)
	for name in p1:
		assert_close(p1[name], mini_golden["orc_after_" + name], atol=3e-4, rtol=1e-4, what="param " + name)


@pytest.mark.parametrize("planes", [16, 6])
def test_batchnorm_backward_folded_into_the_convolution(bnd, mini_golden, planes):
	"""Conv2D -> BatchNorm2D backward with the BN's apply pass evaluated inside the convolution's backward-data /
	backward-filter gathers (pz_conv2d_bwd_*_bn): gradients equal the unfolded path to fp32 rounding and the oracle to the
	usual tolerances; the folded kernels must actually have been taken."""
	from puzzlelib_amd import nets, train, nn, backend
	from puzzlelib_amd.surface import bound

	gpuarray = bound().gpuarray
	# planes 16: block outputs of 64 / 128 maps, eligible 1x1 convolutions; planes 6: 24 / 48 maps, not a multiple of 16 ->
	# the convolution declines and the handle is materialised by the BN's own apply pass (same numbers either way)
	spec = nets.resnet_spec(stages=((planes, 1), (2 * planes, 2)), classes=10, stem=16, softmax=False)
	spec = [l if l[0] != "avgpool" else ("avgpool", l[1], 8, 1, 0) for l in spec]
	rng = np.random.RandomState(3)
	data = rng.randn(4, 3, 64, 64).astype(np.float32)
	labels = rng.randint(0, 10, size=(4, )).astype(np.int32)

	taken = []
	original = backend.DnnContext.convNdBackwardData

	def spy(self, grad, *args, **kwargs):
		taken.append(isinstance(grad, backend.DeferredBNGrad))
		return original(self, grad, *args, **kwargs)

	results = {}
	for fold in (False, True):
		nn.Sequential.fuseBnBackward = fold
		backend.DnnContext.convNdBackwardData = spy
		taken.clear()
		try:
			np.random.seed(7)
			net = nets.build(spec, name="mini16", initscheme="he", actInplace=True)
			init = {name: var.data.get() for name, var in nets.namedVariables(net).items()}
			optimizer = train.Adam(alpha=1e-3)
			optimizer.setupOn(net, useGlobalState=True)
			cost = train.CrossEntropy()
			net.trainMode()
			pred = net(gpuarray.to_gpu(data))
			grad = cost(pred, gpuarray.to_gpu(labels), queryError=False)
			optimizer.zeroGradParams()
			net.backward(grad, updGrad=False)
			results[fold] = ({name: var.grad.get() for name, var in nets.namedVariables(net).items()}, init, sum(taken))
		finally:
			backend.DnnContext.convNdBackwardData = original
			nn.Sequential.fuseBnBackward = True

	(g0, init0, n0), (g1, init1, n1) = results[False], results[True]
	assert n0 == 0 and n1 >= 3, "the BN handed its convolution a handle %d times" % n1
	for name in g0:
		scale = np.abs(g0[name]).max() + 1e-12
		assert_close(g1[name], g0[name], atol=2e-5 * scale, rtol=2e-4, what="grad " + name)

	# oracle
	params = init1
	_, ashapes = nets.spec_param_shapes(spec)
	attrs = {k: (np.zeros(s, np.float32) if k.endswith(".mean") else np.ones(s, np.float32)) for k, s in ashapes.items()}
	cnet = N.CpuNet(spec, params, attrs)
	cnet.train = True
	pred_ref = cnet.forward(data)
	_, grad_ref = R.cross_entropy(pred_ref, labels)
	cnet.zero_grads()
	cnet.backward(grad_ref)
	for name in g1:
		ref = cnet.grads[name]
		scale = np.abs(ref).max() + 1e-6
		assert_close(g1[name], ref, atol=2e-3 * scale, rtol=2e-3, what="grad vs oracle " + name)


def allModules(container):
	from puzzlelib_amd import nn
	for mod in (container.graph if hasattr(container, "graph") else container.modules.values()):
		if isinstance(mod, nn.Container):
			yield from allModules(mod)
		else:
			yield mod


def test_validator_and_eval_mode(bnd, mini_golden):
	from puzzlelib_amd import train

	net, spec, gpuarray = build_mini(mini_golden)
	data, labels = mini_golden["data"], mini_golden["labels"]

	validator = train.Validator(net, train.CrossEntropy(), batchsize=2)
	err = validator.validate(gpuarray.to_gpu(data), gpuarray.to_gpu(labels))

	params = {k[5:]: mini_golden[k] for k in mini_golden.keys() if k.startswith("init_")}
	from puzzlelib_amd import nets
	_, ashapes = nets.spec_param_shapes(spec)
	attrs = {k: (np.zeros(s, np.float32) if k.endswith(".mean") else np.ones(s, np.float32)) for k, s in ashapes.items()}
	cnet = N.CpuNet(spec, params, attrs)
	cnet.train = False
	pred = cnet.forward(data)
	assert err == np.mean(np.argmax(pred, axis=1) != labels)


def test_nin_forward_backward_runs_and_matches_oracle(bnd):
	"""Config 3 (CIFAR-10 NiN): one training step with a fixed dropout mask on a reduced batch vs the oracle."""
	from puzzlelib_amd import nets, train, nn
	from puzzlelib_amd.surface import bound

	gpuarray = bound().gpuarray
	np.random.seed(1234)
	net = nets.buildNiN()
	spec = nets.nin_spec()

	rng = np.random.RandomState(5)
	data = rng.randn(8, 3, 32, 32).astype(np.float32)
	labels = rng.randint(0, 10, size=(8, )).astype(np.int32)

	params = {name: var.data.get() for name, var in nets.namedVariables(net).items()}
	cnet = N.CpuNet(spec, params)

	# the device RNG is Philox, not the reference's XORWOW: parity is checked with the mask fed to the oracle
	class FixedRng:
		def __init__(self):
			self.masks = {}
		def fillInteger(self, ary):
			bnd.globalRng.fillInteger(ary)
			self.last = ary.get()

	for mod in net.getAllByType(nn.Dropout):
		mod.rng = FixedRng()

	net.trainMode()
	pred = net(gpuarray.to_gpu(data))
	for mod in net.getAllByType(nn.Dropout):
		cnet.dropmasks[mod.name] = mod.rng.last

	cnet.train = True
	pred_ref = cnet.forward(data)
	assert_close(pred.get(), pred_ref, atol=1e-4, rtol=1e-3, what="NiN forward")

	cost = train.CrossEntropy()
	grad = cost(pred, gpuarray.to_gpu(labels), queryError=False)
	err_ref, grad_ref = R.cross_entropy(pred_ref, labels)

	for var in nets.namedVariables(net).values():
		var.grad.fill(0)
	net.backward(grad, updGrad=False)
	cnet.zero_grads()
	cnet.backward(grad_ref)

	for name, var in nets.namedVariables(net).items():
		ref = cnet.grads[name]
		got = var.grad.get()
		scale = np.abs(ref).max() + 1e-8
		# nine ReLU layers deep, a pre-activation within rounding of zero may gate differently on the two sides and moves
		# single gradient entries by O(1e-2 * max): the tensor as a whole is held to 2e-3 (relative L2), entries to 2e-2 * max
		rel = np.linalg.norm((got - ref).astype(np.float64)) / (np.linalg.norm(ref.astype(np.float64)) + 1e-30)
		assert rel < 2e-3, "NiN grad %s: relative L2 error %.3e" % (name, rel)
		assert_close(got, ref, atol=2e-2 * scale, rtol=5e-3, what="NiN grad " + name)
