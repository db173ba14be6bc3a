"""
GPU parity tests on whole networks through the executor (puzzlelib_amd/engine.py + optim.py), which sends the backend
the reference modules' call sequence (tests/test_dryrun_backend.py): LeNet b64 (config 1 — the forward logits in the
fixture were computed by the REFERENCE's own CPU backend on the same seed), a two-stage mini-ResNet training step (conv,
BN, ReLU, pooling, residual add, linear, cross-entropy, Adam; oracle fixture), NiN (config 3) and the batch loops.
Tolerances: forward logits atol 1e-4; parameters after one step atol 2e-5 (updates are O(lr)); gradients rtol 1e-3.
"""
import os

import numpy as np
import pytest

import cpu_ref as R
import cpu_net as N
from conftest import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def cleanSwitches():
	from puzzlelib_amd import lazy, backend, engine
	lazy.enabled, lazy.disabled = True, set()
	backend.DnnContext.convStatsPolicy = "adaptive"
	engine.Net.skipInputGrad = False
	yield
	lazy.enabled, lazy.disabled = True, set()
	backend.DnnContext.convStatsPolicy = "adaptive"
	engine.Net.skipInputGrad = False


def test_lenet_reference_forward_and_step(bnd, lenet_golden):
	from puzzlelib_amd import nets, optim
	from puzzlelib_amd.surface import bound

	gpuarray = bound().gpuarray

	np.random.seed(1234)                               # TestLib/CnnMnistLenet.py:18
	net = nets.loadLeNet(None, initscheme=None)
	data = np.random.randn(64, 1, 28, 28).astype(np.float32)
	labels = np.random.randint(0, 10, size=(64, )).astype(np.int32)
	assert np.array_equal(labels, lenet_golden["labels"])

	for name, param in net.namedParams().items():
		head = param.data.get().ravel()[:64]
		assert np.array_equal(head, lenet_golden["ref_init_head_" + name]), "same seed must give the reference's init: " + name

	net.evalMode()
	logits = net(gpuarray.to_gpu(data)).get()
	assert_close(logits, lenet_golden["ref_logits"], atol=1e-4, rtol=1e-4, what="LeNet forward vs reference CPU backend")

	optimizer = optim.MomentumSGD(learnRate=0.1, momRate=0.9)
	optimizer.setupOn(net, useGlobalState=True)
	cost = optim.CrossEntropy()
	trainer = optim.Trainer(net, cost, optimizer, batchsize=64)
	trainer.train(gpuarray.to_gpu(data), gpuarray.to_gpu(labels), random=False)

	assert np.isclose(cost.getMeanError() * 64, lenet_golden["orc_err"][0], rtol=1e-4)

	for name, param in net.namedParams().items():
		p, g = param.data.get().ravel(), param.grad.get().ravel()
		assert_close(g[:256], lenet_golden["orc_grad_head_" + name], atol=1e-5, rtol=1e-3, what="grad " + name)
		assert_close(p[:256], lenet_golden["orc_after_head_" + name], atol=2e-5, rtol=1e-4, what="param " + name)
		ref_sum, ref_abs = lenet_golden["orc_after_sum_" + name]
		assert abs(p.sum(dtype=np.float64) - ref_sum) <= 1e-3 + 1e-4 * ref_abs, "param checksum " + name


def miniSpec(planes=8, stem=8):
	from puzzlelib_amd import nets
	spec = nets.resnet_spec(stages=((planes, 1), (2 * planes, 2)), classes=10, stem=stem, softmax=False)
	return [l if l[0] != "avgpool" else ("avgpool", l[1], 8, 1, 0) for l in spec]


def build_mini(mini_golden, actInplace=False):
	from puzzlelib_amd import nets
	from puzzlelib_amd.surface import bound

	spec = miniSpec()
	np.random.seed(7)
	net = nets.build(spec, name="mini", initscheme="he", actInplace=actInplace)
	for name, param in net.namedParams().items():
		param.data.set(mini_golden["init_" + name])
	return net, spec, bound().gpuarray


@pytest.mark.parametrize("actInplace", [False, True])
def test_mini_resnet_training_step(bnd, mini_golden, actInplace):
	from puzzlelib_amd import optim

	net, spec, gpuarray = build_mini(mini_golden, actInplace)
	data, labels = mini_golden["data"], mini_golden["labels"]

	optimizer = optim.Adam(alpha=1e-3)
	optimizer.setupOn(net, useGlobalState=True)
	cost = optim.CrossEntropy()

	net.trainMode()
	pred = net(gpuarray.to_gpu(data))
	assert_close(pred.get(), mini_golden["orc_logits"], atol=2e-4, rtol=1e-3, what="logits")

	grad = cost(pred, gpuarray.to_gpu(labels), queryError=False)
	assert np.isclose(cost.devErr.get(), mini_golden["orc_err"][0], rtol=1e-4)

	optimizer.zeroGradParams()
	net.backward(grad, updGrad=False)

	for name, param in net.namedParams().items():
		ref = mini_golden["orc_grad_" + name]
		scale = np.abs(ref).max() + 1e-6
		assert_close(param.grad.get(), ref, atol=2e-3 * scale, rtol=2e-3, what="grad " + name)

	optimizer.update()

	for name, param in net.namedParams().items():
		assert_close(param.data.get(), mini_golden["orc_after_" + name], atol=3e-4, rtol=1e-4, what="param " + name)
	for name, attr in net.namedAttrs().items():
		assert_close(attr.get(), mini_golden["orc_attr_" + name], atol=1e-5, rtol=1e-4, what="running stat " + name)


def test_mini_resnet_matches_oracle_for_several_steps(bnd, mini_golden):
	"""3 Adam steps on device vs 3 oracle steps (loss trajectory; Adam's sign-like first steps amplify tiny gradient
	differences, hence the loss is the invariant that is checked). From the second step on the batch-norm statistics come
	from the convolutions' epilogues (adaptive policy)."""
	from puzzlelib_amd import nets, optim, lazy

	net, spec, gpuarray = build_mini(mini_golden, actInplace=True)
	data, labels = mini_golden["data"], mini_golden["labels"]

	params = {k[5:]: mini_golden[k] for k in mini_golden.keys() if k.startswith("init_")}
	pshapes, ashapes = nets.spec_param_shapes(spec)
	attrs = {k: (np.zeros(s, np.float32) if k.endswith(".mean") else np.ones(s, np.float32)) for k, s in ashapes.items()}
	cnet = N.CpuNet(spec, params, attrs)
	copt = N.CpuAdam(cnet, alpha=1e-3)

	optimizer = optim.Adam(alpha=1e-3)
	optimizer.setupOn(net, useGlobalState=True)
	cost = optim.CrossEntropy()
	trainer = optim.Trainer(net, cost, optimizer, batchsize=4)

	gdata, glabels = gpuarray.to_gpu(data), gpuarray.to_gpu(labels)
	lazy.counters.clear()
	for step in range(3):
		_, err = N.train_step(cnet, copt, data, labels)
		trainer.train(gdata, glabels, random=False)
		assert np.isclose(cost.getMeanError() * 4, err, rtol=2e-3), "step %d: device %s vs oracle %s" % (
			step, cost.getMeanError() * 4, err
		)
	assert lazy.counters.get("conv_stats", 0) > 0


def oneStep(spec, init, data, labels, seed=7):
	"""logits, loss, gradients and parameters after one Adam step, plus the fusion counters of that step"""
	from puzzlelib_amd import nets, optim, lazy
	from puzzlelib_amd.surface import bound
	gpuarray = bound().gpuarray

	np.random.seed(seed)
	net = nets.build(spec, name="mini", initscheme="he", actInplace=True)
	if init is not None:
		for name, param in net.namedParams().items():
			param.data.set(init[name])
	init = {name: p.data.get() for name, p in net.namedParams().items()}

	optimizer = optim.Adam(alpha=1e-3)
	optimizer.setupOn(net, useGlobalState=True)
	cost = optim.CrossEntropy()
	net.trainMode()

	lazy.counters.clear()
	pred = net(gpuarray.to_gpu(data))
	logits = pred.get()
	grad = cost(pred, gpuarray.to_gpu(labels), queryError=False)
	optimizer.zeroGradParams()
	net.backward(grad, updGrad=False)
	grads = {name: p.grad.get() for name, p in net.namedParams().items()}
	optimizer.update()
	params = {name: p.data.get() for name, p in net.namedParams().items()}
	return dict(logits=logits, err=float(cost.devErr.get()), grads=grads, params=params, init=init, taken=dict(lazy.counters))


def test_lazy_fusion_is_bit_identical_to_the_literal_sequence_and_matches_oracle(bnd, mini_golden):
	"""One mini-ResNet Adam step (actInplace=True, Models/Nets/ResNet.py:33,58) under the lazy layer, with the layer off
	(every reference call launches its own kernels), and with each fusion pattern switched off alone. Every variant must
	give the same bits (the BN-backward fold, whose rounding differs, is checked separately below; epilogue statistics
	are pinned off for the same reason) and the oracle's loss / parameters within the usual tolerances."""
	from puzzlelib_amd import lazy, backend

	spec = miniSpec()
	data, labels = mini_golden["data"], mini_golden["labels"]
	init = {k[5:]: mini_golden[k] for k in mini_golden.keys() if k.startswith("init_")}
	backend.DnnContext.convStatsPolicy = "never"

	results = {}
	variants = [(), ("bnadd", ), ("gatestats", ), ("up2", ), ("mask", ), ("sidestream", ), ("bnrelu", ), ("bnrelubwd", ), ("gate", ),
				("addrelu", ), ("addgate", ), ("sum", ), ("bnapply", ), ("bnpool", ), "literal"]
	for variant in variants:
		if variant == "literal":
			lazy.enabled = False
		else:
			lazy.disabled = {"bnbwdfold"} | set(variant)
		results[variant] = oneStep(spec, init, data, labels)
		lazy.enabled, lazy.disabled = True, set()

	full = results[()]
	taken = full["taken"]
	for key, n in (("bn_apply_add", 3), ("bn_apply_relu", 6), ("bn_pool", 1), ("bn_bwd_gate", 7), ("gate_stats", 1), ("gate_stats_up2", 1),
				   ("compact_dgrad", 2), ("gate_by_mask", 2), ("bn_bwd_from_partials", 4)):
		assert taken.get(key, 0) == n, "%s taken %d times, expected %d: %s" % (key, taken.get(key, 0), n, taken)
	assert set(results["literal"]["taken"]) <= {"bn_apply"}, "with the layer off nothing is deferred"
	assert results[("bnadd", )]["taken"].get("bn_apply_add", 0) == 0 and results[("up2", )]["taken"].get("compact_dgrad", 0) == 0
	assert results[("mask", )]["taken"].get("gate_by_mask", 0) == 0
	assert results[("bnpool", )]["taken"].get("bn_pool", 0) == 0 and results[("bnpool", )]["taken"].get("bn_apply_relu", 0) == 7

	for variant in variants[1:]:
		other = results[variant]
		assert np.array_equal(other["logits"], full["logits"]) and other["err"] == full["err"], variant
		for name in full["grads"]:
			assert np.array_equal(other["grads"][name], full["grads"][name]), "grad %s (%s)" % (name, variant)
			assert np.array_equal(other["params"][name], full["params"][name]), "param %s (%s)" % (name, variant)

	assert_close(full["logits"], mini_golden["orc_logits"], atol=2e-4, rtol=1e-3, what="logits")
	assert np.isclose(full["err"], mini_golden["orc_err"][0], rtol=1e-4)
	for name in full["params"]:
		assert_close(full["params"][name], mini_golden["orc_after_" + name], atol=3e-4, rtol=1e-4, what="param " + name)


@pytest.mark.parametrize("planes", [16, 6])
def test_batchnorm_backward_folded_into_the_convolution(bnd, planes):
	"""Conv2D -> BatchNorm2D backward with the BN's input gradient only described and evaluated inside the 1x1
	convolution's backward-data / backward-filter gathers (pz_conv2d_bwd_*_bn): gradients equal the unfolded path to fp32
	rounding and the oracle to the usual tolerances. planes 6: the 24-map layers are not a multiple of 16 -> their
	convolution declines and the description is written out by pz_bn_bwd_apply_coef (same numbers)."""
	from puzzlelib_amd import lazy, nets, backend
	backend.DnnContext.convStatsPolicy = "never"      # (adaptive: keyed by filter addresses, which the second build may reuse)

	spec = miniSpec(planes, stem=16)
	rng = np.random.RandomState(3)
	data = rng.randn(4, 3, 64, 64).astype(np.float32)
	labels = rng.randint(0, 10, size=(4, )).astype(np.int32)

	lazy.disabled = {"bnbwdfold"}
	plain = oneStep(spec, None, data, labels)
	lazy.disabled = set()
	folded = oneStep(spec, plain["init"], data, labels)

	assert plain["taken"].get("dgrad_bn_fold", 0) == 0
	if planes == 16:
		assert folded["taken"].get("dgrad_bn_fold", 0) == 4 and folded["taken"].get("wgrad_bn_fold", 0) == 4
	else:
		# stage 2's 48 maps are a multiple of 16 (folded), stage 1's 24 are not (written out)
		assert folded["taken"].get("dgrad_bn_fold", 0) == 2 and folded["taken"].get("bn_bwd_apply", 0) == 2
	for name in plain["grads"]:
		scale = np.abs(plain["grads"][name]).max() + 1e-12
		assert_close(folded["grads"][name], plain["grads"][name], atol=3e-5 * scale, rtol=3e-4, what="grad " + name)

	_, ashapes = nets.spec_param_shapes(spec)
	attrs = {k: (np.zeros(s, np.float32) if k.endswith(".mean") else np.ones(s, np.float32)) for k, s in ashapes.items()}
	cnet = N.CpuNet(spec, plain["init"], attrs)
	cnet.train = True
	pred_ref = cnet.forward(data)
	_, grad_ref = R.cross_entropy(pred_ref, labels)
	cnet.zero_grads()
	cnet.backward(grad_ref)
	for name in folded["grads"]:
		ref = cnet.grads[name]
		scale = np.abs(ref).max() + 1e-6
		assert_close(folded["grads"][name], ref, atol=2e-3 * scale, rtol=2e-3, what="grad vs oracle " + name)


def test_skipping_the_first_layers_input_gradient_changes_nothing_else(bnd, mini_golden):
	"""engine.Net.skipInputGrad (the harness's one deviation from the reference's call sequence: updGrad=False honoured)
	leaves every parameter gradient bit-identical."""
	from puzzlelib_amd import engine, backend
	backend.DnnContext.convStatsPolicy = "never"      # (adaptive: keyed by filter addresses, which the second build may reuse)
	spec = miniSpec()
	data, labels = mini_golden["data"], mini_golden["labels"]
	a = oneStep(spec, None, data, labels)
	engine.Net.skipInputGrad = True
	b = oneStep(spec, a["init"], data, labels)
	for name in a["grads"]:
		assert np.array_equal(a["grads"][name], b["grads"][name]), name


def test_batchnorm_first_network_trains(bnd):
	"""a network whose first layer is a BatchNorm (its parameter gradients come out of the same backward call
	as the input gradient) must train through the batch loop; checked against the oracle."""
	from puzzlelib_amd import nets, optim
	from puzzlelib_amd.surface import bound
	gpuarray = bound().gpuarray
	spec = [("bn", "bn0", 3), ("relu", "r0"), ("conv", "c1", 3, 8, 3, 1, 1, True), ("relu", "r1"), ("avgpool", "p", 8, 1, 0),
			("flatten", "f"), ("linear", "fc", 8, 5)]
	rng = np.random.RandomState(1)
	data, labels = rng.randn(6, 3, 8, 8).astype(np.float32), rng.randint(0, 5, size=(6, )).astype(np.int32)

	for inplace in (False, True):
		np.random.seed(2)
		net = nets.build(spec, initscheme="he", actInplace=inplace)
		params = {k: p.data.get() for k, p in net.namedParams().items()}
		_, ashapes = nets.spec_param_shapes(spec)
		attrs = {k: (np.zeros(s, np.float32) if k.endswith(".mean") else np.ones(s, np.float32)) for k, s in ashapes.items()}
		cnet = N.CpuNet(spec, params, attrs)
		copt = N.CpuMomentumSGD(cnet, learnRate=0.05, momRate=0.9)

		optimizer = optim.MomentumSGD(learnRate=0.05, momRate=0.9)
		optimizer.setupOn(net, useGlobalState=True)
		cost = optim.CrossEntropy()
		trainer = optim.Trainer(net, cost, optimizer, batchsize=6)
		for step in range(2):
			_, err = N.train_step(cnet, copt, data, labels)
			trainer.train(gpuarray.to_gpu(data), gpuarray.to_gpu(labels), random=False)
			assert np.isclose(cost.getMeanError() * 6, err, rtol=1e-3)
		for name, p in net.namedParams().items():
			assert_close(p.data.get(), cnet.params[name], atol=1e-4, rtol=1e-3, what=name)


def test_validator_and_eval_mode(bnd, mini_golden):
	from puzzlelib_amd import optim, nets

	net, spec, gpuarray = build_mini(mini_golden)
	data, labels = mini_golden["data"], mini_golden["labels"]

	validator = optim.Validator(net, optim.CrossEntropy(), batchsize=2)
	err = validator.validate(gpuarray.to_gpu(data), gpuarray.to_gpu(labels))

	params = {k[5:]: mini_golden[k] for k in mini_golden.keys() if k.startswith("init_")}
	_, ashapes = nets.spec_param_shapes(spec)
	attrs = {k: (np.zeros(s, np.float32) if k.endswith(".mean") else np.ones(s, np.float32)) for k, s in ashapes.items()}
	cnet = N.CpuNet(spec, params, attrs)
	cnet.train = False
	pred = cnet.forward(data)
	assert err == np.mean(np.argmax(pred, axis=1) != labels)


def adoptDeviceGates(cnet, net, spec):
	"""Makes the oracle's backward gate with what the DEVICE gated with: ReLU outputs and max-pool inputs / outputs read back
	from the executor's layers replace the oracle's own in its cache (flat specs: cache key = spec index). Nine ReLU layers
	deep a pre-activation within rounding of zero can have different signs on the two sides; that single flip moves a whole
	filter gradient by O(1e-3) relative although both sides are right to rounding (profiles/r03_nin_seed_sweep.txt: 100
	seeds x {one / two streams} x {lazy on / off} — all device runs bit-identical, error with the device's gates <= 7.1e-7,
	with the oracle's own gates up to 5.3e-3 on the 13 seeds that had a flip). Returns the number of flipped ReLU gates; the
	forward tensors themselves are held to the oracle's separately by the caller."""
	flips = 0
	for idx, layer in enumerate(net.layers):
		key = str(idx)
		if layer.kind == "act":
			y = layer.y.get()
			assert np.abs(y - cnet.cache[key]).max() <= 1e-4 * max(1.0, np.abs(y).max()), "activation %s" % layer.name
			flips += int(((y > 0) != (cnet.cache[key] > 0)).sum())
			cnet.cache[key] = y
		elif layer.kind == "pool" and spec[idx][0] == "maxpool":
			cnet.cache[key] = (layer.x.get(), layer.y.get())
	return flips


@pytest.mark.parametrize("batch", [8, 128])
def test_nin_step_matches_oracle(bnd, batch):
	"""Config 3 (CIFAR-10 NiN, TestLib/CnnCifar10NIN.py): one full training step — forward, cross-entropy, backward,
	WeightDecay hook, MomentumSGD update — at a reduced batch and at the configuration's own batch of 128. Dropout words
	come from a SEEDED device generator and are fed to the oracle (Philox here, XORWOW there: statistical parity only);
	the oracle's backward gates with the device's ReLU / max-pool decisions (adoptDeviceGates). Tolerance: every parameter
	gradient within 5e-5 relative L2 of the oracle's (measured <= 7.1e-7 at batch 8 over 100 seeds)."""
	from puzzlelib_amd import nets, optim, backend
	from puzzlelib_amd.surface import bound

	gpuarray = bound().gpuarray
	np.random.seed(1234)
	net = nets.buildNiN()
	spec = nets.nin_spec()

	rng = np.random.RandomState(5)
	data = rng.randn(batch, 3, 32, 32).astype(np.float32)
	labels = rng.randint(0, 10, size=(batch, )).astype(np.int32)

	params = {name: p.data.get() for name, p in net.namedParams().items()}
	cnet = N.CpuNet(spec, params)

	devrng = backend.RandomNumberGenerator(seed=20260929)
	drops = [layer for layer in net.walk() if layer.kind == "dropout"]
	for layer in drops:
		layer.cfg["rng"] = devrng

	optimizer = optim.MomentumSGD(learnRate=0.1, momRate=0.9)
	optimizer.setupOn(net, useGlobalState=True)
	optimizer.addHook(optim.WeightDecay(1e-4))
	for target, _ in optimizer.targets:
		target.wc = 1.0
	cost = optim.CrossEntropy()

	# the device step runs undisturbed (nothing is read back before the backward pass has been issued)
	net.trainMode()
	pred = net(gpuarray.to_gpu(data))
	grad = cost(pred, gpuarray.to_gpu(labels), queryError=False)
	optimizer.zeroGradParams()
	net.backward(grad, updGrad=False)

	for layer in drops:
		cnet.dropmasks[layer.name] = layer.aux[0].get()
	cnet.train = True
	pred_ref = cnet.forward(data)
	assert_close(pred.get(), pred_ref, atol=1e-4, rtol=1e-3, what="NiN forward")
	err_ref, grad_ref = R.cross_entropy(pred_ref, labels)
	assert np.isclose(float(cost.devErr.get()), err_ref, rtol=1e-4)
	assert_close(grad.get(), grad_ref, atol=1e-6, rtol=1e-4, what="NiN cross-entropy gradient")

	adoptDeviceGates(cnet, net, spec)
	cnet.zero_grads()
	cnet.backward(grad_ref)

	for name, p in net.namedParams().items():
		ref = cnet.grads[name]
		got = p.grad.get()
		rel = np.linalg.norm((got - ref).astype(np.float64)) / (np.linalg.norm(ref.astype(np.float64)) + 1e-30)
		assert rel < 5e-5, "NiN b%d grad %s: relative L2 error %.3e" % (batch, name, rel)
		assert_close(got, ref, atol=1e-4 * (np.abs(ref).max() + 1e-8), rtol=1e-3, what="NiN grad " + name)

	# the update of the real step: WeightDecay(1e-4) hook + MomentumSGD(0.1, 0.9) (TestLib/CnnCifar10NIN.py:63-66)
	optimizer.update()
	copt = N.CpuMomentumSGD(cnet, learnRate=0.1, momRate=0.9, weightDecay=1e-4)
	copt.update()
	for name, p in net.namedParams().items():
		ref = cnet.params[name]
		assert_close(p.data.get(), ref, atol=2e-5 * (np.abs(ref).max() + 1e-8), rtol=1e-4, what="NiN param after the step " + name)


@pytest.mark.parametrize("form", ["unique", "full"])
def test_network_loads_what_the_reference_saved(bnd, form):
	"""Modules/Module.py:233-283 load of a file Module.save wrote (:179-231; both naming forms, Models/Nets/ResNet.py:118-119 uses
	assumeUniqueNames=True): tests/golden/refckpt_mini_<form>.npz is the reference's OWN output for a small ResNet built from
	its residBlock (oracle/make_checkpoint_fixture.py; .npz mirror of the HDF5 tree), refckpt_mini_io.npz the reference's
	evaluation-mode scores on a fixed input (numpy CPU backend). Every parameter and running statistic has a distinct value:
	the device network reproduces the scores only if each tensor lands in its place."""
	from conftest import GOLDEN, Golden
	from puzzlelib_amd import nets, checkpoint
	from puzzlelib_amd.surface import bound
	gpuarray = bound().gpuarray
	io = Golden("refckpt_mini_io.npz")
	spec = nets.resnet_spec(stages=((4, 2), (8, 1)), classes=10, stem=8, softmax=False)
	spec = [l if l[0] != "avgpool" else ("avgpool", l[1], 4, 1, 0) for l in spec]
	net = nets.build(spec, name="ResNet-mini", initscheme="gaussian")
	meta = checkpoint.load(net, os.path.join(GOLDEN, "refckpt_mini_%s.npz" % form))
	assert meta == {}
	net.evalMode()
	scores = net(gpuarray.to_gpu(io["data"])).get()
	assert_close(scores, io["scores"], atol=2e-5 * np.abs(io["scores"]).max(), rtol=1e-4, what="scores after loading the reference's file")

	# ... and against the oracle network fed the file's tensors directly
	tensors = checkpoint.read(os.path.join(GOLDEN, "refckpt_mini_%s.npz" % form))
	link, attrOf = checkpoint.resolver(tensors, "links", form), checkpoint.resolver(tensors, "attrs", form)
	params = {name: tensors["params/%d" % int(tensors[link(name)])] for name in net.namedParams()}
	attrs = {name: tensors[attrOf(name)] for name in net.namedAttrs()}
	cnet = N.CpuNet(spec, params, attrs)
	cnet.train = False
	assert_close(cnet.forward(io["data"]), io["scores"], atol=2e-5 * np.abs(io["scores"]).max(), rtol=1e-4, what="oracle network on the file's tensors")


@pytest.mark.parametrize("layers", ["50", "101", "152"])
def test_resnet_variants_run(bnd, layers):
	"""Models/Nets/ResNet.py:124-143 (the reference's unittest): all three depths build and push one image through."""
	from puzzlelib_amd import nets
	from puzzlelib_amd.surface import bound
	gpuarray = bound().gpuarray
	np.random.seed(5)
	net = nets.loadResNet(None, layers=layers, initscheme="he")        # (the reference's "gaussian" draws overflow 100+ layers deep)
	assert net.name == "ResNet-%s" % layers
	net.evalMode()
	out = net(gpuarray.to_gpu(np.random.randn(1, 3, 224, 224).astype(np.float32))).get()
	assert out.shape == (1, 1000) and np.isfinite(out).all() and abs(float(out.sum()) - 1.0) < 1e-4
	del net
	gpuarray.memoryPool.freeHeld()


def test_checkpoint_round_trip_continues_bit_for_bit(bnd, mini_golden, tmp_path):
	"""Modules/Module.py:179-283 save / load (params + links + attrs) in the harness's container, plus optimizer state: train
	2 steps, save, restore into a freshly built network + optimizer, train 2 more — equal to 4 uninterrupted steps bit for
	bit (parameters, running statistics, Adam moments, batch-norm momentum schedule)."""
	from puzzlelib_amd import nets, optim, checkpoint, backend
	from puzzlelib_amd.surface import bound
	gpuarray = bound().gpuarray
	spec = miniSpec()
	data, labels = gpuarray.to_gpu(mini_golden["data"]), gpuarray.to_gpu(mini_golden["labels"])
	# (the adaptive policy switches a convolution's epilogue statistics on after its first pass — run-time state that is
	# not part of a checkpoint and moves results by summation order only; pinned so that both runs round identically)
	backend.DnnContext.convStatsPolicy = "always"

	def fresh(seed):
		np.random.seed(seed)
		net = nets.build(spec, name="mini", initscheme="he", actInplace=True)
		opt = optim.Adam(alpha=1e-3)
		opt.setupOn(net, useGlobalState=True)
		return net, opt, optim.Trainer(net, optim.CrossEntropy(), opt, batchsize=4)

	net, opt, trainer = fresh(7)
	for _ in range(2):
		trainer.train(data, labels, random=False)
	path = str(tmp_path / "mini_checkpoint")                             # no extension: the container is found by its magic
	checkpoint.save(net, path, optimizer=opt)
	for _ in range(2):
		trainer.train(data, labels, random=False)
	straight = {k: p.data.get() for k, p in net.namedParams().items()}
	straight.update({k: a.get() for k, a in net.namedAttrs().items()})

	tensors = checkpoint.read(path)
	assert "links/conv1.W" in tensors and "attrs/bn_conv1.mean" in tensors and "optimizer/0/mg" in tensors
	assert tensors["params/%d" % int(tensors["links/conv1.W"])].shape == (8, 3, 7, 7)

	net2, opt2, trainer2 = fresh(99)                                     # different initial values: everything must come from the file
	meta = checkpoint.load(net2, path, optimizer=opt2)
	assert meta["optimizer"]["t"] == 2 and opt2.t == 2
	for _ in range(2):
		trainer2.train(data, labels, random=False)
	resumed = {k: p.data.get() for k, p in net2.namedParams().items()}
	resumed.update({k: a.get() for k, a in net2.namedAttrs().items()})
	for key in straight:
		assert np.array_equal(straight[key], resumed[key]), key

	with pytest.raises(ValueError):
		checkpoint.load(net2, path, optimizer=optim.MomentumSGD())


def test_descriptions_that_outlive_an_update_copy_only_what_they_read_of_a_large_arena(bnd):
	"""a described pre-activation convolution (Conv2D(bias) -> Activation(relu) out of place) that has to survive the
	optimizer's write to the arena used to snapshot the WHOLE arena. Above lazy.snapshotWhole it now copies only its own filter
	and bias bytes. Three NiN steps with the threshold at 0 (every description takes private pieces) must leave exactly the
	parameters of three steps with one shared whole-arena copy per step — and read the same stale pre-activation values."""
	from puzzlelib_amd import nets, optim, lazy, backend
	from puzzlelib_amd.surface import bound
	gpuarray = bound().gpuarray
	rng = np.random.RandomState(6)
	data = gpuarray.to_gpu(rng.randn(8, 3, 32, 32).astype(np.float32))
	labels = gpuarray.to_gpu(rng.randint(0, 10, size=(8, )).astype(np.int32))

	def run(threshold):
		before, lazy.snapshotWhole = lazy.snapshotWhole, threshold
		try:
			np.random.seed(1234)
			net = nets.buildNiN()
			devrng = backend.RandomNumberGenerator(seed=20260930)      # (the two runs must draw identical dropout words)
			for layer in net.walk():
				if layer.kind == "dropout":
					layer.cfg["rng"] = devrng
			optimizer = optim.MomentumSGD(learnRate=0.05, momRate=0.9)
			optimizer.setupOn(net, useGlobalState=True)
			trainer = optim.Trainer(net, optim.CrossEntropy(), optimizer, batchsize=8)
			net.trainMode()
			lazy.counters.clear()
			kept = None
			for step in range(3):
				trainer.step([data, labels])
				if step == 1:
					first = next(l for l in net.walk() if l.kind == "conv")
					kept = first.y                            # the convolution's own (pre-activation) output: only described so far
				net.reset()
			stale = None if kept is None else kept.get()      # read AFTER two more updates of the parameters it was computed from
			return {k: p.data.get() for k, p in net.namedParams().items()}, stale, dict(lazy.counters)
		finally:
			lazy.snapshotWhole = before

	whole, stale_w, cw = run(1 << 40)
	pieces, stale_p, cp = run(0)
	assert cw.get("param_snapshot", 0) > 0 and cw.get("param_snapshot_piece", 0) == 0, cw
	assert cp.get("param_snapshot_piece", 0) > 0 and cp.get("param_snapshot", 0) == 0, cp
	for name in whole:
		assert np.isfinite(whole[name]).all() and np.array_equal(whole[name], pieces[name]), name
	assert stale_w is not None and np.isfinite(stale_w).all() and np.array_equal(stale_w, stale_p)
