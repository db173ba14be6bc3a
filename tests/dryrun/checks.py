"""
Host-logic checks that need the backend's Python side to *run* without a device: executed in a child process with
PUZZLE_MI355_DRYRUN=1 (the library then records C-ABI calls instead of executing them — puzzlelib_amd/lib.py) by
tests/test_dryrun_backend.py. Usage: python tests/dryrun/checks.py <check name>
"""
import json, os, sys

assert os.environ.get("PUZZLE_MI355_DRYRUN") == "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np

from puzzlelib_amd import nets, optim, lib, lazy, fusion, grid
from puzzlelib_amd.surface import bound

SKIP = ("pz_pool_", "pz_event_", "pz_stream_", "pz_malloc", "pz_free", "pz_device_", "pz_init")


def names(trace=None):
	return [n for n, _ in (lib.trace if trace is None else trace) if not n.startswith(SKIP)]


def trainerFor(net, batch):
	opt = optim.Adam(alpha=1e-3)
	opt.setupOn(net, useGlobalState=True)
	return optim.Trainer(net, optim.CrossEntropy(), opt, batchsize=batch), opt


# ------------------------------------------------------------------------------------------------ drop-in call traces
def executor_trace(case):
	g = bound().gpuarray
	if case == "resnet50_b8":
		net = nets.loadResNet(None, "50", actInplace=True, initscheme="none")
		net.layers.pop()                                   # the trailing SoftMax (training on raw scores)
		shape = (8, 3, 224, 224)
	elif case == "nin_b8":
		np.random.seed(1234)
		net, shape = nets.buildNiN(), (8, 3, 32, 32)
	else:
		net, shape = nets.loadLeNet(None), (16, 1, 28, 28)
	nsteps = 2
	if case == "lenet_dp_b16":                             # the reference's MomentumSGD(nodeinfo=...) on a one-rank grid, its own arena order
		optim.Optimizer.arenaLayout = "sorted"
		opt = optim.MomentumSGD(learnRate=0.1, momRate=0.9, nodeinfo=singleRankNode())
		opt.setupOn(net, useGlobalState=True)
		trainer = optim.Trainer(net, optim.CrossEntropy(), opt, batchsize=shape[0])
		nsteps = 5
	elif case == "nin_b8":                                   # TestLib/CnnCifar10NIN.py:68-72
		opt = optim.MomentumSGD(learnRate=0.1, momRate=0.9)
		opt.addHook(optim.WeightDecay(0.0001))
		opt.setupOn(net, useGlobalState=True)
		trainer = optim.Trainer(net, optim.CrossEntropy(maxlabels=10), opt, batchsize=shape[0])
	else:
		trainer, _ = trainerFor(net, shape[0])
	data, labels = g.to_gpu(np.zeros(shape, np.float32)), g.to_gpu(np.zeros(shape[:1], np.int32))
	steps = []
	for _ in range(nsteps):
		lib.trace.clear()
		trainer.train(data, labels, random=False)
		steps.append([[n, list(a)] for n, a in lib.trace if not n.startswith(SKIP)])
	return json.loads(json.dumps(steps))


def check_trace(case):
	"""the executor sends the backend what the reference's own modules send (fixture recorded by oracle/make_trace.py)"""
	ref = json.load(open(os.path.join(ROOT, "tests", "golden", "trace_%s.json" % case)))["steps"]
	ours = executor_trace(case)
	for step, (a, b) in enumerate(zip(ours, ref)):
		assert len(a) == len(b), "%s step %d: %d calls, the reference's modules make %d" % (case, step, len(a), len(b))
		for i, (x, y) in enumerate(zip(a, b)):
			assert x == y, "%s step %d call %d: executor %s, reference modules %s" % (case, step, i, x, y)
	print("trace %s: identical (%d + %d calls)" % (case, len(ref[0]), len(ref[1])))


def trace_resnet50():
	check_trace("resnet50_b8")


def trace_lenet():
	check_trace("lenet_b16")


def trace_lenet_dp():
	check_trace("lenet_dp_b16")


def trace_nin():
	check_trace("nin_b8")


# ------------------------------------------------------------------------------------------------ lazy-buffer semantics
def lazy_barriers():
	surf = bound()
	g, bnd, dnn = surf.gpuarray, surf.backend, surf.backend.dnn
	f32 = np.float32

	def bnInputs(c=8):
		return (g.to_gpu(np.ones((1, c, 1, 1), f32)), g.zeros((1, c, 1, 1), f32), g.zeros((1, c, 1, 1), f32),
				g.to_gpu(np.ones((1, c, 1, 1), f32)))

	x = g.to_gpu(np.zeros((4, 8, 16, 16), f32))
	scale, bias, mean, var = bnInputs()

	# (1) batchNormNd describes its output; nothing but the coefficient launch happens until somebody reads it
	lib.trace.clear()
	y, sm, si = surf.Dnn.batchNormNd(x, scale, bias, mean, var, 1e-5, 1.0, False)
	assert names() == ["pz_bn_fwd_train_coef"] and isinstance(lazy.pending(y), fusion.BnApply)
	surf.ElementWise.reluKer(f32)(y, y)                                   # in place: joins the description
	assert names() == ["pz_bn_fwd_train_coef"] and lazy.pending(y).relu
	y.get()                                                               # a read writes it, fused
	assert names()[1:] == ["pz_bn_apply_add", "pz_memcpy_d2h"] and lazy.pending(y) is None
	assert lazy.fact(y, "bnapply") is not None

	# (2) a write to the input of a pending description settles the description first (it needs the old values) ...
	lib.trace.clear()
	y2, _, _ = surf.Dnn.batchNormNd(x, scale, bias, mean, var, 1e-5, 1.0, False)
	x.fill(0)
	assert names() == ["pz_bn_fwd_train_coef", "pz_bn_apply_add", "pz_memset_d32"] or \
		names()[:2] == ["pz_bn_fwd_train_coef", "pz_bn_apply_add"], names()
	assert lazy.pending(y2) is None
	# ... and drops facts derived from the old contents (y's "this is relu(bn(x))")
	assert lazy.fact(y, "bnapply") is None and lazy.fact(y2, "bnapply") is None

	# (3) overwriting a described tensor entirely drops the description without running it
	lib.trace.clear()
	y3, _, _ = surf.Dnn.batchNormNd(x, scale, bias, mean, var, 1e-5, 1.0, False)
	y3.set(np.zeros(y3.shape, f32))
	assert "pz_bn_apply_add" not in names() and lazy.pending(y3) is None

	# (4) a partial write (a view) runs it first
	lib.trace.clear()
	y4, _, _ = surf.Dnn.batchNormNd(x, scale, bias, mean, var, 1e-5, 1.0, False)
	y4[0:1].set(np.zeros((1, 8, 16, 16), f32))
	assert names()[:2] == ["pz_bn_fwd_train_coef", "pz_bn_apply_add"]

	# (5) the residual sum: zeros + axpy + axpy + relu is one kernel, whoever reads it; a third reader sees a plain tensor
	a, _, _ = surf.Dnn.batchNormNd(x, scale, bias, mean, var, 1e-5, 1.0, False)
	b = g.to_gpu(np.zeros(x.shape, f32))
	lib.trace.clear()
	total = g.empty(x.shape, dtype=f32)
	total.fill(0)
	surf.Blas.toVectorAddVector(total.ravel(), a.ravel())
	surf.Blas.toVectorAddVector(total.ravel(), b.ravel())
	surf.ElementWise.reluKer(f32)(total, total)
	assert names() == [] and isinstance(lazy.pending(total), fusion.Sum)
	total.get()
	assert names() == ["pz_bn_apply_add_mask", "pz_memcpy_d2h"], names()
	assert lazy.pending(a) is not None, "the BatchNorm output itself was never written"
	assert lazy.fact(total, "relumask") is not None and len(lazy.fact(total, "bnterms")) == 1

	# (6) alpha != 1 or an accumulator somebody already read: the literal kernels
	lib.trace.clear()
	acc = g.empty(x.shape, dtype=f32)
	acc.fill(0)
	surf.Blas.toVectorAddVector(acc.ravel(), b.ravel(), alpha=0.5)
	assert names() == ["pz_memset_d32", "pz_eltwise"], names()

	# (7) with the layer switched off every call launches on the spot
	lazy.enabled = False
	lib.trace.clear()
	y7, _, _ = surf.Dnn.batchNormNd(x, scale, bias, mean, var, 1e-5, 1.0, False)
	surf.ElementWise.reluKer(f32)(y7, y7)
	assert names() == ["pz_bn_fwd_train_coef", "pz_bn_apply_add", "pz_eltwise"], names()
	lazy.enabled = True

	# (8) a kernel on a borrowed stream leaves its event on what it wrote; the main stream waits when it touches it
	stream = g.streamManager.borrow(1)[0]
	p, q = g.to_gpu(np.zeros(64, f32)), g.to_gpu(np.zeros(64, f32))
	lib.trace.clear()
	surf.ElementWise.toVectorAddVectorKer(f32)(p, q, 1.0, stream=stream)
	assert p.gpudata.root.lz.wev is not None and q.gpudata.root.lz.rev is not None
	before = len([n for n, _ in lib.trace if n == "pz_stream_wait_event"])
	p.get()
	after = len([n for n, _ in lib.trace if n == "pz_stream_wait_event"])
	assert after == before + 1 and p.gpudata.root.lz.wev is None

	# (9) ... and only for the bytes it wrote: a neighbour block of the same allocation (the optimizer's arena holds every
	# gradient) is touched by the main stream without waiting
	arena = g.to_gpu(np.zeros(256, f32))
	left, right = arena[:128], arena[128:]
	surf.ElementWise.toVectorAddVectorKer(f32)(left, g.to_gpu(np.zeros(128, f32)), 1.0, stream=stream)
	waits = lambda: len([n for n, _ in lib.trace if n == "pz_stream_wait_event"])
	before = waits()
	surf.ElementWise.toVectorAddVectorKer(f32)(right, g.to_gpu(np.zeros(128, f32)), 1.0)
	assert waits() == before, "a write next to the foreign stream's bytes must not wait for it"
	surf.ElementWise.toVectorAddVectorKer(f32)(arena, g.to_gpu(np.zeros(256, f32)), 1.0)
	assert waits() == before + 1 and arena.gpudata.root.lz.wev is None

	# (10) ADVICE r2: overwriting a described tensor entirely must first serve whoever captured it by reference — B is a
	# sum over A while A is still a pending zero fill; A.set(...) may drop A's description only after B has been written
	lib.trace.clear()
	A = g.empty(x.shape, dtype=f32)
	A.fill(0)
	B = g.empty(x.shape, dtype=f32)
	B.fill(0)
	surf.Blas.toVectorAddVector(B.ravel(), A.ravel())
	assert names() == [] and isinstance(lazy.pending(B), fusion.Sum)
	A.set(np.ones(A.shape, f32))
	got = names()
	# (two zero fills — A's own and the literal 0 + A of B — and B's axpy, all before the upload lands in A)
	assert lazy.pending(B) is None and got.count("pz_memset_d32") == 2 and got[-1] == "pz_memcpy_h2d", got

	# (11) ADVICE r2: an in-place ReLU joining a description is a write — an accumulator that took the described tensor
	# by reference (3-d BatchNorm output: the "arr" term of a Sum) must get bn(x), not relu(bn(x))
	x3 = g.to_gpu(np.zeros((4, 8, 64), f32))
	lib.trace.clear()
	y11, _, _ = surf.Dnn.batchNormNd(x3, scale, bias, mean, var, 1e-5, 1.0, False)
	acc = g.empty(x3.shape, dtype=f32)
	acc.fill(0)
	surf.Blas.toVectorAddVector(acc.ravel(), y11.ravel())
	assert isinstance(lazy.pending(acc), fusion.Sum) and lazy.pending(acc).terms[0][0] == "arr"
	surf.ElementWise.reluKer(f32)(y11, y11)
	assert lazy.pending(acc) is None, "the accumulator was settled before the ReLU changed the tensor it refers to"
	relus = [a for n, a in lib.trace if n == "pz_bn_apply_add"]
	assert relus and all(int(a[-2]) == 0 for a in relus), "bn(x) was written without the ReLU"
	assert "pz_eltwise" in names(), "the ReLU itself then ran as a kernel on the written tensor"

	# (12) the same for a further term joining a described sum that somebody refers to
	lib.trace.clear()
	s1 = g.empty(x.shape, dtype=f32)
	s1.fill(0)
	surf.Blas.toVectorAddVector(s1.ravel(), b.ravel())
	s2 = g.empty(x.shape, dtype=f32)
	s2.fill(0)
	surf.Blas.toVectorAddVector(s2.ravel(), s1.ravel())            # s2 = 0 + s1 (s1 by reference)
	surf.Blas.toVectorAddVector(s1.ravel(), b.ravel())             # s1 changes: s2 is written first
	assert lazy.pending(s2) is None

	# (13) ADVICE r2 (high): a convolution's per-channel strip sums are the statistics of a BatchNorm over that very
	# tensor only — not of a BatchNorm over a reshape or a slice of it (second pass: the adaptive policy is armed)
	from puzzlelib_amd import backend as B_
	w = g.to_gpu(np.zeros((16, 8, 3, 3), f32))
	def statsArg():
		call = [a for n, a in lib.trace if n == "pz_bn_fwd_train_coef"][-1]
		return call[12], call[13]                 # (statistics given: 1 / 0, strips)
	for attempt in range(3):
		lib.trace.clear()
		yc = dnn.convNd(x, w, None, stride=1, pad=1)
		yb, _, _ = surf.Dnn.batchNormNd(yc, *bnInputs(16), 1e-5, 1.0, False)
		if attempt > 0:
			assert statsArg()[0] == 1 and statsArg()[1] > 0, "conv -> BN takes the strip sums from the second pass on"
		yc = dnn.convNd(x, w, None, stride=1, pad=1)
		view = yc.reshape(4, 32, 8, 16)
		surf.Dnn.batchNormNd(view, *bnInputs(32), 1e-5, 1.0, False)
		assert statsArg() == (0, 0), "BN over a reshape of the convolution's output computes its own statistics"
		yc = dnn.convNd(x, w, None, stride=1, pad=1)
		surf.Dnn.batchNormNd(yc[:2], *bnInputs(16), 1e-5, 1.0, False)
		assert statsArg() == (0, 0), "BN over a slice of the convolution's output computes its own statistics"
	print("lazy barriers: OK")


def fused_step_counts():
	"""a mini-ResNet step takes every fused path; the same step with the layer off takes none and launches more"""
	g = bound().gpuarray
	spec = nets.resnet_spec(stages=((32, 1), (64, 2)), classes=10, stem=16, softmax=False)
	spec = [l if l[0] != "avgpool" else ("avgpool", l[1], 8, 1, 0) for l in spec]
	data, labels = g.to_gpu(np.zeros((4, 3, 64, 64), np.float32)), g.to_gpu(np.zeros((4, ), np.int32))

	counts = {}
	for mode in (True, False):
		lazy.enabled = mode
		np.random.seed(1)
		net = nets.build(spec, name="mini", initscheme="he", actInplace=True)
		trainer, _ = trainerFor(net, 4)
		trainer.train(data, labels, random=False)
		lib.trace.clear()
		lazy.counters.clear()
		trainer.train(data, labels, random=False)
		counts[mode] = (len(names()), dict(lazy.counters))
	lazy.enabled = True

	fusedCalls, fused = counts[True]
	literalCalls, literal = counts[False]
	# (bn_apply_relu: the three relu(bn(.)) tensors in front of the 3x3 layers are written; the three in front of the pointwise
	# layers are evaluated in those layers' forward and filter-gradient gathers — conv_xbn / wgrad_xbn)
	for key, n in (("bn_apply_add", 3), ("bn_apply_relu", 3), ("conv_xbn", 3), ("wgrad_xbn", 3), ("bn_pool", 1), ("bn_bwd_gate", 7), ("wgrad_bn_fold", 4), ("dgrad_bn_fold", 4),
				   ("gate_stats", 1), ("gate_stats_up2", 1), ("compact_dgrad", 2), ("conv_stats", 12), ("gate_by_mask", 2)):
		assert fused.get(key, 0) == n, "%s taken %d times, expected %d (%s)" % (key, fused.get(key, 0), n, fused)
	assert not any(k in literal for k in ("bn_apply_add", "bn_bwd_gate", "wgrad_bn_fold", "gate_stats", "compact_dgrad"))
	assert literalCalls > fusedCalls + 30, (literalCalls, fusedCalls)
	print("fused step: %d launches against %d with the lazy layer off" % (fusedCalls, literalCalls))


def dgrad_stats_counts():
	"""round 6, opt-in (PUZZLE_MI355_DGRAD_STATS=1 sets the pattern "dgradstats"): the backward-data launch of a pointwise layer whose
	input was relu(bn(x)), never written, sums that BatchNorm's backward statistics in its epilogue (pz_conv2d_bwd_data_bnstats);
	the BatchNorm's backward then is ONE pass (pz_bn_bwd_gate_from_partials) — in the mini-ResNet: the three bottleneck tails.
	With the pattern off (the default) nothing of it runs."""
	g = bound().gpuarray
	spec = nets.resnet_spec(stages=((32, 1), (64, 2)), classes=10, stem=16, softmax=False)
	spec = [l if l[0] != "avgpool" else ("avgpool", l[1], 8, 1, 0) for l in spec]
	data, labels = g.to_gpu(np.zeros((4, 3, 64, 64), np.float32)), g.to_gpu(np.zeros((4, ), np.int32))
	seen = {}
	for on in (False, True):
		(lazy.requested.add if on else lazy.requested.discard)("dgradstats")
		np.random.seed(1)
		net = nets.build(spec, name="mini", initscheme="he", actInplace=True)
		trainer, _ = trainerFor(net, 4)
		trainer.train(data, labels, random=False)
		lib.trace.clear()
		lazy.counters.clear()
		trainer.train(data, labels, random=False)
		seen[on] = (names(), dict(lazy.counters))
	lazy.requested.discard("dgradstats")
	off, on = seen[False], seen[True]
	assert "pz_conv2d_bwd_data_bnstats" not in off[0] and "dgrad_bnstats" not in off[1]
	assert on[0].count("pz_conv2d_bwd_data_bnstats") == 3 and on[0].count("pz_bn_bwd_gate_from_partials") == 3, on[1]
	assert on[1]["dgrad_bnstats"] == 3 and on[1]["bn_bwd_gate_from_partials"] == 3
	assert on[1].get("bn_bwd_gate", 0) == off[1]["bn_bwd_gate"] - 3 and on[1]["dgrad_bn_fold"] == off[1]["dgrad_bn_fold"]
	dgrad = lambda calls: sum(calls.count(k) for k in ("pz_conv2d_bwd_data", "pz_conv2d_bwd_data_bn", "pz_conv2d_bwd_data_bnstats"))
	assert dgrad(on[0]) == dgrad(off[0]) and on[0].count("pz_bn_bwd_gate") == off[0].count("pz_bn_bwd_gate") - 3
	assert len(on[0]) == len(off[0]), "one launch replaced by one launch, three times: %d vs %d calls" % (len(on[0]), len(off[0]))
	print("dgrad statistics: 3 backward-data launches carry the BatchNorm sums, 3 BatchNorm backwards run as one pass")


def dgrad_stats_resnet50():
	"""the bench's own step (ResNet-50, batch 256) with the pattern on: all 32 BatchNorms in front of a bottleneck's 3x3 and last 1x1
	layer get their backward sums from the backward-data launch behind them (16 implicit-GEMM epilogues, 16 F(4x4) Winograd
	epilogues); of the 33 two-pass BatchNorm backwards only the stem's is left"""
	lazy.requested.add("dgradstats")
	g = bound().gpuarray
	net = nets.loadResNet(None, "50", actInplace=True, initscheme="none")
	trainer, _ = trainerFor(net, 256)
	data, labels = g.empty((256, 3, 224, 224), np.float32), g.empty((256, ), np.int32)
	net.trainMode()
	for _ in range(3):
		lib.trace.clear()
		lazy.counters.clear()
		trainer.step([data, labels])
		net.reset()
	lazy.requested.discard("dgradstats")
	calls, counts = names(), dict(lazy.counters)
	assert counts.get("dgrad_bnstats") == 32 and counts.get("bn_bwd_gate_from_partials") == 32 and counts.get("bn_bwd_gate") == 1, counts
	assert calls.count("pz_conv2d_bwd_data_bnstats") == 32 and calls.count("pz_bn_bwd_gate") == 1
	print("ResNet-50 b256: 32 backward-data launches carry BatchNorm sums, %d two-pass BatchNorm backward left" % calls.count("pz_bn_bwd_gate"))


def dgrad_stats_values_on_emulation():
	"""the same step WITH VALUES, the C ABI emulated on host buffers (oracle/emu_cabi.py: the header's contract executed with the
	numpy oracle): every parameter gradient of a mini-ResNet step with the statistics taken from the backward-data epilogue
	equals the gradient of the default path to fp32 rounding (the sums are formed in another order), and both match the CPU
	oracle's step. Judges the Python glue — which tensor, which coefficients, which saved mean travel with the gate — not the
	kernel (tests/test_gpu_6_fulltensor.py does that on the device)."""
	sys.path.insert(0, os.path.join(ROOT, "oracle"))
	sys.path.insert(0, os.path.join(ROOT, "tests"))
	import emu_cabi, cpu_net as N, cpu_ref as R
	emu_cabi.install()
	g = bound().gpuarray
	spec = nets.resnet_spec(stages=((8, 1), (16, 2)), classes=10, stem=8, softmax=False)
	spec = [l if l[0] != "avgpool" else ("avgpool", l[1], 8, 1, 0) for l in spec]
	rng = np.random.RandomState(3)
	data, labels = rng.randn(4, 3, 64, 64).astype(np.float32), rng.randint(0, 10, size=(4, )).astype(np.int32)
	grads = {}
	for on in (False, True):
		(lazy.requested.add if on else lazy.requested.discard)("dgradstats")
		np.random.seed(11)
		net = nets.build(spec, name="mini", initscheme="he", actInplace=True)
		params = {name: p.data.get() for name, p in net.namedParams().items()}
		opt = optim.Adam(alpha=1e-3)
		opt.setupOn(net, useGlobalState=True)
		cost = optim.CrossEntropy()
		net.trainMode()
		lazy.counters.clear()
		grad = cost(net(g.to_gpu(data)), g.to_gpu(labels), queryError=False)
		opt.zeroGradParams()
		net.backward(grad, updGrad=False)
		grads[on] = {name: p.grad.get() for name, p in net.namedParams().items()}
		assert (lazy.counters.get("dgrad_bnstats", 0) > 0) == on, lazy.counters
		net.reset()
	lazy.requested.discard("dgradstats")

	_, ashapes = nets.spec_param_shapes(spec)
	attrs = {k: (np.zeros(sh, np.float32) if k.endswith(".mean") else np.ones(sh, np.float32)) for k, sh in ashapes.items()}
	cnet = N.CpuNet(spec, params, attrs)
	cnet.train = True
	_, cgrad = R.cross_entropy(cnet.forward(data), labels)
	cnet.zero_grads()
	cnet.backward(cgrad)

	worst = [0.0, 0.0]
	for name, ref in cnet.grads.items():
		scale = float(np.abs(ref).max()) + 1e-12
		a, b = grads[False][name], grads[True][name]
		worst[0] = max(worst[0], float(np.abs(a - b).max()) / scale)
		worst[1] = max(worst[1], float(np.abs(b - ref).max()) / scale)
	assert worst[0] < 2e-5, "epilogue statistics change a gradient by %.2e of its scale" % worst[0]
	assert worst[1] < 2e-3, "gradients with epilogue statistics differ from the oracle by %.2e of their scale" % worst[1]
	print("dgrad statistics on the emulated C ABI: gradients within %.1e of the default path, %.1e of the oracle" % tuple(worst))


def reference_checkpoint_names():
	"""checkpoint.load resolves the entries of a file the REFERENCE wrote (tests/golden/refckpt_mini_*.npz: Module.save output of
	a small network built from the reference's own residBlock, both naming forms), refuses ambiguous and missing entries."""
	import numpy as np
	from puzzlelib_amd import nets, checkpoint
	golden = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden")
	spec = nets.resnet_spec(stages=((4, 2), (8, 1)), classes=10, stem=8, softmax=False)
	spec = [l if l[0] != "avgpool" else ("avgpool", l[1], 4, 1, 0) for l in spec]
	net = nets.build(spec, name="ResNet-mini")
	for tag in ("unique", "full"):
		path = os.path.join(golden, "refckpt_mini_%s.npz" % tag)
		tensors = checkpoint.read(path)
		assert "meta/json" not in tensors
		link = checkpoint.resolver(tensors, "links", path)
		seen = set()
		for name, param in net.namedParams().items():
			key = link(name)
			assert key.endswith(name) and key not in seen
			seen.add(key)
			assert tensors["params/%d" % int(tensors[key])].shape == param.data.shape, name
		assert len(seen) == len([k for k in tensors if k.startswith("links/")])
		assert checkpoint.load(net, path) == {}
		assert all(l.cfg["passes"] == 0 for l in net.walk() if l.kind == "bn")

	tensors = {"links/netA.blk1.conv.W": np.array(0), "links/netA.blk2.conv.W": np.array(1), "links/fc.W": np.array(2)}
	find = checkpoint.resolver(tensors, "links", "<test>")
	assert find("fc.W") == "links/fc.W"
	for bad, word in (("conv.W", "ambiguous"), ("conv2.W", "no entry")):
		try:
			find(bad)
		except KeyError as e:
			assert word in str(e)
		else:
			raise AssertionError("%s resolved" % bad)


def no_leaks_without_gc():
	"""device memory is returned by reference counting alone: after a step and reset() no activation buffer survives
	(a cycle through a buffer would keep gigabytes until the collector happens to run)"""
	import gc
	from puzzlelib_amd import driver
	g = bound().gpuarray
	spec = nets.resnet_spec(stages=((32, 1), (64, 2)), classes=10, stem=16, softmax=False)
	spec = [l if l[0] != "avgpool" else ("avgpool", l[1], 8, 1, 0) for l in spec]
	net = nets.build(spec, name="mini", initscheme="he", actInplace=True)
	trainer, _ = trainerFor(net, 4)
	data, labels = g.to_gpu(np.zeros((4, 3, 64, 64), np.float32)), g.to_gpu(np.zeros((4, ), np.int32))
	gc.collect()
	gc.disable()
	try:
		live = []
		for _ in range(3):
			trainer.step([data, labels])
			net.reset()
			live.append(sum(1 for o in gc.get_objects() if isinstance(o, driver.Buffer) and o.owner))
	finally:
		gc.enable()
	assert live[0] == live[1] == live[2], "owning buffers alive after each step: %s" % live
	print("no leaks: %d owning buffers alive between steps" % live[0])


# ------------------------------------------------------------------------------------------------ data-parallel planning
def dp_bucket_progress():
	"""the real ResNet-50 arena, in the order the optimizer lays it out: as backward reports layers, all-reduce buckets
	must go out progressively — at least 75 % of the bytes before the last layer reports"""
	net = nets.loadResNet(None, "50", actInplace=True, initscheme="none")
	opt = optim.Adam()
	opt.setupOn(net, useGlobalState=True)
	blocks = grid.arenaBlocks(opt.grads)
	assert [b[0] for b in blocks] == optim.Optimizer.arenaOrder(net)
	assert blocks[0][0].startswith("fc1000") and blocks[-1][0].startswith(("conv1", "bn_conv1"))

	class Ops:
		def markReady(self): return None
		def allreduce(self, *a): pass
		def finish(self, scale): pass

	red = grid.GradReducer(blocks, Ops(), gridsize=8, bucketBytes=25 << 20)
	red.beginStep()
	order = []
	net.gradsReady = lambda layer: [order.append("%s.%s" % (layer.name, k)) for k in layer.params]
	g = bound().gpuarray
	trainer = optim.Trainer(net, optim.CrossEntropy(), opt, batchsize=2)
	trainer.step([g.to_gpu(np.zeros((2, 3, 224, 224), np.float32)), g.to_gpu(np.zeros((2, ), np.int32))])
	assert order == [b[0] for b in blocks], "backward finishes gradients in arena order"

	for name in order:
		red.variableReady(name)
	total = blocks[-1][1] + blocks[-1][2]
	before_last = red.launchedBytes[-2] / total
	assert before_last >= 0.75, "only %.0f %% of the gradient bytes were handed to the transport before the last layer" % (100 * before_last)
	assert len(red.buckets) >= 4 and red.launchedBytes[-1] == sum(b.stop - b.start for b in red.buckets)
	half = next(i for i, b in enumerate(red.launchedBytes) if b / total >= 0.5) / len(order)
	print("dp buckets: %d buckets, %.0f %% of bytes out before the last layer, half of them after %.0f %% of the layers" % (
		len(red.buckets), 100 * before_last, 100 * half
	))

# ------------------------------------------------------------------------------------------------ overlap for an unpatched caller
def singleRankNode():
	"""a one-rank grid in this process (dry run: the communicator is a stand-in, every pz_comm_* call is recorded)"""
	import socket
	with socket.socket() as sock:
		sock.bind(("127.0.0.1", 0))
		port = sock.getsockname()[1]
	return grid.connectNode(0, 1, 0, "127.0.0.1", port)


def dp_auto_overlap_sorted_arena():
	"""the data-parallel exchange overlaps with backward for a caller that only ever calls
	nodeinfo.sumTensor (the reference's unpatched Optimizer): ResNet-50 with the gradient arena in the REFERENCE's sorted-name
	order (Optimizers/Optimizer.py:66-68), no executor hook (`net.gradsReady` unset, `grid.enableOverlap` not called). The
	watcher on the arena learns the completion order from the first two steps; from the third on, scattered completion-set
	buckets go out during backward — at least 80 % of the gradient bytes before the last layer's gradients are written."""
	optim.Optimizer.arenaLayout = "sorted"
	node = singleRankNode()
	net = nets.loadResNet(None, "50", actInplace=True, initscheme="none")
	opt = optim.Adam(nodeinfo=node)
	opt.setupOn(net, useGlobalState=True)
	names_ = [b[0] for b in grid.arenaBlocks(opt.grads)]
	assert names_ == sorted(names_) and net.gradsReady is None and not node.reducers

	g = bound().gpuarray
	trainer = optim.Trainer(net, optim.CrossEntropy(), opt, batchsize=2)
	data, labels = g.to_gpu(np.zeros((2, 3, 224, 224), np.float32)), g.to_gpu(np.zeros((2, ), np.int32))
	per_step = []
	for step in range(5):
		lib.trace.clear()
		trainer.step([data, labels])
		net.reset()
		per_step.append(names())
	watcher = node.watcherOf("grad")
	assert watcher.reducer is not None and watcher.steps == 5
	red = watcher.reducer
	assert len(red.buckets) >= 4 and any(len(b.ranges) > 1 for b in red.buckets), "sorted-name arena -> scattered buckets"
	total = sum(b.nbytes for b in red.buckets)
	assert total >= sum(b[2] for b in watcher.blocks)

	# step 1 ends with the first sumTensor (the watcher is attached there), steps 2-3 are observed: one whole-arena collective at
	# update time each; steps 4-5: grouped collectives inside backward
	for calls in per_step[:3]:
		assert calls.count("pz_comm_allreduce_sum_f32") == 1 and "pz_comm_allreduce_sum_f32_ranges" not in calls
	for calls in per_step[3:]:
		comm = [i for i, n in enumerate(calls) if n.startswith("pz_comm_allreduce")]
		assert len(comm) == len(red.buckets)
		last_wgrad = max(i for i, n in enumerate(calls) if n.startswith("pz_conv2d_bwd_filter"))
		early = [i for i in comm if i < last_wgrad]
		assert len(early) >= len(red.buckets) - 2, "buckets must leave during backward: %s of %s did" % (len(early), len(comm))
		assert calls.index("pz_eltwise", comm[-1]) > comm[-1], "the update follows the last collective"

	# byte progress over the write events of the last step: what was in flight before the LAST block (the stem) was written
	prog = watcher_progress(node, trainer, data, labels, net)
	before_last = prog[-2] / total
	assert before_last >= 0.80, "only %.0f %% of the gradient bytes were in flight before the last layer's gradient" % (100 * before_last)
	half = next(i for i, b in enumerate(prog) if b / total >= 0.5) / len(prog)
	print("auto overlap, sorted-name arena: %d buckets (%d byte ranges), %.0f %% of bytes out before the last gradient write, half after %.0f %% of the writes" % (
		len(red.buckets), sum(len(b.ranges) for b in red.buckets), 100 * before_last, 100 * half))
	node.close()


def watcher_progress(node, trainer, data, labels, net):
	watcher = node.watcherOf("grad")
	log = watcher.launchedBytes = []
	trainer.step([data, labels])
	net.reset()
	return list(log)


def dp_auto_overlap_with_hook():
	"""a hook that writes the whole arena before sumTensor (weight decay; the reference runs hooks first, Optimizer.py:160-167):
	the exchange — early buckets and the rest — completes and the mean is applied as a pass BEFORE the hook's kernel; the
	following sumTensor adds nothing. Config 3's network with the reference script's optimizer and hook."""
	optim.Optimizer.arenaLayout = "sorted"
	node = singleRankNode()
	np.random.seed(1234)
	net = nets.buildNiN()
	opt = optim.MomentumSGD(learnRate=0.1, momRate=0.9, nodeinfo=node)
	opt.addHook(optim.WeightDecay(1e-4))
	opt.setupOn(net, useGlobalState=True)
	opt.targets[0][0].wc = 1.0            # (the flat variable's decay factor is 0 by default, in the reference too: the hook would do nothing)
	g = bound().gpuarray
	trainer = optim.Trainer(net, optim.CrossEntropy(maxlabels=10), opt, batchsize=8)
	data, labels = g.to_gpu(np.zeros((8, 3, 32, 32), np.float32)), g.to_gpu(np.zeros((8, ), np.int32))
	for step in range(5):
		lib.trace.clear()
		trainer.step([data, labels])
		net.reset()
	calls = [(n, a) for n, a in lib.trace if not n.startswith(SKIP)]
	idx = {"comm": [i for i, (n, _) in enumerate(calls) if n.startswith("pz_comm_allreduce")],
		   "decay": [i for i, (n, a) in enumerate(calls) if n == "pz_eltwise" and a[0] == lib.OP_WEIGHT_DECAY],
		   "scale": [i for i, (n, a) in enumerate(calls) if n == "pz_eltwise" and a[0] == lib.OP_LINEAR]}
	assert node.watcherOf("grad").reducer is not None and idx["comm"] and len(idx["decay"]) == 1
	assert max(idx["comm"]) < idx["decay"][0], "every collective is queued before the hook touches the gradients"
	assert any(max(idx["comm"]) < i < idx["decay"][0] for i in idx["scale"]), "the mean is applied before the hook"
	print("auto overlap with a whole-arena hook: %d collectives, mean pass, then the hook" % len(idx["comm"]))
	node.close()


def dp_auto_overlap_conv_bias():
	"""one launch may take several arena write addresses before it is queued — convNdBackwardParams with the
	fused bias path takes wgrad's and the bias gradient's (two barriers, then pz_conv2d_bwd_filter). A block is final only when
	the launch that writes it was ISSUED: with buckets so small that every filter closes one by itself, no bucket's collective
	may be queued before the last library call that carries an address inside that bucket. LeNet (convolutions with biases)
	under the reference script's optimizer, sorted-name arena, no executor hook."""
	import ctypes
	optim.Optimizer.arenaLayout = "sorted"
	node = singleRankNode()
	node.bucketBytes = 256
	np.random.seed(1234)
	net = nets.loadLeNet(None)
	opt = optim.MomentumSGD(learnRate=0.1, momRate=0.9, nodeinfo=node)
	opt.setupOn(net, useGlobalState=True)
	g = bound().gpuarray
	trainer = optim.Trainer(net, optim.CrossEntropy(maxlabels=10), opt, batchsize=8)
	data, labels = g.to_gpu(np.zeros((8, 1, 28, 28), np.float32)), g.to_gpu(np.zeros((8, ), np.int32))

	arena = opt.grads.ary
	base, end = arena.gpudata.ptr, arena.gpudata.ptr + arena.nbytes
	seen = []                                  # (entry point, [arena byte offsets among its arguments]) of every call, in order

	def observe(name, args):
		offsets = []
		for arg in args:
			if type(arg) is int and base <= arg < end:
				offsets.append(arg - base)
			elif isinstance(arg, ctypes.Array) and getattr(arg, "_type_", None) is ctypes.c_void_p:
				offsets += [ptr - base for ptr in arg if ptr is not None and base <= ptr < end]
		seen.append((name, offsets, args))

	lib.issueWatchers.append(observe)
	for step in range(5):
		del seen[:]
		trainer.step([data, labels])
		net.reset()
	lib.issueWatchers.remove(observe)

	watcher = node.watcherOf("grad")
	red = watcher.reducer
	assert red is not None and len(red.buckets) >= 6
	fused = [i for i, (n, offs, _) in enumerate(seen) if n == "pz_conv2d_bwd_filter" and len(offs) == 2]
	assert fused, "LeNet's filter-gradient launches carry the bias gradient's address too"

	comm = [(i, n, args) for i, (n, _, args) in enumerate(seen) if n.startswith("pz_comm_allreduce")]
	assert len(comm) == len(red.buckets)
	early = 0
	last_write = max(i for i, (n, offs, _) in enumerate(seen) if offs and not n.startswith(("pz_comm_", "pz_eltwise")))
	for i, n, args in comm:
		if n == "pz_comm_allreduce_sum_f32":
			ranges = [(args[1] - base, args[1] - base + 4 * args[3])]
		else:
			ranges = [(4 * o, 4 * (o + c)) for o, c in zip(list(args[2]), list(args[3]))]
		writers = [j for j, (m, offs, _) in enumerate(seen) if j <= last_write and not m.startswith("pz_comm_") and
				   any(lo <= off < hi for off in offs for lo, hi in ranges)]
		assert writers and max(writers) < i, "bucket %s was handed to the all-reduce at call %d, its last writer is call %d (%s)" % (
			ranges, i, max(writers), seen[max(writers)][0])
		early += i < last_write
	assert early >= len(comm) // 2, "buckets still leave during backward: %d of %d" % (early, len(comm))
	print("auto overlap, two barriers before one launch: %d buckets, %d left during backward, none before its writer was issued" % (
		len(comm), early))
	node.close()


# ------------------------------------------------------------------------------------------------ runGrid (Grid.py:4-35)
def gridTrain(nodeinfo, verbose, epochs=2):
	"""the body of TestLib/MultiGPUMnist.py:6-57 on synthetic data: identical seeds on every node, LeNet,
	MomentumSGD(nodeinfo=nodeinfo) in global-state mode, a Trainer over this node's shard, nodeinfo.meanValue of the errors"""
	np.random.seed(1234)
	net = nets.loadLeNet(None)
	optimizer = optim.MomentumSGD(learnRate=0.1, momRate=0.9, nodeinfo=nodeinfo)
	optimizer.setupOn(net, useGlobalState=True)
	cost = optim.CrossEntropy(maxlabels=10)
	batch = 128 // nodeinfo.gridsize
	trainer = optim.Trainer(net, cost, optimizer, batchsize=batch)

	rng = np.random.RandomState(7)
	data, labels = rng.randn(1024, 1, 28, 28).astype(np.float32), rng.randint(0, 10, size=(1024, )).astype(np.int32)
	part = data.shape[0] // nodeinfo.gridsize
	start, end = nodeinfo.index * part, (nodeinfo.index + 1) * part
	for epoch in range(epochs):
		trainer.trainFromHost(data[start:end], labels[start:end], macroBatchSize=part)
		trerr = nodeinfo.meanValue(cost.getMeanError())
		if nodeinfo.index == 0 and verbose:
			print("Epoch %s global train error: %s" % (epoch + 1, trerr))
		optimizer.learnRate *= 0.9
	calls = names()
	assert calls.count("pz_comm_init_rank") == 1 and calls.count("pz_comm_broadcast") == 1
	steps = epochs * (part // batch)
	assert sum(1 for n in calls if n.startswith("pz_comm_allreduce")) >= steps, "one exchange per step at least"
	print("node %d of %d on device %d: %d steps, %d collectives" % (nodeinfo.index, nodeinfo.gridsize, nodeinfo.device, steps,
																	sum(1 for n in calls if n.startswith("pz_comm_allreduce"))))


def run_grid(size):
	grid.runGrid(target=gridTrain, size=size, verbose=True, devices=[0] * size)       # (dry run: every node on the simulated device)
	try:
		grid.runGrid(target=gridFails, size=2, devices=[0, 0])
	except RuntimeError as e:
		assert "exited with status" in str(e)
	else:
		raise AssertionError("a dying node must make runGrid raise")
	print("runGrid OK")


def gridFails(nodeinfo):
	if nodeinfo.index == 1:
		raise SystemExit(3)


def run_grid_2():
	run_grid(2)


def run_grid_8():
	run_grid(8)


if __name__ == "__main__":
	globals()[sys.argv[1]]()
