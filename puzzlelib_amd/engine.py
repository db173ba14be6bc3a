"""
Spec interpreter: runs a plain-data network description (puzzlelib_amd/nets.py) on the MI355X backend.

This is the build's own harness, not a module system: one `Net` object walks the spec and, for every layer kind, issues
exactly the calls the reference's module of that kind makes on the dispatch surface — same wrapper, same positional
arguments, same order (forward in spec order; backward in reverse with the data gradient before the parameter gradient
of each layer, residual branches first-to-last) — so that what the backend sees is what an unmodified PuzzleLib on top
of it would send. The call sequences this file has to reproduce are cited per handler; tests/test_host_logic.py replays
the reference's own Modules/Containers against the same backend object in dry-run mode and compares the traces.

Nothing here knows about fusion: layers exchange ordinary GPUArrays. Whatever is fused is fused behind the surface
(puzzlelib_amd/lazy.py).

Layer kinds: conv, deconv, linear, bn, act (relu / sigmoid / tanh / leakyRelu / elu / softPlus / clip), maxpool, avgpool,
dropout, flatten, softmax, identity, resid (Replicate -> Parallel(branch, shortcut) -> Add).
"""
import math
from collections import OrderedDict

import numpy as np

from puzzlelib_amd.settings import Config
from puzzlelib_amd.surface import bound as S


class NetError(Exception):
	pass


class Param:
	"""A trainable tensor and its gradient accumulator (the role of Variable.py:5-58); per-parameter multipliers as there."""
	__slots__ = ("name", "data", "grad", "learnRate", "momRate", "wc")

	def __init__(self, name, data, grad=None):
		self.name, self.data = name, data
		self.grad = S().gpuarray.zeros(data.shape, dtype=data.dtype) if grad is None else grad
		self.learnRate, self.momRate, self.wc = 1.0, 1.0, 0.0


class Layer:
	"""One spec entry with its run-time state. `x`/`y` are the forward input/output, `dx` the input gradient; `aux` holds
	what the backward needs besides those (pool workspace, BN saved statistics, dropout words, ...)."""
	__slots__ = ("kind", "name", "cfg", "params", "attrs", "x", "y", "dx", "aux", "branches", "train", "hook")

	def __init__(self, kind, name, **cfg):
		self.kind, self.name, self.cfg = kind, name, cfg
		self.params, self.attrs = OrderedDict(), OrderedDict()
		self.x = self.y = self.dx = self.aux = None
		self.branches = ()
		self.train = not Config.globalEvalMode
		self.hook = None

	def drop(self):
		self.x = self.y = self.dx = self.aux = None
		for branch in self.branches:
			for layer in branch:
				layer.drop()


def pair(v):
	return (int(v), int(v)) if isinstance(v, (int, np.integer)) else tuple(int(a) for a in v)


def fans(shape, transposed=False):
	"""(fan_out, fan_in) of a parameter tensor: a vector counts itself, a matrix is (in, out), a filter bank is
	(out maps, in maps) times the receptive field — the quantities Modules/Module.py:455-476 feeds the init schemes."""
	if len(shape) == 1:
		return shape[0], shape[0]
	if len(shape) == 2:
		fin, fout = shape
	else:
		field = int(np.prod(shape[2:]))
		fout, fin = shape[0] * field, shape[1] * field
	return (fin, fout) if transposed else (fout, fin)


def drawWeights(scheme, shape, wscale, transposed=False):
	"""Initial values with the reference's numpy RNG call per scheme (Modules/Module.py:406-452), so that a seed gives the
	same parameters there and here. `scheme` may be (name, "in" | "out" | "avg")."""
	basis = "in"
	if isinstance(scheme, (tuple, list)):
		scheme, basis = scheme
	fout, fin = fans(shape, transposed)
	factor = {"in": fin, "out": fout, "avg": (fin + fout) / 2}[basis]

	if scheme == "none":
		return None
	if scheme is None or scheme == "xavier_uniform":
		lim = math.sqrt(3.0 / factor)
		return np.random.uniform(-lim, lim, shape).astype(np.float32)
	if scheme in ("xavier", "xavier_normal"):
		return np.random.normal(0, math.sqrt(1.0 / factor), shape).astype(np.float32)
	if scheme == "he":
		return np.random.normal(0.0, math.sqrt(2.0 / factor), shape).astype(np.float32)
	if scheme == "gaussian":
		return np.random.normal(0.0, wscale, shape).astype(np.float32)
	if scheme == "uniform":
		return np.random.uniform(-wscale, wscale, shape).astype(np.float32)
	raise NotImplementedError(scheme)


ACTIVATIONS = {       # name -> (extra scalar arguments) — Modules/Activation.py:57-61
	"relu": (), "sigmoid": (), "tanh": (), "softPlus": (), "leakyRelu": (0.01, ), "elu": (1.0, ), "clip": (0.0, 6.0)
}


class Net:
	def __init__(self, spec, name=None, initscheme=None, wscale=1.0, actInplace=False, bnInplace=False):
		self.name = name
		self.initscheme, self.wscale, self.actInplace, self.bnInplace = initscheme, wscale, actInplace, bnInplace
		self.layers = [self.make(entry, idx) for idx, entry in enumerate(spec)]
		self.train = not Config.globalEvalMode
		self.y = self.dx = None
		self.gradsReady = None        # callable(layer) once a layer's parameter gradients are final (puzzlelib_amd/grid.py)

	# ------------------------------------------------------------------------------------------ construction
	def make(self, entry, idx):
		kind = entry[0]
		surf = S()
		gpuarray = surf.gpuarray

		def named(name):
			return str(idx) if name is None or str(name).isdigit() else name

		def weights(layer, shape, transposed=False):
			init = drawWeights(self.initscheme, shape, self.wscale, transposed)
			data = gpuarray.empty(shape, dtype=np.float32) if init is None else gpuarray.to_gpu(init)
			layer.params["W"] = Param(layer.name + ".W", data)

		if kind in ("conv", "deconv"):
			_, name, cin, cout, size, stride, padding, bias = entry[:8]
			opts = entry[8] if len(entry) > 8 else {}
			groups = opts.get("groups", 1)
			layer = Layer(kind, named(name), stride=pair(stride), pad=pair(padding), dilation=pair(opts.get("dilation", 1)),
						  groups=groups, postpad=pair(opts.get("postpad", 0)))
			if cin % groups or cout % groups:
				raise NetError("%s: %d -> %d maps cannot be split into %d groups" % (name, cin, cout, groups))
			if kind == "conv":
				weights(layer, (cout, cin // groups) + pair(size))
			else:
				weights(layer, (cin, cout // groups) + pair(size), transposed=True)
			if bias:
				layer.params["b"] = Param(layer.name + ".b", gpuarray.zeros((1, cout, 1, 1), dtype=np.float32))
			dnn = surf.Dnn
			layer.cfg["algos"] = (dnn.ConvFwdAlgo.auto, dnn.ConvBwdDataAlgo.auto, dnn.ConvBwdFilterAlgo.auto)
			return layer

		if kind == "linear":
			_, name, nin, nout = entry[:4]
			layer = Layer(kind, named(name))
			init = drawWeights(self.initscheme, (nin, nout), self.wscale)
			data = gpuarray.empty((nin, nout), dtype=np.float32) if init is None else gpuarray.to_gpu(init)
			layer.params["W"] = Param(layer.name + ".W", data)
			layer.params["b"] = Param(layer.name + ".b", gpuarray.zeros((nout, ), dtype=np.float32))
			return layer

		if kind == "bn":
			_, name, maps = entry[:3]
			opts = entry[3] if len(entry) > 3 else {}
			layer = Layer(kind, named(name), maps=maps, epsilon=opts.get("epsilon", 1e-5), initFactor=1.0, minFactor=0.1,
						  inplace=self.bnInplace, passes=0)
			shape = (1, maps, 1, 1)
			gamma = np.random.normal(1.0, opts.get("sscale", 0.01), shape).astype(np.float32)    # Modules/BatchNormND.py:36
			layer.params["scale"] = Param(layer.name + ".scale", gpuarray.to_gpu(gamma))
			layer.params["bias"] = Param(layer.name + ".bias", gpuarray.zeros(shape, dtype=np.float32))
			layer.attrs["mean"] = gpuarray.zeros(shape, dtype=np.float32)
			layer.attrs["var"] = gpuarray.to_gpu(np.ones(shape, dtype=np.float32))
			return layer

		if kind in ACTIVATIONS:
			opts = entry[2] if len(entry) > 2 else {}
			return Layer("act", named(entry[1]), fn=kind, args=tuple(opts.get("args", ACTIVATIONS[kind])),
						 inplace=opts.get("inplace", self.actInplace), slc=opts.get("slice", None))

		if kind in ("maxpool", "avgpool"):
			_, name, size, stride, padding = entry[:5]
			PoolMode = surf.Dnn.PoolMode
			mode = PoolMode.max if kind == "maxpool" else PoolMode.avgWithPad
			return Layer("pool", named(name), size=pair(size), stride=pair(stride), pad=pair(padding), mode=mode)

		if kind == "dropout":
			return Layer(kind, named(entry[1]), p=entry[2], rng=None)

		if kind in ("flatten", "softmax", "identity"):
			return Layer(kind, named(entry[1]) if len(entry) > 1 else str(idx))

		if kind == "resid":
			layer = Layer(kind, "resid%d" % idx)
			main = [self.make(e, i) for i, e in enumerate(entry[1])]
			short = [self.make(e, i) for i, e in enumerate(entry[2])] if len(entry[2]) > 0 else [Layer("identity", "0")]
			layer.branches = (main, short)
			return layer

		raise NotImplementedError(kind)

	# ------------------------------------------------------------------------------------------ bookkeeping
	def walk(self, layers=None):
		"""every leaf layer, in construction order"""
		for layer in (self.layers if layers is None else layers):
			if layer.kind == "resid":
				for branch in layer.branches:
					yield from self.walk(branch)
			else:
				yield layer

	def namedParams(self):
		"""{"<layer>.<param>": Param} in construction order (layer names are unique in the shipped specs)"""
		out = OrderedDict()
		for layer in self.walk():
			for key, param in layer.params.items():
				out["%s.%s" % (layer.name, key)] = param
		return out

	def namedAttrs(self):
		out = OrderedDict()
		for layer in self.walk():
			for key, attr in layer.attrs.items():
				out["%s.%s" % (layer.name, key)] = attr
		return out

	def layerByName(self, name):
		for layer in self.walk():
			if layer.name == name:
				return layer
		return None

	def rebind(self, name, data, grad):
		"""points a parameter at new storage (the optimizer's flat arenas, puzzlelib_amd/optim.py)"""
		lname, key = name.rsplit(".", 1)
		param = self.layerByName(lname).params[key]
		param.data, param.grad = data, grad

	def numOfParams(self):
		return sum(p.data.size for p in self.namedParams().values())

	def trainMode(self):
		self.setMode(True)

	def evalMode(self):
		self.setMode(False)

	def setMode(self, train):
		self.train = train
		for layer in self.walk():
			layer.train = train
		self.reset()

	def reset(self):
		"""drops activations and gradients (Module.reset after every batch, Handlers/Handler.py:58): buffers return to the pool"""
		self.y = self.dx = None
		for layer in self.layers:
			layer.drop()

	def zeroGradParams(self):
		for param in self.namedParams().values():
			param.grad.fill(0)

	def optimizeForShape(self, shape, memlimit=None):
		"""picks, per convolution, the fastest kernel family for this input shape (Modules/ConvND.py:52-61)"""
		dnn = S().Dnn
		limit = float("inf") if memlimit is None else memlimit
		for layer in self.layers:
			if layer.kind == "conv":
				fwd, bwdFilter, bwdData = dnn.convNdbenchmark(
					shape, layer.params["W"].data.shape, layer.cfg["stride"], layer.cfg["pad"], layer.cfg["dilation"],
					layer.cfg["groups"], transpose=False
				)
				layer.cfg["algos"] = (
					next(dnn.ConvFwdAlgo(r.algo.value) for r in fwd if r.memory <= limit),
					next(dnn.ConvBwdDataAlgo(r.algo.value) for r in bwdData if r.memory <= limit),
					next(dnn.ConvBwdFilterAlgo(r.algo.value) for r in bwdFilter if r.memory <= limit)
				)
			shape = outShape(layer, shape)

	# ------------------------------------------------------------------------------------------ execution
	def __call__(self, data):
		return self.forward(data)

	def forward(self, data):
		self.y = self.runForward(self.layers, data)
		return self.y

	def runForward(self, layers, data):
		for idx, layer in enumerate(layers):
			if not Config.disableDtypeShapeChecks:
				checkInput(layer, data, idx)
			layer.x, layer.y = data, None
			data = layer.y = FORWARD[layer.kind](self, layer, data)
		return data

	def backward(self, grad, updParamGrads=True, updGrad=True, scale=1.0, momentum=1.0):
		"""Containers/Sequential.py:212-232: every layer computes its input gradient and then, in training mode,
		accumulates its parameter gradients (`momentum` 1.0 = add to what is there; the trainer zeroes first).
		`updGrad=False` is accepted and — as in the reference, whose branch for it is unreachable (Sequential.py:215-218)
		— changes nothing unless `Net.skipInputGrad` is set."""
		self.dx = self.runBackward(self.layers, grad, updParamGrads, scale, momentum, top=not updGrad)
		return self.dx

	skipInputGrad = False      # honour updGrad=False: do not compute the first layer's input gradient (nobody reads it)

	def runBackward(self, layers, grad, updParamGrads, scale, momentum, top=False):
		for idx in range(len(layers) - 1, -1, -1):
			layer = layers[idx]
			layer.dx = None
			wantInput = not (top and idx == 0 and Net.skipInputGrad)
			grad = BACKWARD[layer.kind](self, layer, grad, updParamGrads and layer.train, scale, momentum, wantInput)
			layer.dx = grad
			if updParamGrads and layer.train and layer.params and self.gradsReady is not None:
				self.gradsReady(layer)
		return grad


# ================================================================================================ shape checks
def outShape(layer, shape):
	kind, cfg = layer.kind, layer.cfg
	if kind == "conv":
		n, _, h, w = shape
		k, _, r, s = layer.params["W"].data.shape
		(sh, sw), (ph, pw), (dh, dw) = cfg["stride"], cfg["pad"], cfg["dilation"]
		return n, k, (h + 2 * ph - dh * (r - 1) - 1) // sh + 1, (w + 2 * pw - dw * (s - 1) - 1) // sw + 1
	if kind == "deconv":
		n, _, h, w = shape
		_, kg, r, s = layer.params["W"].data.shape
		(sh, sw), (ph, pw), (dh, dw), (qh, qw) = cfg["stride"], cfg["pad"], cfg["dilation"], cfg["postpad"]
		return n, kg * cfg["groups"], (h - 1) * sh + dh * (r - 1) - 2 * ph + 1 + qh, (w - 1) * sw + dw * (s - 1) - 2 * pw + 1 + qw
	if kind == "pool":
		n, c, h, w = shape
		(fh, fw), (sh, sw), (ph, pw) = cfg["size"], cfg["stride"], cfg["pad"]
		return n, c, (h + 2 * ph - fh) // sh + 1, (w + 2 * pw - fw) // sw + 1
	if kind == "flatten":
		return shape[0], int(np.prod(shape[1:]))
	if kind == "linear":
		return shape[0], layer.params["W"].data.shape[1]
	if kind == "resid":
		for sub in layer.branches[0]:
			shape = outShape(sub, shape)
	return shape


def checkInput(layer, data, idx):
	"""the shape / dtype errors the reference modules raise before touching the device (checkDataShape / checkDataType)"""
	kind = layer.kind
	if kind == "resid":
		return
	if data.dtype != np.float32:
		raise NetError("layer %d (%s %s): expected dtype float32, got %s" % (idx, kind, layer.name, data.dtype))
	if kind in ("conv", "deconv", "bn", "pool") and data.ndim != 4:
		raise NetError("layer %d (%s %s): data must be a 4d tensor" % (idx, kind, layer.name))
	if kind == "conv":
		want = layer.params["W"].data.shape[1] * layer.cfg["groups"]
		if data.shape[1] != want:
			raise NetError("layer %d (%s): data has %d maps (expected %d)" % (idx, layer.name, data.shape[1], want))
		_, _, r, s = layer.params["W"].data.shape
		for axis, (f, label) in enumerate(((r, "height"), (s, "width"))):
			ext = data.shape[2 + axis] + 2 * layer.cfg["pad"][axis]
			need = layer.cfg["dilation"][axis] * (f - 1) + 1
			if ext < need:
				raise NetError("layer %d (%s): data maps %s is too small (got %d, expected at least %d)" % (
					idx, layer.name, label, ext, need
				))
	elif kind == "bn" and data.shape[1] != layer.cfg["maps"]:
		raise NetError("layer %d (%s): data has %d maps (expected %d)" % (idx, layer.name, data.shape[1], layer.cfg["maps"]))
	elif kind == "linear":
		if data.ndim != 2:
			raise NetError("layer %d (%s): data must be a 2d matrix" % (idx, layer.name))
		if data.shape[1] != layer.params["W"].data.shape[0]:
			raise NetError("layer %d (%s): expected %d data dimensions, %d were given" % (
				idx, layer.name, layer.params["W"].data.shape[0], data.shape[1]
			))


# ================================================================================================ forward handlers
def paramData(layer, key):
	param = layer.params.get(key, None)
	return None if param is None else param.data


def fwdConv(net, layer, x):
	# Modules/ConvND.py:77-81
	c = layer.cfg
	return S().Dnn.convNd(
		x, paramData(layer, "W"), paramData(layer, "b"), stride=c["stride"], pad=c["pad"], dilation=c["dilation"],
		groups=c["groups"], algo=c["algos"][0]
	)


def fwdDeconv(net, layer, x):
	# Modules/DeconvND.py:82-86 (forward = the convolution's backward-data family)
	c = layer.cfg
	return S().Dnn.deconvNd(
		x, paramData(layer, "W"), paramData(layer, "b"), stride=c["stride"], pad=c["pad"], dilation=c["dilation"],
		postpad=c["postpad"], groups=c["groups"], algo=c["algos"][1]
	)


def fwdLinear(net, layer, x):
	# Modules/Linear.py:37-41
	surf = S()
	y = surf.Blas.mulMatrixOnMatrix(x, paramData(layer, "W"), transpB=False)
	surf.MatVec.addVecToMat(paramData(layer, "b"), y, axis=1, out=y)
	return y


def fwdBn(net, layer, x):
	# Modules/BatchNormND.py:50-72
	c, dnn = layer.cfg, S().Dnn
	scale, bias = paramData(layer, "scale"), paramData(layer, "bias")
	mean, var = layer.attrs["mean"], layer.attrs["var"]
	if layer.train:
		if c["inplace"]:
			raise NetError("%s: using inplace flag in train mode is prohibited" % layer.name)
		c["passes"] += 1
		factor = max(c["initFactor"] / c["passes"], c["minFactor"])
		y, savemean, saveinvvar = dnn.batchNormNd(x, scale, bias, mean, var, c["epsilon"], factor, False)
		layer.aux = (savemean, saveinvvar)
		return y
	return dnn.batchNormNd(x, scale, bias, mean, var, c["epsilon"], 0, True, out=x if c["inplace"] else None)


def fwdAct(net, layer, x):
	# Modules/Activation.py:52-55
	surf, c = S(), layer.cfg
	y = x if c["inplace"] else surf.gpuarray.empty(x.shape, dtype=x.dtype, allocator=surf.gpuarray.memoryPool)
	getattr(surf.ElementWise, c["fn"] + "Ker")(x.dtype)(y, x, *c["args"], slice=c["slc"])
	return y


def fwdPool(net, layer, x):
	# Modules/MaxPool2D.py:36-39, AvgPool2D.py
	c = layer.cfg
	y, layer.aux = S().Dnn.poolNd(x, size=c["size"], stride=c["stride"], pad=c["pad"], mode=c["mode"], test=not layer.train)
	return y


def fwdDropout(net, layer, x):
	# Modules/Dropout.py:33-58
	if not layer.train:
		return x
	surf, c = S(), layer.cfg
	gpuarray = surf.gpuarray
	y = gpuarray.empty(x.shape, dtype=x.dtype, allocator=gpuarray.memoryPool)
	words = gpuarray.empty(((x.nbytes + 3) // 4, ), dtype=np.uint32, allocator=gpuarray.memoryPool)
	(gpuarray.globalRng if c["rng"] is None else c["rng"]).fillInteger(words.view(np.uint32))
	keep = 1.0 - c["p"]
	threshold = int(keep * np.iinfo(np.uint32).max)
	surf.ElementWise.dropoutKer(x.dtype)(y, x, words, threshold, np.float32(keep), slice=None)
	layer.aux = (words, threshold)
	return y


def fwdFlatten(net, layer, x):
	return x.reshape(x.shape[0], int(np.prod(x.shape[1:])))


def fwdSoftmax(net, layer, x):
	# Modules/SoftMax.py:18-23
	lifted = x.shape + (1, ) * max(0, 4 - x.ndim)
	return S().Dnn.softmaxNd(x.reshape(lifted)).reshape(x.shape)


def fwdIdentity(net, layer, x):
	return x


def fwdResid(net, layer, x):
	# Replicate (Modules/Replicate.py:19-20) hands the same tensor to both branches of the Parallel
	# (Containers/Parallel.py:96-113), Add sums the results into a zeroed tensor (Modules/Add.py:15-22)
	surf = S()
	outs = [net.runForward(branch, x) for branch in layer.branches]
	y = surf.gpuarray.empty(outs[0].shape, dtype=outs[0].dtype, allocator=surf.gpuarray.memoryPool)
	y.fill(0)
	for term in outs:
		if term.shape != outs[0].shape:
			raise NetError("%s: shape %s is not equal to initial shape %s" % (layer.name, term.shape, outs[0].shape))
		surf.Blas.toVectorAddVector(y.ravel(), term.ravel())
	return y


FORWARD = {
	"conv": fwdConv, "deconv": fwdDeconv, "linear": fwdLinear, "bn": fwdBn, "act": fwdAct, "pool": fwdPool,
	"dropout": fwdDropout, "flatten": fwdFlatten, "softmax": fwdSoftmax, "identity": fwdIdentity, "resid": fwdResid
}


# ================================================================================================ backward handlers
# handler(net, layer, grad, accumulate, scale, momentum, wantInput) -> input gradient

def paramGrad(layer, key):
	param = layer.params.get(key, None)
	return None if param is None else param.grad


def bwdConv(net, layer, g, accumulate, scale, momentum, wantInput):
	# Modules/ConvND.py:84-95
	c, dnn = layer.cfg, S().Dnn
	dx = None
	if wantInput:
		dx = dnn.convNdBackwardData(
			g, paramData(layer, "W"), data=layer.x, stride=c["stride"], pad=c["pad"], dilation=c["dilation"],
			groups=c["groups"], algo=c["algos"][1]
		)
	if accumulate:
		dnn.convNdBackwardParams(
			layer.x, g, paramData(layer, "W"), paramData(layer, "b"), stride=c["stride"], pad=c["pad"], dilation=c["dilation"],
			groups=c["groups"], wgrad=paramGrad(layer, "W"), bgrad=paramGrad(layer, "b"), scale=scale, momentum=momentum,
			algo=c["algos"][2]
		)
	return dx


def bwdDeconv(net, layer, g, accumulate, scale, momentum, wantInput):
	# Modules/DeconvND.py:89-101
	c, dnn = layer.cfg, S().Dnn
	dx = None
	if wantInput:
		dx = dnn.deconvNdBackwardData(
			g, paramData(layer, "W"), data=layer.x, stride=c["stride"], pad=c["pad"], dilation=c["dilation"],
			groups=c["groups"], algo=c["algos"][0]
		)
	if accumulate:
		dnn.deconvNdBackwardParams(
			layer.x, g, paramData(layer, "W"), paramData(layer, "b"), stride=c["stride"], pad=c["pad"], dilation=c["dilation"],
			groups=c["groups"], wgrad=paramGrad(layer, "W"), bgrad=paramGrad(layer, "b"), scale=scale, momentum=momentum,
			algo=c["algos"][2]
		)
	return dx


def bwdLinear(net, layer, g, accumulate, scale, momentum, wantInput):
	# Modules/Linear.py:44-54
	blas = S().Blas
	dx = blas.mulMatrixOnMatrix(g, paramData(layer, "W"), transpB=True) if wantInput else None
	if accumulate:
		blas.mulMatrixOnMatrix(layer.x, g, out=paramGrad(layer, "W"), transpA=True, alpha=scale, beta=momentum)
		blas.sumOnMatrix(g, out=paramGrad(layer, "b"), alpha=scale, beta=momentum)
	return dx


def bwdBn(net, layer, g, accumulate, scale, momentum, wantInput):
	# Modules/BatchNormND.py:75-92: the backward call yields all three gradients; the two parameter gradients are then
	# added into the accumulators with addVectorToVector
	surf = S()
	savemean, saveinvvar = layer.aux
	dx, dscale, dbias = surf.Dnn.batchNormNdBackward(
		layer.x, g, paramData(layer, "scale"), savemean, saveinvvar, layer.cfg["epsilon"]
	)
	if accumulate:
		for key, fresh in (("scale", dscale), ("bias", dbias)):
			acc = paramGrad(layer, key)
			surf.Blas.addVectorToVector(fresh.ravel(), acc.ravel(), out=acc.ravel(), alpha=scale, beta=momentum)
	return dx


def bwdAct(net, layer, g, accumulate, scale, momentum, wantInput):
	# Modules/Activation.py:58-60
	surf, c = S(), layer.cfg
	dx = g if c["inplace"] else surf.gpuarray.empty(g.shape, dtype=g.dtype, allocator=surf.gpuarray.memoryPool)
	getattr(surf.ElementWise, c["fn"] + "DerKer")(g.dtype)(dx, g, layer.y, *c["args"], slice=c["slc"])
	return dx


def bwdPool(net, layer, g, accumulate, scale, momentum, wantInput):
	# Modules/MaxPool2D.py:42-47
	c = layer.cfg
	return S().Dnn.poolNdBackward(layer.x, layer.y, g, layer.aux, size=c["size"], stride=c["stride"], pad=c["pad"], mode=c["mode"])


def bwdDropout(net, layer, g, accumulate, scale, momentum, wantInput):
	# Modules/Dropout.py:61-72
	if not layer.train:
		return g
	surf = S()
	words, threshold = layer.aux
	dx = surf.gpuarray.empty(g.shape, dtype=g.dtype, allocator=surf.gpuarray.memoryPool)
	surf.ElementWise.dropoutKer(g.dtype)(dx, g, words, threshold, 1.0 - layer.cfg["p"], slice=None)
	return dx


def bwdFlatten(net, layer, g, accumulate, scale, momentum, wantInput):
	return g.reshape(layer.x.shape)


def bwdSoftmax(net, layer, g, accumulate, scale, momentum, wantInput):
	# Modules/SoftMax.py:26-33
	lifted = g.shape + (1, ) * max(0, 4 - g.ndim)
	return S().Dnn.softmaxNdBackward(layer.y.reshape(lifted), g.reshape(lifted)).reshape(g.shape)


def bwdIdentity(net, layer, g, accumulate, scale, momentum, wantInput):
	return g


def bwdResid(net, layer, g, accumulate, scale, momentum, wantInput):
	# Add hands the same gradient to every branch (Modules/Add.py:25-26), the Parallel runs its branches' backward passes
	# in order (Containers/Parallel.py:127-142), Replicate sums what comes back into a zeroed tensor (Replicate.py:22-29)
	surf = S()
	grads = [net.runBackward(branch, g, accumulate, scale, momentum) for branch in layer.branches]
	dx = surf.gpuarray.empty(grads[0].shape, dtype=grads[0].dtype, allocator=surf.gpuarray.memoryPool)
	dx.fill(0)
	for term in grads:
		surf.Blas.toVectorAddVector(dx.ravel(), term.ravel())
	return dx


BACKWARD = {
	"conv": bwdConv, "deconv": bwdDeconv, "linear": bwdLinear, "bn": bwdBn, "act": bwdAct, "pool": bwdPool,
	"dropout": bwdDropout, "flatten": bwdFlatten, "softmax": bwdSoftmax, "identity": bwdIdentity, "resid": bwdResid
}
