"""
Losses, parameter-update rules and the batch loop of the stand-alone harness (the roles of the reference's Cost/,
Optimizers/ and Handlers/ packages), written against the dispatch surface only.

What has to match the reference is what reaches the backend, and in which order (Handlers/Trainer.py:28-35): per batch
forward, loss + its gradient, zero the gradient accumulators, backward in accumulate mode, update. The update rules are
one table (`RULES`): kernel object, state tensors, scalar arguments — the argument lists the reference's updateVar
methods pass (Optimizers/{SGD,MomentumSGD,NesterovSGD,Adam,RMSProp,AdaGrad,AdaDelta,RMSPropGraves,SMORMS3}.py).

With `useGlobalState` every parameter and every gradient is a view into one flat fp32 arena each (the backend's
SharedArray, Cuda/Utils.py:19-64), so one kernel launch updates the whole model and one bucketed collective reduces all
gradients (Optimizers/Optimizer.py:66-111). The reference lays the arena out in sorted-name order; here the order is the
build's to choose and it is *reverse execution order*: backward finalises gradients front to back in the arena, so the
data-parallel reducer's buckets (puzzlelib_amd/grid.py) complete progressively instead of all at the end.
"""
import math
from collections import OrderedDict

import os

import numpy as np

from puzzlelib_amd.surface import bound as S


# ================================================================================================ losses
class Loss:
	"""Device-side error bookkeeping of Cost/Cost.py:10-118: `devErr` = this batch's summed error, `accumErr` = running
	sum since resetAccumulator(); nothing is copied to the host unless asked."""

	def __init__(self):
		gpuarray = S().gpuarray
		self.accumErr = gpuarray.empty((), dtype=np.float32)
		self.devErr = gpuarray.empty((), dtype=np.float32)
		self.grad = self.valError = None
		self.batchsize = self.numOfSamples = 0
		self.resetAccumulator()

	def resetAccumulator(self):
		self.accumErr.fill(0.0)
		self.batchsize = self.numOfSamples = 0

	def getError(self):
		return self.devErr.get() / self.batchsize

	def getMeanError(self):
		return self.accumErr.get() / self.numOfSamples

	def __call__(self, pred, target, queryError=True):
		assert pred.shape[0] == target.shape[0]
		self.check(pred, target)
		self.grad = self.gradient(pred, target)
		self.accumulate(pred, target)
		self.batchsize = pred.shape[0]
		self.numOfSamples += self.batchsize
		return (self.getError(), self.grad) if queryError else self.grad

	def validate(self, pred, target):
		assert pred.shape[0] == target.shape[0]
		self.check(pred, target)
		self.valError = self.score(pred, target)
		return self.valError

	def check(self, pred, target):
		pass


class CrossEntropy(Loss):
	"""Cost/CrossEntropy.py:13-83: softmax cross-entropy on raw scores, error = summed negative log-likelihood,
	validation = misclassification rate."""

	def __init__(self, maxlabels=None, weights=None):
		super().__init__()
		self.maxlabels = maxlabels
		self.weights = S().gpuarray.to_gpu(weights) if isinstance(weights, np.ndarray) else weights

	def check(self, scores, labels):
		assert scores.ndim > 1 and labels.ndim == scores.ndim - 1 and labels.dtype == np.int32
		assert scores.ndim == 2 or scores.shape[2:] == labels.shape[1:]
		assert not self.maxlabels or scores.shape[1] == self.maxlabels
		assert self.weights is None or self.weights.shape[0] == scores.shape[1]

	def gradient(self, scores, labels):
		self.devErr, grad = S().Costs.crossEntropyKernel(scores, labels, weights=self.weights, error=self.devErr)
		return grad

	def accumulate(self, scores, labels):
		self.accumErr += self.devErr

	def score(self, scores, labels):
		surf = S()
		if scores.ndim == 2:
			best = surf.MatVec.argmax(scores, axis=1)
		else:
			flat = scores.reshape(*scores.shape[:2], int(np.prod(scores.shape[2:])))
			best = surf.MatVec.argmaxBatch(flat, axis=1).reshape(labels.shape)
		wrong = surf.Costs.getAccuracyKernel("calcAccuracy")(best, labels, allocator=surf.gpuarray.memoryPool)
		return wrong.get() / np.prod(labels.shape)


class MSE(Loss):
	"""Cost/MSE.py:7-37: gradient (target - pred) / size, error ||target - pred||^2 / (2 * size) per batch element sum."""

	def check(self, pred, target):
		assert pred.shape[1:] == target.shape[1:]

	def gradient(self, pred, target):
		c = 1.0 / float(np.prod(target.shape))
		return S().Blas.addVectorToVector(target.ravel(), pred.ravel(), alpha=c, beta=-c).reshape(pred.shape)

	def accumulate(self, pred, target):
		g = self.grad.ravel()
		self.devErr.fill(S().Blas.dot(g, g) * float(np.prod(self.grad.shape)) * self.grad.shape[0] / 2.0)
		self.accumErr += self.devErr

	def score(self, pred, target):
		blas = S().Blas
		diff = blas.addVectorToVector(target.ravel(), pred.ravel(), alpha=1.0, beta=-1.0)
		return blas.dot(diff, diff) / (2.0 * float(np.prod(target.shape)))


# ================================================================================================ update rules
def adamStep(opt):
	# bias corrections folded into the step size on the host; the kernel gets (1 - beta) (Optimizers/Adam.py:37-46)
	fix1, fix2 = 1.0 - opt.beta1 ** opt.t, 1.0 - opt.beta2 ** opt.t
	opt.learnRate = opt.alpha * math.sqrt(fix2) / fix1
	return opt.learnRate


# name -> (kernel attribute, state tensors [(key, initial value)], scalar arguments as a function of (opt, param))
RULES = {
	"SGD": ("toVectorAddVectorKer", [], lambda o, p: (o.learnRate * p.learnRate, )),
	"MomentumSGD": ("classicMomSGDKer", [("mom", 0.0)], lambda o, p: (o.learnRate * p.learnRate, o.momRate * p.momRate)),
	"NesterovSGD": ("nesterovMomSGDKer", [("mom", 0.0)], lambda o, p: (o.learnRate * p.learnRate, o.momRate * p.momRate)),
	"Adam": ("adamKer", [("mg", 0.0), ("ms", 0.0)],
			 lambda o, p: (adamStep(o) * p.learnRate, 1.0 - o.beta1, 1.0 - o.beta2, o.epsilon)),
	"RMSProp": ("rmspropKer", [("ms", 0.0)], lambda o, p: (o.learnRate * p.learnRate, o.factor, o.epsilon)),
	"AdaGrad": ("adagradKer", [("h", 0.0)], lambda o, p: (o.learnRate * p.learnRate, o.epsilon)),
	"AdaDelta": ("adadeltaKer", [("msg", 0.0), ("msdx", 0.0)], lambda o, p: (o.rho, o.epsilon)),
	"RMSPropGraves": ("rmspropGravesKer", [("mg", 0.0), ("ms", 0.0), ("delta", 0.0)],
					  lambda o, p: (o.learnRate * p.learnRate, o.alpha, o.momRate * p.momRate, o.epsilon)),
	"SMORMS3": ("smorms3Ker", [("mem", 1.0), ("mg", 0.0), ("ms", 0.0)], lambda o, p: (o.learnRate * p.learnRate, o.epsilon)),
}

DEFAULTS = {
	"SGD": dict(learnRate=1e-3), "MomentumSGD": dict(learnRate=1e-3, momRate=0.9), "NesterovSGD": dict(learnRate=1e-3, momRate=0.9),
	"Adam": dict(alpha=1e-3, beta1=0.9, beta2=0.999, epsilon=1e-8), "RMSProp": dict(learnRate=1e-3, factor=0.9, epsilon=1e-5),
	"AdaGrad": dict(learnRate=1e-3, epsilon=1e-8), "AdaDelta": dict(rho=0.95, epsilon=1e-6),
	"RMSPropGraves": dict(learnRate=1e-4, alpha=0.95, momRate=0.9, epsilon=1e-4), "SMORMS3": dict(learnRate=1e-3, epsilon=1e-16),
}


class WeightDecay:
	"""Optimizers/Hooks.py: grad -= rate * wc * param before the update, for parameters with wc > 0."""

	def __init__(self, rate):
		self.rate = rate

	def __call__(self, param, state, stream=None):
		if param.wc > 0.0:
			S().ElementWise.weightDecayKer(param.grad, param.data, self.rate * param.wc, stream=stream)


class Flat:
	"""The whole model seen as one parameter (what Optimizer.setupGlobalState's globalVar is)."""
	__slots__ = ("name", "data", "grad", "learnRate", "momRate", "wc")

	def __init__(self, data, grad):
		self.name, self.data, self.grad = "*", data, grad
		self.learnRate, self.momRate, self.wc = 1.0, 1.0, 0.0


class Optimizer:
	def __init__(self, rule, nodeinfo=None, **hyper):
		unknown = set(hyper) - set(DEFAULTS[rule])
		if unknown:
			raise TypeError("%s has no hyper-parameter %s" % (rule, sorted(unknown)))
		self.rule, self.nodeinfo = rule, nodeinfo
		self.learnRate = 0.0
		for key, value in dict(DEFAULTS[rule], **hyper).items():
			setattr(self, key, value)

		self.t, self.net, self.hooks = 0, None, []
		self.targets = []                 # [(Param | Flat, {state key: GPUArray})]
		self.params = self.grads = None   # the arenas (SharedArray) under useGlobalState
		self.onGradsFinal = None          # puzzlelib_amd/grid.py: flush / wait for the gradient exchange before the update

	def addHook(self, hook):
		self.hooks.append(hook)

	def newState(self, like):
		gpuarray = S().gpuarray
		state = OrderedDict()
		for key, value in RULES[self.rule][1]:
			state[key] = gpuarray.zeros(like.shape, dtype=np.float32)
			if value != 0.0:
				state[key].set(np.full(like.shape, value, dtype=np.float32))
		return state

	arenaLayout = os.environ.get("PUZZLE_MI355_ARENA", "execution")

	@staticmethod
	def arenaOrder(net):
		"""parameter names in the order backward finishes them: last layer first; inside a residual block the main branch
		(run first by the Parallel's backward, Containers/Parallel.py:127-142) from its tail, then the shortcut"""
		def visit(layers):
			for layer in reversed(layers):
				if layer.kind == "resid":
					for branch in layer.branches:
						yield from visit(branch)
				else:
					for key in layer.params:
						yield "%s.%s" % (layer.name, key)
		return list(visit(net.layers))

	def setupOn(self, net, useGlobalState=False):
		assert self.nodeinfo is None or useGlobalState, "data-parallel training reduces the flat gradient arena"
		self.net = net
		named = net.namedParams()

		if not useGlobalState:
			self.targets = [(param, self.newState(param.data)) for param in named.values()]
			return

		SharedArray = S().gpuarray.SharedArray
		self.params, self.grads = SharedArray(np.float32), SharedArray(np.float32)
		# "execution": blocks in the order backward finishes them (contiguous completion-set buckets); "sorted": the reference's
		# own layout, variables sorted by name (Optimizers/Optimizer.py:66-68) — what an unpatched PuzzleLib hands the backend
		order = self.arenaOrder(net) if self.arenaLayout == "execution" else sorted(named)
		assert sorted(order) == sorted(named)
		for name in order:
			self.params.register(named[name].data.shape, np.float32, name)
			self.grads.register(named[name].grad.shape, np.float32, name)
		self.params.build()
		self.grads.build()
		self.grads.ary.fill(0)                 # alignment gaps must not feed NaNs into the flat update

		for name in order:
			data, grad = self.params[name], self.grads[name]
			data.set(named[name].data)
			grad.set(named[name].grad)
			net.rebind(name, data, grad)

		flat = Flat(self.params.ary, self.grads.ary)
		if self.nodeinfo is not None:
			self.nodeinfo.broadcastBuffer("data", flat.data.gpudata)
		self.targets = [(flat, self.newState(flat.data))]

	def zeroGradParams(self):
		for target, _ in self.targets:
			target.grad.fill(0)

	def update(self, useStreams=False, sync=True):
		"""Optimizers/Optimizer.py:150-196: hooks, then (data-parallel) the gradient mean, then the rule's kernel — per
		parameter, or once over the arena. `useStreams` spreads per-parameter updates over borrowed streams; ordering
		against the main stream is carried by the buffers themselves (lazy.foreignBegin / foreignEnd)."""
		self.t += 1
		surf = S()
		kernel = getattr(surf.ElementWise, RULES[self.rule][0])(np.float32)
		scalars = RULES[self.rule][2]

		flat = self.params is not None
		streams = surf.gpuarray.streamManager.borrow(len(self.targets)) if (useStreams and not flat) else None

		exchange = flat and self.nodeinfo is not None
		# with the overlapped reducer, all-reduces of this step's buckets are still in flight on the communication stream:
		# the exchange is completed before any hook touches the gradients (for the reference's hooks — weight decay with
		# identical parameters on every rank — mean-then-hook equals hook-then-mean)
		# (only for a reducer registered by grid.enableOverlap; without one the reference's order stands — hooks, then
		# sumTensor — and the arena's own watcher completes an overlapped exchange in front of a hook's write, grid.ArenaWatcher)
		early = exchange and "grad" in (getattr(self.nodeinfo, "reducers", None) or {})

		for idx, (target, state) in enumerate(self.targets):
			stream = None if streams is None else streams[idx]
			if early:
				self.nodeinfo.sumTensor("grad", target.grad)
			for hook in self.hooks:
				hook(target, state, stream)
			if exchange and not early:
				self.nodeinfo.sumTensor("grad", target.grad)
			if target.learnRate > 0.0:
				kernel(target.data, target.grad, *state.values(), *scalars(self, target), stream=stream)

		if streams is not None:
			if sync:
				for stream in streams:
					stream.synchronize()
			surf.gpuarray.streamManager.give(streams)


def SGD(learnRate=1e-3, nodeinfo=None):
	return Optimizer("SGD", nodeinfo, learnRate=learnRate)


def MomentumSGD(learnRate=1e-3, momRate=0.9, nodeinfo=None):
	return Optimizer("MomentumSGD", nodeinfo, learnRate=learnRate, momRate=momRate)


def NesterovSGD(learnRate=1e-3, momRate=0.9, nodeinfo=None):
	return Optimizer("NesterovSGD", nodeinfo, learnRate=learnRate, momRate=momRate)


def Adam(alpha=1e-3, beta1=0.9, beta2=0.999, epsilon=1e-8, nodeinfo=None):
	return Optimizer("Adam", nodeinfo, alpha=alpha, beta1=beta1, beta2=beta2, epsilon=epsilon)


def RMSProp(learnRate=1e-3, factor=0.9, epsilon=1e-5, nodeinfo=None):
	return Optimizer("RMSProp", nodeinfo, learnRate=learnRate, factor=factor, epsilon=epsilon)


def AdaGrad(learnRate=1e-3, epsilon=1e-8, nodeinfo=None):
	return Optimizer("AdaGrad", nodeinfo, learnRate=learnRate, epsilon=epsilon)


def AdaDelta(rho=0.95, epsilon=1e-6, nodeinfo=None):
	return Optimizer("AdaDelta", nodeinfo, rho=rho, epsilon=epsilon)


def RMSPropGraves(learnRate=1e-4, alpha=0.95, momRate=0.9, epsilon=1e-4, nodeinfo=None):
	return Optimizer("RMSPropGraves", nodeinfo, learnRate=learnRate, alpha=alpha, momRate=momRate, epsilon=epsilon)


def SMORMS3(learnRate=1e-3, epsilon=1e-16, nodeinfo=None):
	return Optimizer("SMORMS3", nodeinfo, learnRate=learnRate, epsilon=epsilon)


# ================================================================================================ batch loops
def batches(total, size, shuffle):
	count = (total + size - 1) // size
	order = np.random.permutation(count) if shuffle else np.arange(count)
	for n in order:
		yield int(n) * size, min(total, (int(n) + 1) * size)


class Loop:
	"""Cuts (macro-)batches and calls `step` on each (Handlers/Handler.py:20-60). Host data is staged through pinned
	memory on a copy stream one macro-batch ahead (pipeline.HostStager) unless `asyncUpload` is off, which gives the
	reference's synchronous upload."""
	asyncUpload = True

	def __init__(self, net, batchsize=128, onBatchFinish=None):
		self.net, self.batchsize, self.onBatchFinish = net, batchsize, onBatchFinish
		self.stager = None

	def overDevice(self, tensors, shuffle):
		for lo, hi in batches(tensors[0].shape[0], self.batchsize, shuffle):
			self.step([t[lo:hi] for t in tensors])
			self.net.reset()                  # activations go back to the pool
			if self.onBatchFinish:
				self.onBatchFinish(self)

	def overHost(self, arrays, macroBatchSize, shuffle, onMacroBatchFinish=None):
		spans = list(batches(arrays[0].shape[0], macroBatchSize, shuffle))
		if not Loop.asyncUpload:
			to_gpu = S().gpuarray.to_gpu
			for lo, hi in spans:
				self.overDevice([to_gpu(a[lo:hi]) for a in arrays], shuffle)
				if onMacroBatchFinish:
					onMacroBatchFinish(self)
			return

		from puzzlelib_amd.pipeline import HostStager
		S()
		if self.stager is None:
			self.stager = HostStager()            # pinned memory is expensive to allocate: one per loop object
		cut = lambda span: [a[span[0]:span[1]] for a in arrays]
		ticket = self.stager.submit(cut(spans[0])) if spans else None
		for i in range(len(spans)):
			onDevice = self.stager.acquire(ticket)
			current, ticket = ticket, (self.stager.submit(cut(spans[i + 1])) if i + 1 < len(spans) else None)
			self.overDevice(onDevice, shuffle)
			self.stager.release(current)
			if onMacroBatchFinish:
				onMacroBatchFinish(self)


class Trainer(Loop):
	def __init__(self, net, cost, optimizer, onBatchFinish=None, batchsize=128):
		super().__init__(net, batchsize, onBatchFinish)
		self.cost, self.optimizer = cost, optimizer

	def step(self, batch):
		"""Handlers/Trainer.py:28-35"""
		data, target = batch
		grad = self.cost(self.net(data), target, queryError=False)
		self.optimizer.zeroGradParams()
		self.net.backward(grad, updGrad=False)
		self.optimizer.update()

	def train(self, data, target, random=True):
		self.cost.resetAccumulator()
		self.net.trainMode()
		self.overDevice([data, target], random)

	def trainFromHost(self, data, target, macroBatchSize=10000, onMacroBatchFinish=None, random=True):
		self.cost.resetAccumulator()
		self.net.trainMode()
		self.overHost([data, target], macroBatchSize, random, onMacroBatchFinish)


class Validator(Loop):
	def __init__(self, net, cost, onBatchFinish=None, batchsize=128):
		super().__init__(net, batchsize, onBatchFinish)
		self.cost, self.error = cost, 0.0

	def step(self, batch):
		data, target = batch
		self.error += self.cost.validate(self.net(data), target) * data.shape[0]

	def validate(self, data, target):
		self.net.evalMode()
		self.error = 0.0
		self.overDevice([data, target], False)
		self.error /= data.shape[0]
		return self.error

	def validateFromHost(self, data, target, macroBatchSize=10000, onMacroBatchFinish=None):
		self.net.evalMode()
		self.error = 0.0
		self.overHost([data, target], macroBatchSize, False, onMacroBatchFinish)
		self.error /= data.shape[0]
		return self.error
