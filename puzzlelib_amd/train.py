"""
Training-loop pieces of the stand-alone harness: costs, optimizers, hooks, Trainer / Validator.

Behaviour follows the reference (cited per class); everything numeric is a HIP kernel launch through the surface.
  Cost / CrossEntropy / MSE       Cost/Cost.py:10-118, Cost/CrossEntropy.py:13-50, Cost/MSE.py:7-27
  Optimizer (+ global state)      Optimizers/Optimizer.py:11-199 — flat SharedArray arena, ONE update launch, nodeinfo hook
  SGD / MomentumSGD / Adam        Optimizers/SGD.py:11-18, MomentumSGD.py:12-27, Adam.py:13-43
  WeightDecay hook                Optimizers/Hooks.py:11-19
  Trainer / Validator             Handlers/Handler.py:6-86, Trainer.py:7-35, Validator.py
"""
import math
from collections import OrderedDict

import numpy as np

from puzzlelib_amd.settings import Config
from puzzlelib_amd.surface import bound as S
from puzzlelib_amd.nn import Variable


# ================================================================================================ costs
class CostError(Exception):
	pass


class Cost:
	def __init__(self):
		gpuarray = S().gpuarray
		self.accumErr = gpuarray.empty((), dtype=np.float32)
		self.devErr = gpuarray.empty((), dtype=np.float32)

		self.error = self.valError = self.grad = None
		self.batchsize = self.numOfSamples = None
		self.dirty = True

		self.resetAccumulator()


	def resetAccumulator(self):
		self.accumErr.fill(0.0)
		self.batchsize = self.numOfSamples = 0


	def getError(self):
		if self.dirty:
			self.error = self.devErr.get() / self.batchsize
			self.dirty = False
		return self.error


	def getMeanError(self):
		return self.accumErr.get() / self.numOfSamples


	def getValError(self):
		return self.valError


	def __call__(self, pred, target, queryError=True):
		GPUArray = S().gpuarray.GPUArray
		if isinstance(target, GPUArray) and isinstance(pred, GPUArray):
			assert pred.shape[0] == target.shape[0]

		self.checkDataShape(pred, target)
		self.error = self.valError = self.grad = None

		self.grad = self.calcGrad(pred, target)
		self.calcError(pred, target)
		self.dirty = True

		self.batchsize = pred.shape[0]
		self.numOfSamples += self.batchsize

		if queryError:
			self.error = self.getError()
			return self.error, self.grad

		return self.grad


	def validate(self, pred, target):
		self.checkValDataShape(pred, target)
		self.valError = self.calcVal(pred, target)
		return self.valError


	def calcGrad(self, pred, target):
		raise NotImplementedError()


	def calcError(self, pred, target):
		raise NotImplementedError()


	def calcVal(self, pred, target):
		raise NotImplementedError()


	def checkDataShape(self, pred, target):
		pass


	def checkValDataShape(self, pred, target):
		pass


class CrossEntropy(Cost):
	def __init__(self, maxlabels=None, weights=None):
		super().__init__()
		self.maxlabels, self.mostProb = maxlabels, None
		self.weights = S().gpuarray.to_gpu(weights) if isinstance(weights, np.ndarray) else weights


	def calcGrad(self, scores, labels):
		if Config.verifyData:
			self.verifyLabels(scores, labels)

		self.devErr, grad = S().Costs.crossEntropyKernel(scores, labels, weights=self.weights, error=self.devErr)
		return grad


	def calcError(self, scores, labels):
		self.accumErr += self.devErr


	def calcVal(self, scores, labels):
		surf = S()
		if scores.ndim == 2:
			self.mostProb = surf.MatVec.argmax(scores, axis=1)
		else:
			flat = scores.reshape(*scores.shape[:2], int(np.prod(scores.shape[2:])))
			self.mostProb = surf.MatVec.argmaxBatch(flat, axis=1).reshape(labels.shape)

		mismatches = surf.Costs.getAccuracyKernel("calcAccuracy")(
			self.mostProb, labels, allocator=surf.gpuarray.memoryPool
		)
		return mismatches.get() / np.prod(labels.shape)


	def checkDataShape(self, scores, labels):
		assert scores.ndim > 1 and labels.ndim == scores.ndim - 1
		assert labels.dtype == np.int32
		if scores.ndim > 2:
			assert scores.shape[2:] == labels.shape[1:]
		if self.maxlabels:
			assert scores.shape[1] == self.maxlabels
		if self.weights is not None:
			assert self.weights.shape[0] == scores.shape[1]


	checkValDataShape = checkDataShape


	@staticmethod
	def verifyLabels(scores, labels):
		gpuarray = S().gpuarray
		mn, mx = gpuarray.minimum(labels).get(), gpuarray.maximum(labels).get()
		if mn < 0:
			raise CostError("Cross entropy labels verification failed, found index %s (< 0)" % mn)
		if mx >= scores.shape[1]:
			raise CostError("Cross entropy labels verification failed, found index %s (> %s)" % (mx, scores.shape[1] - 1))


class MSE(Cost):
	def calcGrad(self, pred, target):
		c = 1.0 / np.prod(target.shape)
		return S().Blas.addVectorToVector(target.ravel(), pred.ravel(), alpha=c, beta=-c).reshape(pred.shape)


	def calcError(self, pred, target):
		g = self.grad.ravel()
		self.devErr.fill(S().Blas.dot(g, g) * np.prod(self.grad.shape) * self.grad.shape[0] / 2.0)
		self.accumErr += self.devErr


	def calcVal(self, pred, target):
		Blas = S().Blas
		diff = Blas.addVectorToVector(target.ravel(), pred.ravel(), alpha=1.0, beta=-1.0)
		return Blas.dot(diff, diff) / (2.0 * np.prod(target.shape))


	def checkDataShape(self, pred, target):
		assert pred.shape[1:] == target.shape[1:]


	checkValDataShape = checkDataShape


# ================================================================================================ optimizers
class Hook:
	def __call__(self, var, state, stream=None):
		raise NotImplementedError()


class WeightDecay(Hook):
	def __init__(self, rate):
		self.rate = rate


	def __call__(self, var, state, stream=None):
		assert var.grad.dtype == np.float32
		if var.wc > 0.0:
			S().ElementWise.weightDecayKer(var.grad, var.data, self.rate * var.wc, stream=stream)


class Optimizer:
	def __init__(self, nodeinfo=None):
		self.t, self.learnRate = 0, 0.0
		self.module, self.nodeinfo = None, nodeinfo

		self.states, self.hooks, self.customVars = {}, [], []
		self.shParams, self.shGrads = {}, {}

		self.globalState, self.globalVar = False, OrderedDict()


	def addHook(self, hook):
		self.hooks.append(hook)


	def setupOn(self, mod, useGlobalState=False):
		if self.nodeinfo is not None:
			assert useGlobalState

		self.module = mod
		vartable = mod.getVarTable()

		if useGlobalState:
			self.globalState = True
			self.setupGlobalState(vartable)
		else:
			for var, names in vartable.items():
				if var.hasUpdater:
					self.customVars.append(names[0])
				else:
					self.states[names[0]] = self.setupState(var)


	def setupGlobalState(self, vartable):
		"""All parameters (and all gradients) move into one flat arena each, blocks sorted by variable name and aligned
		to 16 B; module variables become views into it, so one kernel launch updates every parameter and one
		collective reduces every gradient (Optimizers/Optimizer.py:66-111)."""
		SharedArray = S().gpuarray.SharedArray
		variables = sorted(((names, var) for var, names in vartable.items()), key=lambda elem: elem[0][0])

		for names, var in variables:
			if var.hasUpdater:
				assert self.nodeinfo is None
				self.customVars.append(names[0])
				continue

			dtype = var.data.dtype.type
			self.shParams.setdefault(dtype, SharedArray(dtype)).register(var.data.shape, dtype, names[0])
			self.shGrads.setdefault(dtype, SharedArray(dtype)).register(var.grad.shape, dtype, names[0])

		for dtype in self.shParams:
			self.shParams[dtype].build()
			self.shGrads[dtype].build()
			self.shGrads[dtype].ary.fill(0)        # alignment gaps must not feed NaNs into the flat update

			self.globalVar[self.shParams[dtype].dtype] = Variable(self.shParams[dtype].ary, grad=self.shGrads[dtype].ary)

		for names, var in variables:
			if var.hasUpdater:
				continue

			dtype = var.data.dtype.type
			data, grad = self.shParams[dtype][names[0]], self.shGrads[dtype][names[0]]
			data.set(var.data)
			grad.set(var.grad)

			for name in names:
				self.module.setVar(name, Variable(data, grad=grad))

		for dtype, globalVar in self.globalVar.items():
			if self.nodeinfo is not None:
				self.nodeinfo.broadcastBuffer("data", globalVar.data.gpudata)
			self.states[dtype] = self.setupState(globalVar)


	def setupState(self, var):
		return {}


	def zeroGradParams(self):
		if self.globalState:
			for globalVar in self.globalVar.values():
				globalVar.grad.fill(0)
		else:
			for name in self.states:
				var = self.module.getVar(name)
				if not var.hasUpdater:
					var.grad.fill(0)


	def update(self, useStreams=False, sync=True):
		self.t += 1

		if self.globalState:
			for dtype, globalVar in self.globalVar.items():
				state = self.states[dtype]

				for hook in self.hooks:
					hook(globalVar, state)

				if self.nodeinfo is not None:
					self.nodeinfo.sumTensor("grad", globalVar.grad)

				if globalVar.learnRate > 0.0:
					self.updateVar(globalVar, state)

		else:
			streams = S().gpuarray.streamManager.borrow(len(self.states)) if useStreams else None

			for i, (name, state) in enumerate(self.states.items()):
				var = self.module.getVar(name)
				assert var.grad is not None and var.data.shape == var.grad.shape
				stream = streams[i] if useStreams else None

				for hook in self.hooks:
					hook(var, state, stream)
				if var.learnRate > 0.0:
					self.updateVar(var, state, stream)

			if useStreams:
				if sync:
					for stream in streams:
						stream.synchronize()
				S().gpuarray.streamManager.give(streams)

		for name in self.customVars:
			self.module.getVar(name).update(self.learnRate)


	def updateVar(self, var, state, stream=None):
		raise NotImplementedError()


class SGD(Optimizer):
	def __init__(self, learnRate=1e-3, nodeinfo=None):
		super().__init__(nodeinfo)
		self.learnRate = learnRate


	def updateVar(self, var, state, stream=None):
		S().ElementWise.toVectorAddVectorKer(var.data.dtype)(
			var.data, var.grad, self.learnRate * var.learnRate, stream=stream
		)


class MomentumSGD(SGD):
	def __init__(self, learnRate=1e-3, momRate=0.9, nodeinfo=None):
		super().__init__(learnRate, nodeinfo)
		self.momRate = momRate


	def setupState(self, var):
		return {"mom": S().gpuarray.zeros(var.data.shape, dtype=var.data.dtype)}


	def updateVar(self, var, state, stream=None):
		S().ElementWise.classicMomSGDKer(var.data.dtype)(
			var.data, var.grad, state["mom"], self.learnRate * var.learnRate, self.momRate * var.momRate, stream=stream
		)


class Adam(Optimizer):
	def __init__(self, alpha=1e-3, beta1=0.9, beta2=0.999, epsilon=1e-8, nodeinfo=None):
		super().__init__(nodeinfo)
		self.alpha, self.beta1, self.beta2, self.epsilon = alpha, beta1, beta2, epsilon


	def setupState(self, var):
		gpuarray = S().gpuarray
		return {
			"mg": gpuarray.zeros(var.data.shape, dtype=np.float32), "ms": gpuarray.zeros(var.data.shape, dtype=np.float32)
		}


	def updateVar(self, var, state, stream=None):
		# bias correction folded into the step size on the host; the kernel sees (1-beta) factors
		fix1, fix2 = 1.0 - self.beta1**self.t, 1.0 - self.beta2**self.t
		self.learnRate = self.alpha * math.sqrt(fix2) / fix1

		S().ElementWise.adamKer(var.data.dtype)(
			var.data, var.grad, state["mg"], state["ms"], self.learnRate * var.learnRate, 1.0 - self.beta1,
			1.0 - self.beta2, self.epsilon, stream=stream
		)


# ================================================================================================ handlers
class Handler:
	def __init__(self, mod, onBatchFinish=None, batchsize=128):
		self.module, self.onBatchFinish, self.batchsize = mod, onBatchFinish, batchsize
		self.currBatch = self.totalBatches = 0
		self.currMacroBatch = self.totalMacroBatches = 0


	@staticmethod
	def dataSize(data):
		while isinstance(data, list):
			data = data[0]
		return data.shape[0]


	@classmethod
	def sliceData(cls, data, idx, size, post):
		if isinstance(data, list):
			return [cls.sliceData(d, idx, size, post) for d in data]
		return post(data[idx * size:(idx + 1) * size])


	asyncUpload = True       # stage macro-batches through pinned memory on a copy stream (pipeline.HostStager)


	def handleFromHost(self, data, state=None, macroBatchSize=10000, onMacroBatchFinish=None, random=True):
		"""Handlers/Handler.py:20-36 uploads each macro-batch synchronously before training on it. Same loop, same
		order and callbacks; with `asyncUpload` the next macro-batch is staged in pinned memory and copied on a side
		stream while the current one trains (Handler.asyncUpload = False gives the reference's literal behaviour)."""
		self.totalMacroBatches = (self.dataSize(data) + macroBatchSize - 1) // macroBatchSize
		order = np.random.permutation(self.totalMacroBatches) if random else np.arange(self.totalMacroBatches)

		if not Handler.asyncUpload:
			to_gpu = S().gpuarray.to_gpu
			for i, n in enumerate(order):
				macrobatch = self.sliceData(data, n, macroBatchSize, to_gpu)
				self.currMacroBatch = i + 1

				self.handle(macrobatch, state, random=random)
				if onMacroBatchFinish:
					onMacroBatchFinish(self)
			return

		from .pipeline import HostStager
		S()                                      # the backend (device, pool) must be up before streams are made
		stager = getattr(self, "stager", None)  # pinned memory is expensive to allocate: one stager per handler
		if stager is None:
			stager = self.stager = HostStager()
		hostSlice = lambda idx: self.sliceData(data, idx, macroBatchSize, lambda dat: dat)

		ticket = stager.submit(hostSlice(order[0])) if len(order) > 0 else None
		for i, n in enumerate(order):
			macrobatch = stager.acquire(ticket)
			current, ticket = ticket, (stager.submit(hostSlice(order[i + 1])) if i + 1 < len(order) else None)
			self.currMacroBatch = i + 1

			self.handle(macrobatch, state, random=random)
			stager.release(current)
			if onMacroBatchFinish:
				onMacroBatchFinish(self)


	def handle(self, data, state=None, random=True):
		self.totalBatches = (self.dataSize(data) + self.batchsize - 1) // self.batchsize
		order = np.random.permutation(self.totalBatches) if random else np.arange(self.totalBatches)

		for i, n in enumerate(order):
			batch = self.sliceData(data, n, self.batchsize, lambda dat: dat)
			self.currBatch = i + 1

			self.handleBatch(batch, n, state)
			self.module.reset()           # drops activations -> buffers go back to the pool

			if self.onBatchFinish:
				self.onBatchFinish(self)


	def handleBatch(self, batch, idx, state):
		raise NotImplementedError()


class Trainer(Handler):
	def __init__(self, mod, cost, optimizer, onBatchFinish=None, batchsize=128):
		super().__init__(mod, onBatchFinish, batchsize)
		self.cost, self.optimizer = cost, optimizer


	def trainFromHost(self, data, target, macroBatchSize=10000, onMacroBatchFinish=None, random=True):
		self.cost.resetAccumulator()
		self.module.trainMode()
		self.handleFromHost([data, target], None, macroBatchSize, onMacroBatchFinish, random=random)


	def train(self, data, target, random=True):
		self.cost.resetAccumulator()
		self.module.trainMode()
		self.handle([data, target], None, random=random)


	def handleBatch(self, batch, idx, state):
		"""forward, cost, zero grads, backward (accumulate mode), update — Handlers/Trainer.py:28-35."""
		data, target = batch
		grad = self.cost(self.module(data), target, queryError=False)

		self.optimizer.zeroGradParams()
		self.module.backward(grad, updGrad=False)
		self.optimizer.update()


class Validator(Handler):
	def __init__(self, mod, cost, onBatchFinish=None, batchsize=128):
		super().__init__(mod, onBatchFinish, batchsize)
		self.cost, self.error = cost, 0.0


	def validateFromHost(self, data, target, macroBatchSize=10000, onMacroBatchFinish=None):
		self.module.evalMode()
		self.error = 0.0
		self.handleFromHost([data, target], None, macroBatchSize, onMacroBatchFinish, random=False)
		self.error /= self.dataSize(data)
		return self.error


	def validate(self, data, target):
		self.module.evalMode()
		self.error = 0.0
		self.handle([data, target], None, random=False)
		self.error /= self.dataSize(data)
		return self.error


	def handleBatch(self, batch, idx, state):
		data, target = batch
		self.error += self.cost.validate(self.module(data), target) * self.dataSize(data)
