"""
Module layer of the stand-alone harness: the modules and containers the reference's configs are built from, written
against the dispatch surface (puzzlelib_amd/surface.py) and therefore running entirely on the HIP kernels.

Same class names, constructor arguments, attribute names and call protocol as the reference so that user code and the
parity tests read alike — Module (`__call__`/`backward`/`updateData`/`updateGrad`/`accGradParams`,
Modules/Module.py:124-167), Conv2D (Modules/ConvND.py:14-95, Conv2D.py:11-78), Linear (Modules/Linear.py:12-54),
BatchNorm2D (Modules/BatchNormND.py:12-97), Activation (Modules/Activation.py:37-76), MaxPool2D / AvgPool2D
(Modules/Pool2D.py, MaxPool2D.py:11-47, AvgPool2D.py:9-24), Add, Replicate, Identity, Flatten, SoftMax, Dropout,
and the Sequential / Parallel containers (Containers/Container.py:13-136, Sequential.py:186-232, Parallel.py:96-150).
Checkpoint IO (HDF5), blueprints and graph nodes are out of scope.
"""
import math
from enum import Enum

import numpy as np

from puzzlelib_amd.settings import Config
from puzzlelib_amd.surface import bound as S


class ModuleError(Exception):
	pass


class ContainerError(ModuleError):
	pass


class Variable:
	"""Parameter + gradient pair — Variable.py:5-58."""
	index = 0

	def __init__(self, data, name=None, withgrad=True, grad=None, updater=None, postUpdater=None):
		if name is None:
			name, Variable.index = str(Variable.index), Variable.index + 1

		self.name, self.data, self.updater = name, data, updater
		if updater is not None:
			return

		self.postUpdater = postUpdater
		self.grad = grad

		if grad is None and withgrad and not Config.globalEvalMode:
			self.grad = S().gpuarray.zeros(shape=data.shape, dtype=data.dtype)

		self.learnRate, self.momRate, self.wc = 1.0, 1.0, 0.0


	hasUpdater = property(lambda self: self.updater is not None)
	hasPostUpdater = property(lambda self: getattr(self, "postUpdater", None) is not None)


	def update(self, learnRate):
		self.updater(self, learnRate)


	def postUpdate(self):
		self.postUpdater(self)


	def set(self, variable):
		self.data.set(variable.data)
		if self.grad is not None:
			self.grad.set(variable.grad)


def initTensor(scheme, shape, wscale, factorShape=None, dtype=np.float32, factorTranspose=False):
	"""Parameter initialisation with the reference's RNG call sequence (Modules/Module.py:406-476), so that the same
	numpy seed yields the same initial parameters."""
	factorType = "in"
	if isinstance(scheme, (tuple, list)):
		scheme, factorType = scheme

	fshape = shape if factorShape is None else factorShape
	if len(fshape) == 1:
		outs = ins = fshape[0]
	elif len(fshape) == 2:
		ins, outs = fshape
	else:
		rf = int(np.prod(fshape[2:]))
		outs, ins = fshape[0] * rf, fshape[1] * rf

	if factorTranspose:        # deconvolution filters are stored (inmaps, outmaps, ...): Modules/Module.py:471
		outs, ins = ins, outs

	factor = {"in": ins, "out": outs, "avg": (outs + ins) / 2}[factorType]

	if scheme == "none":
		return None
	elif scheme is None or scheme == "xavier_uniform":
		bound = math.sqrt(3.0 / factor)
		return np.random.uniform(-bound, bound, shape).astype(dtype)
	elif scheme in ("xavier", "xavier_normal"):
		return np.random.normal(0, math.sqrt(1.0 / factor), shape).astype(dtype)
	elif scheme == "he":
		return np.random.normal(0.0, math.sqrt(2.0 / factor), shape).astype(dtype)
	elif scheme == "gaussian":
		return np.random.normal(0.0, wscale, shape).astype(dtype)
	elif scheme == "uniform":
		return np.random.uniform(-wscale, wscale, shape).astype(dtype)
	else:
		raise NotImplementedError(scheme)


def repeat(val, ntimes):
	return (val, ) * ntimes if isinstance(val, (int, np.integer)) else tuple(val)


def shapesOf(data):
	return [shapesOf(d) for d in data] if isinstance(data, (tuple, list)) else data.shape


def dtypesOf(data):
	return [dtypesOf(d) for d in data] if isinstance(data, (tuple, list)) else data.dtype


class backwardScope:
	"""Brackets a module-driven backward pass for the backend: filter gradients issued inside may run on its side stream
	(DnnContext.overlapFilterGrad); leaving the outermost scope joins the streams."""
	def __enter__(self):
		S().Dnn.beginBackward()

	def __exit__(self, *exc):
		S().Dnn.endBackward()
		return False


class Module:
	paramGradsHook = None      # set by puzzlelib_amd.grid.enableOverlap: called after a module's parameter gradients are final

	def __init__(self, name=None):
		self.name = name
		self.vars, self.attrs = {}, {}

		self.gradUsesOutData = self.movesData = self.movesGrad = False
		self.inData = self.data = self.grad = None

		self.train = not Config.globalEvalMode
		self.calctype = np.float32


	# ---- parameters
	def setVar(self, name, var):
		setattr(self, name, var.data)
		self.vars[name] = var


	def getVar(self, name):
		return self.vars[name]


	def setAttr(self, name, attr):
		setattr(self, name, attr)
		self.attrs[name] = attr


	def hasAttr(self, name):
		return name in self.attrs


	def getVarTable(self, vartable=None, name=None, root=True):
		if root and name is None:
			name = self.name if self.name is not None else ""

		vartable = {} if vartable is None else vartable
		for paramName, var in self.vars.items():
			vartable.setdefault(var, []).append("%s%s" % (name, paramName))

		return vartable


	# ---- call protocol
	def __call__(self, data):
		if not Config.disableDtypeShapeChecks:
			self.checkDataShape(shapesOf(data))
			self.checkDataType(dtypesOf(data))

		self.data, self.inData = None, data
		self.updateData(data)
		return self.data


	def backward(self, grad, updParamGrads=True, updGrad=True, scale=1.0, momentum=0.0):
		if not Config.disableDtypeShapeChecks:
			self.checkGradShape(shapesOf(grad))
			self.checkGradType(dtypesOf(grad))

		self.grad = None
		with backwardScope():
			if updGrad:
				self.updateGrad(grad)
			if updParamGrads and self.train:
				self.accGradParams(grad, scale=scale, momentum=momentum)

				if Module.paramGradsHook is not None and self.vars:
					Module.paramGradsHook(self)


	def updateData(self, data):
		raise NotImplementedError()


	def updateGrad(self, grad):
		raise NotImplementedError()


	def accGradParams(self, grad, scale=1.0, momentum=0.0):
		pass


	def zeroGradParams(self):
		for var in self.vars.values():
			if not var.hasUpdater:
				var.grad.fill(0)


	def updateParams(self, learnRate):
		for var in self.vars.values():
			S().Blas.toVectorAddVector(var.data.ravel(), var.grad.ravel(), alpha=learnRate)


	def optimizeForShape(self, shape, memlimit=None):
		pass


	# ---- modes
	def trainMode(self):
		self.train = True
		self.reset()


	def evalMode(self):
		self.train = False
		self.reset()


	def calcMode(self, T):
		if T != np.float32:
			raise ModuleError("Unsupported dtype %s" % T)
		self.calctype = T


	def reset(self):
		self.inData = self.data = self.grad = None


	# ---- checks
	def checkDataShape(self, shape):
		pass


	def checkGradShape(self, shape):
		pass


	def checkDataType(self, dtype):
		self.genericCheckDataType(dtype)


	def checkGradType(self, dtype):
		self.genericCheckDataType(dtype)


	def genericCheckDataType(self, dtype):
		if isinstance(dtype, (tuple, list)):
			for d in dtype:
				self.genericCheckDataType(d)
		elif dtype != self.calctype:
			raise ModuleError("Expected dtype %s, got %s" % (self.calctype, dtype))


	def dataShapeFrom(self, shape):
		raise NotImplementedError()


	def gradShapeFrom(self, shape):
		raise NotImplementedError()


	def numOfParams(self):
		return sum(var.data.size for var in self.vars.values())


	def __str__(self):
		return "Module %s (name: %s)" % (self.__class__.__name__, self.name)


# ================================================================================================ layers
class Conv2D(Module):
	def __init__(self, inmaps, outmaps, size, stride=1, pad=0, dilation=1, wscale=1.0, useBias=True, name=None,
				 initscheme=None, empty=False, groups=1):
		super().__init__(name)

		self.stride, self.pad, self.dilation = repeat(stride, 2), repeat(pad, 2), repeat(dilation, 2)
		self.useBias, self.groups = useBias, groups

		dnn = S().Dnn
		self.fwdAlgo, self.bwdFilterAlgo, self.bwdDataAlgo = \
			dnn.ConvFwdAlgo.auto, dnn.ConvBwdFilterAlgo.auto, dnn.ConvBwdDataAlgo.auto

		if inmaps % groups != 0 or outmaps % groups != 0:
			raise ModuleError(
				"Number of input and output maps must be divisible by number of groups "
				"(%d inmaps, %d outmaps, %d groups)" % (inmaps, outmaps, groups)
			)

		self.W = self.b = None
		if empty:
			return

		gpuarray = S().gpuarray
		Wshape = (outmaps, inmaps // groups, *repeat(size, 2))
		W = initTensor(initscheme, Wshape, wscale)
		self.setVar("W", Variable(gpuarray.empty(Wshape, dtype=self.calctype) if W is None else gpuarray.to_gpu(W)))

		if useBias:
			self.setVar("b", Variable(gpuarray.zeros((1, outmaps, 1, 1), dtype=self.calctype)))


	def updateData(self, data):
		emit = getattr(self, "emitStats", False)      # set per forward pass by Sequential.planFusion
		result = S().Dnn.convNd(
			data, self.W, self.b, stride=self.stride, pad=self.pad, dilation=self.dilation, groups=self.groups,
			algo=self.fwdAlgo, withStats=emit
		)
		self.data, self.outStats = result if emit else (result, None)


	def reset(self):
		super().reset()
		self.outStats = None


	compactGrad = False      # set per forward pass by the enclosing Sequential (planFusion, fuseStridedGrad)

	def updateGrad(self, grad):
		self.grad = S().Dnn.convNdBackwardData(
			grad, self.W, data=self.inData, stride=self.stride, pad=self.pad, dilation=self.dilation,
			groups=self.groups, algo=self.bwdDataAlgo, compact=self.compactGrad
		)


	def accGradParams(self, grad, scale=1.0, momentum=0.0):
		S().Dnn.convNdBackwardParams(
			self.inData, grad, self.W, self.b, stride=self.stride, pad=self.pad, dilation=self.dilation,
			groups=self.groups, wgrad=self.vars["W"].grad, bgrad=self.vars["b"].grad if self.b is not None else None,
			scale=scale, momentum=momentum, algo=self.bwdFilterAlgo
		)


	def optimizeForShape(self, shape, memlimit=None):
		dnn = S().Dnn
		fwd, bwdFilter, bwdData = dnn.convNdbenchmark(
			shape, self.W.shape, self.stride, self.pad, self.dilation, self.groups, transpose=False
		)
		limit = float("inf") if memlimit is None else memlimit

		self.fwdAlgo = next(dnn.ConvFwdAlgo(res.algo.value) for res in fwd if res.memory <= limit)
		self.bwdFilterAlgo = next(dnn.ConvBwdFilterAlgo(res.algo.value) for res in bwdFilter if res.memory <= limit)
		self.bwdDataAlgo = next(dnn.ConvBwdDataAlgo(res.algo.value) for res in bwdData if res.memory <= limit)


	def checkDataShape(self, shape):
		if len(shape) != 4:
			raise ModuleError("Data must be 4d tensor")

		_, inmaps, inh, inw = shape
		_, _, fh, fw = self.W.shape

		if inmaps != self.W.shape[1] * self.groups:
			raise ModuleError("Data has %d maps (expected: %d)" % (inmaps, self.W.shape[1] * self.groups))

		exth, extw = inh + 2 * self.pad[0], inw + 2 * self.pad[1]
		extfh, extfw = self.dilation[0] * (fh - 1) + 1, self.dilation[1] * (fw - 1) + 1

		if exth < extfh:
			raise ModuleError("Data maps height is too small (got %d, expected at least %d)" % (exth, extfh))
		if extw < extfw:
			raise ModuleError("Data maps width is too small (got %d, expected at least %d)" % (extw, extfw))


	def checkGradShape(self, shape):
		if len(shape) != 4:
			raise ModuleError("Grad must be 4d tensor")
		if shape[1] != self.W.shape[0]:
			raise ModuleError("Grad has %d maps (expected: %d)" % (shape[1], self.W.shape[0]))


	def dataShapeFrom(self, shape):
		n, _, inh, inw = shape
		outmaps, _, fh, fw = self.W.shape
		outh = (inh + 2 * self.pad[0] - self.dilation[0] * (fh - 1) - 1) // self.stride[0] + 1
		outw = (inw + 2 * self.pad[1] - self.dilation[1] * (fw - 1) - 1) // self.stride[1] + 1
		return n, outmaps, outh, outw


	def gradShapeFrom(self, shape):
		n, _, outh, outw = shape
		_, inmaps, fh, fw = self.W.shape
		inh = (outh - 1) * self.stride[0] + self.dilation[0] * (fh - 1) - 2 * self.pad[0] + 1
		inw = (outw - 1) * self.stride[1] + self.dilation[1] * (fw - 1) - 2 * self.pad[1] + 1
		return n, inmaps * self.groups, inh, inw


class Deconv2D(Module):
	"""Transposed convolution (Modules/DeconvND.py + Modules/Deconv2D.py): forward is the convolution's backward-data pass
	plus a bias over the produced maps, backward-data is the convolution's forward, and the parameter gradient is the
	convolution's filter gradient with the tensor roles swapped. W is (inmaps, outmaps / groups, fh, fw)."""

	def __init__(self, inmaps, outmaps, size, stride=1, pad=0, dilation=1, postpad=0, wscale=1.0, useBias=True,
				 name=None, initscheme=None, empty=False, groups=1):
		super().__init__(name)

		self.stride, self.pad, self.dilation = repeat(stride, 2), repeat(pad, 2), repeat(dilation, 2)
		self.postpad = repeat(postpad, 2)
		if any(p >= max(s, d) for p, s, d in zip(self.postpad, self.stride, self.dilation)):
			raise ModuleError("Postpad must be smaller than stride and dilation")

		self.useBias, self.groups = useBias, groups

		dnn = S().Dnn
		self.fwdAlgo, self.bwdFilterAlgo, self.bwdDataAlgo = \
			dnn.ConvFwdAlgo.auto, dnn.ConvBwdFilterAlgo.auto, dnn.ConvBwdDataAlgo.auto

		if inmaps % groups != 0 or outmaps % groups != 0:
			raise ModuleError(
				"Number of input and output maps must be divisible by number of groups "
				"(%d inmaps, %d outmaps, %d groups)" % (inmaps, outmaps, groups)
			)

		self.W = self.b = None
		if empty:
			return

		gpuarray = S().gpuarray
		Wshape = (inmaps, outmaps // groups, *repeat(size, 2))
		W = initTensor(initscheme, Wshape, wscale, factorTranspose=True)
		self.setVar("W", Variable(gpuarray.empty(Wshape, dtype=self.calctype) if W is None else gpuarray.to_gpu(W)))

		if useBias:
			# the reference sizes the bias (1, outmaps / groups, 1, 1) (Modules/DeconvND.py:51), which only fits the produced
			# maps for groups == 1; the bias here always covers every produced map
			self.setVar("b", Variable(gpuarray.zeros((1, outmaps, 1, 1), dtype=self.calctype)))


	def updateData(self, data):
		self.data = S().Dnn.deconvNd(
			data, self.W, self.b, stride=self.stride, pad=self.pad, dilation=self.dilation, postpad=self.postpad,
			groups=self.groups, algo=self.bwdDataAlgo
		)


	def updateGrad(self, grad):
		self.grad = S().Dnn.deconvNdBackwardData(
			grad, self.W, data=self.inData, stride=self.stride, pad=self.pad, dilation=self.dilation,
			groups=self.groups, algo=self.fwdAlgo
		)


	def accGradParams(self, grad, scale=1.0, momentum=0.0):
		S().Dnn.deconvNdBackwardParams(
			self.inData, grad, self.W, self.b, stride=self.stride, pad=self.pad, dilation=self.dilation,
			groups=self.groups, wgrad=self.vars["W"].grad, bgrad=self.vars["b"].grad if self.b is not None else None,
			scale=scale, momentum=momentum, algo=self.bwdFilterAlgo
		)


	def checkDataShape(self, shape):
		if len(shape) != 4:
			raise ModuleError("Data must be 4d tensor")
		if shape[1] != self.W.shape[0]:
			raise ModuleError("Data has %d maps (expected: %d)" % (shape[1], self.W.shape[0]))


	def checkGradShape(self, shape):
		if len(shape) != 4:
			raise ModuleError("Grad must be 4d tensor")

		_, outmaps, outh, outw = shape
		_, _, fh, fw = self.W.shape

		if outmaps != self.W.shape[1] * self.groups:
			raise ModuleError("Grad has %d maps (expected: %d)" % (outmaps, self.W.shape[1] * self.groups))
		if outh + 2 * self.pad[0] < self.dilation[0] * (fh - 1) + 1:
			raise ModuleError("Grad maps height is too small (got %d, expected at least %d)" % (
				outh + 2 * self.pad[0], self.dilation[0] * (fh - 1) + 1
			))
		if outw + 2 * self.pad[1] < self.dilation[1] * (fw - 1) + 1:
			raise ModuleError("Grad maps width is too small (got %d, expected at least %d)" % (
				outw + 2 * self.pad[1], self.dilation[1] * (fw - 1) + 1
			))


	def dataShapeFrom(self, shape):
		n, _, inh, inw = shape
		_, outmaps, fh, fw = self.W.shape
		outh = (inh - 1) * self.stride[0] + self.dilation[0] * (fh - 1) - 2 * self.pad[0] + 1 + self.postpad[0]
		outw = (inw - 1) * self.stride[1] + self.dilation[1] * (fw - 1) - 2 * self.pad[1] + 1 + self.postpad[1]
		return n, outmaps * self.groups, outh, outw


	def gradShapeFrom(self, shape):
		n, _, outh, outw = shape
		inmaps, _, fh, fw = self.W.shape
		inh = (outh + 2 * self.pad[0] - self.dilation[0] * (fh - 1) - 1) // self.stride[0] + 1
		inw = (outw + 2 * self.pad[1] - self.dilation[1] * (fw - 1) - 1) // self.stride[1] + 1
		return n, inmaps, inh, inw


class Linear(Module):
	def __init__(self, insize, outsize, wscale=1.0, useBias=True, initscheme=None, name=None, empty=False,
				 transpose=False):
		super().__init__(name)
		self.transpose, self.useBias = transpose, useBias
		self.W = self.b = None

		if empty:
			return

		gpuarray = S().gpuarray
		Wshape, bshape = ((outsize, insize), (insize, )) if transpose else ((insize, outsize), (outsize, ))
		W = initTensor(initscheme, Wshape, wscale, factorShape=Wshape)

		self.setVar("W", Variable(gpuarray.empty(Wshape, dtype=self.calctype) if W is None else gpuarray.to_gpu(W)))
		if useBias:
			self.setVar("b", Variable(gpuarray.zeros(bshape, dtype=self.calctype)))


	def updateData(self, data):
		self.data = S().Blas.mulMatrixOnMatrix(data, self.W, transpB=self.transpose)
		if self.useBias:
			S().MatVec.addVecToMat(self.b, self.data, axis=1, out=self.data)


	def updateGrad(self, grad):
		self.grad = S().Blas.mulMatrixOnMatrix(grad, self.W, transpB=not self.transpose)


	def accGradParams(self, grad, scale=1.0, momentum=0.0):
		Blas = S().Blas
		A, B = (self.inData, grad) if not self.transpose else (grad, self.inData)
		Blas.mulMatrixOnMatrix(A, B, out=self.vars["W"].grad, transpA=True, alpha=scale, beta=momentum)

		if self.useBias:
			Blas.sumOnMatrix(grad, out=self.vars["b"].grad, alpha=scale, beta=momentum)


	def checkDataShape(self, shape):
		if len(shape) != 2:
			raise ModuleError("Data must be 2d matrix")
		expected = self.W.shape[1] if self.transpose else self.W.shape[0]
		if shape[1] != expected:
			raise ModuleError("Expected %d data dimensions, %d were given" % (expected, shape[1]))


	def checkGradShape(self, shape):
		if len(shape) != 2:
			raise ModuleError("Grad must be 2d matrix")
		expected = self.W.shape[0] if self.transpose else self.W.shape[1]
		if shape[1] != expected:
			raise ModuleError("Expected %d grad dimensions, %d were given" % (expected, shape[1]))


	def dataShapeFrom(self, shape):
		return shape[0], (self.W.shape[0] if self.transpose else self.W.shape[1])


	def gradShapeFrom(self, shape):
		return shape[0], (self.W.shape[1] if self.transpose else self.W.shape[0])


class BatchNorm2D(Module):
	def __init__(self, maps, epsilon=1e-5, initFactor=1.0, minFactor=0.1, sscale=0.01, affine=True, name=None,
				 empty=False, inplace=False):
		super().__init__(name)
		self.inplace = inplace
		self.maps, self.epsilon = maps, epsilon
		self.initFactor, self.minFactor, self.numOfProps = initFactor, minFactor, 0
		self.affine = affine

		self.scale = self.bias = self.mean = self.var = None
		self.savemean = self.saveinvvar = self.scalegrad = self.biasgrad = None
		self.fusedRelu = False       # set per forward pass by Sequential (planFusion): output is relu(bn(x))
		self.statsFrom = None        # ... and the Conv2D right in front whose epilogue sums this layer's input per channel
		self.deferApply = False      # ... the only consumer is a residual Add that normalises on the fly (DeferredBN output)
		self.bwdPartials = None      # (grad tensor, partial sums) left by the Replicate fan-in that produced this layer's grad
		self.foldBackwardInto = None # the Conv2D in front that applies this layer's backward while gathering its gradient

		if empty:
			return

		gpuarray = S().gpuarray
		shape = (1, maps, 1, 1)
		scale = np.random.normal(1.0, sscale if affine else 0.0, shape).astype(self.calctype)

		self.setVar("scale", Variable(gpuarray.to_gpu(scale)))
		self.setVar("bias", Variable(gpuarray.zeros(shape, dtype=self.calctype)))
		self.setAttr("mean", gpuarray.zeros(shape, dtype=self.calctype))
		self.setAttr("var", gpuarray.to_gpu(np.ones(shape, dtype=self.calctype)))


	def updateData(self, data):
		dnn = S().Dnn

		if self.train:
			if self.inplace:
				raise ModuleError("%s: using inplace flag in train mode is prohibited" % self)

			self.numOfProps += 1
			factor = max(self.initFactor / self.numOfProps, self.minFactor)

			self.data, self.savemean, self.saveinvvar = dnn.batchNormNd(
				data, self.scale, self.bias, self.mean, self.var, self.epsilon, factor, False, fuseRelu=self.fusedRelu,
				convStats=self.statsFrom.outStats if self.statsFrom is not None else None,
				defer=self.deferApply and not self.fusedRelu
			)
		else:
			self.data = dnn.batchNormNd(
				data, self.scale, self.bias, self.mean, self.var, self.epsilon, 0, True,
				out=data if self.inplace else None
			)


	def backward(self, grad, updParamGrads=True, updGrad=True, scale=1.0, momentum=0.0):
		"""Module.backward (Modules/Module.py) runs updateGrad then accGradParams; when both are wanted the parameter
		gradients are accumulated by the backward kernel itself (pz_bn_bwd_acc) instead of two extra vector kernels."""
		self.accumulate = None
		if updGrad and updParamGrads and self.train and self.affine:
			self.accumulate = (self.vars["scale"].grad, self.vars["bias"].grad, scale, momentum)
		try:
			super().backward(grad, updParamGrads=updParamGrads, updGrad=updGrad, scale=scale, momentum=momentum)
		finally:
			self.accumulate = None


	def updateGrad(self, grad):
		partials, self.bwdPartials = self.bwdPartials, None
		tup = S().Dnn.batchNormNdBackward(
			self.inData, grad, self.scale, self.savemean, self.saveinvvar, self.epsilon,
			bias=self.bias if self.fusedRelu else None, fuseRelu=self.fusedRelu,
			accumulate=getattr(self, "accumulate", None),
			partials=partials[1] if partials is not None and partials[0] is grad and not self.fusedRelu else None,
			lazyGrad=self.foldBackwardInto is not None
		)
		if self.affine:
			self.grad, self.scalegrad, self.biasgrad = tup
		else:
			self.grad = tup[0]


	def accGradParams(self, grad, scale=1.0, momentum=0.0):
		if not self.affine or getattr(self, "accumulate", None) is not None:     # already done inside updateGrad
			return

		Blas = S().Blas
		for name, fresh in (("scale", self.scalegrad), ("bias", self.biasgrad)):
			dst = self.vars[name].grad.ravel()
			Blas.addVectorToVector(fresh.ravel(), dst, out=dst, alpha=scale, beta=momentum)


	def reset(self):
		super().reset()
		self.savemean = self.saveinvvar = None
		if self.affine:
			self.scalegrad = self.biasgrad = None


	def checkDataShape(self, shape):
		if len(shape) != 4:
			raise ModuleError("Data must be 4d tensor")
		if shape[1] != self.maps:
			raise ModuleError("Data has %d maps (expected: %d)" % (shape[1], self.maps))


	def checkGradShape(self, shape):
		if len(shape) != 4:
			raise ModuleError("Grad must be 4d tensor")
		if shape[1] != self.maps:
			raise ModuleError("Grad has %d maps (expected: %d)" % (shape[1], self.maps))


	def dataShapeFrom(self, shape):
		return shape


	def gradShapeFrom(self, shape):
		return shape


class ActivationType(str, Enum):
	sigmoid = "sigmoid"
	tanh = "tanh"
	relu = "relu"
	leakyRelu = "leakyRelu"
	elu = "elu"
	softPlus = "softPlus"
	clip = "clip"


sigmoid, tanh, relu, leakyRelu, elu, softPlus, clip = (
	ActivationType.sigmoid, ActivationType.tanh, ActivationType.relu, ActivationType.leakyRelu, ActivationType.elu,
	ActivationType.softPlus, ActivationType.clip
)


class Activation(Module):
	defaultArgs = {ActivationType.leakyRelu: (0.01, ), ActivationType.elu: (1.0, ), ActivationType.clip: (0.0, 6.0)}

	def __init__(self, activation, slc=None, inplace=False, name=None, args=()):
		super().__init__(name)
		self.gradUsesOutData = True
		self.inplace, self.slc = inplace, slc

		self.activation = ActivationType(activation)
		kernels = S().ElementWise
		self.actFunc = getattr(kernels, "%sKer" % self.activation.value)
		self.actFuncDer = getattr(kernels, "%sDerKer" % self.activation.value)
		self.actArgs = tuple(args) if len(args) > 0 else self.defaultArgs.get(self.activation, ())

		# set per forward pass by Sequential (planFusion) for in-place ReLUs only:
		self.dataFused = False       # the producer (BatchNorm2D / Add) already wrote relu(.) into `data`
		self.gradFused = False       # the ReLU derivative is applied by the producer's or the consumer's backward


	def fusable(self):
		return self.activation == ActivationType.relu and self.inplace and self.slc is None


	def allocLike(self, ary):
		gpuarray = S().gpuarray
		return gpuarray.empty(ary.shape, dtype=ary.dtype, allocator=gpuarray.memoryPool)


	def updateData(self, data):
		if self.dataFused:
			self.data = data
			return

		self.data = data if self.inplace else self.allocLike(data)
		self.actFunc(data.dtype)(self.data, data, *self.actArgs, slice=self.slc)


	def updateGrad(self, grad):
		if self.gradFused:
			self.grad = grad
			return

		self.grad = grad if self.inplace else self.allocLike(grad)
		self.actFuncDer(grad.dtype)(self.grad, grad, self.data, *self.actArgs, slice=self.slc)


	def dataShapeFrom(self, shape):
		return shape


	def gradShapeFrom(self, shape):
		return shape


class Pool2D(Module):
	def __init__(self, size=2, stride=2, pad=0, name=None):
		super().__init__(name)
		self.gradUsesOutData = True
		self.size, self.stride, self.pad = repeat(size, 2), repeat(stride, 2), repeat(pad, 2)
		self.workspace = None
		self.mode = None


	def updateData(self, data):
		self.data, self.workspace = S().Dnn.poolNd(
			data, size=self.size, stride=self.stride, pad=self.pad, mode=self.mode, test=not self.train
		)


	def updateGrad(self, grad):
		self.grad = S().Dnn.poolNdBackward(
			self.inData, self.data, grad, self.workspace, size=self.size, stride=self.stride, pad=self.pad,
			mode=self.mode
		)


	def reset(self):
		super().reset()
		self.workspace = None


	def checkDataShape(self, shape):
		if len(shape) != 4:
			raise ModuleError("Data must be 4d tensor")

		for axis, label in ((0, "height"), (1, "width")):
			ext = shape[2 + axis] + 2 * self.pad[axis]
			if ext < self.size[axis]:
				raise ModuleError("Data maps %s is too small (got %d, expected at least %d)" % (label, ext, self.size[axis]))


	def checkGradShape(self, shape):
		if len(shape) != 4:
			raise ModuleError("Grad must be 4d tensor")


	def dataShapeFrom(self, shape):
		n, maps, inh, inw = shape
		outh = (inh + 2 * self.pad[0] - self.size[0]) // self.stride[0] + 1
		outw = (inw + 2 * self.pad[1] - self.size[1]) // self.stride[1] + 1
		return n, maps, outh, outw


	def gradShapeFrom(self, shape):
		n, maps, outh, outw = shape
		inh = (outh - 1) * self.stride[0] - 2 * self.pad[0] + self.size[0]
		inw = (outw - 1) * self.stride[1] - 2 * self.pad[1] + self.size[1]
		return n, maps, inh, inw


class MaxPool2D(Pool2D):
	def __init__(self, size=2, stride=2, pad=0, useMask=False, name=None):
		super().__init__(size, stride, pad, name)
		if useMask:
			raise NotImplementedError("mask pooling (Backend/Kernels/Pool.py) is outside the implemented operator path")
		self.mode = S().Dnn.PoolMode.max


class AvgPool2D(Pool2D):
	def __init__(self, size=2, stride=2, pad=0, includePad=True, name=None):
		super().__init__(size, stride, pad, name)
		PoolMode = S().Dnn.PoolMode
		self.mode = PoolMode.avgWithPad if includePad else PoolMode.avgNoPad


def isDeferred(obj):
	return type(obj).__name__ == "DeferredBN"


def sumTensors(tensors, relu=False, gate=None):
	"""memset + one axpy per input in the reference (Modules/Add.py:15-22, Replicate.py:22-29: 28 B/elem for two
	inputs); here the first two inputs are summed by one 3-operand kernel (12 B/elem), further ones by axpy.
	0 + a + b == a + b exactly in fp32, so results are bit-identical.
	Two-input sums can absorb a neighbouring in-place ReLU: relu=True gives relu(a + b), gate=y gives (a + b)*(y > 0)."""
	surf = S()
	first = tensors[0]
	out = surf.gpuarray.empty(first.shape, dtype=first.dtype, allocator=surf.gpuarray.memoryPool)

	if relu or gate is not None:
		assert len(tensors) == 2
		if relu:
			surf.ElementWise.add3ReluKer(out, tensors[0], tensors[1])
		else:
			surf.ElementWise.add3GateKer(out, tensors[0], tensors[1], gate)
		return out

	if len(tensors) == 1:
		out.set(first)
		return out

	surf.ElementWise.add3Ker(out, tensors[0], tensors[1])
	for extra in tensors[2:]:
		surf.Blas.toVectorAddVector(out.ravel(), extra.ravel())

	return out


class Add(Module):
	def __init__(self, name=None):
		super().__init__(name)
		self.movesGrad = True
		self.fusedRelu = False       # set per forward pass by Sequential (planFusion)
		self.emitMask = False        # ... the fused ReLU's sign mask is wanted by the gradient fan-in behind it (fuseReluMask)
		self.reluMask = None


	def updateData(self, data):
		self.reluMask = None
		lazy = [isDeferred(d) for d in data]
		if any(lazy):
			if len(data) == 2:            # normalise the deferred BatchNorm outputs while summing them
				first, second = (data[0], data[1]) if lazy[0] else (data[1], data[0])
				if self.fusedRelu and self.emitMask:
					self.data, self.reluMask = S().Dnn.bnApplyAdd(first, second, relu=True, withMask=True)
				else:
					self.data = S().Dnn.bnApplyAdd(first, second, relu=self.fusedRelu)
				return
			data = [d.materialize() if isDeferred(d) else d for d in data]

		if self.fusedRelu and len(data) != 2:
			self.data = sumTensors(data)
			S().ElementWise.reluKer(self.data.dtype)(self.data, self.data)
		else:
			self.data = sumTensors(data, relu=self.fusedRelu)


	def updateGrad(self, grad):
		self.grad = [grad] * len(self.inData)


	def checkDataShape(self, shapes):
		for shape in shapes:
			if shape != shapes[0]:
				raise ModuleError("Shape %s is not equal to initial shape %s" % (shape, shapes[0]))


	def dataShapeFrom(self, shape):
		return shape[0]


	def gradShapeFrom(self, shape):
		return [shape] * len(self.inData)


class Replicate(Module):
	def __init__(self, times, name=None):
		super().__init__(name)
		self.movesData = True
		self.times = times
		self.gateGrad = False        # set per forward pass by Sequential (planFusion): input is an in-place ReLU's output
		self.statsFor = []           # ... and BatchNorms of the block in front whose backward statistics the fan-in also sums
		self.maskFrom = None         # ... and the Add that produced the input together with its sign mask


	def updateData(self, data):
		self.data = [data] * self.times


	def updateGrad(self, grad):
		targets = [bn for bn in self.statsFor if bn.train and bn.savemean is not None and not bn.fusedRelu and
				   bn.inData is not None and bn.inData.shape == self.inData.shape]

		if self.gateGrad and len(grad) == 2 and targets:
			# fan-in + ReLU derivative + the statistics pass of the BatchNorm backward(s) this gradient goes to next
			# (two compact stride-2 gradients are expanded on the fly, see planFusion / fuseStridedGrad)
			mask = self.maskFrom.reluMask if self.maskFrom is not None else None
			self.grad, parts = S().Dnn.bnGateStats(
				grad[0], grad[1], self.inData, [(bn.inData, bn.savemean) for bn in targets], mask=mask
			)
			for bn, part in zip(targets, parts):
				bn.bwdPartials = (self.grad, part)
		else:
			grad = [g.materialize() if hasattr(g, "compact") else g for g in grad]
			self.grad = sumTensors(grad, gate=self.inData if self.gateGrad else None)


	def dataShapeFrom(self, shape):
		return [shape] * self.times


	def gradShapeFrom(self, shape):
		return shape[0]


class Identity(Module):
	def __init__(self, name=None):
		super().__init__(name)
		self.movesData = self.movesGrad = True


	def updateData(self, data):
		self.data = data


	def updateGrad(self, grad):
		self.grad = grad


	def dataShapeFrom(self, shape):
		return shape


	def gradShapeFrom(self, shape):
		return shape


class Flatten(Module):
	def __init__(self, name=None):
		super().__init__(name)
		self.movesData = self.movesGrad = True
		self.inshape = None


	def updateData(self, data):
		self.inshape = data.shape
		self.data = data.reshape(data.shape[0], int(np.prod(data.shape[1:])))


	def updateGrad(self, grad):
		self.grad = grad.reshape(self.inshape)


	def dataShapeFrom(self, shape):
		return shape[0], int(np.prod(shape[1:]))


	def gradShapeFrom(self, shape):
		return (shape[0], ) + tuple(self.inshape[1:])


class SoftMax(Module):
	def __init__(self, name=None):
		super().__init__(name)
		self.gradUsesOutData = True


	@staticmethod
	def as4d(ary):
		return ary.reshape(ary.shape + (1, ) * max(0, 4 - ary.ndim))


	def updateData(self, data):
		self.data = S().Dnn.softmaxNd(self.as4d(data)).reshape(data.shape)


	def updateGrad(self, grad):
		self.grad = S().Dnn.softmaxNdBackward(self.as4d(self.data), self.as4d(grad)).reshape(grad.shape)


	def dataShapeFrom(self, shape):
		return shape


	def gradShapeFrom(self, shape):
		return shape


class Dropout(Module):
	def __init__(self, p=0.5, rng=None, slicing=None, inplace=False, name=None):
		super().__init__(name)
		self.p, self.rng = p, rng
		self.slice, self.inplace = slicing, inplace
		self.rands, self.partition = None, None


	def target(self, ary):
		gpuarray = S().gpuarray
		if self.inplace:
			return ary
		if self.slice is not None:
			return gpuarray.copy(None, ary)
		return gpuarray.empty(ary.shape, dtype=ary.dtype, allocator=gpuarray.memoryPool)


	def updateData(self, data):
		if not self.train:
			self.data = data
			return

		surf = S()
		self.data = self.target(data)

		self.rands = surf.gpuarray.empty((data.size, ), dtype=np.uint32, allocator=surf.gpuarray.memoryPool)
		(surf.gpuarray.globalRng if self.rng is None else self.rng).fillInteger(self.rands)

		keep = 1.0 - self.p
		self.partition = int(keep * np.iinfo(np.uint32).max)
		surf.ElementWise.dropoutKer(data.dtype)(self.data, data, self.rands, self.partition, keep, slice=self.slice)


	def updateGrad(self, grad):
		if not self.train:
			self.grad = grad
			return

		self.grad = self.target(grad)
		S().ElementWise.dropoutKer(grad.dtype)(
			self.grad, grad, self.rands, self.partition, 1.0 - self.p, slice=self.slice
		)


	def reset(self):
		super().reset()
		self.rands = None


	def dataShapeFrom(self, shape):
		return shape


	def gradShapeFrom(self, shape):
		return shape


# ================================================================================================ containers
class Container(Module):
	def __init__(self, name=None):
		super().__init__(name)
		self.modules = {}


	def append(self, mod, acquire=True):
		if mod.name is None:
			mod.name = str(len(self.modules))

		if mod.name in self.modules:
			if not acquire:
				raise ContainerError("Module with name '%s' is already in container" % mod.name)
			mod.name = str(len(self.modules))

		self.modules[mod.name] = mod
		return self


	def removeModule(self, mod):
		self.modules.pop(mod.name)
		return mod


	def getByName(self, name):
		if name in self.modules:
			return self.modules[name]

		for mod in self.modules.values():
			if isinstance(mod, Container):
				found = mod.getByName(name)
				if found is not None:
					return found

		return None


	def getAllByType(self, typ):
		found = []
		for mod in self.modules.values():
			if isinstance(mod, typ):
				found.append(mod)
			elif isinstance(mod, Container):
				found.extend(mod.getAllByType(typ))
		return found


	def setVar(self, name, var):
		head, _, tail = name.partition(".")
		if not tail:
			raise ContainerError("Cannot find dot-delimiter in variable name: %s" % name)
		self.modules[head].setVar(tail, var)


	def getVar(self, name):
		head, _, tail = name.partition(".")
		if not tail:
			raise ContainerError("Cannot find dot-delimiter in variable name: %s" % name)
		return self.modules[head].getVar(tail)


	def getVarTable(self, vartable=None, name=None, root=True):
		prefix = "" if root else name
		vartable = {} if vartable is None else vartable

		for mod in self.modules.values():
			mod.getVarTable(vartable, "%s%s." % (prefix, mod.name), root=False)

		return vartable


	def zeroGradParams(self):
		for mod in self.modules.values():
			mod.zeroGradParams()


	def updateParams(self, learnRate):
		for mod in self.modules.values():
			mod.updateParams(learnRate)


	def trainMode(self):
		super().trainMode()
		for mod in self.modules.values():
			mod.trainMode()


	def evalMode(self):
		super().evalMode()
		for mod in self.modules.values():
			mod.evalMode()


	def calcMode(self, T):
		for mod in self.modules.values():
			mod.calcMode(T)


	def reset(self):
		super().reset()
		for mod in self.modules.values():
			mod.reset()


	def numOfParams(self):
		return sum(var.data.size for var in self.getVarTable().keys())


	def genericCheckDataType(self, dtype):
		pass


	def __getitem__(self, item):
		if isinstance(item, str):
			return self.modules[item]
		raise NotImplementedError(type(item).__name__)


class Sequential(Container):
	honourUpdGrad = True
	fuseInplaceRelu = True       # backend-internal fusion around in-place ReLUs (see planFusion)
	fuseConvStats = True         # BatchNorm statistics from the preceding convolution's epilogue (see planFusion)
	fuseBnAdd = True             # residual Add normalises its BatchNorm inputs on the fly (see planFusion)
	fuseGateStats = True         # gradient fan-in also sums the next BatchNorm backward's statistics (see planFusion)
	fuseBnBackward = True        # ... and the BatchNorm's apply pass is folded into the backward of the Conv2D in front
	fuseStridedGrad = True       # input gradients of a down-sampling block's stride-2 1x1 convolutions stay compact
	fuseReluMask = True          # Add+ReLU leaves its output's sign mask for the fan-in that would read the output back

	def __init__(self, name=None):
		super().__init__(name)
		self.graph = []


	def planFusion(self):
		"""Marks, for the coming forward/backward pass, which neighbouring modules share a kernel (fusion
		inside the backend, module API untouched). Every pattern has its switch on Sequential and is checked against the
		unfused sequence in tests/test_gpu_nets.py.

		fuseInplaceRelu — around Activation(relu, inplace=True) (Models/Nets/ResNet.py:33,58 with actInplace=True; with
		the in-place flag the pre-activation values are overwritten in the reference too, so no observable buffer changes):
		  BatchNorm2D (train) -> ReLU : BN writes relu(bn(x)); its backward gates the incoming grad with (bn(x) > 0)
		  Add (2 inputs)      -> ReLU : the sum kernel writes relu(a + b)
		  ReLU -> Replicate(2)        : the fan-in kernel writes (g0 + g1) * (y > 0), y = the ReLU's output
		  the ReLU module itself then only forwards data / grad; bit-identical.
		fuseConvStats  — Conv2D -> BatchNorm2D (train): the convolution's epilogue leaves per-strip channel sums, the BN
		                 skips its statistics pass (same mean/variance up to fp32 summation order).
		fuseBnAdd      — Parallel(... Conv2D -> BatchNorm2D, ...) -> Add: those BatchNorms only produce per-channel
		                 coefficients (DeferredBN); the Add kernel normalises while summing; bit-identical.
		fuseGateStats  — [Parallel, Add, ReLU, Replicate]: the fan-in also sums the backward statistics of the Parallel's
		                 branch-tail BatchNorms, which then run their apply pass only; bit-identical.
		fuseBnBackward — Conv2D (no bias) -> BatchNorm2D whose statistics came from the fan-in: the BN hands the
		                 convolution a DeferredBNGrad and the 1x1 convolution's backward kernels evaluate the BN backward
		                 while gathering (equal to the separate pass up to fp32 rounding)."""
		graph, on = self.graph, Sequential.fuseInplaceRelu

		for i, mod in enumerate(graph):
			if isinstance(mod, Activation):
				mod.dataFused = mod.gradFused = False
			elif isinstance(mod, (BatchNorm2D, Add)):
				mod.fusedRelu = False
			elif isinstance(mod, Replicate):
				mod.gateGrad, mod.statsFor, mod.maskFrom = False, [], None
			if isinstance(mod, Add):
				mod.emitMask = False

			if isinstance(mod, Conv2D):
				mod.emitStats = False

			# Parallel -> Add: a branch ending in Conv2D -> BatchNorm2D hands the Add an un-normalised tensor plus
			# per-channel coefficients (the BN's output has no other reader), see BatchNorm2D.deferApply
			if isinstance(mod, Parallel):
				nxt = graph[i + 1] if i + 1 < len(graph) else None
				for branch in mod.graph:
					tail = branch.graph[-2:] if isinstance(branch, Sequential) else []
					if len(tail) == 2 and isinstance(tail[1], BatchNorm2D) and isinstance(tail[0], Conv2D):
						tail[1].deferApply = (
							Sequential.fuseBnAdd and Sequential.fuseConvStats and tail[1].train and isinstance(nxt, Add) and
							len(mod.graph) == 2
						)
			if isinstance(mod, BatchNorm2D):
				prev = graph[i - 1] if i > 0 else None
				mod.statsFrom = None
				if Sequential.fuseConvStats and mod.train and isinstance(prev, Conv2D):
					prev.emitStats, mod.statsFrom = True, prev
				# Conv2D -> BatchNorm2D: the convolution's backward kernels can evaluate the BN backward while gathering its
				# gradient (only taken when the BN's statistics were summed by the fan-in and the conv is eligible)
				mod.foldBackwardInto = prev if (
					Sequential.fuseBnBackward and mod.train and isinstance(prev, Conv2D) and prev.b is None
				) else None

		# first convolutions of the branches behind every Replicate: decided below, off unless decided otherwise
		for i, mod in enumerate(graph):
			if isinstance(mod, Replicate) and i + 1 < len(graph) and isinstance(graph[i + 1], Parallel):
				for branch in graph[i + 1].graph:
					if isinstance(branch, Sequential) and branch.graph and isinstance(branch.graph[0], Conv2D):
						branch.graph[0].compactGrad = False

		if not on:
			return

		gated = []
		for i, mod in enumerate(graph):
			if not (isinstance(mod, Activation) and mod.fusable() and i > 0):
				continue

			prev = graph[i - 1]
			nxt = graph[i + 1] if i + 1 < len(graph) else None

			if isinstance(prev, BatchNorm2D) and prev.train:
				prev.fusedRelu = mod.dataFused = mod.gradFused = True

			elif isinstance(prev, Add):
				prev.fusedRelu = mod.dataFused = True

			if not mod.gradFused and isinstance(nxt, Replicate) and nxt.times == 2:
				nxt.gateGrad = mod.gradFused = True
				gated.append(i + 1)

				# [Parallel, Add, ReLU, Replicate]: the fan-in's output is the gradient of the Parallel's branch-tail
				# BatchNorms (Add passes it through unchanged) -> it also sums their backward statistics
				if Sequential.fuseGateStats and i >= 2 and isinstance(prev, Add) and isinstance(graph[i - 2], Parallel):
					nxt.statsFor = [
						branch.graph[-1] for branch in graph[i - 2].graph
						if isinstance(branch, Sequential) and branch.graph and isinstance(branch.graph[-1], BatchNorm2D)
					][:2]
					# fuseReluMask: the Add's kernel also leaves (y > 0) as one bit per element and the fan-in gates with
					# that instead of reading y back (4 B -> 1/4 B per element; same predicate, bit-identical)
					if Sequential.fuseReluMask and nxt.statsFor and prev.fusedRelu:
						prev.emitMask, nxt.maskFrom = True, prev


		# fuseStridedGrad: [ReLU, Replicate(2), Parallel] whose two branches both start with a stride-2 pointwise convolution
		# (the down-sampling blocks, Models/Nets/ResNet.py:36-46): their input gradients are non-zero on every other pixel
		# of every other row only. They stay compact (a quarter of the tensor, no memset, no strided stores) and the
		# fan-in kernel, which sums them and knows where the zeros are, expands them on the fly; bit-identical.
		for r in gated:
			rep, par = graph[r], graph[r + 1] if r + 1 < len(graph) else None
			if not (Sequential.fuseStridedGrad and rep.statsFor and isinstance(par, Parallel) and len(par.graph) == 2):
				continue
			heads = [b.graph[0] if isinstance(b, Sequential) and b.graph else None for b in par.graph]
			if all(isinstance(h, Conv2D) and h.groups == 1 and
				   S().Dnn.compactGradSupported(h.W, h.stride, h.pad, h.dilation) for h in heads):
				for h in heads:
					h.compactGrad = True


	def append(self, mod, acquire=True):
		super().append(mod, acquire)
		self.graph.append(mod)
		return self


	def extend(self, container, acquire=True):
		for mod in container.graph:
			self.append(mod, acquire)
		return self


	def pop(self):
		mod = self.graph.pop()
		self.removeModule(mod)
		return mod


	def __getitem__(self, item):
		if isinstance(item, int):
			return self.graph[item]
		return super().__getitem__(item)


	def getByIndex(self, index):
		return self.graph[index]


	def optimizeForShape(self, shape, memlimit=None):
		for mod in self.graph:
			mod.optimizeForShape(shape, memlimit)
			shape = mod.dataShapeFrom(shape)


	def updateData(self, data):
		self.planFusion()

		for i, mod in enumerate(self.graph):
			try:
				mod(data)
			except ModuleError as e:
				raise ModuleError("%s:\nData error in module %d (%s):\n%s" % (self, i, mod, e))
			data = mod.data

		self.data = data


	def backward(self, grad, updParamGrads=True, updGrad=True, scale=1.0, momentum=1.0):
		"""Accumulate mode by default (momentum=1.0), like Containers/Sequential.py:212-232. `updGrad=False` — what
		Trainer.handleBatch passes (Handlers/Trainer.py:33) — skips the data gradient of the FIRST module, whose result
		nobody consumes. The reference means to do the same but its branch is dead code (`i < len(self.graph)` is always
		true, Sequential.py:215-218), so it always pays for conv1's backward-data; parameters, loss and every other
		gradient are unaffected. Set Sequential.honourUpdGrad = False for the reference's literal behaviour."""
		last = len(self.graph) - 1

		with backwardScope():
			for i, mod in enumerate(reversed(self.graph)):
				first = i == last and Sequential.honourUpdGrad
				try:
					mod.backward(grad, updParamGrads=updParamGrads, updGrad=updGrad if first else True, scale=scale,
								 momentum=momentum)
				except ModuleError as e:
					raise ModuleError("%s:\nGrad error in module %d (%s):\n%s" % (self, len(self.graph) - 1 - i, mod, e))
				grad = mod.grad

		self.grad = grad


	def dataShapeFrom(self, shape):
		for mod in self.graph:
			shape = mod.dataShapeFrom(shape)
		return shape


	def gradShapeFrom(self, shape):
		for mod in reversed(self.graph):
			shape = mod.gradShapeFrom(shape)
		return shape


class Parallel(Container):
	def __init__(self, name=None):
		super().__init__(name)
		self.graph = []


	def append(self, mod, acquire=True):
		super().append(mod, acquire)
		self.graph.append(mod)
		return self


	def getByIndex(self, index):
		return self.graph[index]


	def updateData(self, data):
		assert len(data) == len(self.graph)
		self.data = []

		for i, mod in enumerate(self.graph):
			try:
				mod(data[i])
			except ModuleError as e:
				raise ModuleError("%s:\nData error in module %d (%s):\n%s" % (self, i, mod, e))
			self.data.append(mod.data)


	def backward(self, grad, updParamGrads=True, updGrad=True, scale=1.0, momentum=1.0):
		assert len(grad) == len(self.graph)
		self.grad = []

		with backwardScope():
			for i, mod in enumerate(self.graph):
				try:
					mod.backward(grad[i], updParamGrads=updParamGrads, updGrad=updGrad, scale=scale, momentum=momentum)
				except ModuleError as e:
					raise ModuleError("%s:\nGrad error in module %d (%s):\n%s" % (self, i, mod, e))
				self.grad.append(mod.grad)


	def dataShapeFrom(self, shapes):
		return [mod.dataShapeFrom(shapes[i]) for i, mod in enumerate(self.graph)]


	def gradShapeFrom(self, shapes):
		return [mod.gradShapeFrom(shapes[i]) for i, mod in enumerate(self.graph)]
