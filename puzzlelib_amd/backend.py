"""
The backend object of the MI355X operator backend — the thing `Backend.getBackend(deviceIdx, initmode, logger)` returns.

It reproduces the object shape PuzzleLib's dispatch surface reads from `PuzzleLib.Hip.Backend`
(Backend/gpuarray.py:60-113, Backend/Blas.py:43-102, Backend/Dnn.py:124-338, Backend/Kernels/*.py; the original is
Hip/Backend.py:19-71 on top of Cuda/GPUBackend.py:17-433): GPUArray, memoryPool, blas, dnn, matmod, costmod, the
`<name>Ker` kernel objects, enums, SharedArray, stream/event managers, RNG, copy/concatenate/split/tile, timeKernel.
Underneath every entry is one or two calls into libpuzzle_mi355.so — no MIOpen, no rocBLAS, no JIT.
"""
import os, weakref, sys, time, ctypes
from ctypes import byref, c_int, c_size_t, c_void_p
from collections import OrderedDict, deque
from types import SimpleNamespace

import numpy as np

from puzzlelib_amd import lib, driver, lazy, fusion, rtc
from puzzlelib_amd.lib import HipError, ConvDesc, PoolDesc
from puzzlelib_amd.driver import streamHandle
from puzzlelib_amd.gpuarray import GPUArray, prod, eltwise, contiguousStrides
from puzzlelib_amd.common import (
	ConvFwdAlgo, ConvBwdFilterAlgo, ConvBwdDataAlgo, PoolMode, SoftMaxMode, BatchNormMode, LRNMode, RNNMode, DirectionMode, RNNAlgo,
	GroupFormat, ConvPerf, toAlgoId, pair, requireF32, rptrOf
)
from puzzlelib_amd.blas import BlasContext
from puzzlelib_amd.dnn import DnnContext, conv3d
from puzzlelib_amd.modules import (
	MatModule, ReductionCallable, CostModule, MemModule, PoolModule, StubModule, PointwiseCost, PReluModule, PadModule,
	UpsampleModule, CTCModule, EmbedModule
)
from puzzlelib_amd.kernels import EltwiseKernel, memoizedKernel, AddKernelFactory, absorbRelu, absorbReluDer, absorbAxpy


# ---------------------------------------------------------------------------------------------- helpers
class SharedArray:
	"""The flat arena the optimizers and the data-parallel exchange work on: tensors are registered by name, `build()` places
	them in ONE allocation at 16-byte aligned offsets in registration order and exposes them as views — `.ary` is the
	whole arena (1-d), `.blocks[name]` / `[name]` a tensor inside it. Contract of Cuda/Utils.py:19-64 (the reference's
	Optimizer.setupOn(useGlobalState=True) and Grid both rely on the order and the alignment)."""
	alignment = 16

	def __init__(self, dtype=np.float32, allocator=None):
		self.dtype, self.allocator = np.dtype(dtype), allocator
		self.ary = None
		self.blocks = OrderedDict()            # name -> GPUArray once built
		self.plan, self.end = [], 0            # [(name, shape, byte offset)] while registering; bytes laid out so far

	@classmethod
	def align(cls, nbytes):
		return -(-nbytes // cls.alignment) * cls.alignment

	def register(self, shape, dtype, name):
		if self.ary is not None:
			raise ValueError("SharedArray is already built")
		if np.dtype(dtype) != self.dtype:
			raise ValueError("SharedArray of %s cannot hold %s %s" % (self.dtype, name, np.dtype(dtype)))
		if any(name == other for other, _, _ in self.plan):
			raise ValueError("%s is registered twice" % name)
		self.plan.append((name, tuple(shape), self.end))
		self.end += self.align(prod(shape) * self.dtype.itemsize)

	def build(self):
		self.ary = GPUArray.empty((self.end // self.dtype.itemsize, ), dtype=self.dtype, allocator=self.allocator)
		for name, shape, offset in self.plan:
			nbytes = prod(shape) * self.dtype.itemsize
			self.blocks[name] = GPUArray(shape, self.dtype, gpudata=self.ary.gpudata[offset:offset + nbytes])
		# the allocation knows it is an arena: the data-parallel exchange plans its buckets from the blocks (grid.ArenaWatcher)
		lazy.stateOf(self.ary.gpudata.root).arena = [
			(name, offset, prod(shape) * self.dtype.itemsize) for name, shape, offset in self.plan
		]

	def __getitem__(self, name):
		return self.blocks[name]


class QueueManager:
	"""Pool of reusable driver objects (streams, events): `borrow(n)` hands out n of them, creating what the pool lacks;
	`give(items)` takes them back; `reserve(n)` creates ahead of need. The streamManager / eventManager attributes of the
	backend object (contract: Cuda/Utils.py:67-94; Optimizers/Optimizer.py:152-190 borrows one stream per parameter)."""

	def __init__(self, objtype):
		self.objtype = objtype
		self.free = deque()

	def reserve(self, nitems):
		for _ in range(nitems):
			self.free.append(self.objtype())

	def borrow(self, nitems):
		return [self.free.pop() if self.free else self.objtype() for _ in range(nitems)]

	def give(self, items):
		self.free.extend(items)

	def clear(self):
		self.free.clear()


class RandomNumberGenerator:
	"""fillInteger / fillUniform / fillNormal — Cuda/Source/Libs/CuRand.c:231-234 (Philox here, XORWOW there)."""

	def __init__(self, type=None, seed=0):
		self.type, self.seed = "philox4x32-10", int(seed) & 0xffffffffffffffff
		handle = c_void_p()
		lib.pz_rng_create(self.seed, byref(handle))
		self.handle = handle.value


	def fillInteger(self, data):
		assert data.contiguous and data.dtype.itemsize == 4
		lib.pz_rng_fill_u32(self.handle, data.optr, data.size, None)


	def fillUniform(self, data):
		requireF32(data)
		lib.pz_rng_fill_uniform(self.handle, data.optr, data.size, None)


	def fillNormal(self, data, mean=0.0, stddev=1.0):
		requireF32(data)
		lib.pz_rng_fill_normal(self.handle, data.optr, data.size, mean, stddev, None)


	def __del__(self):
		handle, self.handle = getattr(self, "handle", None), None
		if handle is not None:
			try:
				lib.pz_rng_destroy(handle)
			except Exception:
				pass


# ---------------------------------------------------------------------------------------------- the backend object
class Mi355Backend:
	BackendName = "Hip"

	warpSize = 64
	nthreads = 256

	Driver = driver
	GPUArray = GPUArray
	Error = HipError
	SharedArray = SharedArray

	# the reference backend's `Rand` module attribute (Cuda/GPUBackend.py:33,62-63; Hip/Backend.py:29): generator class + type id
	Rand = SimpleNamespace(__name__="puzzlelib_amd.rng", RandomNumberGenerator=RandomNumberGenerator, RAND_RNG_PSEUDO_PHILOX4_32_10=0)
	RandomNumberGenerator = RandomNumberGenerator

	# kernels a caller defines at run time (Cuda/GPUBackend.py:27-30; Cuda/SourceModule.py): hiprtc for gfx950, puzzlelib_amd/rtc.py
	SourceModule, ElementwiseKernel, ElementHalf2Kernel, ReductionKernel = rtc.SourceModule, rtc.ElementwiseKernel, rtc.ElementHalf2Kernel, rtc.ReductionKernel

	GroupFormat = GroupFormat
	ConvPerf = ConvPerf
	ConvFwdAlgo, ConvBwdDataAlgo, ConvBwdFilterAlgo = ConvFwdAlgo, ConvBwdDataAlgo, ConvBwdFilterAlgo
	PoolMode, SoftMaxMode, BatchNormMode, LRNMode = PoolMode, SoftMaxMode, BatchNormMode, LRNMode
	RNNAlgo, RNNMode, DirectionMode = RNNAlgo, RNNMode, DirectionMode


	def __init__(self, deviceIdx, initmode=0, logger=None):
		self.deviceIdx = deviceIdx

		ndevices = driver.Device.count()
		if ndevices == 0:
			raise HipError("No %s enabled device found" % self.BackendName)
		if deviceIdx >= ndevices:
			raise HipError("Invalid %s config device index" % self.BackendName)

		self.device = driver.Device(deviceIdx).set()

		if logger is not None:
			logger.info("Using device #%s (%s, %s)", deviceIdx, self.device.name(), self.device.arch())

		self.memoryPool = driver.MemoryPool()
		GPUArray.defaultAllocator = self.memoryPool

		seed = int(np.random.randint(sys.maxsize, dtype=np.intp))
		self.globalRng = RandomNumberGenerator(seed=seed)

		self.streamManager = QueueManager(objtype=driver.Stream)
		self.eventManager = QueueManager(objtype=driver.Event)

		self.blas, self.dnn = None, None
		self.costmod, self.matmod = None, None
		self.ctcmod = self.embedmod = self.padmod = self.poolmod = self.prelumod = self.upsamplemod = self.memmod = None
		self.getAccuracyKernel = None

		self.initmode = 0
		self.updateBackend(initmode, logger=logger)


	def updateBackend(self, initmode, logger=None):
		"""Raises the initialisation level (Backend/__init__ levels: 0 = arrays only, 1 = + blas / dnn contexts, 2 = + kernel
		objects and modules); a level once reached stays."""
		for level, bringUp in ((1, lambda: self.initLibs(logger)), (2, self.initKernels)):
			if self.initmode < level <= initmode:
				bringUp()
		self.initmode = max(self.initmode, initmode)


	def initLibs(self, logger=None):
		if self.blas is None:
			self.blas, self.dnn = BlasContext(self), DnnContext(self)
		if logger is not None:
			logger.debug("Created blas/dnn contexts (%s; %s)", self.blas.getVersion(), self.dnn.getVersion())


	def initKernels(self):
		if self.dnn is None:
			self.initLibs()

		self.matmod = MatModule(self)
		self.costmod = CostModule(self)
		self.getAccuracyKernel = self.costmod.getAccuracyKernel

		self.memmod = MemModule(self)
		self.poolmod = PoolModule(self)
		self.embedmod, self.padmod = EmbedModule(self), PadModule(self)
		self.prelumod, self.upsamplemod = PReluModule(self.matmod), UpsampleModule(self)
		self.ctcmod = CTCModule(self)

		K = memoizedKernel
		self.sigmoidKer = K(lib.OP_SIGMOID, 2, 0, "sigmoidKer")
		self.sigmoidDerKer = K(lib.OP_SIGMOID_DER, 3, 0, "sigmoidDerKer")
		self.tanhKer = K(lib.OP_TANH, 2, 0, "tanhKer")
		self.tanhDerKer = K(lib.OP_TANH_DER, 3, 0, "tanhDerKer")
		self.reluKer = K(lib.OP_RELU, 2, 0, "reluKer")
		self.reluDerKer = K(lib.OP_RELU_DER, 3, 0, "reluDerKer")
		self.leakyReluKer = K(lib.OP_LEAKY_RELU, 2, 1, "leakyReluKer")
		self.leakyReluDerKer = K(lib.OP_LEAKY_RELU_DER, 3, 1, "leakyReluDerKer")
		self.eluKer = K(lib.OP_ELU, 2, 1, "eluKer")
		self.eluDerKer = K(lib.OP_ELU_DER, 3, 1, "eluDerKer")
		self.softPlusKer = K(lib.OP_SOFTPLUS, 2, 0, "softPlusKer")
		self.softPlusDerKer = K(lib.OP_SOFTPLUS_DER, 3, 0, "softPlusDerKer")
		self.clipKer = K(lib.OP_CLIP, 2, 2, "clipKer")
		self.clipDerKer = K(lib.OP_CLIP_DER, 3, 2, "clipDerKer")
		self.geluKer = K(lib.OP_GELU, 2, 0, "geluKer")
		self.geluDerKer = K(lib.OP_GELU_DER, 3, 0, "geluDerKer")

		self.dropoutKer = K(lib.OP_DROPOUT, 3, 2, "dropoutKer", rawScalar=(0, ))
		self.dropout2dKer = K(lib.OP_DROPOUT2D, 3, 3, "dropout2dKer", rawScalar=(0, 2))
		self.toVectorAddVectorKer = K(lib.OP_AXPY, 2, 1, "toVectorAddVectorKer")

		self.classicMomSGDKer = K(lib.OP_CLASSIC_MOM_SGD, 3, 2, "classicMomSGDKer")
		self.nesterovMomSGDKer = K(lib.OP_NESTEROV_MOM_SGD, 3, 2, "nesterovMomSGDKer")
		self.rmspropKer = K(lib.OP_RMSPROP, 3, 3, "rmspropKer")
		self.adamKer = K(lib.OP_ADAM, 4, 4, "adamKer")
		self.rmspropGravesKer = K(lib.OP_RMSPROP_GRAVES, 5, 4, "rmspropGravesKer")
		self.adagradKer = K(lib.OP_ADAGRAD, 3, 2, "adagradKer")
		self.adadeltaKer = K(lib.OP_ADADELTA, 4, 2, "adadeltaKer")
		self.smorms3Ker = K(lib.OP_SMORMS3, 5, 2, "smorms3Ker")

		self.linearKer = K(lib.OP_LINEAR, 2, 2, "linearKer")
		self.mulKer = K(lib.OP_MUL, 3, 0, "mulKer")
		self.addKer = AddKernelFactory()

		# direct callables (not factories) in the reference: Cuda/GPUBackend.py:169-172,194-195,207,211-215
		self.rbmKer = EltwiseKernel(lib.OP_RBM, 3, 0, "rbmKer")
		self.absKer = EltwiseKernel(lib.OP_ABS, 2, 0, "absKer")
		self.weightDecayKer = EltwiseKernel(lib.OP_WEIGHT_DECAY, 2, 1, "weightDecayKer")
		self.l1penaltyKer = EltwiseKernel(lib.OP_L1_PENALTY, 3, 1, "l1penaltyKer")
		self.l1gradKer = EltwiseKernel(lib.OP_L1_GRAD, 3, 1, "l1gradKer")

		# castFP16toFP32(outdata, indata) / castFP32toFP16(outdata, indata) — Cuda/Kernels/ElementWise.py:1143-1156, read by
		# Modules/Cast.py:69-70. fp16 is a storage type here: the casts exist, no operator computes in it
		def castKernel(name, entry, src, dst):
			def cast(outdata, indata, slice=None, stream=None):
				if slice is not None or outdata.dtype != dst or indata.dtype != src or outdata.size != indata.size or \
						not (outdata.contiguous and indata.contiguous):
					raise ValueError("%s(outdata %s, indata %s): contiguous %s -> %s arrays of one size, no slice" % (
						name, outdata.dtype, indata.dtype, np.dtype(src).name, np.dtype(dst).name))
				entry(outdata.optr, indata.rptr, indata.size, None if stream is None else stream.handle)
			cast.__name__ = name
			return cast
		self.castFP16toFP32 = castKernel("castFP16toFP32", lib.pz_cast_f16_f32, np.float16, np.float32)
		self.castFP32toFP16 = castKernel("castFP32toFP16", lib.pz_cast_f32_f16, np.float32, np.float16)
		# point-wise cost kernels with the reference's positional arguments (Cuda/Kernels/Costs.py:8-72)
		self.bceKer = PointwiseCost(lib.COST_BCE, "bceKer")
		self.hingeKer = PointwiseCost(lib.COST_HINGE, "hingeKer")
		self.smoothL1Ker = PointwiseCost(lib.COST_SMOOTH_L1, "smoothL1Ker")
		self.l1HingeKer = PointwiseCost(lib.COST_L1_HINGE, "l1HingeKer")

		# fused residual sum / gradient fan-in (one 12 B/elem pass instead of memset + 2 axpy)
		self.add3Ker = EltwiseKernel(lib.OP_ADD3, 3, 0, "add3Ker")
		self.add3ReluKer = EltwiseKernel(lib.OP_ADD3_RELU, 3, 0, "add3ReluKer")
		self.add3GateKer = EltwiseKernel(lib.OP_ADD3_GATE, 4, 0, "add3GateKer")


	@staticmethod
	def dtypesSupported():
		return [(np.float32, 1e-5)]


	@staticmethod
	def copy(dest, source, allocator=None):
		if dest is None:
			return source.copy(allocator=allocator)
		dest.set(source)
		return dest


	def fillUniform(self, data, minval=0.0, maxval=1.0, rng=None):
		assert data.dtype == np.float32
		rng = self.globalRng if rng is None else rng
		rng.fillUniform(data)
		self.linearKer(data.dtype)(data, data, maxval - minval, minval)


	def fillNormal(self, data, mean=0.0, stddev=1.0, rng=None):
		rng = self.globalRng if rng is None else rng
		rng.fillNormal(data, mean=mean, stddev=stddev)


	@staticmethod
	def axisBand(ary, axis, start, stop):
		"""view of `ary` restricted to [start, stop) along `axis`"""
		return ary[(slice(None), ) * axis + (slice(start, stop), )]


	@staticmethod
	def copyBand(dst, src, axis):
		"""dst <- src, same shape, either side a band along `axis` of a dense tensor. 4-byte element types: one launch of the
		strided-copy kernel. Any other element size (float16, int8 / uint8 masks, int64, float64): one pitched copy, as the
		reference does for every type (Cuda/GPUBackend.py:296,320) — rows = the axes in front of `axis`, row pitch = their stride"""
		if dst.dtype.itemsize == 4:
			dst.stridedCopyFrom(src)
			return
		inner = dst.dtype.itemsize
		for d in range(axis + 1, dst.ndim):
			inner *= dst.shape[d]
		height = 1
		for d in range(axis):
			height *= dst.shape[d]
		width = dst.shape[axis] * inner
		if width == 0 or height == 0:
			return
		pitches = []
		for ary in (dst, src):
			st = ary.strides
			dense = st[-1] == ary.dtype.itemsize and all(st[d] == st[d + 1] * ary.shape[d + 1] for d in range(axis, ary.ndim - 1))
			rows = all(st[d] == st[d + 1] * ary.shape[d + 1] for d in range(axis - 1))
			if not (dense and rows):
				raise NotImplementedError("concatenate / split of a %s view with strides %s" % (ary.dtype, st))
			pitches.append(st[axis - 1] if axis > 0 else width)
		lib.pz_memcpy_2d(dst.wptr, pitches[0], src.rptr, pitches[1], width, height, None)


	def concatenate(self, tup, axis, out=None, allocator=None):
		"""np.concatenate along `axis` (Backend/gpuarray.py; contract of Cuda/GPUBackend.py:275-300): every input lands in its
		band of the output with one strided copy"""
		first = tup[0]
		total = 0
		for a in tup:
			if a.dtype != first.dtype or a.ndim != first.ndim or any(
				a.shape[d] != first.shape[d] for d in range(first.ndim) if d != axis
			):
				raise ValueError("concatenate: %s %s does not match %s %s off axis %d" % (a.shape, a.dtype, first.shape, first.dtype, axis))
			total += a.shape[axis]

		shape = first.shape[:axis] + (total, ) + first.shape[axis + 1:]
		if out is None:
			out = GPUArray.empty(shape, dtype=first.dtype, allocator=allocator)
		elif out.shape != shape or out.dtype != first.dtype:
			raise ValueError("concatenate: output is %s %s, expected %s %s" % (out.shape, out.dtype, shape, first.dtype))

		at = 0
		for a in tup:
			self.copyBand(self.axisBand(out, axis, at, at + a.shape[axis]), a, axis)
			at += a.shape[axis]
		return out


	def split(self, ary, sections, axis, allocator=None):
		"""inverse of concatenate: `sections` = sizes along `axis` (Cuda/GPUBackend.py:303-325's contract)"""
		if sum(sections) != ary.shape[axis]:
			raise ValueError("split: sections %s do not add up to %d" % (list(sections), ary.shape[axis]))
		outs, at = [], 0
		for size in sections:
			piece = GPUArray.empty(ary.shape[:axis] + (size, ) + ary.shape[axis + 1:], dtype=ary.dtype, allocator=allocator)
			self.copyBand(piece, self.axisBand(ary, axis, at, at + size), axis)
			outs.append(piece)
			at += size
		return outs


	def tile(self, ary, repeats, axis, allocator=None):
		"""`repeats` copies of `ary` side by side along `axis`"""
		return self.concatenate((ary, ) * repeats, axis, allocator=allocator)


	def timeKernel(self, func, args, kwargs=None, looplength=1000, log=True, logname=None, normalize=False,
				   hotpass=True):
		"""Event-pair timing of `looplength` back-to-back calls — Cuda/GPUBackend.py:332-368."""
		kwargs = {} if kwargs is None else kwargs
		if hotpass:
			func(*args, **kwargs)

		start, end = driver.Event(), driver.Event()

		hostStart = time.time()
		start.record()
		for _ in range(looplength):
			func(*args, **kwargs)
		end.record()
		hostEnd = time.time()

		end.synchronize()
		devsecs, hostsecs = start.timeTill(end) * 1e-3, hostEnd - hostStart

		if normalize:
			devsecs /= looplength
			hostsecs /= looplength

		if log:
			logname = getattr(func, "__name__", func.__class__.__name__) if logname is None else logname
			print("%s device time: %s secs" % (logname, devsecs))
			print("%s host time: %s secs" % (logname, hostsecs))

		return devsecs, hostsecs


	def convNdbenchmark(self, datashape, Wshape, dtype, stride=1, pad=0, dilation=1, groups=1, algoCount=10):
		results = self.dnn.convNdbenchmark(datashape, Wshape, dtype, stride, pad, dilation, groups, algoCount)
		return tuple(
			[ConvPerf(algotype(values[0]), *values[1:]) for values in sub] for algotype, sub in
			zip((ConvFwdAlgo, ConvBwdDataAlgo, ConvBwdFilterAlgo), results)
		)


	def instanceNorm2d(self, data, scale, bias, epsilon=1e-5, out=None, allocator=None):
		"""Cuda/GPUBackend.py:381-400: instance normalisation is batch normalisation of the (1, n*c, h, w) view with the
		affine pair tiled over the batch; returns (out, savemean, saveinvvar, tiled scale)."""
		n, c, h, w = data.shape
		ext = n * c
		mean = GPUArray.empty((ext, ), dtype=np.float32, allocator=allocator)
		var = GPUArray.empty((ext, ), dtype=np.float32, allocator=allocator)
		mean.fill(0.0)
		var.fill(1.0)         # (running statistics nobody reads: the reference passes them uninitialised)
		if n > 1:
			scale, bias = self.tile(scale, n, axis=0, allocator=allocator), self.tile(bias, n, axis=0, allocator=allocator)
		outdata, savemean, saveinvvar = self.dnn.batchNormNd(
			data.reshape(1, ext, h, w), mean, var, scale, bias, epsilon, test=False,
			out=None if out is None else out.reshape(1, ext, h, w), allocator=allocator
		)
		return outdata.reshape(data.shape), savemean, saveinvvar, scale


	def instanceNorm2dBackward(self, grad, data, extscale, savemean, saveinvvar, epsilon, affine=True, out=None,
							   allocator=None):
		"""Cuda/GPUBackend.py:403-420"""
		n, c, h, w = grad.shape
		ext = n * c
		ingrad, scalegrad, bgrad = self.dnn.batchNormNdBackward(
			grad.reshape(1, ext, h, w), data.reshape(1, ext, h, w), extscale, savemean, saveinvvar, epsilon,
			out=None if out is None else out.reshape(1, ext, h, w), allocator=allocator
		)
		if affine and n > 1:
			scalegrad = self.matmod.matsum(scalegrad.reshape(n, -1), axis=0, allocator=allocator)
			bgrad = self.matmod.matsum(bgrad.reshape(n, -1), axis=0, allocator=allocator)
		return (ingrad.reshape(grad.shape), scalegrad, bgrad) if affine else ingrad.reshape(grad.shape)


	def createRnn(self, *args, **kwargs):
		raise NotImplementedError("RNNs are outside the implemented operator path")


	acquireRnnParams = updateRnnParams = createRnn


	@staticmethod
	def deviceSupportsBatchHint():
		return False


backendCache = {}


def getDeviceCount():
	return driver.Device.count()


def getBackend(deviceIdx=0, initmode=0, logger=None):
	bnd = backendCache.get(deviceIdx, None)

	if bnd is None:
		bnd = Mi355Backend(deviceIdx, initmode, logger=logger)
		backendCache[deviceIdx] = bnd
	else:
		bnd.updateBackend(initmode, logger=logger)

	return bnd
