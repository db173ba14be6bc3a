"""
The backend object of the MI355X operator backend — the thing `Backend.getBackend(deviceIdx, initmode, logger)` returns.

It reproduces the object shape PuzzleLib's dispatch surface reads from `PuzzleLib.Hip.Backend`
(Backend/gpuarray.py:60-113, Backend/Blas.py:43-102, Backend/Dnn.py:124-338, Backend/Kernels/*.py; the original is
Hip/Backend.py:19-71 on top of Cuda/GPUBackend.py:17-433): GPUArray, memoryPool, blas, dnn, matmod, costmod, the
`<name>Ker` kernel objects, enums, SharedArray, stream/event managers, RNG, copy/concatenate/split/tile, timeKernel.
Underneath every entry is one or two calls into libpuzzle_mi355.so — no MIOpen, no rocBLAS, no JIT.
"""
import os, sys, time, ctypes
from ctypes import byref, c_int, c_size_t, c_void_p
from enum import Enum
from collections import OrderedDict

import numpy as np

from puzzlelib_amd import lib, driver
from puzzlelib_amd.lib import HipError, ConvDesc, PoolDesc
from puzzlelib_amd.driver import streamHandle
from puzzlelib_amd.gpuarray import GPUArray, prod, eltwise


# ---------------------------------------------------------------------------------------------- enums
class ConvFwdAlgo(Enum):              # Hip/Wrappers/MIOpen.py:24-31 (values are this library's algo ids)
	auto = -1
	gemm = 0
	direct = 1
	fft = 2
	winograd = 3
	implicitGemm = 5
	staticGemm = 4


class ConvBwdFilterAlgo(Enum):        # Hip/Wrappers/MIOpen.py:34-39
	auto = -1
	gemm = 0
	direct = 1
	winograd = 3
	implicitGemm = 5


class ConvBwdDataAlgo(Enum):          # Hip/Wrappers/MIOpen.py:42-49
	auto = -1
	gemm = 0
	direct = 1
	fft = 2
	winograd = 3
	transposeGemm = 4
	implicitGemm = 5


class PoolMode(Enum):                 # Hip/Wrappers/MIOpen.py:52-55
	max = 0
	avgWithPad = 1
	avgNoPad = 2


class SoftMaxMode(Enum):              # Hip/Wrappers/MIOpen.py:64-66
	perActivation = 0
	spatial = 1


class BatchNormMode(Enum):            # Hip/Wrappers/MIOpen.py:69-71
	perActivation = 0
	spatial = 1


class LRNMode(Enum):
	map = 0
	cross = 1


class RNNMode(Enum):
	relu = 0
	tanh = 1
	lstm = 2
	gru = 3


class DirectionMode(Enum):
	uni = 0
	bi = 1


class RNNAlgo(Enum):
	default = 0


class GroupFormat(Enum):              # Hip/Backend.py:39-41
	gbp = 0
	bgp = 1


class ConvPerf:                        # Hip/Wrappers/MIOpen.py:82-100
	def __init__(self, algo, time, memory):
		self.algo, self.time, self.memory = algo, time, memory

	def toString(self):
		return "%-40s %-25s %-28s" % (
			"Algo %s" % self.algo, "time %.6f secs" % self.time, "memory %.6f mbytes" % (self.memory / 1024**2)
		)

	__str__ = __repr__ = toString


def toAlgoId(algo):
	"""The reference's algo ids: `direct` is the one-thread-per-output kernel, `winograd` asks for F(2x2, 3x3) where it
	applies (3x3 stride-1 forward / backward-data), `implicitGemm` pins the MFMA implicit GEMM, every other id leaves the
	choice to the library."""
	algo = algo.value if isinstance(algo, Enum) else algo
	return {1: lib.CONV_ALGO_DIRECT, 3: lib.CONV_ALGO_WINOGRAD, 5: lib.CONV_ALGO_IMPLICIT_GEMM}.get(algo, lib.CONV_ALGO_AUTO)


def pair(v):
	return (int(v), int(v)) if isinstance(v, (int, np.integer)) else tuple(int(a) for a in v)


class DeferredBN:
	"""A batch-normalised tensor that was not written: the un-normalised input plus per-channel {a, b} of y = a*x + b
	(pz_bn_fwd_train_defer). Only DnnContext.bnApplyAdd consumes it; `materialize()` writes it out for anyone else."""
	__slots__ = ["tensor", "coef", "dnn"]

	def __init__(self, tensor, coef, dnn):
		self.tensor, self.coef, self.dnn = tensor, coef, dnn

	@property
	def shape(self):
		return self.tensor.shape

	@property
	def dtype(self):
		return self.tensor.dtype

	def materialize(self, allocator=None):
		return self.dnn.bnApplyAdd(self, None, relu=False, allocator=allocator)

	def get(self, stream=None):
		"""the normalised values on the host (what BatchNorm2D.data.get() gives when nothing is deferred)"""
		return self.materialize().get(stream)


class DeferredBNGrad:
	"""The input gradient of a BatchNorm that was not written: dx = A*grad + B*data + C per channel (pz_bn_bwd_coef).
	DnnContext.convNdBackwardData / convNdBackwardParams of the convolution in front evaluate it while gathering;
	`materialize()` runs the apply pass for anyone else."""
	__slots__ = ["grad", "data", "coef", "apply", "dense"]

	def __init__(self, grad, data, coef, apply):
		self.grad, self.data, self.coef, self.apply, self.dense = grad, data, coef, apply, None

	@property
	def shape(self):
		return self.grad.shape

	@property
	def dtype(self):
		return self.grad.dtype

	def materialize(self):
		if self.dense is None:
			self.dense = self.apply()
		return self.dense

	def get(self, stream=None):
		return self.materialize().get(stream)


class StridedGrad:
	"""The input gradient of a stride-2 pointwise convolution that was not zero-filled: `compact` holds the values of the
	pixels (2i, 2j) — (n, c, ceil(h/2), ceil(w/2)) — every other pixel of the (n, c, h, w) gradient is zero. Produced by
	DnnContext.convNdBackwardData(compact=True), consumed by DnnContext.bnGateStats; `materialize()` zero-fills for
	anyone else."""
	__slots__ = ["compact", "shape", "dense"]

	def __init__(self, compact, shape):
		self.compact, self.shape, self.dense = compact, tuple(shape), None

	@property
	def dtype(self):
		return self.compact.dtype

	@property
	def ndim(self):
		return len(self.shape)

	def materialize(self):
		if self.dense is None:
			self.dense = GPUArray.zeros(self.shape, dtype=self.compact.dtype)
			self.dense[:, :, ::2, ::2].set(self.compact)
		return self.dense

	def get(self, stream=None):
		return self.materialize().get(stream)


class ReluMask:
	"""(y > 0) of a fused ReLU's output `tensor`, one bit per element (pz_bn_apply_add_mask); valid for exactly that tensor
	object. Lets the gradient fan-in gate without reading y back."""
	__slots__ = ["tensor", "bits"]

	def __init__(self, tensor, bits):
		self.tensor, self.bits = tensor, bits


class ConvStats:
	"""Per-strip channel sums of a convolution output (pz_conv2d_fwd_stats), valid for exactly that tensor object."""
	__slots__ = ["tensor", "stats"]

	def __init__(self, tensor, stats):
		self.tensor, self.stats = tensor, stats


def requireF32(*arrays):
	for ary in arrays:
		if ary is None:
			continue
		if ary.dtype != np.float32:
			raise ValueError("float32 gpuarray expected, got %s" % ary.dtype)
		if not ary.contiguous:
			raise ValueError("gpuarray is not contiguous")


def ptrOf(ary):
	return None if ary is None else ary.ptr


# ---------------------------------------------------------------------------------------------- BLAS
class BlasContext:
	"""gemm / dot / l1norm / l2norm — BlasContext of Cuda/Source/Libs/CuBlas.c:486-499 (RocBlas on HIP)."""

	def __init__(self, backend):
		self.backend = backend


	def enableTensorOps(self, _):
		return self


	@staticmethod
	def getVersion():
		return "puzzle-mi355 mfma-f32 gemm %d" % lib.pz_version()


	def gemm(self, A, B, out=None, transpA=False, transpB=False, alpha=1.0, beta=0.0, allocator=None):
		requireF32(A, B, out)
		if A.ndim != 2 or B.ndim != 2:
			raise ValueError("gemm operands must be matrices")
		if transpA and transpB:
			raise ValueError("gemm with both operands transposed is not supported")

		m, k = (A.shape[1], A.shape[0]) if transpA else A.shape
		kb, n = (B.shape[1], B.shape[0]) if transpB else B.shape
		if k != kb:
			raise ValueError("gemm inner dimensions do not match (%d vs %d)" % (k, kb))

		if out is None:
			out = GPUArray.empty((m, n), dtype=A.dtype, allocator=allocator)
		elif out.shape != (m, n):
			raise ValueError("gemm output has shape %s, expected %s" % (out.shape, (m, n)))

		lib.pz_gemm(
			int(transpA), int(transpB), m, n, k, alpha, A.ptr, A.shape[1], B.ptr, B.shape[1], beta, out.ptr, n, None
		)
		return out


	def gemmBatched(self, *args, **kwargs):
		raise NotImplementedError("batched GEMM (GroupLinear) is outside the ResNet/NiN/LeNet operator path")


	def scalarOut(self):
		return GPUArray.empty((), dtype=np.float32, allocator=self.backend.memoryPool)


	def dot(self, x, y):
		requireF32(x, y)
		out = self.scalarOut()
		lib.pz_dot(x.ptr, y.ptr, x.size, out.ptr, None)
		return float(out.get())


	def l1norm(self, x):
		requireF32(x)
		out = self.scalarOut()
		lib.pz_asum(x.ptr, x.size, out.ptr, None)
		return float(out.get())


	def l2norm(self, x):
		return float(np.sqrt(self.dot(x, x)))


# ---------------------------------------------------------------------------------------------- DNN
class DnnContext:
	"""conv / pool / softmax / batch-norm entry points with the signatures of Hip/Wrappers/MIOpen.py:333-751."""

	def __init__(self, backend):
		self.backend = backend


	def enableTensorOps(self, _):
		return self


	@staticmethod
	def getVersion():
		return "puzzle-mi355 implicit-gemm conv %d" % lib.pz_version()


	@staticmethod
	def convDesc(dataShape, Wshape, stride, pad, dilation, groups):
		if len(dataShape) != 4 or len(Wshape) != 4:
			raise NotImplementedError("only 2-D convolution (4-d tensors) is implemented on this backend")

		(sh, sw), (ph, pw), (dh, dw) = pair(stride), pair(pad), pair(dilation)
		n, c, h, w = dataShape
		k, _, r, s = Wshape
		return ConvDesc(n, c, h, w, k, r, s, sh, sw, ph, pw, dh, dw, groups)


	def workspace(self, nbytes, allocator):
		if nbytes == 0:
			return None
		return GPUArray.empty((nbytes, ), dtype=np.uint8, allocator=allocator)


	# ---- filter gradients on a side stream. Backward-data and backward-filter of a layer read the same incoming gradient
	# and nothing of each other: with overlapFilterGrad the filter-gradient launches (pack / main kernel / slab reduce) go
	# to a second stream behind an event, so that the two chains fill each other's tails and tiny launches. Every tensor
	# those launches touch is kept referenced until joinFilterGrads() — the pool must not hand its memory to the main
	# stream while the side stream still uses it.
	# Only inside a module-driven backward pass (beginBackward / endBackward, nn.backwardScope): its end joins the streams,
	# so nobody sees a half-finished gradient; direct calls of convNdBackwardParams stay on the main stream.
	overlapFilterGrad = os.environ.get("PUZZLE_MI355_OVERLAP_WGRAD", "1") == "1"
	backwardDepth = 0

	def beginBackward(self):
		self.backwardDepth += 1


	def endBackward(self):
		self.backwardDepth -= 1
		if self.backwardDepth == 0:
			self.joinFilterGrads()


	def filterGradStream(self):
		if not DnnContext.overlapFilterGrad or self.backwardDepth == 0:
			return None
		if getattr(self, "sideStream", None) is None:
			self.sideStream, self.sideRefs, self.sideLaunches = driver.Stream(), [], 0
		self.sideLaunches += 1
		ready = driver.Event()
		ready.record(None)                          # everything issued so far on the main stream (the incoming gradient)
		self.sideStream.waitEvent(ready)
		self.sideRefs.append(ready)
		return self.sideStream


	def filterGradEvent(self):
		"""An event behind everything queued on the side stream so far (None when it is idle or off): what a consumer of
		freshly accumulated filter gradients on another stream has to wait for besides the main stream."""
		if getattr(self, "sideStream", None) is None or not self.sideRefs:
			return None
		event = driver.Event()
		event.record(self.sideStream)
		return event


	def joinFilterGrads(self):
		"""Main stream waits for the side stream; the references held for it are dropped."""
		if getattr(self, "sideStream", None) is None or not self.sideRefs:
			return
		done = driver.Event()
		done.record(self.sideStream)
		lib.pz_stream_wait_event(None, done.handle)
		self.sideRefs = [done]                       # (the event itself must outlive the wait it was queued for)


	def convNd(self, data, W, bias=None, stride=1, pad=0, dilation=1, groups=1, algo=ConvFwdAlgo.auto.value,
			   out=None, allocator=None, withStats=False):
		"""`withStats` (backend-internal): also return the per-strip channel sums of the output for a BatchNorm that
		reads it next -> (out, ConvStats | None); see batchNormNd(convStats=)."""
		assert data.ndim == W.ndim and data.shape[1] == W.shape[1] * groups
		requireF32(data, W, bias, out)

		desc = self.convDesc(data.shape, W.shape, stride, pad, dilation, groups)
		p, q = c_int(0), c_int(0)
		lib.pz_conv2d_out_shape(byref(desc), byref(p), byref(q))
		outshape = (data.shape[0], W.shape[0], p.value, q.value)

		out = GPUArray.empty(outshape, dtype=data.dtype, allocator=allocator) if out is None else out
		if out.shape != outshape:
			raise ValueError("conv output has shape %s, expected %s" % (out.shape, outshape))

		algo = toAlgoId(algo)
		size = c_size_t(0)
		lib.pz_conv2d_workspace_bytes(byref(desc), lib.CONV_FWD, algo, byref(size))
		ws = self.workspace(size.value, allocator)

		if not withStats:
			lib.pz_conv2d_fwd(byref(desc), data.ptr, W.ptr, ptrOf(bias), out.ptr, algo, ptrOf(ws), size.value, None)
			return out

		strips = c_int(0)
		lib.pz_conv2d_fwd_stats_strips(byref(desc), algo, byref(strips))
		if strips.value == 0:
			lib.pz_conv2d_fwd(byref(desc), data.ptr, W.ptr, ptrOf(bias), out.ptr, algo, ptrOf(ws), size.value, None)
			return out, None

		stats = GPUArray.empty((W.shape[0], strips.value, 4), dtype=np.float32, allocator=allocator)
		lib.pz_conv2d_fwd_stats(
			byref(desc), data.ptr, W.ptr, ptrOf(bias), out.ptr, stats.ptr, algo, ptrOf(ws), size.value, None
		)
		return out, ConvStats(out, stats)


	def convAlgoUsed(self, desc, which, algo):
		"""The kernel family (`direct` / `winograd` / `implicitGemm` id) a request resolves to."""
		used = c_int(0)
		lib.pz_conv2d_algo_used(byref(desc), which, toAlgoId(algo), byref(used))
		return used.value


	def bnFoldSupported(self, desc, algo):
		flag = c_int(0)
		lib.pz_conv2d_bn_fold_supported(byref(desc), algo, byref(flag))
		return bool(flag.value)


	@staticmethod
	def compactGradSupported(W, stride, pad, dilation):
		"""Stride-2 pointwise convolution without padding: its backward-data is a stride-1 problem on the output grid."""
		return tuple(W.shape[2:]) == (1, 1) and pair(stride) == (2, 2) and pair(pad) == (0, 0) and pair(dilation) == (1, 1)


	def convNdBackwardData(self, grad, W, bias=None, data=None, stride=1, pad=0, dilation=1, postpad=0, groups=1,
						   algo=ConvBwdDataAlgo.auto.value, out=None, allocator=None, compact=False):
		if compact and data is not None and bias is None and out is None and self.compactGradSupported(W, stride, pad, dilation):
			# backend-internal (Sequential.planFusion): dx[.., 2i, 2j] = W^T dy[.., i, j] and zero elsewhere — computed on the
			# compact grid; StridedGrad carries it to the fan-in kernel that knows where the zeros are
			small = self.convNdBackwardData(grad, W, None, None, 1, 0, 1, 0, groups, algo, None, allocator)
			assert small.shape[2:] == tuple((d + 1) // 2 for d in data.shape[2:])
			return StridedGrad(small, data.shape)

		lazy = grad if isinstance(grad, DeferredBNGrad) else None      # backend-internal: BN backward folded into the gather
		if lazy is not None:
			grad = lazy.grad
		assert grad.ndim == W.ndim and grad.shape[1] == W.shape[0]
		requireF32(grad, W, bias, out)

		(sh, sw), (ph, pw), (dh, dw) = pair(stride), pair(pad), pair(dilation)

		if data is not None:
			inshape = data.shape
		else:
			poh, pow_ = pair(postpad if postpad is not None else 0)
			_, _, oh, ow = grad.shape
			_, cg, r, s = W.shape
			inshape = (
				grad.shape[0], cg * groups, (oh - 1) * sh + dh * (r - 1) - 2 * ph + 1 + poh,
				(ow - 1) * sw + dw * (s - 1) - 2 * pw + 1 + pow_
			)

		desc = self.convDesc(inshape, W.shape, stride, pad, dilation, groups)
		p, q = c_int(0), c_int(0)
		lib.pz_conv2d_out_shape(byref(desc), byref(p), byref(q))
		if (p.value, q.value) != grad.shape[2:]:
			raise ValueError("gradient maps %s do not match the convolution geometry %s" % (grad.shape[2:], (p.value, q.value)))

		out = GPUArray.empty(inshape, dtype=grad.dtype, allocator=allocator) if out is None else out

		algo = toAlgoId(algo)
		size = c_size_t(0)
		lib.pz_conv2d_workspace_bytes(byref(desc), lib.CONV_BWD_DATA, algo, byref(size))
		ws = self.workspace(size.value, allocator)

		if lazy is not None and self.bnFoldSupported(desc, algo):
			lib.pz_conv2d_bwd_data_bn(
				byref(desc), grad.ptr, lazy.data.ptr, lazy.coef.ptr, W.ptr, out.ptr, algo, ptrOf(ws), size.value, None
			)
		else:
			if lazy is not None:
				grad = lazy.materialize()
			lib.pz_conv2d_bwd_data(byref(desc), grad.ptr, W.ptr, out.ptr, algo, ptrOf(ws), size.value, None)

		if bias is not None:           # deconvolution forward: bias over the produced maps, rows of the (n*maps, pixels) view
			assert bias.size == out.shape[1]
			lib.pz_bias_add(
				out.ptr, out.ptr, bias.ptr, 1, out.shape[0] * out.shape[1], prod(out.shape[2:]), out.shape[1], 0, None
			)

		return out


	def convNdBackwardParams(self, data, grad, W, stride=1, pad=0, dilation=1, groups=1, withbias=False, deconv=False,
							 wgrad=None, bgrad=None, scale=1.0, momentum=0.0, algo=ConvBwdFilterAlgo.auto.value,
							 allocator=None):
		lazy = grad if isinstance(grad, DeferredBNGrad) else None
		if lazy is not None:
			grad = lazy.grad
		assert data.ndim == grad.ndim and grad.shape[1] == W.shape[0] and data.shape[1] == W.shape[1] * groups
		requireF32(data, grad, wgrad, bgrad)
		# deconv=True (Backend/Dnn.py wrapDeconvNdBackwardParams passes the deconvolution's output gradient as `data` and its
		# input as `grad`): the filter gradient is the same contraction; only the bias gradient sums over `data`'s maps
		# instead of `grad`'s (Hip/Wrappers/MIOpen.py:435-436)
		biasof = data if deconv else grad

		desc = self.convDesc(data.shape, W.shape, stride, pad, dilation, groups)

		# accumulate contract of Hip/Wrappers/MIOpen.py:414-433,441-455: a destination that was passed in AND
		# (scale, momentum) != (1, 0) -> dst = momentum*dst + scale*d; otherwise dst = d
		accumulate = scale != 1.0 or momentum != 0.0
		wcoef = (scale, momentum) if (wgrad is not None and accumulate) else (1.0, 0.0)
		bcoef = (scale, momentum) if (bgrad is not None and accumulate) else (1.0, 0.0)

		wgrad = GPUArray.empty(W.shape, dtype=W.dtype, allocator=allocator) if wgrad is None else wgrad

		algo = toAlgoId(algo)
		size = c_size_t(0)
		lib.pz_conv2d_workspace_bytes(byref(desc), lib.CONV_BWD_FILTER, algo, byref(size))
		ws = self.workspace(size.value, allocator)

		bg = None
		if withbias:
			bg = GPUArray.empty((biasof.shape[1], ), dtype=data.dtype, allocator=allocator) if bgrad is None else bgrad

		fused = withbias and bcoef == wcoef and not deconv    # one library call reduces dw and db with the same (alpha, beta)
		folded = lazy is not None and not withbias and self.bnFoldSupported(desc, algo)
		if lazy is not None and not folded:
			grad = lazy.materialize()

		side = self.filterGradStream() if (not withbias or fused) else None
		st = side.handle if side is not None else None
		if side is not None:
			self.sideRefs.append((data, grad, lazy, ws, wgrad, bg))

		if folded:
			lib.pz_conv2d_bwd_filter_bn(
				byref(desc), data.ptr, grad.ptr, lazy.data.ptr, lazy.coef.ptr, wgrad.ptr, wcoef[0], wcoef[1], algo,
				ptrOf(ws), size.value, st
			)
			return wgrad

		lib.pz_conv2d_bwd_filter(
			byref(desc), data.ptr, grad.ptr, wgrad.ptr, ptrOf(bg) if fused else None, wcoef[0], wcoef[1], algo,
			ptrOf(ws), size.value, st
		)

		if withbias and not fused:
			n, k = biasof.shape[:2]
			persample = self.backend.matmod.matsum(biasof.reshape(n * k, prod(biasof.shape[2:])), axis=1, allocator=allocator)
			self.backend.matmod.matsum(persample.reshape(n, k), axis=0, out=bg, alpha=bcoef[0], beta=bcoef[1])

		return (wgrad, bg) if withbias else wgrad


	def convNdbenchmark(self, datashape, Wshape, dtype, stride=1, pad=0, dilation=1, groups=1, algoCount=10,
						exhaustive=False):
		"""Times the kernel families that serve each pass (implicit GEMM, Winograd where it applies, direct) on scratch
		tensors: (algo id, seconds, workspace bytes) triples, the result shape of Hip/Wrappers/MIOpen.py:465-519."""
		bnd = self.backend
		data = GPUArray.zeros(datashape, dtype=dtype, allocator=bnd.memoryPool)
		W = GPUArray.zeros(Wshape, dtype=dtype, allocator=bnd.memoryPool)
		desc = self.convDesc(datashape, Wshape, stride, pad, dilation, groups)

		out = self.convNd(data, W, None, stride, pad, dilation, groups, allocator=bnd.memoryPool)
		results = []

		for which, run in (
			(lib.CONV_FWD, lambda a: self.convNd(data, W, None, stride, pad, dilation, groups, a, None, bnd.memoryPool)),
			(lib.CONV_BWD_DATA, lambda a: self.convNdBackwardData(
				out, W, None, data, stride, pad, dilation, 0, groups, a, None, bnd.memoryPool
			)),
			(lib.CONV_BWD_FILTER, lambda a: self.convNdBackwardParams(
				data, out, W, stride, pad, dilation, groups, False, False, None, None, 1.0, 0.0, a, bnd.memoryPool
			)),
		):
			perfs = []
			for algo in (ConvFwdAlgo.implicitGemm.value, ConvFwdAlgo.winograd.value, ConvFwdAlgo.direct.value):
				if self.convAlgoUsed(desc, which, algo) != algo:        # e.g. Winograd asked of a layer it does not serve
					continue
				size = c_size_t(0)
				lib.pz_conv2d_workspace_bytes(byref(desc), which, toAlgoId(algo), byref(size))
				secs, _ = bnd.timeKernel(run, (algo, ), looplength=3, log=False, normalize=True)
				perfs.append((algo, secs, size.value))

			results.append(sorted(perfs, key=lambda perf: perf[1])[:algoCount])

		return tuple(results)


	@staticmethod
	def poolDesc(shape, size, stride, pad, mode):
		(fh, fw), (sh, sw), (ph, pw) = pair(size), pair(stride), pair(pad)
		n, c, h, w = shape
		return PoolDesc(n, c, h, w, fh, fw, sh, sw, ph, pw, mode)


	def poolNd(self, data, size=2, stride=2, pad=0, mode=PoolMode.max.value, test=False, out=None, allocator=None):
		assert data.ndim == 4
		requireF32(data, out)

		desc = self.poolDesc(data.shape, size, stride, pad, mode)
		p, q = c_int(0), c_int(0)
		lib.pz_pool2d_out_shape(byref(desc), byref(p), byref(q))
		outshape = data.shape[:2] + (p.value, q.value)

		out = GPUArray.empty(outshape, dtype=data.dtype, allocator=allocator) if out is None else out

		workspace = None
		if not test:
			# training mode returns the arg-max workspace (1 byte per output element; dummy for average pooling)
			nbytes = prod(outshape) if mode == PoolMode.max.value else 4
			workspace = GPUArray.empty((nbytes, ), dtype=np.uint8, allocator=allocator)

		index = workspace.ptr if (workspace is not None and mode == PoolMode.max.value) else None
		lib.pz_pool2d_fwd(byref(desc), data.ptr, out.ptr, index, None)

		return out if test else (out, workspace)


	def poolNdBackward(self, grad, indata, outdata, workspace, size=2, stride=2, pad=0, mode=PoolMode.max.value,
					   out=None, allocator=None):
		assert grad.ndim == 4
		requireF32(grad, indata, outdata, out)

		desc = self.poolDesc(indata.shape, size, stride, pad, mode)
		out = GPUArray.empty(indata.shape, dtype=grad.dtype, allocator=allocator) if out is None else out

		index = workspace.ptr if (workspace is not None and mode == PoolMode.max.value) else None
		lib.pz_pool2d_bwd(byref(desc), grad.ptr, indata.ptr, outdata.ptr, index, out.ptr, None)
		return out


	@staticmethod
	def softmaxGeometry(data, mode):
		n, c = data.shape[0], data.shape[1]
		spatial = prod(data.shape[2:])

		if mode == SoftMaxMode.perActivation.value:
			c, spatial = c * spatial, 1

		return n, c, spatial


	def softmaxNd(self, data, mode=SoftMaxMode.spatial.value, algo=None, out=None, allocator=None):
		requireF32(data, out)
		out = GPUArray.empty(data.shape, dtype=data.dtype, allocator=allocator) if out is None else out

		n, c, spatial = self.softmaxGeometry(data, mode)
		lib.pz_softmax_fwd(data.ptr, out.ptr, n, c, spatial, None)
		return out


	def softmaxNdBackward(self, grad, outdata, mode=SoftMaxMode.spatial.value, algo=None, out=None, allocator=None):
		requireF32(grad, outdata, out)
		out = GPUArray.empty(grad.shape, dtype=grad.dtype, allocator=allocator) if out is None else out

		n, c, spatial = self.softmaxGeometry(grad, mode)
		lib.pz_softmax_bwd(grad.ptr, outdata.ptr, out.ptr, n, c, spatial, None)
		return out


	def bnWorkspace(self, n, c, hw, allocator):
		size = c_size_t(0)
		lib.pz_bn_workspace_bytes(n, c, hw, byref(size))
		return GPUArray.empty((size.value, ), dtype=np.uint8, allocator=allocator), size.value


	def bnGateStats(self, grad0, grad1, outdata, targets, allocator=None, mask=None):
		"""Backend-internal (Sequential.planFusion): g = (grad0 + grad1) * (outdata > 0) plus, for each of the one or two
		`targets` = (bnInput, savemean), the partial sums a following batchNormNdBackward(g, bnInput, ..., partials=)
		would otherwise recompute. Returns (g, [partials...])."""
		up2 = isinstance(grad0, StridedGrad) and isinstance(grad1, StridedGrad)
		if not up2:
			grad0 = grad0.materialize() if isinstance(grad0, StridedGrad) else grad0
			grad1 = grad1.materialize() if isinstance(grad1, StridedGrad) else grad1
		requireF32(grad0 if not up2 else grad0.compact, grad1 if not up2 else grad1.compact, outdata)
		assert 1 <= len(targets) <= 2 and tuple(grad0.shape) == tuple(grad1.shape) == tuple(outdata.shape)
		n, c, hw = grad0.shape[0], grad0.shape[1], prod(grad0.shape[2:])

		out = GPUArray.empty(outdata.shape, dtype=outdata.dtype, allocator=allocator)
		size = c_size_t(0)
		lib.pz_bn_workspace_bytes(n, c, hw, byref(size))
		parts = [GPUArray.empty((size.value // 4, ), dtype=np.float32, allocator=allocator) for _ in targets]

		(xa, ma), (xb, mb) = targets[0], (targets[1] if len(targets) == 2 else (None, None))
		assert xa.shape == outdata.shape and (xb is None or xb.shape == outdata.shape)
		# `mask` (ReluMask of exactly `outdata`, from bnApplyAdd): the gate comes from one bit per element, outdata is not read
		mptr = mask.bits.ptr if mask is not None and mask.tensor is outdata else None
		if up2:
			lib.pz_bn_gate_stats_up2(
				grad0.compact.ptr, grad1.compact.ptr, outdata.ptr, mptr, out.ptr, n, c, outdata.shape[2], outdata.shape[3],
				xa.ptr, ma.ptr, parts[0].ptr, ptrOf(xb), ptrOf(mb), parts[1].ptr if xb is not None else None, None
			)
			return out, parts
		lib.pz_bn_gate_stats(
			grad0.ptr, grad1.ptr, outdata.ptr, mptr, out.ptr, n, c, hw, xa.ptr, ma.ptr, parts[0].ptr,
			ptrOf(xb), ptrOf(mb), parts[1].ptr if xb is not None else None, None
		)
		return out, parts


	def bnApplyAdd(self, first, second, relu=False, allocator=None, withMask=False):
		"""out = act(bn(first) + second') for a DeferredBN `first` and `second` = DeferredBN | GPUArray | None
		(None: out = bn(first), no activation). Backend-internal (see Sequential.planFusion).
		`withMask` (with relu): also return the ReluMask of `out` -> (out, mask)."""
		x1 = first.tensor
		n, c, hw = x1.shape[0], x1.shape[1], prod(x1.shape[2:])
		if isinstance(second, DeferredBN):
			x2, coef2 = second.tensor, second.coef
		else:
			x2, coef2 = second, None
		if x2 is not None and x2.shape != x1.shape:
			raise ValueError("bnApplyAdd: operand shapes %s and %s differ" % (x1.shape, x2.shape))
		requireF32(x1, x2)

		out = GPUArray.empty(x1.shape, dtype=x1.dtype, allocator=allocator)
		if withMask and relu:
			size = c_size_t(0)
			lib.pz_relu_mask_bytes(n, c, hw, byref(size))
			bits = GPUArray.empty((size.value, ), dtype=np.uint8, allocator=allocator)
			lib.pz_bn_apply_add_mask(x1.ptr, first.coef.ptr, ptrOf(x2), ptrOf(coef2), out.ptr, bits.ptr, n, c, hw, 1, None)
			return out, ReluMask(out, bits)
		lib.pz_bn_apply_add(x1.ptr, first.coef.ptr, ptrOf(x2), ptrOf(coef2), out.ptr, n, c, hw, int(bool(relu)), None)
		return (out, None) if withMask else out


	def batchNormNd(self, data, mean, var, scale, bias, epsilon=1e-5, factor=1.0, test=False,
					mode=BatchNormMode.spatial.value, out=None, allocator=None, fuseRelu=False, convStats=None,
					defer=False):
		"""`fuseRelu` (backend-internal, train mode only): write relu(bn(data)) — used by Sequential for a BatchNorm
		followed by an in-place ReLU; the matching backward is batchNormNdBackward(..., bias=, fuseRelu=True)."""
		assert mean.ndim == 1 and var.ndim == 1 and scale.ndim == 1 and bias.ndim == 1
		assert data.dimAt(1) == mean.dimAt(0)
		requireF32(data, mean, var, scale, bias, out)
		if mode != BatchNormMode.spatial.value:
			raise NotImplementedError("per-activation batch normalisation is not implemented on this backend")

		n, c, hw = data.shape[0], data.shape[1], prod(data.shape[2:])

		if defer and not test and not fuseRelu and out is None and convStats is not None and convStats.tensor is data:
			# statistics from the convolution's strip sums, normalisation left to the consumer (bnApplyAdd)
			savemean = GPUArray.empty(mean.shape, dtype=data.dtype, allocator=allocator)
			saveinvvar = GPUArray.empty(var.shape, dtype=data.dtype, allocator=allocator)
			coef = GPUArray.empty((c, 2), dtype=data.dtype, allocator=allocator)
			ws, nbytes = self.bnWorkspace(n, c, hw, allocator)
			lib.pz_bn_fwd_train_defer(
				n, c, hw, scale.ptr, bias.ptr, mean.ptr, var.ptr, savemean.ptr, saveinvvar.ptr, epsilon, factor,
				convStats.stats.ptr, convStats.stats.shape[1], coef.ptr, ws.ptr, nbytes, None
			)
			return DeferredBN(data, coef, self), savemean, saveinvvar

		out = GPUArray.empty(data.shape, dtype=data.dtype, allocator=allocator) if out is None else out

		if test:
			lib.pz_bn_fwd_infer(data.ptr, out.ptr, n, c, hw, scale.ptr, bias.ptr, mean.ptr, var.ptr, epsilon, None)
			return out

		savemean = GPUArray.empty(mean.shape, dtype=data.dtype, allocator=allocator)
		saveinvvar = GPUArray.empty(var.shape, dtype=data.dtype, allocator=allocator)
		ws, nbytes = self.bnWorkspace(n, c, hw, allocator)

		act = lib.BN_ACT_RELU if fuseRelu else lib.BN_ACT_NONE

		if convStats is not None and convStats.tensor is data:
			# the producing convolution already summed this tensor per strip: no statistics pass over `data`
			lib.pz_bn_fwd_train_pre(
				data.ptr, out.ptr, n, c, hw, scale.ptr, bias.ptr, mean.ptr, var.ptr, savemean.ptr, saveinvvar.ptr,
				epsilon, factor, act, convStats.stats.ptr, convStats.stats.shape[1], ws.ptr, nbytes, None
			)
		else:
			lib.pz_bn_fwd_train_act(
				data.ptr, out.ptr, n, c, hw, scale.ptr, bias.ptr, mean.ptr, var.ptr, savemean.ptr, saveinvvar.ptr,
				epsilon, factor, act, ws.ptr, nbytes, None
			)
		return out, savemean, saveinvvar


	def batchNormNdBackward(self, grad, data, scale, savemean=None, saveinvvar=None, epsilon=1e-5,
							mode=BatchNormMode.spatial.value, out=None, allocator=None, bias=None, fuseRelu=False,
							accumulate=None, partials=None, lazyGrad=False):
		"""`accumulate` (backend-internal) = (scalegradDst, biasgradDst, alpha, beta): additionally
		dst = alpha*fresh + beta*dst for both parameter gradients inside the same launch."""
		assert data.ndim == grad.ndim
		requireF32(grad, data, scale, savemean, saveinvvar, out, bias)
		if fuseRelu and bias is None:
			raise ValueError("batchNormNdBackward: the fused ReLU gate needs the layer's bias")
		if savemean is None or saveinvvar is None:
			raise ValueError("batchNormNdBackward needs the saved mean / inverse variance of the forward pass")

		out = GPUArray.empty(grad.shape, dtype=grad.dtype, allocator=allocator) if out is None else out
		scalegrad = GPUArray.empty(scale.shape, dtype=scale.dtype, allocator=allocator)
		bgrad = GPUArray.empty(scale.shape, dtype=scale.dtype, allocator=allocator)

		n, c, hw = data.shape[0], data.shape[1], prod(data.shape[2:])
		ws, nbytes = self.bnWorkspace(n, c, hw, allocator)

		sdst, bdst, alpha, beta = accumulate if accumulate is not None else (None, None, 1.0, 0.0)
		requireF32(sdst, bdst)

		if partials is not None and not fuseRelu and lazyGrad:
			# statistics already summed (bnGateStats) and the consumer folds the apply pass into its gathers
			coef = GPUArray.empty((c, 4), dtype=np.float32, allocator=allocator)
			lib.pz_bn_bwd_coef(
				n, c, hw, scale.ptr, savemean.ptr, saveinvvar.ptr, scalegrad.ptr, bgrad.ptr, ptrOf(sdst), ptrOf(bdst), alpha, beta,
				partials.ptr, coef.ptr, None
			)

			def apply():
				lib.pz_bn_bwd_from_partials(
					data.ptr, grad.ptr, out.ptr, n, c, hw, scale.ptr, savemean.ptr, saveinvvar.ptr, scalegrad.ptr, bgrad.ptr,
					None, None, 1.0, 0.0, partials.ptr, None
				)
				return out

			return DeferredBNGrad(grad, data, coef, apply), scalegrad, bgrad

		if partials is not None and not fuseRelu:           # statistics already summed by bnGateStats: apply pass only
			lib.pz_bn_bwd_from_partials(
				data.ptr, grad.ptr, out.ptr, n, c, hw, scale.ptr, savemean.ptr, saveinvvar.ptr, scalegrad.ptr, bgrad.ptr,
				ptrOf(sdst), ptrOf(bdst), alpha, beta, partials.ptr, None
			)
			return out, scalegrad, bgrad
		lib.pz_bn_bwd_acc(
			data.ptr, grad.ptr, out.ptr, n, c, hw, scale.ptr, bias.ptr if fuseRelu else None, savemean.ptr,
			saveinvvar.ptr, scalegrad.ptr, bgrad.ptr, lib.BN_ACT_RELU if fuseRelu else lib.BN_ACT_NONE,
			ptrOf(sdst), ptrOf(bdst), alpha, beta, ws.ptr, nbytes, None
		)
		return out, scalegrad, bgrad


	def lrn(self, *args, **kwargs):
		raise NotImplementedError("LRN is outside the implemented operator path")


	lrnBackward = lrn


# ---------------------------------------------------------------------------------------------- matrix-vector module
class MatModule:
	"""matsum / addVecToMat / argmax — Cuda/Kernels/MatVec.py:231-374."""

	def __init__(self, backend):
		self.backend, self.GPUArray = backend, backend.GPUArray


	def matsum(self, tensor, axis=0, out=None, alpha=1.0, beta=0.0, allocator=None):
		requireF32(tensor, out)
		assert 0 <= axis < tensor.ndim

		outshape = tensor.shape[:axis] + tensor.shape[axis + 1:]
		if out is None:
			out = GPUArray.zeros(outshape, dtype=tensor.dtype, allocator=allocator)
		else:
			assert out.shape == outshape

		if axis == tensor.ndim - 1:
			lib.pz_reduce_sum_rows(tensor.ptr, prod(tensor.shape[:-1]), tensor.shape[-1], out.ptr, alpha, beta, None)
		else:
			z, h, w = prod(tensor.shape[:axis]), tensor.shape[axis], prod(tensor.shape[axis + 1:])
			lib.pz_reduce_sum_cols(tensor.ptr, z, h, w, out.ptr, alpha, beta, None)

		return out


	def argmax(self, tensor, axis=0, allocator=None):
		requireF32(tensor)
		assert 0 <= axis < tensor.ndim

		idx = GPUArray.empty(tensor.shape[:axis] + tensor.shape[axis + 1:], dtype=np.int32, allocator=allocator)

		if axis == tensor.ndim - 1:
			lib.pz_argmax_rows(tensor.ptr, prod(tensor.shape[:-1]), tensor.shape[-1], idx.ptr, None)
		else:
			z, h, w = prod(tensor.shape[:axis]), tensor.shape[axis], prod(tensor.shape[axis + 1:])
			lib.pz_argmax_cols(tensor.ptr, z, h, w, idx.ptr, None)

		return idx


	def argmin(self, tensor, axis=0, allocator=None):
		raise NotImplementedError("argmin is not on the implemented operator path")


	def matvec(self, *args, **kwargs):
		raise NotImplementedError("matvec (GroupLinear) is outside the implemented operator path")


	def addVecToMat(self, vec, mat, axis=0, out=None, allocator=None, tiled=False):
		requireF32(vec, mat, out)
		assert vec.ndim == mat.ndim - 1 and 0 <= axis < 2
		assert mat.shape[:-2] == vec.shape[:-1] or tiled

		out = GPUArray.empty(mat.shape, dtype=mat.dtype, allocator=allocator) if out is None else out
		z = prod(mat.shape[:-2])
		n, m = mat.shape[-2:]

		if tiled:          # one vector shared by every matrix of the batch
			for b in range(z):
				lib.pz_bias_add(out.ptr + b * n * m * 4, mat.ptr + b * n * m * 4, vec.ptr, 1, n, m, vec.shape[-1], axis, None)
			return out

		if axis == 1:
			assert mat.dimAt(-1) % vec.dimAt(-1) == 0
		else:
			assert mat.dimAt(-2) == vec.dimAt(-1)

		lib.pz_bias_add(out.ptr, mat.ptr, vec.ptr, z, n, m, vec.dimAt(-1), axis, None)
		return out


# ---------------------------------------------------------------------------------------------- cost module
class ReductionCallable:
	def __init__(self, fn):
		self.fn = fn

	def __call__(self, *args, **kwargs):
		return self.fn(*args, **kwargs)


class CostModule:
	"""crossEntropy + accuracy kernels — Cuda/Kernels/Costs.py:160-247."""

	def __init__(self, backend):
		self.backend, self.GPUArray, self.dnn = backend, backend.GPUArray, backend.dnn
		self.accKernelCache = {}


	def getAccuracyKernel(self, name):
		krl = self.accKernelCache.get(name, None)

		if krl is None:
			if name != "calcAccuracy":
				raise NotImplementedError(name)

			def calcAccuracy(x, y, allocator=None):
				assert x.dtype == np.int32 and y.dtype == np.int32 and x.size == y.size
				out = GPUArray.empty((), dtype=np.float32, allocator=allocator)
				lib.pz_count_neq_i32(x.ptr, y.ptr, x.size, out.ptr, None)
				return out

			krl = self.accKernelCache[name] = ReductionCallable(calcAccuracy)

		return krl


	def crossEntropy(self, scores, labels, weights=None, error=None, allocator=None):
		assert scores.dtype == np.float32 and labels.dtype == np.int32
		requireF32(scores, weights)

		n, c = scores.shape[:2]
		spatial = prod(scores.shape[2:])

		grad = GPUArray.empty(scores.shape, dtype=np.float32, allocator=allocator)
		if error is None:
			error = GPUArray.empty((), dtype=np.float32, allocator=allocator)

		ws = GPUArray.empty((n * spatial, ), dtype=np.float32, allocator=allocator)
		lib.pz_cross_entropy(
			scores.ptr, labels.ptr, ptrOf(weights), n, c, spatial, grad.ptr, error.ptr, ws.ptr, ws.nbytes, None
		)
		return error, grad


	def svm(self, *args, **kwargs):
		raise NotImplementedError("SVM cost is outside the implemented operator path")


class MemModule:
	"""transpose / moveaxis / swapaxes / depthConcat / depthSplit — Cuda/Kernels/Memory.py:81-203. The reference
	instantiates a `transformNd` kernel per rank; here every case is one strided copy (pz_strided_copy, up to 6 axes)
	between a tensor and a strided VIEW of the other side, so the index arithmetic lives in the view's strides."""

	def __init__(self, backend):
		self.backend = backend


	@staticmethod
	def viewLike(ary, shape, strides, offsetBytes=0):
		return GPUArray(shape, ary.dtype, gpudata=ary.gpudata[offsetBytes:], strides=strides)


	def transpose(self, tensor, axes=None, out=None, allocator=None):
		if axes is not None and len(axes) != tensor.ndim:
			raise ValueError("axes do not match the tensor rank")
		if tensor.dtype.itemsize != 4:
			raise NotImplementedError("memmod: 4-byte element types only (this backend computes in float32)")

		axes = tuple(reversed(range(tensor.ndim))) if axes is None else tuple(axes)
		shape = tuple(tensor.dimAt(axis) for axis in axes)

		if out is None:
			out = GPUArray.empty(shape, dtype=tensor.dtype, allocator=allocator)
		elif out.shape != shape:
			raise ValueError("transpose output has shape %s, expected %s" % (out.shape, shape))

		outstrides = [0] * len(axes)
		for i, axis in enumerate(axes):
			outstrides[axis] = out.strideAt(i)

		if tensor.size > 0:
			self.viewLike(out, tensor.shape, outstrides).stridedCopyFrom(tensor)
		return out


	def moveaxis(self, data, src, dst, out=None, allocator=None):
		if src < dst:
			axes = tuple(range(src)) + tuple(range(src + 1, dst + 1)) + (src, ) + tuple(range(dst + 1, data.ndim))
		else:
			axes = tuple(range(dst)) + (src, ) + tuple(range(dst, src)) + tuple(range(src + 1, data.ndim))
		return self.transpose(data, axes, out=out, allocator=allocator)


	def swapaxes(self, data, axis1, axis2, out=None, allocator=None):
		axes = list(range(data.ndim))
		axes[axis1], axes[axis2] = axes[axis2], axes[axis1]
		return self.transpose(data, tuple(axes), out=out, allocator=allocator)


	@staticmethod
	def centred(big, small):
		"""byte offset that centres `small`'s maps inside `big`'s (Memory.py:178,194)"""
		return (big.dimAt(2) - small.dimAt(2)) // 2 * big.strideAt(2) + (big.dimAt(3) - small.dimAt(3)) // 2 * big.strideAt(3)


	def depthConcat(self, tensors, out=None, allocator=None):
		assert all(tn.ndim == 4 and tn.dtype == tensors[0].dtype for tn in tensors)
		assert all(tn.dimAt(0) == tensors[0].dimAt(0) for tn in tensors)

		depth = sum(tn.dimAt(1) for tn in tensors)
		h, w = max(tn.dimAt(2) for tn in tensors), max(tn.dimAt(3) for tn in tensors)
		shape = (tensors[0].dimAt(0), depth, h, w)

		if out is None:
			out = GPUArray.zeros(shape, dtype=tensors[0].dtype, allocator=allocator)
		elif out.shape != shape:
			raise ValueError("depthConcat output has shape %s, expected %s" % (out.shape, shape))

		offset = 0
		for tn in tensors:
			self.viewLike(out, tn.shape, out.strides, offset + self.centred(out, tn)).stridedCopyFrom(tn)
			offset += out.strideAt(1) * tn.dimAt(1)
		return out


	def depthSplit(self, grad, tensors, allocator=None):
		assert all(tn.ndim == 4 and tn.dtype == tensors[0].dtype for tn in tensors)
		ingrads = [GPUArray.empty(tn.shape, dtype=tn.dtype, allocator=allocator) for tn in tensors]

		offset = 0
		for gr in ingrads:
			gr.stridedCopyFrom(self.viewLike(grad, gr.shape, grad.strides, offset + self.centred(grad, gr)))
			offset += grad.strideAt(1) * gr.dimAt(1)
		return ingrads


class StubModule:
	def __init__(self, name):
		self.stubName = name

	def __getattr__(self, item):
		def raiser(*args, **kwargs):
			raise NotImplementedError("%s.%s is outside the implemented operator path" % (self.stubName, item))
		return raiser


# ---------------------------------------------------------------------------------------------- element-wise kernel objects
class EltwiseKernel:
	"""Callable with the launch signature of the reference kernel objects:
	ker(*arrays_then_scalars, slice=None, stream=None) — Cuda/SourceModule.py:203-226."""

	def __init__(self, op, narrays, nscalars, name, rawScalar=()):
		self.op, self.narrays, self.nscalars, self.name = op, narrays, nscalars, name
		self.rawScalar = rawScalar      # indices of scalars that are integers travelling as raw 32-bit words


	def __call__(self, *args, **kwargs):
		if len(args) != self.narrays + self.nscalars:
			raise TypeError("%s expects %d arguments, got %d" % (self.name, self.narrays + self.nscalars, len(args)))

		arrays, scalars = args[:self.narrays], args[self.narrays:]
		for ary in arrays:
			if not ary.contiguous:
				raise ValueError("gpuarray is not contiguous")

		words = np.empty(len(scalars), dtype=np.float32)
		for i, value in enumerate(scalars):
			if i in self.rawScalar:
				words.view(np.uint32)[i] = np.uint32(int(value))
			else:
				words[i] = value

		eltwise(self.op, arrays[0].size, arrays, words, slc=kwargs.get("slice", None), stream=kwargs.get("stream", None))


def memoizedKernel(op, narrays, nscalars, name, rawScalar=()):
	"""`ker(dtype) -> callable` factories (the @memoize'd kernels of Cuda/Kernels/ElementWise.py)."""
	kernel = EltwiseKernel(op, narrays, nscalars, name, rawScalar)

	def factory(dtype):
		if np.dtype(dtype) != np.float32:
			raise NotImplementedError("%s: dtype %s (this backend computes in float32)" % (name, dtype))
		return kernel

	factory.__name__ = name
	return factory


# ---------------------------------------------------------------------------------------------- helpers
class SharedArray:
	"""Flat parameter/gradient arena — Cuda/Utils.py:19-64 (16-byte aligned blocks in registration order)."""
	alignment = 16

	def __init__(self, dtype=np.float32, allocator=None):
		self.ary = None
		self.blocks = OrderedDict()
		self.dtype = np.dtype(dtype)
		self.allocator = allocator


	def register(self, shape, dtype, name):
		assert name not in self.blocks
		assert dtype == self.dtype
		self.blocks[name] = (shape, prod(shape) * self.dtype.itemsize)


	def build(self):
		total = sum(self.align(nbytes) for _, nbytes in self.blocks.values())
		self.ary = GPUArray.empty((total // self.dtype.itemsize, ), dtype=self.dtype, allocator=self.allocator)

		blocks, offset = OrderedDict(), 0
		for name, (shape, nbytes) in self.blocks.items():
			blocks[name] = GPUArray(shape, self.dtype, gpudata=self.ary.gpudata[offset:offset + nbytes])
			offset += self.align(nbytes)

		self.blocks = blocks


	def __getitem__(self, item):
		return self.blocks[item]


	@classmethod
	def align(cls, nbytes):
		return (nbytes + cls.alignment - 1) // cls.alignment * cls.alignment


class QueueManager:
	"""borrow/give pool of Stream or Event objects — Cuda/Utils.py:67-94."""

	def __init__(self, objtype):
		self.objtype, self.items = objtype, []

	def reserve(self, nitems):
		self.items.extend(self.objtype() for _ in range(nitems))

	def borrow(self, nitems):
		if len(self.items) < nitems:
			self.reserve(nitems - len(self.items))
		end = len(self.items) - nitems
		borrowed, self.items = self.items[end:], self.items[:end]
		return borrowed

	def give(self, items):
		self.items.extend(items)

	def clear(self):
		self.items.clear()


class RandomNumberGenerator:
	"""fillInteger / fillUniform / fillNormal — Cuda/Source/Libs/CuRand.c:231-234 (Philox here, XORWOW there)."""

	def __init__(self, type=None, seed=0):
		self.type, self.seed = "philox4x32-10", int(seed) & 0xffffffffffffffff
		handle = c_void_p()
		lib.pz_rng_create(self.seed, byref(handle))
		self.handle = handle.value


	def fillInteger(self, data):
		assert data.contiguous and data.dtype.itemsize == 4
		lib.pz_rng_fill_u32(self.handle, data.ptr, data.size, None)


	def fillUniform(self, data):
		requireF32(data)
		lib.pz_rng_fill_uniform(self.handle, data.ptr, data.size, None)


	def fillNormal(self, data, mean=0.0, stddev=1.0):
		requireF32(data)
		lib.pz_rng_fill_normal(self.handle, data.ptr, data.size, mean, stddev, None)


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
		if initmode > 0 >= self.initmode:
			self.initLibs(logger)
		if initmode > 1 >= self.initmode:
			self.initKernels()
		self.initmode = max(initmode, self.initmode)


	def initLibs(self, logger=None):
		self.blas = BlasContext(self)
		self.dnn = DnnContext(self)

		if logger is not None:
			logger.debug("Created blas/dnn contexts (%s; %s)", self.blas.getVersion(), self.dnn.getVersion())


	def initKernels(self):
		if self.dnn is None:
			self.initLibs()

		self.matmod = MatModule(self)
		self.costmod = CostModule(self)
		self.getAccuracyKernel = self.costmod.getAccuracyKernel

		self.memmod = MemModule(self)
		for name in ("ctcmod", "embedmod", "padmod", "poolmod", "prelumod", "upsamplemod"):
			setattr(self, name, StubModule(name))

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

		def unsupported(*args, **kwargs):
			raise NotImplementedError("this kernel is outside the implemented operator path (fp32-only backend)")

		self.castFP16toFP32 = self.castFP32toFP16 = unsupported
		self.bceKer = self.hingeKer = self.smoothL1Ker = self.l1HingeKer = unsupported

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


	def concatenate(self, tup, axis, out=None, allocator=None):
		ary = tup[0]
		dtype, reduced = ary.dtype, ary.shape[:axis] + ary.shape[axis + 1:]
		assert all(a.dtype == dtype and a.shape[:axis] + a.shape[axis + 1:] == reduced for a in tup[1:])

		shape = reduced[:axis] + (sum(a.dimAt(axis) for a in tup), ) + reduced[axis:]
		if out is None:
			out = GPUArray.empty(shape, dtype=dtype, allocator=allocator)
		else:
			assert out.shape == shape and out.dtype == dtype

		dstPitch = out.strideAt(axis - 1) if axis > 0 else out.nbytes
		height, offset = prod(shape[:axis]), 0

		for a in tup:
			width = a.strideAt(axis - 1) if axis > 0 else a.nbytes
			driver.memcpy2D(width, height, a.gpudata, width, out.gpudata, dstPitch, dstX=offset)
			offset += width

		return out


	def split(self, ary, sections, axis, allocator=None):
		shape = ary.shape
		assert sum(sections) == shape[axis]

		outs = [
			GPUArray.empty(shape[:axis] + (sec, ) + shape[axis + 1:], dtype=ary.dtype, allocator=allocator)
			for sec in sections
		]

		srcPitch = ary.strideAt(axis - 1) if axis > 0 else ary.nbytes
		height, offset = prod(shape[:axis]), 0

		for out in outs:
			width = out.strideAt(axis - 1) if axis > 0 else out.nbytes
			driver.memcpy2D(width, height, ary.gpudata, srcPitch, out.gpudata, width, srcX=offset)
			offset += width

		return outs


	def tile(self, ary, repeats, axis, allocator=None):
		return self.concatenate([ary] * repeats, axis=axis, allocator=allocator)


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


	def instanceNorm2d(self, *args, **kwargs):
		raise NotImplementedError("instance normalisation is outside the implemented operator path")


	instanceNorm2dBackward = instanceNorm2d


	def createRnn(self, *args, **kwargs):
		raise NotImplementedError("RNNs are outside the implemented operator path")


	acquireRnnParams = updateRnnParams = createRnn


	@staticmethod
	def deviceSupportsBatchHint():
		return False


class AddKernelFactory:
	"""addKer(dtype)(out, x, alpha, y, beta): out = alpha*x + beta*y — Cuda/Kernels/ElementWise.py:1017-1045
	(note the interleaved array/scalar argument order)."""

	def __call__(self, dtype):
		if np.dtype(dtype) != np.float32:
			raise NotImplementedError("addKer: dtype %s" % dtype)
		return self.launch

	@staticmethod
	def launch(out, x, alpha, y, beta, slice=None, stream=None):
		eltwise(lib.OP_ADD, out.size, (out, x, y), np.array([alpha, beta], dtype=np.float32), slc=slice, stream=stream)


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
