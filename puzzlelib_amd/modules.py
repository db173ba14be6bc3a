"""
The kernel-module objects of the backend — `matmod`, `costmod`, `memmod`, `poolmod`, `prelumod`, `padmod`, `upsamplemod`,
`ctcmod`, `embedmod` (Backend/Kernels/*.py read them off the backend object; originals: Cuda/Kernels/MatVec.py, Costs.py,
Memory.py, Pool.py, PRelu.py, Pad.py, Upsample.py, CTC.py, Embedder.py). Signature glue over the C entry points; nothing
here defers work.
"""
import os, weakref, sys, time, ctypes
from ctypes import byref, c_int, c_size_t, c_void_p

import numpy as np

from puzzlelib_amd import lib, driver, lazy, fusion
from puzzlelib_amd.lib import HipError, ConvDesc, PoolDesc
from puzzlelib_amd.driver import streamHandle
from puzzlelib_amd.gpuarray import GPUArray, prod, eltwise, contiguousStrides
from puzzlelib_amd.common import (
	ConvFwdAlgo, ConvBwdFilterAlgo, ConvBwdDataAlgo, PoolMode, SoftMaxMode, BatchNormMode, LRNMode, RNNMode, DirectionMode, RNNAlgo,
	GroupFormat, ConvPerf, toAlgoId, pair, requireF32, rptrOf
)


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
			lib.pz_reduce_sum_rows(tensor.rptr, prod(tensor.shape[:-1]), tensor.shape[-1], out.wptr, alpha, beta, None)
		else:
			z, h, w = prod(tensor.shape[:axis]), tensor.shape[axis], prod(tensor.shape[axis + 1:])
			lib.pz_reduce_sum_cols(tensor.rptr, z, h, w, out.wptr, alpha, beta, None)

		return out


	def argmax(self, tensor, axis=0, allocator=None):
		requireF32(tensor)
		assert 0 <= axis < tensor.ndim

		idx = GPUArray.empty(tensor.shape[:axis] + tensor.shape[axis + 1:], dtype=np.int32, allocator=allocator)

		if axis == tensor.ndim - 1:
			lib.pz_argmax_rows(tensor.rptr, prod(tensor.shape[:-1]), tensor.shape[-1], idx.optr, None)
		else:
			z, h, w = prod(tensor.shape[:axis]), tensor.shape[axis], prod(tensor.shape[axis + 1:])
			lib.pz_argmax_cols(tensor.rptr, z, h, w, idx.optr, None)

		return idx


	def argmin(self, tensor, axis=0, allocator=None):
		requireF32(tensor)
		assert 0 <= axis < tensor.ndim

		idx = GPUArray.empty(tensor.shape[:axis] + tensor.shape[axis + 1:], dtype=np.int32, allocator=allocator)
		if axis == tensor.ndim - 1:
			lib.pz_argmin_rows(tensor.rptr, prod(tensor.shape[:-1]), tensor.shape[-1], idx.optr, None)
		else:
			z, h, w = prod(tensor.shape[:axis]), tensor.shape[axis], prod(tensor.shape[axis + 1:])
			lib.pz_argmin_cols(tensor.rptr, z, h, w, idx.optr, None)
		return idx


	def matvec(self, mat, vec, axis=0, out=None, alpha=1.0, beta=0.0, allocator=None):
		"""Cuda/Kernels/MatVec.py:302-345: per leading index z, out[z] = alpha * mat[z] @ vec[z] (axis 1: over the last
		axis) or alpha * mat[z].T @ vec[z] (axis 0) + beta * out[z]."""
		requireF32(mat, vec, out)
		assert vec.ndim == mat.ndim - 1 and 0 <= axis < 2
		h, w = mat.shape[-2:]
		assert vec.dimAt(-1) == (w if axis == 1 else h)

		oshape = mat.shape[:-1] if axis == 1 else mat.shape[:-2] + (w, )
		if out is None:
			out = GPUArray.zeros(oshape, dtype=mat.dtype, allocator=allocator)
		else:
			assert out.shape == oshape
		lib.pz_matvec(mat.rptr, vec.rptr, out.wptr, prod(mat.shape[:-2]), h, w, axis, alpha, beta, None)
		return out


	def addVecToMat(self, vec, mat, axis=0, out=None, allocator=None, tiled=False):
		requireF32(vec, mat, out)
		assert vec.ndim == mat.ndim - 1 and 0 <= axis < 2
		assert mat.shape[:-2] == vec.shape[:-1] or tiled

		out = GPUArray.empty(mat.shape, dtype=mat.dtype, allocator=allocator) if out is None else out
		z = prod(mat.shape[:-2])
		n, m = mat.shape[-2:]

		if tiled:          # one vector shared by every matrix of the batch
			for b in range(z):
				lib.pz_bias_add(out.wptr + b * n * m * 4, mat.rptr + b * n * m * 4, vec.rptr, 1, n, m, vec.shape[-1], axis, None)
			return out

		if axis == 1:
			assert mat.dimAt(-1) % vec.dimAt(-1) == 0
		else:
			assert mat.dimAt(-2) == vec.dimAt(-1)

		lib.pz_bias_add(out.wptr, mat.rptr, vec.rptr, z, n, m, vec.dimAt(-1), axis, None)
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
			def calcAccuracy(x, y, allocator=None):
				assert x.dtype == np.int32 and y.dtype == np.int32 and x.size == y.size
				out = GPUArray.empty((), dtype=np.float32, allocator=allocator)
				lib.pz_count_neq_i32(x.rptr, y.rptr, x.size, out.optr, None)
				return out

			def costAccuracy(kind):
				def kernel(x, labels, allocator=None):
					assert x.dtype == np.float32 and labels.dtype == np.int32 and x.size == labels.size
					out = GPUArray.empty((), dtype=np.float32, allocator=allocator)
					lib.pz_cost_accuracy(kind, x.rptr, labels.rptr, x.size, out.optr, None)
					return out
				return kernel

			def klDivergence(x, y, grad, gradnorm, allocator=None):
				requireF32(x, y, grad)
				assert x.size == y.size == grad.size
				out = GPUArray.empty((), dtype=np.float32, allocator=allocator)
				lib.pz_kl_divergence(x.rptr, y.rptr, grad.optr, gradnorm, x.size, out.optr, None)
				return out

			kernels = {"calcAccuracy": calcAccuracy, "calcBCEAccuracy": costAccuracy(0), "l1HingeAccuracy": costAccuracy(1),
					   "klDivergence": klDivergence}
			if name not in kernels:
				raise NotImplementedError(name)
			krl = self.accKernelCache[name] = ReductionCallable(kernels[name])

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
			scores.rptr, labels.rptr, rptrOf(weights), n, c, spatial, grad.optr, error.optr, ws.optr, ws.nbytes, None
		)
		return error, grad


	def svm(self, scores, labels, mode, error=None, allocator=None):
		"""Cuda/Kernels/Costs.py:250-276 (mode "l1" | "l2")"""
		assert scores.dtype == np.float32 and labels.dtype == np.int32 and mode in ("l1", "l2")
		requireF32(scores)
		n, c = scores.shape[:2]
		spatial = prod(scores.shape[2:])

		grad = GPUArray.empty(scores.shape, dtype=np.float32, allocator=allocator)
		if error is None:
			error = GPUArray.empty((), dtype=np.float32, allocator=allocator)
		terms = GPUArray.empty((scores.size, ), dtype=np.float32, allocator=allocator)
		lib.pz_svm_cost(scores.rptr, labels.rptr, n, c, spatial, int(mode == "l2"), grad.optr, terms.optr, None)
		lib.pz_asum(terms.rptr, terms.size, error.optr, None)
		return error, grad


class MemModule:
	"""transpose / moveaxis / swapaxes / depthConcat / depthSplit — Cuda/Kernels/Memory.py:81-203. The reference
	instantiates a `transformNd` kernel per rank; here every case is one strided copy (pz_strided_copy, up to 6 axes)
	between a tensor and a strided VIEW of the other side, so the index arithmetic lives in the view's strides."""

	def __init__(self, backend):
		self.backend, self.GPUArray = backend, backend.GPUArray       # (module.GPUArray: Cuda/Kernels/Memory.py:84, Pool.py:120)


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


class PoolModule:
	"""maxpool2d / maxpool2dBackward / maxunpool2d / maxunpool2dBackward with index masks — Cuda/Kernels/Pool.py:117-213
	(MaxPool2D(useMask=True), MaxUnpool2D)."""

	def __init__(self, backend):
		self.backend, self.GPUArray = backend, backend.GPUArray       # (module.GPUArray: Cuda/Kernels/Memory.py:84, Pool.py:120)


	@staticmethod
	def desc(shape, size, stride, pad):
		(fh, fw), (sh, sw), (ph, pw) = pair(size), pair(stride), pair(pad)
		n, c, h, w = shape
		return PoolDesc(n, c, h, w, fh, fw, sh, sw, ph, pw, PoolMode.max.value)


	def maxpool2d(self, data, size, stride, pad, allocator=None):
		assert data.dtype == np.float32 and data.ndim == 4
		requireF32(data)
		desc = self.desc(data.shape, size, stride, pad)
		p, q = c_int(0), c_int(0)
		lib.pz_pool2d_out_shape(byref(desc), byref(p), byref(q))
		shape = data.shape[:2] + (p.value, q.value)
		outdata = GPUArray.empty(shape, dtype=np.float32, allocator=allocator)
		mask = GPUArray.empty(shape, dtype=np.int32, allocator=allocator)
		lib.pz_maskpool2d_fwd(byref(desc), data.rptr, outdata.optr, mask.optr, None)
		return outdata, mask


	def maxpool2dBackward(self, grad, origshape, mask, size, stride, pad, allocator=None):
		assert grad.dtype == np.float32 and mask.dtype == np.int32
		requireF32(grad)
		desc = self.desc(tuple(grad.shape[:2]) + tuple(origshape[2:]), size, stride, pad)
		ingrad = GPUArray.empty(tuple(grad.shape[:2]) + tuple(origshape[2:]), dtype=np.float32, allocator=allocator)
		lib.pz_maskpool2d_bwd(byref(desc), grad.rptr, mask.rptr, ingrad.optr, None)
		return ingrad


	def maxunpool2d(self, data, origshape, mask, allocator=None):
		assert data.dtype == np.float32 and mask.dtype == np.int32
		requireF32(data)
		n, c, inh, inw = data.shape
		outh, outw = origshape[2], origshape[3]
		outdata = GPUArray.empty((n, c, outh, outw), dtype=np.float32, allocator=allocator)
		lib.pz_maxunpool2d_fwd(data.rptr, mask.rptr, outdata.optr, n * c, inh * inw, outh * outw, None)
		return outdata


	def maxunpool2dBackward(self, grad, poolshape, mask, allocator=None):
		assert grad.dtype == np.float32 and mask.dtype == np.int32
		requireF32(grad)
		n, c, outh, outw = grad.shape
		inh, inw = poolshape[2], poolshape[3]
		ingrad = GPUArray.empty((n, c, inh, inw), dtype=np.float32, allocator=allocator)
		lib.pz_maxunpool2d_bwd(grad.rptr, mask.rptr, ingrad.optr, n * c, inh * inw, outh * outw, None)
		return ingrad


class StubModule:
	def __init__(self, name):
		self.stubName = name

	def __getattr__(self, item):
		def raiser(*args, **kwargs):
			raise NotImplementedError("%s.%s is outside the implemented operator path" % (self.stubName, item))
		return raiser


class PointwiseCost:
	"""bceKer / hingeKer / smoothL1Ker / l1HingeKer — direct callables with the reference's argument lists
	(Cuda/Kernels/Costs.py:8-72; callers Cost/BCE.py:20, Hinge.py, SmoothL1.py, L1Hinge.py):
	  bceKer(scores, labels, totalError, grad, numsamples, spatialDim)        hingeKer(scores, labels, totalError, grad, numsamples, numcases)
	  smoothL1Ker(pred, target, totalError, grad, norm, fullnorm)             l1HingeKer(x1, x2, labels, totalError, g1, g2, numsamples, numcases)
	The error is ADDED to totalError (the reference's kernels atomicAdd into it; the cost modules zero it first)."""

	def __init__(self, kind, name):
		self.kind, self.name = kind, name

	def __call__(self, *args, slice=None, stream=None, allocator=None):
		assert slice is None, "%s takes whole tensors" % self.name
		kind = self.kind
		grad2 = labels = other = None
		norm = fullnorm = 0.0
		numsamples = numcases = 1
		if kind in (lib.COST_BCE, lib.COST_HINGE):
			a, labels, error, grad, numsamples, numcases = args
			assert labels.dtype == np.int32 and labels.size == a.size
		elif kind == lib.COST_SMOOTH_L1:
			a, other, error, grad, norm, fullnorm = args
			assert other.dtype == np.float32 and other.size == a.size
		else:
			a, other, labels, error, grad, grad2, numsamples, numcases = args
			assert labels.dtype == np.int32 and other.size == a.size and grad2.size == a.size
			assert labels.size * int(numcases) == a.size
		requireF32(a, error, grad)
		assert grad.size == a.size
		terms = GPUArray.empty((a.size, ), dtype=np.float32, allocator=allocator)
		lib.pz_cost_pointwise(
			kind, a.rptr, rptrOf(other), rptrOf(labels), error.wptr, grad.optr,
			None if grad2 is None else grad2.optr, terms.optr, a.size, int(numsamples), int(numcases), float(norm), float(fullnorm),
			streamHandle(stream)
		)


class PReluModule:
	"""prelu / preluBackwardData / preluBackwardParams — Cuda/Kernels/PRelu.py:58-133"""

	def __init__(self, matmod):
		self.matmod, self.backend, self.GPUArray = matmod, matmod.backend, GPUArray


	@staticmethod
	def geometry(data, slopes, sharedMaps):
		assert slopes.shape == (1, ) if sharedMaps else data.shape[1] == slopes.shape[0]
		return data.shape[0], data.shape[1], prod(data.shape[2:])


	def prelu(self, data, slopes, inplace=False, sharedMaps=False, allocator=None):
		requireF32(data, slopes)
		n, maps, mapsize = self.geometry(data, slopes, sharedMaps)
		outdata = data if inplace else GPUArray.empty(data.shape, dtype=np.float32, allocator=allocator)
		lib.pz_prelu_fwd(data.rptr, slopes.rptr, outdata.wptr if inplace else outdata.optr, n, maps, mapsize, int(sharedMaps), None)
		return outdata


	def preluBackwardData(self, grad, slopes, indata, sharedMaps=False, allocator=None):
		requireF32(grad, slopes, indata)
		assert grad.shape == indata.shape
		n, maps, mapsize = self.geometry(grad, slopes, sharedMaps)
		ingrad = GPUArray.empty(grad.shape, dtype=np.float32, allocator=allocator)
		lib.pz_prelu_bwd_data(grad.rptr, slopes.rptr, indata.rptr, ingrad.optr, n, maps, mapsize, int(sharedMaps), None)
		return ingrad


	def preluBackwardParams(self, indata, outgrad, sharedMaps=False, allocator=None):
		requireF32(indata, outgrad)
		assert indata.shape == outgrad.shape
		n, maps, mapsize = indata.shape[0], indata.shape[1], prod(indata.shape[2:])
		permap = GPUArray.empty((maps, ), dtype=np.float32, allocator=allocator)
		lib.pz_prelu_bwd_params(indata.rptr, outgrad.rptr, permap.optr, n, maps, mapsize, None)
		return self.matmod.matsum(permap.reshape(1, maps), axis=1, allocator=allocator) if sharedMaps else permap


class PadModule:
	"""reflectpad / reflectpadBackward — Cuda/Kernels/Pad.py:146-230 (3-d tensors pad the last axis with (l, r), 4-d
	tensors the last two with (u, b, l, r))"""

	def __init__(self, backend):
		self.backend, self.GPUArray = backend, GPUArray


	def reflectpad(self, data, pad, allocator=None):
		requireF32(data)
		if data.ndim == 3:
			(n, maps, inw), inh, (upad, bpad, lpad, rpad) = data.shape, 1, (0, 0) + tuple(pad)
			assert inw >= max(lpad, rpad) + 1
			outshape = (n, maps, inw + lpad + rpad)
		elif data.ndim == 4:
			(n, maps, inh, inw), (upad, bpad, lpad, rpad) = data.shape, pad
			assert inh >= max(upad, bpad) + 1 and inw >= max(lpad, rpad) + 1
			outshape = (n, maps, inh + upad + bpad, inw + lpad + rpad)
		else:
			raise NotImplementedError(data.ndim)
		outdata = GPUArray.empty(outshape, dtype=data.dtype, allocator=allocator)
		lib.pz_reflectpad2d_fwd(data.rptr, outdata.optr, n * maps, inh, inw, upad, bpad, lpad, rpad, None)
		return outdata


	def reflectpadBackward(self, grad, pad, allocator=None):
		requireF32(grad)
		if grad.ndim == 3:
			(n, maps, outw), (upad, bpad, lpad, rpad) = grad.shape, (0, 0) + tuple(pad)
			inh, inw = 1, outw - lpad - rpad
			inshape = (n, maps, inw)
		elif grad.ndim == 4:
			(n, maps, outh, outw), (upad, bpad, lpad, rpad) = grad.shape, pad
			inh, inw = outh - upad - bpad, outw - lpad - rpad
			inshape = (n, maps, inh, inw)
		else:
			raise NotImplementedError(grad.ndim)
		ingrad = GPUArray.empty(inshape, dtype=grad.dtype, allocator=allocator)
		lib.pz_reflectpad2d_bwd(grad.rptr, ingrad.optr, n * maps, inh, inw, upad, bpad, lpad, rpad, None)
		return ingrad


class UpsampleModule:
	"""upsample2d / upsample3d (+Backward), modes "nearest" and "linear" — Cuda/Kernels/Upsample.py:301-455"""

	def __init__(self, backend):
		self.backend, self.GPUArray = backend, GPUArray


	@staticmethod
	def linearFlag(mode):
		if mode not in ("nearest", "linear"):
			raise NotImplementedError(mode)
		return int(mode == "linear")


	def run(self, data, scale, mode, allocator, nd, backward):
		requireF32(data)
		assert data.ndim == nd + 2
		scales = (int(scale), ) * nd if isinstance(scale, (int, np.integer)) else tuple(int(v) for v in scale)
		sd, sh, sw = ((1, ) + scales) if nd == 2 else scales
		dims = ((1, ) + tuple(data.shape[2:])) if nd == 2 else tuple(data.shape[2:])
		n, maps = data.shape[:2]
		if backward:
			ind, inh, inw = dims[0] // sd, dims[1] // sh, dims[2] // sw
			outshape = (n, maps) + ((inh, inw) if nd == 2 else (ind, inh, inw))
			out = GPUArray.empty(outshape, dtype=data.dtype, allocator=allocator)
			lib.pz_upsample_bwd(data.rptr, out.optr, n * maps, ind, inh, inw, sd, sh, sw, self.linearFlag(mode), None)
		else:
			ind, inh, inw = dims
			outshape = (n, maps) + ((inh * sh, inw * sw) if nd == 2 else (ind * sd, inh * sh, inw * sw))
			out = GPUArray.empty(outshape, dtype=data.dtype, allocator=allocator)
			lib.pz_upsample_fwd(data.rptr, out.optr, n * maps, ind, inh, inw, sd, sh, sw, self.linearFlag(mode), None)
		return out


	def upsample2d(self, data, scale, mode="nearest", allocator=None):
		return self.run(data, scale, mode, allocator, 2, False)

	def upsample2dBackward(self, grad, scale, mode="nearest", allocator=None):
		return self.run(grad, scale, mode, allocator, 2, True)

	def upsample3d(self, data, scale, mode="nearest", allocator=None):
		return self.run(data, scale, mode, allocator, 3, False)

	def upsample3dBackward(self, grad, scale, mode="nearest", allocator=None):
		return self.run(grad, scale, mode, allocator, 3, True)


class CTCModule:
	"""ctcLoss — Cuda/Kernels/CTC.py:232-270 (Backend/Kernels/Costs.py:68-69 -> Cost/CTC.py:23-30)"""

	def __init__(self, backend):
		self.backend, self.GPUArray, self.dnn = backend, GPUArray, backend.dnn


	def ctcLoss(self, data, datalen, labels, lengths, blank, error=None, normalized=False, returnAlphas=False, allocator=None):
		requireF32(data)
		assert data.ndim == 3 and datalen.dtype == np.int32 and labels.dtype == np.int32
		T, batchsize, vocabsize = data.shape
		lengths = np.asarray(lengths, dtype=np.int32)
		assert lengths.shape == (batchsize, ) and datalen.size == batchsize

		if not normalized:
			data = self.dnn.softmaxNd(data.reshape(T * batchsize, vocabsize, 1, 1), allocator=allocator).reshape(
				T, batchsize, vocabsize
			)

		offsets = np.zeros(batchsize + 1, dtype=np.int32)
		offsets[1:] = np.cumsum(lengths, dtype=np.int32)
		total = int(offsets[-1])

		# positions of every sample's extended label sequence grouped by label (stable: ascending position inside a group) —
		# the reference sorts inside its kernel; the label lengths are host data in its API and the labels follow them here
		hostLabels = labels.get()
		order = np.empty(2 * total + batchsize, dtype=np.int32)
		segStart, segLabel, segOff = [], [], np.zeros(batchsize + 1, dtype=np.int32)
		for b in range(batchsize):
			L = int(lengths[b])
			ext = np.full(2 * L + 1, blank, dtype=np.int32)
			ext[1::2] = hostLabels[offsets[b]:offsets[b] + L]
			by = np.argsort(ext, kind="stable").astype(np.int32)
			order[2 * offsets[b] + b:2 * offsets[b] + b + 2 * L + 1] = by
			keys = ext[by]
			starts = np.flatnonzero(np.concatenate(([True], keys[1:] != keys[:-1]))).astype(np.int32)
			segStart.append(np.concatenate((starts, [2 * L + 1])).astype(np.int32))
			segLabel.append(keys[starts])
			segOff[b + 1] = segOff[b] + starts.size

		toGpu = lambda a: GPUArray.toGpu(np.ascontiguousarray(a, dtype=np.int32), allocator=allocator)
		alphas = GPUArray.empty((T * (2 * total + batchsize), ), dtype=np.float32, allocator=allocator)
		nll = GPUArray.empty((batchsize, ), dtype=np.float32, allocator=allocator)
		error = GPUArray.zeros((), dtype=np.float32, allocator=allocator) if error is None else error
		grad = GPUArray.zeros(data.shape, dtype=np.float32, allocator=allocator)

		# (the index tables stay referenced until the launch is queued: a temporary would go back to the pool — and to the
		# next table — before the call)
		tables = [toGpu(a) for a in (offsets, order, np.concatenate(segStart), np.concatenate(segLabel), segOff)]
		lib.pz_ctc_loss(
			data.rptr, datalen.rptr, labels.rptr, tables[0].rptr, tables[1].rptr, tables[2].rptr, tables[3].rptr, tables[4].rptr,
			T, batchsize, vocabsize, int(blank), int(2 * lengths.max() + 1), alphas.optr, nll.optr, grad.wptr, error.wptr, None
		)
		return (error, grad) if not returnAlphas else (error, grad, alphas)


class EmbedModule:
	"""embed / embedBackwardParams — Cuda/Kernels/Embedder.py:57-88 (word index -1: padding)"""

	def __init__(self, backend):
		self.backend, self.GPUArray = backend, GPUArray


	def embed(self, data, W, allocator=None):
		assert data.dtype == np.int32 and data.ndim == 2 and W.ndim == 2
		requireF32(W)
		batchsize, sentlen = data.shape
		embsize = W.shape[1]
		outdata = GPUArray.empty((batchsize, sentlen, embsize), dtype=W.dtype, allocator=allocator)
		lib.pz_embed_fwd(data.rptr, W.rptr, outdata.optr, batchsize * sentlen, embsize, None)
		return outdata


	def embedBackwardParams(self, indata, grad, W, scale):
		assert indata.shape == grad.shape[:2] and W.shape[1] == grad.shape[2]
		assert indata.dtype == np.int32
		requireF32(grad, W)
		lib.pz_embed_bwd_params(indata.rptr, grad.rptr, W.wptr, float(scale), indata.size, W.shape[1], None)
