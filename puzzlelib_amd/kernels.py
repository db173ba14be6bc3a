"""
The element-wise kernel objects of the backend (`reluKer(dtype)(out, x)`, `adamKer`, `addKer`, ... — Backend/Kernels/
ElementWise.py reads them off the backend object; originals: Cuda/Kernels/ElementWise.py) and the element-wise half of the
fusion policy: which calls join a tensor's pending description instead of launching (absorbRelu / absorbReluDer /
absorbAxpy; fusion.Scaled consumed by the update rules). The DNN half is puzzlelib_amd/dnn.py.
"""
import os, weakref, sys, time, ctypes
from ctypes import byref, c_int, c_size_t, c_void_p

import numpy as np

from puzzlelib_amd import lib, driver, lazy, fusion
from puzzlelib_amd.lib import HipError, ConvDesc, PoolDesc
from puzzlelib_amd.driver import streamHandle
from puzzlelib_amd.gpuarray import GPUArray, prod, eltwise, contiguousStrides


def absorbRelu(arrays, scalars):
	"""reluKer(out, in): in place on a described tensor the ReLU joins the description (Modules/Activation.py:52-60 with
	inplace=True after BatchNorm2D or Add); out of place on a described BatchNorm output, `out` gets the description."""
	out, inp = arrays
	waiting = lazy.pending(inp)
	if waiting is None:
		return False

	if lazy.sameBuffer(out, inp):
		waiting = lazy.editable(inp)                         # in place = a write: readers by reference are settled first
		if waiting is None:
			return False
		if isinstance(waiting, lazy.Zero):
			return True                                      # relu(0) = 0
		if isinstance(waiting, fusion.BnApply) and not waiting.relu and lazy.on("bnrelu"):
			waiting.relu = True
			return True
		if isinstance(waiting, fusion.Sum) and not waiting.relu and waiting.gate is None and lazy.on("addrelu"):
			waiting.relu = True
			return True
		if isinstance(waiting, fusion.ConvFwd) and not waiting.relu and lazy.on("convrelu"):
			waiting.relu = True
			return True
		return False

	if isinstance(waiting, fusion.ConvFwd) and not waiting.relu and lazy.on("convrelu") and lazy.whole(out) and \
			out.shape == inp.shape and lazy.pending(out) is None:
		# out of place (Modules/Activation.py:52-55 default): ONE launch writes relu(conv) into `out`; the convolution's own
		# output keeps its description, moved onto a snapshot of the parameters (nobody reads it in a training step, and the
		# optimizer's update must not force it)
		out.optr
		facts = waiting.twin(True).run(out)
		lazy.setFact(out, "convrelu", facts["convrelu"])
		waiting.detach()
		return True

	if isinstance(waiting, fusion.BnApply) and not waiting.relu and lazy.on("bnrelu") and lazy.whole(out) and \
			out.shape == inp.shape and lazy.pending(out) is None:
		out.optr
		lazy.attach(out, fusion.BnApply(waiting.x, waiting.coef, relu=True))
		return True
	return False


def absorbReluDer(arrays, scalars):
	"""reluDerKer(ingrad, outgrad, outdata) in place (Modules/Activation.py:62-70, inplace=True): the gate (outdata > 0)
	joins a described fan-in, or becomes a description of its own on a written gradient — the batch-norm backward that
	reads it next applies it while loading."""
	ingrad, outgrad, outdata = arrays
	if not lazy.sameBuffer(ingrad, outgrad):
		# out of place: a backward-data launch that is still only described takes the gate into its epilogue and writes `ingrad`
		waiting = lazy.pending(outgrad, fusion.ConvBwdData) if lazy.on("convgate") else None
		if waiting is None or waiting.gate is not None or not lazy.whole(ingrad) or ingrad.shape != outgrad.shape or \
				ingrad.shape != outdata.shape or lazy.pending(ingrad) is not None or lazy.sameBuffer(ingrad, outdata):
			return False
		ingrad.optr
		waiting.twin(outdata).run(ingrad)
		waiting.detach()
		return True
	if not lazy.whole(ingrad) or ingrad.shape != outdata.shape or lazy.sameBuffer(ingrad, outdata):
		return False

	waiting = lazy.editable(ingrad)
	root = ingrad.gpudata.root
	if isinstance(waiting, fusion.ConvBwdData) and waiting.gate is None and lazy.on("convgate"):
		waiting.gate = outdata
		lazy.depend(outdata, root)
		return True
	if isinstance(waiting, fusion.Sum) and not waiting.relu and waiting.gate is None and lazy.on("addgate"):
		waiting.gate = outdata
		lazy.depend(outdata, root)
		return True

	if lazy.on("gate"):
		# sums the producing backward-data launch left for the BatchNorm behind this ReLU (a fact of the stored gradient: read
		# before the write barrier drops it) stay with the gate as long as it is THIS output the gate is made of
		parts = lazy.fact(ingrad, "gatedparts") if lazy.on("dgradstats") else None
		if parts is not None and not lazy.sameBuffer(parts[0], outdata):
			parts = None
		ingrad.wptr                                          # whatever is pending gets written; dependents are settled
		lazy.attach(ingrad, fusion.Gate(outdata, parts))
		return True
	return False


def absorbAxpy(arrays, scalars):
	"""toVectorAddVectorKer(y, x, alpha) with alpha == 1 onto a zero-filled / summed accumulator (Modules/Add.py:20-22,
	Replicate.py:27-29): x becomes a term of y's description."""
	y, x = arrays
	if float(scalars[0]) != 1.0 or x.size != y.size or x.dtype != y.dtype or not lazy.on("sum"):
		return False
	waiting = lazy.editable(y)
	if not isinstance(waiting, (lazy.Zero, fusion.Sum)) or x.gpudata.root is y.gpudata.root:
		return False
	if isinstance(waiting, fusion.Sum) and (waiting.relu or waiting.gate is not None or len(waiting.terms) >= 4):
		return False

	root = y.gpudata.root
	if isinstance(waiting, lazy.Zero):
		total = fusion.Sum()
		total.shape, total.dtype = waiting.shape, waiting.dtype
		root.lz.thunk = waiting = total

	src = lazy.pending(x)
	if isinstance(src, fusion.BnApply) and not src.relu and lazy.on("bnadd") and len(waiting.shape) == 4 and \
			tuple(src.x.shape) == tuple(waiting.shape):
		term = ("bn", src.x, src.coef)
	elif isinstance(src, fusion.Up2) and lazy.on("up2") and len(waiting.shape) == 4:
		term = ("up2", src.compact)
	else:
		term = ("arr", x)

	waiting.terms.append(term)
	lazy.depend(term[1], root)
	return True


class EltwiseKernel:
	"""Callable with the launch signature of the reference kernel objects:
	ker(*arrays_then_scalars, slice=None, stream=None) — Cuda/SourceModule.py:203-226."""

	# optimizer updates write every array but the gradient (index 1); everything else writes its first array only
	writesAll = frozenset((
		lib.OP_ADAM, lib.OP_CLASSIC_MOM_SGD, lib.OP_NESTEROV_MOM_SGD, lib.OP_RMSPROP, lib.OP_ADAGRAD, lib.OP_ADADELTA,
		lib.OP_RMSPROP_GRAVES, lib.OP_SMORMS3
	))

	def __init__(self, op, narrays, nscalars, name, rawScalar=()):
		self.op, self.narrays, self.nscalars, self.name = op, narrays, nscalars, name
		self.rawScalar = rawScalar      # indices of scalars that are integers travelling as raw 32-bit words
		self.readonly = (1, ) if op in self.writesAll else tuple(range(1, narrays))
		# a call the lazy-buffer layer can absorb into a tensor's description returns without launching (fusion.py)
		self.absorb = {lib.OP_RELU: absorbRelu, lib.OP_RELU_DER: absorbReluDer, lib.OP_AXPY: absorbAxpy}.get(op, None)
		# update rules whose kernel takes the gradient's scale as a trailing scalar (csrc/eltwise.hip): a gradient arena that
		# is only described as "sum over ranks x 1/N" (fusion.Scaled) is consumed as it stands
		self.gradScale = op in (lib.OP_ADAM, lib.OP_CLASSIC_MOM_SGD, lib.OP_NESTEROV_MOM_SGD)


	def __call__(self, *args, **kwargs):
		if len(args) != self.narrays + self.nscalars:
			raise TypeError("%s expects %d arguments, got %d" % (self.name, self.narrays + self.nscalars, len(args)))

		arrays, scalars = args[:self.narrays], args[self.narrays:]
		for ary in arrays:
			if not ary.contiguous:
				raise ValueError("gpuarray is not contiguous")

		slc, stream = kwargs.get("slice", None), kwargs.get("stream", None)
		if self.absorb is not None and lazy.enabled and slc is None and stream is None and self.absorb(arrays, scalars):
			return

		words = np.empty(len(scalars) + int(self.gradScale), dtype=np.float32)
		for i, value in enumerate(scalars):
			if i in self.rawScalar:
				words.view(np.uint32)[i] = np.uint32(int(value))
			else:
				words[i] = value

		rawIdx = -1
		if self.gradScale:
			words[-1] = 1.0
			scaled = lazy.pending(arrays[1], fusion.Scaled) if (lazy.enabled and slc is None and stream is None) else None
			if scaled is not None and not any(lazy.sameBuffer(arrays[1], a) for a in (arrays[0], ) + tuple(arrays[2:])):
				glz = arrays[1].gpudata.root.lz
				if glz.small is not None:               # queued small accumulates into the arena are part of its stored value
					lazy.touchSmall(glz, arrays[1].gpudata, False)
				lazy.rawRead(arrays[1])                 # foreign-stream writers (the exchange was joined by sumTensor already)
				words[-1], rawIdx = scaled.scale, 1
				lazy.count("grad_scale_folded")

		eltwise(self.op, arrays[0].size, arrays, words, slc=slc, stream=stream, readonly=self.readonly, rawIdx=rawIdx)


def memoizedKernel(op, narrays, nscalars, name, rawScalar=()):
	"""`ker(dtype) -> callable` factories (the @memoize'd kernels of Cuda/Kernels/ElementWise.py)."""
	kernel = EltwiseKernel(op, narrays, nscalars, name, rawScalar)

	def factory(dtype):
		if np.dtype(dtype) != np.float32:
			raise NotImplementedError("%s: dtype %s (this backend computes in float32)" % (name, dtype))
		return kernel

	factory.__name__ = name
	return factory


class AddKernelFactory:
	"""addKer(dtype)(out, x, alpha, y, beta): out = alpha*x + beta*y — Cuda/Kernels/ElementWise.py:1017-1045
	(note the interleaved array/scalar argument order)."""

	def __call__(self, dtype):
		if np.dtype(dtype) != np.float32:
			raise NotImplementedError("addKer: dtype %s" % dtype)
		return self.launch

	@staticmethod
	def launch(out, x, alpha, y, beta, slice=None, stream=None):
		if slice is None and stream is None and 0 < out.size <= 4096 and out.size == x.size == y.size and lazy.on("smalladd") \
				and out.contiguous and x.contiguous and y.contiguous and x.dtype == y.dtype == out.dtype == np.float32:
			lazy.deferAdd(out, x, y, alpha, beta)            # runs with its neighbours in one launch (lazy.flushSmall)
			return
		eltwise(lib.OP_ADD, out.size, (out, x, y), np.array([alpha, beta], dtype=np.float32), slc=slice, stream=stream)
