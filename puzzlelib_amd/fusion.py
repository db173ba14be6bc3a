"""
The tensor descriptions (lazy.Thunk subclasses) this backend keeps instead of writing a tensor, and the fused kernels
each of them settles into. See lazy.py for the barrier machinery; DnnContext / the element-wise kernel objects in
backend.py create and consume these.

  description            created by (reference call, unchanged signature)            settled by
  --------------------------------------------------------------------------------------------------------------------
  lazy.Zero              GPUArray.fill(0) on a fresh tensor (Modules/Add.py:18,       memset
                         Replicate.py:25)
  BnApply(x, {a,b})      batchNormNd in training mode (Modules/BatchNormND.py:60-72)  pz_bn_apply_add (1 operand)
    .relu                + reluKer(y, y) in place (Modules/Activation.py:52-60)
  Sum(terms)             toVectorAddVector onto a Zero / Sum (Modules/Add.py:20-22)   pz_bn_apply_add[_mask] when a term is
    .relu                + reluKer in place                                           a BnApply, add3 / add3Relu / add3Gate,
    .gate = y            + reluDerKer(g, g, y) in place (Activation.py:62-70)         pz_bn_gate_stats[_up2] when the reader
                                                                                      is batchNormNdBackward
  Gate(y)                reluDerKer(g, g, y) in place on a written tensor             reluDer kernel; pz_bn_bwd_gate when
                                                                                      the reader is this BN's backward
  BnBwdApply(dy,x,{ABC}) batchNormNdBackward whose statistics are already known       pz_bn_bwd_apply_coef; folded into
                                                                                      pz_conv2d_bwd_{data,filter}_bn
  Up2(compact)           convNdBackwardData of a stride-2 pointwise convolution       zero fill + strided copy; folded into
                                                                                      pz_bn_gate_stats_up2
  ConvFwd(x, W, b)       convNd with a bias and no BatchNorm behind it (Conv2D ->   one convolution launch; + reluKer: the ReLU
    .relu                Activation(relu), TestLib/CnnCifar10NIN.py:16-45)            in its epilogue (pz_conv2d_fwd_relu)
  ConvBwdData(dy, W)     convNdBackwardData into the output of such a ReLU           one launch; + reluDerKer: the gate in its
    .gate = y                                                                         epilogue (pz_conv2d_bwd_data_gate)
  Scaled(1/N)            nodeinfo.sumTensor on the gradient arena (the mean of the    linear kernel in place; folded into the
                         data-parallel exchange, Optimizers/Optimizer.py:166-167)     Adam / momentum-SGD update kernels
"""
import ctypes
from ctypes import byref, c_size_t

import numpy as np

from puzzlelib_amd import lib, lazy
from puzzlelib_amd.lazy import Thunk
from puzzlelib_amd.gpuarray import GPUArray, prod


def raw(ary):
	"""address of an array that is known to be settled and private to a description (coefficients, partial sums)"""
	return ary.gpudata.ptr


def nchw(shape):
	return shape[0], shape[1], prod(shape[2:])


def eltwiseRaw(op, out, ins):
	ptrs = (ctypes.c_void_p * (1 + len(ins)))(out.gpudata.ptr, *[a.rptr for a in ins])
	lib.pz_eltwise(op, out.size, ptrs, 1 + len(ins), None, 0, 0, out.size, 1, None)


class BnApply(Thunk):
	"""y = a*x + b per channel, optionally through ReLU; {a, b} = `coef` (c, 2), private to the description."""

	def __init__(self, x, coef, relu=False):
		self.x, self.coef, self.relu = x, coef, relu

	def inputs(self):
		return (self.x, )

	def run(self, out):
		n, c, hw = nchw(self.x.shape)
		lib.pz_bn_apply_add(self.x.rptr, raw(self.coef), None, None, out.gpudata.ptr, n, c, hw, int(self.relu), None)
		lazy.count("bn_apply_relu" if self.relu else "bn_apply")
		return {"bnapply": (self.x, self.coef, self.relu)}


class Up2(Thunk):
	"""Zero except at pixels (2i, 2j), which hold `compact` (n, c, ceil(h/2), ceil(w/2))."""

	def __init__(self, compact):
		self.compact = compact

	def inputs(self):
		return (self.compact, )

	def run(self, out):
		lib.pz_memset_d32(out.gpudata.ptr, 0, out.size, None)
		view = out[:, :, ::2, ::2]
		shape = (ctypes.c_int64 * 4)(*view.shape)
		lib.pz_strided_copy(
			view.gpudata.ptr, view.elemStrides(), self.compact.rptr, self.compact.elemStrides(), shape, 4, None
		)
		lazy.count("up2_expand")


class BnBwdApply(Thunk):
	"""dx = A*dy + (B*x + C) per channel; {A, B, C, -} = `coef` (c, 4), private."""

	def __init__(self, dy, x, coef):
		self.dy, self.x, self.coef = dy, x, coef

	def inputs(self):
		return (self.dy, self.x)

	def run(self, out):
		n, c, hw = nchw(self.x.shape)
		lib.pz_bn_bwd_apply_coef(self.x.rptr, self.dy.rptr, out.gpudata.ptr, n, c, hw, raw(self.coef), None)
		lazy.count("bn_bwd_apply")


class Gate(Thunk):
	"""In place: the buffer holds g, its value is g * (y > 0). `parts`: (y, x, coef, mean, partials) when the backward-data launch
	that wrote g already summed {q, q (x - mean)} of the gated gradient for the BatchNorm y = relu(coef.a x + coef.b) came out of
	(dnn.convNdBackwardData, pz_conv2d_bwd_data_bnstats) — that BatchNorm's backward then needs no statistics pass."""

	def __init__(self, y, parts=None):
		self.y, self.parts = y, parts

	def inputs(self):
		return (self.y, ) if self.parts is None else (self.y, self.parts[1], self.parts[3])

	def run(self, out):
		ptrs = (ctypes.c_void_p * 3)(out.gpudata.ptr, out.gpudata.ptr, self.y.rptr)
		lib.pz_eltwise(lib.OP_RELU_DER, out.size, ptrs, 3, None, 0, 0, out.size, 1, None)
		lazy.count("gate")


class ConvFwd(Thunk):
	"""y = conv(x, W) + b, optionally through ReLU (x * (x > 0), the element-wise kernel's form) in the kernel's epilogue.
	Modules/Activation.py:52-55 is out of place by default: the ReLU's output then gets the activated launch and THIS
	description stays on the convolution's own output, which nobody reads in a training step — `detach()` moves it onto a
	snapshot of the parameters so that the optimizer's update does not have to materialise it."""

	def __init__(self, dnn, desc, algo, x, W, bias, relu=False):
		self.dnn, self.desc, self.algo, self.x, self.W, self.bias, self.relu = dnn, desc, algo, x, W, bias, relu
		self.detached = False

	def inputs(self):
		return (self.x, self.W, self.bias)

	def dependsOn(self, root):
		return any(t.gpudata.root is root for t in self.inputs())

	def twin(self, relu):
		return ConvFwd(self.dnn, self.desc, self.algo, self.x, self.W, self.bias, relu)

	def detach(self):
		self.W, self.bias, self.detached = lazy.snapshot(self.W), lazy.snapshot(self.bias), True

	def run(self, out):
		self.dnn.launchForward(self.desc, self.algo, self.x, self.W, self.bias, out, self.relu, prepared=not self.detached, settling=True)
		lazy.count("conv_relu" if self.relu else "conv_deferred")
		return {"convrelu": True} if self.relu else None


class ConvBwdData(Thunk):
	"""dx = bwd_data(dy, W), optionally gated by `gate` > 0 (reluDerKer's form g * (y > 0)) in the kernel's epilogue; see ConvFwd
	for `detach`."""

	def __init__(self, dnn, desc, algo, dy, W, gate=None):
		self.dnn, self.desc, self.algo, self.dy, self.W, self.gate = dnn, desc, algo, dy, W, gate
		self.detached = False

	def inputs(self):
		return (self.dy, self.W) + ((self.gate, ) if self.gate is not None else ())

	def dependsOn(self, root):
		return any(t.gpudata.root is root for t in self.inputs())

	def twin(self, gate):
		return ConvBwdData(self.dnn, self.desc, self.algo, self.dy, self.W, gate)

	def detach(self):
		self.W, self.detached = lazy.snapshot(self.W), True

	def run(self, out):
		self.dnn.launchBackwardData(self.desc, self.algo, self.dy, self.W, out, self.gate, prepared=not self.detached, settling=True)
		lazy.count("dgrad_gate" if self.gate is not None else "dgrad_deferred")


class Scaled(Thunk):
	"""In place: the buffer holds g (the sum of the ranks' gradients), its value is g * scale."""

	def __init__(self, scale):
		self.scale = float(scale)

	def run(self, out):
		ptrs = (ctypes.c_void_p * 2)(out.gpudata.ptr, out.gpudata.ptr)
		words = (ctypes.c_float * 2)(self.scale, 0.0)
		lib.pz_eltwise(lib.OP_LINEAR, out.size, ptrs, 2, words, 2, 0, out.size, 1, None)
		lazy.count("grad_scale_pass")


class Sum(Thunk):
	"""[relu](t0 + t1 + ...) [* (gate > 0)]. A term is ("arr", GPUArray) | ("bn", x, coef) | ("up2", compact); summed left
	to right exactly as the axpy sequence it stands for would (0 + t0 is t0)."""

	def __init__(self):
		self.terms, self.relu, self.gate = [], False, None

	def inputs(self):
		return tuple(t[1] for t in self.terms) + ((self.gate, ) if self.gate is not None else ())

	# ---- helpers
	@staticmethod
	def dense(term, like):
		"""a term as a written tensor"""
		if term[0] == "arr":
			return term[1]
		tmp = GPUArray.empty(like.shape, dtype=like.dtype)
		if term[0] == "bn":
			BnApply(term[1], term[2]).run(tmp)
		else:
			Up2(term[1]).run(tmp)
		return tmp

	def maskFor(self, out):
		"""the sign mask of the gate tensor, if the kernel that produced it left one"""
		if self.gate is None or not lazy.on("mask"):
			return None
		self.gate.rptr                                       # (a gate tensor that is itself still described gets written now)
		bits = lazy.fact(self.gate, "relumask")
		return bits if bits is not None and self.gate.shape == out.shape else None

	# ---- settle
	def run(self, out, statsFor=None):
		"""`statsFor` = [(bnInput, savemean), ...] (at most 2): also return the partial sums of those BatchNorm backwards
		over the produced gradient (only from batchNormNdBackward, only for gated two-term sums)."""
		terms, n = self.terms, len(self.terms)
		kinds = [t[0] for t in terms]
		optr = out.gpudata.ptr
		facts = None

		if self.gate is not None and n == 2 and statsFor and out.ndim == 4:
			return self.runGateStats(out, statsFor)

		if self.gate is None and n in (1, 2) and "bn" in kinds and out.ndim >= 2:
			if kinds[0] != "bn":                       # a + b == b + a
				terms = [terms[1], terms[0]]
			first, second = terms[0], (terms[1] if n == 2 else None)
			x2 = coef2 = None
			if second is not None:
				if second[0] == "bn":
					x2, coef2 = second[1], second[2]
				else:
					x2 = self.dense(second, out)
			nn_, c, hw = nchw(first[1].shape)
			if self.relu and lazy.on("mask") and n == 2:
				size = c_size_t(0)
				lib.pz_relu_mask_bytes(nn_, c, hw, byref(size))
				bits = GPUArray.empty((size.value, ), dtype=np.uint8)
				lib.pz_bn_apply_add_mask(
					first[1].rptr, raw(first[2]), x2.rptr, None if coef2 is None else raw(coef2), optr, raw(bits), nn_, c, hw, 1,
					None
				)
				facts = {"relumask": bits}
			else:
				lib.pz_bn_apply_add(
					first[1].rptr, raw(first[2]), None if x2 is None else x2.rptr, None if coef2 is None else raw(coef2), optr,
					nn_, c, hw, int(self.relu), None
				)
			lazy.count("bn_apply_add")
			# which BatchNorm inputs this tensor was summed from: their backward passes will read one common gradient
			facts = dict(facts or {}, bnterms=[t[1] for t in terms if t[0] == "bn"])
			return facts

		if n == 2 and (self.gate is None or not self.relu):
			a, b = self.dense(terms[0], out), self.dense(terms[1], out)
			if self.gate is not None:
				eltwiseRaw(lib.OP_ADD3_GATE, out, (a, b, self.gate))
				lazy.count("add3_gate")
			else:
				eltwiseRaw(lib.OP_ADD3_RELU if self.relu else lib.OP_ADD3, out, (a, b))
				lazy.count("add3_relu" if self.relu else "add3")
			return None

		# anything else: the literal sequence (memset, one axpy per term, ReLU, gate)
		lib.pz_memset_d32(optr, 0, out.size, None)
		one = np.ones(1, dtype=np.float32).ctypes.data_as(ctypes.POINTER(ctypes.c_float))
		for term in terms:
			src = self.dense(term, out)
			ptrs = (ctypes.c_void_p * 2)(optr, src.rptr)
			lib.pz_eltwise(lib.OP_AXPY, out.size, ptrs, 2, one, 1, 0, out.size, 1, None)
		if self.relu:
			ptrs = (ctypes.c_void_p * 2)(optr, optr)
			lib.pz_eltwise(lib.OP_RELU, out.size, ptrs, 2, None, 0, 0, out.size, 1, None)
		if self.gate is not None:
			ptrs = (ctypes.c_void_p * 3)(optr, optr, self.gate.rptr)
			lib.pz_eltwise(lib.OP_RELU_DER, out.size, ptrs, 3, None, 0, 0, out.size, 1, None)
		lazy.count("sum_generic")
		return None

	def runGateStats(self, out, statsFor):
		"""g = (t0 + t1) * (gate > 0) and, in the same pass, the backward statistics of the BatchNorm(s) g goes to."""
		terms = self.terms
		n, c, hw = nchw(out.shape)
		size = c_size_t(0)
		lib.pz_bn_workspace_bytes(n, c, hw, byref(size))
		parts = [GPUArray.empty((size.value // 4, ), dtype=np.float32) for _ in statsFor]

		(xa, ma), (xb, mb) = statsFor[0], (statsFor[1] if len(statsFor) == 2 else (None, None))
		bits = self.maskFor(out)
		yptr = self.gate.rptr if bits is None else None        # with the bits the gate tensor itself is not read
		if yptr is None:
			yptr = self.gate.gpudata.ptr
		mptr = None if bits is None else raw(bits)
		pb = None if xb is None else raw(parts[1])

		up2 = terms[0][0] == "up2" and terms[1][0] == "up2" and lazy.on("up2")
		if up2:
			lib.pz_bn_gate_stats_up2(
				terms[0][1].rptr, terms[1][1].rptr, yptr, mptr, out.gpudata.ptr, n, c, out.shape[2], out.shape[3],
				xa.rptr, ma.rptr, raw(parts[0]), None if xb is None else xb.rptr, None if mb is None else mb.rptr, pb, None
			)
			lazy.count("gate_stats_up2")
		else:
			a, b = self.dense(terms[0], out), self.dense(terms[1], out)
			lib.pz_bn_gate_stats(
				a.rptr, b.rptr, yptr, mptr, out.gpudata.ptr, n, c, hw, xa.rptr, ma.rptr, raw(parts[0]),
				None if xb is None else xb.rptr, None if mb is None else mb.rptr, pb, None
			)
			lazy.count("gate_stats")
		if bits is not None:
			lazy.count("gate_by_mask")
		return {"bwdparts": [(x, mean, part) for (x, mean), part in zip(statsFor, parts)]}


def settleWithStats(grad, statsFor):
	"""Settles a pending gated two-term Sum on `grad` together with the BatchNorm-backward statistics of `statsFor`;
	returns the partial sums [(x, savemean, partials), ...] or None if `grad` is not such a tensor."""
	thunk = lazy.pending(grad, Sum)
	if thunk is None or thunk.gate is None or len(thunk.terms) != 2 or thunk.relu or not lazy.on("gatestats"):
		return None
	root = grad.gpudata.root
	lz = root.lz
	lz.thunk = None
	out = GPUArray(thunk.shape, thunk.dtype, gpudata=root)
	facts = thunk.run(out, statsFor=statsFor)
	lz.meta = dict(lz.meta or {})
	lz.meta.update(facts)
	for (x, mean) in statsFor:
		lazy.depend(x, root)
	return facts["bwdparts"]
