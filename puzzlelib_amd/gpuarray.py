"""
GPUArray: the N-d strided device array type of the MI355X backend.

Keeps the attribute/method set of the reference type (C part Cuda/Source/Core/Array.c:1462-1512: shape, strides,
ptr, nbytes, dtype, contiguous, device, ndim, size, gpudata, reshape/view/ravel/get/set/copy/dimAt/strideAt,
empty/zeros/emptyLike/zerosLike/toGpu, __getitem__; Python part Cuda/GPUArray.py:20-262 -> Hip/GPUArray.py:9:
fill/astype/min/max/+/*/+=/*=/__setitem__), re-implemented over raw device pointers and the C ABI. Views share the
parent's Buffer; ops never retain their inputs. All arithmetic runs in HIP kernels (pz_eltwise / pz_reduce_*).
"""
import ctypes, math, os
import numpy as np

from puzzlelib_amd import lib, lazy
from puzzlelib_amd.driver import Buffer, Device, streamHandle


def prod(seq):
	out = 1
	for v in seq:
		out *= int(v)
	return out


def contiguousStrides(shape, itemsize):
	strides, acc = [], itemsize
	for dim in reversed(shape):
		strides.append(acc)
		acc *= max(int(dim), 1)
	return tuple(reversed(strides))


def viewStridesForReshape(oldshape, oldstrides, newshape):
	"""Strides of a no-copy reshape of a strided view, or None if the data would have to move
	(the rule GPUArray.reshape follows for non-contiguous views, Cuda/GPUArray.py:302-307 memoryTest)."""
	olddims = [(d, s) for d, s in zip(oldshape, oldstrides) if d != 1]
	newstrides = [0] * len(newshape)

	oi, oj, ni, nj = 0, 1, 0, 1
	nold, nnew = len(olddims), len(newshape)

	while ni < nnew and oi < nold:
		np_, op = newshape[ni], olddims[oi][0]

		while np_ != op:
			if np_ < op:
				np_ *= newshape[nj]
				nj += 1
			else:
				op *= olddims[oj][0]
				oj += 1

		for ok in range(oi, oj - 1):
			if olddims[ok][1] != olddims[ok + 1][0] * olddims[ok + 1][1]:
				return None

		newstrides[nj - 1] = olddims[oj - 1][1]
		for nk in range(nj - 1, ni, -1):
			newstrides[nk - 1] = newstrides[nk] * newshape[nk]

		ni, nj = nj, nj + 1
		oi, oj = oj, oj + 1

	last = newstrides[ni - 1] if ni >= 1 else 0
	for nk in range(ni, nnew):
		newstrides[nk] = last

	return tuple(newstrides)


DTYPES = {t: np.dtype(t) for t in (np.float32, np.int32, np.uint32, np.uint8, np.float64, np.int64, np.float16, np.int8)}
PTRS = {n: ctypes.c_void_p * n for n in range(1, 9)}
FLOATS = {n: ctypes.c_float * n for n in range(1, 9)}


class GPUArray:
	__slots__ = ["shape", "_strides", "dtype", "gpudata", "size", "ndim", "nbytes", "contiguous", "__weakref__"]

	defaultAllocator = None        # set by the backend: its memory pool
	debugFill = os.environ.get("PUZZLE_MI355_DEBUG_ALLOC", "0") == "1"     # setupDebugAllocator (Cuda/Utils.py:97-114)


	def __init__(self, shape, dtype, allocator=None, gpudata=None, strides=None):
		if isinstance(shape, (int, np.integer)):
			shape = (int(shape), )

		self.shape = shape = tuple(map(int, shape))
		self.dtype = dtype = dtype if type(dtype) is np.dtype else DTYPES.get(dtype) or np.dtype(dtype)
		self.ndim = len(shape)
		self.size = size = math.prod(shape)
		self.nbytes = size * dtype.itemsize

		if strides is None:                      # dense: the strides are derived on demand (most arrays never need them)
			self._strides, self.contiguous = None, True
		else:
			cstrides = contiguousStrides(self.shape, self.dtype.itemsize)
			self._strides = tuple(int(s) for s in strides)
			self.contiguous = self.size <= 1 or all(
				s == cs for s, cs, d in zip(self._strides, cstrides, self.shape) if d != 1
			)

		if gpudata is None:
			allocator = GPUArray.defaultAllocator if allocator is None else allocator
			gpudata = allocator.allocate(self.nbytes) if allocator is not None else Buffer.allocate(self.nbytes)
			if GPUArray.debugFill and self.nbytes >= 4:
				# the reference's debug allocator poisons fresh memory with NaNs so that reads of unwritten data show up
				# (Cuda/Utils.py:97-114, switched on by Unittester.py:52-55)
				lib.pz_memset_d32(gpudata.ptr, 0x7fc00000, self.nbytes // 4, None)

		elif self.contiguous and gpudata.size < self.nbytes:
			raise ValueError("gpudata buffer is too small (%d < %d bytes)" % (gpudata.size, self.nbytes))

		self.gpudata = gpudata


	# ------------------------------------------------------------------ properties
	@classmethod
	def dense(cls, shape, dtype, gpudata, size):
		"""a contiguous array over existing memory, arguments already normalised (the hot path of reshape / ravel)"""
		self = object.__new__(cls)
		self.shape, self.dtype, self.gpudata, self.size = shape, dtype, gpudata, size
		self.ndim, self.nbytes, self._strides, self.contiguous = len(shape), size * dtype.itemsize, None, True
		return self


	@property
	def strides(self):
		if self._strides is None:
			self._strides = contiguousStrides(self.shape, self.dtype.itemsize)
		return self._strides


	# Device addresses are handed out behind the lazy-buffer barriers (puzzlelib_amd/lazy.py): `rptr` for reading, `wptr`
	# for partial or read-modify-write access, `optr` when the whole array is about to be overwritten, `ptr` (the
	# reference attribute, Array.c:1462-1512) when the use is not known = read + write.
	@property
	def rptr(self):
		buf = self.gpudata
		root = buf.root
		if root.lz is not None:
			lazy.readBarrier(root, buf)
		return buf.ptr


	@property
	def wptr(self):
		buf = self.gpudata
		root = buf.root
		if root.lz is not None:
			lazy.writeBarrier(root, buf)
		return buf.ptr


	@property
	def optr(self):
		buf = self.gpudata
		root = buf.root
		if root.lz is not None:
			lazy.writeBarrier(root, buf, self.contiguous and buf.ptr == root.ptr and self.nbytes == root.size)
		return buf.ptr


	ptr = wptr


	def ptrOn(self, stream, write):
		"""Address for a launch on a foreign stream: pending contents are settled (on the main stream, which the foreign
		stream is made to follow), events of *other* streams are waited for by `stream`."""
		buf = self.gpudata
		root = buf.root
		if root.lz is not None:
			if write:
				lazy.writeBarrier(root, buf, False, stream)
			else:
				lazy.readBarrier(root, buf, stream)
		return buf.ptr


	@property
	def device(self):
		return Device.current


	def dimAt(self, index):
		return self.shape[index]


	def strideAt(self, index):
		return self.strides[index]


	# ------------------------------------------------------------------ constructors
	@staticmethod
	def empty(shape, dtype, allocator=None, gpudata=None):
		return GPUArray(shape, dtype, allocator=allocator, gpudata=gpudata)


	@staticmethod
	def zeros(shape, dtype, allocator=None, gpudata=None):
		ary = GPUArray(shape, dtype, allocator=allocator, gpudata=gpudata)
		if ary.nbytes > 0:
			ary.fill(0)
		return ary


	@staticmethod
	def emptyLike(ary, allocator=None):
		return GPUArray(ary.shape, ary.dtype, allocator=allocator)


	@staticmethod
	def zerosLike(ary, allocator=None):
		return GPUArray.zeros(ary.shape, ary.dtype, allocator=allocator)


	@staticmethod
	def toGpu(ary, allocator=None):
		ary = np.ascontiguousarray(ary)
		out = GPUArray(ary.shape, ary.dtype, allocator=allocator)
		out.set(ary)
		return out


	# ------------------------------------------------------------------ host <-> device
	def enforceContiguous(self):
		if not self.contiguous:
			raise ValueError("gpuarray is not contiguous")


	def enforceWordSized(self):
		if self.dtype.itemsize != 4:
			raise ValueError("strided access supports 4-byte element types only (got %s)" % self.dtype)


	def elemStrides(self):
		return (ctypes.c_int64 * max(self.ndim, 1))(*[s // self.dtype.itemsize for s in self.strides])


	def stridedCopyFrom(self, src, stream=None):
		self.enforceWordSized()
		shape = (ctypes.c_int64 * max(self.ndim, 1))(*self.shape)
		lib.pz_strided_copy(self.wptr, self.elemStrides(), src.rptr, src.elemStrides(), shape, self.ndim, streamHandle(stream))


	def get(self, stream=None):
		src = self if self.contiguous else self.copy()
		out = np.empty(self.shape, dtype=self.dtype)

		if self.nbytes > 0:
			lib.pz_memcpy_d2h(out.ctypes.data, src.rptr, self.nbytes, streamHandle(stream))
			lib.pz_stream_sync(streamHandle(stream))

		return out


	def set(self, ary, stream=None):
		if isinstance(ary, GPUArray):
			if ary.shape != self.shape or ary.dtype != self.dtype:
				raise ValueError("gpuarray shape/dtype mismatch in set (%s %s vs %s %s)" % (
					ary.shape, ary.dtype, self.shape, self.dtype
				))

			if self.contiguous and ary.contiguous:
				lib.pz_memcpy_d2d(self.optr, ary.rptr, self.nbytes, streamHandle(stream))
			else:
				self.stridedCopyFrom(ary, stream)
			return

		ary = np.asarray(ary)
		if ary.shape != self.shape:
			raise ValueError("array shape %s does not match gpuarray shape %s" % (ary.shape, self.shape))
		if ary.dtype != self.dtype:
			raise ValueError("array dtype %s does not match gpuarray dtype %s" % (ary.dtype, self.dtype))

		ary = np.ascontiguousarray(ary)
		if self.nbytes == 0:
			return

		if self.contiguous:
			lib.pz_memcpy_h2d(self.optr, ary.ctypes.data, self.nbytes, streamHandle(stream))
			lib.pz_stream_sync(streamHandle(stream))      # the host array may die right after this call
		else:
			self.stridedCopyFrom(GPUArray.toGpu(ary), stream)


	def copy(self, allocator=None):
		out = GPUArray(self.shape, self.dtype, allocator=allocator)
		out.set(self)
		return out


	# ------------------------------------------------------------------ views
	def reshape(self, *shape):
		if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
			shape = tuple(shape[0])

		shape = [int(d) for d in shape]
		if shape.count(-1) > 1:
			raise ValueError("can only specify one unknown dimension")
		if -1 in shape:
			known = prod(d for d in shape if d != -1)
			shape[shape.index(-1)] = self.size // known if known else 0

		shape = tuple(shape)
		if prod(shape) != self.size:
			raise ValueError("cannot reshape array of size %d into shape %s" % (self.size, shape))

		if self.contiguous:
			return GPUArray.dense(shape, self.dtype, self.gpudata, self.size)

		strides = viewStridesForReshape(self.shape, self.strides, shape)
		if strides is None:
			raise ValueError("gpuarray view cannot be reshaped without a copy")

		return GPUArray(shape, self.dtype, gpudata=self.gpudata, strides=strides)


	def ravel(self):
		if self.contiguous:
			return GPUArray.dense((self.size, ), self.dtype, self.gpudata, self.size)
		return self.reshape(self.size)


	def view(self, dtype):
		dtype = np.dtype(dtype)
		self.enforceContiguous()

		if self.ndim == 0 or (self.shape[-1] * self.dtype.itemsize) % dtype.itemsize != 0:
			raise ValueError("gpuarray cannot be viewed as %s" % dtype)

		shape = self.shape[:-1] + (self.shape[-1] * self.dtype.itemsize // dtype.itemsize, )
		return GPUArray(shape, dtype, gpudata=self.gpudata)


	def __getitem__(self, item):
		if not isinstance(item, tuple):
			item = (item, )
		if len(item) > self.ndim:
			raise IndexError("too many indices for gpuarray")

		offset, shape, strides = 0, [], []

		for axis, idx in enumerate(item):
			dim, stride = self.shape[axis], self.strides[axis]

			if isinstance(idx, (int, np.integer)):
				idx = int(idx)
				if idx < 0:
					idx += dim
				if not 0 <= idx < dim:
					raise IndexError("index %d is out of bounds for axis %d with size %d" % (idx, axis, dim))
				offset += idx * stride

			elif isinstance(idx, slice):
				start, stop, step = idx.indices(dim)
				if step <= 0:
					raise IndexError("gpuarray slices need a positive step")
				n = max(0, (stop - start + step - 1) // step)
				offset += start * stride
				shape.append(n)
				strides.append(stride * step)

			else:
				raise IndexError("unsupported gpuarray index %r" % (idx, ))

		shape.extend(self.shape[len(item):])
		strides.extend(self.strides[len(item):])

		# the view's buffer spans exactly the bytes it can touch (lazy.py orders foreign streams per byte range)
		extent = self.dtype.itemsize + sum((d - 1) * st for d, st in zip(shape, strides)) if all(d > 0 for d in shape) else 0
		return GPUArray(tuple(shape), self.dtype, gpudata=self.gpudata[offset:offset + extent], strides=tuple(strides))


	def __setitem__(self, key, value):
		self[key].set(value)


	# ------------------------------------------------------------------ arithmetic (HIP kernels)
	def fill(self, value):
		self.enforceContiguous()
		if self.size == 0:
			return self

		item = self.dtype.type(value)
		if self.dtype.itemsize == 4:
			word = int(np.array(item).view(np.uint32))
			if word == 0 and lazy.enabled and self.dtype == np.float32 and self.nbytes >= lazy.Zero.threshold and \
					self.gpudata.root is self.gpudata and self.nbytes == self.gpudata.size:
				# a big tensor zeroed right after its allocation is the accumulator of a residual Add / gradient fan-in
				# (Modules/Add.py:15-22, Replicate.py:22-29): the zeros are written only if somebody asks for them
				self.optr
				lazy.attach(self, lazy.Zero())
				return self
			lib.pz_memset_d32(self.optr, word, self.size, None)

		elif self.dtype.itemsize == 1 and self.nbytes % 4 == 0:
			byte = int(np.array(item).view(np.uint8))
			lib.pz_memset_d32(self.optr, byte * 0x01010101, self.nbytes // 4, None)

		else:
			self.set(np.full(self.shape, item, dtype=self.dtype))

		return self


	def astype(self, dtype):
		self.enforceContiguous()
		dtype = np.dtype(dtype)
		out = GPUArray(self.shape, dtype, allocator=findParentAllocator(self.gpudata))

		if dtype == self.dtype:
			out.set(self)
		elif self.dtype == np.int32 and dtype == np.float32:
			lib.pz_cast_i32_f32(out.optr, self.rptr, self.size, None)
		elif self.dtype == np.float32 and dtype == np.int32:
			lib.pz_cast_f32_i32(out.optr, self.rptr, self.size, None)
		elif self.dtype == np.float32 and dtype == np.float16:           # (storage only: no operator computes in fp16)
			lib.pz_cast_f32_f16(out.optr, self.rptr, self.size, None)
		elif self.dtype == np.float16 and dtype == np.float32:
			lib.pz_cast_f16_f32(out.optr, self.rptr, self.size, None)
		else:
			raise NotImplementedError("astype %s -> %s" % (self.dtype, dtype))

		return out


	def minmax(self, isMax):
		self.enforceContiguous()
		out = GPUArray((), self.dtype, allocator=findParentAllocator(self.gpudata))

		if self.dtype == np.float32:
			lib.pz_reduce_minmax_f32(self.rptr, self.size, int(isMax), out.optr, None)
		elif self.dtype == np.int32:
			lib.pz_reduce_minmax_i32(self.rptr, self.size, int(isMax), out.optr, None)
		else:
			raise NotImplementedError(self.dtype)

		return out


	def min(self):
		return self.minmax(False)


	def max(self):
		return self.minmax(True)


	def enforceSame(self, other):
		self.enforceContiguous()
		other.enforceContiguous()

		if self.shape != other.shape:
			raise ValueError("gpuarray shapes are not equal")
		if self.dtype != other.dtype:
			raise ValueError("gpuarray datatypes are not equal")
		if self.dtype != np.float32:
			raise NotImplementedError("gpuarray arithmetic is implemented for float32")


	def binary(self, other, op):
		self.enforceSame(other)
		out = GPUArray(self.shape, self.dtype, allocator=findParentAllocator(self.gpudata, other.gpudata))
		eltwise(op, out.size, (out, self, other))
		return out


	def __add__(self, other):
		return self.binary(other, lib.OP_ADD3)


	def __mul__(self, other):
		return self.binary(other, lib.OP_MUL)


	def __iadd__(self, other):
		self.enforceSame(other)
		eltwise(lib.OP_IADD, self.size, (self, other))
		return self


	def __imul__(self, other):
		self.enforceSame(other)
		eltwise(lib.OP_IMUL, self.size, (self, other))
		return self


	def __repr__(self):
		return "GPUArray(shape=%s, dtype=%s, ptr=0x%x)" % (self.shape, self.dtype, self.gpudata.ptr or 0)


def findParentAllocator(*buffers):
	from puzzlelib_amd.driver import MemoryPool

	for buf in buffers:
		parent = buf
		while parent is not None and not isinstance(parent, MemoryPool):
			parent = getattr(parent, "parent", None)

		if parent is not None:
			return parent

	return None


def eltwise(op, count, arrays, scalars=(), slc=None, stream=None, readonly=None, rawIdx=-1):
	"""One launch of the element-wise family: pz_eltwise(op, count, ptrs, scalars, start, stop, step).
	`readonly`: indices of arrays the op only reads (default: every array but the first). `rawIdx`: an operand whose pending
	description the caller evaluates itself (through a scalar of the op): its address is taken without the read barrier."""
	nptrs = len(arrays)
	if readonly is None:
		readonly = range(1, nptrs)

	lazy.writeOp = op
	try:
		if stream is None:
			ptrs = (PTRS.get(nptrs) or ctypes.c_void_p * nptrs)(*[
				a.gpudata.ptr if i == rawIdx else a.rptr if i in readonly else a.wptr for i, a in enumerate(arrays)
			])
		else:
			# a borrowed stream (Optimizer.update(useStreams=True)): it follows the main stream up to here, and what it writes
			# carries its completion event, so the main stream waits exactly when it touches those buffers again
			ptrs = (ctypes.c_void_p * nptrs)(*[a.ptrOn(stream, i not in readonly) for i, a in enumerate(arrays)])
			ready = lazy.foreignBegin(stream)
	finally:
		lazy.writeOp = None

	# scalars travel as raw float32 words (bit patterns such as the dropout threshold must survive untouched)
	if isinstance(scalars, np.ndarray):
		sc = scalars
		nsc = sc.size
		scptr = sc.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) if nsc > 0 else None
	else:                                      # plain numbers: rounded to float32 by ctypes exactly as numpy would
		nsc = len(scalars)
		scptr = (FLOATS.get(nsc) or ctypes.c_float * nsc)(*scalars) if nsc > 0 else None

	if slc is None:
		start, stop, step = 0, count, 1
	else:
		start = 0 if slc.start is None else slc.start
		stop = count if slc.stop is None else slc.stop
		step = 1 if slc.step is None else slc.step

	lib.pz_eltwise(op, count, ptrs, nptrs, scptr, nsc, start, stop, step, streamHandle(stream))

	if stream is not None:
		lazy.foreignEnd(
			stream, ready, reads=[a for i, a in enumerate(arrays) if i in readonly],
			writes=[a for i, a in enumerate(arrays) if i not in readonly]
		)
