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
from enum import Enum
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np

from puzzlelib_amd import lib, driver, lazy, fusion
from puzzlelib_amd.lib import HipError, ConvDesc, PoolDesc
from puzzlelib_amd.driver import streamHandle
from puzzlelib_amd.gpuarray import GPUArray, prod, eltwise, contiguousStrides


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


def requireF32(*arrays):
	for ary in arrays:
		if ary is None:
			continue
		if ary.dtype != np.float32:
			raise ValueError("float32 gpuarray expected, got %s" % ary.dtype)
		if not ary.contiguous:
			raise ValueError("gpuarray is not contiguous")


def rptrOf(ary):
	return None if ary is None else ary.rptr


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

		size = c_size_t(0)
		lib.pz_gemm_workspace_bytes(m, n, k, byref(size))
		ws = GPUArray.empty((size.value, ), dtype=np.uint8, allocator=allocator) if size.value > 0 else None
		lib.pz_gemm_ws(
			int(transpA), int(transpB), m, n, k, alpha, A.rptr, A.shape[1], B.rptr, B.shape[1], beta,
			out.optr if beta == 0.0 else out.wptr, n, None if ws is None else ws.optr, size.value, None
		)
		return out


	def gemmBatched(self, A, B, formatA=GroupFormat.gbp.value, formatB=GroupFormat.gbp.value, formatOut=GroupFormat.gbp.value,
					transpA=False, transpB=False, alpha=1.0, beta=0.0, out=None, allocator=None):
		"""One GEMM per group (Cuda/Source/Libs/CuBlas.c:308-312, used by GroupLinear): a tensor is "gbp" =
		(groups, rows, cols) or "bgp" = (rows, groups, cols); in the second layout a group's matrix is the same memory read
		with a row pitch of groups * cols — which pz_gemm's lda / ldb / ldc express directly."""
		requireF32(A, B, out)
		if A.ndim != 3 or B.ndim != 3:
			raise ValueError("gemmBatched operands must be 3-d tensors")
		if transpA and transpB:
			raise ValueError("gemm with both operands transposed is not supported")
		gbp = GroupFormat.gbp.value

		def view(t, fmt):        # (groups, rows, cols, pitch in elements, element offset of group i as a function)
			if fmt == gbp:
				g, r, c = t.shape
				return g, r, c, c, lambda i: i * r * c
			r, g, c = t.shape
			return g, r, c, g * c, lambda i: i * c

		ga, ra, ca, lda, offA = view(A, formatA)
		gb, rb, cb, ldb, offB = view(B, formatB)
		if ga != gb:
			raise ValueError("gemmBatched: %d groups in A, %d in B" % (ga, gb))
		m, k = (ca, ra) if transpA else (ra, ca)
		kb, n = (cb, rb) if transpB else (rb, cb)
		if k != kb:
			raise ValueError("gemm inner dimensions do not match (%d vs %d)" % (k, kb))

		oshape = (ga, m, n) if formatOut == gbp else (m, ga, n)
		if out is None:
			out = GPUArray.empty(oshape, dtype=A.dtype, allocator=allocator)
		elif out.shape != oshape:
			raise ValueError("gemmBatched output has shape %s, expected %s" % (out.shape, oshape))
		_, _, _, ldc, offC = view(out, formatOut)

		pa, pb, pc = A.rptr, B.rptr, (out.optr if beta == 0.0 else out.wptr)
		for i in range(ga):
			lib.pz_gemm(int(transpA), int(transpB), m, n, k, alpha, pa + 4 * offA(i), lda, pb + 4 * offB(i), ldb, beta,
						pc + 4 * offC(i), ldc, None)
		return out


	def scalarOut(self):
		return GPUArray.empty((), dtype=np.float32, allocator=self.backend.memoryPool)


	def dot(self, x, y):
		requireF32(x, y)
		out = self.scalarOut()
		lib.pz_dot(x.rptr, y.rptr, x.size, out.optr, None)
		return float(out.get())


	def l1norm(self, x):
		requireF32(x)
		out = self.scalarOut()
		lib.pz_asum(x.rptr, x.size, out.optr, None)
		return float(out.get())


	def l2norm(self, x):
		return float(np.sqrt(self.dot(x, x)))


# ---------------------------------------------------------------------------------------------- DNN
class DnnContext:
	"""conv / pool / softmax / batch-norm / LRN entry points with the signatures of Hip/Wrappers/MIOpen.py:333-751 —
	nothing more: every fusion this backend does is decided here from what the tensors carry (lazy.py, fusion.py)."""

	# Conv2D -> BatchNorm2D: the convolution's epilogue can leave per-strip channel sums so that the BatchNorm skips its
	# statistics pass. "adaptive": a convolution starts doing so once a BatchNorm has been seen reading its output
	# (keyed by the filter's address); "always" / "never" pin it (tests).
	convStatsPolicy = os.environ.get("PUZZLE_MI355_CONV_STATS", "adaptive")

	# How the MFMA kernels multiply fp32 operands (include/puzzle_mi355.h, pz_conv_math_set): "f32" = the fp32 MFMA;
	# "split6" / "split9" = exact 3-way bf16 split of every operand, 6 / 9 bf16 partial products, fp32 accumulation
	MATH = {"f32": 0, "split6": 6, "split9": 9}
	convMathDefault = os.environ.get("PUZZLE_MI355_MATH", "f32")
	# Output tile of the Winograd 3x3 kernels (pz_conv_winograd_tile_set): 0 = per layer by multiplication count, 2 / 4 pinned
	winogradTileDefault = int(os.environ.get("PUZZLE_MI355_WINO_TILE", "0"))
	sideStreamMaxGflop = float(os.environ.get("PUZZLE_MI355_SIDE_MAX_GFLOP", "15"))      # mean GFLOP per filter-gradient launch
	sideWorkMean = 0.0

	def __init__(self, backend):
		self.backend = backend
		self.statsWanted = weakref.WeakKeyDictionary()      # allocation of a filter -> byte offsets of filters a BatchNorm follows
		self.geometry = {}
		self.sideStream = None
		self.sideLaunches = 0
		self.poolBnCache = {}
		self.packCache = weakref.WeakKeyDictionary()        # allocation of a filter -> {(offset, pass, algo, geometry): PackEntry}
		self.convMath = None
		self.setConvMath(self.convMathDefault)
		self.setWinogradTile(self.winogradTileDefault)


	def setConvMath(self, name):
		"""process-wide; workspace sizes depend on it, so the geometry cache starts over"""
		if name not in self.MATH:
			raise ValueError("PUZZLE_MI355_MATH / setConvMath: %r is not one of %s" % (name, sorted(self.MATH)))
		lib.pz_conv_math_set(self.MATH[name])
		self.geometry.clear()
		DnnContext.descCache.clear()
		self.packCache.clear()
		self.convMath = name
		return self


	def setWinogradTile(self, tile):
		"""process-wide like the math mode: workspace sizes and prepared filter operands depend on it"""
		lib.pz_conv_winograd_tile_set(int(tile))
		self.geometry.clear()
		DnnContext.descCache.clear()
		self.packCache.clear()
		self.winogradTile = int(tile)
		return self


	def enableTensorOps(self, _):
		return self


	@staticmethod
	def getVersion():
		return "puzzle-mi355 implicit-gemm conv %d" % lib.pz_version()


	@staticmethod
	def to4d(shape):
		"""1-D and 3-D convolutions run on the 2-D core: (n, c, w) is (n, c, 1, w); 3-D is handled by the caller."""
		return tuple(shape[:2]) + (1, ) * (4 - len(shape)) + tuple(shape[2:])


	descCache = {}       # call-site arguments -> descriptor: a network asks for the same few dozen every step

	@staticmethod
	def convDesc(dataShape, Wshape, stride, pad, dilation, groups):
		"""The library's descriptor of a 2-D convolution; `.key` = its fields as a tuple, `.geo` = what the library
		answered about it per (pass, algo) (convGeometry). One object per distinct argument list: small networks are bound
		by the host's call rate, and building / hashing descriptors was a tenth of a convolution call."""
		try:
			args = (dataShape, Wshape, stride, pad, dilation, groups)
			return DnnContext.descCache[args]
		except KeyError:
			pass
		except TypeError:                        # (lists as stride / pad: not hashable — no caching)
			args = None
		if len(dataShape) != 4 or len(Wshape) != 4:
			raise NotImplementedError("convolution descriptors are 2-D (1-D tensors are lifted by the callers)")

		(sh, sw), (ph, pw), (dh, dw) = pair(stride), pair(pad), pair(dilation)
		n, c, h, w = dataShape
		k, _, r, s = Wshape
		desc = ConvDesc(n, c, h, w, k, r, s, sh, sw, ph, pw, dh, dw, groups)
		desc.key, desc.geo = (n, c, h, w, k, r, s, sh, sw, ph, pw, dh, dw, groups), {}
		if args is not None:
			if len(DnnContext.descCache) > 4096:
				DnnContext.descCache.clear()
			DnnContext.descCache[args] = desc
		return desc


	def workspace(self, nbytes, allocator):
		if nbytes == 0:
			return None
		return GPUArray.empty((nbytes, ), dtype=np.uint8, allocator=allocator)


	# ---- filter operands prepared once per parameter version -------------------------------------------------------------
	# What a pass derives from the filter tensor alone (the implicit GEMM's packed forward operand and gather table, the
	# Winograd kernels' transformed filters) used to be one ~5 us launch per layer and pass, every step, each on the
	# critical path: 69 of ResNet-50's launches. They are kept per (filter, pass, geometry) instead and are current while
	# the write-version of the filter's allocation stands (lazy.State.version: every write barrier bumps it — the
	# optimizer's update, .set(), a foreign stream's write). The first convolution that finds its operand stale prepares
	# ALL operands of that allocation that were used since the last time, in one batched launch per kernel family
	# (pz_conv2d_prepack): with the parameters in one flat arena that is once per training step, behind the optimizer.
	class PackEntry:
		__slots__ = ("offset", "desc", "which", "algo", "packed", "version", "used")

	def prepared(self, W, desc, which, algo):
		"""address of the prepared filter operand of this pass, or None when the pass reads the filter tensor itself"""
		if not lazy.on("prepack"):
			return None
		root = W.gpudata.root
		entries = self.packCache.get(root)
		if entries is None:
			entries = self.packCache[root] = {}
		offset = W.gpudata.ptr - root.ptr
		key = (offset, which, algo, desc.key)
		entry = entries.get(key)
		if entry is None:
			if len(entries) >= 1024:                    # (a process that keeps changing batch sizes: start over rather than grow)
				entries.clear()
			nbytes = c_size_t(0)
			lib.pz_conv2d_prepack_bytes(byref(desc), which, algo, byref(nbytes))
			entry = entries[key] = self.PackEntry()
			entry.offset, entry.which, entry.algo, entry.version, entry.used = offset, which, algo, -1, False
			entry.desc = ConvDesc.from_buffer_copy(desc)
			entry.packed = GPUArray.empty((nbytes.value, ), dtype=np.uint8) if nbytes.value > 0 else None
		if entry.packed is None:
			return None
		entry.used = True
		lz = lazy.stateOf(root)
		if entry.version != lz.version:
			lazy.readBarrier(root)                      # pending contents written, foreign writers waited for (whole allocation)
			stale = [e for e in entries.values() if e.packed is not None and e.used and e.version != lz.version]
			jobs = (lib.PrepackJob * len(stale))()
			for job, e in zip(jobs, stale):
				job.desc, job.which, job.algo, job.w, job.packed = e.desc, e.which, e.algo, root.ptr + e.offset, e.packed.gpudata.ptr
				e.version, e.used = lz.version, False
			entry.used = True
			lib.pz_conv2d_prepack(jobs, len(stale), None)
			lazy.count("prepack_launch")
		return entry.packed.gpudata.ptr


	def convGeometry(self, desc, which, algo):
		"""(P, Q, workspace bytes, statistics strips) of a convolution pass — host-side queries of the library, asked once
		per (geometry, pass, algo): small networks are bound by the host's call rate (NiN: ~120 launches in 3 ms)."""
		hit = desc.geo.get((which, algo))
		if hit is None:
			p, q, size, strips = c_int(0), c_int(0), c_size_t(0), c_int(0)
			lib.pz_conv2d_out_shape(byref(desc), byref(p), byref(q))
			lib.pz_conv2d_workspace_bytes(byref(desc), which, algo, byref(size))
			if which == lib.CONV_FWD:
				lib.pz_conv2d_fwd_stats_strips(byref(desc), algo, byref(strips))
			fold = c_int(0)
			if which != lib.CONV_FWD:
				lib.pz_conv2d_bn_fold_supported(byref(desc), algo, byref(fold))
			hit = desc.geo[(which, algo)] = (p.value, q.value, size.value, strips.value, bool(fold.value))
		return hit


	# ---- 1-D / 3-D convolutions on the 2-D core (Modules/ConvND.py:14-95 passes nd-tuples straight through)
	@staticmethod
	def lift(ary, nd):
		"""(n, c, w) -> (n, c, 1, w)"""
		return ary if ary is None or nd == 2 else ary.reshape(ary.shape[:2] + (1, ) + ary.shape[2:])

	@staticmethod
	def lift1(v, fill):
		v = (v, ) if isinstance(v, (int, np.integer)) else tuple(v)
		return (fill, int(v[0]))

	@staticmethod
	def unlift(ary, nd):
		return ary if nd == 2 else ary.reshape(ary.shape[:2] + ary.shape[3:])


	def convNd(self, data, W, bias=None, stride=1, pad=0, dilation=1, groups=1, algo=ConvFwdAlgo.auto.value,
			   out=None, allocator=None):
		assert data.ndim == W.ndim and data.shape[1] == W.shape[1] * groups
		nd = data.ndim - 2
		if nd == 1:
			res = self.convNd(
				self.lift(data, 1), self.lift(W, 1), bias, self.lift1(stride, 1), self.lift1(pad, 0), self.lift1(dilation, 1),
				groups, algo, self.lift(out, 1), allocator
			)
			return out if out is not None else self.unlift(res, 1)
		if nd == 3:
			return conv3d.forward(self, data, W, bias, stride, pad, dilation, groups, algo, out, allocator)
		requireF32(data, W, bias, out)
		if lazy.held:
			lazy.prune()             # tensors the filter-gradient stream has finished with go back to the pool

		desc = self.convDesc(data.shape, W.shape, stride, pad, dilation, groups)
		algo = toAlgoId(algo)
		p, q, wsbytes, nstrips, _ = self.convGeometry(desc, lib.CONV_FWD, algo)
		outshape = (data.shape[0], W.shape[0], p, q)

		given = out is not None
		out = GPUArray.empty(outshape, dtype=data.dtype, allocator=allocator) if out is None else out
		if out.shape != outshape:
			raise ValueError("conv output has shape %s, expected %s" % (out.shape, outshape))

		ws = self.workspace(wsbytes, allocator)

		# (the filter is identified by its allocation OBJECT and offset, not by its address: a network built later in a
		# recycled address range starts with no history, so a network's n-th step takes the same path in every process)
		wroot = W.gpudata.root
		key = (wroot, W.gpudata.ptr - wroot.ptr)
		policy = DnnContext.convStatsPolicy
		want = lazy.on("convstats") and not given and (
			policy == "always" or (policy == "adaptive" and key[1] in self.statsWanted.get(wroot, ()))
		)

		stats = None
		if want and nstrips > 0:
			stats = GPUArray.empty((W.shape[0], nstrips, 4), dtype=np.float32, allocator=allocator)
		packed = self.prepared(W, desc, lib.CONV_FWD, algo)
		if packed is not None:
			lib.pz_conv2d_fwd_pre(
				byref(desc), data.rptr, packed, rptrOf(bias), out.optr, None if stats is None else stats.optr, algo, rptrOf(ws),
				wsbytes, None
			)
		elif stats is None:
			lib.pz_conv2d_fwd(byref(desc), data.rptr, W.rptr, rptrOf(bias), out.optr, algo, rptrOf(ws), wsbytes, None)
		else:
			lib.pz_conv2d_fwd_stats(
				byref(desc), data.rptr, W.rptr, rptrOf(bias), out.optr, stats.optr, algo, rptrOf(ws), wsbytes, None
			)
		if stats is not None:
			lazy.setFact(out, "convstats", (stats, outshape))
			lazy.count("conv_stats")

		if lazy.enabled and not given:
			lazy.setFact(out, "fromconv", key)
		return out


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
						   algo=ConvBwdDataAlgo.auto.value, out=None, allocator=None):
		assert grad.ndim == W.ndim and grad.shape[1] == W.shape[0]
		nd = grad.ndim - 2
		if nd == 1:
			res = self.convNdBackwardData(
				self.lift(grad, 1), self.lift(W, 1), bias, self.lift(data, 1), self.lift1(stride, 1), self.lift1(pad, 0),
				self.lift1(dilation, 1), self.lift1(postpad if postpad is not None else 0, 0), groups, algo, self.lift(out, 1),
				allocator
			)
			return out if out is not None else self.unlift(res, 1)
		if nd == 3:
			return conv3d.backwardData(self, grad, W, bias, data, stride, pad, dilation, postpad, groups, algo, out, allocator)

		if data is not None and bias is None and out is None and lazy.on("up2") and groups == 1 and \
				self.compactGradSupported(W, stride, pad, dilation) and data.shape[2] > 1 and data.shape[3] > 1:
			# dx[.., 2i, 2j] = W^T dy[.., i, j] and zero elsewhere: computed on the compact grid (a quarter of the tensor, no
			# memset, dense stores); whoever reads dx either knows where the zeros are (the gradient fan-in,
			# pz_bn_gate_stats_up2) or has it expanded first
			small = self.convNdBackwardData(grad, W, None, None, 1, 0, 1, 0, groups, algo, None, allocator)
			assert small.shape[2:] == tuple((d + 1) // 2 for d in data.shape[2:])
			out = GPUArray.empty(data.shape, dtype=grad.dtype, allocator=allocator)
			lazy.attach(out, fusion.Up2(small))
			lazy.count("compact_dgrad")
			return out

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
		algo = toAlgoId(algo)
		p, q, wsbytes, _, foldable = self.convGeometry(desc, lib.CONV_BWD_DATA, algo)
		if (p, q) != grad.shape[2:]:
			raise ValueError("gradient maps %s do not match the convolution geometry %s" % (grad.shape[2:], (p, q)))

		out = GPUArray.empty(inshape, dtype=grad.dtype, allocator=allocator) if out is None else out
		ws = self.workspace(wsbytes, allocator)

		# the gradient is the un-written input gradient of a BatchNorm (fusion.BnBwdApply): evaluate it while gathering
		bn = lazy.pending(grad, fusion.BnBwdApply) if lazy.on("bnbwdfold") else None
		if bn is not None and foldable:
			lib.pz_conv2d_bwd_data_bn(
				byref(desc), bn.dy.rptr, bn.x.rptr, fusion.raw(bn.coef), W.rptr, out.optr, algo, rptrOf(ws), wsbytes, None
			)
			lazy.count("dgrad_bn_fold")
		else:
			packed = self.prepared(W, desc, lib.CONV_BWD_DATA, algo)
			if packed is not None:
				lib.pz_conv2d_bwd_data_pre(byref(desc), grad.rptr, packed, out.optr, algo, rptrOf(ws), wsbytes, None)
			else:
				lib.pz_conv2d_bwd_data(byref(desc), grad.rptr, W.rptr, out.optr, algo, rptrOf(ws), wsbytes, None)

		if bias is not None:           # deconvolution forward: bias over the produced maps, rows of the (n*maps, pixels) view
			assert bias.size == out.shape[1]
			lib.pz_bias_add(
				out.wptr, out.rptr, bias.rptr, 1, out.shape[0] * out.shape[1], prod(out.shape[2:]), out.shape[1], 0, None
			)

		return out


	# ---- filter gradients on a side stream. Backward-data and backward-filter of a layer read the same incoming gradient
	# and nothing of each other, so every filter-gradient call (pack / main kernel / slab reduce) goes to a second HIP
	# stream behind an event of the main stream; the two chains fill each other's tails and tiny launches. Nobody has to
	# join the streams explicitly: the launch leaves its completion event on the buffers it touched (lazy.foreignEnd) —
	# the optimizer, `.get()`, the all-reduce or the next step's zero fill wait for it when they touch the gradient arena,
	# and the tensors the side stream reads stay referenced (and guarded against overwrites) until the event has passed.
	def filterGradStream(self, gflop=0.0):
		# The split modes run everything on ONE stream: on gfx950 a packed-fp32 instruction whose low lane reads the high
		# half of a source (v_pk_mul_f32 ... op_sel:[0,1] — hipcc's SLP pass emits them all over the BatchNorm / element-wise
		# kernels) returns a wrong low lane while another wave of the SIMD executes a bf16 MFMA
		# (tools/probes/pk_forms_probe.hip): no kernel of this library may overlap a split kernel.
		if not lazy.on("sidestream") or self.convMath != "f32":
			return None
		# A second queue pays while the launches are short (it hides launch latency and fills partial rounds: NiN at batch
		# 128, 3.28 -> 3.08 ms per step). Long kernels from two queues only share the CUs, and share them badly: a
		# filter-gradient launch next to its layer's backward-data launch or next to a BatchNorm pass takes 20-60 % of the
		# shorter kernel LONGER than the two in sequence (tools/pair_overlap.py, profiles/r02_pair_overlap.txt; ResNet-50 at
		# batch 256: 61.9 -> 61.3 ms on one stream, and the host is not held back by the bound on outstanding side launches).
		# The decision follows the running mean of the filter-gradient work per launch, so a network stays on one side of it.
		self.sideWorkMean += 0.1 * (gflop - self.sideWorkMean)
		if self.sideWorkMean > self.sideStreamMaxGflop:
			return None
		if self.sideStream is None:
			prio = os.environ.get("PUZZLE_MI355_SIDE_PRIORITY", "")
			self.sideStream = driver.Stream(priority={"low": -1, "mid": 0, "high": 1}.get(prio))
		self.sideLaunches += 1
		return self.sideStream


	def sideEvent(self):
		"""An event behind everything queued on the side stream so far (None when it never ran): what a consumer of fresh
		filter gradients on a third stream (the all-reduce) waits for besides the main stream."""
		if self.sideStream is None:
			return None
		event = driver.Event()
		event.record(self.sideStream)
		return event


	def convNdBackwardParams(self, data, grad, W, stride=1, pad=0, dilation=1, groups=1, withbias=False, deconv=False,
							 wgrad=None, bgrad=None, scale=1.0, momentum=0.0, algo=ConvBwdFilterAlgo.auto.value,
							 allocator=None):
		assert data.ndim == grad.ndim and grad.shape[1] == W.shape[0] and data.shape[1] == W.shape[1] * groups
		nd = data.ndim - 2
		if nd == 1:
			res = self.convNdBackwardParams(
				self.lift(data, 1), self.lift(grad, 1), self.lift(W, 1), self.lift1(stride, 1), self.lift1(pad, 0),
				self.lift1(dilation, 1), groups, withbias, deconv, self.lift(wgrad, 1), bgrad, scale, momentum, algo, allocator
			)
			if not withbias:
				return wgrad if wgrad is not None else self.unlift(res, 1)
			return (wgrad if wgrad is not None else self.unlift(res[0], 1)), res[1]
		if nd == 3:
			return conv3d.backwardParams(
				self, data, grad, W, stride, pad, dilation, groups, withbias, deconv, wgrad, bgrad, scale, momentum, algo, allocator
			)
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
		_, _, wsbytes, _, foldable = self.convGeometry(desc, lib.CONV_BWD_FILTER, algo)
		ws = self.workspace(wsbytes, allocator)

		bg = None
		if withbias:
			bg = GPUArray.empty((biasof.shape[1], ), dtype=data.dtype, allocator=allocator) if bgrad is None else bgrad

		fused = withbias and bcoef == wcoef and not deconv    # one library call reduces dw and db with the same (alpha, beta)
		bn = lazy.pending(grad, fusion.BnBwdApply) if lazy.on("bnbwdfold") else None
		folded = bn is not None and not withbias and foldable

		gflop = 2e-9 * prod(grad.shape) * prod(W.shape[1:])
		side = self.filterGradStream(gflop) if (not withbias or fused) else None
		st = side.handle if side is not None else None
		reads = [data, bn.dy, bn.x] if folded else [data, grad]
		writes = [wgrad] + ([bg] if fused else [])

		def rp(ary):
			return ary.rptr if side is None else ary.ptrOn(side, False)

		def wp(ary):
			return ary.wptr if side is None else ary.ptrOn(side, True)

		rptrs = [rp(a) for a in reads]
		wptrs = [wp(a) for a in writes]
		ready = lazy.foreignBegin(side) if side is not None else None

		if folded:
			lib.pz_conv2d_bwd_filter_bn(
				byref(desc), rptrs[0], rptrs[1], rptrs[2], fusion.raw(bn.coef), wptrs[0], wcoef[0], wcoef[1], algo,
				rptrOf(ws), wsbytes, st
			)
			lazy.count("wgrad_bn_fold")
		else:
			lib.pz_conv2d_bwd_filter(
				byref(desc), rptrs[0], rptrs[1], wptrs[0], wptrs[1] if fused else None, wcoef[0], wcoef[1], algo,
				rptrOf(ws), wsbytes, st
			)

		if side is not None:
			lazy.foreignEnd(side, ready, reads=reads, writes=writes, keep=(ws, bn.coef if folded else None))

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
		desc = self.convDesc(self.to4d(datashape), self.to4d(Wshape), self.lift1(stride, 1) if len(datashape) == 3 else stride,
							 self.lift1(pad, 0) if len(datashape) == 3 else pad,
							 self.lift1(dilation, 1) if len(datashape) == 3 else dilation, groups)

		out = self.convNd(data, W, None, stride, pad, dilation, groups, allocator=bnd.memoryPool)
		results = []

		for which, run in (
			(lib.CONV_FWD, lambda a: self.convNd(data, W, None, stride, pad, dilation, groups, a, None, bnd.memoryPool)),
			(lib.CONV_BWD_DATA, lambda a: self.convNdBackwardData(
				out, W, None, data, stride, pad, dilation, 0, groups, a, None, bnd.memoryPool
			).rptr),
			(lib.CONV_BWD_FILTER, lambda a: self.convNdBackwardParams(
				data, out, W, stride, pad, dilation, groups, False, False, None, None, 1.0, 0.0, a, bnd.memoryPool
			).rptr),
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
		if data.ndim == 3:
			res = self.poolNd(
				self.lift(data, 1), self.lift1(size, 1), self.lift1(stride, 1), self.lift1(pad, 0), mode, test, self.lift(out, 1),
				allocator
			)
			return self.unlift(res, 1) if test else (self.unlift(res[0], 1), res[1])
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

		index = workspace.optr if (workspace is not None and mode == PoolMode.max.value) else None

		# max pooling over a BatchNorm(+ReLU) that is still only described (the ResNet stem): the band kernel normalises the
		# rows while it stages them, and the normalised tensor is never written — nobody else reads it (the pooling's
		# backward works from the arg-max bytes, the BatchNorm's from its own input)
		bn = lazy.pending(data, fusion.BnApply) if (lazy.on("bnpool") and mode == PoolMode.max.value) else None
		if bn is not None and bn.x.shape == data.shape and self.poolFusesBn(desc):
			lib.pz_pool2d_fwd_bn(byref(desc), bn.x.rptr, fusion.raw(bn.coef), int(bn.relu), out.optr, index, None)
			lazy.count("bn_pool")
		else:
			lib.pz_pool2d_fwd(byref(desc), data.rptr, out.optr, index, None)

		return out if test else (out, workspace)


	def poolFusesBn(self, desc):
		key = tuple(getattr(desc, f) for f, _ in desc._fields_)
		known = self.poolBnCache.get(key)
		if known is None:
			flag = c_int(0)
			lib.pz_pool2d_fwd_bn_supported(byref(desc), byref(flag))
			known = self.poolBnCache[key] = bool(flag.value)
		return known


	def poolNdBackward(self, grad, indata, outdata, workspace, size=2, stride=2, pad=0, mode=PoolMode.max.value,
					   out=None, allocator=None):
		if grad.ndim == 3:
			res = self.poolNdBackward(
				self.lift(grad, 1), self.lift(indata, 1), self.lift(outdata, 1), workspace, self.lift1(size, 1),
				self.lift1(stride, 1), self.lift1(pad, 0), mode, self.lift(out, 1), allocator
			)
			return self.unlift(res, 1)
		assert grad.ndim == 4
		requireF32(grad, indata, outdata, out)

		desc = self.poolDesc(indata.shape, size, stride, pad, mode)
		out = GPUArray.empty(indata.shape, dtype=grad.dtype, allocator=allocator) if out is None else out

		index = workspace.rptr if (workspace is not None and mode == PoolMode.max.value) else None
		# with the arg-max bytes the kernel reads neither tensor: not asking for their addresses leaves a described input
		# (a BatchNorm the forward pooling normalised on the fly) unwritten
		xptr = indata.rptr if index is None else None
		yptr = outdata.rptr if index is None else None
		lib.pz_pool2d_bwd(byref(desc), grad.rptr, xptr, yptr, index, out.optr, None)
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
		lib.pz_softmax_fwd(data.rptr, out.optr, n, c, spatial, None)
		return out


	def softmaxNdBackward(self, grad, outdata, mode=SoftMaxMode.spatial.value, algo=None, out=None, allocator=None):
		requireF32(grad, outdata, out)
		out = GPUArray.empty(grad.shape, dtype=grad.dtype, allocator=allocator) if out is None else out

		n, c, spatial = self.softmaxGeometry(grad, mode)
		lib.pz_softmax_bwd(grad.rptr, outdata.rptr, out.optr, n, c, spatial, None)
		return out


	def bnWorkspace(self, n, c, hw, allocator):
		nbytes = self.geometry.get(("bn", n, c, hw))
		if nbytes is None:
			size = c_size_t(0)
			lib.pz_bn_workspace_bytes(n, c, hw, byref(size))
			nbytes = self.geometry[("bn", n, c, hw)] = size.value
		return GPUArray.empty((nbytes, ), dtype=np.uint8, allocator=allocator), nbytes


	@staticmethod
	def bnGeometry(data, mode):
		"""(n, channels, pixels) of the statistics: per channel over (n, h, w) for spatial mode; per activation over n only
		— the tensor then is n slabs of c*h*w one-pixel channels (Hip/Wrappers/MIOpen.py:634-664 honours `mode`)."""
		if mode == BatchNormMode.spatial.value:
			return data.shape[0], data.shape[1], prod(data.shape[2:])
		return data.shape[0], prod(data.shape[1:]), 1


	def batchNormNd(self, data, mean, var, scale, bias, epsilon=1e-5, factor=1.0, test=False,
					mode=BatchNormMode.spatial.value, out=None, allocator=None):
		assert mean.ndim == 1 and var.ndim == 1 and scale.ndim == 1 and bias.ndim == 1
		requireF32(data, mean, var, scale, bias, out)
		n, c, hw = self.bnGeometry(data, mode)
		assert c == mean.dimAt(0)

		given = out is not None
		out = GPUArray.empty(data.shape, dtype=data.dtype, allocator=allocator) if out is None else out

		if test:
			lib.pz_bn_fwd_infer(data.rptr, out.optr, n, c, hw, scale.rptr, bias.rptr, mean.rptr, var.rptr, epsilon, None)
			return out

		savemean = GPUArray.empty(mean.shape, dtype=data.dtype, allocator=allocator)
		saveinvvar = GPUArray.empty(var.shape, dtype=data.dtype, allocator=allocator)
		coef = GPUArray.empty((c, 2), dtype=data.dtype, allocator=allocator)
		ws, nbytes = self.bnWorkspace(n, c, hw, allocator)

		# statistics: the producing convolution's strip sums when it left them; otherwise tell that convolution (by its
		# filter's address) that a BatchNorm reads its output, so that it does from the next pass on
		# (per-channel sums of the convolution's (n, k, p, q) output: they are this BatchNorm's statistics only if it
		# normalises that very tensor over the same channel axis — not a reshape, a slice or per-activation mode)
		stats = lazy.fact(data, "convstats") if mode == BatchNormMode.spatial.value and data.ndim == 4 else None
		if stats is not None:
			stats = stats[0] if tuple(stats[1]) == tuple(data.shape) and stats[0].shape[0] == c else None
		if stats is None and DnnContext.convStatsPolicy == "adaptive" and data.ndim == 4:
			key = lazy.fact(data, "fromconv")
			if key is not None:
				self.statsWanted.setdefault(key[0], set()).add(key[1])

		lib.pz_bn_fwd_train_coef(
			data.rptr, n, c, hw, scale.rptr, bias.rptr, mean.wptr, var.wptr, savemean.optr, saveinvvar.optr, epsilon, factor,
			None if stats is None else stats.rptr, 0 if stats is None else stats.shape[1], coef.optr, ws.optr, nbytes, None
		)

		# the normalisation itself is only described: y = a*x + b (fusion.BnApply). An in-place ReLU joins the description,
		# a residual Add / a convolution's gather applies it on the fly, anyone else has it written first.
		thunk = fusion.BnApply(data.reshape(n, c, hw, 1) if data.ndim != 4 or mode != BatchNormMode.spatial.value else data, coef)
		if lazy.on("bnapply") and not given and lazy.whole(out) and not lazy.sameBuffer(out, data):
			lazy.attach(out, thunk)
			if lazy.enabled:
				lazy.setFact(data, "bnsaved", savemean)
		else:
			out.optr
			thunk.run(out)
		return out, savemean, saveinvvar


	def batchNormNdBackward(self, grad, data, scale, savemean=None, saveinvvar=None, epsilon=1e-5,
							mode=BatchNormMode.spatial.value, out=None, allocator=None):
		assert data.ndim == grad.ndim
		requireF32(grad, data, scale, savemean, saveinvvar, out)
		if savemean is None or saveinvvar is None:
			raise ValueError("batchNormNdBackward needs the saved mean / inverse variance of the forward pass")

		given = out is not None
		out = GPUArray.empty(grad.shape, dtype=grad.dtype, allocator=allocator) if out is None else out
		scalegrad = GPUArray.empty(scale.shape, dtype=scale.dtype, allocator=allocator)
		bgrad = GPUArray.empty(scale.shape, dtype=scale.dtype, allocator=allocator)

		n, c, hw = self.bnGeometry(data, mode)
		spatial = mode == BatchNormMode.spatial.value and data.ndim == 4

		# (1) the gradient still carries the derivative of THIS layer's in-place ReLU (reluDerKer(g, g, y) with
		# y = relu(bn(x)), fusion.Gate): gate while loading, y re-created from x with the forward's own {a, b}
		gate = lazy.pending(grad, fusion.Gate) if (spatial and lazy.on("bnrelubwd")) else None
		if gate is not None:
			desc = lazy.fact(gate.y, "bnapply")
			if desc is None:
				waiting = lazy.pending(gate.y, fusion.BnApply)
				desc = None if waiting is None else (waiting.x, waiting.coef, waiting.relu)
			if desc is not None and desc[2] and lazy.sameBuffer(desc[0], data):
				ws, nbytes = self.bnWorkspace(n, c, hw, allocator)
				lib.pz_bn_bwd_gate(
					data.rptr, lazy.rawRead(grad), out.optr, n, c, hw, scale.rptr, savemean.rptr, saveinvvar.rptr,
					scalegrad.optr, bgrad.optr, fusion.raw(desc[1]), ws.optr, nbytes, None
				)
				lazy.count("bn_bwd_gate")
				return out, scalegrad, bgrad

		# (2) the gradient is an un-written gated fan-in (fusion.Sum): write it and sum this layer's backward statistics —
		# and those of the other BatchNorm that fed the same residual Add — in the same pass
		parts = None
		if spatial and lazy.on("gatestats"):
			waiting = lazy.pending(grad, fusion.Sum)
			if waiting is not None and waiting.gate is not None:
				targets = [(data, savemean)]
				waiting.gate.rptr                            # (a gate tensor that is itself still described gets written now)
				for other in (lazy.fact(waiting.gate, "bnterms") or ()):
					saved = lazy.fact(other, "bnsaved")
					if saved is not None and not lazy.sameBuffer(other, data) and other.shape == data.shape and len(targets) < 2:
						targets.append((other, saved))
				fusion.settleWithStats(grad, targets)
			for x, mean_, part in (lazy.fact(grad, "bwdparts") or ()):
				if lazy.sameBuffer(x, data) and lazy.sameBuffer(mean_, savemean):
					parts = part

		if parts is None:
			ws, nbytes = self.bnWorkspace(n, c, hw, allocator)
			lib.pz_bn_bwd_acc(
				data.rptr, grad.rptr, out.optr, n, c, hw, scale.rptr, None, savemean.rptr, saveinvvar.rptr, scalegrad.optr,
				bgrad.optr, lib.BN_ACT_NONE, None, None, 1.0, 0.0, ws.optr, nbytes, None
			)
			return out, scalegrad, bgrad

		# (3) statistics known: the input gradient is dx = A*dy + B*x + C per channel — described, not written; the 1x1
		# convolution in front evaluates it inside its backward gathers (pz_conv2d_bwd_{data,filter}_bn)
		lazy.count("bn_bwd_from_partials")
		if lazy.on("bnbwdfold") and not given and lazy.whole(out):
			coef = GPUArray.empty((c, 4), dtype=np.float32, allocator=allocator)
			lib.pz_bn_bwd_coef(
				n, c, hw, scale.rptr, savemean.rptr, saveinvvar.rptr, scalegrad.optr, bgrad.optr, None, None, 1.0, 0.0,
				fusion.raw(parts), coef.optr, None
			)
			lazy.attach(out, fusion.BnBwdApply(grad, data, coef))
		else:
			lib.pz_bn_bwd_from_partials(
				data.rptr, grad.rptr, out.optr, n, c, hw, scale.rptr, savemean.rptr, saveinvvar.rptr, scalegrad.optr, bgrad.optr,
				None, None, 1.0, 0.0, fusion.raw(parts), None
			)
		return out, scalegrad, bgrad


	def lrn(self, data, N=5, alpha=1e-4, beta=0.75, K=2.0, mode=LRNMode.map.value, test=False, out=None, allocator=None):
		"""Hip/Wrappers/MIOpen.py:708-731. Training mode returns (out, workspace): the workspace holds the normaliser
		s = K + alpha/|window| * sum x^2 per element, which the backward reads."""
		assert data.ndim == 4
		requireF32(data, out)
		out = GPUArray.empty(data.shape, dtype=data.dtype, allocator=allocator) if out is None else out
		workspace = None if test else GPUArray.empty(data.shape, dtype=np.float32, allocator=allocator)
		n, c, h, w = data.shape
		lib.pz_lrn_fwd(
			data.rptr, out.optr, None if workspace is None else workspace.optr, n, c, h, w, N, alpha, beta, K,
			int(mode == LRNMode.cross.value), None
		)
		return out if test else (out, workspace)


	def lrnBackward(self, grad, indata, outdata, workspace, N=5, alpha=1e-4, beta=0.75, K=2.0, mode=LRNMode.map.value,
					out=None, allocator=None):
		"""Hip/Wrappers/MIOpen.py:734-751"""
		requireF32(grad, indata, out)
		mode = mode.value if isinstance(mode, Enum) else mode
		if workspace is None:                   # (a forward pass in inference mode keeps no normaliser: recompute it)
			_, workspace = self.lrn(indata, N, alpha, beta, K, mode, False, None, allocator)
		out = GPUArray.empty(grad.shape, dtype=grad.dtype, allocator=allocator) if out is None else out
		n, c, h, w = indata.shape
		lib.pz_lrn_bwd(
			indata.rptr, grad.rptr, workspace.rptr, out.optr, n, c, h, w, N, alpha, beta, K, int(mode == LRNMode.cross.value), None
		)
		return out


class conv3d:
	"""3-D convolutions (Modules/Conv3D.py through the same Dnn.convNd* entries) on the 2-D MFMA core: the depth taps are
	unfolded into channels — xu[(n, d), (c, t), h, w] = x[n, c, d*sd + t*dd - pd, h, w] (zero outside), T strided copies
	— after which all three passes are the 2-D passes with filters (K, C*T, R, S) = the 5-d filter tensor reshaped:
	forward = conv2d(xu) transposed to (N, K, D', P, Q); backward-filter = the 2-D filter gradient, already in 5-d
	order; backward-data = the 2-D backward-data folded back over the taps."""

	@staticmethod
	def triple(v):
		return (int(v), ) * 3 if isinstance(v, (int, np.integer)) else tuple(int(a) for a in v)


	@staticmethod
	def taps(D, Dout, T, sd, pd, dd):
		"""per depth tap t: (first output depth, count, first input depth) of the in-range part"""
		for t in range(T):
			d0 = max(0, -((t * dd - pd) // sd))                    # smallest d with d*sd + t*dd - pd >= 0
			d1 = min(Dout - 1, (D - 1 + pd - t * dd) // sd)
			if d1 >= d0:
				yield t, d0, d1 - d0 + 1, d0 * sd + t * dd - pd


	@classmethod
	def unfold(cls, data, T, Dout, sd, pd, dd, allocator):
		n, c, D, h, w = data.shape
		xu = GPUArray.zeros((n, Dout, c, T, h, w), dtype=data.dtype, allocator=allocator)
		sN, sD, sC, sT, sH, sW = xu.strides
		for t, d0, count, z0 in cls.taps(D, Dout, T, sd, pd, dd):
			src = data[:, :, z0:z0 + (count - 1) * sd + 1:sd]
			dst = MemModule.viewLike(xu, (n, c, count, h, w), (sN, sC, sD, sH, sW), d0 * sD + t * sT)
			dst.stridedCopyFrom(src)
		return xu.reshape(n * Dout, c * T, h, w)


	@classmethod
	def geometry(cls, dshape, Wshape, stride, pad, dilation):
		(sd, sh, sw), (pd, ph, pw), (dd, dh, dw) = cls.triple(stride), cls.triple(pad), cls.triple(dilation)
		T = Wshape[2]
		Dout = (dshape[2] + 2 * pd - dd * (T - 1) - 1) // sd + 1
		return (sd, pd, dd, T, Dout), dict(stride=(sh, sw), pad=(ph, pw), dilation=(dh, dw))


	@classmethod
	def forward(cls, dnn, data, W, bias, stride, pad, dilation, groups, algo, out, allocator):
		(sd, pd, dd, T, Dout), kw = cls.geometry(data.shape, W.shape, stride, pad, dilation)
		n, k = data.shape[0], W.shape[0]
		xu = cls.unfold(data, T, Dout, sd, pd, dd, allocator)
		W2 = W.reshape(k, W.shape[1] * T, W.shape[3], W.shape[4])
		y2 = dnn.convNd(xu, W2, bias, groups=groups, algo=algo, allocator=allocator, **kw)
		y5 = y2.reshape(n, Dout, k, y2.shape[2], y2.shape[3])
		return dnn.backend.memmod.transpose(y5, (0, 2, 1, 3, 4), out=out, allocator=allocator)


	@classmethod
	def backwardData(cls, dnn, grad, W, bias, data, stride, pad, dilation, postpad, groups, algo, out, allocator):
		if data is None:
			# deconvolution forward (Modules/Deconv3D.py through Dnn.deconvNd): the produced shape follows from the geometry
			(sd_, sh_, sw_), (pd_, ph_, pw_), (dd_, dh_, dw_) = cls.triple(stride), cls.triple(pad), cls.triple(dilation)
			qd, qh, qw = cls.triple(postpad if postpad is not None else 0)
			n_, _, od, oh, ow = grad.shape
			_, cg, T_, R_, S_ = W.shape
			data = cls.Shape((
				n_, cg * groups, (od - 1) * sd_ + dd_ * (T_ - 1) - 2 * pd_ + 1 + qd, (oh - 1) * sh_ + dh_ * (R_ - 1) - 2 * ph_ + 1 + qh,
				(ow - 1) * sw_ + dw_ * (S_ - 1) - 2 * pw_ + 1 + qw
			))
		(sd, pd, dd, T, Dout), kw = cls.geometry(data.shape, W.shape, stride, pad, dilation)
		if Dout != grad.shape[2]:
			raise ValueError("gradient depth %d does not match the convolution geometry (%d)" % (grad.shape[2], Dout))
		n, c, D, h, w = data.shape
		k = W.shape[0]
		memmod = dnn.backend.memmod
		g2 = memmod.transpose(grad, (0, 2, 1, 3, 4), allocator=allocator).reshape(n * Dout, k, grad.shape[3], grad.shape[4])
		W2 = W.reshape(k, W.shape[1] * T, W.shape[3], W.shape[4])
		dxu = dnn.convNdBackwardData(
			g2, W2, None, cls.Shape((n * Dout, c * T, h, w)), groups=groups, algo=algo, allocator=allocator, **kw
		)
		dx = GPUArray.zeros(data.shape, dtype=grad.dtype, allocator=allocator) if out is None else out
		if out is not None:
			out.fill(0)
		sN, sD, sC, sT, sH, sW = contiguousStrides((n, Dout, c, T, h, w), 4)
		for t, d0, count, z0 in cls.taps(D, Dout, T, sd, pd, dd):
			part = GPUArray.zeros(data.shape, dtype=grad.dtype, allocator=allocator)
			src = MemModule.viewLike(dxu, (n, c, count, h, w), (sN, sC, sD, sH, sW), d0 * sD + t * sT)
			part[:, :, z0:z0 + (count - 1) * sd + 1:sd].stridedCopyFrom(src)
			dnn.backend.toVectorAddVectorKer(np.float32)(dx.ravel(), part.ravel(), 1.0)
		if bias is not None:           # deconvolution forward: bias over the produced maps (rows of the (n*maps, voxels) view)
			assert bias.size == dx.shape[1]
			lib.pz_bias_add(dx.wptr, dx.rptr, bias.rptr, 1, dx.shape[0] * dx.shape[1], prod(dx.shape[2:]), dx.shape[1], 0, None)
		return dx


	class Shape:
		"""stands in for the `data` argument of convNdBackwardData where only its shape is read"""
		def __init__(self, shape):
			self.shape, self.ndim = tuple(shape), len(shape)


	@classmethod
	def backwardParams(cls, dnn, data, grad, W, stride, pad, dilation, groups, withbias, deconv, wgrad, bgrad, scale, momentum,
					   algo, allocator):
		if deconv and withbias:
			# deconvolution (`data` = the gradient of what the deconvolution produced, `grad` = its input): the filter gradient is
			# the same contraction; the bias gradient sums over `data`'s maps (Hip/Wrappers/MIOpen.py:435-436)
			wg = cls.backwardParams(dnn, data, grad, W, stride, pad, dilation, groups, False, False, wgrad, None, scale, momentum,
									algo, allocator)
			accumulate = bgrad is not None and (scale != 1.0 or momentum != 0.0)
			nn_, maps = data.shape[:2]
			bg = GPUArray.empty((maps, ), dtype=data.dtype, allocator=allocator) if bgrad is None else bgrad
			matmod = dnn.backend.matmod
			persample = matmod.matsum(data.reshape(nn_ * maps, prod(data.shape[2:])), axis=1, allocator=allocator)
			matmod.matsum(persample.reshape(nn_, maps), axis=0, out=bg, alpha=scale if accumulate else 1.0,
						  beta=momentum if accumulate else 0.0)
			return wg, bg
		(sd, pd, dd, T, Dout), kw = cls.geometry(data.shape, W.shape, stride, pad, dilation)
		n, k = data.shape[0], W.shape[0]
		xu = cls.unfold(data, T, Dout, sd, pd, dd, allocator)
		g2 = dnn.backend.memmod.transpose(grad, (0, 2, 1, 3, 4), allocator=allocator).reshape(n * Dout, k, grad.shape[3], grad.shape[4])
		shape2 = (k, W.shape[1] * T, W.shape[3], W.shape[4])
		res = dnn.convNdBackwardParams(
			xu, g2, W.reshape(shape2), groups=groups, withbias=withbias, deconv=False,
			wgrad=None if wgrad is None else wgrad.reshape(shape2), bgrad=bgrad, scale=scale, momentum=momentum, algo=algo,
			allocator=allocator, **kw
		)
		if withbias:
			return (wgrad if wgrad is not None else res[0].reshape(W.shape)), res[1]
		return wgrad if wgrad is not None else res.reshape(W.shape)


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
			if name != "calcAccuracy":
				raise NotImplementedError(name)

			def calcAccuracy(x, y, allocator=None):
				assert x.dtype == np.int32 and y.dtype == np.int32 and x.size == y.size
				out = GPUArray.empty((), dtype=np.float32, allocator=allocator)
				lib.pz_count_neq_i32(x.rptr, y.rptr, x.size, out.optr, None)
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


class PoolModule:
	"""maxpool2d / maxpool2dBackward / maxunpool2d / maxunpool2dBackward with index masks — Cuda/Kernels/Pool.py:117-213
	(MaxPool2D(useMask=True), MaxUnpool2D)."""

	def __init__(self, backend):
		self.backend = backend


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


# ---------------------------------------------------------------------------------------------- element-wise kernel objects
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
		return False

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
	if not lazy.sameBuffer(ingrad, outgrad) or not lazy.whole(ingrad) or ingrad.shape != outdata.shape or \
			lazy.sameBuffer(ingrad, outdata):
		return False

	waiting = lazy.editable(ingrad)
	root = ingrad.gpudata.root
	if isinstance(waiting, fusion.Sum) and not waiting.relu and waiting.gate is None and lazy.on("addgate"):
		waiting.gate = outdata
		lazy.depend(outdata, root)
		return True

	if lazy.on("gate"):
		ingrad.wptr                                          # whatever is pending gets written; dependents are settled
		lazy.attach(ingrad, fusion.Gate(outdata))
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

		words = np.empty(len(scalars), dtype=np.float32)
		for i, value in enumerate(scalars):
			if i in self.rawScalar:
				words.view(np.uint32)[i] = np.uint32(int(value))
			else:
				words[i] = value

		eltwise(self.op, arrays[0].size, arrays, words, slc=slc, stream=stream, readonly=self.readonly)


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

		def unsupported(*args, **kwargs):
			raise NotImplementedError("this kernel is outside the implemented operator path (fp32-only backend)")

		self.castFP16toFP32 = self.castFP32toFP16 = unsupported
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
