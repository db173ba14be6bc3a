"""
BlasContext — the `blas` attribute of the backend object: Backend/Blas.py:43-102 calls gemm / gemmBatched / dot / l1norm /
l2norm on it (original: Cuda/Source/Libs/CuBlas.c through Cuda/Wrappers/CuBlas.py). Signature glue over pz_gemm* /
pz_dot / pz_*norm; no policy lives here.
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
		# an operand with ONE group serves every group of the other (batch stride 0, Cuda/Source/Libs/CuBlas.c:251-257,303-304:
		# GroupLinear with wmode / inmode "one", Modules/Sum via a single ones-matrix)
		if ga != gb and ga != 1 and gb != 1:
			raise ValueError("gemmBatched: %d groups in A, %d in B" % (ga, gb))
		if ga == 1 and gb > 1:
			offA = lambda i: 0
		if gb == 1 and ga > 1:
			offB = lambda i: 0
		ga = max(ga, gb)
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
