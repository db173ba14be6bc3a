"""
Enumerations and small argument helpers shared by the context objects of the backend (puzzlelib_amd/blas.py, dnn.py,
modules.py, kernels.py, backend.py): the ids the reference's wrappers expose (Hip/Wrappers/MIOpen.py:23-88,
Cuda/Wrappers/CuBlas.py) and the checks every entry point makes on its operands.
"""
from enum import Enum

import numpy as np

from puzzlelib_amd import lib


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
