"""
The dispatch surface, bound to the MI355X backend.

In PuzzleLib the modules import function objects from Backend/{gpuarray,Blas,Dnn}.py and Backend/Kernels/*.py, whose
`initHip()`/`initGPU()` wrap a backend object (Backend/gpuarray.py:60-113, Backend/Blas.py:43-102,
Backend/Dnn.py:124-268, Backend/Kernels/{ElementWise,MatVec,Costs}.py). Those files stay as they are when this backend
is plugged into PuzzleLib itself. For the stand-alone harness of this repository (the reference
package does not travel to the GPU box) the same wrappers are restated here with exactly the reference's positional
signatures and nothing else — no extra keyword, no extra entry: whatever the backend fuses, it fuses behind these calls
(puzzlelib_amd/lazy.py). Namespaces carry the reference's names: `bound().gpuarray`, `.Blas`, `.Dnn`, `.ElementWise`,
`.MatVec`, `.Costs`. tests/test_host_logic.py replays the reference's own modules against the same backend object in
dry-run mode and checks that this harness issues the identical call sequence.

Binding is lazy (first call to `bound()`), so host-only code can import the module graph without a device; the
device is then required — there is no other backend to fall back to.
"""
from types import SimpleNamespace

from puzzlelib_amd.settings import Config

_surface = None


def bound():
	global _surface
	if _surface is None:
		_surface = bind()
	return _surface


def bind():
	Config.requireHip()
	if not Config.shouldInit():
		raise Config.ConfigError("backend initialisation outside the main process needs Config.allowMultiContext = True")

	from puzzlelib_amd import backend as Backend
	bnd = Backend.getBackend(Config.deviceIdx, initmode=2, logger=Config.getLogger())

	GPUArray, memoryPool, blas, dnn, matmod, costmod = \
		bnd.GPUArray, bnd.memoryPool, bnd.blas, bnd.dnn, bnd.matmod, bnd.costmod

	# ------------------------------------------------------------------ gpuarray (Backend/gpuarray.py:60-113)
	gpuarray = SimpleNamespace(
		backend=bnd, GPUArray=GPUArray, to_gpu=GPUArray.toGpu, empty=GPUArray.empty, zeros=GPUArray.zeros,
		minimum=GPUArray.min, maximum=GPUArray.max, getDeviceName=lambda: bnd.device.name(),
		SharedArray=bnd.SharedArray, memoryPool=memoryPool, streamManager=bnd.streamManager, globalRng=bnd.globalRng,
		copy=lambda dest, source: bnd.copy(dest, source, allocator=memoryPool),
		concatenate=lambda tup, axis, out=None: bnd.concatenate(tup, axis, out, allocator=memoryPool),
		split=lambda ary, sections, axis: bnd.split(ary, sections, axis, allocator=memoryPool),
		tile=lambda ary, times, axis: bnd.tile(ary, times, axis, allocator=memoryPool),
		fillUniform=lambda data, minval, maxval, rng: bnd.fillUniform(data, minval, maxval, rng),
		fillNormal=lambda data, mean, stddev, rng: bnd.fillNormal(data, mean, stddev, rng),
		dtypesSupported=bnd.dtypesSupported, timeKernel=bnd.timeKernel
	)

	# ------------------------------------------------------------------ Blas (Backend/Blas.py:43-75)
	def toVectorAddVector(y, x, alpha=1.0):
		bnd.toVectorAddVectorKer(y.dtype)(y, x, alpha)
		return y

	def addVectorToVector(x, y, out=None, alpha=1.0, beta=1.0):
		if out is None:
			out = GPUArray.empty(x.shape, dtype=x.dtype, allocator=memoryPool)
		else:
			assert out.shape == x.shape
		bnd.addKer(out.dtype)(out, x, alpha, y, beta)
		return out

	def mulMatrixOnMatrix(A, B, out=None, transpA=False, transpB=False, alpha=1.0, beta=0.0):
		return blas.gemm(A, B, out, transpA, transpB, alpha, beta, memoryPool)

	def sumOnMatrix(A, out=None, cols=True, alpha=1.0, beta=0.0):
		assert A.ndim == 2
		return matmod.matsum(A, 0 if cols else 1, out, alpha, beta, memoryPool)

	Blas = SimpleNamespace(
		toVectorAddVector=toVectorAddVector, addVectorToVector=addVectorToVector, dot=blas.dot,
		vectorL1Norm=blas.l1norm, mulMatrixOnMatrix=mulMatrixOnMatrix, sumOnMatrix=sumOnMatrix
	)

	# ------------------------------------------------------------------ Dnn (Backend/Dnn.py:124-268)
	def convNd(data, W, bias, stride, pad, dilation, groups, algo):
		return dnn.convNd(
			data, W, bias.ravel() if bias is not None else None, stride, pad, dilation, groups, algo.value, None, memoryPool
		)

	def convNdBackwardData(grad, W, data, stride, pad, dilation, groups, algo):
		return dnn.convNdBackwardData(grad, W, None, data, stride, pad, dilation, None, groups, algo.value, None, memoryPool)

	def convNdBackwardParams(data, grad, W, bias, stride, pad, dilation, groups, wgrad, bgrad, scale, momentum, algo):
		return dnn.convNdBackwardParams(
			data, grad, W, stride, pad, dilation, groups, bias is not None, False, wgrad,
			bgrad.ravel() if bgrad is not None else None, scale, momentum, algo.value, memoryPool
		)

	# deconvolution = the convolution's passes with their roles swapped (Backend/Dnn.py:211-231)
	def deconvNd(data, W, bias, stride, pad, dilation, postpad, groups, algo):
		return dnn.convNdBackwardData(
			data, W, bias.ravel() if bias is not None else None, None, stride, pad, dilation, postpad, groups, algo.value, None,
			memoryPool
		)

	def deconvNdBackwardData(grad, W, data, stride, pad, dilation, groups, algo):
		assert data is not None
		return dnn.convNd(grad, W, None, stride, pad, dilation, groups, algo.value, None, memoryPool)

	def deconvNdBackwardParams(data, grad, W, bias, stride, pad, dilation, groups, wgrad, bgrad, scale, momentum, algo):
		return dnn.convNdBackwardParams(
			grad, data, W, stride, pad, dilation, groups, bias is not None, True, wgrad,
			bgrad.ravel() if bgrad is not None else None, scale, momentum, algo.value, memoryPool
		)

	def poolNd(data, size, stride, pad, mode, test):
		result = dnn.poolNd(data, size, stride, pad, mode.value, test, None, memoryPool)
		return result if not test else (result, None)

	def poolNdBackward(indata, outdata, grad, workspace, size, stride, pad, mode):
		return dnn.poolNdBackward(grad, indata, outdata, workspace, size, stride, pad, mode.value, None, memoryPool)

	def batchNormNd(data, scale, bias, mean, var, epsilon, factor, test, mode=bnd.BatchNormMode.spatial, out=None):
		shape = scale.shape
		result = dnn.batchNormNd(
			data, mean.ravel(), var.ravel(), scale.ravel(), bias.ravel(), epsilon, factor, test, mode.value, out=out,
			allocator=memoryPool
		)
		if test:
			return result

		outdata, savemean, saveinvvar = result
		return outdata, savemean.reshape(shape), saveinvvar.reshape(shape)

	def batchNormNdBackward(data, grad, scale, savemean, saveinvvar, epsilon, mode=bnd.BatchNormMode.spatial):
		shape = scale.shape
		ingrad, scalegrad, bgrad = dnn.batchNormNdBackward(
			grad, data, scale.ravel(), savemean.ravel(), saveinvvar.ravel(), epsilon, mode.value, allocator=memoryPool
		)
		return ingrad, scalegrad.reshape(shape), bgrad.reshape(shape)

	def softmaxNd(data, mode=bnd.SoftMaxMode.spatial):
		return dnn.softmaxNd(data, mode.value, allocator=memoryPool)

	def softmaxNdBackward(outdata, grad):
		return dnn.softmaxNdBackward(grad, outdata, allocator=memoryPool)

	def convNdbenchmark(datashape, Wshape, stride, pad, dilation, groups, transpose):
		import numpy as np
		fwd, bwdData, bwdParam = bnd.convNdbenchmark(datashape, Wshape, np.float32, stride, pad, dilation, groups)
		return fwd, bwdParam, bwdData

	Dnn = SimpleNamespace(
		ConvFwdAlgo=bnd.ConvFwdAlgo, ConvBwdDataAlgo=bnd.ConvBwdDataAlgo, ConvBwdFilterAlgo=bnd.ConvBwdFilterAlgo,
		PoolMode=bnd.PoolMode, BatchNormMode=bnd.BatchNormMode, SoftMaxMode=bnd.SoftMaxMode,
		RNNMode=bnd.RNNMode, DirectionMode=bnd.DirectionMode,
		convNd=convNd, convNdBackwardData=convNdBackwardData, convNdBackwardParams=convNdBackwardParams,
		deconvNd=deconvNd, deconvNdBackwardData=deconvNdBackwardData, deconvNdBackwardParams=deconvNdBackwardParams,
		convNdbenchmark=convNdbenchmark, poolNd=poolNd, poolNdBackward=poolNdBackward,
		batchNormNd=batchNormNd, batchNormNdBackward=batchNormNdBackward,
		softmaxNd=softmaxNd, softmaxNdBackward=softmaxNdBackward,
		deviceSupportsBatchHint=bnd.deviceSupportsBatchHint
	)

	# ------------------------------------------------------------------ kernels (Backend/Kernels/ElementWise.py:71-121)
	kernelNames = [
		"sigmoidKer", "sigmoidDerKer", "tanhKer", "tanhDerKer", "reluKer", "reluDerKer", "leakyReluKer",
		"leakyReluDerKer", "eluKer", "eluDerKer", "softPlusKer", "softPlusDerKer", "clipKer", "clipDerKer", "geluKer",
		"geluDerKer", "dropoutKer", "dropout2dKer", "toVectorAddVectorKer", "classicMomSGDKer", "nesterovMomSGDKer",
		"rmspropKer", "adamKer", "rmspropGravesKer", "adagradKer", "adadeltaKer", "smorms3Ker", "addKer", "mulKer",
		"linearKer", "rbmKer", "absKer", "weightDecayKer", "l1penaltyKer", "l1gradKer"
	]
	ElementWise = SimpleNamespace(**{name: getattr(bnd, name) for name in kernelNames})

	# ------------------------------------------------------------------ MatVec / Costs (Backend/Kernels/MatVec.py:35-57, Costs.py:47-73)
	MatVec = SimpleNamespace(
		addVecToMat=lambda vec, mat, axis, out: matmod.addVecToMat(vec, mat, axis, out, memoryPool),
		argmax=lambda tensor, axis: matmod.argmax(tensor, axis, memoryPool),
		addVecToMatBatch=lambda vec, mat, axis, out: matmod.addVecToMat(vec, mat, axis, out, memoryPool),
		argmaxBatch=lambda tensor, axis: matmod.argmax(tensor, axis, memoryPool)
	)

	Costs = SimpleNamespace(
		getAccuracyKernel=bnd.getAccuracyKernel,
		crossEntropyKernel=lambda scores, labels, weights=None, error=None: costmod.crossEntropy(
			scores, labels, weights, error, memoryPool
		)
	)

	return SimpleNamespace(
		backend=bnd, gpuarray=gpuarray, Blas=Blas, Dnn=Dnn, ElementWise=ElementWise, MatVec=MatVec, Costs=Costs
	)
