"""
DnnContext — the `dnn` attribute of the backend object (Backend/Dnn.py:124-338 dispatches convNd / poolNd / batchNormNd /
softmaxNd / lrn and their backward passes to it; original: Hip/Wrappers/MIOpen.py:333-700).

This file holds signature glue AND the backend's fusion policy for the DNN operators — what a call defers to a tensor
description instead of launching (puzzlelib_amd/lazy.py, fusion.py): convolution epilogue statistics for a following
BatchNorm (convStatsPolicy / statsWanted), prepared filter operands (prepared), the BatchNorm backward folded into the
1x1 convolutions' backward gathers, compact stride-2 input gradients, the filter-gradient stream policy
(filterGradStream), BatchNorm forward / backward descriptions. The element-wise side of the policy (ReLU / gate / axpy
absorption) is puzzlelib_amd/kernels.py.
"""
import os, weakref, sys, time, ctypes
from ctypes import byref, c_int, c_size_t, c_void_p
from enum import Enum

import numpy as np

from puzzlelib_amd import lib, driver, lazy, fusion
from puzzlelib_amd.lib import HipError, ConvDesc, PoolDesc
from puzzlelib_amd.driver import streamHandle
from puzzlelib_amd.gpuarray import GPUArray, prod, eltwise, contiguousStrides
from puzzlelib_amd.common import (
	ConvFwdAlgo, ConvBwdFilterAlgo, ConvBwdDataAlgo, PoolMode, SoftMaxMode, BatchNormMode, LRNMode, RNNMode, DirectionMode, RNNAlgo,
	GroupFormat, ConvPerf, toAlgoId, pair, requireF32, rptrOf
)


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
		self.earlyReady = None       # (event, gradient allocation, data allocation, their versions): see convNdBackwardData
		self.poolBnCache = {}
		self.packCache = weakref.WeakKeyDictionary()        # allocation of a filter -> {(offset, pass, algo, geometry): PackEntry}
		self.packEpoch = lib.modeEpoch
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
	descEpoch = 0        # lib.modeEpoch the cached descriptors (and the sizes kept on them) were made under

	@staticmethod
	def convDesc(dataShape, Wshape, stride, pad, dilation, groups):
		"""The library's descriptor of a 2-D convolution; `.key` = its fields as a tuple, `.geo` = what the library
		answered about it per (pass, algo) (convGeometry). One object per distinct argument list: small networks are bound
		by the host's call rate, and building / hashing descriptors was a tenth of a convolution call."""
		if DnnContext.descEpoch != lib.modeEpoch:     # the math mode / Winograd tile changed (by anyone): cached sizes are stale
			DnnContext.descCache.clear()
			DnnContext.descEpoch = lib.modeEpoch
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
		if self.packEpoch != lib.modeEpoch:           # layouts of prepared operands follow the math mode / Winograd tile
			self.packCache.clear()
			self.packEpoch = lib.modeEpoch
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


	def workspaceWithPrepared(self, desc, which, algo):
		"""workspace bytes of a pass that is handed its prepared filter operand (no room for a second copy of the packed filters)"""
		hit = desc.geo.get((which, algo, "pre"))
		if hit is None:
			size = c_size_t(0)
			lib.pz_conv2d_workspace_bytes_pre(byref(desc), which, algo, byref(size))
			hit = desc.geo[(which, algo, "pre")] = size.value
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

		# (the filter is identified by its allocation OBJECT and offset, not by its address: a network built later in a
		# recycled address range starts with no history, so a network's n-th step takes the same path in every process)
		wroot = W.gpudata.root
		key = (wroot, W.gpudata.ptr - wroot.ptr)
		policy = DnnContext.convStatsPolicy
		want = lazy.on("convstats") and not given and (
			policy == "always" or (policy == "adaptive" and key[1] in self.statsWanted.get(wroot, ()))
		)

		# Conv2D(useBias) -> Activation(relu) (TestLib/CnnCifar10NIN.py:16-45): the launch waits for the ReLU that may follow
		# and takes it into its epilogue (fusion.ConvFwd; kernels.absorbRelu). Whoever reads the tensor first runs it.
		if bias is not None and not given and not want and lazy.on("convrelu") and lazy.whole(out) and \
				self.epilogueSupported(desc, lib.CONV_FWD, algo):
			lazy.attach(out, fusion.ConvFwd(self, desc, algo, data, W, bias))
			if lazy.enabled:
				# (a BatchNorm reading this tensor must still find out which filter made it: the adaptive policy registers the
				# filter in statsWanted and the NEXT step's convolution leaves strip sums instead of waiting for a ReLU)
				lazy.setFact(out, "fromconv", key)
			return out

		stats = None
		if want and nstrips > 0:
			stats = GPUArray.empty((W.shape[0], nstrips, 4), dtype=np.float32, allocator=allocator)
		# the input is a BatchNorm (+ in-place ReLU) that was only described, and this layer can apply it while gathering: the
		# normalised tensor stays unwritten (its other reader, this layer's filter gradient, does the same)
		xbn = self.describedInput(data)
		if xbn is not None and not self.xbnSupported(desc, lib.CONV_FWD, algo):
			xbn = None
		self.launchForward(desc, algo, data, W, bias, out, False, stats=stats, allocator=allocator, xbn=xbn)
		if stats is not None:
			lazy.setFact(out, "convstats", (stats, outshape))
			lazy.count("conv_stats")

		if lazy.enabled and not given:
			lazy.setFact(out, "fromconv", key)
		return out


	def launchForward(self, desc, algo, data, W, bias, out, relu, stats=None, allocator=None, prepared=True, settling=False, xbn=None):
		"""the forward launch itself (out: the whole output, overwritten). prepared=False: W is not a live parameter (a snapshot):
		its packed operand is made inside the call instead of being kept per parameter version. settling=True: called while
		`out`'s own description runs (lazy.settle) — its address is taken without a write barrier (the barrier that led here
		has done that work; another one would run `out`'s dependents BEFORE this launch wrote what they read)."""
		optr = out.gpudata.ptr if settling else out.optr
		wsbytes = self.convGeometry(desc, lib.CONV_FWD, algo)[2]
		packed = self.prepared(W, desc, lib.CONV_FWD, algo) if prepared else None
		if packed is not None:
			wsbytes = self.workspaceWithPrepared(desc, lib.CONV_FWD, algo)
		ws = self.workspace(wsbytes, allocator)
		if xbn is not None:
			assert not relu
			lib.pz_conv2d_fwd_xbn(
				byref(desc), xbn.x.rptr, fusion.raw(xbn.coef), int(xbn.relu), None if packed is not None else W.rptr, packed, rptrOf(bias),
				optr, None if stats is None else stats.optr, algo, rptrOf(ws), wsbytes, None
			)
			lazy.count("conv_xbn")
		elif relu:
			assert stats is None
			lib.pz_conv2d_fwd_relu(
				byref(desc), data.rptr, None if packed is not None else W.rptr, packed, rptrOf(bias), optr, algo, rptrOf(ws),
				wsbytes, None
			)
		elif packed is not None:
			lib.pz_conv2d_fwd_pre(
				byref(desc), data.rptr, packed, rptrOf(bias), optr, None if stats is None else stats.optr, algo, rptrOf(ws),
				wsbytes, None
			)
		elif stats is None:
			lib.pz_conv2d_fwd(byref(desc), data.rptr, W.rptr, rptrOf(bias), optr, algo, rptrOf(ws), wsbytes, None)
		else:
			lib.pz_conv2d_fwd_stats(
				byref(desc), data.rptr, W.rptr, rptrOf(bias), optr, stats.optr, algo, rptrOf(ws), wsbytes, None
			)


	def launchBackwardData(self, desc, algo, grad, W, out, gate=None, allocator=None, prepared=True, settling=False):
		optr = out.gpudata.ptr if settling else out.optr
		wsbytes = self.convGeometry(desc, lib.CONV_BWD_DATA, algo)[2]
		if gate is not None:
			ws = self.workspace(wsbytes, allocator)
			lib.pz_conv2d_bwd_data_gate(byref(desc), grad.rptr, W.rptr, gate.rptr, optr, algo, rptrOf(ws), wsbytes, None)
			return ws
		packed = self.prepared(W, desc, lib.CONV_BWD_DATA, algo) if prepared else None
		if packed is not None:
			wsbytes = self.workspaceWithPrepared(desc, lib.CONV_BWD_DATA, algo)
			ws = self.workspace(wsbytes, allocator)
			lib.pz_conv2d_bwd_data_pre(byref(desc), grad.rptr, packed, optr, algo, rptrOf(ws), wsbytes, None)
		else:
			ws = self.workspace(wsbytes, allocator)
			lib.pz_conv2d_bwd_data(byref(desc), grad.rptr, W.rptr, optr, algo, rptrOf(ws), wsbytes, None)
		return ws          # (the caller may have to keep it from being handed out again: markBeforeBackwardData)


	def backwardStatsTarget(self, desc, algo, data):
		"""(x, coef, mean, bytes of the partials) when `data` — the input of the layer whose backward-data is about to run — is
		y = relu(a x + b), the described (or written) output of a BatchNorm over x with known coefficients and saved mean, and
		the launch has the statistics epilogue (pz_conv2d_bwd_data_bnstats); else None"""
		if not isinstance(data, GPUArray) or data.ndim != 4:
			return None
		info = lazy.fact(data, "bnapply")
		if info is None:
			waiting = lazy.pending(data, fusion.BnApply)
			info = None if waiting is None else (waiting.x, waiting.coef, waiting.relu)
		if info is None or not info[2] or tuple(info[0].shape) != tuple(data.shape):
			return None
		mean = lazy.fact(info[0], "bnsaved")
		if mean is None:
			return None
		nbytes = desc.geo.get((lib.CONV_BWD_DATA, algo, "bst"))
		if nbytes is None:
			size = c_size_t(0)
			lib.pz_conv2d_bwd_data_bnstats_bytes(byref(desc), algo, byref(size))
			nbytes = desc.geo[(lib.CONV_BWD_DATA, algo, "bst")] = size.value
		return (info[0], info[1], mean, nbytes) if nbytes > 0 else None


	def epilogueSupported(self, desc, which, algo):
		"""can this pass take an activation into its epilogue (pz_conv2d_fwd_relu / pz_conv2d_bwd_data_gate)?"""
		hit = desc.geo.get((which, algo, "epi"))
		if hit is None:
			flag = c_int(0)
			lib.pz_conv2d_epilogue_supported(byref(desc), which, algo, byref(flag))
			hit = desc.geo[(which, algo, "epi")] = bool(flag.value)
		return hit


	def convAlgoUsed(self, desc, which, algo):
		"""The kernel family (`direct` / `winograd` / `implicitGemm` id) a request resolves to."""
		used = c_int(0)
		lib.pz_conv2d_algo_used(byref(desc), which, toAlgoId(algo), byref(used))
		return used.value


	def xbnSupported(self, desc, which, algo):
		"""can this pass evaluate a preceding BatchNorm (+ ReLU) while it gathers its input (pz_conv2d_fwd_xbn / _bwd_filter_xbn)?"""
		hit = desc.geo.get((which, algo, "xbn"))
		if hit is None:
			flag = c_int(0)
			lib.pz_conv2d_xbn_supported(byref(desc), which, algo, byref(flag))
			hit = desc.geo[(which, algo, "xbn")] = bool(flag.value)
		return hit


	@staticmethod
	def describedInput(data):
		"""`data` as a convolution may read it without making it exist: the description relu?(a * x + b) still pending on it
		(fusion.BnApply over a tensor of data's own shape), or None"""
		if not lazy.on("xbn") or not isinstance(data, GPUArray) or data.ndim != 4:
			return None
		waiting = lazy.pending(data, fusion.BnApply)
		if waiting is None or waiting.x.shape != data.shape or not waiting.x.contiguous:
			return None
		return waiting


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

		given = out is not None
		out = GPUArray.empty(inshape, dtype=grad.dtype, allocator=allocator) if out is None else out

		# the gradient is the un-written input gradient of a BatchNorm (fusion.BnBwdApply): evaluate it while gathering
		bn = lazy.pending(grad, fusion.BnBwdApply) if lazy.on("bnbwdfold") else None
		bst = self.backwardStatsTarget(desc, algo, data) if (lazy.on("dgradstats") and not given and bias is None) else None
		if bst is not None:
			# this layer read y = relu(bn(x)), never written: its input gradient goes through that ReLU's derivative and into that
			# BatchNorm's backward next — the epilogue that holds the gradient tile sums the gated gradient's statistics
			x2, coef2, mean2, nbytes = bst
			fold = bn is not None and foldable
			if not fold:
				self.markBeforeBackwardData(grad, W, data)
			parts = GPUArray.empty((nbytes, ), dtype=np.uint8, allocator=allocator)
			ws = self.workspace(wsbytes, allocator)
			lib.pz_conv2d_bwd_data_bnstats(
				byref(desc), bn.dy.rptr if fold else grad.rptr, bn.x.rptr if fold else None, fusion.raw(bn.coef) if fold else None,
				W.rptr, out.optr, x2.rptr, fusion.raw(coef2), mean2.rptr, parts.optr, algo, rptrOf(ws), wsbytes, None
			)
			lazy.setFact(out, "gatedparts", (data, x2, coef2, mean2, parts), sources=(x2, mean2))
			lazy.count("dgrad_bnstats")
			if fold:
				lazy.count("dgrad_bn_fold")
			elif self.earlyReady is not None:
				self.earlyReady += (ws, )
		elif bn is not None and foldable:
			ws = self.workspace(wsbytes, allocator)
			lib.pz_conv2d_bwd_data_bn(
				byref(desc), bn.dy.rptr, bn.x.rptr, fusion.raw(bn.coef), W.rptr, out.optr, algo, rptrOf(ws), wsbytes, None
			)
			lazy.count("dgrad_bn_fold")
		elif given is False and bias is None and data is not None and lazy.on("convgate") and lazy.whole(out) and \
				isinstance(data, GPUArray) and self.epilogueSupported(desc, lib.CONV_BWD_DATA, algo) and (
					lazy.fact(data, "convrelu") or getattr(lazy.pending(data, fusion.ConvFwd), "relu", False)):
			# this convolution's input came out of a ReLU fused into the convolution in front: that ReLU's backward
			# (reluDerKer on this very gradient) comes next and joins the launch as its epilogue (kernels.absorbReluDer)
			lazy.attach(out, fusion.ConvBwdData(self, desc, algo, grad, W))
			return out
		else:
			self.markBeforeBackwardData(grad, W, data)
			ws = self.launchBackwardData(desc, algo, grad, W, out, allocator=allocator)
			if self.earlyReady is not None:
				# a launch that waits for the mark only may run NEXT to this kernel: the pool must not hand it this kernel's
				# workspace (freed here, on the host, it would be the next block of its size class)
				self.earlyReady += (ws, )

		if bias is not None:           # deconvolution forward: bias over the produced maps, rows of the (n*maps, pixels) view
			assert bias.size == out.shape[1]
			lib.pz_bias_add(
				out.wptr, out.rptr, bias.rptr, 1, out.shape[0] * out.shape[1], prod(out.shape[2:]), out.shape[1], 0, None
			)

		return out


	def markBeforeBackwardData(self, grad, W, data):
		"""The filter gradient of this layer comes next (Modules/ConvND.py:84-95: updateGrad, then accGradParams on the same
		gradient) and, on the filter-gradient stream, has to wait for the producers of `grad` and `data` — not for this
		layer's backward-data kernel, which it would if it waited for the main stream's position at ITS call. So the position
		is marked here, in front of the backward-data launch, together with the two allocations' write versions; the
		filter-gradient call uses the mark if nothing touched them since (convNdBackwardParams). NiN b128: the 5x5 layer's
		filter gradient no longer starts 0.38 ms late, the first layer's not behind its input gradient."""
		self.earlyReady = None
		if not isinstance(data, GPUArray) or not lazy.on("sidestream") or not lazy.on("earlyready") or self.convMath != "f32" or \
				self.sideWorkMean > self.sideStreamMaxGflop:
			return
		grad.rptr, W.rptr                                    # whatever is pending is launched in FRONT of the mark
		groot, droot = grad.gpudata.root, data.gpudata.root
		glz, dlz = lazy.stateOf(groot), lazy.stateOf(droot)
		if glz.thunk is not None or dlz.thunk is not None:
			return
		event = lazy.newEvent()
		event.record(None)
		self.earlyReady = (event, groot, droot, glz.version, dlz.version)


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

		fresh = wgrad is None or (withbias and bgrad is None)      # destinations allocated (and, in debug mode, filled) by this call
		wgrad = GPUArray.empty(W.shape, dtype=W.dtype, allocator=allocator) if wgrad is None else wgrad

		algo = toAlgoId(algo)
		_, _, wsbytes, _, foldable = self.convGeometry(desc, lib.CONV_BWD_FILTER, algo)

		bg = None
		if withbias:
			bg = GPUArray.empty((biasof.shape[1], ), dtype=data.dtype, allocator=allocator) if bgrad is None else bgrad

		fused = withbias and bcoef == wcoef and not deconv    # one library call reduces dw and db with the same (alpha, beta)
		bn = lazy.pending(grad, fusion.BnBwdApply) if lazy.on("bnbwdfold") else None
		folded = bn is not None and not withbias and foldable
		# the layer's input was never written (a described BatchNorm + ReLU, see convNd): this pass reads the BatchNorm's input too
		xbn = self.describedInput(data) if not withbias and not deconv else None
		if xbn is not None and not self.xbnSupported(desc, lib.CONV_BWD_FILTER, algo):
			xbn = None

		gflop = 2e-9 * prod(grad.shape) * prod(W.shape[1:])
		side = self.filterGradStream(gflop) if (not withbias or fused) else None
		st = side.handle if side is not None else None
		src = data if xbn is None else xbn.x
		reads = [src, bn.dy, bn.x] if folded else [src, grad]
		writes = [wgrad] + ([bg] if fused else [])

		def rp(ary):
			return ary.rptr if side is None else ary.ptrOn(side, False)

		def wp(ary):
			return ary.wptr if side is None else ary.ptrOn(side, True)

		# The launch follows the main stream from the mark in front of this layer's backward-data kernel (markBeforeBackwardData)
		# when nothing touched what it reads since, and what it writes was not produced by this call on the main stream.
		early = self.earlyReady          # (kept until this call's own allocations are made: it holds backward-data's workspace)
		mark = None
		if side is not None and early is not None and not folded and xbn is None and not fresh and grad.gpudata.root is early[1] and \
				data.gpudata.root is early[2] and lazy.cleanSince(early[1], early[3]) and lazy.cleanSince(early[2], early[4]) and \
				all(lazy.quiet(a) for a in writes):          # (a settle — a pending zero fill of the arena — would be behind the mark)
			mark = early[0]
			lazy.count("wgrad_early_start")

		# the workspace: the debug allocator's NaN fill is a main-stream launch BEHIND the mark, so with a mark it is made on
		# the launch's own stream instead
		poison = mark is not None and GPUArray.debugFill and wsbytes >= 4
		if poison:
			GPUArray.debugFill = False
		try:
			ws = self.workspace(wsbytes, allocator)
		finally:
			if poison:
				GPUArray.debugFill = True

		rptrs = [rp(a) for a in reads]
		wptrs = [wp(a) for a in writes]
		ready = lazy.foreignBegin(side, mark) if side is not None else None
		if poison:
			lib.pz_memset_d32(ws.gpudata.ptr, 0x7fc00000, wsbytes // 4, st)

		if xbn is not None:
			lib.pz_conv2d_bwd_filter_xbn(
				byref(desc), rptrs[0], fusion.raw(xbn.coef), int(xbn.relu), rptrs[1], rptrs[2] if folded else None,
				fusion.raw(bn.coef) if folded else None, wptrs[0], wcoef[0], wcoef[1], algo, rptrOf(ws), wsbytes, st
			)
			lazy.count("wgrad_xbn")
			if folded:
				lazy.count("wgrad_bn_fold")
		elif folded:
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

		self.earlyReady = early = None
		if side is not None:
			lazy.foreignEnd(side, ready, reads=reads, writes=writes, keep=(ws, bn.coef if folded else None, xbn.coef if xbn is not None else None))

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
			parts = gate.parts if desc is not None and lazy.on("dgradstats") else None
			if parts is not None and desc[2] and lazy.sameBuffer(desc[0], data) and lazy.sameBuffer(parts[1], data) and \
					parts[2] is desc[1] and lazy.sameBuffer(parts[3], savemean):
				# the backward-data launch that wrote this gradient summed the gated gradient's statistics in its epilogue
				lib.pz_bn_bwd_gate_from_partials(
					data.rptr, lazy.rawRead(grad), out.optr, n, c, hw, scale.rptr, savemean.rptr, saveinvvar.rptr,
					scalegrad.optr, bgrad.optr, fusion.raw(desc[1]), fusion.raw(parts[4]), None
				)
				lazy.count("bn_bwd_gate_from_partials")
				return out, scalegrad, bgrad
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
			from puzzlelib_amd.modules import MemModule
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
			from puzzlelib_amd.modules import MemModule
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
