"""
GPU test of the drop-in boundary: the MI355X backend object offers every attribute PuzzleLib's dispatch surface reads
from `Hip.Backend` (list extracted from the reference's Backend/*.py by oracle/list_backend_attrs.py and committed as
tests/golden/backend_attrs.json), with the call signatures the wrappers use (Backend/Dnn.py:131,135,179-193,238-253).
"""
import inspect, json, os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_backend_object_covers_the_dispatch_surface(bnd):
	contract = json.load(open(os.path.join(GOLDEN, "backend_attrs.json")))

	missing = [name for name in contract["backend"] if not hasattr(bnd, name)]
	assert not missing, "backend object lacks: %s" % missing

	for obj in ("blas", "dnn", "matmod", "costmod", "GPUArray"):
		target = getattr(bnd, obj)
		missing = [name for name in contract[obj] if not hasattr(target, name)]
		assert not missing, "%s lacks: %s" % (obj, missing)

	for name in ("allocate", "freeHeld", "getStats"):
		assert hasattr(bnd.memoryPool, name)


def test_positional_signatures_match_the_wrappers(bnd):
	def params(fn):
		return [p for p in inspect.signature(fn).parameters]

	# Backend/Dnn.py:179-193
	assert params(bnd.dnn.convNd) == ["data", "W", "bias", "stride", "pad", "dilation", "groups", "algo", "out", "allocator"]
	assert params(bnd.dnn.convNdBackwardData) == [
		"grad", "W", "bias", "data", "stride", "pad", "dilation", "postpad", "groups", "algo", "out", "allocator"
	]
	assert params(bnd.dnn.convNdBackwardParams) == [
		"data", "grad", "W", "stride", "pad", "dilation", "groups", "withbias", "deconv", "wgrad", "bgrad", "scale",
		"momentum", "algo", "allocator"
	]
	# Backend/Dnn.py:131,135
	assert params(bnd.dnn.poolNd) == ["data", "size", "stride", "pad", "mode", "test", "out", "allocator"]
	assert params(bnd.dnn.poolNdBackward) == [
		"grad", "indata", "outdata", "workspace", "size", "stride", "pad", "mode", "out", "allocator"
	]
	# Backend/Dnn.py:238-253
	assert params(bnd.dnn.batchNormNd)[:9] == ["data", "mean", "var", "scale", "bias", "epsilon", "factor", "test", "mode"]
	assert params(bnd.dnn.batchNormNdBackward)[:7] == ["grad", "data", "scale", "savemean", "saveinvvar", "epsilon", "mode"]
	# Backend/Blas.py:61
	assert params(bnd.blas.gemm) == ["A", "B", "out", "transpA", "transpB", "alpha", "beta", "allocator"]
	assert params(bnd.matmod.matsum) == ["tensor", "axis", "out", "alpha", "beta", "allocator"]
	assert params(bnd.costmod.crossEntropy) == ["scores", "labels", "weights", "error", "allocator"]

	for enum in ("ConvFwdAlgo", "ConvBwdDataAlgo", "ConvBwdFilterAlgo"):
		assert isinstance(getattr(bnd, enum).auto.value, int)
	assert {m.name for m in bnd.PoolMode} == {"max", "avgWithPad", "avgNoPad"}
	assert bnd.dtypesSupported() == [(np.float32, 1e-5)]
	assert isinstance(bnd.device.name(), str) and len(bnd.device.name()) > 0
	assert bnd.device.arch().startswith("gfx950")


def test_out_of_scope_entries_fail_loudly(bnd):
	x = bnd.GPUArray.zeros((2, 3, 4, 4), dtype=np.float32)
	for call in (lambda: bnd.dnn.lrn(x), lambda: bnd.createRnn(4, 4, np.float32), lambda: bnd.blas.gemmBatched(x, x),
				 lambda: bnd.poolmod.maxpool2d(x), lambda: bnd.instanceNorm2d(x, x, x)):
		with pytest.raises(NotImplementedError):
			call()


def test_rccl_single_rank_roundtrip(bnd):
	"""World size 1 exercises the RCCL plumbing (unique id, communicator, all-reduce, broadcast) on the one GPU the
	test box has; the multi-rank arithmetic is covered on CPU by tests/test_dp_gloo.py."""
	import ctypes
	from puzzlelib_amd import lib, grid

	buf = ctypes.create_string_buffer(lib.COMM_ID_BYTES)
	lib.pz_comm_unique_id(buf)
	node = grid.RcclNodeInfo(0, 1, 0, buf.raw)

	rng = np.random.RandomState(0)
	host = rng.randn(1 << 16).astype(np.float32)
	g = bnd.GPUArray.toGpu(host)

	node.sumTensor("grad", g)                      # N = 1: mean == identity
	assert np.array_equal(g.get(), host)

	blocks = [("a", 0, 1 << 17), ("b", 1 << 17, 1 << 17)]
	reducer = node.attach("grad", g, blocks)
	reducer.beginStep()
	reducer.variableReady("b")
	reducer.variableReady("a")
	node.sumTensor("grad", g)
	assert np.array_equal(g.get(), host)

	node.broadcastBuffer("data", g.gpudata)
	assert np.array_equal(g.get(), host)
	node.close()
