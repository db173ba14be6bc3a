import os, sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
	if p not in sys.path:
		sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
	config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
	def __init__(self, name):
		self.z = np.load(os.path.join(GOLDEN, name))

	def __getitem__(self, key):
		return self.z[key]

	def keys(self):
		return self.z.files


@pytest.fixture(scope="session")
def ops():
	return Golden("ops.npz")


@pytest.fixture(scope="session")
def lenet_golden():
	return Golden("lenet.npz")


@pytest.fixture(scope="session")
def mini_golden():
	return Golden("miniresnet.npz")


@pytest.fixture(scope="session")
def bnd():
	"""The MI355X backend object (initmode=2). GPU tests only: raises if the native library or the device is missing —
	there is no fallback to hide behind."""
	from puzzlelib_amd import backend, gpuarray
	# the reference's unit-test runner poisons fresh allocations (Cuda/Utils.py:97-114, Unittester.py:52-55): every GPU
	# test here runs with NaN-filled `empty()` buffers, so a kernel reading memory nobody wrote shows up as NaNs
	gpuarray.GPUArray.debugFill = os.environ.get("PUZZLE_MI355_DEBUG_ALLOC", "1") == "1"
	bnd_ = backend.getBackend(0, initmode=2)
	# the backend seeds its global generator from numpy's (unseeded) global state, as the reference does
	# (Cuda/GPUBackend.py: globalRng seeded from np.random.randint): every test run here draws the same device words —
	# re-seeded in place, so that whoever already holds the object (surface.bound()) sees the same generator
	bnd_.globalRng.__init__(seed=0x5eed2026)
	return bnd_


def assert_close(actual, desired, atol=1e-5, rtol=1e-5, what=""):
	actual, desired = np.asarray(actual), np.asarray(desired)
	assert actual.shape == desired.shape, "%s: shape %s vs %s" % (what, actual.shape, desired.shape)
	err = np.abs(actual.astype(np.float64) - desired.astype(np.float64))
	tol = atol + rtol * np.abs(desired.astype(np.float64))
	bad = err > tol
	assert not bad.any(), "%s: %d/%d elements off, max abs err %.3e (tol atol=%g rtol=%g)" % (
		what, int(bad.sum()), bad.size, float(err.max()), atol, rtol
	)
