"""GEMM timings through the C ABI (NN / NT / TN): TFLOP/s against the 157.3 TFLOP/s fp32 MFMA peak.
    python tools/gemm_bench.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from puzzlelib_amd import backend, lib

bnd = backend.getBackend(0, initmode=2)
G = bnd.GPUArray
rng = np.random.RandomState(0)
SHAPES = [(256, 2048, 1000), (64, 800, 1024), (256, 1000, 2048), (2048, 256, 1000), (1024, 1024, 1024), (4096, 4096, 4096),
		  (8192, 1024, 8192), (128, 25088, 4096)]
print("%-22s %-3s %10s %9s %7s" % ("M x K x N", "op", "us", "TFLOP/s", "of peak"))
for m, k, n in SHAPES:
	for tag, ta, tb in (("NN", False, False), ("NT", False, True), ("TN", True, False)):
		A = G.toGpu(rng.randn(*((k, m) if ta else (m, k))).astype(np.float32))
		B = G.toGpu(rng.randn(*((n, k) if tb else (k, n))).astype(np.float32))
		out = G.empty((m, n), dtype=np.float32)
		fn = lambda: bnd.blas.gemm(A, B, out, ta, tb, 1.0, 0.0, bnd.memoryPool)
		secs, _ = bnd.timeKernel(fn, (), looplength=20, log=False, normalize=True)
		tf = 2.0 * m * n * k / secs / 1e12
		print("%-22s %-3s %10.1f %9.1f %6.0f%%" % ("%d x %d x %d" % (m, k, n), tag, secs * 1e6, tf, tf / 157.3 * 100))
