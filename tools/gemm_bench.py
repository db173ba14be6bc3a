"""GEMM timings through the C ABI (NN / NT / TN): TFLOP/s against the 157.3 TFLOP/s fp32 MFMA peak.
    python tools/gemm_bench.py"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from puzzlelib_amd import backend, lib

bnd = backend.getBackend(0, initmode=2)
G = bnd.GPUArray
rng = np.random.RandomState(0)
ap = argparse.ArgumentParser()
ap.add_argument("--data", default="randn", help="randn | uniform | q12 (12 significant bits) | zeros | ones: operand values (power / clock sensitivity)")
ap.add_argument("--big", action="store_true", help="only the MFMA-bound shapes")
ap.add_argument("--loop", type=int, default=20, help="launches per timing (>= 300: clocks settled under the kernel's own load; 20-launch bursts read up to 10 %% off either way)")
args = ap.parse_args()


def values(*shape):
	if args.data == "randn": return rng.randn(*shape).astype(np.float32)
	if args.data == "uniform": return rng.uniform(-1, 1, shape).astype(np.float32)
	if args.data == "q12": return (rng.randint(0, 4096, shape) / 4096.0 - 0.5).astype(np.float32)
	if args.data == "zeros": return np.zeros(shape, np.float32)
	return np.ones(shape, np.float32)


SHAPES = [(256, 2048, 1000), (64, 800, 1024), (256, 1000, 2048), (2048, 256, 1000), (1024, 1024, 1024), (4096, 4096, 4096),
		  (8192, 1024, 8192), (128, 25088, 4096)]
if args.big: SHAPES = [(4096, 4096, 4096), (8192, 1024, 8192), (2048, 2048, 2048), (1024, 256, 50176), (512, 128, 200704)]
print("data: %s" % args.data)
print("%-22s %-3s %10s %9s %7s" % ("M x K x N", "op", "us", "TFLOP/s", "of peak"))
for m, k, n in SHAPES:
	for tag, ta, tb in (("NN", False, False), ("NT", False, True), ("TN", True, False)):
		A = G.toGpu(values(*((k, m) if ta else (m, k))))
		B = G.toGpu(values(*((n, k) if tb else (k, n))))
		out = G.empty((m, n), dtype=np.float32)
		fn = lambda: bnd.blas.gemm(A, B, out, ta, tb, 1.0, 0.0, bnd.memoryPool)
		secs, _ = bnd.timeKernel(fn, (), looplength=args.loop, log=False, normalize=True)
		tf = 2.0 * m * n * k / secs / 1e12
		print("%-22s %-3s %10.1f %9.1f %6.0f%%" % ("%d x %d x %d" % (m, k, n), tag, secs * 1e6, tf, tf / 157.3 * 100))
