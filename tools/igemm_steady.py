"""Asymptotic rate of the implicit-GEMM forward kernel: 1x1 convolutions large enough that launch-level effects (ramp,
in-phase prologues / epilogues, the last partial round) vanish next to the k-loop. Kernel-tuning aid only."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from puzzlelib_amd import backend, lib, lazy
lazy.disabled.add("sidestream")
bnd = backend.getBackend(0, initmode=2)
G = bnd.GPUArray
rng = np.random.RandomState(0)
SHAPES = [(1024, 256, 14, 256), (4096, 256, 14, 256), (1024, 1024, 14, 256), (1024, 1024, 28, 256), (256, 1024, 28, 256),
		  (256, 1024, 14, 256), (64, 256, 56, 256), (512, 128, 28, 256)]
for c, k, hw, n in SHAPES:
	x = G.toGpu(rng.randn(n, c, hw, hw).astype(np.float32))
	W = G.toGpu((rng.randn(k, c, 1, 1) / np.sqrt(c)).astype(np.float32))
	fn = lambda: bnd.dnn.convNd(x, W, None, 1, 0, allocator=bnd.memoryPool)
	fn(); lib.pz_device_sync()
	start, end = bnd.Driver.Event(), bnd.Driver.Event()
	reps = 10
	start.record()
	for _ in range(reps):
		fn()
	end.record(); end.synchronize()
	ms = start.timeTill(end) / reps
	gf = 2.0 * n * k * hw * hw * c / 1e9
	tiles = ((k + 127) // 128) * ((n * hw * hw + 127) // 128)
	print("C=%5d K=%5d %2dx%2d N=%d: %8.3f ms %7.1f TFLOP/s  (%d tiles of 128x128 = %.2f per CU, %d k-tiles)" % (
		c, k, hw, hw, n, ms, gf / ms, tiles, tiles / 256.0, c // 16))
	del x, W
