"""Backward-filter of pointwise layers in the split math modes against the fp32-MFMA kernel and a float64 reference of
a few filter elements; run twice (determinism). Tuning / debugging aid."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from puzzlelib_amd import backend, lazy

lazy.disabled.add("sidestream")
bnd = backend.getBackend(0, initmode=2)
G = bnd.GPUArray
rng = np.random.RandomState(0)
shapes = [(256, 64, 55, 55, 256), (64, 256, 14, 14, 1024), (32, 1024, 14, 14, 256), (64, 512, 7, 7, 2048), (16, 64, 55, 55, 64),
          (8, 128, 28, 28, 512), (3, 48, 5, 7, 80), (2, 16, 3, 3, 16)]
for (n, c, h, w, k) in shapes:
	x = rng.randn(n, c, h, w).astype(np.float32)
	dy = rng.randn(n, k, h, w).astype(np.float32)
	W = np.zeros((k, c, 1, 1), np.float32)
	gx, gdy, gW = G.toGpu(x), G.toGpu(dy), G.toGpu(W)
	out = {}
	for mode in ("f32", "split6", "split6", "split9"):
		bnd.dnn.setConvMath(mode)
		wg = G.zeros(W.shape, dtype=np.float32)
		bnd.dnn.convNdBackwardParams(gx, gdy, gW, 1, 0, wgrad=wg, scale=1.0, momentum=0.0, allocator=bnd.memoryPool)
		out.setdefault(mode, []).append(wg.get())
	ref = np.einsum("nkp,ncp->kc", dy[:, :4].reshape(n, 4, -1).astype(np.float64), x.reshape(n, c, -1).astype(np.float64))
	scale = np.abs(ref).max()
	line = "%-28s" % ((n, c, h, w, k), )
	for mode in ("f32", "split6", "split9"):
		err = np.abs(out[mode][0][:4, :, 0, 0] - ref).max() / scale
		line += " %s err %.2e" % (mode, err)
	line += "  split6 twice equal: %s, nan: %s" % (np.array_equal(out["split6"][0], out["split6"][1]), np.isnan(out["split6"][0]).any())
	line += "  max|split6-f32|/max %.2e" % (np.abs(out["split6"][0] - out["f32"][0]).max() / np.abs(out["f32"][0]).max())
	print(line)
