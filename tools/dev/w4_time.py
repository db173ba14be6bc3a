"""development aid: time of the Winograd forward on the four 3x3 layer shapes of ResNet-50 (b256), tile from --tile"""
import os, sys, argparse
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ap = argparse.ArgumentParser(); ap.add_argument("--tile", type=int, default=4); ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--bwd", action="store_true")
args = ap.parse_args()
from puzzlelib_amd import backend, lib
bnd = backend.getBackend(0, initmode=2)
G, dnn = bnd.GPUArray, bnd.dnn
dnn.setWinogradTile(args.tile)
rng = np.random.RandomState(0)
out = []
for (n, c, h, w, k) in [(256, 64, 55, 55, 64), (256, 128, 28, 28, 128), (256, 256, 14, 14, 256), (256, 512, 7, 7, 512)]:
	x = G.toGpu(rng.randn(n, c, h, w).astype(np.float32)); wt = G.toGpu((rng.randn(k, c, 3, 3) / np.sqrt(9 * c)).astype(np.float32))
	y = dnn.convNd(x, wt, None, 1, 1, 1, 1, 3)
	def run():
		if args.bwd:
			dnn.convNdBackwardParams(x, y, wt, 1, 1, 1, 1, False, False, wt0, None, 1.0, 0.0, 3)
		else:
			dnn.convNd(x, wt, None, 1, 1, 1, 1, 3, y)
	if args.bwd:
		wt0 = G.empty(wt.shape, np.float32)
	run(); lib.pz_device_sync()
	a, b = bnd.Driver.Event(), bnd.Driver.Event()
	a.record()
	for _ in range(args.reps): run()
	b.record(); b.synchronize()
	t = a.timeTill(b) / args.reps
	out.append("%dx%d c%d: %.1f us (%.0f TF-eq)" % (h, w, c, t * 1e3, 2.0 * n * h * w * k * c * 9 / t / 1e9))
print(" | ".join(out))
