"""
Winograd (F(2x2, 3x3) or F(4x4, 3x3): --tile) against the implicit GEMM on the device: max abs difference and time per pass
for the 3x3 layers of the ResNet-50 census plus ragged shapes. Development tool (parity proper lives in tests/).

    python tools/wino_check.py [--reps 10] [--tile 0|2|4]
"""
import argparse, os, sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [      # n, c, h, w, k, pad
	(2, 8, 6, 6, 8, 1), (3, 12, 7, 9, 20, 1), (2, 16, 5, 8, 70, 0), (1, 4, 11, 3, 5, 2),
	(256, 64, 55, 55, 64, 1), (256, 128, 28, 28, 128, 1), (256, 256, 14, 14, 256, 1), (256, 512, 7, 7, 512, 1),
	(128, 64, 56, 56, 128, 1),
]


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--reps", type=int, default=10)
	ap.add_argument("--only", type=int, default=-1)
	ap.add_argument("--tile", type=int, default=0)
	args = ap.parse_args()

	from puzzlelib_amd import backend, lib
	bnd = backend.getBackend(0, initmode=2)
	G, dnn = bnd.GPUArray, bnd.dnn
	dnn.setWinogradTile(args.tile)
	rng = np.random.RandomState(0)

	def timed(fn):
		fn()
		lib.pz_device_sync()
		start, end = bnd.Driver.Event(), bnd.Driver.Event()
		start.record()
		for _ in range(args.reps):
			fn()
		end.record()
		end.synchronize()
		return start.timeTill(end) / args.reps

	for idx, (n, c, h, w, k, pad) in enumerate(SHAPES):
		if args.only >= 0 and idx != args.only:
			continue
		x = G.toGpu(rng.randn(n, c, h, w).astype(np.float32))
		wt = G.toGpu((rng.randn(k, c, 3, 3) / np.sqrt(9 * c)).astype(np.float32))
		b = G.toGpu(rng.randn(k).astype(np.float32))

		y5 = dnn.convNd(x, wt, b, 1, pad, 1, 1, 5)
		y3 = dnn.convNd(x, wt, b, 1, pad, 1, 1, 3)
		ef = float(np.abs(y5.get() - y3.get()).max())

		dy = G.toGpu(rng.randn(*y5.shape).astype(np.float32))
		d5 = dnn.convNdBackwardData(dy, wt, None, x, 1, pad, 1, 0, 1, 5)
		d3 = dnn.convNdBackwardData(dy, wt, None, x, 1, pad, 1, 0, 1, 3)
		eb = float(np.abs(d5.get() - d3.get()).max())

		w5 = dnn.convNdBackwardParams(x, dy, wt, 1, pad, 1, 1, False, False, None, None, 1.0, 0.0, 5)
		w3 = dnn.convNdBackwardParams(x, dy, wt, 1, pad, 1, 1, False, False, None, None, 1.0, 0.0, 3)
		ew = float(np.abs(w5.get() - w3.get()).max() / max(1e-30, np.abs(w5.get()).max()))

		flops = 2.0 * n * y5.shape[2] * y5.shape[3] * k * c * 9
		line = "(%d,%d,%d,%d)->%d p%d  fwd err %.2e  dgrad err %.2e  wgrad rel err %.2e" % (n, c, h, w, k, pad, ef, eb, ew)
		if n >= 64:
			t5 = timed(lambda: dnn.convNd(x, wt, b, 1, pad, 1, 1, 5, y5))
			t3 = timed(lambda: dnn.convNd(x, wt, b, 1, pad, 1, 1, 3, y3))
			u5 = timed(lambda: dnn.convNdBackwardData(dy, wt, None, x, 1, pad, 1, 0, 1, 5, d5))
			u3 = timed(lambda: dnn.convNdBackwardData(dy, wt, None, x, 1, pad, 1, 0, 1, 3, d3))
			v5 = timed(lambda: dnn.convNdBackwardParams(x, dy, wt, 1, pad, 1, 1, False, False, w5, None, 1.0, 0.0, 5))
			v3 = timed(lambda: dnn.convNdBackwardParams(x, dy, wt, 1, pad, 1, 1, False, False, w3, None, 1.0, 0.0, 3))
			line += " | fwd igemm %.3f ms (%.0f TF) wino %.3f ms (%.0f TF-eq) | dgrad igemm %.3f wino %.3f | wgrad igemm %.3f wino %.3f" % (
				t5, flops / t5 / 1e9, t3, flops / t3 / 1e9, u5, u3, v5, v3
			)
		print(line, flush=True)


if __name__ == "__main__":
	main()
