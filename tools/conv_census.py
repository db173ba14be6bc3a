"""
Per-layer timing of the ResNet-50 convolution census at batch 256 on one MI355X: forward, backward-data
and backward-filter of every distinct conv shape through the C ABI, TFLOP/s per pass against the 157.3 TFLOP/s fp32 MFMA
peak. Used to decide which tile shapes / kernels to work on; not part of the parity suite.

    python tools/conv_census.py [--batch 256] [--reps 5]
"""
import argparse, ctypes, os, sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (C, H, W) -> (K, size, stride, pad), count in the network
CENSUS = [
	((3, 224, 224), (64, 7, 2, 3), 1), ((64, 55, 55), (64, 1, 1, 0), 1), ((64, 55, 55), (64, 3, 1, 1), 3),
	((64, 55, 55), (256, 1, 1, 0), 4), ((256, 55, 55), (64, 1, 1, 0), 2), ((256, 55, 55), (128, 1, 2, 0), 1),
	((128, 28, 28), (128, 3, 1, 1), 4), ((128, 28, 28), (512, 1, 1, 0), 4), ((256, 55, 55), (512, 1, 2, 0), 1),
	((512, 28, 28), (128, 1, 1, 0), 3), ((512, 28, 28), (256, 1, 2, 0), 1), ((256, 14, 14), (256, 3, 1, 1), 6),
	((256, 14, 14), (1024, 1, 1, 0), 6), ((512, 28, 28), (1024, 1, 2, 0), 1), ((1024, 14, 14), (256, 1, 1, 0), 5),
	((1024, 14, 14), (512, 1, 2, 0), 1), ((512, 7, 7), (512, 3, 1, 1), 3), ((512, 7, 7), (2048, 1, 1, 0), 3),
	((1024, 14, 14), (2048, 1, 2, 0), 1), ((2048, 7, 7), (512, 1, 1, 0), 2),
]


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--batch", type=int, default=256)
	ap.add_argument("--reps", type=int, default=5)
	ap.add_argument("--only", type=int, default=-1)
	ap.add_argument("--rotate", type=int, default=1, help="cycle over this many input tensors (and their fresh outputs): with 1 the "
					"launches of a long run re-read an Infinity-Cache-resident input; a training step never does")
	ap.add_argument("--passes", default="fwd,dgrad,wgrad")
	ap.add_argument("--config2", action="store_true", help="BASELINE.json config 2: Conv2D 3x3, 64 -> 128, 56x56, batch 128")
	ap.add_argument("--nin", action="store_true", help="BASELINE.json config 3: the nine convolutions of the CIFAR-10 NiN, batch 128, with bias")
	args = ap.parse_args()
	if args.nin:
		global CENSUS
		CENSUS = [((3, 32, 32), (192, 5, 1, 2), 1), ((192, 32, 32), (160, 1, 1, 0), 1), ((160, 32, 32), (96, 1, 1, 0), 1),
				  ((96, 16, 16), (192, 5, 1, 2), 1), ((192, 16, 16), (192, 1, 1, 0), 2), ((192, 8, 8), (192, 3, 1, 1), 1),
				  ((192, 8, 8), (192, 1, 1, 0), 1), ((192, 8, 8), (10, 1, 1, 0), 1)]
		args.batch = 128
	if args.config2:
		CENSUS, args.batch = [((64, 56, 56), (128, 3, 1, 1), 1)], 128

	from puzzlelib_amd import backend, lib, lazy
	lazy.disabled.add("sidestream")          # per-pass timings: the filter gradient runs on the timed (main) stream
	lazy.disabled.add("up2")                 # ... and stride-2 input gradients are written out in full
	bnd = backend.getBackend(0, initmode=2)
	G = bnd.GPUArray
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

	total = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
	ideal = 0.0
	print("%-34s %5s | %9s %7s | %9s %7s | %9s %7s" % ("layer", "x", "fwd ms", "TF", "dgrad ms", "TF", "wgrad ms", "TF"))

	for idx, ((c, h, w), (k, size, stride, pad), count) in enumerate(CENSUS):
		if args.only >= 0 and idx != args.only:
			continue
		n = args.batch
		x = G.toGpu(rng.randn(n, c, h, w).astype(np.float32))
		W = G.toGpu((rng.randn(k, c, size, size) / np.sqrt(c * size * size)).astype(np.float32))
		y = bnd.dnn.convNd(x, W, None, stride, pad, allocator=bnd.memoryPool)
		dy = G.toGpu(rng.randn(*y.shape).astype(np.float32))
		wg = G.zeros(W.shape, dtype=np.float32)
		p, q = y.shape[2:]
		gflop = 2.0 * n * k * p * q * c * size * size / 1e9

		passes = args.passes.split(",")
		if args.rotate > 1:
			xs = [x] + [G.toGpu(rng.randn(n, c, h, w).astype(np.float32)) for _ in range(args.rotate - 1)]
			dys = [dy] + [G.toGpu(rng.randn(*y.shape).astype(np.float32)) for _ in range(args.rotate - 1)]
			outs, state = [None] * args.rotate, [0]

			def fwd_rot():
				i = state[0] = (state[0] + 1) % args.rotate
				outs[i] = bnd.dnn.convNd(xs[i], W, None, stride, pad, allocator=bnd.memoryPool)

			def dgrad_rot():
				i = state[0] = (state[0] + 1) % args.rotate
				outs[i] = bnd.dnn.convNdBackwardData(dys[i], W, None, xs[i], stride, pad, allocator=bnd.memoryPool)
			tf = timed(fwd_rot) if "fwd" in passes else 1e9
			outs = [None] * args.rotate
			td = timed(dgrad_rot) if "dgrad" in passes else 1e9
			del xs, dys, outs
		else:
			tf = timed(lambda: bnd.dnn.convNd(x, W, None, stride, pad, allocator=bnd.memoryPool)) if "fwd" in passes else 1e9
			td = timed(lambda: bnd.dnn.convNdBackwardData(dy, W, None, x, stride, pad, allocator=bnd.memoryPool)) \
				if "dgrad" in passes else 1e9
		tw = timed(lambda: bnd.dnn.convNdBackwardParams(x, dy, W, stride, pad, wgrad=wg, scale=1.0, momentum=1.0,
													  allocator=bnd.memoryPool)) if "wgrad" in passes else 1e9

		name = "(%d,%d,%d)->(%d,%dx%d,s%d,p%d)" % (c, h, w, k, size, size, stride, pad)
		print("%-34s %5d | %9.3f %7.1f | %9.3f %7.1f | %9.3f %7.1f" % (
			name, count, tf, gflop / tf, td, gflop / td, tw, gflop / tw
		))
		total["fwd"] += tf * count
		total["dgrad"] += td * count * (0 if idx == 0 else 1)
		total["wgrad"] += tw * count
		ideal += gflop * count / 157.3

		del x, W, y, dy, wg

	print("per step: fwd %.1f ms, dgrad %.1f ms (conv1 excluded), wgrad %.1f ms, sum %.1f ms; at 157.3 TFLOP/s each pass "
		  "would take %.1f ms" % (total["fwd"], total["dgrad"], total["wgrad"], sum(total.values()), ideal))


if __name__ == "__main__":
	main()
