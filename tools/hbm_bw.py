"""
What the streaming kernels reach on this HBM (development tool): element-wise kernels on 1 GiB operands, and the
channel-walking batch-norm inference kernel (read 1, write 1) on aligned and odd-sized planes.

    python tools/hbm_bw.py
"""
import os, sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
	from puzzlelib_amd import backend, lib
	bnd = backend.getBackend(0, initmode=2)
	G = bnd.GPUArray

	def timed(fn, reps=10):
		fn()
		lib.pz_device_sync()
		s, e = bnd.Driver.Event(), bnd.Driver.Event()
		s.record()
		for _ in range(reps):
			fn()
		e.record()
		e.synchronize()
		return s.timeTill(e) / reps

	n = 1 << 28
	x, y, z = (G.empty((n, ), dtype=np.float32) for _ in range(3))
	x.fill(1.0), y.fill(2.0)
	relu, add = bnd.reluKer(np.float32), bnd.addKer(np.float32)
	t = timed(lambda: relu(y, x))
	print("relu  r1 w1: %.3f ms  %.2f TB/s" % (t, 2 * 4 * n / t / 1e9))
	t = timed(lambda: add(z, x, 1.0, y, 1.0))
	print("add   r2 w1: %.3f ms  %.2f TB/s" % (t, 3 * 4 * n / t / 1e9))
	del x, y, z

	for shape in [(256, 256, 55, 55), (256, 256, 56, 56), (256, 512, 28, 28), (256, 1024, 14, 14), (256, 2048, 7, 7), (256, 2048, 8, 8)]:
		c = shape[1]
		a = G.empty(shape, dtype=np.float32)
		a.fill(1.0)
		mean, bi = G.zeros((c, ), dtype=np.float32), G.zeros((c, ), dtype=np.float32)
		var, sc = G.empty((c, ), dtype=np.float32), G.empty((c, ), dtype=np.float32)
		var.fill(1.0), sc.fill(1.0)
		out = G.empty(shape, dtype=np.float32)
		t = timed(lambda: bnd.dnn.batchNormNd(a, mean, var, sc, bi, 1e-5, 1.0, True, out=out), 5)
		print("bn infer r1 w1 %-22s: %.3f ms  %.2f TB/s" % (shape, t, 2 * a.nbytes / t / 1e9))
		del a, out


if __name__ == "__main__":
	main()
