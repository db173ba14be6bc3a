"""development aid: F(4x4,3x3) forward on structured inputs, error maps against a direct fp64 sum"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
np.set_printoptions(linewidth=250, precision=3, suppress=True)
from puzzlelib_amd import backend
bnd = backend.getBackend(0, initmode=2)
G, dnn = bnd.GPUArray, bnd.dnn
dnn.setWinogradTile(4)

def direct(x, w, pad):
	n, c, h, ww = x.shape
	k = w.shape[0]
	xp = np.zeros((n, c, h + 2 * pad, ww + 2 * pad)); xp[:, :, pad:pad + h, pad:pad + ww] = x
	P, Q = h + 2 * pad - 2, ww + 2 * pad - 2
	y = np.zeros((n, k, P, Q))
	for r in range(3):
		for s in range(3):
			y += np.einsum("nchw,kc->nkhw", xp[:, :, r:r + P, s:s + Q], w[:, :, r, s].astype(np.float64))
	return y

rng = np.random.RandomState(0)
def case(name, x, w, pad=1, show=True):
	y = dnn.convNd(G.toGpu(x), G.toGpu(w), None, 1, pad, 1, 1, 3).get()
	ref = direct(x, w, pad)
	err = np.abs(y - ref)
	print("==", name, "max err %.3e" % err.max(), "of", np.abs(ref).max())
	if show and err.max() > 1e-3:
		n, k = np.unravel_index(err.reshape(err.shape[0], err.shape[1], -1).max(axis=2).argmax(), err.shape[:2])
		print("worst (n,k) =", n, k); print("got"); print(y[n, k]); print("ref"); print(ref[n, k])
		bad = (err > 1e-3)
		print("bad fraction %.3f; bad per k:" % bad.mean(), bad.mean(axis=(0, 2, 3))[:40])

n, c, k, h, w_ = 1, 4, 4, 8, 8
x = np.ones((n, c, h, w_), np.float32); wt = np.zeros((k, c, 3, 3), np.float32); wt[np.arange(k), np.arange(c), 1, 1] = 1
case("ones, identity 8x8", x, wt)
x = rng.randn(n, c, h, w_).astype(np.float32)
case("randn, identity 8x8", x, wt)
wt = rng.randn(k, c, 3, 3).astype(np.float32)
case("randn, randn 8x8 c4", x, wt)
x = rng.randn(1, 8, 8, 8).astype(np.float32); wt = rng.randn(4, 8, 3, 3).astype(np.float32)
case("randn 8x8 c8 (2 chunks)", x, wt)
x = rng.randn(1, 16, 8, 8).astype(np.float32); wt = rng.randn(40, 16, 3, 3).astype(np.float32)
case("randn 8x8 c16 k40", x, wt)
x = rng.randn(1, 4, 6, 6).astype(np.float32); wt = rng.randn(4, 4, 3, 3).astype(np.float32)
case("randn 6x6 (ragged)", x, wt)
x = rng.randn(3, 4, 16, 16).astype(np.float32); wt = rng.randn(4, 4, 3, 3).astype(np.float32)
case("randn 3x16x16 (48 tiles)", x, wt)
