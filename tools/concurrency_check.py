"""
Repro of an unexplained cross-kernel interaction on MI355X: bn_gate_stats_kernel<TWO> (main
stream; its -O3 code is full of SLP-packed v_pk_{add,mul,fma}_f32) returns wrong partial sums for a few workgroups per
launch while wgrad_split_kernel (bf16 MFMA) runs on a second stream. Clean on an idle chip, with the fp32-MFMA backward-
filter kernel as the neighbour (PUZZLE_MI355_MATH=f32), with the MFMAs of the split kernel compiled out or replaced by fp32
MFMAs (tools/ablate.sh: wg_nomfma, wg_f32mfma), and with bn.hip built -fno-slp-vectorize / -O1.

    PUZZLE_MI355_MATH=split6 python tools/concurrency_check.py
"""
import os, sys, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from puzzlelib_amd import backend, lib, lazy, driver
from ctypes import byref, c_size_t
lazy.disabled.add("sidestream")
bnd = backend.getBackend(0, initmode=2)
G = bnd.GPUArray
rng = np.random.RandomState(0)
dnn = bnd.dnn
side = driver.Stream()
def wgrad_job(n, c, h, w, k):
	x = G.toGpu(rng.randn(n, c, h, w).astype(np.float32)); y = G.toGpu(rng.randn(n, k, h, w).astype(np.float32))
	W = G.zeros((k, c, 1, 1), dtype=np.float32)
	desc = dnn.convDesc(x.shape, W.shape, 1, 0, 1, 1)
	P, Q, wsbytes, _, _ = dnn.convGeometry(desc, lib.CONV_BWD_FILTER, 0)
	ws = G.empty((max(wsbytes, 1), ), dtype=np.uint8)
	return lambda: lib.pz_conv2d_bwd_filter(byref(desc), x.ptr, y.ptr, W.ptr, None, 1.0, 0.0, 0, ws.ptr, wsbytes, side.handle)
n, c, h, w = 64, 1024, 14, 14
hw = h * w
host = [rng.randn(n, c, h, w).astype(np.float32) for _ in range(5)]
g0, g1, y, xa, xb = [G.toGpu(a) for a in host]
ma, mb = G.toGpu(rng.randn(c).astype(np.float32)), G.toGpu(rng.randn(c).astype(np.float32))
sc, iv = G.toGpu(rng.rand(c).astype(np.float32) + 0.5), G.toGpu(rng.rand(c).astype(np.float32) + 0.5)
size = c_size_t(0); lib.pz_bn_workspace_bytes(n, c, hw, byref(size))
pa, pb = G.empty((size.value // 4, ), dtype=np.float32), G.empty((size.value // 4, ), dtype=np.float32)
gout, dxa, dxb = [G.empty(g0.shape, dtype=np.float32) for _ in range(3)]
ds, db, ds2, db2 = [G.empty((c, ), dtype=np.float32) for _ in range(4)]
def gs(): lib.pz_bn_gate_stats(g0.ptr, g1.ptr, y.ptr, None, gout.ptr, n, c, hw, xa.ptr, ma.ptr, pa.ptr, xb.ptr, mb.ptr, pb.ptr, None)
def A(d1=ds, d2=db): lib.pz_bn_bwd_from_partials(xa.ptr, gout.ptr, dxa.ptr, n, c, hw, sc.ptr, ma.ptr, iv.ptr, d1.ptr, d2.ptr, None, None, 1.0, 0.0, pa.ptr, None)
def Bk(d1=ds, d2=db): lib.pz_bn_bwd_from_partials(xb.ptr, gout.ptr, dxb.ptr, n, c, hw, sc.ptr, mb.ptr, iv.ptr, d1.ptr, d2.ptr, None, None, 1.0, 0.0, pb.ptr, None)
variants = {
	"as is": lambda: (gs(), A(), Bk()),
	"sync after stats": lambda: (gs(), lib.pz_device_sync(), A(), Bk()),
	"B first": lambda: (gs(), Bk(), A()),
	"only A": lambda: (gs(), A()),
	"separate dscale/dbias": lambda: (gs(), A(), Bk(ds2, db2)),
}
s = wgrad_job(64, 1024, 14, 14, 256)
gs(); A(); Bk(); lib.pz_device_sync()
ref = [a.get().copy() for a in (gout, dxa, dxb, pa)]
for name, fn in variants.items():
	bad = [0, 0, 0, 0]
	for it in range(10):
		for _ in range(3): s()
		fn()
		for _ in range(2): s()
		lib.pz_device_sync()
		for i, (a, r) in enumerate(zip((gout, dxa, dxb, pa), ref)):
			got = a.get()
			bad[i] += not np.array_equal(got, r)
			if i == 3 and it == 0 and not np.array_equal(got, r):
				d = np.flatnonzero(got != r)
				print("   pa differs at", d.size, "of", got.size, "floats; first idx", d[:6], "vals", got[d[:4]], "ref", r[d[:4]], " offset floats of partials:", 4 * c)
	print("%-24s wrong (gout, dxa, dxb, pa): %s" % (name, bad))
lib.pz_device_sync()
gs(); A(); Bk(); lib.pz_device_sync()
print("serial again wrong:", [int(not np.array_equal(a.get(), r)) for a, r in zip((gout, dxa, dxb, pa), ref)])
for name, a, hst in zip(("g0", "g1", "y", "xa", "xb"), (g0, g1, y, xa, xb), host):
	d = a.get() != hst
	print(name, "modified elements:", int(d.sum()), "first at", np.flatnonzero(d.ravel())[:3])
