"""Does an HBM-bound pass overlap with an MFMA-bound filter-gradient kernel when the two are launched as a PAIR on two
streams (nothing else in flight)? H = pz_bn_apply_add over a block-output tensor (read 2, write 1), W = backward-filter of
a pointwise layer (registers / LDS leave room for H's waves: 2 x 138 of 512 registers per SIMD). Prints H alone, W alone,
H then W on one stream, and H || W. Kernel-tuning aid only."""
import ctypes, os, sys
from ctypes import byref, c_size_t, c_float
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from puzzlelib_amd import backend, lib, driver

bnd = backend.getBackend(0, initmode=2)
G = bnd.GPUArray
rng = np.random.RandomState(0)
side = driver.Stream()

def ev():
	h = ctypes.c_void_p(); lib.pz_event_create(byref(h)); return h

def run(label, hshape, conv, reps=20):
	n, c, hw = hshape
	x1 = G.toGpu(rng.randn(n, c, hw).astype(np.float32)); x2 = G.toGpu(rng.randn(n, c, hw).astype(np.float32))
	out = G.empty((n, c, hw), dtype=np.float32)
	coef = G.toGpu(np.ones((c, 2), np.float32))
	(ci, h, w), (k, size, stride, pad) = conv
	d = lib.ConvDesc(256, ci, h, w, k, size, size, stride, stride, pad, pad, 1, 1, 1)
	p = (h + 2 * pad - size) // stride + 1
	cx = G.toGpu(rng.randn(256, ci, h, w).astype(np.float32)); dy = G.toGpu(rng.randn(256, k, p, p).astype(np.float32))
	dw = G.zeros((k, ci, size, size), dtype=np.float32)
	need = c_size_t(0); lib.pz_conv2d_workspace_bytes(byref(d), lib.CONV_BWD_FILTER, lib.CONV_ALGO_AUTO, byref(need))
	ws = G.empty((need.value // 4 + 64, ), dtype=np.float32)

	H = lambda st: lib.pz_bn_apply_add(x1.ptr, coef.ptr, x2.ptr, None, out.ptr, n, c, hw, 1, st)
	W = lambda st: lib.pz_conv2d_bwd_filter(byref(d), cx.ptr, dy.ptr, dw.ptr, None, 1.0, 0.0, lib.CONV_ALGO_AUTO, ws.ptr, need.value, st)
	e0, e1, ej, es = ev(), ev(), ev(), ev()

	def timed(body):
		body(); lib.pz_device_sync()
		lib.pz_event_record(e0, None)
		for _ in range(reps):
			body()
		lib.pz_event_record(e1, None); lib.pz_event_sync(e1)
		ms = c_float(0); lib.pz_event_elapsed_ms(e0, e1, byref(ms)); return ms.value / reps * 1e3

	def pair():
		lib.pz_event_record(es, None); lib.pz_stream_wait_event(side.handle, es)       # side starts where main is
		W(side.handle); H(None)
		lib.pz_event_record(ej, side.handle); lib.pz_stream_wait_event(None, ej)       # main joins the side stream

	th, tw = timed(lambda: H(None)), timed(lambda: W(None))
	ts, tp = timed(lambda: (H(None), W(None))), timed(pair)
	print("%-46s H %6.1f us  W %6.1f us  H;W %6.1f us  H||W %6.1f us  (hidden: %4.0f%% of the shorter)" % (
		label, th, tw, ts, tp, 100.0 * (ts - tp) / min(th, tw)))

run("stage-2 block output + wgrad 256->64 (55^2)", (256, 256, 3025), ((256, 55, 55), (64, 1, 1, 0)))
run("stage-2 block output + wgrad 64->256 (55^2)", (256, 256, 3025), ((64, 55, 55), (256, 1, 1, 0)))
run("stage-3 block output + wgrad 128->512 (28^2)", (256, 512, 784), ((128, 28, 28), (512, 1, 1, 0)))
run("stage-3 block output + wgrad 512->128 (28^2)", (256, 512, 784), ((512, 28, 28), (128, 1, 1, 0)))
run("stage-4 block output + wgrad 1024->256 (14^2)", (256, 1024, 196), ((1024, 14, 14), (256, 1, 1, 0)))
run("stage-3 mid tensor + wino wgrad 128 (28^2)", (256, 128, 784), ((128, 28, 28), (128, 3, 1, 1)))
run("stage-2 block output + igemm-sized wgrad 3x3 64", (256, 256, 3025), ((64, 55, 55), (64, 3, 1, 1)))


def run_mm(label, conv, reps=20):
	"""backward-data || backward-filter of one layer (both MFMA-bound): the pair the backward pass actually overlaps"""
	(ci, h, w), (k, size, stride, pad) = conv
	d = lib.ConvDesc(256, ci, h, w, k, size, size, stride, stride, pad, pad, 1, 1, 1)
	p = (h + 2 * pad - size) // stride + 1
	cx = G.toGpu(rng.randn(256, ci, h, w).astype(np.float32)); dy = G.toGpu(rng.randn(256, k, p, p).astype(np.float32))
	wt = G.toGpu(rng.randn(k, ci, size, size).astype(np.float32)); dx = G.empty((256, ci, h, w), dtype=np.float32)
	dw = G.zeros((k, ci, size, size), dtype=np.float32)
	nf, nd = c_size_t(0), c_size_t(0)
	lib.pz_conv2d_workspace_bytes(byref(d), lib.CONV_BWD_FILTER, lib.CONV_ALGO_AUTO, byref(nf))
	lib.pz_conv2d_workspace_bytes(byref(d), lib.CONV_BWD_DATA, lib.CONV_ALGO_AUTO, byref(nd))
	wf, wd = G.empty((nf.value // 4 + 64, ), dtype=np.float32), G.empty((nd.value // 4 + 64, ), dtype=np.float32)
	D = lambda st: lib.pz_conv2d_bwd_data(byref(d), dy.ptr, wt.ptr, dx.ptr, lib.CONV_ALGO_AUTO, wd.ptr, nd.value, st)
	W = lambda st: lib.pz_conv2d_bwd_filter(byref(d), cx.ptr, dy.ptr, dw.ptr, None, 1.0, 0.0, lib.CONV_ALGO_AUTO, wf.ptr, nf.value, st)
	e0, e1, ej, es = ev(), ev(), ev(), ev()

	def timed(body):
		body(); lib.pz_device_sync()
		lib.pz_event_record(e0, None)
		for _ in range(reps):
			body()
		lib.pz_event_record(e1, None); lib.pz_event_sync(e1)
		ms = c_float(0); lib.pz_event_elapsed_ms(e0, e1, byref(ms)); return ms.value / reps * 1e3

	def pair():
		lib.pz_event_record(es, None); lib.pz_stream_wait_event(side.handle, es)
		W(side.handle); D(None)
		lib.pz_event_record(ej, side.handle); lib.pz_stream_wait_event(None, ej)

	td, tw = timed(lambda: D(None)), timed(lambda: W(None))
	ts, tp = timed(lambda: (D(None), W(None))), timed(pair)
	print("%-46s D %6.1f us  W %6.1f us  D;W %6.1f us  D||W %6.1f us  (hidden: %4.0f%% of the shorter)" % (
		label, td, tw, ts, tp, 100.0 * (ts - tp) / min(td, tw)))


run_mm("dgrad || wgrad 256->64 (55^2)", ((256, 55, 55), (64, 1, 1, 0)))
run_mm("dgrad || wgrad 128->512 (28^2)", ((128, 28, 28), (512, 1, 1, 0)))
run_mm("dgrad || wgrad 1024->256 (14^2)", ((1024, 14, 14), (256, 1, 1, 0)))
run_mm("dgrad || wgrad 512->2048 (7^2)", ((512, 7, 7), (2048, 1, 1, 0)))
run_mm("dgrad || wgrad 3x3 128 (28^2, Winograd)", ((128, 28, 28), (128, 3, 1, 1)))
run_mm("dgrad || wgrad 3x3 512 (7^2, Winograd)", ((512, 7, 7), (512, 3, 1, 1)))
