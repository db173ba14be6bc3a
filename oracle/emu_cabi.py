"""
TEST INFRASTRUCTURE (build container only; never imported by puzzlelib_amd/).

A host-memory emulation of the C ABI of libpuzzle_mi355.so (include/puzzle_mi355.h): every device-facing `pz_*` entry is
executed on HOST buffers with the numpy oracle (oracle/cpu_ref.py), following the contract the header states for it.
"Device" addresses are addresses of numpy buffers, so views, offsets and arenas work as they do on the device.

Purpose (SURVEY f3's finish line, Unittester.py:114-122): the REFERENCE's own module / optimizer / trainer
unit tests can then run, with values, through this repository's Python glue — backend.py, dnn.py, modules.py, kernels.py,
lazy.py, fusion.py, the ≈3 000 lines that turn the reference's calls into C-ABI calls and decide every fusion — in a
container without a GPU, and the reference's own asserts judge the results (oracle/make_reftests.py). The package is put in
its dry-run mode (PUZZLE_MI355_DRYRUN=1: the real library answers host-side queries — shapes, workspace sizes, kernel-family
resolution — and nothing touches a device) and `lib.callHook` hands every other entry to `dispatch` below.

The emulation states each entry's CONTRACT, not the kernels' internals: buffers whose layout is private to the library
(convolution strip sums, BatchNorm partial sums, sign masks, arg-max bytes, prepared filter operands) get a layout of this
file's own that fits the sizes the real library reports for them; producer and consumer are both emulated here.
"""
import ctypes

import numpy as np

import cpu_ref as R

F = np.float32


class Emu:
	def __init__(self):
		self.blocks = {}          # address -> numpy uint8 buffer (kept alive; a released block stays readable, as on the device)
		self.rngs = {}
		self.calls = {}
		import reftape
		self.newState = reftape.LoggedState          # (a recording tape puts its own factory here: Tape.deviceState)

	# ------------------------------------------------------------------ memory
	def alloc(self, nbytes):
		buf = np.empty(max(int(nbytes), 1) + 64, dtype=np.uint8)
		buf[:] = 0xA5                                        # poison: reading unwritten memory shows
		addr = (buf.ctypes.data + 63) // 64 * 64
		self.blocks[addr] = buf
		return addr

	@staticmethod
	def view(ptr, count, dtype=F):
		if count == 0:
			return np.empty(0, dtype=dtype)
		assert ptr, "null pointer dereferenced by the emulated library"
		ptr = ptr if isinstance(ptr, int) else ptr.value
		raw = (ctypes.c_char * (int(count) * np.dtype(dtype).itemsize)).from_address(ptr)
		return np.frombuffer(raw, dtype=dtype)

	def f(self, ptr, *shape):
		return self.view(ptr, int(np.prod(shape)), F).reshape(shape)

	def opt(self, ptr, *shape):
		return None if not ptr else self.f(ptr, *shape)


EMU = Emu()
V, Fv = EMU.view, EMU.f


def out(ref, value):
	ref._obj.value = value


def desc(ref):
	return ref._obj


# ---------------------------------------------------------------------------------------------------- memory / runtime
def pz_malloc(ref, nbytes):
	out(ref, EMU.alloc(nbytes))


def pz_pool_alloc(pool, nbytes, ref):
	out(ref, EMU.alloc(nbytes))


def pz_host_alloc_pinned(ref, nbytes):
	out(ref, EMU.alloc(nbytes))


def pz_memcpy(dst, src, nbytes, stream):
	V(dst, nbytes, np.uint8)[:] = V(src, nbytes, np.uint8)


pz_memcpy_h2d = pz_memcpy_d2h = pz_memcpy_d2d = pz_memcpy


def pz_memcpy_2d(dst, dpitch, src, spitch, width, height, stream):
	for r in range(height):
		V(dst + r * dpitch, width, np.uint8)[:] = V(src + r * spitch, width, np.uint8)


def pz_memset_d32(dst, value, count, stream):
	V(dst, count, np.uint32)[:] = np.uint32(value)


def pz_strided_copy(dst, dstrides, src, sstrides, shape, ndim, stream):
	shape = tuple(int(shape[i]) for i in range(ndim))
	ds = tuple(int(dstrides[i]) * 4 for i in range(ndim))
	ss = tuple(int(sstrides[i]) * 4 for i in range(ndim))
	if int(np.prod(shape)) == 0:
		return
	span = lambda st: sum((n - 1) * s for n, s in zip(shape, st)) // 4 + 1
	d = np.lib.stride_tricks.as_strided(V(dst, span(ds), np.uint32), shape, ds)
	s = np.lib.stride_tricks.as_strided(V(src, span(ss), np.uint32), shape, ss)
	d[...] = s


def pz_cast_i32_f32(o, i, count, stream):
	V(o, count, F)[:] = V(i, count, np.int32).astype(F)


def pz_cast_f32_i32(o, i, count, stream):
	V(o, count, np.int32)[:] = V(i, count, F).astype(np.int32)


def pz_cast_f32_f16(o, i, count, stream):
	V(o, count, np.float16)[:] = V(i, count, F).astype(np.float16)


def pz_cast_f16_f32(o, i, count, stream):
	V(o, count, F)[:] = V(i, count, np.float16).astype(F)


# ---------------------------------------------------------------------------------------------------- convolution
def conv_geom(d):
	kw = dict(stride=(d.stride_h, d.stride_w), pad=(d.pad_h, d.pad_w), dilation=(d.dil_h, d.dil_w), groups=d.groups)
	p, q = R.conv_outshape((d.h, d.w), (d.r, d.s), kw["stride"], kw["pad"], kw["dilation"])
	return kw, (d.n, d.c, d.h, d.w), (d.k, d.c // d.groups, d.r, d.s), (d.n, d.k, p, q)


def filterOf(w, packed, wshape):
	"""the filter tensor of a pass: `w`, or the prepared operand (this emulation's own layout: the filter values themselves)"""
	return Fv(packed if packed else w, *wshape)


def write_strips(stats, strips, y):
	"""this emulation's strip layout: channel k, strip s -> {shift, sum(v - shift), sum((v - shift)^2), count} over the s-th of
	`strips` equal chunks of the flattened (n, p, q) axis (the real kernels: 64-pixel strips / 32-tile blocks)"""
	n, k, p, q = y.shape
	flat = y.transpose(1, 0, 2, 3).reshape(k, -1).astype(np.float64)
	npix = flat.shape[1]
	chunk = -(-npix // strips)
	st = Fv(stats, k, strips, 4)
	st[...] = 0
	for s in range(strips):
		seg = flat[:, s * chunk:(s + 1) * chunk]
		if seg.shape[1] == 0:
			continue
		shift = seg[:, :1]
		st[:, s, 0] = shift[:, 0]
		st[:, s, 1] = (seg - shift).sum(axis=1)
		st[:, s, 2] = ((seg - shift) ** 2).sum(axis=1)
		st[:, s, 3] = seg.shape[1]


def read_strips(stats, strips, c):
	"""-> (mean, biased variance, count) per channel, merged in fp64"""
	st = Fv(stats, c, strips, 4).astype(np.float64)
	cnt = st[:, :, 3]
	total = cnt.sum(axis=1)
	sums = (st[:, :, 0] * cnt + st[:, :, 1]).sum(axis=1)
	mean = sums / total
	# sum (v - mean)^2 = sum (v - shift)^2 - 2 (mean - shift) sum(v - shift) + cnt (mean - shift)^2
	dm = mean[:, None] - st[:, :, 0]
	m2 = (st[:, :, 2] - 2 * dm * st[:, :, 1] + cnt * dm * dm).sum(axis=1)
	return mean, m2 / total, total


def conv_fwd_impl(d, x, w, packed, bias, y, stats, strips_of, relu=False):
	d = desc(d)
	kw, xs, ws, ys = conv_geom(d)
	res = R.conv2d_fwd(Fv(x, *xs), filterOf(w, packed, ws), EMU.opt(bias, d.k), **kw)
	if relu:
		res = R.relu(res)
	Fv(y, *ys)[...] = res
	if stats:
		write_strips(stats, strips_of(d), res)


def strips_for(algo):
	def get(d):
		from puzzlelib_amd import lib
		n = ctypes.c_int(0)
		lib.pz_conv2d_fwd_stats_strips(ctypes.byref(d), algo, ctypes.byref(n))
		assert n.value > 0, "strip sums requested from a configuration that cannot produce them"
		return n.value
	return get


def pz_conv2d_fwd(d, x, w, bias, y, algo, ws, wsb, stream):
	conv_fwd_impl(d, x, w, None, bias, y, None, None)


def pz_conv2d_fwd_relu(d, x, w, packed, bias, y, algo, ws, wsb, stream):
	conv_fwd_impl(d, x, w, packed, bias, y, None, None, relu=True)


def pz_conv2d_fwd_stats(d, x, w, bias, y, stats, algo, ws, wsb, stream):
	conv_fwd_impl(d, x, w, None, bias, y, stats, strips_for(algo))


def pz_conv2d_fwd_pre(d, x, packed, bias, y, stats, algo, ws, wsb, stream):
	conv_fwd_impl(d, x, None, packed, bias, y, stats, strips_for(algo))


def xbn_input(d, x, xcoef, xrelu):
	"""relu?(a * x + b) per input channel: the tensor pz_bn_apply_add would have written"""
	d = desc(d)
	X = affine(Fv(x, d.n, d.c, d.h * d.w), xcoef, d.c)
	return np.ascontiguousarray(R.relu(X) if xrelu else X).reshape(d.n, d.c, d.h, d.w)


def pz_conv2d_fwd_xbn(d, x, xcoef, xrelu, w, packed, bias, y, stats, algo, ws, wsb, stream):
	X = xbn_input(d, x, xcoef, xrelu)
	conv_fwd_impl(d, X.ctypes.data, w, packed, bias, y, stats, strips_for(algo))


def pz_conv2d_bwd_filter_xbn(d, x, xcoef, xrelu, dy, bnx, bncoef, dw, alpha, beta, algo, ws, wsb, stream):
	X = xbn_input(d, x, xcoef, xrelu)
	bwd_filter_impl(d, X.ctypes.data, dy, dw, None, alpha, beta, bn=(bnx, bncoef) if bnx else None)


def pz_conv2d_prepack(jobs, njobs, stream):
	for i in range(njobs):
		job = jobs[i]
		d = job.desc
		count = d.k * (d.c // d.groups) * d.r * d.s
		V(job.packed, count, F)[:] = V(job.w, count, F)


def bwd_data_impl(d, dy, w, packed, dx, gate=None, bn=None):
	d = desc(d)
	kw, xs, ws, ys = conv_geom(d)
	g = Fv(dy, *ys)
	if bn is not None:
		bnx, coef = bn
		co = Fv(coef, d.k, 4)
		g = co[None, :, 0, None, None] * g + (co[None, :, 1, None, None] * Fv(bnx, *ys) + co[None, :, 2, None, None])
	res = R.conv2d_bwd_data(g.astype(F), filterOf(w, packed, ws), xs, **kw)
	if gate:
		res = R.relu_der(res, Fv(gate, *xs))
	Fv(dx, *xs)[...] = res


def pz_conv2d_bwd_data(d, dy, w, dx, algo, ws, wsb, stream):
	bwd_data_impl(d, dy, w, None, dx)


def pz_conv2d_bwd_data_pre(d, dy, packed, dx, algo, ws, wsb, stream):
	bwd_data_impl(d, dy, None, packed, dx)


def pz_conv2d_bwd_data_gate(d, dy, w, gate, dx, algo, ws, wsb, stream):
	bwd_data_impl(d, dy, w, None, dx, gate=gate)


def pz_conv2d_bwd_data_bn(d, dy, bnx, bncoef, w, dx, algo, ws, wsb, stream):
	bwd_data_impl(d, dy, w, None, dx, bn=(bnx, bncoef))


def pz_conv2d_bwd_data_bnstats(d, dy, bnx, bncoef, w, dx, gx, gab, gmean, partials, algo, ws, wsb, stream):
	"""dx as pz_conv2d_bwd_data[_bn] stores it, and in `partials` the sums of the gated gradient q = dx * (gab.x * gx + gab.y > 0)
	for the BatchNorm whose input gx is (the emulation's own partials layout: write_partials)"""
	bwd_data_impl(d, dy, w, None, dx, bn=(bnx, bncoef) if bnx else None)
	dd = desc(d)
	n, c, hw = dd.n, dd.c, dd.h * dd.w
	X = Fv(gx, n, c, hw)
	q = (Fv(dx, n, c, hw) * (affine(X, gab, c) > 0)).astype(F)
	write_partials(partials, q, X, gmean, c)


def bwd_filter_impl(d, x, dy, dw, db, alpha, beta, bn=None):
	d = desc(d)
	kw, xs, ws, ys = conv_geom(d)
	g = Fv(dy, *ys)
	if bn is not None:
		bnx, coef = bn
		co = Fv(coef, d.k, 4)
		g = (co[None, :, 0, None, None] * g + (co[None, :, 1, None, None] * Fv(bnx, *ys) + co[None, :, 2, None, None])).astype(F)
	res = R.conv2d_bwd_filter(Fv(x, *xs), g, ws, withbias=bool(db), **kw)
	gw, gb = res if db else (res, None)
	W = Fv(dw, *ws)
	W[...] = F(beta) * W + F(alpha) * gw if beta != 0.0 else F(alpha) * gw
	if db:
		B = Fv(db, d.k)
		B[...] = F(beta) * B + F(alpha) * gb if beta != 0.0 else F(alpha) * gb


def pz_conv2d_bwd_filter(d, x, dy, dw, db, alpha, beta, algo, ws, wsb, stream):
	bwd_filter_impl(d, x, dy, dw, db, alpha, beta)


def pz_conv2d_bwd_filter_bn(d, x, dy, bnx, bncoef, dw, alpha, beta, algo, ws, wsb, stream):
	bwd_filter_impl(d, x, dy, dw, None, alpha, beta, bn=(bnx, bncoef))


# ---------------------------------------------------------------------------------------------------- GEMM / reductions
def gemm_impl(ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc):
	def mat(ptr, rows, cols, ld):
		return np.lib.stride_tricks.as_strided(V(ptr, (rows - 1) * ld + cols, F), (rows, cols), (ld * 4, 4))
	A = mat(a, k, m, lda).T if ta else mat(a, m, k, lda)
	B = mat(b, n, k, ldb).T if tb else mat(b, k, n, ldb)
	C = mat(c, m, n, ldc)
	prod = F(alpha) * np.dot(A.astype(F), B.astype(F))
	C[...] = prod + F(beta) * C if beta != 0.0 else prod


def pz_gemm(ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, stream):
	gemm_impl(ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc)


def pz_gemm_ws(ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, ws, wsb, stream):
	gemm_impl(ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc)


def acc_out(o, value, alpha, beta):
	o[...] = F(alpha) * value + F(beta) * o if beta != 0.0 else F(alpha) * value


def pz_reduce_sum_rows(t, rows, cols, o, alpha, beta, stream):
	acc_out(V(o, rows, F), Fv(t, rows, cols).sum(axis=1, dtype=F), alpha, beta)


def pz_reduce_sum_cols(t, z, h, w, o, alpha, beta, stream):
	acc_out(Fv(o, z, w), Fv(t, z, h, w).sum(axis=1, dtype=F), alpha, beta)


def pz_argmax_rows(t, rows, cols, o, stream):
	V(o, rows, np.int32)[:] = np.argmax(Fv(t, rows, cols), axis=1)


def pz_argmax_cols(t, z, h, w, o, stream):
	V(o, z * w, np.int32).reshape(z, w)[...] = np.argmax(Fv(t, z, h, w), axis=1)


def pz_argmin_rows(t, rows, cols, o, stream):
	V(o, rows, np.int32)[:] = np.argmin(Fv(t, rows, cols), axis=1)


def pz_argmin_cols(t, z, h, w, o, stream):
	V(o, z * w, np.int32).reshape(z, w)[...] = np.argmin(Fv(t, z, h, w), axis=1)


def pz_bias_add(o, mat, vec, z, n, m, veclen, axis, stream):
	M, vv = Fv(mat, z, n, m), Fv(vec, z, veclen)
	if axis == 1:
		res = M + vv[:, None, np.arange(m) % veclen]
	else:
		res = M + vv[:, np.arange(n) % veclen, None]
	Fv(o, z, n, m)[...] = res


def pz_count_neq_i32(x, y, count, o, stream):
	V(o, 1, F)[0] = R.count_neq(V(x, count, np.int32), V(y, count, np.int32))


def pz_cost_accuracy(kind, x, labels, count, o, stream):
	xv, lab = V(x, count, F), V(labels, count, np.int32)
	wrong = np.where(lab == 1, xv <= 0, xv > 0) if kind == 0 else ((xv <= 1).astype(np.int32) != lab)
	V(o, 1, F)[0] = F(wrong.sum())


def pz_kl_divergence(x, y, grad, gradnorm, count, o, stream):
	xv, yv = V(x, count, F), V(y, count, F)
	V(grad, count, F)[:] = (yv - xv) * F(gradnorm)
	pos = yv > 0
	V(o, 1, F)[0] = F((yv[pos].astype(np.float64) * (np.log(yv[pos].astype(np.float64)) - np.log(xv[pos].astype(np.float64)))).sum())


def pz_reduce_minmax_f32(x, count, is_max, o, stream):
	v = V(x, count, F)
	V(o, 1, F)[0] = v.max() if is_max else v.min()


def pz_reduce_minmax_i32(x, count, is_max, o, stream):
	v = V(x, count, np.int32)
	V(o, 1, np.int32)[0] = v.max() if is_max else v.min()


def pz_dot(x, y, count, o, stream):
	V(o, 1, F)[0] = R.dot(V(x, count, F), V(y, count, F))


def pz_asum(x, count, o, stream):
	V(o, 1, F)[0] = R.l1norm(V(x, count, F))


def pz_matvec(mat, vec, o, z, h, w, axis, alpha, beta, stream):
	M = Fv(mat, z, h, w)
	if axis == 1:
		res = np.einsum("zhw,zw->zh", M, Fv(vec, z, w))
		acc_out(Fv(o, z, h), res.astype(F), alpha, beta)
	else:
		res = np.einsum("zhw,zh->zw", M, Fv(vec, z, h))
		acc_out(Fv(o, z, w), res.astype(F), alpha, beta)


# ---------------------------------------------------------------------------------------------------- batch normalisation
def bn_stats_of(x):
	xa = x.astype(np.float64)
	m = xa.shape[0] * xa.shape[2]
	mean = xa.sum(axis=(0, 2)) / m
	var = ((xa - mean[None, :, None]) ** 2).sum(axis=(0, 2)) / m
	return mean, var, m


def bn_finalize(mean, var, m, c, scale, bias, run_mean, run_var, save_mean, save_invvar, epsilon, factor):
	invstd = 1.0 / np.sqrt(var + np.float64(F(epsilon)))
	rm, rv = V(run_mean, c, F), V(run_var, c, F)
	f = np.float64(F(factor))
	rm[:] = ((1 - f) * rm + f * mean).astype(F)
	rv[:] = ((1 - f) * rv + f * var * m / max(m - 1, 1)).astype(F)
	V(save_mean, c, F)[:] = mean
	V(save_invvar, c, F)[:] = invstd
	a = V(scale, c, F) * invstd.astype(F)
	b = V(bias, c, F) - mean.astype(F) * a
	return a.astype(F), b.astype(F)


def pz_bn_fwd_train_coef(x, n, c, hw, scale, bias, run_mean, run_var, save_mean, save_invvar, epsilon, factor, stats, strips, coef,
						 ws, wsb, stream):
	if stats:
		mean, var, m = read_strips(stats, strips, c)
		assert int(m[0]) == n * hw, "strip sums of %d values handed to a BatchNorm over %d" % (int(m[0]), n * hw)
		m = n * hw
	else:
		mean, var, m = bn_stats_of(Fv(x, n, c, hw))
	a, b = bn_finalize(mean, var, m, c, scale, bias, run_mean, run_var, save_mean, save_invvar, epsilon, factor)
	co = Fv(coef, c, 2)
	co[:, 0], co[:, 1] = a, b


def affine(x, coef, c):
	co = Fv(coef, c, 2)
	return (co[None, :, 0, None] * x + co[None, :, 1, None]).astype(F)


def pz_bn_fwd_infer(x, y, n, c, hw, scale, bias, mean, var, epsilon, stream):
	Fv(y, n, c, hw)[...] = R.bn_fwd_infer(Fv(x, n, c, hw), V(scale, c, F), V(bias, c, F), V(mean, c, F), V(var, c, F), epsilon)


def apply_add(x1, coef1, x2, coef2, n, c, hw, relu):
	res = affine(Fv(x1, n, c, hw), coef1, c)
	if x2:
		other = Fv(x2, n, c, hw)
		res = res + (affine(other, coef2, c) if coef2 else other)
	return R.relu(res) if relu else res.astype(F)


def pz_bn_apply_add(x1, coef1, x2, coef2, o, n, c, hw, relu, stream):
	Fv(o, n, c, hw)[...] = apply_add(x1, coef1, x2, coef2, n, c, hw, relu)


def mask_shape(n, c, hw):
	return n * c, -(-hw // 4)


def pz_bn_apply_add_mask(x1, coef1, x2, coef2, o, mask, n, c, hw, relu, stream):
	res = apply_add(x1, coef1, x2, coef2, n, c, hw, relu)
	Fv(o, n, c, hw)[...] = res
	planes, nb = mask_shape(n, c, hw)
	bits = np.zeros((planes, nb * 4), dtype=np.uint8)
	bits[:, :hw] = (res.reshape(planes, hw) > 0)
	V(mask, planes * nb, np.uint8).reshape(planes, nb)[...] = (bits.reshape(planes, nb, 4) << np.arange(4, dtype=np.uint8)).sum(axis=2)


def gate_of(y, mask, n, c, hw):
	if mask:
		planes, nb = mask_shape(n, c, hw)
		by = V(mask, planes * nb, np.uint8).reshape(planes, nb, 1)
		return ((by >> np.arange(4, dtype=np.uint8)) & 1).reshape(planes, nb * 4)[:, :hw].reshape(n, c, hw).astype(bool)
	return Fv(y, n, c, hw) > 0


def write_partials(part, g, x, mean, c):
	"""this emulation's layout of BatchNorm-backward partial sums: part[2k] = sum g, part[2k + 1] = sum g (x - mean[k])"""
	p = Fv(part, c, 2)
	g64 = g.astype(np.float64)
	p[:, 0] = g64.sum(axis=(0, 2))
	p[:, 1] = (g64 * (x.astype(np.float64) - V(mean, c, F).astype(np.float64)[None, :, None])).sum(axis=(0, 2))


def gate_stats_impl(g, y, mask, gout, n, c, hw, xa, mean_a, part_a, xb, mean_b, part_b):
	res = (g * gate_of(y, mask, n, c, hw)).astype(F)
	Fv(gout, n, c, hw)[...] = res
	if xa:
		write_partials(part_a, res, Fv(xa, n, c, hw), mean_a, c)
	if xb:
		write_partials(part_b, res, Fv(xb, n, c, hw), mean_b, c)


def pz_bn_gate_stats(g0, g1, y, mask, gout, n, c, hw, xa, mean_a, part_a, xb, mean_b, part_b, stream):
	g = Fv(g0, n, c, hw) + Fv(g1, n, c, hw) if g1 else Fv(g0, n, c, hw)
	gate_stats_impl(g, y, mask, gout, n, c, hw, xa, mean_a, part_a, xb, mean_b, part_b)


def pz_bn_gate_stats_up2(g0c, g1c, y, mask, gout, n, c, h, w, xa, mean_a, part_a, xb, mean_b, part_b, stream):
	hc, wc = (h + 1) // 2, (w + 1) // 2
	g = np.zeros((n, c, h, w), dtype=F)
	g[:, :, ::2, ::2] = Fv(g0c, n, c, hc, wc) + Fv(g1c, n, c, hc, wc)
	gate_stats_impl(g.reshape(n, c, h * w), y, mask, gout, n, c, h * w, xa, mean_a, part_a, xb, mean_b, part_b)


def pz_bn_bwd_stats(x, dy, n, c, hw, save_mean, partials, stream):
	write_partials(partials, Fv(dy, n, c, hw), Fv(x, n, c, hw), save_mean, c)


def param_grads(c, hw_n, scale, save_mean, save_invvar, dscale, dbias, dscale_acc, dbias_acc, alpha, beta, partials):
	p = Fv(partials, c, 2).astype(np.float64)
	inv = V(save_invvar, c, F).astype(np.float64)
	s1, s2 = p[:, 0], p[:, 1]
	ds, db = (s2 * inv).astype(F), s1.astype(F)
	if dscale:
		V(dscale, c, F)[:] = ds
	if dbias:
		V(dbias, c, F)[:] = db
	if dscale_acc:
		acc_out(V(dscale_acc, c, F), ds, alpha, beta)
	if dbias_acc:
		acc_out(V(dbias_acc, c, F), db, alpha, beta)
	sc, mu = V(scale, c, F).astype(np.float64), V(save_mean, c, F).astype(np.float64)
	A = sc * inv
	B = -sc * inv ** 3 * s2 / hw_n
	C = -A * s1 / hw_n - B * mu
	return A, B, C


def pz_bn_bwd_coef(n, c, hw, scale, save_mean, save_invvar, dscale, dbias, dscale_acc, dbias_acc, alpha, beta, partials, coef, stream):
	A, B, C = param_grads(c, n * hw, scale, save_mean, save_invvar, dscale, dbias, dscale_acc, dbias_acc, alpha, beta, partials)
	co = Fv(coef, c, 4)
	co[:, 0], co[:, 1], co[:, 2], co[:, 3] = A, B, C, 0


def pz_bn_bwd_apply_coef(x, dy, dx, n, c, hw, coef, stream):
	co = Fv(coef, c, 4)
	Fv(dx, n, c, hw)[...] = co[None, :, 0, None] * Fv(dy, n, c, hw) + (co[None, :, 1, None] * Fv(x, n, c, hw) + co[None, :, 2, None])


def pz_bn_bwd_from_partials(x, dy, dx, n, c, hw, scale, save_mean, save_invvar, dscale, dbias, dscale_acc, dbias_acc, alpha, beta,
							partials, stream):
	A, B, C = param_grads(c, n * hw, scale, save_mean, save_invvar, dscale, dbias, dscale_acc, dbias_acc, alpha, beta, partials)
	Fv(dx, n, c, hw)[...] = (A[None, :, None] * Fv(dy, n, c, hw) + (B[None, :, None] * Fv(x, n, c, hw) + C[None, :, None])).astype(F)


def bn_bwd_impl(x, g, dx, n, c, hw, scale, save_mean, save_invvar, dscale, dbias, dscale_acc=None, dbias_acc=None, alpha=1.0, beta=0.0):
	X = Fv(x, n, c, hw)
	tmp = np.empty((c, 2), dtype=F)
	write_partials(tmp.ctypes.data, g, X, save_mean, c)
	A, B, C = param_grads(c, n * hw, scale, save_mean, save_invvar, dscale, dbias, dscale_acc, dbias_acc, alpha, beta, tmp.ctypes.data)
	Fv(dx, n, c, hw)[...] = (A[None, :, None] * g + (B[None, :, None] * X + C[None, :, None])).astype(F)


def pz_bn_bwd(x, dy, dx, n, c, hw, scale, save_mean, save_invvar, dscale, dbias, ws, wsb, stream):
	bn_bwd_impl(x, Fv(dy, n, c, hw), dx, n, c, hw, scale, save_mean, save_invvar, dscale, dbias)


def pz_bn_bwd_gate(x, dy, dx, n, c, hw, scale, save_mean, save_invvar, dscale, dbias, gate_coef, ws, wsb, stream):
	g = (Fv(dy, n, c, hw) * (affine(Fv(x, n, c, hw), gate_coef, c) > 0)).astype(F)
	bn_bwd_impl(x, g, dx, n, c, hw, scale, save_mean, save_invvar, dscale, dbias)


def pz_bn_bwd_gate_from_partials(x, dy, dx, n, c, hw, scale, save_mean, save_invvar, dscale, dbias, gate_coef, partials, stream):
	X = Fv(x, n, c, hw)
	g = (Fv(dy, n, c, hw) * (affine(X, gate_coef, c) > 0)).astype(F)
	A, B, C = param_grads(c, n * hw, scale, save_mean, save_invvar, dscale, dbias, None, None, 1.0, 0.0, partials)
	Fv(dx, n, c, hw)[...] = (A[None, :, None] * g + (B[None, :, None] * X + C[None, :, None])).astype(F)


def pz_bn_bwd_acc(x, dy, dx, n, c, hw, scale, bias, save_mean, save_invvar, dscale, dbias, act, dscale_acc, dbias_acc, alpha, beta,
				  ws, wsb, stream):
	g = Fv(dy, n, c, hw)
	if act:
		a = V(scale, c, F) * V(save_invvar, c, F)
		b = V(bias, c, F) - V(save_mean, c, F) * a
		g = (g * ((a[None, :, None] * Fv(x, n, c, hw) + b[None, :, None]) > 0)).astype(F)
	bn_bwd_impl(x, g, dx, n, c, hw, scale, save_mean, save_invvar, dscale, dbias, dscale_acc, dbias_acc, alpha, beta)


def pz_bn_fwd_train(x, y, n, c, hw, scale, bias, run_mean, run_var, save_mean, save_invvar, epsilon, factor, ws, wsb, stream):
	mean, var, m = bn_stats_of(Fv(x, n, c, hw))
	a, b = bn_finalize(mean, var, m, c, scale, bias, run_mean, run_var, save_mean, save_invvar, epsilon, factor)
	Fv(y, n, c, hw)[...] = a[None, :, None] * Fv(x, n, c, hw) + b[None, :, None]


# ---------------------------------------------------------------------------------------------------- pooling
def pool_geom(d):
	p = (d.h + 2 * d.pad_h - d.size_h) // d.stride_h + 1
	q = (d.w + 2 * d.pad_w - d.size_w) // d.stride_w + 1
	return (d.n, d.c, d.h, d.w), (d.n, d.c, p, q), dict(size=(d.size_h, d.size_w), stride=(d.stride_h, d.stride_w), pad=(d.pad_h, d.pad_w), mode=d.mode)


def pool_fwd_impl(d, X, y, index_ws):
	xs, ys, kw = pool_geom(d)
	Fv(y, *ys)[...] = R.pool2d_fwd(X, **kw)
	if index_ws and d.mode == 0:
		win, _ = R._pool_windows(X, kw["size"], kw["stride"], kw["pad"], -np.inf)
		V(index_ws, int(np.prod(ys)), np.uint8)[:] = np.argmax(win.reshape(ys + (-1, )), axis=4).ravel()


def pz_pool2d_fwd(d, x, y, index_ws, stream):
	d = desc(d)
	pool_fwd_impl(d, Fv(x, d.n, d.c, d.h, d.w), y, index_ws)


def pz_pool2d_fwd_bn(d, x, coef, relu, y, index_ws, stream):
	d = desc(d)
	X = affine(Fv(x, d.n, d.c, d.h * d.w), coef, d.c)
	pool_fwd_impl(d, (R.relu(X) if relu else X).reshape(d.n, d.c, d.h, d.w), y, index_ws)


def pz_pool2d_bwd(d, dy, x, y, index_ws, dx, stream):
	d = desc(d)
	xs, ys, kw = pool_geom(d)
	DY = Fv(dy, *ys)
	if d.mode == 0 and index_ws:
		n, c, p, q = ys
		idx = V(index_ws, int(np.prod(ys)), np.uint8).reshape(ys).astype(np.int64)
		r, s = idx // d.size_w, idx % d.size_w
		dxp = np.zeros((n, c, d.h + 2 * d.pad_h, d.w + 2 * d.pad_w), dtype=F)
		nn, cc, pp, qq = np.meshgrid(np.arange(n), np.arange(c), np.arange(p), np.arange(q), indexing="ij")
		np.add.at(dxp, (nn, cc, pp * d.stride_h + r, qq * d.stride_w + s), DY)
		Fv(dx, *xs)[...] = dxp[:, :, d.pad_h:d.pad_h + d.h, d.pad_w:d.pad_w + d.w]
	else:
		Fv(dx, *xs)[...] = R.pool2d_bwd(DY, Fv(x, *xs) if x else None, None, **kw)


# ---------------------------------------------------------------------------------------------------- softmax / costs
def pz_softmax_fwd(x, y, n, c, spatial, stream):
	Fv(y, n, c, spatial)[...] = R.softmax_fwd(Fv(x, n, c, spatial))


def pz_softmax_bwd(dy, y, dx, n, c, spatial, stream):
	Fv(dx, n, c, spatial)[...] = R.softmax_bwd(Fv(dy, n, c, spatial), Fv(y, n, c, spatial))


def pz_cross_entropy(scores, labels, weights, n, c, spatial, grad, error, ws, wsb, stream):
	err, g = R.cross_entropy(Fv(scores, n, c, spatial), V(labels, n * spatial, np.int32).reshape(n, spatial), EMU.opt(weights, c))
	Fv(grad, n, c, spatial)[...] = g
	V(error, 1, F)[0] = err


# ---------------------------------------------------------------------------------------------------- element-wise family
def elt_ops():
	from puzzlelib_amd import lib as L

	def inplace(fn):
		return lambda a, s: fn(*a, *s)

	def store(fn):
		def run(a, s):
			a[0][...] = fn(*a[1:], *s)
		return run

	def adam(a, s):
		R.adam(a[0], a[1] * F(s[4]), a[2], a[3], *s[:4])

	def mom(fn):
		return lambda a, s: fn(a[0], a[1] * F(s[2]), a[2], s[0], s[1])

	def dropout(a, s):
		a[0][...] = R.dropout(a[1], a[2].view(np.uint32), np.float32(s[0]).view(np.uint32), s[1])

	def dropout2d(a, s):
		a[0][...] = R.dropout2d(a[1], a[2].view(np.uint32), np.float32(s[0]).view(np.uint32), s[1], int(np.float32(s[2]).view(np.int32)))

	def iadd(a, s):
		a[0][...] = a[0] + a[1]

	def imul(a, s):
		a[0][...] = a[0] * a[1]

	return {
		L.OP_SIGMOID: store(R.sigmoid), L.OP_SIGMOID_DER: store(R.sigmoid_der), L.OP_TANH: store(R.tanh), L.OP_TANH_DER: store(R.tanh_der),
		L.OP_RELU: store(R.relu), L.OP_RELU_DER: store(R.relu_der), L.OP_LEAKY_RELU: store(R.leaky_relu),
		L.OP_LEAKY_RELU_DER: store(R.leaky_relu_der), L.OP_ELU: store(R.elu), L.OP_ELU_DER: store(R.elu_der),
		L.OP_SOFTPLUS: store(R.softplus), L.OP_SOFTPLUS_DER: store(R.softplus_der), L.OP_CLIP: store(R.clip), L.OP_CLIP_DER: store(R.clip_der),
		L.OP_GELU: store(R.gelu), L.OP_GELU_DER: store(R.gelu_der), L.OP_DROPOUT: dropout, L.OP_DROPOUT2D: dropout2d,
		L.OP_AXPY: inplace(R.axpy), L.OP_ADD: lambda a, s: R.add_scaled(a[1], s[0], a[2], s[1], out=a[0]), L.OP_MUL: store(R.mul),
		L.OP_LINEAR: store(R.linear), L.OP_ABS: store(lambda x: np.abs(x)), L.OP_WEIGHT_DECAY: inplace(R.weight_decay),
		L.OP_ADAM: adam, L.OP_CLASSIC_MOM_SGD: mom(R.classic_mom_sgd), L.OP_NESTEROV_MOM_SGD: mom(R.nesterov_mom_sgd),
		L.OP_RMSPROP: inplace(R.rmsprop), L.OP_ADAGRAD: inplace(R.adagrad), L.OP_ADADELTA: inplace(R.adadelta),
		L.OP_RMSPROP_GRAVES: inplace(R.rmsprop_graves), L.OP_SMORMS3: inplace(R.smorms3),
		L.OP_ADD3: store(lambda a, b: (a + b).astype(F)), L.OP_IADD: iadd, L.OP_IMUL: imul,
		L.OP_ADD3_RELU: store(lambda a, b: R.relu((a + b).astype(F))), L.OP_ADD3_GATE: store(lambda a, b, y: R.relu_der((a + b).astype(F), y)),
		L.OP_L1_PENALTY: store(lambda g, d, a: (g - F(a) * ((F(0) <= d).astype(F) - (d < F(0)).astype(F))).astype(F)),
		L.OP_L1_GRAD: store(lambda pred, target, norm: np.where(pred - target > F(0), -F(norm), F(norm)).astype(F)),      # l1gradKer, Cuda/Kernels/ElementWise.py:1135-1140
		L.OP_RBM: store(lambda x, uni: (uni < F(1) / (F(1) + np.exp(-x))).astype(F)),
	}


ELT = None


def pz_eltwise(op, count, ptrs, nptrs, scalars, nscalars, start, stop, step, stream):
	global ELT
	if ELT is None:
		ELT = elt_ops()
	sc = [F(scalars[i]) for i in range(nscalars)]
	# strided variant: elements start, start + step, ... < stop of every operand (Cuda/SourceModule.py:216-226)
	arrays = [V(ptrs[i], count, F)[start:stop:step] for i in range(nptrs)]
	if op not in ELT:
		raise NotImplementedError("emulated pz_eltwise: op %d" % op)
	ELT[op](arrays, sc)


def pz_multi_add(njobs, outs, xs, ys, alphas, betas, sizes, stream):
	for j in range(njobs):
		n = int(sizes[j])
		R.add_scaled(V(xs[j], n, F), alphas[j], V(ys[j], n, F), betas[j], out=V(outs[j], n, F))


# ---------------------------------------------------------------------------------------------------- RNG (statistical parity only)
def pz_rng_create(seed, ref):
	handle = EMU.alloc(8)
	EMU.rngs[handle] = EMU.newState(seed)
	out(ref, handle)


# what a fill draws is written down in tests/reftape.py (DEVICE_DRAWS): the tapes regenerate these values on the GPU box
def pz_rng_fill_u32(rng, o, count, stream):
	V(o, count, np.uint32)[:] = EMU.rngs[rng].draw("u32", count)


def pz_rng_fill_uniform(rng, o, count, stream):
	V(o, count, F)[:] = EMU.rngs[rng].draw("uniform", count)


def pz_rng_fill_normal(rng, o, count, mean, stddev, stream):
	V(o, count, F)[:] = EMU.rngs[rng].draw("normal", count, float(mean), float(stddev))


# ---------------------------------------------------------------------------------------------------- beside the hot path (f3)
def pz_maskpool2d_fwd(d, x, y, mask, stream):
	d = desc(d)
	xs, ys, kw = pool_geom(d)
	res, m = R.maskpool2d_fwd(Fv(x, *xs), kw["size"], kw["stride"], kw["pad"])
	Fv(y, *ys)[...] = res
	V(mask, int(np.prod(ys)), np.int32).reshape(ys)[...] = m


def pz_maskpool2d_bwd(d, dy, mask, dx, stream):
	d = desc(d)
	xs, ys, kw = pool_geom(d)
	Fv(dx, *xs)[...] = R.maskpool2d_bwd(Fv(dy, *ys), V(mask, int(np.prod(ys)), np.int32).reshape(ys), xs)


def pz_maxunpool2d_fwd(x, mask, y, planes, in_plane, out_plane, stream):
	m = V(mask, planes * in_plane, np.int32).reshape(1, planes, in_plane)
	Fv(y, planes * out_plane)[...] = R.maxunpool2d_fwd(Fv(x, 1, planes, in_plane), m, (1, planes, 1, out_plane)).ravel()


def pz_maxunpool2d_bwd(dy, mask, dx, planes, in_plane, out_plane, stream):
	m = V(mask, planes * in_plane, np.int32).reshape(1, planes, in_plane)
	Fv(dx, planes * in_plane)[...] = R.maxunpool2d_bwd(Fv(dy, 1, planes, out_plane), m).ravel()


def pz_lrn_fwd(x, y, scale, n, c, h, w, size, alpha, beta, k, cross, stream):
	X = Fv(x, n, c, h, w)
	Fv(y, n, c, h, w)[...] = R.lrn_fwd(X, size, alpha, beta, k, bool(cross))
	if scale:
		Fv(scale, n, c, h, w)[...] = R.lrn_norms(X, size, alpha, k, bool(cross))


def pz_lrn_bwd(x, dy, scale, dx, n, c, h, w, size, alpha, beta, k, cross, stream):
	Fv(dx, n, c, h, w)[...] = R.lrn_bwd(Fv(x, n, c, h, w), Fv(dy, n, c, h, w), size, alpha, beta, k, bool(cross))


def pz_svm_cost(scores, labels, samples, cases, spatial, squared, grad, terms, stream):
	S = Fv(scores, samples, cases, spatial)
	lab = V(labels, samples * spatial, np.int32).reshape(samples, spatial)
	err, g = R.svm_cost(S, lab, "l2" if squared else "l1")
	Fv(grad, samples, cases, spatial)[...] = g
	t = Fv(terms, samples * cases * spatial)
	t[...] = 0
	t[0] = err                                           # the caller sums the terms (pz_asum)


def pz_cost_pointwise(kind, a, b, labels, error, grad, grad2, terms, total, numsamples, numcases, norm, fullnorm, stream):
	A = V(a, total, F)
	if kind == 0:
		err, g = R.bce_cost(A, V(labels, total, np.int32), numsamples, numcases)
	elif kind == 1:
		err, g = R.hinge_cost(A, V(labels, total, np.int32), numsamples, numcases)
	elif kind == 2:
		err, g = R.smooth_l1_cost(A, V(b, total, F), norm, fullnorm)
	else:
		err, g, g2 = R.l1_hinge_cost(A.reshape(numsamples, numcases), V(b, total, F).reshape(numsamples, numcases),
									 V(labels, numsamples, np.int32), numsamples, numcases)
		V(grad2, total, F)[:] = g2.ravel()
	V(grad, total, F)[:] = g.ravel()
	V(error, 1, F)[0] += err


def pz_prelu_fwd(x, slopes, y, n, maps, mapsize, shared, stream):
	Fv(y, n, maps, mapsize)[...] = R.prelu_fwd(Fv(x, n, maps, mapsize), V(slopes, 1 if shared else maps, F), bool(shared))


def pz_prelu_bwd_data(dy, slopes, x, dx, n, maps, mapsize, shared, stream):
	Fv(dx, n, maps, mapsize)[...] = R.prelu_bwd_data(Fv(dy, n, maps, mapsize), V(slopes, 1 if shared else maps, F), Fv(x, n, maps, mapsize), bool(shared))


def pz_prelu_bwd_params(x, dy, per_map, n, maps, mapsize, stream):
	V(per_map, maps, F)[:] = R.prelu_bwd_params(Fv(x, n, maps, mapsize), Fv(dy, n, maps, mapsize), False)


def pz_reflectpad2d_fwd(x, y, planes, inh, inw, upad, bpad, lpad, rpad, stream):
	Fv(y, 1, planes, inh + upad + bpad, inw + lpad + rpad)[...] = R.reflectpad_fwd(Fv(x, 1, planes, inh, inw), (upad, bpad, lpad, rpad))


def pz_reflectpad2d_bwd(dy, dx, planes, inh, inw, upad, bpad, lpad, rpad, stream):
	Fv(dx, 1, planes, inh, inw)[...] = R.reflectpad_bwd(Fv(dy, 1, planes, inh + upad + bpad, inw + lpad + rpad), (upad, bpad, lpad, rpad))


def pz_upsample_fwd(x, y, planes, ind, inh, inw, sd, sh, sw, linear, stream):
	Fv(y, 1, planes, ind * sd, inh * sh, inw * sw)[...] = R.upsample_fwd(Fv(x, 1, planes, ind, inh, inw), (sd, sh, sw), "linear" if linear else "nearest")


def pz_upsample_bwd(dy, dx, planes, ind, inh, inw, sd, sh, sw, linear, stream):
	Fv(dx, 1, planes, ind, inh, inw)[...] = R.upsample_bwd(Fv(dy, 1, planes, ind * sd, inh * sh, inw * sw), (sd, sh, sw), "linear" if linear else "nearest")


def pz_embed_fwd(words, vocab, o, tokens, embsize, stream):
	w = V(words, tokens, np.int32)
	rows = int(w.max()) + 1 if tokens else 0
	Fv(o, tokens, embsize)[...] = R.embed_fwd(w, Fv(vocab, max(rows, 1), embsize))


def pz_embed_bwd_params(words, grad, vocab, scale, tokens, embsize, stream):
	w = V(words, tokens, np.int32)
	rows = int(w.max()) + 1 if tokens else 0
	R.embed_bwd_params(w, Fv(grad, tokens, embsize), Fv(vocab, max(rows, 1), embsize), scale)


def pz_ctc_loss(probs, datalen, labels, offsets, order, seg_start, seg_label, seg_off, T, batch, vocab, blank, max_positions,
				alphas, nll, grad, error, stream):
	"""contract of the header: probs are softmax OUTPUTS; writes the forward variables, nll per sample, the gradient inside
	datalen (the caller zero-filled the rest) and ADDS the summed nll to *error. The sort tables are the kernel's business."""
	off = V(offsets, batch + 1, np.int32)
	lengths = np.diff(off)
	lab = V(labels, int(off[-1]), np.int32)
	dl = V(datalen, batch, np.int32)
	total, g, al = R.ctc_loss(Fv(probs, T, batch, vocab), dl, lab, lengths, blank, normalized=True)
	Fv(alphas, al.size)[...] = al
	# per-sample nll: the oracle returns their sum; each sample alone gives its own
	per = Fv(nll, batch)
	for b in range(batch):
		one, _, _ = R.ctc_loss(Fv(probs, T, batch, vocab)[:, b:b + 1], dl[b:b + 1], lab[off[b]:off[b + 1]], lengths[b:b + 1], blank, normalized=True)
		per[b] = one
	Fv(grad, T, batch, vocab)[...] = g
	Fv(error, 1)[0] += F(total)


# ---------------------------------------------------------------------------------------------------- run-time compiled kernels
# pz_rtc_compile itself is NOT emulated: the real library compiles the generated HIP source with hiprtc (no device needed), so a
# template that does not compile fails here as it would on the device. What cannot run here is the code object: for kernels
# generated by puzzlelib_amd/rtc.py's templates, the recipe they carry in their first line ("// pz-rtc: {...}") is turned into
# the same loops in plain C, compiled with gcc and called on the host buffers. A hand-written SourceModule has no recipe:
# NotImplementedError.
class HostKernels:
	def __init__(self):
		self.sources, self.modules, self.functions = {}, {}, {}

	def remember(self, real):
		"""wraps lib.pz_rtc_compile: compile for real, note which source the code object came from"""
		def compile_(source, name, options, noptions, code, size, log, logbytes):
			real(source, name, options, noptions, code, size, log, logbytes)
			self.sources[code._obj.value] = source.decode()
		return compile_

	def load(self, code):
		import json, os, subprocess, tempfile
		source = self.sources.get(code if isinstance(code, int) else code.value)
		assert source is not None, "pz_module_load: a code object pz_rtc_compile did not produce"
		first = source.split("\n", 1)[0]
		if not first.startswith("// pz-rtc: "):
			raise NotImplementedError("the C-ABI emulation cannot run a hand-written SourceModule (no generation recipe)")
		r = json.loads(first[len("// pz-rtc: "):])
		params = ", ".join("%s %s" % (c, n) for c, n in r["args"])
		if r["kind"] == "eltwise":
			text = ("#include <math.h>\n#include <stdint.h>\n"
					"void %(name)s(%(params)s, long long size) { for (long long i = 0; i < size; ++i) { %(op)s; } }\n"
					"void %(name)s_strided(%(params)s, long long start, long long stop, long long step) "
					"{ for (long long i = start; i < stop; i += step) { %(op)s; } }\n") % dict(name=r["name"], params=params, op=r["operation"])
			entries = {r["name"]: [c for c, _ in r["args"]] + ["long long"], r["name"] + "_strided": [c for c, _ in r["args"]] + ["long long"] * 3}
		else:
			# (the device's own order: every thread of a workgroup strides over the elements, then a halving tree — bit-equal sums)
			text = ("#include <math.h>\n#include <stdint.h>\ntypedef %(T)s pz_acc_t;\n"
					"static pz_acc_t red(pz_acc_t a, pz_acc_t b) { return (%(reduceExpr)s); }\n"
					"static pz_acc_t tree_(pz_acc_t *t) { for (int h = %(block)d / 2; h > 0; h >>= 1) for (int k = 0; k < h; ++k) t[k] = red(t[k], t[k + h]); return t[0]; }\n"
					"void %(name)s_stage1(%(params)s, pz_acc_t *partials, long long size, int blocks) {\n"
					"	for (int blk = 0; blk < blocks; ++blk) { pz_acc_t t[%(block)d];\n"
					"		for (int th = 0; th < %(block)d; ++th) { pz_acc_t acc = %(neutral)s;\n"
					"			for (long long i = (long long)blk * %(block)d + th; i < size; i += (long long)blocks * %(block)d) acc = red(acc, (pz_acc_t)(%(mapExpr)s));\n"
					"			t[th] = acc; }\n"
					"		partials[blk] = tree_(t); } }\n"
					"void %(name)s_stage2(const pz_acc_t *partials, pz_acc_t *out, int count) {\n"
					"	pz_acc_t t[%(block)d];\n"
					"	for (int th = 0; th < %(block)d; ++th) { pz_acc_t acc = %(neutral)s; for (int i = th; i < count; i += %(block)d) acc = red(acc, partials[i]); t[th] = acc; }\n"
					"	out[0] = tree_(t); }\n") % dict(r, params=params)
			entries = {r["name"] + "_stage1": [c for c, _ in r["args"]] + ["pz_acc_t *", "long long", "+blocks"],
					   r["name"] + "_stage2": ["pz_acc_t *", "pz_acc_t *", "int"]}
		scratch = tempfile.mkdtemp(prefix="emu_rtc_")
		path = os.path.join(scratch, "k.c")
		open(path, "w").write(text)
		subprocess.run(["gcc", "-O1", "-shared", "-fPIC", "-std=gnu99", "-o", path[:-2] + ".so", path, "-lm"], check=True, capture_output=True)
		handle = EMU.alloc(8)
		self.modules[handle] = (ctypes.CDLL(path[:-2] + ".so"), entries)
		return handle

	def function(self, module, name):
		dll, entries = self.modules[module if isinstance(module, int) else module.value]
		name = name.decode()
		assert name in entries, "the module has no kernel %s" % name
		handle = EMU.alloc(8)
		self.functions[handle] = (getattr(dll, name), entries[name])
		return handle

	CT = {"float": ctypes.c_float, "double": ctypes.c_double, "int": ctypes.c_int, "unsigned int": ctypes.c_uint, "long long": ctypes.c_longlong,
		  "unsigned long long": ctypes.c_ulonglong, "short": ctypes.c_short, "unsigned short": ctypes.c_ushort, "signed char": ctypes.c_byte,
		  "unsigned char": ctypes.c_ubyte}

	def launch(self, function, grid, args, nbytes):
		fn, types = self.functions[function if isinstance(function, int) else function.value]
		raw = bytes(args[:nbytes]) if isinstance(args, (bytes, bytearray)) else ctypes.string_at(args, nbytes)
		values, offset = [], 0
		for cname in types:
			if cname == "+blocks":
				values.append(ctypes.c_int(int(grid[0])))
				continue
			base = " ".join(w for w in cname.split() if w not in ("const", "volatile", "__restrict__", "restrict"))
			ct = ctypes.c_void_p if base.endswith("*") else self.CT[base]
			size = ctypes.sizeof(ct)
			offset += -offset % size
			values.append(ct.from_buffer_copy(raw[offset:offset + size]))
			offset += size
		fn.restype = None
		fn(*values)


HOST = HostKernels()


def pz_module_load(code, ref):
	out(ref, HOST.load(code))


def pz_module_unload(module):
	pass


def pz_module_function(module, name, ref):
	out(ref, HOST.function(module, name))


def pz_function_launch(function, grid, block, shared, args, nbytes, stream):
	HOST.launch(function, grid, args, nbytes)


# ---------------------------------------------------------------------------------------------------- dispatch
NOOPS = {
	"pz_init", "pz_device_sync", "pz_free", "pz_pool_destroy", "pz_pool_release", "pz_pool_free_held", "pz_host_free_pinned",
	"pz_stream_destroy", "pz_stream_sync", "pz_stream_wait_event", "pz_event_destroy", "pz_event_record", "pz_event_sync",
	"pz_rng_destroy", "pz_conv_profile_enable",
}
# answered by lib.py's own dry-run stand-ins (handles, device properties, event queries)
DRY = {
	"pz_pool_create", "pz_stream_create", "pz_stream_create_priority", "pz_event_create", "pz_device_count", "pz_device_num_cus",
	"pz_device_name", "pz_device_arch", "pz_device_mem_info", "pz_pool_stats", "pz_event_elapsed_ms", "pz_event_query",
}


def dispatch(name, args):
	"""lib.callHook: True = executed here; False = leave it to the dry-run stand-in"""
	if name in DRY:
		return False
	EMU.calls[name] = EMU.calls.get(name, 0) + 1
	if name in NOOPS:
		return True
	fn = globals().get(name)
	if fn is None:
		raise NotImplementedError("the C-ABI emulation has no %s" % name)
	fn(*args)
	return True


def install():
	"""puts the package's ctypes layer (already imported in dry-run mode) onto this emulation"""
	from puzzlelib_amd import lib
	assert lib.DRYRUN, "import puzzlelib_amd with PUZZLE_MI355_DRYRUN=1 before installing the emulation"
	lib.callHook = dispatch
	lib.pz_rtc_compile = HOST.remember(lib.pz_rtc_compile)
	return EMU
