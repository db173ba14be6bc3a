"""
TEST INFRASTRUCTURE — CPU oracle for the PuzzleLib operator hot path (numpy restatement).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module. Nothing under `puzzlelib_amd/` imports it; the product path has no CPU fallback.

Every function restates the algorithm of a reference function and cites it
(paths relative to the reference repo root, puzzlelib/PuzzleLib v1.0.2):

* forward conv / pool / batch-norm inference / GEMM / column-sum follow the reference's numpy CPU
  backend (CPU/Wrappers/NumpyDnn.py, CPU/Wrappers/NumpyBlas.py) — im2col via as_strided + np.dot;
* element-wise and optimizer kernels follow CPU/Kernels/ElementWise.py (the gcc-JIT'd C loops);
* ops the reference CPU backend does not implement (conv backward, pool backward, BN training
  forward/backward, softmax, cross-entropy — see Backend/Dnn.py:341-371) follow the brute-force host
  formulas inside the reference's own unit tests (Cuda/Wrappers/CuDnn.py, CuDnnNorm.py,
  Cuda/Kernels/Costs.py, Modules/*.py), restated in vectorised form ("restatement-extended").

Parity pinning: `oracle/make_golden.py` (runs in the build container only) checks this file against
the imported reference CPU backend and against the reference's `bnd`-parameterised unit tests, and
writes the fixtures in tests/golden/. One convention is not pinned by any reference test: the
batch-norm *running variance* update (we use the cuDNN/MIOpen convention: unbiased batch variance).

Arithmetic is float32 by default (like the reference); pass ``acc=np.float64`` for a tighter
yardstick when checking long reductions.
"""
import numpy as np


# --------------------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------------------

def pair(v):
	"""repeatValue(val, 2) — CPU/Wrappers/NumpyDnn.py:15-23"""
	if isinstance(v, (int, np.integer)):
		return int(v), int(v)
	v = tuple(int(a) for a in v)
	assert len(v) == 2
	return v


def conv_outshape(inhw, size, stride, pad, dilation):
	"""outshape — CPU/Wrappers/NumpyDnn.py:26-36 (same formula as Modules/Conv2D.py:50-51)"""
	(inh, inw), (fh, fw), (sh, sw), (ph, pw), (dh, dw) = inhw, size, stride, pad, dilation
	outh = (inh + 2 * ph - dh * (fh - 1) - 1) // sh + 1
	outw = (inw + 2 * pw - dw * (fw - 1) - 1) // sw + 1
	return outh, outw


def im2col(data, size, stride, pad, dilation, padval=0):
	"""im2col — CPU/Wrappers/NumpyDnn.py:39-61: np.pad then as_strided to (N*P*Q, C*R*S)."""
	fh, fw = size
	sh, sw = stride
	ph, pw = pad
	dh, dw = dilation

	n, c, inh, inw = data.shape
	outh, outw = conv_outshape((inh, inw), size, stride, pad, dilation)

	if ph > 0 or pw > 0:
		data = np.pad(data, ((0, 0), (0, 0), (ph, ph), (pw, pw)), mode="constant", constant_values=padval)

	st = data.strides
	col = np.lib.stride_tricks.as_strided(
		data, shape=(n, outh, outw, c, fh, fw),
		strides=(st[0], sh * st[2], sw * st[3], st[1], st[2] * dh, st[3] * dw)
	)
	return col.reshape(n * outh * outw, c * fh * fw)


def col2im(data, maps, shape):
	"""col2im — CPU/Wrappers/NumpyDnn.py:64-71: (N*P*Q, K) -> NCHW."""
	h, w = shape
	return np.ascontiguousarray(np.moveaxis(data.reshape(-1, h, w, maps), 3, 1))


# --------------------------------------------------------------------------------------------------
# convolution (a1, a2, a3)
# --------------------------------------------------------------------------------------------------

def conv2d_fwd(x, w, bias=None, stride=1, pad=0, dilation=1, groups=1, acc=np.float32):
	"""
	y = x (*) w (+ b): NumpyDnn.conv2d — CPU/Wrappers/NumpyDnn.py:83-99 (groups=1);
	groups>1 as the host loop of convGroupTest — Cuda/Wrappers/CuDnn.py:137-162.
	bias may be shaped (K,) or (1,K,1,1).
	"""
	stride, pad, dilation = pair(stride), pair(pad), pair(dilation)
	n, c, inh, inw = x.shape
	k, cg, fh, fw = w.shape
	assert c == cg * groups and k % groups == 0

	outh, outw = conv_outshape((inh, inw), (fh, fw), stride, pad, dilation)
	kg = k // groups
	outs = []

	for g in range(groups):
		col = im2col(x[:, g * cg:(g + 1) * cg].astype(acc, copy=False), (fh, fw), stride, pad, dilation)
		wmat = w[g * kg:(g + 1) * kg].reshape(kg, -1).T.astype(acc, copy=False)
		outs.append(np.dot(col, wmat))

	out = outs[0] if groups == 1 else np.concatenate(outs, axis=1)
	if bias is not None:
		out = out + np.asarray(bias, dtype=acc).reshape(1, k)

	return col2im(out, k, (outh, outw)).astype(np.float32)


def conv2d_bwd_data(dy, w, xshape, stride=1, pad=0, dilation=1, groups=1, acc=np.float32):
	"""
	dx[n,c,p*s+r*d-pad, ...] += w[k,c,r,s] * dy[n,k,p,q]: host loops of conv2dTest —
	Cuda/Wrappers/CuDnn.py:51-65 and multiMapsWithPadsTest — Modules/Conv2D.py:215-230
	(stride/pad/dilation); vectorised as dy_col . W followed by a scatter-add over the R*S taps.
	"""
	stride, pad, dilation = pair(stride), pair(pad), pair(dilation)
	n, c, inh, inw = xshape
	k, cg, fh, fw = w.shape
	(sh, sw), (ph, pw), (dh, dw) = stride, pad, dilation
	_, _, outh, outw = dy.shape
	kg = k // groups

	dxp = np.zeros((n, c, inh + 2 * ph, inw + 2 * pw), dtype=acc)

	for g in range(groups):
		dyg = np.moveaxis(dy[:, g * kg:(g + 1) * kg].astype(acc, copy=False), 1, 3).reshape(-1, kg)
		wmat = w[g * kg:(g + 1) * kg].reshape(kg, -1).astype(acc, copy=False)

		dcol = np.dot(dyg, wmat).reshape(n, outh, outw, cg, fh, fw)

		for r in range(fh):
			for s in range(fw):
				tgt = dxp[:, g * cg:(g + 1) * cg, r * dh:r * dh + sh * outh:sh, s * dw:s * dw + sw * outw:sw]
				tgt += np.moveaxis(dcol[:, :, :, :, r, s], 3, 1)

	return np.ascontiguousarray(dxp[:, :, ph:ph + inh, pw:pw + inw]).astype(np.float32)


def conv2d_bwd_filter(x, dy, wshape, stride=1, pad=0, dilation=1, groups=1, withbias=False,
					  wgrad=None, bgrad=None, scale=1.0, momentum=0.0, acc=np.float32):
	"""
	dw[k,c,r,s] = sum x[n,c,p*s+r*d-pad,...] * dy[n,k,p,q]; db[k] = sum dy — host loops of conv2dTest
	Cuda/Wrappers/CuDnn.py:67-80, Modules/Conv2D.py:232-248.
	Accumulate contract (Hip/Wrappers/MIOpen.py:414-433,441-455): when wgrad is given and
	(scale != 1 or momentum != 0): wgrad <- momentum*wgrad + scale*dw (same for bgrad); otherwise
	wgrad <- dw. Returns wgrad or (wgrad, bgrad).
	"""
	stride, pad, dilation = pair(stride), pair(pad), pair(dilation)
	k, cg, fh, fw = wshape
	kg = k // groups
	dws = []

	for g in range(groups):
		col = im2col(x[:, g * cg:(g + 1) * cg].astype(acc, copy=False), (fh, fw), stride, pad, dilation)
		dyg = np.moveaxis(dy[:, g * kg:(g + 1) * kg].astype(acc, copy=False), 1, 3).reshape(-1, kg)
		dws.append(np.dot(dyg.T, col).reshape(kg, cg, fh, fw))

	dw = (dws[0] if groups == 1 else np.concatenate(dws, axis=0)).astype(np.float32)

	def accumulate(dst, val):
		if dst is not None and (scale != 1.0 or momentum != 0.0):
			dst[...] = (np.float32(momentum) * dst + np.float32(scale) * val.reshape(dst.shape)).astype(np.float32)
			return dst
		if dst is None:
			return val
		dst[...] = val.reshape(dst.shape)
		return dst

	wgrad = accumulate(wgrad, dw)
	if not withbias:
		return wgrad

	db = np.sum(dy, axis=(0, 2, 3), dtype=acc).astype(np.float32)
	bgrad = accumulate(bgrad, db)
	return wgrad, bgrad


# --------------------------------------------------------------------------------------------------
# GEMM / matrix-vector helpers (a4, a5)
# --------------------------------------------------------------------------------------------------

def gemm(A, B, out=None, transpA=False, transpB=False, alpha=1.0, beta=0.0, acc=np.float32):
	"""
	out = alpha*op(A)*op(B) + beta*out, row-major; NumpyBlas.mulMatrixOnMatrix —
	CPU/Wrappers/NumpyBlas.py:66-91 (alpha=1, beta=0 there); alpha/beta semantics of BlasContext.gemm —
	Cuda/Source/Libs/CuBlas.c:327-402 (not both transposed).
	"""
	assert not (transpA and transpB)
	a = A.T if transpA else A
	b = B.T if transpB else B
	res = np.dot(a.astype(acc, copy=False), b.astype(acc, copy=False))

	if alpha != 1.0:
		res = acc(alpha) * res
	if out is not None and beta != 0.0:
		res = res + acc(beta) * out
	res = res.astype(np.float32)

	if out is None:
		return res
	out[...] = res
	return out


def add_vec_to_mat(vec, mat, axis=1, out=None):
	"""
	MatModule.addVecToMat — Cuda/Kernels/MatVec.py:346-374: axis=1 adds vec along columns (bias add; vec may
	tile when mat width is a multiple of its length), axis=0 adds vec[row] to every row element.
	Batched: mat (z,n,m), vec (z,len).
	"""
	if axis == 1:
		reps = mat.shape[-1] // vec.shape[-1]
		v = np.tile(vec, reps) if vec.ndim == 1 else np.tile(vec, (1, reps))
		res = mat + (v[np.newaxis, :] if vec.ndim == 1 else v[:, np.newaxis, :])
	else:
		res = mat + (vec[:, np.newaxis] if vec.ndim == 1 else vec[:, :, np.newaxis])

	if out is None:
		return res.astype(np.float32)
	out[...] = res
	return out


def matsum(tensor, axis=0, out=None, alpha=1.0, beta=0.0, acc=np.float32):
	"""MatModule.matsum — Cuda/Kernels/MatVec.py:273-308; NumpyBlas.sumOnMatrix — CPU/Wrappers/NumpyBlas.py:7-22."""
	s = np.sum(tensor, axis=axis, dtype=acc)
	if out is None:
		return (acc(alpha) * s).astype(np.float32)
	out[...] = (acc(beta) * out + acc(alpha) * s).astype(np.float32)
	return out


def argmax(tensor, axis):
	"""MatModule.argmax — Cuda/Kernels/MatVec.py:231-266 (first maximum wins on ties, as np.argmax)."""
	return np.argmax(tensor, axis=axis).astype(np.int32)


def dot(x, y):
	"""NumpyBlas.dot — CPU/Wrappers/NumpyBlas.py:56-63"""
	return float(np.vdot(x.ravel(), y.ravel()))


def l1norm(x):
	"""NumpyBlas.vectorL1Norm — CPU/Wrappers/NumpyBlas.py:48-53"""
	return float(np.sum(np.abs(x.ravel())))


# --------------------------------------------------------------------------------------------------
# batch normalisation (a6, a7)
# --------------------------------------------------------------------------------------------------

def bn_fwd_infer(x, scale, bias, mean, var, epsilon=1e-5):
	"""NumpyDnn.batchNorm2d — CPU/Wrappers/NumpyDnn.py:117-129: y = scale/sqrt(var+eps)*(x-mean)+bias."""
	shp = (1, -1) + (1, ) * (x.ndim - 2)
	s = scale.reshape(shp) / np.sqrt(var.reshape(shp) + np.float32(epsilon))
	return (s * (x - mean.reshape(shp)) + bias.reshape(shp)).astype(np.float32)


def bn_fwd_train(x, scale, bias, mean, var, epsilon=1e-5, factor=1.0, acc=np.float32):
	"""
	Spatial BN training forward as pinned by batchNorm2dTest — Cuda/Wrappers/CuDnnNorm.py:23-52 and
	Modules/BatchNorm2D.py:34-59: biased variance for normalisation/saveinvvar; running stats updated
	IN PLACE on `mean`/`var` as (1-f)*old + f*new (cuDNN/MIOpen contract, Cuda/Source/Libs/CuDnnNorm.c).
	Running *variance* uses the unbiased batch variance (cuDNN convention; not pinned by a reference test).
	Returns (y, savemean, saveinvvar) with stats shaped (C,).
	"""
	axes = (0, ) + tuple(range(2, x.ndim))
	shp = (1, -1) + (1, ) * (x.ndim - 2)
	norm = x.size // x.shape[1]

	xa = x.astype(acc, copy=False)
	mu = np.sum(xa, axis=axes, dtype=acc) / norm
	v = np.sum((xa - mu.reshape(shp))**2, axis=axes, dtype=acc) / norm
	rstd = 1.0 / np.sqrt(v + acc(epsilon))

	y = (xa - mu.reshape(shp)) * rstd.reshape(shp) * scale.reshape(shp) + bias.reshape(shp)

	f = acc(factor)
	unbiased = v * norm / max(norm - 1, 1)
	mean[...] = ((1 - f) * mean.reshape(-1) + f * mu).astype(np.float32).reshape(mean.shape)
	var[...] = ((1 - f) * var.reshape(-1) + f * unbiased).astype(np.float32).reshape(var.shape)

	return y.astype(np.float32), mu.astype(np.float32), rstd.astype(np.float32)


def bn_bwd(dy, x, scale, savemean, saveinvvar, acc=np.float32):
	"""
	BN backward — formulas of batchNorm2dTest Cuda/Wrappers/CuDnnNorm.py:55-63 (== Modules/BatchNorm2D.py:61-82):
	dscale = sum dy*xhat; dbias = sum dy; dx = dy*scale*rstd + (2*dvar*(x-mu) + dmean)/norm.
	Returns (dx, dscale, dbias) with (C,) parameter grads.
	"""
	axes = (0, ) + tuple(range(2, x.ndim))
	shp = (1, -1) + (1, ) * (x.ndim - 2)
	norm = x.size // x.shape[1]

	xa, dya = x.astype(acc, copy=False), dy.astype(acc, copy=False)
	mu, rstd, sc = savemean.reshape(shp).astype(acc), saveinvvar.reshape(shp).astype(acc), scale.reshape(shp).astype(acc)

	xc = xa - mu
	dscale = np.sum(dya * xc * rstd, axis=axes, dtype=acc)
	dbias = np.sum(dya, axis=axes, dtype=acc)

	dmean = -rstd * dbias.reshape(shp) * sc
	dvar = -0.5 * np.sum(dya * xc, axis=axes, dtype=acc).reshape(shp) * sc * rstd**3
	dx = dya * sc * rstd + (2 * dvar * xc + dmean) / norm

	return dx.astype(np.float32), dscale.astype(np.float32), dbias.astype(np.float32)


# --------------------------------------------------------------------------------------------------
# pooling (a9)
# --------------------------------------------------------------------------------------------------

POOL_MAX, POOL_AVG_WITH_PAD, POOL_AVG_NO_PAD = 0, 1, 2


def _pool_windows(x, size, stride, pad, padval):
	n, c, inh, inw = x.shape
	(fh, fw), (sh, sw), (ph, pw) = size, stride, pad
	outh, outw = conv_outshape((inh, inw), size, stride, pad, (1, 1))

	xp = np.pad(x, ((0, 0), (0, 0), (ph, ph), (pw, pw)), mode="constant", constant_values=padval) \
		if (ph or pw) else x
	st = xp.strides
	win = np.lib.stride_tricks.as_strided(
		xp, shape=(n, c, outh, outw, fh, fw), strides=(st[0], st[1], sh * st[2], sw * st[3], st[2], st[3])
	)
	return win, (outh, outw)


def pool2d_fwd(x, size=2, stride=2, pad=0, mode=POOL_MAX):
	"""
	NumpyDnn.pool2d — CPU/Wrappers/NumpyDnn.py:102-114 (max: pads with -inf; avg only exact for pad=0 there).
	avgWithPad divides by the full window (zero padding counted) as Modules/AvgPool2D.py:40-50;
	avgNoPad divides by the number of in-bounds taps (MIOpen "miopenPoolingAverage").
	"""
	size, stride, pad = pair(size), pair(stride), pair(pad)

	if mode == POOL_MAX:
		win, _ = _pool_windows(x, size, stride, pad, -np.inf)
		return win.max(axis=(4, 5)).astype(np.float32)

	win, _ = _pool_windows(x, size, stride, pad, 0.0)
	s = win.sum(axis=(4, 5), dtype=np.float32)

	if mode == POOL_AVG_WITH_PAD:
		return (s / np.float32(size[0] * size[1])).astype(np.float32)

	ones, _ = _pool_windows(np.ones(x.shape[2:], dtype=np.float32)[None, None], size, stride, pad, 0.0)
	return (s / ones.sum(axis=(4, 5))).astype(np.float32)


def pool2d_bwd(dy, x, y, size=2, stride=2, pad=0, mode=POOL_MAX):
	"""
	Max: gradient goes to the arg-max tap of each window — maxpool2dTest Cuda/Wrappers/CuDnn.py:395-410.
	The reference host loop credits *every* tap equal to the max; device libraries (and this build) credit
	the FIRST maximum in row-major window order — identical on tie-free data, which is what tests use.
	Avg: Modules/AvgPool2D.py:52-63 (dy / window for avgWithPad; dy / valid-count for avgNoPad).
	"""
	size, stride, pad = pair(size), pair(stride), pair(pad)
	(fh, fw), (sh, sw), (ph, pw) = size, stride, pad
	n, c, inh, inw = x.shape
	_, _, outh, outw = dy.shape

	dxp = np.zeros((n, c, inh + 2 * ph, inw + 2 * pw), dtype=np.float32)

	if mode == POOL_MAX:
		win, _ = _pool_windows(x, size, stride, pad, -np.inf)
		flat = win.reshape(n, c, outh, outw, fh * fw)
		idx = np.argmax(flat, axis=4)
		r, s = idx // fw, idx % fw

		nn, cc, pp, qq = np.meshgrid(np.arange(n), np.arange(c), np.arange(outh), np.arange(outw), indexing="ij")
		np.add.at(dxp, (nn, cc, pp * sh + r, qq * sw + s), dy)

	else:
		if mode == POOL_AVG_WITH_PAD:
			g = dy / np.float32(fh * fw)
		else:
			ones, _ = _pool_windows(np.ones((1, 1, inh, inw), dtype=np.float32), size, stride, pad, 0.0)
			g = dy / ones.sum(axis=(4, 5))

		for r in range(fh):
			for s in range(fw):
				dxp[:, :, r:r + sh * outh:sh, s:s + sw * outw:sw] += g

	return np.ascontiguousarray(dxp[:, :, ph:ph + inh, pw:pw + inw])


# --------------------------------------------------------------------------------------------------
# softmax / cross-entropy / accuracy (a10, a11)
# --------------------------------------------------------------------------------------------------

def softmax_fwd(x):
	"""Channel softmax over axis 1 ("accurate": max-subtracted) — softmax2dTest Cuda/Wrappers/CuDnn.py:454-470."""
	e = np.exp(x - np.amax(x, axis=1, keepdims=True))
	return (e / np.sum(e, axis=1, keepdims=True)).astype(np.float32)


def softmax_bwd(dy, y):
	"""dx = y*(dy - sum_c(y*dy)) — Cuda/Wrappers/CuDnn.py:472-485."""
	return (y * (dy - np.sum(y * dy, axis=1, keepdims=True))).astype(np.float32)


def cross_entropy(scores, labels, weights=None):
	"""
	CostModule.crossEntropy — Cuda/Kernels/Costs.py:79-106,213-247: p = softmax(scores) over axis 1;
	grad = ((c==label) - p)/N (times weight[c] if given); error = sum(-log p[label])/spatial (weighted).
	Returns (error, grad); error is the *unnormalised-by-batch* device accumulator value.
	"""
	n, ncls = scores.shape[:2]
	spatial = int(np.prod(scores.shape[2:])) if scores.ndim > 2 else 1

	s3 = scores.reshape(n, ncls, spatial)
	p = softmax_fwd(s3)
	lab = labels.reshape(n, spatial)

	onehot = (np.arange(ncls)[None, :, None] == lab[:, None, :])
	grad = (onehot.astype(np.float32) - p) / np.float32(n)

	pl = np.take_along_axis(p, lab[:, None, :].astype(np.int64), axis=1)[:, 0, :]
	if weights is None:
		err = np.sum(-np.log(pl), dtype=np.float64) / spatial
	else:
		grad = grad * weights.reshape(1, ncls, 1)
		err = np.sum(-weights[lab] * np.log(pl), dtype=np.float64) / spatial

	return np.float32(err), grad.reshape(scores.shape).astype(np.float32)


def count_neq(x, y):
	"""calcAccuracy reduction — Cuda/Kernels/Costs.py:178-182: sum(x[i] != y[i]) as float32."""
	return np.float32(np.sum(x.ravel() != y.ravel()))


def mse(pred, target):
	"""
	MSE cost — Cost/MSE.py:8-21: grad = (target - pred)/numel; device error accumulator =
	dot(grad,grad)*numel*batch/2 (so error/batch = sum(diff^2)/(2*numel)). Returns (devErr, grad).
	"""
	numel = pred.size
	grad = ((target - pred) * np.float32(1.0 / numel)).astype(np.float32)
	err = np.float32(np.dot(grad.ravel().astype(np.float64), grad.ravel().astype(np.float64)) * numel * pred.shape[0] / 2.0)
	return err, grad


# --------------------------------------------------------------------------------------------------
# element-wise family (a8, a12, a13) — CPU/Kernels/ElementWise.py, loop bodies restated with ufuncs
# --------------------------------------------------------------------------------------------------

F = np.float32


def sigmoid(x):            return (F(1) / (F(1) + np.exp(-x))).astype(F)             # ElementWise.py:9-16
def sigmoid_der(g, y):     return (g * y * (F(1) - y)).astype(F)                      # :19-29
def tanh(x):               return np.tanh(x).astype(F)                                # :32-39
def tanh_der(g, y):        return (g * (F(1) - y * y)).astype(F)                      # :42-49
def relu(x):               return (x * (x > 0)).astype(F)                             # :52-59
def relu_der(g, y):        return (g * (y > 0)).astype(F)                             # :62-69
def leaky_relu(x, a):      return (x * ((x > 0) + F(a) * (x <= 0))).astype(F)         # :72-79
def leaky_relu_der(g, y, a): return (g * ((y > 0) + F(a) * (y <= 0))).astype(F)       # :82-89
def elu(x, a):             return (x * (x > 0) + F(a) * (np.exp(x) - F(1)) * (x <= 0)).astype(F)   # :92-99
def elu_der(g, y, a):      return (g * ((y > 0) + (y + F(a)) * (y <= 0))).astype(F)   # :102-109
def softplus(x):           return np.log(F(1) + np.exp(x)).astype(F)                  # :112-119
def softplus_der(g, y):    return (g * (F(1) - np.exp(-y))).astype(F)                 # :122-129


def clip(x, a, b):                                                                    # :132-139
	a, b = F(a), F(b)
	return (x * ((x > a) & (x < b)) + a * (x <= a) + b * (x >= b)).astype(F)


def clip_der(g, y, a, b):  return (g * ((y > F(a)) & (y < F(b)))).astype(F)           # :142-152


def gelu(x):
	"""Cuda/Kernels/ElementWise.py:427-456 (no CPU twin): 0.5*x*(1+erf(x/sqrt2))"""
	from math import erf
	return (F(0.5) * x * (F(1) + np.vectorize(erf)(x / np.sqrt(2.0)))).astype(F)


def gelu_der(g, x):
	"""Cuda/Kernels/ElementWise.py:459-492 — the reference's own formula (x/sqrt(pi) factor), restated as is."""
	from math import erf
	cdf = 0.5 * (1.0 + np.vectorize(erf)(x / np.sqrt(2.0)))
	return (g * (cdf + x / np.sqrt(np.pi) * np.exp(-0.5 * x * x))).astype(F)


def dropout(x, bits, v, p):
	"""dropoutKer — CPU/Kernels/ElementWise.py:155-165: out = x*(b<v)/p with a uint32 mask stream."""
	return (x * (bits.reshape(x.shape) < np.uint32(v)) / F(p)).astype(F)


def dropout2d(x, bits, v, p, mapsize):
	"""dropout2dKer — CPU/Kernels/ElementWise.py:168-179: one mask word per feature map."""
	idx = np.arange(x.size) // mapsize
	return (x.ravel() * (bits[idx] < np.uint32(v)) / F(p)).astype(F).reshape(x.shape)


def axpy(y, x, alpha=1.0):
	"""toVectorAddVectorKer — :182-189: y += alpha*x (in place)."""
	y += x * F(alpha)
	return y


def add_scaled(x, alpha, y, beta, out=None):
	"""addKer — :343-353 / Backend/Blas.py:51-58: out = alpha*x + beta*y."""
	res = (F(alpha) * x + F(beta) * y).astype(F)
	if out is None:
		return res
	out[...] = res
	return out


def mul(a, b):             return (a * b).astype(F)                                   # mulKer :356-363
def linear(x, a, b):       return (F(a) * x + F(b)).astype(F)                         # linearKer :366-373
def weight_decay(grad, param, rate):                                                  # weightDecayKer :383-387
	grad -= F(rate) * param
	return grad


def adam(param, grad, mg, ms, learn_rate, fix1, fix2, epsilon):
	"""adamKer — CPU/Kernels/ElementWise.py:235-249 (in place; update is +=)."""
	mg += F(fix1) * (grad - mg)
	ms += F(fix2) * (grad * grad - ms)
	param += F(learn_rate) * mg / (np.sqrt(ms) + F(epsilon))


def classic_mom_sgd(param, grad, mom, learn_rate, mom_rate):
	"""classicMomSGDKer — :252-265"""
	mom[...] = F(mom_rate) * mom + F(learn_rate) * grad
	param += mom


def nesterov_mom_sgd(param, grad, mom, learn_rate, mom_rate):
	"""nesterovMomSGDKer — :268-282"""
	m = mom.copy()
	mom[...] = F(mom_rate) * m + F(learn_rate) * grad
	param += F(mom_rate) * F(mom_rate) * m + (F(1) + F(mom_rate)) * F(learn_rate) * grad


def rmsprop(param, grad, ms, learn_rate, factor, epsilon):
	"""rmspropKer — :285-298"""
	ms[...] = F(factor) * ms + (F(1) - F(factor)) * grad * grad
	param += F(learn_rate) * grad / (np.sqrt(ms) + F(epsilon))


def adagrad(param, grad, h, learn_rate, epsilon):
	"""adagradKer — :219-232"""
	h += grad * grad
	param += F(learn_rate) * grad / (np.sqrt(h) + F(epsilon))


def adadelta(param, grad, msg, msdx, rho, epsilon):
	"""adadeltaKer — :201-216"""
	msg += (F(1) - F(rho)) * (grad * grad - msg)
	dx = np.sqrt((msdx + F(epsilon)) / (msg + F(epsilon))) * grad
	msdx += (F(1) - F(rho)) * (dx * dx - msdx)
	param += dx


def rmsprop_graves(param, grad, mg, ms, delta, learn_rate, alpha, mom_rate, epsilon):
	"""rmspropGravesKer — :301-317"""
	ms[...] = F(alpha) * ms + (F(1) - F(alpha)) * grad * grad
	mg[...] = F(alpha) * mg + (F(1) - F(alpha)) * grad
	delta[...] = F(mom_rate) * delta + F(learn_rate) * grad / np.sqrt(ms - mg * mg + F(epsilon))
	param += delta


def smorms3(param, grad, mem, mg, ms, learn_rate, epsilon):
	"""smorms3Ker — :320-340"""
	r = F(1) / (mem + F(1))
	mgi = (F(1) - r) * mg + r * grad
	msi = (F(1) - r) * ms + r * grad * grad
	x = mgi * mgi / (msi + F(epsilon))
	mem[...] = F(1) + mem * (F(1) - x)
	mg[...] = mgi
	ms[...] = msi
	param += grad * np.minimum(F(learn_rate), x) / (np.sqrt(msi) + F(epsilon))


def grad_mean_allreduce(grads):
	"""Data-parallel exchange — ParentNode.sumTensor Grid.py:123-135: g <- (g_0 + ... + g_{N-1}) / N for every rank."""
	n = len(grads)
	acc = grads[0].astype(np.float32) * F(1.0 / n)
	for g in grads[1:]:
		acc = acc + g * F(1.0 / n)
	return acc


# --------------------------------------------------------------------------------------------------
# operators beside the ResNet / NiN / LeNet path — host formulas of the reference's own unit tests
# --------------------------------------------------------------------------------------------------

def maskpool2d_fwd(x, size, stride, pad):
	"""Cuda/Kernels/Pool.py:9-46 maxpool2d kernel (and its host check :229-262): window maximum and the flat index
	h*W + w of the FIRST maximum inside the input plane (strict >), -1 for an empty window."""
	n, c, h, w = x.shape
	(fh, fw), (sh, sw), (ph, pw) = pair(size), pair(stride), pair(pad)
	oh, ow = (h - fh + 2 * ph) // sh + 1, (w - fw + 2 * pw) // sw + 1
	y = np.full((n, c, oh, ow), np.finfo(np.float32).min, np.float32)
	mask = np.full((n, c, oh, ow), -1, np.int32)
	for p in range(oh):
		for q in range(ow):
			h0, w0 = max(p * sh - ph, 0), max(q * sw - pw, 0)
			h1, w1 = min(p * sh - ph + fh, h), min(q * sw - pw + fw, w)
			if h1 <= h0 or w1 <= w0:
				continue
			win = x[:, :, h0:h1, w0:w1].reshape(n, c, -1)
			arg = win.argmax(axis=2)                              # first maximum
			y[:, :, p, q] = np.take_along_axis(win, arg[..., None], axis=2)[..., 0]
			mask[:, :, p, q] = (h0 + arg // (w1 - w0)) * w + (w0 + arg % (w1 - w0))
	return y, mask


def maskpool2d_bwd(dy, mask, inshape):
	"""Cuda/Kernels/Pool.py:66-97: every pooled gradient goes to the element its mask names."""
	n, c, h, w = inshape
	dx = np.zeros((n, c, h * w), np.float64)
	flat_dy, flat_m = dy.reshape(n, c, -1), mask.reshape(n, c, -1)
	for i in range(n):
		for j in range(c):
			ok = flat_m[i, j] >= 0
			np.add.at(dx[i, j], flat_m[i, j][ok], flat_dy[i, j][ok])
	return dx.reshape(inshape).astype(np.float32)


def maxunpool2d_fwd(x, mask, outshape):
	"""Cuda/Kernels/Pool.py:48-63"""
	n, c = x.shape[:2]
	y = np.zeros((n, c, outshape[2] * outshape[3]), np.float32)
	for i in range(n):
		for j in range(c):
			ok = mask[i, j].ravel() >= 0
			y[i, j][mask[i, j].ravel()[ok]] = x[i, j].ravel()[ok]
	return y.reshape(n, c, outshape[2], outshape[3])


def maxunpool2d_bwd(dy, mask):
	"""Cuda/Kernels/Pool.py:99-114"""
	n, c = dy.shape[:2]
	flat = dy.reshape(n, c, -1)
	safe = np.maximum(mask.reshape(n, c, -1), 0)
	return (np.take_along_axis(flat, safe, axis=2) * (mask.reshape(n, c, -1) >= 0)).reshape(mask.shape).astype(np.float32)


def lrn_norms(x, N, alpha, K, cross):
	"""normaliser of mapLRN2dTest / crossMapLRN2dTest (Cuda/Wrappers/CuDnnNorm.py:183-262)"""
	n, c, h, w = x.shape
	behind = (N - 1) // 2
	ahead = N - behind
	sq = x.astype(np.float64) ** 2
	s = np.empty(x.shape, np.float64)
	if cross:
		for ch in range(c):
			s[:, ch] = K + sq[:, max(0, ch - behind):min(c, ch + ahead)].sum(axis=1) * alpha / N
	else:
		for y in range(h):
			for xx in range(w):
				win = sq[:, :, max(0, y - behind):min(h, y + ahead), max(0, xx - behind):min(w, xx + ahead)]
				s[:, :, y, xx] = K + win.sum(axis=(2, 3)) * alpha / N ** 2
	return s


def lrn_fwd(x, N=5, alpha=1e-4, beta=0.75, K=2.0, cross=False):
	return (x / lrn_norms(x, N, alpha, K, cross) ** beta).astype(np.float32)


def lrn_bwd(x, dy, N=5, alpha=1e-4, beta=0.75, K=2.0, cross=False):
	n, c, h, w = x.shape
	behind = (N - 1) // 2
	ahead = N - behind
	s = lrn_norms(x, N, alpha, K, cross)
	t = dy.astype(np.float64) * x / s ** (beta + 1)
	acc = np.empty(x.shape, np.float64)
	if cross:
		for ch in range(c):
			acc[:, ch] = t[:, max(0, ch - behind):min(c, ch + ahead)].sum(axis=1)
		coef = 2.0 * alpha * beta / N
	else:
		for y in range(h):
			for xx in range(w):
				acc[:, :, y, xx] = t[:, :, max(0, y - behind):min(h, y + ahead), max(0, xx - behind):min(w, xx + ahead)].sum(axis=(2, 3))
		coef = 2.0 * alpha * beta / N ** 2
	return (dy / s ** beta - coef * x * acc).astype(np.float32)


def matvec(mat, vec, axis):
	"""Cuda/Kernels/MatVec.py:430-447 host check: per leading index, mat @ vec (axis 1) or mat.T @ vec (axis 0)"""
	m, v = mat.astype(np.float64), vec.astype(np.float64)
	return (np.einsum("...ij,...j->...i", m, v) if axis == 1 else np.einsum("...ij,...i->...j", m, v)).astype(np.float32)


def svm_cost(scores, labels, mode):
	"""Cuda/Kernels/Costs.py:109-130 + svmTest :327-350: (summed error, gradient)"""
	n, c = scores.shape[:2]
	spatial = int(np.prod(scores.shape[2:]))
	s = scores.reshape(n, c, spatial).astype(np.float64)
	cls = 2.0 * (labels.reshape(n, 1, spatial) == np.arange(c).reshape(1, c, 1)) - 1.0
	margin = np.maximum(0.0, 1.0 - s * cls)
	if mode == "l1":
		grad = np.where(s * cls < 1.0, cls / c / n, 0.0)
		err = margin.sum() / c / spatial
	else:
		grad = 2.0 * cls * margin / c / n
		err = (margin ** 2).sum() / c / spatial
	return np.float32(err), grad.reshape(scores.shape).astype(np.float32)


def conv3d_fwd(x, w, bias, stride, pad, dilation):
	"""Direct 3-d cross-correlation in fp64 (Modules/Conv3D.py over Dnn.convNd; groups = 1)."""
	n, c, D, H, W = x.shape
	k, _, T, R, S = w.shape
	(sd, sh, sw), (pd, ph, pw), (dd, dh, dw) = stride, pad, dilation
	xp = np.pad(x.astype(np.float64), ((0, 0), (0, 0), (pd, pd), (ph, ph), (pw, pw)))
	Do, Ho, Wo = (D + 2 * pd - dd * (T - 1) - 1) // sd + 1, (H + 2 * ph - dh * (R - 1) - 1) // sh + 1, (W + 2 * pw - dw * (S - 1) - 1) // sw + 1
	y = np.zeros((n, k, Do, Ho, Wo), np.float64)
	for t in range(T):
		for r in range(R):
			for s in range(S):
				win = xp[:, :, t * dd:t * dd + (Do - 1) * sd + 1:sd, r * dh:r * dh + (Ho - 1) * sh + 1:sh, s * dw:s * dw + (Wo - 1) * sw + 1:sw]
				y += np.einsum("ncdhw,kc->nkdhw", win, w[:, :, t, r, s].astype(np.float64))
	if bias is not None:
		y += bias.reshape(1, -1, 1, 1, 1)
	return y


def conv3d_bwd_data(dy, w, xshape, stride, pad, dilation):
	"""Adjoint of conv3d_fwd in fp64 (host loops of conv3dTest, Cuda/Wrappers/CuDnn.py:101-112)."""
	n, c, D, H, W = xshape
	k, _, T, R, S = w.shape
	(sd, sh, sw), (pd, ph, pw), (dd, dh, dw) = stride, pad, dilation
	_, _, Do, Ho, Wo = dy.shape
	dxp = np.zeros((n, c, D + 2 * pd, H + 2 * ph, W + 2 * pw), np.float64)
	g = dy.astype(np.float64)
	for t in range(T):
		for r in range(R):
			for s in range(S):
				dxp[:, :, t * dd:t * dd + (Do - 1) * sd + 1:sd, r * dh:r * dh + (Ho - 1) * sh + 1:sh, s * dw:s * dw + (Wo - 1) * sw + 1:sw] += \
					np.einsum("nkdhw,kc->ncdhw", g, w[:, :, t, r, s].astype(np.float64))
	return dxp[:, :, pd:pd + D, ph:ph + H, pw:pw + W]


def conv3d_bwd_filter(x, dy, wshape, stride, pad, dilation):
	"""Filter gradient of conv3d_fwd in fp64 (conv3dTest, Cuda/Wrappers/CuDnn.py:117-128); bias gradient = dy summed over
	everything but the maps."""
	k, c, T, R, S = wshape
	(sd, sh, sw), (pd, ph, pw), (dd, dh, dw) = stride, pad, dilation
	_, _, Do, Ho, Wo = dy.shape
	xp = np.pad(x.astype(np.float64), ((0, 0), (0, 0), (pd, pd), (ph, ph), (pw, pw)))
	g = dy.astype(np.float64)
	dw_ = np.zeros(wshape, np.float64)
	for t in range(T):
		for r in range(R):
			for s in range(S):
				win = xp[:, :, t * dd:t * dd + (Do - 1) * sd + 1:sd, r * dh:r * dh + (Ho - 1) * sh + 1:sh, s * dw:s * dw + (Wo - 1) * sw + 1:sw]
				dw_[:, :, t, r, s] = np.einsum("ncdhw,nkdhw->kc", win, g)
	return dw_, g.sum(axis=(0, 2, 3, 4))


def instance_norm_fwd(x, scale, bias, epsilon=1e-5):
	"""Backend.instanceNorm2d — Cuda/GPUBackend.py:381-400: batch normalisation of the (1, n*c, h, w) view with the affine pair
	tiled over the batch. Returns (y, savemean (n*c,), saveinvvar (n*c,), tiled scale)."""
	n, c, h, w = x.shape
	ext = np.tile(scale.ravel(), n) if n > 1 else scale.ravel()
	extb = np.tile(bias.ravel(), n) if n > 1 else bias.ravel()
	y, sm, si = bn_fwd_train(x.reshape(1, n * c, h, w), ext, extb, np.zeros(n * c, np.float32), np.ones(n * c, np.float32), epsilon, 1.0)
	return y.reshape(x.shape), sm, si, ext.astype(np.float32)


def instance_norm_bwd(dy, x, extscale, savemean, saveinvvar, affine=True):
	"""Backend.instanceNorm2dBackward — Cuda/GPUBackend.py:403-420"""
	n, c, h, w = x.shape
	dx, ds, db = bn_bwd(dy.reshape(1, n * c, h, w), x.reshape(1, n * c, h, w), extscale, savemean, saveinvvar)
	dx = dx.reshape(x.shape)
	if not affine:
		return dx
	if n > 1:
		ds, db = matsum(ds.reshape(n, -1), 0), matsum(db.reshape(n, -1), 0)
	return dx, ds, db


# point-wise cost kernels — Cuda/Kernels/Costs.py:8-72; each returns the SUM the kernel atomicAdds into totalError and the
# gradient(s). Arithmetic in float32 like the kernels (expf / logf), the sum in float64.
def bce_cost(scores, labels, numsamples, spatial):                                    # :8-22
	s = scores.astype(F)
	prob = F(1) / (F(1) + np.exp(-s))
	pos = labels.reshape(s.shape) == 1
	err = np.where(pos, -np.log(prob), -np.log(F(1) - prob)) / F(spatial)
	grad = (pos.astype(F) - prob) / F(numsamples) / F(spatial)
	return np.float32(err.sum(dtype=np.float64)), grad.astype(F)


def hinge_cost(scores, labels, numsamples, numcases):                                 # :25-40
	s, lab = scores.astype(F), labels.reshape(scores.shape)
	err = np.maximum(F(0), F(1) - s * lab) / F(numcases)
	grad = np.where(s * lab < F(1), lab.astype(F) / F(numsamples) / F(numcases), F(0))
	return np.float32(err.sum(dtype=np.float64)), grad.astype(F)


def smooth_l1_cost(pred, target, norm, fullnorm):                                     # :43-56
	diff = (pred - target).astype(F)
	sign = np.where(diff > 0, F(1), F(-1))
	quad = diff * sign < F(1)
	err = np.where(quad, diff * diff / F(2) * F(norm), (sign * diff - F(0.5)) * F(norm))
	grad = np.where(quad, diff * F(fullnorm), sign * F(fullnorm))
	return np.float32(err.sum(dtype=np.float64)), grad.astype(F)


def l1_hinge_cost(x1, x2, labels, numsamples, numcases):                              # :59-76
	diff = (x1 - x2).astype(F)
	sign = np.where(diff > 0, F(1), F(-1))
	lab = np.repeat(labels.ravel(), numcases).reshape(diff.shape)
	ad = np.abs(diff)
	err = np.where(lab == 0, np.maximum(F(0), F(1) - ad), ad) / F(numcases)
	inside = (ad < F(1)).astype(F)
	g1 = np.where(lab == 0, inside * -sign, sign) / F(numsamples) / F(numcases)
	g2 = np.where(lab == 0, inside * sign, -sign) / F(numsamples) / F(numcases)
	return np.float32(err.sum(dtype=np.float64)), g1.astype(F), g2.astype(F)


# PReLU — Cuda/Kernels/PRelu.py:14-56 (slope index = map, or 0 when the slope is shared: maps / divFactor)
def _slopes(x, slopes, shared):
	shape = (1, -1) + (1, ) * (x.ndim - 2)
	return (np.full(x.shape[1], slopes.ravel()[0], F) if shared else slopes.ravel().astype(F)).reshape(shape)


def prelu_fwd(x, slopes, shared=False):
	return (x * np.where(x > 0, F(1), _slopes(x, slopes, shared))).astype(F)


def prelu_bwd_data(dy, slopes, x, shared=False):
	return (dy * ((x > 0).astype(F) + (x <= 0).astype(F) * _slopes(x, slopes, shared))).astype(F)


def prelu_bwd_params(x, dy, shared=False):
	per = (dy.astype(np.float64) * x * (x <= 0)).reshape(x.shape[0], x.shape[1], -1).sum(axis=(0, 2))
	return (np.array([per.sum()]) if shared else per).astype(F)


# reflection pad — Cuda/Kernels/Pad.py:45-145. Source index of output position o along an axis of length `size` padded by
# (lpad, rpad >= 0): map1d's closed form |o - l| - |o - (size + l - 1)| - o + 2l + size - 1 - max(0, l) + max(0, -l)
def _reflect_index(size, lpad, rpad):
	o = np.arange(size + lpad + rpad)
	return np.abs(o - lpad) - np.abs(o - (size + lpad - 1)) - o + 2 * lpad + size - 1 - max(0, lpad) + max(0, -lpad)


def reflectpad_fwd(x, pad):
	if x.ndim == 3:
		return x[:, :, _reflect_index(x.shape[2], *pad)]
	upad, bpad, lpad, rpad = pad
	return x[:, :, _reflect_index(x.shape[2], upad, bpad)][:, :, :, _reflect_index(x.shape[3], lpad, rpad)]


def reflectpad_bwd(dy, pad):
	"""every output gradient is added to the input element it was copied from (the reference scatters with atomicAdd)"""
	if dy.ndim == 3:
		lpad, rpad = pad
		size = dy.shape[2] - lpad - rpad
		dx = np.zeros(dy.shape[:2] + (size, ), np.float64)
		np.add.at(dx, (slice(None), slice(None), _reflect_index(size, lpad, rpad)), dy)
		return dx.astype(F)
	upad, bpad, lpad, rpad = pad
	inh, inw = dy.shape[2] - upad - bpad, dy.shape[3] - lpad - rpad
	rows = np.zeros(dy.shape[:2] + (inh, dy.shape[3]), np.float64)
	np.add.at(rows, (slice(None), slice(None), _reflect_index(inh, upad, bpad)), dy)
	dx = np.zeros(dy.shape[:2] + (inh, inw), np.float64)
	np.add.at(dx, (slice(None), slice(None), slice(None), _reflect_index(inw, lpad, rpad)), rows)
	return dx.astype(F)


# up-sampling — Cuda/Kernels/Upsample.py:8-298 (2-d and 3-d; `scale` an int or one int per spatial axis)
def _scales(nd, scale):
	return (int(scale), ) * nd if isinstance(scale, (int, np.integer)) else tuple(int(v) for v in scale)


def _linear_taps(insize, outsize):
	"""(i0, i1, w0, w1) per output position with the kernels' float32 arithmetic: pos = r*o, i0 = (int)pos,
	i1 = i0 + (i0 < in - 1), w1 = pos - i0, w0 = 1 - w1; r = float32((in - 1) / (out - 1)) (Upsample.py:336,408)"""
	r = F((insize - 1) / (outsize - 1)) if outsize > 1 else F(0)
	pos = r * np.arange(outsize, dtype=F)
	i0 = pos.astype(np.int64)
	i1 = i0 + (i0 < insize - 1)
	w1 = (pos - i0.astype(F)).astype(F)
	return i0, i1, (F(1) - w1).astype(F), w1


def _linear_matrix(insize, outsize):
	"""(out, in) interpolation matrix of one axis"""
	i0, i1, w0, w1 = _linear_taps(insize, outsize)
	m = np.zeros((outsize, insize), np.float64)
	np.add.at(m, (np.arange(outsize), i0), w0)
	np.add.at(m, (np.arange(outsize), i1), w1)
	return m


def upsample_fwd(x, scale, mode="nearest"):
	nd = x.ndim - 2
	scales = _scales(nd, scale)
	y = x.astype(np.float64)
	for axis, s in enumerate(scales):
		if mode == "nearest":
			y = np.repeat(y, s, axis=2 + axis)
		else:
			insize = x.shape[2 + axis]
			y = np.moveaxis(np.tensordot(_linear_matrix(insize, insize * s), y, axes=(1, 2 + axis)), 0, 2 + axis)
	return y.astype(F)


def upsample_bwd(dy, scale, mode="nearest"):
	nd = dy.ndim - 2
	scales = _scales(nd, scale)
	if mode == "nearest":
		# float32 running sum over the block in (depth, row, column) order, as upsample3dNearestBackward does
		# (Upsample.py:84-104; the 2-d kernel :36-54 walks columns first — same terms)
		sd, sh, sw = ((1, ) + scales) if nd == 2 else scales
		g5 = dy.reshape(dy.shape[:2] + ((1, ) if nd == 2 else ()) + dy.shape[2:]).astype(F)
		acc = np.zeros(g5.shape[:2] + (g5.shape[2] // sd, g5.shape[3] // sh, g5.shape[4] // sw), F)
		for a in range(sd):
			for b in range(sh):
				for c in range(sw):
					acc = (acc + g5[:, :, a::sd, b::sh, c::sw]).astype(F)
		return acc.reshape(acc.shape[:2] + acc.shape[(3 if nd == 2 else 2):])
	g = dy.astype(np.float64)
	for axis, s in enumerate(scales):
		outsize = dy.shape[2 + axis]
		insize = outsize // s
		if mode == "nearest":
			shape = g.shape[:2 + axis] + (insize, s) + g.shape[3 + axis:]
			g = g.reshape(shape).sum(axis=3 + axis)
		else:
			g = np.moveaxis(np.tensordot(_linear_matrix(insize, outsize).T, g, axes=(1, 2 + axis)), 0, 2 + axis)
	return g.astype(F)


# embedding — Cuda/Kernels/Embedder.py:10-45 (word index -1 = padding: zero row, no update)
def embed_fwd(words, vocab):
	out = vocab[np.maximum(words, 0)].astype(F)
	out[words == -1] = 0
	return out


def embed_bwd_params(words, grad, vocab, scale):
	"""vocabulary[word] += scale * grad[token] for every token (in place; rows repeat, so the adds accumulate)"""
	ok = words.ravel() != -1
	np.add.at(vocab, words.ravel()[ok], (F(scale) * grad.reshape(-1, grad.shape[-1])[ok]).astype(F))
	return vocab


# CTC loss — Cuda/Kernels/CTC.py:9-270 (kernels calcAlphas / calcBetas and the module's softmax / offsets bookkeeping)
def ctc_loss(data, datalen, labels, lengths, blank, normalized=False):
	"""(summed negative log-likelihood, grad (T, batch, vocab), alphas flat) for scores `data` (T, batch, vocab); the forward
	and backward variables in log space over the extended label sequence blank, l0, blank, l1, ..., blank of each sample,
	float32 like the kernels; grad[t, b, v] = p - exp(logsum_{i: ext[i] = v}(alpha + beta) - log p + nll_b), negated — what
	the kernel leaves (Cost/CTC.py hands it on as the descent direction); zero beyond datalen and for impossible samples."""
	T, batch, vocab = data.shape
	if normalized:
		p = data.astype(F)
	else:
		e = np.exp(data - data.max(axis=-1, keepdims=True))
		p = (e / e.sum(axis=-1, keepdims=True)).astype(F)
	offsets = np.concatenate(([0], np.cumsum(lengths))).astype(np.int64)
	alphas = np.full(T * (2 * int(offsets[-1]) + batch), np.nan, F)
	grad = np.zeros(p.shape, F)
	total = np.float64(0.0)
	NEG = F(-np.inf)

	def lp(a, b):                  # logPlus, element-wise, float32
		with np.errstate(invalid="ignore", divide="ignore"):
			m, d = np.maximum(a, b), -np.abs(a - b)
			out = (np.log1p(np.exp(d, dtype=F), dtype=F) + m).astype(F)
		out = np.where(np.isneginf(a), b, out)
		return np.where(np.isneginf(b), a, out).astype(F)

	for b in range(batch):
		L, Tb = int(lengths[b]), int(datalen[b])
		S = 2 * L + 1
		ext = np.full(S, blank, np.int64)
		ext[1::2] = labels[offsets[b]:offsets[b] + L]
		skip = np.zeros(S, bool)                               # position i may also be reached from i - 2
		skip[2:] = (ext[2:] != blank) & (ext[2:] != ext[:-2])
		base = (2 * int(offsets[b]) + b) * T
		alpha = alphas[base:base + T * S].reshape(T, S)
		with np.errstate(divide="ignore"):
			logp = np.log(p[:, b, :][:, ext]).astype(F)        # (T, S)

		alpha[0] = NEG
		alpha[0, :2] = logp[0, :2]
		for t in range(1, Tb):
			prev = alpha[t - 1].copy()
			prev[1:] = lp(prev[1:], alpha[t - 1, :-1])
			via2 = np.full(S, NEG, F)
			via2[2:] = alpha[t - 1, :-2]
			prev = np.where(skip, lp(prev, via2), prev)
			alpha[t] = prev + logp[t]
		loglike = lp(alpha[Tb - 1, S - 2], alpha[Tb - 1, S - 1]) if S > 1 else alpha[Tb - 1, 0]
		nll = F(-loglike)
		total += np.float64(nll)
		if not np.isfinite(nll):
			continue

		fwd_skip = np.zeros(S, bool)                           # position i may also go on to i + 2
		fwd_skip[:-2] = (ext[:-2] != blank) & (ext[:-2] != ext[2:])
		beta = np.full(S, NEG, F)
		beta[max(S - 2, 0):] = logp[Tb - 1, max(S - 2, 0):]
		for t in range(Tb - 1, -1, -1):
			if t < Tb - 1:
				nxt = beta.copy()
				nxt[:-1] = lp(nxt[:-1], beta[1:])
				via2 = np.full(S, NEG, F)
				via2[:-2] = beta[2:]
				nxt = np.where(fwd_skip, lp(nxt, via2), nxt)
				beta = (nxt + logp[t]).astype(F)
			ab = (alpha[t] + beta).astype(F)
			g = -p[t, b].copy()
			for v in np.unique(ext):
				acc = NEG
				for i in np.flatnonzero(ext == v):             # ascending position, as the sorted segments are walked
					acc = lp(acc, ab[i])
				if p[t, b, v] > 0:
					g[v] += np.exp(F(acc) - np.log(p[t, b, v]) + nll, dtype=F)
			grad[t, b] = g
	return np.float32(total), grad, alphas
