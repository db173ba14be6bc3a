"""
Lazy device buffers: how this backend fuses across the reference's operator calls without any change to their signatures.

The reference's modules talk to a backend through fixed wrappers (Backend/Dnn.py:131-268, Backend/Blas.py:43-75,
Backend/Kernels/ElementWise.py) — one call per operator, tensors in, tensors out. A ResNet block therefore arrives as
`convNd`, `batchNormNd`, `reluKer(y, y)`, ..., `fill(0)`, `toVectorAddVector` x2, `reluKer` — each of which a literal
backend turns into at least one pass over HBM. This backend instead lets an operator return a tensor whose contents are
*described* but not yet written (a pending `Thunk` on the tensor's device buffer) whenever the description is cheap to
keep — a batch-norm's affine pair, a zero fill, a sum of terms, a ReLU on top, a gate below — and runs one fused kernel
when somebody needs the values. Consumers that can evaluate the description on the fly (the residual sum, the gradient
fan-in, the 1x1 convolution's backward gathers) never make the tensor exist at all.

Correctness does not depend on who calls what next: every access to device memory goes through `GPUArray.rptr`
(read), `.wptr` (read-modify-write or partial write), `.optr` (whole overwrite) or `.ptr` (unknown use = both), and
those are the barriers —
  read   : a pending thunk on the buffer is run first; a write by another stream is waited for;
  write  : additionally every buffer whose thunk or derived facts (`meta`) depend on the old contents is settled first
           (thunks run, facts dropped), and reads by another stream are waited for.
State lives on the *root* allocation (`Buffer.root.lz`), so reshapes / ravels / slices of a tensor share it. A buffer
without state (the overwhelmingly common case) costs one attribute test per access.

The same machinery orders work across streams: a kernel launched on a foreign stream (the filter-gradient stream of
DnnContext, an optimizer's borrowed streams) leaves its completion event on the buffers it touched; the main stream
waits only when — and if — it touches them.

`enabled = False` (PUZZLE_MI355_LAZY=0) runs every producer's thunk on the spot: the literal one-kernel-per-call
behaviour, used by the parity tests as the reference point the fused paths must reproduce bit for bit.
"""
import os, weakref
from collections import deque

from puzzlelib_amd import lib

enabled = os.environ.get("PUZZLE_MI355_LAZY", "1") == "1"
disabled = set(filter(None, os.environ.get("PUZZLE_MI355_LAZY_OFF", "").split(",")))      # individual patterns, for tests
# Patterns that are OFF unless asked for (`requested`; tests assign `disabled` freely, so this is a set of its own):
# "dgradstats" (round 6: BatchNorm-backward statistics from the producing backward-data epilogue, pz_conv2d_bwd_data_bnstats)
# changes the summation order of those statistics — opt-in until measured on the device (PUZZLE_MI355_DGRAD_STATS=1)
OPT_IN = frozenset(("dgradstats", ))
requested = set(p for p, env in (("dgradstats", "PUZZLE_MI355_DGRAD_STATS"), ) if os.environ.get(env, "0") == "1")
counters = {}                      # pattern name -> times taken (tests and tools read it)
writeOp = None                     # element-wise op id whose operands are being fetched (gpuarray.eltwise): a watcher on a written
                                   # allocation (State.watch) can tell a known kernel — the weight-decay hook — from an unknown write


def count(name):
	counters[name] = counters.get(name, 0) + 1


def on(pattern):
	return enabled and pattern not in disabled and (pattern not in OPT_IN or pattern in requested)


class State:
	"""thunk: pending contents; deps: weak references to allocations whose thunk / facts derive from this one's contents;
	meta: facts about the contents; wev / rev: [(event, stream, lo, hi)] — byte ranges a foreign stream still writes / reads"""
	__slots__ = ("thunk", "deps", "meta", "wev", "rev", "base", "small", "version", "snap", "arena", "watch")

	def __init__(self, base):
		self.thunk, self.deps, self.meta, self.wev, self.rev, self.base = None, None, None, None, None, base
		self.small = None         # [(lo, hi, written)] byte ranges that queued small adds (deferAdd) will write / read
		self.version = 0          # bumped by every write barrier: what was derived from the contents (prepared filter
		                          # operands, DnnContext.prepared) is current while the number stands
		self.snap = None          # (version, Buffer): a copy of the allocation as of that version (lazy.snapshot)
		self.arena = None         # [(name, byte offset, nbytes)] when the allocation is a flat arena (backend.SharedArray.build)
		self.watch = None         # watch(lo, hi): told of every write barrier's byte range before the write is issued — the
		                          # data-parallel exchange follows the gradient arena's writes with it (grid.ArenaWatcher)


def stateOf(root):
	lz = root.lz
	if lz is None:
		lz = root.lz = State(root.ptr)
	return lz


class Thunk:
	"""Description of a buffer's contents. `inputs()` lists the GPUArrays it reads (kept alive by the thunk);
	`run(out)` writes the values into `out` (a GPUArray over the whole buffer) and may return facts for `meta`."""
	shape = dtype = None

	def inputs(self):
		return ()

	def run(self, out):
		raise NotImplementedError()

	def dependsOn(self, root):
		"""does the description (still) read allocation `root`? (a description may have moved to a snapshot of it)"""
		return True


def attach(ary, thunk):
	"""Makes `thunk` the pending contents of `ary`'s buffer (which must be covered entirely by `ary`)."""
	root = ary.gpudata.root
	lz = stateOf(root)
	assert lz.thunk is None
	thunk.shape, thunk.dtype = ary.shape, ary.dtype
	lz.thunk = thunk
	for src in thunk.inputs():
		depend(src, root)
	if not enabled:
		settle(root)


def depend(src, root):
	"""`root`'s pending contents or facts derive from `src`'s current contents."""
	slz = stateOf(src.gpudata.root)
	if slz.deps is None:
		slz.deps = []
	slz.deps.append(weakref.ref(root))


def pending(ary, kind=None):
	"""The thunk waiting on `ary`'s buffer if `ary` covers that buffer entirely (and is of class `kind`), else None."""
	buf = ary.gpudata
	lz = buf.root.lz
	if lz is None or lz.thunk is None:
		return None
	if ary.nbytes != buf.root.size or buf.ptr != buf.root.ptr or not ary.contiguous:
		return None
	return lz.thunk if kind is None or isinstance(lz.thunk, kind) else None


def fact(ary, key):
	"""A fact recorded about the contents of `ary`'s allocation — only if `ary` IS that allocation (contiguous, whole):
	facts live on the root buffer, and a slice or strided view of it is a different tensor. Facts that depend on how the
	bytes are cut into axes (per-channel statistics) carry the shape they hold for; their readers compare it."""
	root = ary.gpudata.root
	lz = root.lz
	if lz is None or lz.meta is None:
		return None
	buf = ary.gpudata
	if buf.ptr != root.ptr or ary.nbytes != root.size or not ary.contiguous:
		return None
	return lz.meta.get(key, None)


def setFact(ary, key, value, sources=()):
	root = ary.gpudata.root
	lz = stateOf(root)
	if lz.meta is None:
		lz.meta = {}
	lz.meta[key] = value
	for src in sources:
		depend(src, root)


def settle(root):
	"""Runs the pending thunk of a root buffer."""
	lz = root.lz
	thunk = lz.thunk
	if thunk is None:
		return
	lz.thunk = None                       # first: run() reads its inputs through the barriers, never itself
	from puzzlelib_amd.gpuarray import GPUArray
	out = GPUArray(thunk.shape, thunk.dtype, gpudata=root)
	facts = thunk.run(out)
	if facts:
		if lz.meta is None:
			lz.meta = {}
		lz.meta.update(facts)
		for src in thunk.inputs():          # the facts describe `out` in terms of the inputs' contents
			depend(src, root)


def waitEvents(lz, read, stream, buf):
	"""Cross-stream ordering: before the accessing stream reads (writes) the bytes of `buf`, foreign writes (and reads) of
	overlapping bytes finish. Events are kept per byte range of the allocation: the optimizer's flat gradient arena is one
	allocation in which the filter-gradient stream and the main stream write different parameters' blocks side by side."""
	lo = buf.ptr - lz.base
	hi = lo + buf.size
	for name in (("wev", ) if read else ("wev", "rev")):
		entries = getattr(lz, name)
		if not entries:
			continue
		keep = []
		for entry in entries:
			event, owner, elo, ehi = entry
			if owner is stream or ehi <= lo or elo >= hi:
				keep.append(entry)
				continue
			lib.pz_stream_wait_event(None if stream is None else stream.handle, event.handle)
			if stream is not None:
				keep.append(entry)              # only the main stream's wait retires the entry (everything later follows it)
		setattr(lz, name, keep or None)


def readBarrier(root, buf=None, stream=None):
	lz = root.lz
	if lz.small is not None:
		touchSmall(lz, root if buf is None else buf, False)
	if lz.thunk is not None:
		settle(root)
	if lz.wev is not None:
		waitEvents(lz, True, stream, root if buf is None else buf)


def settleDependents(root):
	"""Everything whose pending description or facts derive from `root`'s CURRENT value is made independent of it: thunks
	run (they read `root` through the read barrier, which writes `root`'s own pending description if they need it), facts
	are dropped. Called before the value changes — by a write, or by an operator that is about to edit `root`'s pending
	description in place (a ReLU / gate / further term joining it)."""
	lz = root.lz
	if lz is None or lz.deps is None:
		return
	deps, lz.deps = lz.deps, None
	for ref in deps:
		other = ref()
		if other is not None and other.lz is not None and other is not root:
			if other.lz.thunk is not None and other.lz.thunk.dependsOn(root):
				settle(other)
			other.lz.meta = None
	lz.deps = None                            # (a dependent that settled re-registered its facts: they were just dropped)


def writeBarrier(root, buf=None, whole=False, stream=None):
	lz = root.lz
	lz.version += 1
	lz.snap = None                            # (a copy of the previous version: whoever still reads it holds it)
	if lz.watch is not None:
		lo = 0 if buf is None else buf.ptr - lz.base
		lz.watch(lo, lo + (root.size if buf is None else buf.size))
	if lz.small is not None:
		touchSmall(lz, root if buf is None else buf, True)
	# dependents first: one of them may need this buffer's own pending contents (B = copy of A while A is a pending zero
	# fill, then A is overwritten — dropping A's description before B ran would let B copy unwritten memory)
	if lz.deps is not None:
		settleDependents(root)
	if lz.thunk is not None:
		if whole:
			lz.thunk = None
		else:
			settle(root)
	lz.meta = None
	if lz.wev is not None or lz.rev is not None:
		waitEvents(lz, False, stream, root if buf is None else buf)


def editable(ary, kind=None):
	"""The pending description of `ary` (as `pending`) for an operator that wants to edit it in place. Editing changes the
	tensor's value, so whoever captured the tensor by reference is settled first; if that made the description run there
	is nothing left to edit and the caller falls back to its kernel."""
	thunk = pending(ary, kind)
	if thunk is None:
		return None
	root = ary.gpudata.root
	if root.lz.deps is not None:
		settleDependents(root)
		thunk = pending(ary, kind)
	return thunk


# ---------------------------------------------------------------------------------------------- queued small adds
# `out = alpha*x + beta*y` on a few hundred elements is all launch latency; a training step issues dozens of them back to
# back (two per BatchNorm layer: Modules/BatchNormND.py:86-92). They are queued and run as ONE launch (pz_multi_add) as
# soon as anybody touches bytes one of them writes, or writes bytes one of them reads — the barriers below check the byte
# ranges — or when the queue is full. Operands are settled when the job is queued, so the launch itself needs no barrier.
smallQueue = []          # (out, x, y, alpha, beta) in call order
smallRoots = []          # allocations that carry ranges of queued jobs


def deferAdd(out, x, y, alpha, beta):
	ptrs = (out.wptr, x.rptr, y.rptr)                      # settles descriptions, waits for foreign streams — now
	for ary, written in ((out, True), (x, False), (y, False)):
		buf = ary.gpudata
		root = buf.root
		lz = stateOf(root)
		if lz.small is None:
			lz.small = []
			smallRoots.append(root)
		lo = buf.ptr - lz.base
		lz.small.append((lo, lo + buf.size, written))
	smallQueue.append((out, x, y, float(alpha), float(beta), ptrs))
	count("small_add_queued")
	if len(smallQueue) >= lib.MULTI_ADD_MAX:
		flushSmall()


def touchSmall(lz, buf, write):
	lo = buf.ptr - lz.base
	hi = lo + buf.size
	for slo, shi, written in lz.small:
		if slo < hi and lo < shi and (write or written):
			flushSmall()
			return


def flushSmall():
	global smallQueue, smallRoots
	if not smallQueue:
		return
	jobs, smallQueue = smallQueue, []
	for root in smallRoots:
		if root.lz is not None:
			root.lz.small = None
	smallRoots = []

	import ctypes
	n = len(jobs)
	P = ctypes.c_void_p * n
	outs, xs, ys = P(*[j[5][0] for j in jobs]), P(*[j[5][1] for j in jobs]), P(*[j[5][2] for j in jobs])
	alphas = (ctypes.c_float * n)(*[j[3] for j in jobs])
	betas = (ctypes.c_float * n)(*[j[4] for j in jobs])
	sizes = (ctypes.c_uint32 * n)(*[j[0].size for j in jobs])
	lib.pz_multi_add(n, outs, xs, ys, alphas, betas, sizes, None)
	count("small_add_launches")


# ---------------------------------------------------------------------------------------------- foreign streams
held = deque()           # (event, objects) kept alive until the event has passed: memory a foreign stream still uses
spare = []               # events ready for reuse (creating / destroying one per launch is two library calls too many)
# The host issues a step faster than the device runs it, and what a foreign-stream launch reads (activations, gradients,
# its workspace) stays allocated until the device has passed it: without a bound the footprint grows with the host's
# lead and differs from step to step, so the pool keeps missing (measured: 60-70 hipMalloc calls, 25-30 ms of host time,
# per ResNet-50 step). The host therefore waits once more than this many foreign launches are outstanding.
maxHeld = int(os.environ.get("PUZZLE_MI355_MAX_FOREIGN", "12"))


def newEvent():
	from puzzlelib_amd.driver import Event
	return spare.pop() if spare else Event()


def prune(block=False):
	while held:
		event, _ = held[0]
		if block:
			event.synchronize()
		else:
			done = lib.c_int(0)
			lib.pz_event_query(event.handle, lib.byref(done))
			if not done.value:
				break
		_, objects = held.popleft()
		if len(spare) < 256:
			# (a buffer may still list `event` as a past write: waiting for its next recording instead only waits longer)
			spare.append(event)
			spare.append(objects[0])              # the launch's `ready` event


def foreignBegin(stream, ready=None):
	"""Everything issued on the main stream so far — or up to `ready`, an event the caller recorded on it earlier and knows
	to lie behind every write of what the launch reads — happens before what `stream` is given next."""
	prune()
	if ready is None:
		ready = newEvent()
		ready.record(None)
	stream.waitEvent(ready)
	return ready


def quiet(ary):
	"""touching `ary` (read or write) launches nothing on the main stream: no pending contents, no dependents to settle, no
	queued small adds over its allocation (events of foreign streams do not count: the toucher waits for those itself)"""
	lz = ary.gpudata.root.lz
	return lz is None or (lz.thunk is None and not lz.deps and lz.small is None)


def cleanSince(root, version):
	"""nothing pending on allocation `root` and no write barrier passed since its version counter read `version`"""
	lz = root.lz
	return lz is not None and lz.thunk is None and lz.version == version


def foreignEnd(stream, ready, reads=(), writes=(), keep=()):
	"""Marks the buffers a foreign-stream launch touched with its completion event; the main stream waits for it when (and
	only when) it touches them; the memory stays referenced until the event has passed."""
	done = newEvent()
	done.record(stream)
	for name, arrays in (("wev", writes), ("rev", reads)):
		for ary in arrays:
			buf = ary.gpudata
			lz = stateOf(buf.root)
			lo = buf.ptr - lz.base
			entries = getattr(lz, name) or []
			# a later event of the same stream over the same bytes supersedes the earlier one
			entries = [e for e in entries if not (e[1] is stream and e[2] >= lo and e[3] <= lo + buf.size)]
			entries.append((done, stream, lo, lo + buf.size))
			setattr(lz, name, entries)
	held.append((done, (ready, tuple(reads), tuple(writes), tuple(keep))))
	if len(held) > maxHeld:
		held[0][0].synchronize()
		prune()
	return done


def joinAll():
	"""The main stream waits for everything foreign streams were given (device-wide ordering point)."""
	prune()
	for event, _ in held:
		lib.pz_stream_wait_event(None, event.handle)


# ---------------------------------------------------------------------------------------------- the simplest thunk
class Zero(Thunk):
	"""All zeros (GPUArray.fill(0) on a whole fresh tensor). Terms added to it turn it into fusion.Sum."""
	threshold = 4096          # bytes; smaller tensors are simply memset

	def run(self, out):
		lib.pz_memset_d32(out.gpudata.ptr, 0, out.size, None)


def whole(ary):
	"""`ary` spans its entire allocation contiguously (so a description of the array is a description of the buffer)"""
	buf = ary.gpudata
	root = buf.root
	return ary.contiguous and buf.ptr == root.ptr and ary.nbytes == root.size


snapshotWhole = 32 << 20          # allocations up to this size are snapshot whole (one copy shared by every description of the step)


def snapshot(ary):
	"""A view like `ary` over a COPY of its allocation as it stands now. One copy per allocation and write-version: with the
	parameters in one flat arena, every description of a step that has to outlive the optimizer's update (fusion.ConvFwd /
	ConvBwdData twins whose activated version was written instead) shares a single device-to-device copy."""
	from puzzlelib_amd.gpuarray import GPUArray
	from puzzlelib_amd import driver
	buf = ary.gpudata
	root = buf.root
	lz = stateOf(root)
	readBarrier(root)
	if root.size > snapshotWhole and buf.size < root.size:
		# a large arena (a VGG-sized model: 0.5 GB): copying all of it for one filter would move and hold a second copy of every
		# parameter each step — the caller gets a private copy of the bytes it reads
		alloc = GPUArray.defaultAllocator.allocate if GPUArray.defaultAllocator is not None else driver.Buffer.allocate
		piece = alloc(buf.size)
		lib.pz_memcpy_d2d(piece.ptr, buf.ptr, buf.size, None)
		count("param_snapshot_piece")
		return GPUArray(ary.shape, ary.dtype, gpudata=piece)
	if lz.snap is None or lz.snap[0] != lz.version:
		copy = GPUArray.defaultAllocator.allocate(root.size) if GPUArray.defaultAllocator is not None else driver.Buffer.allocate(root.size)
		lib.pz_memcpy_d2d(copy.ptr, root.ptr, root.size, None)
		lz.snap = (lz.version, copy)
		count("param_snapshot")
	offset = buf.ptr - root.ptr
	return GPUArray(ary.shape, ary.dtype, gpudata=lz.snap[1][offset:offset + buf.size])


def sameBuffer(a, b):
	return a.gpudata.root is b.gpudata.root and a.gpudata.ptr == b.gpudata.ptr and a.nbytes == b.nbytes


def rawRead(ary):
	"""Address of a buffer's *stored* bytes (a pending in-place description such as fusion.Gate is not applied):
	only foreign-stream writes are waited for."""
	lz = ary.gpudata.root.lz
	if lz is not None and lz.wev is not None:
		waitEvents(lz, True, None, ary.gpudata)
	return ary.gpudata.ptr
