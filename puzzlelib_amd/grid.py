"""
Data-parallel training: one process per GPU, mini-batch sharded along N, mean all-reduce of the flat gradient arena.

The reference (Grid.py:4-157) runs one multiprocessing.Process per GPU and reduces gradients through a star: the
parent maps each child's flat gradient buffer by CUDA/HIP IPC handle, adds them one by one with an axpy kernel and
copies the mean back — O(N) serial full-buffer passes, fully serialised with backward. Here the `nodeinfo` object the
optimizers talk to (Optimizers/Optimizer.py:107-109,166-167: broadcastBuffer / sumTensor / meanValue) keeps its API
but sits on RCCL collectives over xGMI:

  broadcastBuffer(name, buffer)  ->  ncclBroadcast from rank 0 (initial parameter sync)
  sumTensor(name, tensor)        ->  g <- (g_0 + ... + g_{N-1}) / N  via ncclAllReduce(sum) + one scale kernel
  meanValue(value)               ->  scalar mean over ranks

and the gradient exchange is bucketed and overlapped with backward: the flat arena — laid out in the order backward
finishes the gradients (puzzlelib_amd/optim.py, Optimizer.arenaOrder) — is cut into buckets of ~25 MB; the executor
reports layers whose parameter gradients are final, and as soon as a bucket's completion set is full its all-reduce is
queued on a dedicated communication stream behind events of the compute stream and of the filter-gradient stream.
`sumTensor` at update time only queues what is still missing, makes the compute stream wait and scales by 1/N.

Ranks are separate processes (started by `python -m torch.distributed.run`, or by bench.py itself); everything they
exchange on the host — the 128-byte RCCL id, yes/no votes, scalar means, barriers — goes over a plain TCP star on
MASTER_ADDR (`HostGroup`): no PyTorch in the product path.
"""
import os, socket, struct, sys, time

import numpy as np


# ---------------------------------------------------------------------------------------------- host-side group (TCP star)
class stdoutToStderr:
	"""librccl prints a version banner with printf on the first communicator: it sits in the C library's stdout buffer
	and comes out when the process exits — behind whatever the program printed last (bench.py's one JSON line must stay
	the last line of stdout). Inside this block file descriptor 1 is descriptor 2, and the C buffers are flushed on both
	sides of it."""

	def __enter__(self):
		import ctypes
		self.libc = ctypes.CDLL(None)
		sys.stdout.flush()
		self.libc.fflush(None)
		self.saved = os.dup(1)
		os.dup2(2, 1)

	def __exit__(self, *exc):
		self.libc.fflush(None)
		os.dup2(self.saved, 1)
		os.close(self.saved)
		return False


class HostGroup:
	"""Rank 0 listens on (MASTER_ADDR, port) and keeps one connection per peer. Collectives are a gather to rank 0
	followed by a scatter of the result: bytes broadcast, float64 reductions, float32 array sums (the fallback gradient
	transport), barrier. Messages are length-prefixed; every call is made by all ranks in the same order."""

	def __init__(self, rank, world, addr, port, timeout=120.0, publish=None):
		"""`port` 0 on rank 0: the system picks a free one; `publish(port)` is told which, once the socket listens"""
		self.rank, self.world = rank, world
		self.peers = []
		if world == 1:
			return

		if rank == 0:
			server = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
			server.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
			server.bind((addr, port))
			server.listen(world)
			if publish is not None:
				publish(server.getsockname()[1])
			server.settimeout(timeout)
			slots = [None] * world
			for _ in range(world - 1):
				conn, _ = server.accept()
				conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
				conn.settimeout(None)
				peer = struct.unpack("<i", self.recvExact(conn, 4))[0]
				slots[peer] = conn
			server.close()
			self.peers = slots
		else:
			deadline = time.time() + timeout
			while True:
				try:
					conn = socket.create_connection((addr, port), timeout=5.0)
					break
				except OSError:
					if time.time() > deadline:
						raise
					time.sleep(0.05)
			conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
			conn.settimeout(None)
			conn.sendall(struct.pack("<i", rank))
			self.root = conn

	@staticmethod
	def recvExact(conn, n):
		chunks = []
		while n > 0:
			part = conn.recv(min(n, 1 << 20))
			if not part:
				raise ConnectionError("peer closed the host group connection")
			chunks.append(part)
			n -= len(part)
		return b"".join(chunks)

	@classmethod
	def send(cls, conn, payload):
		conn.sendall(struct.pack("<q", len(payload)) + payload)

	@classmethod
	def recv(cls, conn):
		(n, ) = struct.unpack("<q", cls.recvExact(conn, 8))
		return cls.recvExact(conn, n)

	def gatherScatter(self, payload, combine):
		"""rank 0 applies `combine([payload_0, ..., payload_{N-1}]) -> bytes`; every rank returns the result"""
		if self.world == 1:
			return combine([payload])
		if self.rank == 0:
			parts = [payload] + [self.recv(self.peers[r]) for r in range(1, self.world)]
			result = combine(parts)
			for r in range(1, self.world):
				self.send(self.peers[r], result)
			return result
		self.send(self.root, payload)
		return self.recv(self.root)

	def broadcast(self, payload):
		return self.gatherScatter(payload if self.rank == 0 else b"", lambda parts: parts[0])

	def reduce(self, value, op="sum"):
		fn = {"sum": sum, "min": min, "max": max}[op]
		out = self.gatherScatter(
			struct.pack("<d", float(value)), lambda parts: struct.pack("<d", fn(struct.unpack("<d", p)[0] for p in parts))
		)
		return struct.unpack("<d", out)[0]

	def sumArray(self, array):
		"""in-place element-wise sum of a float32 array over the ranks (ascending rank order on every element)"""
		def combine(parts):
			total = np.frombuffer(parts[0], dtype=np.float32).copy()
			for part in parts[1:]:
				total += np.frombuffer(part, dtype=np.float32)
			return total.tobytes()
		array[...] = np.frombuffer(self.gatherScatter(array.tobytes(), combine), dtype=np.float32).reshape(array.shape)

	def barrier(self):
		self.reduce(0.0)

	def close(self):
		for conn in self.peers[1:] if self.rank == 0 else ([self.root] if self.world > 1 else []):
			try:
				conn.close()
			except OSError:
				pass


hostGroup = None          # set by nodeFromEnv


# ---------------------------------------------------------------------------------------------- bucket planning (pure host logic)
class Bucket:
	"""a completion set: the parameters `names`, and the byte ranges of the arena they occupy (one range when the set is
	contiguous; several, merged where they touch, when the arena is laid out in another order than backward finishes it)"""
	__slots__ = ["ranges", "names", "pending", "launched"]

	def __init__(self, ranges, names):
		self.ranges, self.names = [tuple(r) for r in ranges], list(names)
		self.pending, self.launched = set(names), False

	@property
	def start(self):
		return self.ranges[0][0]

	@property
	def stop(self):
		return self.ranges[-1][1]

	@property
	def nbytes(self):
		return sum(hi - lo for lo, hi in self.ranges)


def planBuckets(blocks, bucketBytes):
	"""blocks: ordered [(name, byteOffset, nbytes)] of the flat arena (16-B aligned, Cuda/Utils.py:39-55). Returns
	contiguous buckets [(startByte, stopByte, [names])] of at least `bucketBytes` (except the last), covering the arena
	from 0 to the end of the last block without gaps."""
	buckets, names, start = [], [], 0
	end = 0

	for name, offset, nbytes in blocks:
		names.append(name)
		end = offset + nbytes

		if end - start >= bucketBytes:
			buckets.append((start, end, names))
			names, start = [], end

	if names:
		buckets.append((start, end, names))

	# a bucket's stop is the next bucket's start; gaps from alignment belong to the preceding bucket
	fixed = []
	for i, (bstart, bstop, bnames) in enumerate(buckets):
		nxt = buckets[i + 1][0] if i + 1 < len(buckets) else bstop
		fixed.append((bstart, max(bstop, nxt), bnames))

	return fixed


def planScatteredBuckets(blocks, order, bucketBytes):
	"""Completion-set buckets of an arena that is NOT laid out in completion order (the reference registers its variables in
	sorted-name order, Optimizers/Optimizer.py:66-68): `order` lists the names as backward finishes them; consecutive names
	are collected until a set holds `bucketBytes`, and each set's blocks — scattered over the arena — become byte ranges
	[(lo, hi)], sorted and merged where they touch (an alignment gap of < 16 bytes between two blocks of a set is bridged: it
	holds zeros on every rank). Returns [([(lo, hi), ...], [names])]; every block belongs to exactly one set."""
	where = {name: (offset, nbytes) for name, offset, nbytes in blocks}
	assert sorted(order) == sorted(where), "completion order and arena blocks name different parameters"
	sets, names, size = [], [], 0
	for name in order:
		names.append(name)
		size += where[name][1]
		if size >= bucketBytes:
			sets.append(names)
			names, size = [], 0
	if names:
		sets.append(names)

	out = []
	for names in sets:
		spans = sorted((where[n][0], where[n][0] + where[n][1]) for n in names)
		merged = [list(spans[0])]
		for lo, hi in spans[1:]:
			if lo - merged[-1][1] < 16:
				merged[-1][1] = max(merged[-1][1], hi)
			else:
				merged.append([lo, hi])
		out.append(([tuple(r) for r in merged], names))
	return out


class GradReducer:
	"""Completion-set bucketing of one flat gradient arena. Device work is delegated to `ops`:
	  ops.markReady()                 record 'gradients up to here are final' on the compute stream(s) -> token
	  ops.allreduce(start, stop, tok) queue an in-place sum all-reduce of arena bytes [start, stop) after `tok`
	  ops.allreduceRanges(ranges, tok)   the same for a bucket of several byte ranges (optional: default = one call per range)
	  ops.finish(scale)               make compute wait for all queued collectives, then scale the arena by `scale`
	`order` (names in completion order) makes the buckets completion sets of scattered blocks (planScatteredBuckets); without
	it the arena order IS the completion order and buckets are contiguous."""

	def __init__(self, blocks, ops, gridsize, bucketBytes=25 << 20, order=None):
		self.ops, self.gridsize = ops, gridsize
		if order is None:
			self.buckets = [Bucket([(start, stop)], names) for start, stop, names in planBuckets(blocks, bucketBytes)]
		else:
			self.buckets = [Bucket(ranges, names) for ranges, names in planScatteredBuckets(blocks, order, bucketBytes)]
		self.owner = {name: bucket for bucket in self.buckets for name in bucket.names}
		self.launchedBytes = []       # bytes already handed to the transport after each variableReady (tests, tools)

	def beginStep(self):
		self.launchedBytes = []
		for bucket in self.buckets:
			bucket.pending, bucket.launched = set(bucket.names), False

	def variableReady(self, name):
		bucket = self.owner.get(name, None)
		if bucket is not None and not bucket.launched:
			bucket.pending.discard(name)
			if not bucket.pending and (not hasattr(self.ops, "overlapAllowed") or self.ops.overlapAllowed()):
				self.launch(bucket)
		self.launchedBytes.append(sum(b.nbytes for b in self.buckets if b.launched))

	def launch(self, bucket):
		token = self.ops.markReady()
		if len(bucket.ranges) == 1:
			self.ops.allreduce(bucket.start, bucket.stop, token)
		elif hasattr(self.ops, "allreduceRanges"):
			self.ops.allreduceRanges(bucket.ranges, token)
		else:
			for lo, hi in bucket.ranges:
				self.ops.allreduce(lo, hi, token)
		bucket.launched = True

	def unlaunch(self, bucket):
		"""A bucket that is already with the transport is about to be written again (a second backward pass before the update:
		gradient accumulation). What is in flight becomes the SUM over the ranks; the compute stream waits for it, turns it into
		the MEAN (the same on every rank), the caller's write then adds this rank's new contribution, and the bucket goes out
		again at finishStep: sum_r (mean + d_r) = sum_r g_r + sum_r d_r — what one exchange at the end would have produced, up
		to rounding. The bucket no longer leaves early in this step."""
		assert bucket.launched
		self.ops.unlaunch(bucket.ranges, 1.0 / self.gridsize)
		bucket.launched, bucket.pending = False, {None}          # (nothing reports None: only finishStep launches it)

	def finishStep(self):
		for bucket in self.buckets:
			if not bucket.launched:
				self.launch(bucket)

		self.ops.finish(1.0 / self.gridsize)


# ---------------------------------------------------------------------------------------------- nodeinfo API
class NodeInfo:
	def __init__(self, index, gridsize, device):
		self.index, self.gridsize, self.device = index, gridsize, device

	def close(self):
		pass

	def meanValue(self, value):
		raise NotImplementedError()

	def broadcastBuffer(self, name, buffer):
		raise NotImplementedError()

	def sumTensor(self, name, tensor):
		raise NotImplementedError()


class RcclNodeInfo(NodeInfo):
	transport = "rccl"
	commRanks = 0           # ncclCommCount of the live communicator (0: none was created)
	timeout = float(os.environ.get("PUZZLE_MI355_COMM_TIMEOUT_S", "0"))     # > 0: host-side watchdog on every step's exchange

	def __init__(self, index, gridsize, device, uniqueId, group, bucketBytes=25 << 20):
		super().__init__(index, gridsize, device)
		self.uniqueId, self.group, self.bucketBytes = uniqueId, group, bucketBytes
		self.comm = self.commStream = None
		self.reducers = {}
		self.watchers = {}

	def vote(self, ok):
		"""True iff every rank says yes"""
		return bool(self.group.reduce(1.0 if ok else 0.0, "min")) if self.gridsize > 1 else ok

	def ensureComm(self):
		"""Creates the communicator — after a vote that every rank can: ncclCommInitRank is itself a collective, a rank
		entering it alone would wait forever for peers that already fell back."""
		if self.comm is not None or self.transport != "rccl":
			return

		import ctypes
		from puzzlelib_amd import lib, driver

		reason = None
		try:
			lib.pz_comm_probe()
			if self.uniqueId is None:
				reason = "rank 0 could not create an RCCL id"
		except lib.HipError as e:
			reason = str(e)

		if self.vote(reason is None):
			# every rank enters the collective; a refusal (e.g. two ranks on one device) comes back as an error on all of them
			handle = ctypes.c_void_p()
			try:
				with stdoutToStderr():
					lib.pz_comm_init_rank(ctypes.byref(handle), self.gridsize, self.uniqueId, self.index)
			except lib.HipError as e:
				reason = str(e)
			if self.vote(reason is None):
				self.comm, self.commStream = handle.value, driver.Stream()
				# what RCCL itself reports for the live communicator (bench.py prints it next to the transport)
				nranks, rank = ctypes.c_int(0), ctypes.c_int(0)
				lib.pz_comm_info(self.comm, ctypes.byref(nranks), ctypes.byref(rank))
				self.commRanks = nranks.value
				if (nranks.value, rank.value) != (self.gridsize, self.index):
					raise lib.CommError("RCCL reports rank %d of %d, the grid is rank %d of %d" % (
						rank.value, nranks.value, self.index, self.gridsize))
				return
			if reason is None:
				lib.pz_comm_destroy(handle.value)

		if self.gridsize == 1:
			raise lib.CommError(reason)

		# RCCL is the design; if it cannot be brought up on every rank the run continues on a host-staged exchange
		# (correct, not overlapped, slow) and says so loudly — bench.py reports the transport in its config
		self.transport = "host-staged"
		print("[puzzlelib_amd.grid] rank %d: RCCL unavailable (%s) — falling back to a host-staged all-reduce over TCP; "
			  "expect poor scaling" % (self.index, reason if reason is not None else "failed on another rank"),
			  file=sys.stderr, flush=True)

	def close(self):
		if self.comm is not None:
			from puzzlelib_amd import lib
			lib.pz_comm_destroy(self.comm)
			self.comm = None

	def watcherOf(self, name):
		"""the live ArenaWatcher that follows the arena exchanged under `name` (None: not overlapped by a watcher)"""
		found = [w for (n, _), w in self.watchers.items() if n == name and w is not None and w.root() is not None]
		return found[-1] if found else None

	def commSummary(self):
		"""per-step exposed exchange time and per-bucket bus rates of the overlapped reducer (None: nothing measured)"""
		reducer = self.reducers.get("grad", None)
		stats = getattr(getattr(reducer, "ops", None), "stats", None)
		return None if stats is None else stats.summary()

	def meanValue(self, value):
		return value if self.gridsize == 1 else self.group.reduce(value, "sum") / self.gridsize

	def broadcastBuffer(self, name, buffer):
		from puzzlelib_amd import lib
		self.ensureComm()
		if self.transport == "rccl":
			lib.pz_comm_broadcast(self.comm, buffer.access(True), buffer.size, 0, None)
			return

		host = np.empty(buffer.size, dtype=np.uint8)
		lib.pz_memcpy_d2h(host.ctypes.data, buffer.access(), buffer.size, None)
		lib.pz_stream_sync(None)
		host = np.frombuffer(self.group.broadcast(host.tobytes()), dtype=np.uint8)
		lib.pz_memcpy_h2d(buffer.access(True), host.ctypes.data, buffer.size, None)
		lib.pz_stream_sync(None)

	# ---- gradient exchange
	def attach(self, name, tensor, blocks):
		"""Registers the flat arena `tensor` (1-d fp32 GPUArray) with its parameter blocks for overlapped reduction."""
		self.ensureComm()
		ops = HipReduceOps(self, tensor) if self.transport == "rccl" else HostStagedReduceOps(self, tensor)
		self.reducers[name] = GradReducer(blocks, ops, self.gridsize, self.bucketBytes)
		return self.reducers[name]

	def reduceOps(self, tensor):
		return HipReduceOps(self, tensor) if self.transport == "rccl" else HostStagedReduceOps(self, tensor)

	def plainSum(self, tensor):
		"""no bucket plan: one collective over the whole tensor on the compute stream, then the mean"""
		from puzzlelib_amd import lib
		from puzzlelib_amd.gpuarray import eltwise
		if tensor.dtype != np.float32 or not tensor.contiguous:
			# (the reference exchanges one arena per dtype, Optimizers/Optimizer.py:163-167; this backend computes in float32 only)
			raise NotImplementedError("sumTensor: contiguous float32 tensors only (got %s%s)" % (
				tensor.dtype, "" if tensor.contiguous else ", strided"))
		self.ensureComm()
		if self.transport == "rccl":
			ptr = tensor.wptr
			lib.pz_comm_allreduce_sum_f32(self.comm, ptr, ptr, tensor.size, None)
		else:
			HostStagedReduceOps(self, tensor).allreduce(0, tensor.nbytes, None)
		eltwise(lib.OP_LINEAR, tensor.size, (tensor, tensor), np.array([1.0 / self.gridsize, 0.0], dtype=np.float32))

	def sumTensor(self, name, tensor):
		reducer = self.reducers.get(name, None)

		if reducer is None and AUTO_OVERLAP:
			# Nobody registered a bucket plan (the reference's own Optimizer only ever calls broadcastBuffer / sumTensor,
			# Optimizers/Optimizer.py:107-109,166-167): when the tensor is a flat arena (backend.SharedArray), a watcher on
			# its allocation learns in which order backward finishes its blocks and overlaps the exchange from then on
			# (keyed by name AND allocation: the reference calls sumTensor("grad", ...) once per dtype arena under one name)
			root = tensor.gpudata.root
			key = (name, id(root))
			watcher = self.watchers.get(key, None)
			if watcher is not None and watcher.root() is not root:          # (the id was recycled by a new allocation)
				watcher.detach()
				watcher = None
			if watcher is None and key not in self.watchers:
				for other in [k for k, w in self.watchers.items() if w is not None and w.root() is None]:
					self.watchers.pop(other).detach()                       # arenas that no longer exist
				watcher = self.watchers[key] = ArenaWatcher.attach(self, name, tensor)
			if watcher is not None and watcher.sumTensor():
				return

		if reducer is None:
			self.plainSum(tensor)
			return

		reducer.finishStep()
		reducer.beginStep()


class CommStats:
	"""What one rank saw of its gradient exchange, per step: the EXPOSED time (compute stream idle behind the exchange: from
	the event recorded when the optimizer asks for the mean to the completion of the last bucket's collective) and every
	bucket's own duration on the communication stream. Read one step late (the events have completed by then): no host wait
	is added to the step."""

	def __init__(self, gridsize):
		self.gridsize = gridsize
		self.steps, self.exposedMs, self.bucketMs, self.bucketBytes = 0, 0.0, {}, {}
		self.pending = []             # steps whose events may not have completed yet (the host runs ahead of the device), oldest first

	def stepIssued(self, tail, buckets):
		"""buckets: [(start byte, bytes, begin event, done event)] of the step just issued; tail: event on the compute stream"""
		if buckets:
			self.pending.append((tail, buckets))
		self.collect(wait=len(self.pending) > 64)

	def collect(self, wait=False):
		while self.pending:
			tail, buckets = self.pending[0]
			last = buckets[-1][3]
			if wait:
				last.synchronize()
				tail.synchronize()
			elif not (last.query() and tail.query()):
				return
			self.pending.pop(0)
			self.steps += 1
			self.exposedMs += max(0.0, max(tail.timeTill(done) for _, _, _, done in buckets))
			for start, nbytes, begin, done in buckets:
				self.bucketMs[start] = self.bucketMs.get(start, 0.0) + begin.timeTill(done)
				self.bucketBytes[start] = nbytes

	def summary(self):
		self.collect(wait=True)
		if self.steps == 0:
			return None
		n = self.gridsize
		factor = 2.0 * (n - 1) / n if n > 1 else 1.0          # ring all-reduce: bytes each rank sends + receives per payload byte
		rows = []
		for start in sorted(self.bucketMs):
			ms = self.bucketMs[start] / self.steps
			rows.append({"mbytes": self.bucketBytes[start] / 1e6, "ms": ms,
						 "bus_gb_per_s": factor * self.bucketBytes[start] / (ms * 1e-3) / 1e9 if ms > 0 else None})
		return {"steps_measured": self.steps, "exposed_ms_per_step": self.exposedMs / self.steps, "buckets": rows,
				"note": "exposed = compute stream waiting for the last collective after backward's last kernel; bucket ms = "
						"collective alone on the communication stream (includes waiting for slower ranks); bus GB/s = "
						"2(N-1)/N x bytes / ms (payload rate at N = 1)"}


class HipReduceOps:
	def __init__(self, node, tensor):
		self.node, self.tensor = node, tensor
		self.events = []
		self.stats = CommStats(node.gridsize)

	@staticmethod
	def overlapAllowed():
		"""In the split math modes the convolution kernels issue bf16 MFMAs, next to which packed-fp32 instructions of OTHER waves on
		the SIMD can return wrong low lanes (csrc/Makefile) — and whether RCCL's reduction kernels contain such
		instructions is not known. The library never overlaps its own kernels with a split kernel; the exchange follows the same
		rule: its buckets are queued only at update time, when the compute stream does nothing but wait for them."""
		from puzzlelib_amd.surface import bound
		return bound().backend.dnn.convMath == "f32"

	def markReady(self):
		"""'final up to here' = an event on the compute stream plus one behind the filter-gradient stream, where the
		gradients of this bucket were accumulated (DnnContext.filterGradStream)"""
		from puzzlelib_amd import driver, lazy
		from puzzlelib_amd.surface import bound
		lazy.flushSmall()                             # queued small accumulates into the arena are part of "final"
		event = driver.Event()
		event.record(None)
		return (event, bound().backend.dnn.sideEvent())

	def allreduce(self, start, stop, token):
		from puzzlelib_amd import lib, driver
		node = self.node

		main, side = token
		node.commStream.waitEvent(main)
		if side is not None:
			node.commStream.waitEvent(side)
		ptr = self.tensor.gpudata.ptr + start      # ordering is carried by the two events, not by the arena's barrier
		begin = driver.Event()
		begin.record(node.commStream)
		lib.pz_comm_allreduce_sum_f32(node.comm, ptr, ptr, (stop - start) // 4, node.commStream.handle)

		done = driver.Event()
		done.record(node.commStream)
		self.events.append((token, done, start, stop - start, begin))

	def allreduceRanges(self, ranges, token):
		"""a bucket of several byte ranges of the arena: ONE RCCL group on the communication stream (pz_comm_allreduce_sum_f32_ranges)"""
		import ctypes
		from puzzlelib_amd import lib, driver
		node = self.node

		main, side = token
		node.commStream.waitEvent(main)
		if side is not None:
			node.commStream.waitEvent(side)
		n = len(ranges)
		offsets = (ctypes.c_size_t * n)(*[lo // 4 for lo, _ in ranges])
		counts = (ctypes.c_size_t * n)(*[(hi - lo) // 4 for lo, hi in ranges])
		begin = driver.Event()
		begin.record(node.commStream)
		lib.pz_comm_allreduce_sum_f32_ranges(node.comm, self.tensor.gpudata.ptr, offsets, counts, n, node.commStream.handle)
		done = driver.Event()
		done.record(node.commStream)
		self.events.append((token, done, ranges[0][0], sum(hi - lo for lo, hi in ranges), begin))

	def unlaunch(self, ranges, scale):
		"""GradReducer.unlaunch: the compute stream waits for the collectives over `ranges`, scales them to the mean, and the
		host waits for that — the write that follows may be queued on another stream behind an EARLIER mark of this one."""
		from puzzlelib_amd import lib
		lo, hi = ranges[0][0], ranges[-1][1]
		for _, done, start, _, _ in self.events:
			if lo <= start < hi:
				lib.pz_stream_wait_event(None, done.handle)
		P, F = ctypes_ptrs(2), np.array([scale, 0.0], dtype=np.float32)
		base = self.tensor.gpudata.ptr
		for rlo, rhi in ranges:
			ptrs = P(base + rlo, base + rlo)
			lib.pz_eltwise(lib.OP_LINEAR, (rhi - rlo) // 4, ptrs, 2, F.ctypes.data_as(lib.POINTER(lib.c_float)), 2, 0, (rhi - rlo) // 4, 1, None)
		lib.pz_stream_sync(None)

	def finish(self, scale):
		from puzzlelib_amd import lib, lazy, fusion, driver
		from puzzlelib_amd.gpuarray import eltwise

		lib.pz_comm_async_error(self.node.comm)
		tail = driver.Event()
		tail.record(None)                          # the compute stream has nothing left but to wait for the exchange
		if self.node.timeout > 0.0:
			# the watchdog follows EVERY bucket (a rank that never joined bucket k blocks k, not only the last one)
			for _, done, _, _, _ in self.events:
				lib.pz_comm_wait_event(self.node.comm, done.handle, self.node.timeout)
		for _, done, _, _, _ in self.events:
			lib.pz_stream_wait_event(None, done.handle)
		self.stats.stepIssued(tail, [(start, nbytes, begin, done) for _, done, start, nbytes, begin in self.events])
		self.events = []

		# The mean's 1/N: the arena now holds the SUM. Its division rides in the optimizer's update kernel (the description
		# fusion.Scaled; Grid.py:126-133 divides inside its reduce): no pass of its own over the arena. Anything else that
		# touches the gradients first (a hook, a test reading them) makes the description run as the linear pass it stands for.
		if lazy.on("gradscale") and lazy.whole(self.tensor) and lazy.pending(self.tensor) is None:
			self.tensor.wptr                       # write barrier: version, dependents, foreign readers — the value changes
			lazy.attach(self.tensor, fusion.Scaled(scale))
		else:
			eltwise(lib.OP_LINEAR, self.tensor.size, (self.tensor, self.tensor), np.array([scale, 0.0], dtype=np.float32))


class HostStagedReduceOps:
	"""Fallback transport when RCCL cannot be initialised: device -> host -> TCP star -> device, synchronous.
	Same call protocol as HipReduceOps (GradReducer drives both)."""

	def __init__(self, node, tensor):
		self.node, self.tensor = node, tensor

	def markReady(self):
		return None

	def allreduce(self, start, stop, token):
		from puzzlelib_amd import lib

		ptr = self.tensor.wptr + start          # (the barrier makes the main stream wait for the filter-gradient stream)
		host = np.empty((stop - start) // 4, dtype=np.float32)
		lib.pz_memcpy_d2h(host.ctypes.data, ptr, stop - start, None)
		lib.pz_stream_sync(None)
		self.node.group.sumArray(host)
		lib.pz_memcpy_h2d(ptr, host.ctypes.data, stop - start, None)
		lib.pz_stream_sync(None)

	def unlaunch(self, ranges, scale):
		from puzzlelib_amd import lib
		P, F = ctypes_ptrs(2), np.array([scale, 0.0], dtype=np.float32)
		base = self.tensor.gpudata.ptr
		for rlo, rhi in ranges:
			lib.pz_eltwise(lib.OP_LINEAR, (rhi - rlo) // 4, P(base + rlo, base + rlo), 2, F.ctypes.data_as(lib.POINTER(lib.c_float)), 2,
						   0, (rhi - rlo) // 4, 1, None)
		lib.pz_stream_sync(None)

	def finish(self, scale):
		from puzzlelib_amd import lib
		from puzzlelib_amd.gpuarray import eltwise
		eltwise(lib.OP_LINEAR, self.tensor.size, (self.tensor, self.tensor), np.array([scale, 0.0], dtype=np.float32))


def ctypes_ptrs(n):
	import ctypes
	return ctypes.c_void_p * n


def arenaBlocks(sharedArray):
	"""[(name, byteOffset, nbytes)] of a built SharedArray, in arena order."""
	base = sharedArray.ary.gpudata.ptr
	return [(name, block.gpudata.ptr - base, block.nbytes) for name, block in sharedArray.blocks.items()]


def enableOverlap(optimizer, nodeinfo):
	"""Wires the overlapped reducer into an optimizer in global-state mode: registers the flat fp32 gradient arena and
	has the executor report layers whose parameter gradients are final during backward."""
	reducer = nodeinfo.attach("grad", optimizer.grads.ary, arenaBlocks(optimizer.grads))
	reducer.beginStep()

	def onLayerDone(layer):
		for key in layer.params:
			reducer.variableReady("%s.%s" % (layer.name, key))

	optimizer.net.gradsReady = onLayerDone
	return reducer


# ---------------------------------------------------------------------------------------------- overlap without a patched caller
AUTO_OVERLAP = os.environ.get("PUZZLE_MI355_DP_OVERLAP", "1") == "1"


class ArenaWatcher:
	"""Overlapped gradient exchange for a caller that knows nothing about it — the reference's unpatched Optimizer.

	What the backend sees of a data-parallel training step is: writes into views of one flat allocation (the gradient arena
	of backend.SharedArray — every write goes through a lazy-buffer write barrier, puzzlelib_amd/lazy.py), and one
	`nodeinfo.sumTensor("grad", arena)` per step. The watcher sits on the arena's allocation (`State.watch`) and is told of
	every write barrier BEFORE the write is issued, and (lib.issueWatchers) of every library call AFTER it was queued:
	  * the first sumTensor attaches the watcher; the following steps are only observed: the sequence of blocks written (and
	    of whole-arena writes behind them: hooks) between two sumTensor calls. When two consecutive steps wrote the same
	    sequence, the position of each block's LAST write gives the order in which backward finishes the blocks, whatever
	    order the arena is laid out in (the reference: sorted names, Optimizers/Optimizer.py:66-68) — completion-set buckets
	    of scattered byte ranges (planScatteredBuckets) are planned from it;
	  * from then on a block whose last expected write has been ISSUED is reported to the GradReducer, and a bucket whose
	    blocks are all final is all-reduced at once on the communication stream, behind events of the compute and the
	    filter-gradient stream. "Issued" = a library call carrying an address inside the byte range of that write's barrier
	    was made after the barrier: one launch may take several write addresses first (a filter gradient and its bias
	    gradient: two barriers, then pz_conv2d_bwd_filter), so the next barrier on the arena proves nothing by itself.
	Steps that do not follow the learned pattern are served exactly, without overlap, and the pattern is learned again:
	  * any other write sequence: nothing more goes out early, the rest leaves at sumTensor;
	  * a write into a bucket that is already in flight (a second backward pass before the update): the bucket is taken back
	    (GradReducer.unlaunch) and leaves again at sumTensor;
	  * a write of the WHOLE arena behind block writes is a hook (the reference runs hooks before sumTensor,
	    Optimizer.py:160-167). Only the weight-decay kernel (Optimizers/Hooks.py:16-19: linear in the gradient, reads parameters
	    that are identical on every rank) may trade places with the mean: the exchange completes first, the mean is applied as a
	    pass, the hook then works on mean gradients and the following sumTensor finds nothing left to do. A network whose
	    observed steps contain any OTHER whole-arena write is never overlapped (hook, then one collective at sumTensor: the
	    reference's order). If such a write first appears in an overlapped step, what is already in flight cannot be taken
	    back for a non-linear hook: that one step runs mean-then-hook, says so on stderr, and overlap stays off from then on."""

	HOOK_LINEAR, HOOK_OTHER = -1, -2                 # log entries for whole-arena writes behind block writes

	def __init__(self, node, name, tensor, blocks):
		import weakref
		self.node, self.name, self.tensor = node, name, tensor
		# (tensor None: a host-side rehearsal drives onWrite / onIssue itself, tests/test_dp_gloo.py)
		self.root = weakref.ref(tensor.gpudata.root) if tensor is not None else (lambda: None)
		self.base = tensor.gpudata.ptr if tensor is not None else 0
		self.blocks = sorted(blocks, key=lambda b: b[1])
		self.starts = [b[1] for b in self.blocks]
		self.end = self.blocks[-1][1] + self.blocks[-1][2]
		self.log, self.previous = [], None           # block indices written this step / the step before
		self.sequence = self.last = self.reducer = None
		self.pos, self.armed, self.exact, self.done, self.busy = 0, [], True, False, False
		self.relearn = False                         # this step left the learned pattern in a way that needs a new plan
		self.launchedBytes = []                      # (tests, telemetry) bytes in flight after each write event of the step
		self.steps = 0
		self.listening = False

	@classmethod
	def attach(cls, node, name, tensor):
		from puzzlelib_amd import lazy
		root = tensor.gpudata.root
		lz = lazy.stateOf(root)
		blocks = getattr(lz, "arena", None)
		if not blocks or tensor.gpudata.ptr != root.ptr or tensor.nbytes != root.size or tensor.dtype != np.float32:
			return None                               # not a flat fp32 arena: the plain exchange
		node.ensureComm()
		if node.transport != "rccl":
			return None
		watcher = cls(node, name, tensor, blocks)
		lz.watch = watcher.onWrite
		return watcher

	def detach(self):
		"""the arena was replaced (or is gone): its allocation no longer reports to this watcher"""
		self.listen(False)
		root = self.root()
		if root is not None and root.lz is not None and root.lz.watch == self.onWrite:
			root.lz.watch = None
		self.tensor = None

	# ---- called by lazy.writeBarrier with the byte range about to be written
	def onWrite(self, lo, hi):
		if self.done or self.busy:                    # (the exchange's own writes of the arena: collectives, the mean)
			return
		if lo <= self.blocks[0][1] and hi >= self.end:
			if not self.log:
				return                                # the step's zero fill (or any whole-arena write before backward)
			self.hook()
			return
		import bisect
		first = max(bisect.bisect_right(self.starts, lo) - 1, 0)
		for idx in range(first, len(self.blocks)):
			_, offset, nbytes = self.blocks[idx]
			if offset >= hi:
				break
			if offset + nbytes > lo:
				self.blockWritten(idx, lo, hi)

	def blockWritten(self, idx, lo, hi):
		self.log.append(idx)
		if self.reducer is None:
			return
		self.report()                                 # what was armed by earlier barriers AND has been issued since
		name = self.blocks[idx][0]
		bucket = self.reducer.owner[name]
		if bucket.launched:
			# not the learned step (e.g. a second backward pass before the update): take the bucket back, serve the step without
			# further overlap, learn again
			self.busy = True
			try:
				self.reducer.unlaunch(bucket)
			finally:
				self.busy = False
			self.exact, self.armed, self.relearn = False, [], True
		if self.exact and self.pos < len(self.sequence) and self.sequence[self.pos] == idx:
			if self.last[idx] == self.pos:
				self.armed.append([name, lo, hi, False])
				self.listen(True)
			self.pos += 1
		else:
			self.exact, self.armed = False, []       # a different step: nothing more goes out early
			self.listen(False)
		self.launchedBytes.append(sum(b.nbytes for b in self.reducer.buckets if b.launched))

	# ---- called by the library binding after every queued call while blocks are armed (lib.issueWatchers)
	def listen(self, on):
		from puzzlelib_amd import lib
		if on and not self.listening:
			lib.issueWatchers.append(self.onIssue)
		elif not on and self.listening:
			lib.issueWatchers.remove(self.onIssue)
		self.listening = on

	def onIssue(self, name, args):
		if self.busy or name.startswith("pz_comm_"):  # (the exchange's own calls carry the arena's base address, not a gradient write)
			return
		import ctypes
		base, end = self.base, self.base + self.end
		if name == "pz_memset_d32" and args[0] == base and int(args[2]) * 4 >= self.end:
			return                                    # the step's deferred zero fill, run by the first write barrier: not the write itself
		for arg in args:
			if type(arg) is int:
				if base <= arg < end:
					self.issued(arg - base)
			elif isinstance(arg, ctypes.Array):
				if getattr(arg, "_type_", None) is ctypes.c_void_p:
					for ptr in arg:
						if ptr is not None and base <= ptr < end:
							self.issued(ptr - base)
			elif isinstance(arg, ctypes.c_void_p):
				if arg.value is not None and base <= arg.value < end:
					self.issued(arg.value - base)

	def issued(self, offset):
		waiting = False
		for entry in self.armed:
			if entry[1] <= offset < entry[2]:
				entry[3] = True
			waiting = waiting or not entry[3]
		if not waiting:
			self.listen(False)

	def report(self, everything=False):
		"""hands the armed blocks whose write has been issued (at sumTensor: all of them) to the reducer"""
		if everything:
			from puzzlelib_amd import lazy
			lazy.flushSmall()                         # queued small accumulates are issued now
		ready = [entry for entry in self.armed if entry[3] or everything]
		self.armed = [entry for entry in self.armed if not (entry[3] or everything)]
		if not self.armed:
			self.listen(False)
		for entry in ready:
			self.reducer.variableReady(entry[0])

	def hook(self):
		"""a whole-arena write behind block writes, before sumTensor (see the class comment)"""
		from puzzlelib_amd import lazy, lib
		linear = lazy.writeOp == lib.OP_WEIGHT_DECAY
		self.log.append(self.HOOK_LINEAR if linear else self.HOOK_OTHER)
		if self.reducer is None:
			return                                    # observed steps: hook, then one collective at sumTensor (the reference's order)
		expected = self.exact and self.pos < len(self.sequence) and self.sequence[self.pos] == self.log[-1]
		if linear and expected:
			self.pos += 1
			self.finish(asPass=True)
			self.done = True
			return
		self.exact, self.relearn = False, True
		self.armed = []
		self.listen(False)
		if linear or not any(b.launched for b in self.reducer.buckets):
			# weight decay where none was learned: it may still trade places with the mean (it does in every learned step);
			# nothing in flight: hook first, everything leaves at sumTensor
			if linear:
				self.finish(asPass=True)
				self.done = True
			return
		print("[puzzlelib_amd.grid] rank %d: an unknown kernel wrote the whole gradient arena %r before sumTensor while buckets of "
			  "this step were already being all-reduced: THIS step applies it to the mean gradients (exact only for a hook that "
			  "is linear in the gradient); overlap is off from the next step on" % (self.node.index, self.name), file=sys.stderr, flush=True)
		self.finish(asPass=True)
		self.done = True

	def finish(self, asPass=False):
		from puzzlelib_amd import lazy
		self.busy = True
		try:
			if self.reducer is None:
				self.node.plainSum(self.tensor)
				return
			if self.exact:
				self.report(everything=True)
			if asPass:
				lazy.disabled.add("gradscale")
				try:
					self.reducer.finishStep()
				finally:
					lazy.disabled.discard("gradscale")
			else:
				self.reducer.finishStep()
		finally:
			self.busy = False

	# ---- called by RcclNodeInfo.sumTensor; True = the exchange of this step is complete
	def sumTensor(self):
		if not self.done:
			self.finish()
		self.steps += 1
		self.armed = []
		self.listen(False)
		if self.reducer is not None and (self.relearn or not self.exact or (not self.done and self.pos != len(self.sequence))):
			# the step was not the learned one (served without overlap from the point it differed): learn again
			self.node.reducers.pop("auto:" + self.name, None)
			self.sequence = self.last = self.reducer = None
		# learn: two consecutive steps with the same write sequence fix the plan — unless they contain a whole-arena write of an
		# unknown kernel (only hook-then-mean is exact for those: never overlapped)
		if self.reducer is None and self.log and self.log == self.previous and self.HOOK_OTHER not in self.log:
			self.sequence = list(self.log)
			self.last = {}
			for pos, idx in enumerate(self.sequence):
				if idx >= 0:
					self.last[idx] = pos
			order = [self.blocks[idx][0] for idx in sorted(self.last, key=self.last.get)]
			order += [b[0] for i, b in enumerate(self.blocks) if i not in self.last]        # never written: with the last bucket
			ops = self.node.reduceOps(self.tensor)
			self.reducer = GradReducer(self.blocks, ops, self.node.gridsize, self.node.bucketBytes, order=order)
			self.node.reducers["auto:" + self.name] = self.reducer          # (commSummary finds its telemetry)
		self.previous, self.log = self.log, []
		self.pos, self.exact, self.done, self.relearn = 0, True, False, False
		if self.reducer is not None:
			self.reducer.beginStep()
		return True


# ---------------------------------------------------------------------------------------------- runGrid (Grid.py:4-35)
class GridNode:
	"""What runGrid hands each child process: its place in the grid and where the ranks meet (picklable; the live
	RcclNodeInfo is made inside the child by `connect`). `portCell` (a shared integer made by runGrid): node 0 binds a port
	the system picks and writes it there, the others read it — no port is chosen before somebody holds it."""

	def __init__(self, index, gridsize, device, addr, port, bucketBytes=25 << 20, portCell=None):
		self.index, self.gridsize, self.device, self.addr, self.port, self.bucketBytes = index, gridsize, device, addr, port, bucketBytes
		self.portCell = portCell

	def connect(self):
		return connectNode(self.index, self.gridsize, self.device, self.addr, self.port, self.bucketBytes, portCell=self.portCell)


def generateGridInfo(size, devices=None, portCell=None):
	"""Grid.py:15-22: one node description per process, device i for node i unless `devices` says otherwise"""
	devices = list(range(size)) if devices is None else list(devices)
	assert len(devices) >= size, "runGrid(size=%d) with %d devices" % (size, len(devices))
	port = 0
	if portCell is None:
		# (a caller without a shared cell: a free port found now — another process may take it before node 0 binds it)
		with socket.socket() as s:
			s.bind(("127.0.0.1", 0))
			port = s.getsockname()[1]
	return [GridNode(index, size, devices[index], "127.0.0.1", port, portCell=portCell) for index in range(size)]


def nodeRunner(target, nodeinfo, *args, **kwargs):
	"""Grid.py:25-35: select the node's device BEFORE any backend import, run `target(nodeinfo, ...)`, close the node"""
	from puzzlelib_amd.settings import Config
	configs = [Config]
	try:                                     # inside a PuzzleLib checkout its Config is the one the modules read
		from PuzzleLib import Config as RefConfig
		configs.append(RefConfig)
	except ImportError:
		pass
	for cfg in configs:
		cfg.allowMultiContext = True
		cfg.deviceIdx = nodeinfo.device

	node = nodeinfo.connect() if isinstance(nodeinfo, GridNode) else nodeinfo
	try:
		target(node, *args, **kwargs)
	finally:
		node.close()
		if hostGroup is not None:
			hostGroup.close()


def runGrid(target, size, *args, devices=None, **kwargs):
	"""The reference's launcher with the reference's signature (Grid.py:4-12; TestLib/MultiGPUMnist.py:61 calls
	`runGrid(target=train, size=2, verbose=True)`): one process per device, each running `target(nodeinfo, *args, **kwargs)`
	with a nodeinfo that offers index / gridsize / device / meanValue / broadcastBuffer / sumTensor / close — here over RCCL.
	Children are SPAWNED (the parent may hold a HIP context; a forked copy of it is not usable), so `target` must be
	importable: a module-level function, as in the reference's scripts. A child that dies makes runGrid raise: the others —
	which would wait for it in the host group or inside a collective forever — are terminated (the reference's queue-based
	launcher hangs there, Grid.py:105,119,129)."""
	import multiprocessing
	ctx = multiprocessing.get_context("spawn")
	gridinfo = generateGridInfo(size, devices, portCell=ctx.Value("i", 0))
	nodes = [ctx.Process(target=nodeRunner, args=(target, nodeinfo) + args, kwargs=kwargs) for nodeinfo in gridinfo]
	for node in nodes:
		node.start()
	failed = []
	while not failed and any(node.exitcode is None for node in nodes):
		for node in nodes:
			node.join(0.05)
		failed = [(i, node.exitcode) for i, node in enumerate(nodes) if node.exitcode not in (None, 0)]
	if failed:
		for node in nodes:
			if node.exitcode is None:
				node.terminate()
	for node in nodes:
		node.join()
	if failed:
		raise RuntimeError("runGrid: node(s) %s exited with status %s (the other nodes were stopped)" % (
			[i for i, _ in failed], [c for _, c in failed]))


# ---------------------------------------------------------------------------------------------- process bootstrap
def nodeFromEnv(bucketBytes=25 << 20):
	"""Builds the NodeInfo of this rank from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (set by
	torch.distributed.run or by bench.py's own launcher). Returns None for a single-process run."""
	world = int(os.environ.get("WORLD_SIZE", "1"))
	# PUZZLE_MI355_FORCE_COMM=1: a single process still gets a communicator (one rank) — the whole exchange path (RCCL
	# bring-up, buckets on the communication stream, event joins, 1/N scale) then runs on the one GPU a test box has
	if world == 1 and os.environ.get("PUZZLE_MI355_FORCE_COMM", "0") != "1":
		return None
	os.environ.setdefault("RANK", "0")

	rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", os.environ["RANK"]))
	# PUZZLE_MI355_DEVICE pins every rank to one device: a single-GPU rehearsal of the multi-process path (RCCL itself
	# refuses two ranks on one device unless it is built/configured to allow it)
	local = int(os.environ.get("PUZZLE_MI355_DEVICE", local))

	# MASTER_PORT itself belongs to the launcher's rendezvous store; the host group takes the next port unless told otherwise.
	# PUZZLE_MI355_PORT_FILE (bench.py's own launcher): rank 0 binds a port the system picks and writes it there, the others read
	# it — no port is chosen before somebody holds it
	portFile = os.environ.get("PUZZLE_MI355_PORT_FILE")
	port = int(os.environ.get("PUZZLE_MI355_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 1))
	return connectNode(rank, world, local, os.environ.get("MASTER_ADDR", "127.0.0.1"), port, bucketBytes,
					   portCell=None if portFile is None else FilePortCell(portFile))


class FilePortCell:
	"""the `portCell` of connectNode for ranks that share nothing but the file system: .value reads / writes the port in a file"""

	def __init__(self, path):
		self.path = path

	@property
	def value(self):
		try:
			return int(open(self.path).read().strip() or 0)
		except (OSError, ValueError):
			return 0

	@value.setter
	def value(self, port):
		tmp = "%s.%d" % (self.path, os.getpid())
		with open(tmp, "w") as f:
			f.write(str(int(port)))
		os.replace(tmp, self.path)


def connectNode(rank, world, device, addr, port, bucketBytes=25 << 20, portCell=None):
	"""joins the host group of the grid and returns this rank's RcclNodeInfo (the RCCL id travels over the host group)"""
	global hostGroup
	if portCell is not None and world > 1:
		if rank == 0:
			def publish(actual):
				portCell.value = actual
			hostGroup = HostGroup(rank, world, addr, 0, publish=publish)
		else:
			deadline = time.time() + 120.0
			while portCell.value == 0:
				if time.time() > deadline:
					raise TimeoutError("node 0 of the grid never published its port")
				time.sleep(0.01)
			hostGroup = HostGroup(rank, world, addr, portCell.value)
	else:
		hostGroup = HostGroup(rank, world, addr, port)

	from puzzlelib_amd.settings import Config
	Config.deviceIdx = device
	Config.allowMultiContext = True

	from puzzlelib_amd import lib
	import ctypes

	uid = b""
	if rank == 0:
		try:
			buf = ctypes.create_string_buffer(lib.COMM_ID_BYTES)
			lib.pz_comm_unique_id(buf)
			uid = buf.raw
		except lib.HipError:
			uid = b""              # RcclNodeInfo.ensureComm falls back (on every rank) to the host-staged exchange

	uid = hostGroup.broadcast(uid)
	return RcclNodeInfo(rank, world, device, uid if len(uid) == lib.COMM_ID_BYTES else None, hostGroup, bucketBytes=bucketBytes)


def barrier():
	if hostGroup is not None:
		hostGroup.barrier()


def maxOverRanks(value):
	return value if hostGroup is None else hostGroup.reduce(value, "max")
