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

	def __init__(self, rank, world, addr, port, timeout=120.0):
		self.rank, self.world = rank, world
		self.peers = []
		if world == 1:
			return

		if rank == 0:
			server = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
			server.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
			server.bind((addr, port))
			server.listen(world)
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
	__slots__ = ["start", "stop", "names", "pending", "launched"]

	def __init__(self, start, stop, names):
		self.start, self.stop, self.names = start, stop, list(names)
		self.pending, self.launched = set(names), False


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


class GradReducer:
	"""Completion-set bucketing of one flat gradient arena. Device work is delegated to `ops`:
	  ops.markReady()                 record 'gradients up to here are final' on the compute stream(s) -> token
	  ops.allreduce(start, stop, tok) queue an in-place sum all-reduce of arena bytes [start, stop) after `tok`
	  ops.finish(scale)               make compute wait for all queued collectives, then scale the arena by `scale`
	"""

	def __init__(self, blocks, ops, gridsize, bucketBytes=25 << 20):
		self.ops, self.gridsize = ops, gridsize
		self.buckets = [Bucket(*b) for b in planBuckets(blocks, bucketBytes)]
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
			if not bucket.pending:
				self.launch(bucket)
		self.launchedBytes.append(sum(b.stop - b.start for b in self.buckets if b.launched))

	def launch(self, bucket):
		token = self.ops.markReady()
		self.ops.allreduce(bucket.start, bucket.stop, token)
		bucket.launched = True

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

	def sumTensor(self, name, tensor):
		reducer = self.reducers.get(name, None)

		if reducer is None:
			# no bucket plan registered: one collective over the whole tensor on the compute stream, then the mean
			from puzzlelib_amd import lib
			from puzzlelib_amd.gpuarray import eltwise
			self.ensureComm()
			if self.transport == "rccl":
				ptr = tensor.wptr
				lib.pz_comm_allreduce_sum_f32(self.comm, ptr, ptr, tensor.size, None)
			else:
				HostStagedReduceOps(self, tensor).allreduce(0, tensor.nbytes, None)
			eltwise(lib.OP_LINEAR, tensor.size, (tensor, tensor), np.array([1.0 / self.gridsize, 0.0], dtype=np.float32))
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

	def finish(self, scale):
		from puzzlelib_amd import lib
		from puzzlelib_amd.gpuarray import eltwise
		eltwise(lib.OP_LINEAR, self.tensor.size, (self.tensor, self.tensor), np.array([scale, 0.0], dtype=np.float32))


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


# ---------------------------------------------------------------------------------------------- process bootstrap
def nodeFromEnv(bucketBytes=25 << 20):
	"""Builds the NodeInfo of this rank from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (set by
	torch.distributed.run or by bench.py's own launcher). Returns None for a single-process run."""
	global hostGroup
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

	# MASTER_PORT itself belongs to the launcher's rendezvous store; the host group takes the next port unless told otherwise
	port = int(os.environ.get("PUZZLE_MI355_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 1))
	hostGroup = HostGroup(rank, world, os.environ.get("MASTER_ADDR", "127.0.0.1"), port)

	from puzzlelib_amd.settings import Config
	Config.deviceIdx = local
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
	return RcclNodeInfo(rank, world, local, uid if len(uid) == lib.COMM_ID_BYTES else None, hostGroup, bucketBytes=bucketBytes)


def barrier():
	if hostGroup is not None:
		hostGroup.barrier()


def maxOverRanks(value):
	return value if hostGroup is None else hostGroup.reduce(value, "max")
