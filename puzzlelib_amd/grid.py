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

and the gradient exchange is bucketed and overlapped with backward: the flat arena is cut into buckets of ~25 MB
(contiguous ranges, i.e. sets of variables); a module-level hook marks variables complete as backward produces them,
and as soon as a bucket's completion set is full its all-reduce is queued on a dedicated communication stream behind
an event recorded on the compute stream. `sumTensor` at update time only queues what is still missing, makes the
compute stream wait for the communication stream and scales by 1/N.

Processes are started by `python -m torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT);
torch.distributed's gloo group is used ONLY to hand the 128-byte RCCL unique id to the other ranks and for host scalars.
"""
import os

import numpy as np


# ---------------------------------------------------------------------------------------------- bucket planning (pure host logic)
class Bucket:
	__slots__ = ["start", "stop", "names", "pending", "launched"]

	def __init__(self, start, stop, names):
		self.start, self.stop, self.names = start, stop, list(names)
		self.pending, self.launched = set(names), False


def planBuckets(blocks, bucketBytes):
	"""blocks: ordered [(name, byteOffset, nbytes)] of the flat arena (registration order = sorted names, 16-B aligned,
	Cuda/Utils.py:39-55). Returns contiguous buckets [(startByte, stopByte, [names])] of at least `bucketBytes`
	(except the last), covering the arena from 0 to the end of the last block without gaps."""
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
	  ops.markReady()                 record 'gradients up to here are final' on the compute stream -> token
	  ops.allreduce(start, stop, tok) queue an in-place sum all-reduce of arena bytes [start, stop) after `tok`
	  ops.finish(scale)               make compute wait for all queued collectives, then scale the arena by `scale`
	"""

	def __init__(self, blocks, ops, gridsize, bucketBytes=25 << 20):
		self.ops, self.gridsize = ops, gridsize
		self.buckets = [Bucket(*b) for b in planBuckets(blocks, bucketBytes)]
		self.owner = {name: bucket for bucket in self.buckets for name in bucket.names}


	def beginStep(self):
		for bucket in self.buckets:
			bucket.pending, bucket.launched = set(bucket.names), False


	def variableReady(self, name):
		bucket = self.owner.get(name, None)
		if bucket is None or bucket.launched:
			return

		bucket.pending.discard(name)
		if not bucket.pending:
			self.launch(bucket)


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
	def __init__(self, index, gridsize, device, uniqueId, hostGroup=None, bucketBytes=25 << 20):
		super().__init__(index, gridsize, device)
		self.uniqueId, self.hostGroup, self.bucketBytes = uniqueId, hostGroup, bucketBytes

		self.comm = None
		self.commStream = None
		self.reducers = {}
		self.lastEvents = []


	# ---- lazy device side (the backend must be bound to Config.deviceIdx == self.device first)
	transport = "rccl"

	def allRanksOk(self, ok):
		"""True iff every rank reports success (host all-reduce over the bootstrap group)."""
		if self.gridsize == 1:
			return ok

		import torch, torch.distributed as dist
		if not (dist.is_available() and dist.is_initialized()):
			return ok
		flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
		dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.hostGroup)
		return bool(flag.item())


	def ensureComm(self):
		if self.comm is not None or self.transport != "rccl":
			return

		import ctypes, sys
		from puzzlelib_amd import lib, driver

		handle, error = ctypes.c_void_p(), None
		try:
			if self.uniqueId is None:
				raise lib.CommError("no RCCL unique id (librccl could not be loaded on some rank)")
			lib.pz_comm_init_rank(ctypes.byref(handle), self.gridsize, self.uniqueId, self.index)
		except lib.HipError as e:
			error = e

		if self.allRanksOk(error is None):
			self.comm = handle.value
			self.commStream = driver.Stream()
			return

		if self.gridsize == 1 and error is not None:
			raise error

		# RCCL is the design; if it cannot be brought up on every rank the run continues on a host-staged gloo exchange
		# (correct, not overlapped, slow) and says so loudly — bench.py reports the transport in its config
		if error is None and handle.value:
			lib.pz_comm_destroy(handle.value)
		self.transport = "gloo-host-staged"
		print("[puzzlelib_amd.grid] rank %d: RCCL communicator unavailable (%s) — falling back to a host-staged gloo "
			  "all-reduce; expect poor scaling" % (self.index, error if error is not None else "failed on another rank"),
			  file=sys.stderr, flush=True)


	def close(self):
		if self.comm is not None:
			from puzzlelib_amd import lib
			lib.pz_comm_destroy(self.comm)
			self.comm = None


	def meanValue(self, value):
		if self.gridsize == 1:
			return value

		import torch, torch.distributed as dist
		t = torch.tensor([float(value)], dtype=torch.float64)
		dist.all_reduce(t, group=self.hostGroup)
		return t.item() / self.gridsize


	def broadcastBuffer(self, name, buffer):
		from puzzlelib_amd import lib
		self.ensureComm()
		if self.transport == "rccl":
			lib.pz_comm_broadcast(self.comm, buffer.ptr, buffer.size, 0, None)
			return

		import torch, torch.distributed as dist
		host = np.empty(buffer.size, dtype=np.uint8)
		lib.pz_memcpy_d2h(host.ctypes.data, buffer.ptr, buffer.size, None)
		lib.pz_stream_sync(None)
		dist.broadcast(torch.from_numpy(host), src=0, group=self.hostGroup)
		lib.pz_memcpy_h2d(buffer.ptr, host.ctypes.data, buffer.size, None)
		lib.pz_stream_sync(None)


	# ---- gradient exchange
	def attach(self, name, tensor, blocks):
		"""Registers the flat arena `tensor` (1-d fp32 GPUArray) with its variable blocks for overlapped reduction."""
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
				lib.pz_comm_allreduce_sum_f32(self.comm, tensor.ptr, tensor.ptr, tensor.size, None)
			else:
				HostStagedReduceOps(self, tensor).allreduce(0, tensor.nbytes, None)
			eltwise(lib.OP_LINEAR, tensor.size, (tensor, tensor), np.array([1.0 / self.gridsize, 0.0], dtype=np.float32))
			return

		reducer.finishStep()
		reducer.beginStep()


class HipReduceOps:
	def __init__(self, node, tensor):
		self.node, self.tensor = node, tensor
		self.events = []


	def markReady(self):
		from puzzlelib_amd import driver
		from puzzlelib_amd.surface import bound
		event = driver.Event()
		event.record(None)
		# filter gradients may have been queued on the backend's side stream (DnnContext.overlapFilterGrad)
		return (event, bound().Dnn.filterGradEvent())


	def allreduce(self, start, stop, token):
		from puzzlelib_amd import lib, driver
		node = self.node

		main, side = token
		node.commStream.waitEvent(main)
		if side is not None:
			node.commStream.waitEvent(side)
		ptr = self.tensor.ptr + start
		lib.pz_comm_allreduce_sum_f32(node.comm, ptr, ptr, (stop - start) // 4, node.commStream.handle)

		done = driver.Event()
		done.record(node.commStream)
		self.events.append((token, done))


	def finish(self, scale):
		from puzzlelib_amd import lib
		from puzzlelib_amd.gpuarray import eltwise

		for _, done in self.events:
			lib.pz_stream_wait_event(None, done.handle)
		self.events = []

		eltwise(lib.OP_LINEAR, self.tensor.size, (self.tensor, self.tensor), np.array([scale, 0.0], dtype=np.float32))


class HostStagedReduceOps:
	"""Fallback transport when RCCL cannot be initialised: device -> host -> gloo all-reduce -> device, synchronous.
	Same call protocol as HipReduceOps (GradReducer drives both)."""

	def __init__(self, node, tensor):
		self.node, self.tensor = node, tensor


	def markReady(self):
		from puzzlelib_amd import driver
		from puzzlelib_amd.surface import bound
		bound().Dnn.joinFilterGrads()               # (synchronous transport: no point in keeping the side stream apart)
		event = driver.Event()
		event.record(None)
		return event


	def allreduce(self, start, stop, token):
		import torch, torch.distributed as dist
		from puzzlelib_amd import lib

		if token is not None:
			token.synchronize()
		host = np.empty((stop - start) // 4, dtype=np.float32)
		lib.pz_memcpy_d2h(host.ctypes.data, self.tensor.ptr + start, stop - start, None)
		lib.pz_stream_sync(None)
		dist.all_reduce(torch.from_numpy(host), group=self.node.hostGroup)
		lib.pz_memcpy_h2d(self.tensor.ptr + start, host.ctypes.data, stop - start, None)
		lib.pz_stream_sync(None)


	def finish(self, scale):
		from puzzlelib_amd import lib
		from puzzlelib_amd.gpuarray import eltwise
		eltwise(lib.OP_LINEAR, self.tensor.size, (self.tensor, self.tensor), np.array([scale, 0.0], dtype=np.float32))


def arenaBlocks(sharedArray):
	"""[(name, byteOffset, nbytes)] of a built SharedArray, in arena order."""
	base = sharedArray.ary.ptr
	return [(name, block.ptr - base, block.nbytes) for name, block in sharedArray.blocks.items()]


def enableOverlap(optimizer, nodeinfo):
	"""Wires the overlapped reducer into an optimizer in global-state mode: registers the flat fp32 gradient arena and
	installs the module hook that reports finished variables during backward."""
	from puzzlelib_amd import nn

	shGrads = optimizer.shGrads[np.float32]
	reducer = nodeinfo.attach("grad", shGrads.ary, arenaBlocks(shGrads))
	reducer.beginStep()

	# variable object -> arena name
	names = {}
	for var, varnames in optimizer.module.getVarTable().items():
		names[id(var)] = varnames[0]

	def onParamGrads(module):
		for var in module.vars.values():
			name = names.get(id(var), None)
			if name is not None:
				reducer.variableReady(name)

	nn.Module.paramGradsHook = staticmethod(onParamGrads)
	return reducer


# ---------------------------------------------------------------------------------------------- process bootstrap
def nodeFromEnv(bucketBytes=25 << 20):
	"""Builds the NodeInfo of this rank from the torchrun environment. Returns None for a single-process run."""
	world = int(os.environ.get("WORLD_SIZE", "1"))
	if world == 1:
		return None

	rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", os.environ["RANK"]))
	# PUZZLE_MI355_DEVICE pins every rank to one device: a single-GPU rehearsal of the multi-process path (RCCL itself
	# refuses two ranks on one device unless it is built/configured to allow it)
	local = int(os.environ.get("PUZZLE_MI355_DEVICE", local))

	import torch.distributed as dist
	if not dist.is_initialized():
		dist.init_process_group(backend="gloo")

	from puzzlelib_amd.settings import Config
	Config.deviceIdx = local
	Config.allowMultiContext = True

	from puzzlelib_amd import lib
	import ctypes

	ids = [None]
	if rank == 0:
		try:
			buf = ctypes.create_string_buffer(lib.COMM_ID_BYTES)
			lib.pz_comm_unique_id(buf)
			ids = [buf.raw]
		except lib.HipError:
			ids = [None]            # RcclNodeInfo.ensureComm falls back (on every rank) to the host-staged exchange

	dist.broadcast_object_list(ids, src=0)
	return RcclNodeInfo(rank, world, local, ids[0], bucketBytes=bucketBytes)


def barrier():
	import torch.distributed as dist
	if dist.is_available() and dist.is_initialized():
		dist.barrier()


def maxOverRanks(value):
	import torch, torch.distributed as dist
	if not (dist.is_available() and dist.is_initialized()):
		return value
	t = torch.tensor([float(value)], dtype=torch.float64)
	dist.all_reduce(t, op=dist.ReduceOp.MAX)
	return t.item()
