"""
Device runtime objects of the MI355X backend: Device, Stream, Event, Buffer, MemoryPool, memcpy2D.

Host-side mirror of the reference "Driver" extension module (Cuda/Source/Core/{Device,Stream,Buffer,Allocator,
Driver}.c; method tables Device.c:160-173, Stream.c:101-239, Buffer.c:526-536, Allocator.c:359-362), implemented
over the C ABI of libpuzzle_mi355.so. Ownership follows the reference: device memory is owned by Buffer objects,
a pool-backed Buffer returns its block to the pool when the last Python reference dies, views (`buffer[a:b]`)
keep their parent alive.
"""
import ctypes, threading, weakref
from ctypes import byref, c_int, c_size_t, c_void_p, c_float

from puzzlelib_amd import lib
from puzzlelib_amd.lib import HipError


def getDriverVersion():
	return lib.pz_version()


class Device:
	def __init__(self, index=0):
		self.index = index


	@staticmethod
	def count():
		n = c_int(0)
		lib.pz_device_count(byref(n))
		return n.value


	def set(self):
		lib.pz_init(self.index)
		Device.current = self.index
		return self


	@staticmethod
	def getCurrent():
		return Device.current


	@staticmethod
	def synchronize():
		lib.pz_device_sync()


	def name(self):
		buf = ctypes.create_string_buffer(256)
		lib.pz_device_name(self.index, buf, 256)
		return buf.value.decode()


	def arch(self):
		buf = ctypes.create_string_buffer(256)
		lib.pz_device_arch(self.index, buf, 256)
		return buf.value.decode()


	def numCUs(self):
		n = c_int(0)
		lib.pz_device_num_cus(self.index, byref(n))
		return n.value


	@staticmethod
	def memoryInfo():
		free, total = c_size_t(0), c_size_t(0)
		lib.pz_device_mem_info(byref(free), byref(total))
		return free.value, total.value


Device.current = 0


class Stream:
	def __init__(self, priority=None):
		"""priority: None = a plain stream; -1 / 0 / +1 = the device's lowest / middle / highest queue priority"""
		handle = c_void_p()
		if priority is None:
			lib.pz_stream_create(byref(handle))
		else:
			lib.pz_stream_create_priority(byref(handle), int(priority))
		self.handle = handle.value


	def synchronize(self):
		lib.pz_stream_sync(self.handle)


	def waitEvent(self, event):
		lib.pz_stream_wait_event(self.handle, event.handle)


	def __del__(self):
		handle, self.handle = getattr(self, "handle", None), None
		if handle is not None:
			try:
				lib.pz_stream_destroy(handle)
			except Exception:
				pass


class Event:
	def __init__(self):
		handle = c_void_p()
		lib.pz_event_create(byref(handle))
		self.handle = handle.value


	def record(self, stream=None):
		lib.pz_event_record(self.handle, streamHandle(stream))


	def synchronize(self):
		lib.pz_event_sync(self.handle)


	def query(self):
		done = c_int(0)
		lib.pz_event_query(self.handle, byref(done))
		return bool(done.value)


	def timeTill(self, end):
		ms = c_float(0.0)
		lib.pz_event_elapsed_ms(self.handle, end.handle, byref(ms))
		return ms.value


	def __del__(self):
		handle, self.handle = getattr(self, "handle", None), None
		if handle is not None:
			try:
				lib.pz_event_destroy(handle)
			except Exception:
				pass


def streamHandle(stream):
	return None if stream is None else stream.handle


class Buffer:
	"""A span of device memory. `parent` is the MemoryPool (or None for a raw allocation) for an owning buffer and
	the sliced Buffer for a view — the same convention the reference uses (Cuda/GPUArray.py:146-154)."""
	__slots__ = ["ptr", "size", "parent", "owner", "base", "lz", "__weakref__"]


	def __init__(self, ptr, size, parent=None, owner=False):
		self.ptr, self.size, self.parent, self.owner = ptr, size, parent, owner
		# `root`: the allocation this span lies in (itself unless it is a view); `lz`: that allocation's lazy state
		# (puzzlelib_amd/lazy.py — pending contents, dependents, foreign-stream events), None for most buffers
		# (an allocation must not point at itself: a reference cycle would keep device memory until the next gc run)
		self.base = parent.root if isinstance(parent, Buffer) else None
		self.lz = None


	@property
	def root(self):
		base = self.base
		return self if base is None else base


	@classmethod
	def allocate(cls, nbytes):
		ptr = c_void_p()
		try:
			lib.pz_malloc(byref(ptr), max(int(nbytes), 1))
		except lib.HipMemoryError:
			# blocks parked in the pools' Python-side front caches count as live for the native pools: hand them back
			# (and let the pools return what they hold to the driver) before giving up
			if not MemoryPool.reclaimAll():
				raise
			lib.pz_malloc(byref(ptr), max(int(nbytes), 1))
		return cls(ptr.value, int(nbytes), parent=None, owner=True)


	def __getitem__(self, item):
		if not isinstance(item, slice) or item.step not in (None, 1):
			raise ValueError("buffer supports contiguous byte slices only")

		start, stop, _ = item.indices(self.size)
		if stop < start:
			raise ValueError("invalid buffer slice")

		return Buffer(self.ptr + start, stop - start, parent=self, owner=False)


	def free(self):
		if self.owner and self.ptr is not None:
			ptr, self.ptr, self.owner = self.ptr, None, False

			if isinstance(self.parent, MemoryPool):
				self.parent.release(ptr, self.size)
			else:
				lib.pz_free(ptr)


	def __del__(self):
		try:
			self.free()
		except Exception:
			pass


	def access(self, write=False, whole=False):
		"""The span's address after the lazy-buffer barriers (lazy.py) for a main-stream read or write."""
		root = self.root
		if root.lz is not None:
			from puzzlelib_amd import lazy
			if write:
				lazy.writeBarrier(root, self, whole and self.size == root.size)
			else:
				lazy.readBarrier(root, self)
		return self.ptr


	def fillD32(self, value, stream=None):
		lib.pz_memset_d32(self.access(True, True), int(value) & 0xffffffff, self.size // 4, streamHandle(stream))
		return self


	def copy(self, dst=None, allocator=None, stream=None):
		if dst is None:
			dst = allocator.allocate(self.size) if allocator is not None else Buffer.allocate(self.size)
		elif dst.size < self.size:
			raise ValueError("destination buffer is too small")

		lib.pz_memcpy_d2d(dst.access(True), self.access(), self.size, streamHandle(stream))
		return dst


	def set(self, hostptr, nbytes, stream=None):
		lib.pz_memcpy_h2d(self.access(True), hostptr, nbytes, streamHandle(stream))


	def get(self, hostptr, nbytes, stream=None):
		lib.pz_memcpy_d2h(hostptr, self.access(), nbytes, streamHandle(stream))
		lib.pz_stream_sync(streamHandle(stream))


	def getIPCHandle(self):
		raise NotImplementedError(
			"IPC memory handles are replaced by RCCL collectives in this backend (see puzzlelib_amd.grid)"
		)


class MemoryPool:
	"""Size-class free list living in the native library (pz_pool_*): allocate / freeHeld / getStats — the reference's
	Driver.MemoryPool (Cuda/Source/Core/Allocator.c:29-75,359-362). Passed around as `allocator=`."""

	# A training step allocates and frees the same few dozen sizes over and over (NiN: 80 tensors per 3 ms step), and each
	# trip into the native pool is two foreign calls. Blocks released by Python are therefore parked HERE first, per size
	# class (the native pool's own classes: 4 per octave, >= 256 B), and handed out again without leaving the interpreter;
	# the native pool sees them as live until `freeHeld` / an allocation failure returns them.
	# Bounds: bytes in all, and blocks per size class (a step's working set re-uses a handful of blocks per class; what is
	# released beyond that goes to the native pool, whose out-of-memory path can return it to the driver). A release can
	# come from any thread (Buffer.__del__ on the input pipeline's worker): the front cache is guarded by a lock.
	frontLimit = 64 << 30
	frontPerClass = 64
	pools = weakref.WeakSet()

	def __init__(self):
		handle = c_void_p()
		lib.pz_pool_create(byref(handle))
		self.handle = handle.value
		self.holding = True
		self.front, self.frontBytes, self.frontBlocks = {}, 0, 0
		# re-entrant: Buffer.__del__ -> release() can run inside allocate() / release() / flushFront() on the SAME thread when a
		# garbage collection starts on one of their allocations and finalises a Buffer of this pool that sat in a reference cycle
		self.lock = threading.RLock()
		MemoryPool.pools.add(self)


	@classmethod
	def reclaimAll(cls):
		"""every pool's parked and held blocks back to the driver; True if there was anything to give back"""
		any_ = False
		for pool in list(cls.pools):
			if pool.handle is not None:
				any_ = any_ or pool.frontBlocks > 0 or pool.getStats()["heldBytes"] > 0
				pool.flushFront()
				lib.pz_pool_free_held(pool.handle)
		return any_


	@staticmethod
	def classSize(n):
		"""pool_class_size of csrc/runtime.hip"""
		if n <= 256:
			return 256
		step = 1 << max((n - 1).bit_length() - 3, 8)
		return (n + step - 1) // step * step


	def allocate(self, nbytes):
		nbytes = int(nbytes)
		cls = self.classSize(nbytes)
		with self.lock:
			parked = self.front.get(cls)
			if parked:
				self.frontBytes -= cls
				self.frontBlocks -= 1
				return Buffer(parked.pop(), nbytes, parent=self, owner=True)
		ptr = c_void_p()
		try:
			lib.pz_pool_alloc(self.handle, cls, byref(ptr))
		except lib.HipError:
			# what is parked here — or in another pool's front cache — may be what the driver needs back
			if not MemoryPool.reclaimAll():
				raise
			lib.pz_pool_alloc(self.handle, cls, byref(ptr))
		return Buffer(ptr.value, nbytes, parent=self, owner=True)


	def release(self, ptr, nbytes=None):
		if self.handle is None:
			return
		if nbytes is not None and self.holding and self.frontBytes < self.frontLimit:
			cls = self.classSize(nbytes)
			with self.lock:
				parked = self.front.get(cls)
				if parked is None:
					parked = self.front[cls] = []
				if len(parked) < self.frontPerClass:
					parked.append(ptr)
					self.frontBytes += cls
					self.frontBlocks += 1
					return
		lib.pz_pool_release(self.handle, ptr)
		if not self.holding:
			lib.pz_pool_free_held(self.handle)


	def flushFront(self):
		with self.lock:
			front, self.front, self.frontBytes, self.frontBlocks = self.front, {}, 0, 0
		for parked in front.values():
			for ptr in parked:
				lib.pz_pool_release(self.handle, ptr)


	def freeHeld(self):
		self.flushFront()
		lib.pz_pool_free_held(self.handle)


	def stopHolding(self):
		self.holding = False
		self.freeHeld()


	def getStats(self):
		vals = [c_size_t(0) for _ in range(4)]
		lib.pz_pool_stats(self.handle, *[byref(v) for v in vals])
		# (blocks parked on the Python side are held, not live, whatever the native pool thinks)
		return {"heldBytes": vals[0].value + self.frontBytes, "liveBytes": vals[1].value - self.frontBytes,
				"heldBlocks": vals[2].value + self.frontBlocks, "liveBlocks": vals[3].value - self.frontBlocks}


def memcpy2D(width, height, src, srcPitch, dst, dstPitch, srcX=0, dstX=0, stream=None):
	"""Pitched device-to-device copy; argument order of Driver.memcpy2D as used by Cuda/GPUBackend.py:296,320."""
	lib.pz_memcpy_2d(dst.access(True) + dstX, dstPitch, src.access() + srcX, srcPitch, width, height, streamHandle(stream))


def allocateFromIPCHandle(handle, size):
	raise NotImplementedError("IPC memory handles are replaced by RCCL collectives in this backend")
