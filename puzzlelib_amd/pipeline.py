"""
Input-pipeline edge of the training loop: the reference uploads every macro-batch with one synchronous
`gpuarray.to_gpu` from pageable memory before it starts training on it (Handlers/Handler.py:20-36), so the device idles
for the whole transfer. `HostStager` keeps `depth` slots of pinned host memory + device memory and a copy stream:

    host numpy slice --memcpy--> pinned slot --hipMemcpyAsync (copy stream)--> device slot --event--> compute stream

While macro-batch i trains, macro-batch i+1 is staged and copied — by a worker thread, because the staging memcpy
(≈25 ms per 100 MB) would otherwise hold up the Python thread that launches the kernels. Ordering is by events only:
the compute stream waits for a slot's `copied` event before reading it, the copy stream waits for the slot's `consumed` event (recorded on the
compute stream after the last kernel that read it) before overwriting it, and the host waits for `copied` before it
rewrites the pinned half of the slot. Values are untouched — the trained parameters are bit-identical to the
synchronous path (tests/test_gpu_2_boundary.py).
"""
import ctypes
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import lib
from .driver import Stream, Event, Buffer
from .gpuarray import GPUArray


class PinnedBuffer:
	"""Page-locked host memory (pz_host_alloc_pinned) exposed as a numpy byte array."""

	def __init__(self, nbytes):
		ptr = ctypes.c_void_p()
		lib.pz_host_alloc_pinned(ctypes.byref(ptr), max(int(nbytes), 1))
		self.ptr, self.size = ptr.value, int(nbytes)
		self.bytes = np.frombuffer((ctypes.c_ubyte * max(self.size, 1)).from_address(self.ptr), dtype=np.uint8)


	def view(self, shape, dtype):
		nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
		return self.bytes[:nbytes].view(dtype).reshape(shape)


	def __del__(self):
		ptr, self.ptr = getattr(self, "ptr", None), None
		if ptr is not None:
			self.bytes = None
			try:
				lib.pz_host_free_pinned(ptr)
			except Exception:
				pass


class _Leaf:
	__slots__ = ["pinned", "device"]

	def __init__(self):
		self.pinned, self.device = None, None


class _Slot:
	def __init__(self):
		self.leaves = []
		self.copied, self.consumed = Event(), Event()
		self.inFlight = False         # a copy from the pinned half was issued and not yet waited for on the host
		self.everConsumed = False


class HostStager:
	def __init__(self, depth=2, device=None):
		assert depth >= 2
		self.stream = Stream()
		self.slots = [_Slot() for _ in range(depth)]
		self.cursor = 0

		from .driver import Device
		self.device = Device.current if device is None else device
		self.worker = ThreadPoolExecutor(max_workers=1, thread_name_prefix="puzzle-stager")
		self.worker.submit(lib.pz_init, self.device).result()      # the worker thread issues copies on this device


	def close(self):
		self.worker.shutdown(wait=True)


	@staticmethod
	def _flatten(tree, out):
		if isinstance(tree, list):
			for sub in tree:
				HostStager._flatten(sub, out)
		else:
			out.append(tree)
		return out


	@staticmethod
	def _rebuild(tree, leaves):
		if isinstance(tree, list):
			return [HostStager._rebuild(sub, leaves) for sub in tree]
		return next(leaves)


	def submit(self, tree):
		"""Stages the numpy arrays of `tree` (nested lists) and starts their upload on the worker thread; returns a
		ticket for acquire()."""
		slot = self.slots[self.cursor]
		self.cursor = (self.cursor + 1) % len(self.slots)
		return slot, self.worker.submit(self._stage, slot, tree)


	def _stage(self, slot, tree):
		arrays = [np.ascontiguousarray(a) for a in self._flatten(tree, [])]
		while len(slot.leaves) < len(arrays):
			slot.leaves.append(_Leaf())

		if slot.inFlight:                       # the previous copy out of this slot's pinned memory must have finished
			slot.copied.synchronize()
			slot.inFlight = False
		if slot.everConsumed:                   # ... and the kernels reading its device memory too, in stream order
			self.stream.waitEvent(slot.consumed)

		out = []
		for leaf, ary in zip(slot.leaves, arrays):
			if leaf.pinned is None or leaf.pinned.size < ary.nbytes:
				leaf.pinned = PinnedBuffer(ary.nbytes)
			if leaf.device is None or leaf.device.size < ary.nbytes:
				leaf.device = Buffer.allocate(ary.nbytes)

			np.copyto(leaf.pinned.view(ary.shape, ary.dtype), ary)
			lib.pz_memcpy_h2d(leaf.device.ptr, leaf.pinned.ptr, ary.nbytes, self.stream.handle)
			out.append(GPUArray(ary.shape, ary.dtype, gpudata=leaf.device[:ary.nbytes]))

		slot.copied.record(self.stream)
		slot.inFlight = True
		return self._rebuild(tree, iter(out))


	@staticmethod
	def acquire(ticket):
		"""Makes the compute (NULL) stream wait for the upload; returns the tree of GPUArrays."""
		slot, future = ticket
		tree = future.result()                  # the copies are issued (not necessarily finished)
		lib.pz_stream_wait_event(None, slot.copied.handle)
		return tree


	@staticmethod
	def release(ticket):
		"""Call after the last kernel reading the macro-batch was launched."""
		slot, _ = ticket
		from puzzlelib_amd import lazy
		lazy.joinAll()                          # kernels on the filter-gradient stream may still read the macro-batch (conv1)
		slot.consumed.record(None)
		slot.everConsumed = True
