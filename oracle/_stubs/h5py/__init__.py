"""
Stand-in for h5py in the build container (the image has none), used ONLY by the oracle-side scripts that import the Python
reference — which imports h5py at module top level (Modules/Module.py:5-6, Containers/Container.py:4,
Optimizers/Optimizer.py:4). Files are trees of numpy arrays kept IN MEMORY under their absolute path (an empty file is left on
disk so that a test's os.remove finds it): enough for the reference's own Module.save / Module.load round trips inside one
process (Modules/Module.py:179-283 — oracle/make_reftests.py runs unit tests that end with one), nothing more. No byte of
HDF5 is read or written; puzzlelib_amd/checkpoint.py's fixtures come from oracle/make_checkpoint_fixture.py.
"""
import os

import numpy as np

FILES = {}          # absolute path -> File tree


class _Unavailable:
	def __getattr__(self, item):
		raise RuntimeError("h5py is not installed; file images are out of reach of the in-memory stand-in")


h5p = _Unavailable()
h5f = _Unavailable()


class Group(dict):
	def require_group(self, name):
		return self.setdefault(name, Group())

	create_group = require_group

	def create_dataset(self, name, shape=None, dtype=None, data=None, compression=None, **kwargs):
		dict.__setitem__(self, name, np.array(data) if data is not None else np.zeros(shape, dtype=dtype))
		return self[name]

	def __setitem__(self, name, value):
		dict.__setitem__(self, name, value if isinstance(value, Group) else np.array(value))


class File(Group):
	def __init__(self, name, mode="r", **kwargs):
		super().__init__()
		if not isinstance(name, str):
			raise RuntimeError("h5py stand-in: only named files")
		self.filename = os.path.abspath(name)
		if mode.startswith("w"):
			FILES[self.filename] = self
			open(self.filename, "wb").close()
		else:
			if self.filename not in FILES:
				raise OSError("h5py stand-in: %s was not written in this process" % self.filename)
			self.update(FILES[self.filename])

	@property
	def id(self):
		return _Unavailable()

	def flush(self):
		pass

	def close(self):
		pass

	def __enter__(self):
		return self

	def __exit__(self, *exc):
		return False


def special_dtype(**kwargs):
	return object
