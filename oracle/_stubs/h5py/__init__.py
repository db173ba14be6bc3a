"""
Import-time stand-in for h5py, used ONLY by oracle/make_golden.py in the build container so that the
Python reference (which imports h5py at module top level: Modules/Module.py:5-6, Containers/Container.py:4,
Optimizers/Optimizer.py:4) can be imported. No save/load path is ever exercised through it.
"""


class _Unavailable:
	def __getattr__(self, item):
		raise RuntimeError("h5py is not installed; checkpoint IO is out of scope for the oracle")


h5p = _Unavailable()
h5f = _Unavailable()


class File:
	def __init__(self, *args, **kwargs):
		raise RuntimeError("h5py is not installed")


def special_dtype(**kwargs):
	raise RuntimeError("h5py is not installed")
