"""
TEST INFRASTRUCTURE (build container only) — makes the read-only Python reference importable as the
package `PuzzleLib` with its numpy CPU backend selected. Never used on the GPU box: /root/reference
does not exist there and nothing under tests/ -m gpu, smoke() or bench.py imports this file.

Recipe: a scratch directory holding a symlink PuzzleLib -> /root/reference goes on
sys.path, an h5py stub satisfies the top-level imports, bytecode writing is disabled because the
reference tree is read-only, and Config.backend is set to cpu BEFORE any other PuzzleLib import.
"""
import os, sys, tempfile

REFERENCE = "/root/reference"


def available():
	return os.path.isdir(REFERENCE)


def setup():
	if not available():
		raise RuntimeError("reference tree %s not present (this only works in the build container)" % REFERENCE)

	sys.dont_write_bytecode = True

	root = os.path.join(tempfile.gettempdir(), "puzzle_ref_import")
	os.makedirs(root, exist_ok=True)

	link = os.path.join(root, "PuzzleLib")
	if not os.path.islink(link):
		os.symlink(REFERENCE, link)

	stubs = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_stubs")
	for p in (stubs, root):
		if p not in sys.path:
			sys.path.insert(0, p)

	from PuzzleLib import Config
	Config.backend = Config.Backend.cpu
	Config.showWarnings = False
	return Config
