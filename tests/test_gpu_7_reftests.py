"""
The reference's own unit tests, replayed on the MI355X (SURVEY f3's finish line, Unittester.py:114-122).

Each tests/golden/reftests/<Module>.npz is the tape of one `unittest()` of the reference (Modules/Conv2D.py:80-353,
Modules/BatchNorm2D.py:34-82, Optimizers/Optimizer.py:249-325 trainSimpleTest / trainHardTest, Handlers/Trainer.py:38-108, ...),
recorded in the build container by oracle/make_reftests.py while the unmodified reference ran on this repository's backend
object with the C ABI emulated on host buffers: every call the test (through the reference's Modules / Containers / Optimizers)
made on the backend object, every host array it uploaded, and every value it read back — values the reference's own asserts
accepted there. Here the same program runs through the same Python glue on the real library; every value read back must equal
the accepted one within the tape's fp32 tolerance (atol 1e-5, rtol 1e-4: SURVEY.md 8c). Both with the backend's lazy fusion
layer on (its default) and off (one kernel per call).
"""
import glob, json, os

import numpy as np
import pytest

import reftape
from conftest import GOLDEN

TAPES = sorted(glob.glob(os.path.join(GOLDEN, "reftests", "*.npz")))
MANIFEST = json.load(open(os.path.join(GOLDEN, "reftests", "MANIFEST.json"))) if TAPES else {}


def tape_id(path):
	return os.path.basename(path)[:-4]


@pytest.mark.gpu
@pytest.mark.parametrize("fusion", ["fused", "literal"])
@pytest.mark.parametrize("tape", TAPES, ids=tape_id)
def test_reference_unittest_replayed_on_the_device(bnd, tape, fusion):
	from puzzlelib_amd import backend, lazy
	before = (lazy.enabled, set(lazy.disabled))
	lazy.enabled, lazy.disabled = fusion == "fused", set()
	try:
		compared = reftape.replay(tape, lambda initmode: backend.getBackend(0, initmode))
	finally:
		lazy.flushSmall()
		lazy.enabled, lazy.disabled = before
	assert compared == MANIFEST[tape_id(tape)]["values"], "the replay compared %d values, the recording read back %d" % (
		compared, MANIFEST[tape_id(tape)]["values"])


# -m "not gpu": the same replay against the C ABI EMULATED on host buffers (oracle/emu_cabi.py: the header's contract executed with
# the numpy oracle) — the Python glue between the backend object and the C ABI runs with values in the CPU suite too. A sample of the
# tapes (hot-path modules, boundary tests, an optimizer trajectory, a network with audits and generator-drawn inputs, run-time
# compiled kernels); oracle/make_reftests.py --check replays all of them.
CPU_SAMPLE = ["Modules.Conv2D", "Modules.BatchNorm2D", "Modules.Linear", "Modules.MaxPool2D", "Cost.CrossEntropy", "Optimizers.Adam",
			  "Boundary.MIOpen", "Boundary.MatVec", "Boundary.GPUArray", "Boundary.SourceModule", "Hip.Wrappers.MIOpenNorm", "Models.Nets.NiN"]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("name", CPU_SAMPLE)
def test_tape_replays_on_the_emulated_cabi(name):
	import subprocess, sys
	from conftest import ROOT
	path = os.path.join(GOLDEN, "reftests", name + ".npz")
	assert os.path.exists(path), "no committed tape %s" % name
	env = dict(os.environ, PUZZLE_MI355_DRYRUN="1", PYTHONDONTWRITEBYTECODE="1")
	env.pop("PUZZLE_MI355_DEBUG_ALLOC", None)
	res = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "make_reftests.py"), "--replay", path], env=env,
						 capture_output=True, text=True, timeout=580)
	assert res.returncode == 0, res.stderr[-3000:]
	line = next(l for l in res.stdout.splitlines() if l.startswith("REPLAY "))
	assert json.loads(line.split(" ", 1)[1])["compared"] == MANIFEST[name]["values"]


def test_tapes_are_data_and_listed():
	"""-m "not gpu": every committed tape loads, is listed in the manifest with the number of values it checks, and mentions
	nothing but backend attributes, calls, host arrays and scalars"""
	assert len(TAPES) >= 12 and set(MANIFEST) == {tape_id(t) for t in TAPES}
	for path in TAPES:
		header, data = reftape.load(path)
		assert header["name"] == tape_id(path) and header["atol"] <= 1e-5 and header["rtol"] <= 1e-4
		kinds = {op["k"] for op in header["ops"]}
		assert kinds <= {"root", "attr", "setattr", "call", "poke", "del", "audit"}, kinds
		reads = sum(1 for op in header["ops"] if op["k"] in ("attr", "call") and json.dumps(op["r"]).count('"np"') + json.dumps(op["r"]).count('"f"'))
		reads += sum(1 for op in header["ops"] if op["k"] == "audit")       # samples of dropped device arrays (tests that assert nothing)
		assert reads > 0, "%s reads nothing back" % path
		assert reads >= MANIFEST[tape_id(path)]["values"] or "audit" in kinds
		# host inputs drawn from numpy's generator travel as (seed, call list): the calls must be plain data
		for stream in header.get("rng", {}).values():
			assert all(isinstance(call["n"], str) and not call["n"].startswith("_") for call in stream["calls"])
		for key in data.files:
			assert data[key].dtype.kind in "fiub", (path, key, data[key].dtype)
