"""
Config 5 on whatever the box offers: the data-parallel path with REAL ranks through grid.runGrid (Grid.py:4-35,
TestLib/MultiGPUMnist.py:60-65). On a node with >= 2 (>= 8) devices the tests below start 2 (8) processes, one per device,
and require RCCL to be the transport (ncclCommCount == N) and node 0 to end where a single process ends; on the one-GPU
lease they SKIP — except the rehearsal at the bottom, which runs the very same targets with both nodes pinned to device 0
(RCCL refuses two ranks on one device, so the exchange falls back to the host-staged transport: everything around the
transport — spawn, host group, broadcast, buckets, watcher, hook order — is the code the multi-GPU run uses).

Reference counterparts: Grid.py:103-157 (the star reduce these collectives replace), Optimizers/Optimizer.py:107-109,163-170
(broadcastBuffer at setup, hooks -> sumTensor -> updateVar per step).
"""
import os, subprocess, sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
TESTS = os.path.join(ROOT, "tests")
META = ("transport", "comm_ranks", "exposed_ms", "auto_buckets", "auto_ranges")


def deviceCount():
	from puzzlelib_amd import lib
	count = lib.c_int(0)
	lib.pz_device_count(lib.byref(count))
	return count.value


def cleanEnv(**extra):
	env = dict(os.environ, **extra)
	for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PUZZLE_MI355_DEVICE", "PUZZLE_MI355_FORCE_COMM"):
		if key not in extra:
			env.pop(key, None)
	return env


def single(which, out, *args):
	subprocess.run([sys.executable, os.path.join(TESTS, "dp_targets.py"), which, out] + [str(a) for a in args], check=True,
				   env=cleanEnv(), timeout=600)
	return np.load(out)


def gridRun(target, size, out, devices=None):
	"""grid.runGrid in a child process of its own (the pytest process holds a HIP context; runGrid's parent should not need one)"""
	code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
			"import dp_targets\nfrom puzzlelib_amd import grid\n"
			"grid.runGrid(getattr(dp_targets, %r), %d, %r, devices=%r)\n" % (ROOT, TESTS, target, size, out, devices))
	res = subprocess.run([sys.executable, "-c", code], env=cleanEnv(), capture_output=True, text=True, timeout=900)
	assert res.returncode == 0, "runGrid(%s, size=%d) failed:\n%s\n%s" % (target, size, res.stdout[-2000:], res.stderr[-4000:])
	return np.load(out), res.stdout + res.stderr


def compare(one, many, rtol, atol, what):
	worst = 0.0
	for name in one.files:
		if name in META:
			continue
		a, b = one[name], many[name]
		scale = float(np.abs(a).max()) + 1e-12
		worst = max(worst, float(np.abs(a - b).max()) / scale)
		assert np.allclose(a, b, rtol=rtol, atol=atol * max(scale, 1.0)), "%s: %s differs (max %.3e of scale %.3e)" % (
			what, name, np.abs(a - b).max(), scale)
	return worst


def sizes():
	return [n for n in (2, 8) if deviceCount() >= n]


@pytest.mark.parametrize("size", [2, 8])
def test_runGrid_distinct_shards_equal_the_concatenated_batch(size, tmp_path):
	"""N real ranks over RCCL, every rank on its own shard (LeNet: no batch statistics), MomentumSGD(nodeinfo): node 0's
	parameters after six steps equal a single process that saw all shards of each step as one batch — within fp32 (the
	summation order of the mean differs), not bits."""
	if deviceCount() < size:
		pytest.skip("%d devices visible, %d needed" % (deviceCount(), size))
	one = single("lenetWhole", str(tmp_path / "whole.npz"), size)
	many, _ = gridRun("lenetShards", size, str(tmp_path / "grid.npz"))
	assert str(many["transport"]) == "rccl" and int(many["comm_ranks"]) == size, (many["transport"], many["comm_ranks"])
	worst = compare(one, many, rtol=2e-4, atol=2e-5, what="%d ranks vs one process" % size)
	print("runGrid x%d over RCCL: worst parameter difference %.2e of its scale" % (size, worst))


@pytest.mark.parametrize("size", [2, 8])
def test_runGrid_unpatched_caller_overlaps_through_the_arena_watcher(size, tmp_path):
	"""N real ranks, the reference's own call pattern: sorted-name arena, WeightDecay hook, only sumTensor. The watcher must
	plan scattered completion-set buckets and RCCL must carry them; node 0 ends where the single process ends (identical
	shards: the mean of N equal gradients is that gradient; the hook trades places with the mean -> rounding, not bits)."""
	if deviceCount() < size:
		pytest.skip("%d devices visible, %d needed" % (deviceCount(), size))
	one = single("miniResNetWatched", str(tmp_path / "one.npz"))
	many, log = gridRun("miniResNetWatched", size, str(tmp_path / "grid.npz"))
	assert str(many["transport"]) == "rccl" and int(many["comm_ranks"]) == size
	assert int(many["auto_buckets"]) >= 3 and int(many["auto_ranges"]) > int(many["auto_buckets"]), (many["auto_buckets"], many["auto_ranges"])
	assert "config.comm.exposed_ms_per_step" in log and float(many["exposed_ms"]) >= 0.0
	worst = compare(one, many, rtol=2e-4, atol=2e-5, what="%d ranks (watcher) vs one process" % size)
	print("runGrid x%d, arena watcher over RCCL: worst difference %.2e, exposed %.3f ms/step" % (size, worst, float(many["exposed_ms"])))


def test_rehearsal_two_nodes_on_one_device(tmp_path):
	"""the same targets, the same comparisons, both nodes on device 0 (runs on the one-GPU lease): the transport is the
	host-staged fallback, everything else is what the multi-GPU tests above run"""
	one = single("lenetWhole", str(tmp_path / "whole.npz"), 2)
	many, _ = gridRun("lenetShards", 2, str(tmp_path / "grid.npz"), devices=[0, 0])
	assert str(many["transport"]) in ("rccl", "host-staged")
	worst = compare(one, many, rtol=2e-4, atol=2e-5, what="2 nodes on one device vs one process")

	one = single("miniResNetWatched", str(tmp_path / "one.npz"))
	many, _ = gridRun("miniResNetWatched", 2, str(tmp_path / "grid2.npz"), devices=[0, 0])
	worst2 = compare(one, many, rtol=2e-4, atol=2e-5, what="2 nodes (sorted arena + hook) on one device vs one process")
	print("rehearsal: worst differences %.2e (distinct shards), %.2e (sorted arena + weight decay)" % (worst, worst2))
