"""
CPU tests of the backend's Python side *running* — lazy buffers and fusion decisions, the executor's call sequence, the
data-parallel bucket plan — with the native library in dry-run mode (PUZZLE_MI355_DRYRUN=1: C-ABI calls are recorded, not
executed; host-side queries such as output shapes and kernel-family resolution still come from the real library).
Each check runs in a child process because the mode is fixed when puzzlelib_amd.lib is imported.

The two trace tests are the drop-in evidence: tests/golden/trace_*.json hold the C-ABI call sequences of two training
steps recorded while the REFERENCE's own Modules / Containers / Optimizer / Trainer (imported from /root/reference by
oracle/make_trace.py) ran on this backend object; the build's executor must produce the same calls with the same
descriptors and scalars in the same order.
"""
import os, subprocess, sys

import pytest

from conftest import ROOT

SCRIPT = os.path.join(ROOT, "tests", "dryrun", "checks.py")


def run(name, timeout=300):
	env = dict(os.environ, PUZZLE_MI355_DRYRUN="1", PUZZLE_MI355_LAZY="1", PUZZLE_MI355_CONV_STATS="adaptive")
	env.pop("PUZZLE_MI355_DEBUG_ALLOC", None)
	res = subprocess.run([sys.executable, SCRIPT, name], env=env, capture_output=True, text=True, timeout=timeout)
	assert res.returncode == 0, "%s failed:\n%s\n%s" % (name, res.stdout[-3000:], res.stderr[-6000:])
	return res.stdout


@pytest.mark.parametrize("check", ["trace_lenet", "trace_resnet50"])
def test_executor_sends_what_the_reference_modules_send(check):
	assert "identical" in run(check)


def test_lazy_buffer_barriers():
	assert "OK" in run("lazy_barriers")


def test_fused_paths_taken_and_switchable():
	run("fused_step_counts")


def test_buffers_are_freed_by_reference_counting():
	run("no_leaks_without_gc")


def test_gradient_buckets_leave_progressively_for_resnet50():
	run("dp_bucket_progress")
