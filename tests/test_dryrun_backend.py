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


@pytest.mark.parametrize("check", ["trace_lenet", "trace_resnet50", "trace_nin", "trace_lenet_dp"])
def test_executor_sends_what_the_reference_modules_send(check):
	assert "identical" in run(check)


def test_lazy_buffer_barriers():
	assert "OK" in run("lazy_barriers")


def test_fused_paths_taken_and_switchable():
	run("fused_step_counts")


def test_backward_data_epilogue_carries_the_batchnorm_backward_sums_when_asked():
	assert "one pass" in run("dgrad_stats_counts")


def test_all_bottleneck_batchnorms_of_resnet50_get_their_backward_sums_from_an_epilogue_when_asked():
	assert "32 backward-data launches" in run("dgrad_stats_resnet50", timeout=600)


def test_epilogue_statistics_give_the_same_gradients_on_the_emulated_cabi():
	assert "within" in run("dgrad_stats_values_on_emulation")


def test_checkpoints_written_by_the_reference_resolve():
	run("reference_checkpoint_names")


def test_buffers_are_freed_by_reference_counting():
	run("no_leaks_without_gc")


def test_gradient_buckets_leave_progressively_for_resnet50():
	run("dp_bucket_progress")


def test_exchange_overlaps_for_a_caller_that_only_calls_sumTensor():
	"""the reference's unpatched Optimizer (sorted-name arena, no hook into backward): completion order learned from the
	arena's own write barriers, scattered completion-set buckets, >= 80 % of the bytes in flight before the last layer"""
	assert "auto overlap" in run("dp_auto_overlap_sorted_arena", timeout=600)


def test_exchange_completes_before_a_hook_that_writes_the_whole_arena():
	assert "then the hook" in run("dp_auto_overlap_with_hook")


def test_a_bucket_never_leaves_before_the_launch_that_writes_it_was_issued():
	"""two write barriers, then one launch (filter gradient + fused bias gradient): the filter's bucket waits for the launch"""
	assert "none before its writer" in run("dp_auto_overlap_conv_bias")


@pytest.mark.timeout(300)
@pytest.mark.parametrize("size", [2, 8])
def test_runGrid_runs_a_MultiGPUMnist_shaped_target(size):
	"""grid.runGrid(target, size, *args, devices=None, **kwargs) / nodeRunner with the reference's signatures (Grid.py:4-35):
	the body of TestLib/MultiGPUMnist.py's train(nodeinfo, verbose) on synthetic data, one spawned process per node"""
	out = run("run_grid_%d" % size, timeout=280)
	assert "runGrid OK" in out and out.count(" of %d on device" % size) == size


@pytest.mark.timeout(300)
@pytest.mark.parametrize("ranks", [2, 8])
def test_bench_starts_its_own_ranks_and_runs_the_data_parallel_path(ranks):
	"""`python bench.py --gpus N` (N = 2, and the driver's full node: 8) without a launcher (the shape of the driver's command): bench.py spawns the ranks itself,
	they find each other over the TCP host group, exchange the RCCL id, vote, create the communicator, broadcast the
	parameters and run the bucketed, overlapped gradient exchange — all of it in dry-run mode here (both ranks pinned to the
	one simulated device), so every host-side step of the multi-GPU path runs before hardware sees it."""
	import json
	env = dict(os.environ, PUZZLE_MI355_DRYRUN="1", PUZZLE_MI355_DEVICE="0")
	for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
		env.pop(key, None)
	res = subprocess.run(
		[sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--batch", "2",
		 "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True, timeout=280
	)
	assert res.returncode == 0, res.stderr[-4000:]
	lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
	assert len(lines) == 1, "exactly one JSON line (rank 0's)"
	out = json.loads(lines[0])
	assert out["n_gpus"] == ranks and out["config"]["global_batch"] == 2 * ranks and out["scaling"] == "weak"
	assert out["config"]["rccl_nranks"] == ranks and "comm" in out["config"] and "tolerance" in out
	assert out["config"]["grad_allreduce"].startswith("RCCL")
	# the N = 1 point of the same run: rank 0 alone, before the N-rank loop
	assert out["single_gpu_same_run"]["steps"] == 2 and out["single_gpu_same_run"]["unit"] == "images/sec"
	assert all("frac" in fam for fam in out["conv_kernel_families"])
