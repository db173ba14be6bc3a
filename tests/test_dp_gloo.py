"""
CPU test of the data-parallel path with world_size 2 over gloo: the bucketed, completion-set driven gradient reducer
(puzzlelib_amd.grid.GradReducer — the same object that drives RCCL on the GPUs) runs in two processes on host arrays,
with torch.distributed (gloo) standing in for the RCCL transport. Checks the reference's arithmetic
(Grid.py:123-135: g <- (g_0 + ... + g_{N-1}) / N on every rank), bucket-by-bucket overlap ordering, the parameter
broadcast and the scalar mean.
"""
import os, socket, sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
	with socket.socket() as s:
		s.bind(("127.0.0.1", 0))
		return s.getsockname()[1]


def _worker(rank, world, port, outdir):
	sys.path.insert(0, ROOT)
	sys.path.insert(0, os.path.join(ROOT, "oracle"))
	os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))

	import torch, torch.distributed as dist
	from puzzlelib_amd import grid

	dist.init_process_group(backend="gloo", rank=rank, world_size=world)

	# flat arena layout like Optimizer.setupGlobalState: sorted names, 16-byte aligned blocks
	sizes = {"a.W": 1000, "a.b": 10, "b.W": 5000, "c.scale": 7, "c.bias": 7, "d.W": 3000}
	blocks, offset = [], 0
	for name in sorted(sizes):
		blocks.append((name, offset, sizes[name] * 4))
		offset += (sizes[name] * 4 + 15) // 16 * 16
	total = offset // 4

	rng = np.random.RandomState(100 + rank)
	arena = np.zeros(total, dtype=np.float32)
	log = []

	class HostOps:
		"""gloo stand-in for HipReduceOps: same call protocol, host memory."""
		def markReady(self):
			return len(log)

		def allreduce(self, start, stop, token):
			view = torch.from_numpy(arena[start // 4:stop // 4])
			dist.all_reduce(view)
			log.append((start, stop))

		def finish(self, scale):
			arena[...] *= np.float32(scale)

	reducer = grid.GradReducer(blocks, HostOps(), world, bucketBytes=8000)
	reducer.beginStep()

	# "backward": variables become final in reverse (execution) order, each written exactly once
	local = {}
	for name, off, nbytes in reversed(blocks):
		g = rng.randn(nbytes // 4).astype(np.float32)
		local[name] = g
		arena[off // 4:off // 4 + nbytes // 4] = g
		reducer.variableReady(name)

	launched_during_backward = len(log)
	reducer.finishStep()

	# parameter broadcast (Optimizers/Optimizer.py:107-109) and scalar mean (Grid.py:104-111) over the same group
	params = torch.from_numpy(rng.randn(64).astype(np.float32))
	dist.broadcast(params, src=0)
	scalar = torch.tensor([float(rank + 1)], dtype=torch.float64)
	dist.all_reduce(scalar)

	np.savez(os.path.join(outdir, "rank%d.npz" % rank), arena=arena, params=params.numpy(), mean=scalar.numpy() / world,
			 launched=np.array([launched_during_backward, len(log)]), **{"g_" + k: v for k, v in local.items()})
	dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_gradient_mean_over_gloo(tmp_path):
	import torch.multiprocessing as mp
	import cpu_ref as R
	from puzzlelib_amd import grid

	world, port = 2, _free_port()
	mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)

	ranks = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]

	# every rank ends with the same arena, equal to the oracle's mean of the per-rank gradients
	assert np.array_equal(ranks[0]["arena"], ranks[1]["arena"])

	sizes = {"a.W": 1000, "a.b": 10, "b.W": 5000, "c.scale": 7, "c.bias": 7, "d.W": 3000}
	offset = 0
	for name in sorted(sizes):
		expected = R.grad_mean_allreduce([r["g_" + name] for r in ranks])
		got = ranks[0]["arena"][offset // 4:offset // 4 + sizes[name]]
		assert np.allclose(got, expected, atol=1e-6), name
		offset += (sizes[name] * 4 + 15) // 16 * 16

	assert np.array_equal(ranks[0]["params"], ranks[1]["params"])
	assert ranks[0]["mean"][0] == ranks[1]["mean"][0] == 1.5

	# overlap: all buckets but (at most) the one holding the first-executed layers were launched during "backward"
	during, total = ranks[0]["launched"]
	assert total >= 2 and during >= total - 1


def _group_worker(rank, world, port, outdir):
	sys.path.insert(0, ROOT)
	from puzzlelib_amd import grid
	group = grid.HostGroup(rank, world, "127.0.0.1", port)
	blob = group.broadcast(b"id-from-rank-0" * 9 if rank == 0 else b"")
	total, low, high = group.reduce(rank + 1.0, "sum"), group.reduce(rank + 1.0, "min"), group.reduce(rank + 1.0, "max")
	array = np.full(1000, float(rank + 1), dtype=np.float32)
	group.sumArray(array)
	group.barrier()
	np.savez(os.path.join(outdir, "g%d.npz" % rank), blob=np.frombuffer(blob, dtype=np.uint8), red=np.array([total, low, high]),
			 array=array)
	group.close()


@pytest.mark.timeout(120)
def test_host_group_over_tcp_three_ranks(tmp_path):
	"""puzzlelib_amd.grid.HostGroup — the plain-TCP star that replaces torch.distributed for the ranks' host-side traffic
	(RCCL id hand-out, votes, scalar means, barriers, the fallback gradient transport)."""
	import multiprocessing as mp

	world, port = 3, _free_port()
	ctx = mp.get_context("spawn")
	procs = [ctx.Process(target=_group_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
	for p in procs:
		p.start()
	for p in procs:
		p.join(90)
		assert p.exitcode == 0

	outs = [np.load(os.path.join(str(tmp_path), "g%d.npz" % r)) for r in range(world)]
	for out in outs:
		assert bytes(out["blob"]) == b"id-from-rank-0" * 9
		assert list(out["red"]) == [6.0, 1.0, 3.0]
		assert np.array_equal(out["array"], np.full(1000, 6.0, np.float32))


WATCH_SIZES = {"bn1.bias": 16, "bn1.scale": 16, "conv1.W": 1200, "conv2.W": 4000, "fc.W": 2600, "fc.b": 10, "bn2.bias": 64, "bn2.scale": 64}
# one inner list = one launch: its write barriers come first, all of them, then the launch is issued (a filter gradient and its
# bias gradient leave in ONE pz_conv2d_bwd_filter call: puzzlelib_amd/dnn.py convNdBackwardParams)
WATCH_LAUNCHES = [["fc.W", "fc.b"], ["bn2.scale", "bn2.bias"], ["conv2.W"], ["bn1.scale", "bn1.bias"], ["conv1.W"]]       # backward's order
# what each step does: "plain" one backward pass; "wd" + the weight-decay hook (a known kernel, linear in the gradient) before
# sumTensor; "acc" two backward passes before the update (gradient accumulation) + the hook; "clip" + an UNKNOWN whole-arena
# kernel that is not linear in the gradient
WATCH_STEPS = ["plain", "plain", "plain", "plain", "wd", "wd", "wd", "wd", "acc", "wd", "wd", "wd", "clip", "clip", "clip", "clip"]


def _watcher_worker(rank, world, port, outdir):
	"""the exchange of a caller that only calls sumTensor (grid.ArenaWatcher) over gloo: an arena in sorted-name order whose blocks
	are finished in another order; write barriers and issued launches are simulated by calling the watcher the way
	lazy.writeBarrier and the library binding do"""
	sys.path.insert(0, ROOT)
	os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
	import torch, torch.distributed as dist
	from puzzlelib_amd import grid, lazy, lib

	dist.init_process_group(backend="gloo", rank=rank, world_size=world)
	sizes = WATCH_SIZES
	blocks, offset = [], 0
	for name in sorted(sizes):
		blocks.append((name, offset, sizes[name] * 4))
		offset += (sizes[name] * 4 + 15) // 16 * 16
	where = {b[0]: (b[1], b[2]) for b in blocks}
	arena = np.zeros(offset // 4, dtype=np.float32)
	log = []

	class HostOps:
		def markReady(self):
			return len(log)

		def allreduce(self, start, stop, token):
			dist.all_reduce(torch.from_numpy(arena[start // 4:stop // 4]))
			log.append(("one", start, stop))

		def allreduceRanges(self, ranges, token):
			for lo, hi in ranges:
				dist.all_reduce(torch.from_numpy(arena[lo // 4:hi // 4]))
			log.append(("group", len(ranges)))

		def unlaunch(self, ranges, scale):
			for lo, hi in ranges:
				arena[lo // 4:hi // 4] *= np.float32(scale)
			log.append(("unlaunch", len(ranges)))

		def finish(self, scale):
			arena[...] *= np.float32(scale)

	class Node:
		gridsize, bucketBytes, reducers, index = world, 6000, {}, rank

		def reduceOps(self, tensor):
			return HostOps()

		def plainSum(self, tensor):
			dist.all_reduce(torch.from_numpy(arena))
			arena[...] *= np.float32(1.0 / world)
			log.append(("plain", ))

	watcher = grid.ArenaWatcher(Node(), "grad", None, blocks)
	watcher.base = base = 0x7f0000001000
	rng = np.random.RandomState(10 + rank)
	result = {}
	for step, kind in enumerate(WATCH_STEPS):
		del log[:]
		watcher.onWrite(0, arena.nbytes)                     # zeroGradParams
		arena[...] = 0
		local = {name: np.zeros(n, np.float32) for name, n in sizes.items()}
		for _ in range(2 if kind == "acc" else 1):
			for launch in WATCH_LAUNCHES:
				for name in launch:
					off, nbytes = where[name]
					watcher.onWrite(off, off + nbytes)           # every barrier of the launch comes BEFORE the launch is issued
				for name in launch:
					off, nbytes = where[name]
					g = rng.randn(nbytes // 4).astype(np.float32)
					local[name] += g
					arena[off // 4:(off + nbytes) // 4] += g     # (accumulate-mode backward: Containers/Sequential.py:212-216)
				watcher.onIssue("launch", tuple(base + where[name][0] for name in launch))
		early = len([e for e in log if e[0] in ("one", "group")])
		if kind in ("wd", "acc"):                            # Optimizers/Hooks.py:16-19 through gpuarray.eltwise
			lazy.writeOp = lib.OP_WEIGHT_DECAY
			watcher.onWrite(0, arena.nbytes)
			lazy.writeOp = None
			arena += np.float32(0.5)
		elif kind == "clip":
			watcher.onWrite(0, arena.nbytes)
			np.clip(arena, -0.25, 0.25, out=arena)
		watcher.sumTensor()
		result["arena%d" % step] = arena.copy()
		result["early%d" % step] = np.array([early, sum(1 for e in log if e[0] == "group"), sum(1 for e in log if e[0] == "plain"),
											  sum(1 for e in log if e[0] == "unlaunch")])
		for name, g in local.items():
			result["g%d_%s" % (step, name)] = g
	np.savez(os.path.join(outdir, "w%d.npz" % rank), **result)
	dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_watcher_overlaps_a_sorted_name_arena_over_gloo(tmp_path):
	"""two ranks, an arena laid out by sorted names (Optimizers/Optimizer.py:66-68) and finished in backward's order: after two
	observed steps the watcher's completion-set buckets — several byte ranges each — leave during "backward", and only behind the
	launch that writes them (two barriers, one launch: the filter gradient's bucket must not leave at the bias gradient's
	barrier); every step, every rank ends with the mean of the ranks' gradients (Grid.py:123-135) whatever the step does:
	the weight-decay hook may trade places with the mean, a second backward pass takes launched buckets back, a hook that is
	not linear in the gradient is served hook-then-mean (the reference's order, Optimizers/Optimizer.py:160-167) without overlap"""
	import torch.multiprocessing as mp
	import cpu_ref as R

	world, port = 2, _free_port()
	mp.spawn(_watcher_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
	ranks = [np.load(os.path.join(str(tmp_path), "w%d.npz" % r)) for r in range(world)]

	sizes = WATCH_SIZES
	modes = []
	for step, kind in enumerate(WATCH_STEPS):
		assert np.array_equal(ranks[0]["arena%d" % step], ranks[1]["arena%d" % step])
		early, groups, plain, unlaunched = ranks[0]["early%d" % step]
		modes.append("observed" if plain else "overlapped")
		offset = 0
		for name in sorted(sizes):
			grads = [r["g%d_%s" % (step, name)] for r in ranks]
			if kind == "clip" and plain:
				expected = R.grad_mean_allreduce([np.clip(g, -0.25, 0.25) for g in grads])       # hook, then mean: exact
			elif kind == "clip":
				expected = np.clip(R.grad_mean_allreduce(grads), -0.25, 0.25)                    # first appearance in an overlapped step
			else:
				expected = R.grad_mean_allreduce(grads) + (np.float32(0.5) if kind in ("wd", "acc") else 0)
			got = ranks[0]["arena%d" % step][offset // 4:offset // 4 + sizes[name]]
			assert np.allclose(got, expected, atol=2e-6), (step, kind, name)
			offset += (sizes[name] * 4 + 15) // 16 * 16
		if not plain and kind != "acc":
			assert early >= 2 and groups >= 1, "step %d: %d buckets left during backward, %d as groups of ranges" % (step + 1, early, groups)
		if kind == "acc":
			assert not plain and unlaunched >= 2, "the second backward pass takes back the buckets the first one launched"
	#           observe x2 -> overlap; an unlearned weight decay and the accumulate step are served and learned again; the
	#           unknown hook appears in an overlapped step once and is never overlapped afterwards
	#           (a served step's write sequence counts as an observation: one more like it fixes the new plan)
	assert modes == ["observed", "observed", "overlapped", "overlapped", "overlapped", "observed", "overlapped", "overlapped",
					 "overlapped", "observed", "observed", "overlapped", "overlapped", "observed", "observed", "observed"], modes
