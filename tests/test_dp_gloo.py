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
