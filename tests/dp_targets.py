"""
Targets for grid.runGrid in the multi-GPU tests (tests/test_gpu_8_multigpu.py): module-level functions, because runGrid
spawns its nodes and a spawned child finds its target by import (the reference's scripts have the same shape:
TestLib/MultiGPUMnist.py:6-57 `train(nodeinfo, verbose)` handed to Grid.runGrid at :60-65).

Every target trains on its node's shard of a fixed synthetic batch and has node 0 write what the test compares: the
parameters after the last step, the transport that carried the gradients, what RCCL says about the communicator.
"""
import os, sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
	sys.path.insert(0, ROOT)

STEPS = 6


def lenetData(total):
	rng = np.random.RandomState(7)
	return rng.randn(total, 1, 28, 28).astype(np.float32), rng.randint(0, 10, size=(total, )).astype(np.int32)


def miniSpec():
	from puzzlelib_amd import nets
	spec = nets.resnet_spec(stages=((8, 1), (16, 2)), classes=10, stem=8, softmax=False)
	return [l if l[0] != "avgpool" else ("avgpool", l[1], 8, 1, 0) for l in spec]


def report(nodeinfo, net, out, extra=None):
	"""node 0 writes the parameters and the facts about the exchange; every node checks what it can check itself"""
	summary = nodeinfo.commSummary() if hasattr(nodeinfo, "commSummary") else None
	if nodeinfo.index == 0:
		np.savez(out, transport=np.array(getattr(nodeinfo, "transport", "single")),
				 comm_ranks=np.array(getattr(nodeinfo, "commRanks", 0)),
				 exposed_ms=np.array(-1.0 if summary is None else summary["exposed_ms_per_step"]),
				 **(extra or {}), **{"p_" + name: var.data.get() for name, var in net.namedParams().items()})
		if summary is not None:
			print("config.comm.exposed_ms_per_step = %.3f over %d steps, %d buckets" % (
				summary["exposed_ms_per_step"], summary["steps_measured"], len(summary["buckets"])), flush=True)


def lenetShards(nodeinfo, out, perNode=32):
	"""TestLib/MultiGPUMnist.py's train(): identical seeds on every node, LeNet (convolutions WITH biases: filter and bias
	gradient leave in one launch), MomentumSGD(nodeinfo=nodeinfo) in global-state mode, every node on ITS OWN shard. Without
	batch normalisation the mean of the shard gradients is the gradient of the concatenated batch (Grid.py:126-133), so the
	test compares node 0's parameters with a single process that trained on all shards at once."""
	from puzzlelib_amd import nets, optim
	from puzzlelib_amd.surface import bound
	gpuarray = bound().gpuarray

	np.random.seed(1234)
	net = nets.loadLeNet(None, initscheme="xavier")
	optimizer = optim.MomentumSGD(learnRate=0.05, momRate=0.9, nodeinfo=nodeinfo)
	optimizer.setupOn(net, useGlobalState=True)
	trainer = optim.Trainer(net, optim.CrossEntropy(maxlabels=10), optimizer, batchsize=perNode)

	size = 1 if nodeinfo is None else nodeinfo.gridsize
	index = 0 if nodeinfo is None else nodeinfo.index
	data, labels = lenetData(perNode * size * STEPS)
	net.trainMode()
	for step in range(STEPS):
		# step s of the single process sees rows [s*B*size, (s+1)*B*size); node i takes the i-th part of exactly those rows
		lo = (step * size + index) * perNode
		trainer.step([gpuarray.to_gpu(data[lo:lo + perNode]), gpuarray.to_gpu(labels[lo:lo + perNode])])
		net.reset()
	if nodeinfo is None:
		np.savez(out, **{"p_" + name: var.data.get() for name, var in net.namedParams().items()})
	else:
		report(nodeinfo, net, out)


def lenetWhole(out, size, perNode=32):
	"""the single process of lenetShards: the same rows, `size` shards per step in one batch"""
	from puzzlelib_amd import nets, optim
	from puzzlelib_amd.surface import bound
	gpuarray = bound().gpuarray

	np.random.seed(1234)
	net = nets.loadLeNet(None, initscheme="xavier")
	optimizer = optim.MomentumSGD(learnRate=0.05, momRate=0.9)
	optimizer.setupOn(net, useGlobalState=True)
	batch = perNode * size
	trainer = optim.Trainer(net, optim.CrossEntropy(maxlabels=10), optimizer, batchsize=batch)
	data, labels = lenetData(batch * STEPS)
	net.trainMode()
	for step in range(STEPS):
		trainer.step([gpuarray.to_gpu(data[step * batch:(step + 1) * batch]), gpuarray.to_gpu(labels[step * batch:(step + 1) * batch])])
		net.reset()
	np.savez(out, **{"p_" + name: var.data.get() for name, var in net.namedParams().items()})


def miniResNetWatched(nodeinfo, out, bucketBytes=8192):
	"""What an UNPATCHED PuzzleLib sends: the gradient arena in sorted-name order (Optimizers/Optimizer.py:66-68), the
	WeightDecay hook in front of sumTensor (Optimizer.py:160-167), nothing but nodeinfo.sumTensor — the overlap comes from the
	arena's watcher (grid.ArenaWatcher). Batch normalisation keeps per-replica statistics (the reference has no synchronised
	BatchNorm), so every node trains on THE SAME batch: the mean gradient then is the single-process gradient, and node 0 must
	end where a single process ends. Different initial parameters per node: the broadcast of node 0's arena must fix that."""
	from puzzlelib_amd import nets, optim
	from puzzlelib_amd.surface import bound
	gpuarray = bound().gpuarray
	golden = np.load(os.path.join(ROOT, "tests", "golden", "miniresnet.npz"))

	index = 0 if nodeinfo is None else nodeinfo.index
	if nodeinfo is not None:
		nodeinfo.bucketBytes = bucketBytes
	optim.Optimizer.arenaLayout = "sorted"
	np.random.seed(7 + 100 * index)
	net = nets.build(miniSpec(), name="mini", initscheme="he", actInplace=True)
	if index == 0:
		for name, var in net.namedParams().items():
			var.data.set(golden["init_" + name])

	optimizer = optim.MomentumSGD(learnRate=0.05, momRate=0.9, nodeinfo=nodeinfo)
	optimizer.addHook(optim.WeightDecay(1e-3))
	optimizer.setupOn(net, useGlobalState=True)
	optimizer.targets[0][0].wc = 1.0
	trainer = optim.Trainer(net, optim.CrossEntropy(), optimizer, batchsize=4)
	data, labels = gpuarray.to_gpu(golden["data"]), gpuarray.to_gpu(golden["labels"])
	net.trainMode()
	for _ in range(STEPS):
		trainer.step([data, labels])
		net.reset()

	if nodeinfo is None:
		np.savez(out, **{"p_" + name: var.data.get() for name, var in net.namedParams().items()})
		return
	watcher = nodeinfo.watcherOf("grad")
	planned = watcher is not None and watcher.reducer is not None
	report(nodeinfo, net, out, extra={
		"auto_buckets": np.array(len(watcher.reducer.buckets) if planned else 0),
		"auto_ranges": np.array(sum(len(b.ranges) for b in watcher.reducer.buckets) if planned else 0),
	})


if __name__ == "__main__":
	# single-process runs of the same bodies (the test starts them as child processes: one HIP context per process)
	which, out = sys.argv[1], sys.argv[2]
	if which == "lenetWhole":
		lenetWhole(out, int(sys.argv[3]))
	elif which == "miniResNetWatched":
		miniResNetWatched(None, out)
	else:
		raise SystemExit("unknown target %s" % which)
