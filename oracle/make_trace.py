"""
TEST INFRASTRUCTURE (build container only). Runs the REFERENCE's own Python — Models/Nets/ResNet.py on Modules/ and
Containers/, Cost/CrossEntropy.py, Optimizers/Adam.py, Handlers/Trainer.py, imported from /root/reference — on top of
this repository's backend object in dry-run mode (PUZZLE_MI355_DRYRUN=1: no device; every C-ABI call is recorded instead
of executed) and writes the recorded call sequence of two training steps to tests/golden/trace_*.json.

The reference reaches the backend exactly as INTEGRATION.md §2 describes: `PuzzleLib.Hip.Backend` is a module whose
getBackend / getDeviceCount forward to puzzlelib_amd.backend — nothing else of the reference is touched.

tests/test_host_logic.py replays the build's own executor (puzzlelib_amd/engine.py + optim.py) the same way and
requires the identical sequence: same entry points, same order, same descriptors and scalars. That is the evidence that
(a) an unmodified PuzzleLib drops onto this backend, fusions included, and (b) the harness bench.py times sends the
backend what PuzzleLib would.

    PUZZLE_MI355_DRYRUN=1 python oracle/make_trace.py [--check]
"""
import json, os, sys, types

os.environ["PUZZLE_MI355_DRYRUN"] = "1"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np

CASES = {
	# name: (builder of the reference network, input shape, classes)
	"resnet50_b8": (lambda: refResNet(), (8, 3, 224, 224), 1000),
	"lenet_b16": (lambda: refLeNet(), (16, 1, 28, 28), 10),
	# config 3: TestLib/CnnCifar10NIN.py's own buildNet, optimizer and hook (:13-49, :68-70) — Conv2D(bias) -> Activation(relu)
	# out of place, Dropout, both pooling modes
	"nin_b8": (lambda: refNiN(), (8, 3, 32, 32), 10),
	# the data-parallel path of TestLib/MultiGPUMnist.py:6-57: the reference's own MomentumSGD(nodeinfo=...) in global-state mode
	# (Optimizers/Optimizer.py:107-109 broadcastBuffer, :166-167 sumTensor) on a one-rank grid of this backend — five steps, so
	# that the arena's watcher goes from observing to overlapping (puzzlelib_amd/grid.py ArenaWatcher)
	"lenet_dp_b16": (lambda: refLeNet(), (16, 1, 28, 28), 10),
}
SKIP = ("pz_pool_", "pz_event_", "pz_stream_", "pz_malloc", "pz_free", "pz_device_", "pz_init")


def refResNet():
	from PuzzleLib.Models.Nets.ResNet import loadResNet
	net = loadResNet(None, layers="50", actInplace=True, initscheme="none")
	net.pop()                                     # the trailing SoftMax: training uses raw scores (CrossEntropy)
	return net


def refLeNet():
	from PuzzleLib.Models.Nets.LeNet import loadLeNet
	return loadLeNet(None, initscheme="none")


def refNiN():
	# the script pulls in its dataset loader and matplotlib-based visualisation at import time; only buildNet is wanted
	src = open("/root/reference/TestLib/CnnCifar10NIN.py").read()
	head = src[:src.index("def main():")]
	head = "\n".join(l for l in head.splitlines() if "Datasets" not in l and "Visual" not in l)
	mod = types.ModuleType("nin_buildnet")
	exec(compile(head, "CnnCifar10NIN.py (buildNet only)", "exec"), mod.__dict__)
	np.random.seed(1234)
	return mod.buildNet()


def compute(trace):
	"""the recorded calls that do arithmetic or move tensor data, pointers already reduced to given / null"""
	return [[name, list(args)] for name, args in trace if not name.startswith(SKIP)]


def installBackend():
	import refimport
	Config = refimport.setup()
	Config.backend = Config.Backend.hip

	import puzzlelib_amd.backend as ours
	shim = types.ModuleType("PuzzleLib.Hip.Backend")
	shim.getBackend, shim.getDeviceCount = ours.getBackend, ours.getDeviceCount
	sys.modules["PuzzleLib.Hip.Backend"] = shim
	import PuzzleLib.Hip
	PuzzleLib.Hip.Backend = shim
	return Config


def record(case):
	from puzzlelib_amd import lib
	from PuzzleLib.Backend import gpuarray
	from PuzzleLib.Cost.CrossEntropy import CrossEntropy
	from PuzzleLib.Optimizers.Adam import Adam
	from PuzzleLib.Handlers.Trainer import Trainer

	build, shape, classes = CASES[case]
	net = build()
	data = gpuarray.to_gpu(np.zeros(shape, np.float32))
	labels = gpuarray.to_gpu(np.zeros(shape[:1], np.int32))

	nsteps = 2
	if case.startswith("nin"):
		from PuzzleLib.Optimizers.MomentumSGD import MomentumSGD
		from PuzzleLib.Optimizers import Hooks
		optimizer = MomentumSGD(learnRate=0.1, momRate=0.9)
		optimizer.addHook(Hooks.WeightDecay(0.0001))
	elif "_dp_" in case:
		import socket
		from PuzzleLib.Optimizers.MomentumSGD import MomentumSGD
		from puzzlelib_amd import grid
		with socket.socket() as sock:
			sock.bind(("127.0.0.1", 0))
			port = sock.getsockname()[1]
		optimizer = MomentumSGD(learnRate=0.1, momRate=0.9, nodeinfo=grid.connectNode(0, 1, 0, "127.0.0.1", port))
		nsteps = 5
	else:
		optimizer = Adam(alpha=1e-3)
	optimizer.setupOn(net, useGlobalState=True)
	trainer = Trainer(net, CrossEntropy(maxlabels=classes) if case.startswith("nin") else CrossEntropy(), optimizer, batchsize=shape[0])

	steps = []
	lib.trace.clear()
	for _ in range(nsteps):
		trainer.train(data, labels, random=False)
		steps.append(compute(lib.trace))
		lib.trace.clear()
	return steps


def main():
	installBackend()
	check = "--check" in sys.argv
	for case in CASES:
		steps = record(case)
		path = os.path.join(ROOT, "tests", "golden", "trace_%s.json" % case)
		if check:
			assert json.load(open(path))["steps"] == json.loads(json.dumps(steps)), "trace of %s changed" % case
			print("trace %s: unchanged (%s calls)" % (case, " + ".join(str(len(st)) for st in steps)))
		else:
			json.dump({"case": case, "steps": steps}, open(path, "w"), separators=(",", ":"))
			print("wrote %s (%s calls)" % (path, " + ".join(str(len(st)) for st in steps)))


if __name__ == "__main__":
	main()
