"""Input-pipeline edge: wall time of Trainer.trainFromHost on a CIFAR-10-sized host array (NiN, batch 128,
macro-batches of 10000 images) with the reference's synchronous per-macro-batch upload vs the staged asynchronous one."""
import os, sys, time
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from puzzlelib_amd import nets, optim, lib
from puzzlelib_amd.surface import bound

bound()
rng = np.random.RandomState(0)
which = sys.argv[1] if len(sys.argv) > 1 else "nin"
if which == "nin":
	n, shape, classes, batch, macro = 50000, (3, 32, 32), 10, 128, 10000
else:                                                         # ResNet-50, 8 batches of 256 in macro-batches of 2 batches
	n, shape, classes, batch, macro = 2048, (3, 224, 224), 1000, 256, 512
data = rng.randn(n, *shape).astype(np.float32)
labels = rng.randint(0, classes, size=(n, )).astype(np.int32)

for mode in (False, True, False, True):
	optim.Loop.asyncUpload = mode
	np.random.seed(1)
	net = nets.buildNiN() if which == "nin" else nets.loadResNet(None, "50", actInplace=True, initscheme="he")
	optimizer = optim.MomentumSGD(learnRate=0.01, momRate=0.9)
	optimizer.setupOn(net, useGlobalState=True)
	trainer = optim.Trainer(net, optim.CrossEntropy(), optimizer, batchsize=batch)
	trainer.trainFromHost(data[:2 * macro], labels[:2 * macro], macroBatchSize=macro, random=False)      # warm-up (sizes the staging slots)
	lib.pz_device_sync()
	t0 = time.perf_counter()
	trainer.trainFromHost(data, labels, macroBatchSize=macro, random=False)
	lib.pz_device_sync()
	dt = time.perf_counter() - t0
	print("asyncUpload=%-5s  %.3f s per epoch of %d images (%.0f img/s), mean error %.4f" % (mode, dt, n, n / dt, trainer.cost.getMeanError()))
