"""Host issue time vs wall time of the ResNet-50 b256 training step (is the Python/ctypes launch path keeping up?)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from puzzlelib_amd import nets, optim, lib
from puzzlelib_amd.surface import bound

gpuarray = bound().gpuarray
np.random.seed(1234)
net = nets.loadResNet(None, "50", actInplace=True, initscheme="he")
rng = np.random.RandomState(0)
data = gpuarray.to_gpu(rng.randn(256, 3, 224, 224).astype(np.float32))
labels = gpuarray.to_gpu(rng.randint(0, 1000, size=(256, )).astype(np.int32))
optimizer = optim.Adam(alpha=1e-3)
optimizer.setupOn(net, useGlobalState=True)
trainer = optim.Trainer(net, optim.CrossEntropy(), optimizer, batchsize=256)
net.trainMode()
for _ in range(3):
	trainer.step([data, labels]); net.reset()
lib.pz_device_sync()
steps = 10
t0 = time.perf_counter()
for _ in range(steps):
	trainer.step([data, labels]); net.reset()
t1 = time.perf_counter()
lib.pz_device_sync()
t2 = time.perf_counter()
print("ResNet-50 b256: wall %.2f ms/step, host issue %.2f ms/step" % ((t2 - t0) / steps * 1e3, (t1 - t0) / steps * 1e3))

import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
	trainer.step([data, labels]); net.reset()
pr.disable()
lib.pz_device_sync()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
