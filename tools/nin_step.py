"""Config 3 (BASELINE.json): CIFAR-10 NiN full training step, batch 128, MomentumSGD + weight decay — wall time per step
with the data resident on the device (host launch rate vs GPU time; compare with `rocprofv3 --kernel-trace --stats`)."""
import os, sys, time
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from puzzlelib_amd import nets, optim, lib
from puzzlelib_amd.surface import bound

gpuarray = bound().gpuarray
rng = np.random.RandomState(0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
data = gpuarray.to_gpu(rng.randn(128, 3, 32, 32).astype(np.float32))
labels = gpuarray.to_gpu(rng.randint(0, 10, size=(128, )).astype(np.int32))

np.random.seed(1)
net = nets.buildNiN()
optimizer = optim.MomentumSGD(learnRate=0.01, momRate=0.9)
optimizer.setupOn(net, useGlobalState=True)
trainer = optim.Trainer(net, optim.CrossEntropy(), optimizer, batchsize=128)
net.trainMode()

for _ in range(20):
	trainer.step([data, labels])
	net.reset()
lib.pz_device_sync()
t0 = time.perf_counter()
for _ in range(steps):
	trainer.step([data, labels])
	net.reset()
t1 = time.perf_counter()
lib.pz_device_sync()
t2 = time.perf_counter()
from puzzlelib_amd import lazy
print("filter gradients started from the mark in front of their layer's backward-data: %.1f per step" % (lazy.counters.get("wgrad_early_start", 0) / (steps + 20.0)))
print("NiN b128: %.3f ms/step wall (host issue %.3f ms/step), %.0f img/s" % ((t2 - t0) / steps * 1e3, (t1 - t0) / steps * 1e3, 128 * steps / (t2 - t0)))
