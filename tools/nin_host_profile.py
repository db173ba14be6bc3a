"""Config 3 (NiN b128): where the HOST spends a training step (cProfile of 200 steps, data resident). The step is
host-bound (issue time ~= wall time in tools/nin_step.py), so this is the profile that matters for it.
    python tools/nin_host_profile.py [rows]"""
import cProfile, os, pstats, sys
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from puzzlelib_amd import nets, optim, lib
from puzzlelib_amd.surface import bound

gpuarray = bound().gpuarray
rng = np.random.RandomState(0)
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
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
	trainer.step([data, labels])
	net.reset()
pr.disable()
lib.pz_device_sync()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 40)
st.sort_stats("cumtime").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 40)
