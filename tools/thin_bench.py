import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from puzzlelib_amd import backend, lib, lazy
bnd = backend.getBackend(0, initmode=2); G = bnd.GPUArray
rng = np.random.RandomState(0)
x = G.toGpu(rng.randn(256,3,224,224).astype(np.float32)); W = G.toGpu(rng.randn(64,3,7,7).astype(np.float32))
dy = G.toGpu(rng.randn(256,64,112,112).astype(np.float32))
fn = lambda: bnd.dnn.convNdBackwardData(dy, W, None, x, 2, 3, 1, 0, 1, -1, None, bnd.memoryPool)
secs,_ = bnd.timeKernel(fn, (), looplength=10, log=False, normalize=True)
print(os.environ.get("PUZZLE_MI355_LIB","default")[-16:], "thin dgrad %.3f ms" % (secs*1e3))
