import sys, numpy as np
sys.path.insert(0, '/root/repo')
from puzzlelib_amd import backend, lib
bnd = backend.getBackend(0, initmode=2)
G, dnn = bnd.GPUArray, bnd.dnn
rng = np.random.RandomState(0)
n,c,h,w,k,pad = 2,8,6,6,8,1
x = G.toGpu(rng.randn(n,c,h,w).astype(np.float32)); wt = G.toGpu(rng.randn(k,c,3,3).astype(np.float32))
y5 = dnn.convNd(x, wt, None, 1, pad, 1, 1, 5).get(); y3 = dnn.convNd(x, wt, None, 1, pad, 1, 1, 3).get()
d = np.abs(y5-y3).max(axis=1)
np.set_printoptions(precision=2, suppress=True, linewidth=200)
print(d)
