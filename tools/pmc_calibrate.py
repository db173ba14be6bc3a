"""Known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM
section): a 16-B-per-lane stream (dense ReLU), a 4-B-per-lane stream (the `slice=` variant with step 1) and a pure
16-B read (BatchNorm statistics). Each touches 1 GiB per operand so that the 256 MiB Infinity Cache cannot hide it.
Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`; tools/pmc_bench.sh does that and prints the ratios."""
import os, sys
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from puzzlelib_amd.surface import bound
from puzzlelib_amd import lib

surf = bound()
gpuarray = surf.gpuarray
n = 1 << 28                                                   # 1 GiB of float32
x = gpuarray.empty((n, ), dtype=np.float32)
y = gpuarray.empty((n, ), dtype=np.float32)
x.fill(1.0)
lib.pz_device_sync()

relu = surf.ElementWise.reluKer(np.float32)
for _ in range(3):
	relu(y, x)                                                # elt_dense_kernel<OpRelu>: reads 1 GiB, writes 1 GiB
lib.pz_device_sync()
for _ in range(3):
	relu(y, x, slice=slice(0, n, 1))                          # elt_strided_kernel<OpRelu>: same bytes, 4 B per lane
lib.pz_device_sync()

x4 = x.reshape(256, 64, 128, 128)                             # bn_stats_kernel reads 1 GiB and writes ~nothing
bnd = surf.backend if hasattr(surf, "backend") else None
from puzzlelib_amd.backend import getBackend
from puzzlelib_amd.settings import Config
b = getBackend(Config.deviceIdx, 0, None)
c = 64
mean, var = gpuarray.zeros((c, ), dtype=np.float32), gpuarray.zeros((c, ), dtype=np.float32)
scale, bias = gpuarray.zeros((c, ), dtype=np.float32), gpuarray.zeros((c, ), dtype=np.float32)
for _ in range(3):
	b.dnn.batchNormNd(x4, mean, var, scale, bias, 1e-5, 1.0, False, out=y.reshape(x4.shape))
lib.pz_device_sync()
print("each operand: %d bytes" % (4 * n))
