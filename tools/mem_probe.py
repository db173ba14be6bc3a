import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
from puzzlelib_amd import nets, optim, lib, lazy
from puzzlelib_amd.surface import bound
surf = bound(); g = surf.gpuarray; bnd = surf.backend
np.random.seed(1)
net = nets.loadResNet(None, "50", actInplace=True, initscheme="he")
rng = np.random.RandomState(1)
data = g.to_gpu(rng.randn(256,3,224,224).astype(np.float32)); labels = g.to_gpu(rng.randint(0,1000,size=(256,)).astype(np.int32))
opt = optim.Adam(); opt.setupOn(net, useGlobalState=True)
tr = optim.Trainer(net, optim.CrossEntropy(), opt, batchsize=256); net.trainMode()
def stats(tag):
    st = bnd.memoryPool.getStats(); free, total = bnd.device.memoryInfo()
    print("%-18s live %.1f GB held %.1f GB (blocks %d/%d) device used %.1f GB" % (tag, st["liveBytes"]/1e9, st["heldBytes"]/1e9, st["liveBlocks"], st["heldBlocks"], (total-free)/1e9), flush=True)
stats("before")
for i in range(4):
    grad = tr.cost(net(data), labels, queryError=False); stats("step %d after fwd" % i)
    opt.zeroGradParams(); net.backward(grad, updGrad=False); stats("step %d after bwd" % i)
    opt.update(); net.reset(); lib.pz_device_sync(); stats("step %d after reset" % i)
