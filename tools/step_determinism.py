"""ResNet-50 forward + backward three times (fused, fused again, literal call sequence): per-parameter gradient differences.
Everything must be bit-identical; B=<batch> NOSIDE=1 EXTRA_OFF=<lazy patterns> narrow a difference down."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from puzzlelib_amd import nets, optim, lazy, backend
from puzzlelib_amd.surface import bound
gpuarray = bound().gpuarray
backend.DnnContext.convStatsPolicy = "never"
B = int(os.environ.get("B", "256"))
np.random.seed(1234)
net = nets.loadResNet(None, "50", actInplace=True, initscheme="he")
net.layers.pop()
optimizer = optim.Adam(alpha=1e-3)
optimizer.setupOn(net, useGlobalState=True)
cost = optim.CrossEntropy()
rng = np.random.RandomState(1234)
data = gpuarray.to_gpu(rng.randn(B, 3, 224, 224).astype(np.float32))
labels = gpuarray.to_gpu(rng.randint(0, 1000, size=(B, )).astype(np.int32))
net.trainMode()
def passOnce():
	for layer in net.walk():
		if layer.kind == "bn":
			layer.cfg["passes"] = 0
			layer.attrs["mean"].fill(0.0)
			layer.attrs["var"].fill(1.0)
	logits = net(data)
	grad = cost(logits, labels, queryError=False)
	optimizer.zeroGradParams()
	net.backward(grad, updGrad=False)
	out = {name: p.grad.get() for name, p in net.namedParams().items()}
	net.reset()
	return out
lazy.disabled = {"bnbwdfold"} | ({"sidestream"} if os.environ.get("NOSIDE") else set()) | set(filter(None, os.environ.get("EXTRA_OFF", "").split(",")))
fused = passOnce()
fused2 = passOnce()
lazy.enabled = False
literal = passOnce()
for name in fused:
	d = np.abs(fused[name] - literal[name]).max()
	d2 = np.abs(fused[name] - fused2[name]).max()
	if d > 0 or d2 > 0:
		bad = np.flatnonzero((fused[name] != fused2[name]).ravel())
		print("   differing elements: %d of %d, first %s last %s" % (bad.size, fused[name].size, bad[:4], bad[-4:]))
		print("%-28s %-22s fused-literal %.3e  fused-fused %.3e  max %.3e" % (name, fused[name].shape, d, d2, np.abs(literal[name]).max()))
print("done")
