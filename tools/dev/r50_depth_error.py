"""development aid: where the full-depth ResNet-50 training step's distance to the oracle comes from (batch 16):
device `auto` (Winograd 3x3) and device with every convolution pinned to the implicit GEMM, each against the oracle with fp64
accumulation inside the operators; and the fp32-accumulating oracle against the fp64-accumulating one."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cpu_ref as R, cpu_net as N
from test_gpu_6_fulltensor import adoptDeviceGatesNested
from puzzlelib_amd import nets, optim, lazy, backend
from puzzlelib_amd.surface import bound

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
surf = bound()
gpuarray, Dnn = surf.gpuarray, surf.Dnn
np.random.seed(4321)
net = nets.loadResNet(None, "50", actInplace=True, initscheme="he")
net.layers.pop()
spec = nets.resnet50_spec(softmax=False)
optimizer = optim.Adam(alpha=1e-3)
optimizer.setupOn(net, useGlobalState=True)
cost = optim.CrossEntropy()
rng = np.random.RandomState(99)
data = rng.randn(batch, 3, 224, 224).astype(np.float32)
labels = rng.randint(0, 1000, size=(batch, )).astype(np.int32)
gdata, glabels = gpuarray.to_gpu(data), gpuarray.to_gpu(labels)
params = {name: p.data.get() for name, p in net.namedParams().items()}
net.trainMode()

def devicePass():
	for layer in net.walk():
		if layer.kind == "bn":
			layer.cfg["passes"] = 0
			layer.attrs["mean"].fill(0.0)
			layer.attrs["var"].fill(1.0)
	logits = net(gdata)
	grad = cost(logits, glabels, queryError=False)
	optimizer.zeroGradParams()
	net.backward(grad, updGrad=False)
	return logits, grad

def oracle(acc, gates=True):
	attrs = {name: (np.zeros(a.shape, np.float32) if name.endswith(".mean") else np.ones(a.shape, np.float32)) for name, a in net.namedAttrs().items()}
	cnet = N.CpuNet(spec, params, attrs, acc=acc)
	cnet.train = True
	ref_logits = cnet.forward(data)
	err, grad_ref = R.cross_entropy(ref_logits, labels)
	flips = mism = 0
	if gates:
		flips, mism = adoptDeviceGatesNested(cnet, net.layers, spec)
	cnet.zero_grads()
	cnet.backward(grad_ref)
	return cnet, ref_logits, flips, mism

def rel(a, b):
	return np.linalg.norm((a - b).astype(np.float64)) / (np.linalg.norm(b.astype(np.float64)) + 1e-30)

for mode in ("auto", "implicit-gemm"):
	for layer in net.walk():
		if layer.kind == "conv":
			layer.cfg["algos"] = (Dnn.ConvFwdAlgo.auto, Dnn.ConvBwdDataAlgo.auto, Dnn.ConvBwdFilterAlgo.auto) if mode == "auto" else (Dnn.ConvFwdAlgo.implicitGemm, Dnn.ConvBwdDataAlgo.implicitGemm, Dnn.ConvBwdFilterAlgo.implicitGemm)
	devicePass(); net.reset()
	logits, grad = devicePass()
	c64, l64, flips, mism = oracle(np.float64)
	got = {name: p.grad.get() for name, p in net.namedParams().items()}
	rels = sorted(((rel(got[n], c64.grads[n]), n) for n in got), reverse=True)
	print("%-14s vs fp64-acc oracle: logits %.2e of top, %d flips, fwd mismatch %.2e; worst grads %s; median %.2e" % (
		mode, np.abs(logits.get() - l64).max() / np.abs(l64).max(), flips, mism, ", ".join("%s %.2e" % (n, r) for r, n in rels[:4]), rels[len(rels) // 2][0]))
	if mode == "auto":
		c32, l32, _, _ = oracle(np.float32)
		rels = sorted(((rel(c32.grads[n], c64.grads[n]), n) for n in got), reverse=True)
		print("fp32-acc oracle vs fp64-acc oracle (same device gates): logits %.2e of top; worst grads %s; median %.2e" % (
			np.abs(l32 - l64).max() / np.abs(l64).max(), ", ".join("%s %.2e" % (n, r) for r, n in rels[:4]), rels[len(rels) // 2][0]))
	net.reset()
