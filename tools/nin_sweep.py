"""Config 3 (CIFAR-10 NiN) gradient parity over many dropout seeds: is the tail of `grad conv1.W` against the oracle ReLU-gate
flips (rounding-level pre-activations gating differently on the two sides) or a cross-stream hazard?

Per seed the device step runs four times — filter gradients on the second stream or not (PUZZLE_MI355_SIDE_MAX_GFLOP inf / 0)
x lazy layer on / off — with the same seeded Philox words; the four gradient sets are compared bit for bit with each other
and, as relative L2 error per parameter, with the oracle's backward run twice: once gating with ITS OWN ReLU outputs /
max-pool inputs, once with the DEVICE's (read back after the step). A hazard would show as (a) configurations that differ
from each other, or (b) an error that stays large with the device's gates. Usage: nin_sweep.py [seeds=100] [batch=8]
"""
import os, sys, json
import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from puzzlelib_amd import nets, optim, lib, lazy, backend
from puzzlelib_amd.surface import bound
import cpu_net as N, cpu_ref as R           # checker

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8

surf = bound()
gpuarray = surf.gpuarray
bnd = gpuarray.backend
spec = nets.nin_spec()
print("build %s, %d seeds, batch %d" % (lib.buildId(), seeds, batch))

CONFIGS = [("side_inf_lazy1", float("inf"), True), ("side_0_lazy1", 0.0, True), ("side_inf_lazy0", float("inf"), False),
		   ("side_0_lazy0", 0.0, False)]


def deviceStep(seed, data, labels, side, lz):
	backend.DnnContext.sideStreamMaxGflop = side
	bnd.dnn.sideWorkMean = 0.0
	lazy.enabled = lz
	np.random.seed(1234)
	net = nets.buildNiN()
	rng = backend.RandomNumberGenerator(seed=seed)
	for layer in net.walk():
		if layer.kind == "dropout":
			layer.cfg["rng"] = rng
	optimizer = optim.MomentumSGD(learnRate=0.1, momRate=0.9)
	optimizer.setupOn(net, useGlobalState=True)
	cost = optim.CrossEntropy()
	net.trainMode()
	params = {name: p.data.get() for name, p in net.namedParams().items()}
	pred = net(gpuarray.to_gpu(data))
	grad = cost(pred, gpuarray.to_gpu(labels), queryError=False)
	optimizer.zeroGradParams()
	net.backward(grad, updGrad=False)
	grads = {name: p.grad.get() for name, p in net.namedParams().items()}
	# what the device gated with, read AFTER the step so that nothing is settled early
	gates, masks = {}, {}
	for idx, layer in enumerate(net.layers):
		if layer.kind == "act":
			gates[str(idx)] = layer.y.get()
		elif layer.kind == "pool" and spec[idx][0] == "maxpool":
			gates[str(idx)] = (layer.x.get(), layer.y.get())
		elif layer.kind == "dropout":
			masks[layer.name] = layer.aux[0].get()
	return params, pred.get(), grads, gates, masks


def relL2(got, ref):
	return float(np.linalg.norm((got - ref).astype(np.float64)) / (np.linalg.norm(ref.astype(np.float64)) + 1e-30))


rows = []
mismatch = 0
for seed in range(seeds):
	rs = np.random.RandomState(1000 + seed)
	data = rs.randn(batch, 3, 32, 32).astype(np.float32)
	labels = rs.randint(0, 10, size=(batch, )).astype(np.int32)

	runs = {}
	for name, side, lz in CONFIGS:
		runs[name] = deviceStep(seed + 1, data, labels, side, lz)
	params, pred, grads, gates, masks = runs[CONFIGS[0][0]]
	same = {}
	for name, _, _ in CONFIGS[1:]:
		same[name] = all(np.array_equal(grads[k], runs[name][2][k]) for k in grads)
		mismatch += not same[name]

	cnet = N.CpuNet(spec, params)
	cnet.dropmasks.update(masks)
	cnet.train = True
	pred_ref = cnet.forward(data)
	_, grad_ref = R.cross_entropy(pred_ref, labels)
	cnet.zero_grads()
	cnet.backward(grad_ref)
	own = {k: relL2(grads[k], cnet.grads[k]) for k in grads}
	flips = 0
	for key, val in gates.items():
		if isinstance(val, tuple):
			cnet.cache[key] = val
		else:
			flips += int(((val > 0) != (cnet.cache[key] > 0)).sum())
			cnet.cache[key] = val
	cnet.zero_grads()
	cnet.backward(grad_ref)
	dev = {k: relL2(grads[k], cnet.grads[k]) for k in grads}
	worstOwn, worstDev = max(own, key=own.get), max(dev, key=dev.get)
	rows.append({"seed": seed, "bit_identical": same, "relu_gate_flips": flips, "fwd_max_abs_err": float(np.abs(pred - pred_ref).max()),
				 "own_gates": {"worst": worstOwn, "rel_l2": own[worstOwn], "conv1.W": own["conv1.W"]},
				 "device_gates": {"worst": worstDev, "rel_l2": dev[worstDev], "conv1.W": dev["conv1.W"]}})
	print("seed %3d: flips %3d  own gates %.3e (%s)  device gates %.3e (%s)  configs identical %s" % (
		seed, flips, own[worstOwn], worstOwn, dev[worstDev], worstDev, all(same.values())), flush=True)

own = np.array([r["own_gates"]["rel_l2"] for r in rows])
dev = np.array([r["device_gates"]["rel_l2"] for r in rows])
edges = [0, 1e-6, 3e-6, 1e-5, 3e-5, 1e-4, 3e-4, 1e-3, 2e-3, 5e-3, 1e-2, 1.0]
summary = {
	"build": lib.buildId(), "seeds": seeds, "batch": batch, "configs": [c[0] for c in CONFIGS],
	"runs_not_bit_identical_to_first_config": mismatch,
	"hist_edges_rel_l2": edges,
	"hist_worst_param_own_gates": np.histogram(own, bins=edges)[0].tolist(),
	"hist_worst_param_device_gates": np.histogram(dev, bins=edges)[0].tolist(),
	"max_own_gates": float(own.max()), "max_device_gates": float(dev.max()),
	"seeds_over_2e-3_own_gates": int((own > 2e-3).sum()), "seeds_over_5e-4_device_gates": int((dev > 5e-4).sum()),
	"total_relu_gate_flips": int(sum(r["relu_gate_flips"] for r in rows)),
}
print(json.dumps(summary, indent=1))
out = os.environ.get("NIN_SWEEP_OUT")
if out:
	with open(out, "w") as f:
		json.dump({"summary": summary, "rows": rows}, f, indent=1)
