"""
bench.py — ResNet-50 (the reference's variant, Models/Nets/ResNet.py:69-121) training step on synthetic ImageNet-shaped
fp32 data, batch 256 per GPU: forward + cross-entropy + zero-grad + backward + Adam (Handlers/Trainer.py:28-35), data
already resident in HBM. One process per GPU: `python bench.py --gpus N` starts its N ranks itself; under
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
it is one of the ranks (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment). Weak scaling: 256 images per
GPU, gradients mean-all-reduced over RCCL, bucketed and overlapped with backward.

What is timed is the DROP-IN path: the executor (puzzlelib_amd/engine.py + optim.py) sends the backend exactly the call
sequence the reference's own Modules / Containers / Optimizer / Trainer send — tests/golden/trace_resnet50_b8.json was
recorded from the reference's Python on this backend and tests/test_host_logic.py holds the executor to it — every
wrapper with the reference's signature, conv1's (unused) input gradient included as the reference computes it. All
fusion happens behind those calls (puzzlelib_amd/lazy.py).

Prints ONE JSON line on rank 0 with the driver's contract plus
  roofline      dominant kernel family (fp32 MFMA implicit-GEMM convolution): algorithmic FLOP / measured launch time
                (HIP events around every launch, recorded on the launch stream) vs 157.3 TFLOP/s
  cpu_baseline  the numpy oracle (a restatement of the reference's CPU algorithm, extended with backward) timed on this
                host's cores on a bounded sample (rank 0, N == 1 only)
  dropin        the same step with the backend's lazy fusion switched off (one kernel per call, what a literal backend
                does with this call sequence), with the harness allowed to skip conv1's input gradient, and with the 3x3 layers
                forced onto the implicit GEMM (`winograd_off`: the number under SURVEY 8c's strict per-element bound)
  single_gpu_same_run   (N > 1 only) rank 0 alone, the N = 1 configuration, timed in the same process right before the
                data-parallel run — the N = 1 point of a scaling curve measured on the same box in the same minute
  configs       config 2 (Conv2D 3x3 64->128 56x56 b128, three passes, both kernel families) and config 3 (NiN b128 step)
"""
import argparse, json, os, subprocess, sys, time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 256
FLOP_PER_IMAGE = 22.770e9        # fwd + dgrad + wgrad, conv1 dgrad excluded; reported, not used for `value`
FLOP_CONV1_DGRAD = 0.236e9       # per image; executed on the drop-in path, not counted as algorithmic work
WINO_FLOP_PER_IMAGE = 3 * 3.67482e9             # direct-convolution FLOP of the 16 3x3 stride-1 layers, three passes
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
FAMILY = ["igemm_conv_kernel<128,128> (fwd + bwd-data)", "igemm_conv_kernel<64,256> (fwd + bwd-data)",
		  "wgrad_conv_kernel (bwd-filter)", "Winograd 3x3: wino4_conv_kernel F(4x4,3x3) (fwd, bwd-data) + wino_wgrad_kernel F(2x2,3x3) (bwd-filter)"]
NFAM = len(FAMILY)
ROOF_STEPS = 3


def cpu_baseline(sample_batch=32):
	"""Oracle ResNet-50 training step on the host (numpy im2col + sgemm, as the reference CPU backend does forward)."""
	sys.path.insert(0, os.path.join(ROOT, "oracle"))
	import cpu_net as N
	from puzzlelib_amd import nets

	spec = nets.resnet50_spec()
	rng = np.random.RandomState(1234)
	pshapes, ashapes = nets.spec_param_shapes(spec)

	params = {}
	for name, shape in pshapes.items():
		if name.endswith(".W"):
			fan = int(np.prod(shape[1:])) if len(shape) == 4 else shape[0]
			params[name] = (rng.randn(*shape) * np.sqrt(2.0 / fan)).astype(np.float32)
		elif name.endswith(".scale"):
			params[name] = np.ones(shape, np.float32)
		else:
			params[name] = np.zeros(shape, np.float32)
	attrs = {k: (np.zeros(s, np.float32) if k.endswith(".mean") else np.ones(s, np.float32)) for k, s in ashapes.items()}

	net = N.CpuNet(spec, params, attrs)
	opt = N.CpuAdam(net)
	data = rng.randn(sample_batch, 3, 224, 224).astype(np.float32)
	labels = rng.randint(0, 1000, size=(sample_batch, )).astype(np.int32)

	t0 = time.perf_counter()
	N.train_step(net, opt, data, labels)
	dt = time.perf_counter() - t0

	# the forward-only row: inference forward without the trailing SoftMax is everything the reference's numpy CPU backend
	# implements of this network (BASELINE.md sections 2-3: no backward, no training-mode BatchNorm, no SoftMax on CPU) —
	# the "reference-faithful" number; the training-step row above extends the restatement with backward
	infer = N.CpuNet(nets.resnet50_spec(softmax=False), params, attrs)
	infer.train = False
	t0 = time.perf_counter()
	infer.forward(data)
	dt_fwd = time.perf_counter() - t0

	threads, cap = os.cpu_count(), ""
	try:                                               # the threads numpy's BLAS actually runs its GEMMs on
		from threadpoolctl import threadpool_info
		blas = [p for p in threadpool_info() if p.get("user_api") == "blas"]
		if blas:
			threads = max(p["num_threads"] for p in blas)
			if threads < os.cpu_count():
				cap = " (%s %s as shipped in the numpy wheel is built for at most %d threads: the cap is the library's, no "\
					  "environment variable limits it)" % (blas[0].get("internal_api", "BLAS"), blas[0].get("version", ""), threads)
	except Exception:
		pass

	return {
		"value": sample_batch / dt, "unit": "images/sec", "cores": threads, "kind": "port",
		"sample": "1 training step (fwd+CE+bwd+Adam) of the same ResNet-50 at batch %d, %.1f s wall, numpy %s, BLAS threads %d "
				  "of %d host CPUs%s; im2col / element-wise parts are single-threaded numpy" % (
			sample_batch, dt, np.__version__, threads, os.cpu_count(), cap
		),
		"forward_only": {
			"value": sample_batch / dt_fwd, "unit": "images/sec",
			"what": "inference forward (BatchNorm on running statistics, no SoftMax) of the same network and batch, %.1f s wall: "
					"the part of the path the reference's own CPU backend implements (reference-faithful row of BASELINE.md "
					"section 3); the training-step value extends the restatement with backward" % dt_fwd
		}
	}


def spawnRanks(args):
	"""`python bench.py --gpus N` without a launcher: start the N ranks (one per GPU) and relay rank 0's line."""
	import socket, tempfile
	# MASTER_PORT is kept for whoever reads it (nothing here does); the host group's own port is bound by rank 0 (the system picks
	# it) and published through a file, so no port is chosen before somebody holds it (grid.FilePortCell)
	with socket.socket() as s:
		s.bind(("127.0.0.1", 0))
		port = s.getsockname()[1]
	scratch = tempfile.mkdtemp(prefix="puzzle_bench_")

	procs = []
	for rank in range(args.gpus):
		env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
				   MASTER_PORT=str(port), PUZZLE_MI355_PORT_FILE=os.path.join(scratch, "hostgroup.port"),
				   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
		procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
	# a rank that dies leaves the others waiting in the host group or inside a collective: stop them, report its status
	import time
	codes = [None] * len(procs)
	while any(c is None for c in codes):
		for i, p in enumerate(procs):
			if codes[i] is None:
				codes[i] = p.poll()
		if any(c not in (None, 0) for c in codes):
			for i, p in enumerate(procs):
				if codes[i] is None:
					p.terminate()
					codes[i] = p.wait()
			break
		time.sleep(0.05)
	import shutil
	shutil.rmtree(scratch, ignore_errors=True)
	sys.exit(max(abs(c) for c in codes))


def convLayers(net):
	"""every convolution layer of an engine.Net, residual branches included"""
	def visit(layers):
		for layer in layers:
			if layer.kind == "resid":
				for branch in layer.branches:
					yield from visit(branch)
			elif layer.kind == "conv":
				yield layer
	return list(visit(net.layers))


def oomEvents(lib):
	"""allocations that found the device full (e.g. while the driver still reclaims a previous process's memory): each one
	stalls the step it happens in — a non-zero count explains an outlier"""
	import ctypes
	count = ctypes.c_long(0)
	lib.pz_pool_oom_events(ctypes.byref(count))
	return count.value


def driverAllocs(lib):
	"""(pool misses served by hipMalloc, host seconds inside them) so far in this process"""
	import ctypes
	count, secs = ctypes.c_long(0), ctypes.c_double(0.0)
	lib.pz_pool_driver_allocs(ctypes.byref(count), ctypes.byref(secs))
	return count.value, secs.value


def timeSteps(step, n, lib, grid):
	lib.pz_device_sync()
	grid.barrier()
	t0 = time.perf_counter()
	for _ in range(n):
		step()
	lib.pz_device_sync()
	grid.barrier()
	return grid.maxOverRanks(time.perf_counter() - t0)


def sideConfigs(gpuarray, lib, optim, nets, bnd):
	"""configs 2 and 3 of BASELINE.json as extra keys (parity for both lives in tests/; these are timings only)"""
	out = {}
	dnn = bnd.dnn
	x = gpuarray.to_gpu(np.random.randn(128, 64, 56, 56).astype(np.float32))
	W = gpuarray.to_gpu((np.random.randn(128, 64, 3, 3) * 0.05).astype(np.float32))
	dy = gpuarray.to_gpu(np.random.randn(128, 128, 56, 56).astype(np.float32))
	flop = 2.0 * 128 * 128 * 56 * 56 * 64 * 9
	c2 = {}
	for label, algo in (("implicit_gemm", bnd.ConvFwdAlgo.implicitGemm.value), ("winograd_auto", bnd.ConvFwdAlgo.auto.value)):
		row = {}
		for name, fn in (
			("fwd", lambda: dnn.convNd(x, W, None, 1, 1, 1, 1, algo, None, bnd.memoryPool)),
			("bwd_data", lambda: dnn.convNdBackwardData(dy, W, None, x, 1, 1, 1, 0, 1, algo, None, bnd.memoryPool).rptr),
			("bwd_filter", lambda: dnn.convNdBackwardParams(x, dy, W, 1, 1, 1, 1, False, False, None, None, 1.0, 0.0, algo,
															 bnd.memoryPool).rptr),
		):
			secs, _ = bnd.timeKernel(fn, (), looplength=10, log=False, normalize=True)
			row[name + "_ms"] = secs * 1e3
			row[name + "_tflops_direct_equiv"] = flop / secs / 1e12
		c2[label] = row
	out["config2_conv3x3_64to128_56x56_b128"] = c2

	np.random.seed(1234)
	net = nets.buildNiN()
	opt = optim.MomentumSGD(learnRate=0.1, momRate=0.9)
	opt.setupOn(net, useGlobalState=True)
	opt.addHook(optim.WeightDecay(1e-4))
	trainer = optim.Trainer(net, optim.CrossEntropy(), opt, batchsize=128)
	data = gpuarray.to_gpu(np.random.randn(128, 3, 32, 32).astype(np.float32))
	labels = gpuarray.to_gpu(np.random.randint(0, 10, size=(128, )).astype(np.int32))
	net.trainMode()

	def step():
		trainer.step([data, labels])
		net.reset()
	for _ in range(5):
		step()
	lib.pz_device_sync()
	t0 = time.perf_counter()
	for _ in range(30):
		step()
	issued = (time.perf_counter() - t0) / 30 * 1e3
	lib.pz_device_sync()
	ms = (time.perf_counter() - t0) / 30 * 1e3
	out["config3_nin_cifar10_b128"] = {
		"ms_per_step": ms, "images_per_sec": 128 / ms * 1e3, "issue_ms_per_step": issued,
		"note": "host time to issue a step next to its wall time (the bound on outstanding filter-gradient launches ties the "
				"host to the device's pace: tools/nin_trace.sh shows the two streams)"
	}
	# the gather-free test bed of the MFMA loops (csrc/gemm.hip): one 4096^3 fp32 GEMM, uniform operands, 300 launches behind 60
	# warm ones (steady state; 20-launch bursts read up to 10 % off either way, profiles/r04_gemm_variants_probe.txt)
	m = 4096
	A = gpuarray.to_gpu(np.random.uniform(-1, 1, (m, m)).astype(np.float32))
	B = gpuarray.to_gpu(np.random.uniform(-1, 1, (m, m)).astype(np.float32))
	C = gpuarray.empty((m, m), dtype=np.float32)
	for _ in range(60):
		bnd.blas.gemm(A, B, C, False, False, 1.0, 0.0, bnd.memoryPool)
	secs, _ = bnd.timeKernel(lambda: bnd.blas.gemm(A, B, C, False, False, 1.0, 0.0, bnd.memoryPool), (), looplength=300, log=False,
							 normalize=True)
	out["gemm_testbed_4096_nn"] = {"us": secs * 1e6, "tflops": 2.0 * m ** 3 / secs / 1e12, "frac_of_f32_mfma_peak": 2.0 * m ** 3 / secs / 1e12 / PEAK_F32_MFMA_TFLOPS,
								   "operands": "uniform[-1,1)", "launches": 300}
	return out


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--gpus", type=int, default=1)
	ap.add_argument("--steps", type=int, default=10)
	ap.add_argument("--warmup", type=int, default=3)
	ap.add_argument("--batch", type=int, default=BATCH, help=argparse.SUPPRESS)
	ap.add_argument("--no-cpu-baseline", action="store_true")
	ap.add_argument("--no-extras", action="store_true", help="skip the dropin / config 2 / config 3 side measurements")
	args = ap.parse_args()

	if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
		spawnRanks(args)

	world = int(os.environ.get("WORLD_SIZE", "1"))
	rank = int(os.environ.get("RANK", "0"))
	local = int(os.environ.get("PUZZLE_MI355_DEVICE", os.environ.get("LOCAL_RANK", "0")))
	assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)

	from puzzlelib_amd.settings import Config
	from puzzlelib_amd import grid

	Config.deviceIdx = local

	import logging                               # stdout carries the JSON line only; the backend's banner goes to stderr
	log = logging.getLogger("puzzlelib_amd.bench")
	log.setLevel(logging.INFO)
	log.propagate = False
	log.addHandler(logging.StreamHandler(stream=sys.stderr))
	Config.logger = log
	nodeinfo = grid.nodeFromEnv()

	from puzzlelib_amd import nets, optim, lib, lazy, engine
	from puzzlelib_amd.surface import bound
	import ctypes

	surf = bound()
	gpuarray, bnd = surf.gpuarray, surf.backend
	# a variant library (measurement rig, experiment switches: anything but csrc/Makefile's default flags) or one older than
	# the sources must not produce a bench line; A/B runs of experiment builds say so explicitly and carry their flags
	if os.environ.get("PUZZLE_MI355_ALLOW_VARIANT") != "1":
		lib.requireShippedBuild()

	np.random.seed(1234)                        # identical seeds -> identical initial parameters on every rank
	# actInplace=True is the reference's own flag (Models/Nets/ResNet.py:63): ReLUs overwrite their input
	net = nets.loadResNet(None, "50", actInplace=True, initscheme="he")

	rng = np.random.RandomState(1234 + rank)    # each rank trains on its own shard of the global mini-batch
	data = gpuarray.to_gpu(rng.randn(args.batch, 3, 224, 224).astype(np.float32))
	labels = gpuarray.to_gpu(rng.randint(0, 1000, size=(args.batch, )).astype(np.int32))

	# N > 1: the N = 1 point of the scaling curve from this very run — rank 0 trains a second, private copy of the network
	# without a node (exactly the N = 1 configuration) while the other ranks wait at the host barrier
	single = None
	if world > 1:
		if rank == 0 and os.environ.get("PUZZLE_MI355_BENCH_N1", "1") == "1":
			np.random.seed(1234)
			solo = nets.loadResNet(None, "50", actInplace=True, initscheme="he")
			soloOpt = optim.Adam(alpha=1e-3)
			soloOpt.setupOn(solo, useGlobalState=True)
			soloTrainer = optim.Trainer(solo, optim.CrossEntropy(), soloOpt, batchsize=args.batch)
			solo.trainMode()

			def soloStep():
				soloTrainer.step([data, labels])
				solo.reset()
			for _ in range(args.warmup):
				soloStep()
			lib.pz_device_sync()
			t0 = time.perf_counter()
			for _ in range(args.steps):
				soloStep()
			lib.pz_device_sync()
			dt = time.perf_counter() - t0
			single = {"value": args.batch * args.steps / dt, "unit": "images/sec", "ms_per_step": dt / args.steps * 1e3,
					  "steps": args.steps, "warmup": args.warmup,
					  "what": "rank 0 alone, no node, no exchange: bench.py --gpus 1 inside this process, before the %d-rank run" % world}
			del soloTrainer, soloOpt, solo
			bnd.memoryPool.freeHeld()
		grid.barrier()

	optimizer = optim.Adam(alpha=1e-3, nodeinfo=nodeinfo)
	optimizer.setupOn(net, useGlobalState=True)
	if nodeinfo is not None:
		grid.enableOverlap(optimizer, nodeinfo)

	cost = optim.CrossEntropy()
	trainer = optim.Trainer(net, cost, optimizer, batchsize=args.batch)
	net.trainMode()

	def step():
		trainer.step([data, labels])
		net.reset()

	for _ in range(args.warmup):
		step()

	lib.pz_device_sync()
	grid.barrier()

	# the timed region: K steps, nothing but the steps (no per-launch events; those come from a second pass of the same loop)
	allocs0 = driverAllocs(lib)
	side0 = bnd.dnn.sideLaunches
	t0 = time.perf_counter()
	for _ in range(args.steps):
		step()
	issued = time.perf_counter() - t0
	side_launches = bnd.dnn.sideLaunches - side0
	lib.pz_device_sync()
	grid.barrier()
	elapsed = time.perf_counter() - t0
	allocs1 = driverAllocs(lib)
	elapsed = grid.maxOverRanks(elapsed)
	loss = float(cost.getMeanError())
	fusion_counts = dict(lazy.counters)
	comm = nodeinfo.commSummary() if nodeinfo is not None and hasattr(nodeinfo, "commSummary") else None

	# kernel roofline: the SAME loop once more with HIP events around every convolution launch (recorded on the launch
	# stream) — kept out of the timed region, where ~160 event pairs per step would only bias `value` down
	prof_steps = max(1, min(args.steps, ROOF_STEPS if args.no_extras else args.steps))
	lib.pz_conv_profile_enable(1)
	t1 = time.perf_counter()
	for _ in range(prof_steps):
		step()
	lib.pz_device_sync()
	profiled_elapsed = time.perf_counter() - t1
	lib.pz_conv_profile_enable(0)
	ms = (ctypes.c_double * NFAM)()
	flops = (ctypes.c_double * NFAM)()
	launches = (ctypes.c_longlong * NFAM)()
	lib.pz_conv_profile_collect(ms, flops, launches)
	timed = [(ms[i], flops[i], launches[i]) for i in range(NFAM)]

	# Kernel roofline. When the backend put filter-gradient launches on its second stream during the timed region (its
	# policy does so for networks of short kernels, backend.DnnContext.filterGradStream — not for this one at batch 256),
	# two kernels shared the CUs and a launch's event-to-event time contains its neighbour's work: a kernel's own rate is
	# then taken from ROOF_STEPS more steps of the same loop with that stream off (same kernels, same launches, one at a
	# time) and the timed-region figures are reported next to it. Otherwise the timed region itself is the measurement.
	concurrent = side_launches > 0
	if concurrent:
		lazy.disabled.add("sidestream")
		step()
		lib.pz_conv_profile_enable(1)
		for _ in range(ROOF_STEPS):
			step()
		lib.pz_device_sync()
		lib.pz_conv_profile_enable(0)
		lib.pz_conv_profile_collect(ms, flops, launches)
		lazy.disabled.discard("sidestream")
	roof_steps = ROOF_STEPS if concurrent else prof_steps

	# the same call sequence on a literal backend (no lazy fusion: every call launches its own kernel(s)), and the
	# executor's one permitted deviation (conv1's input gradient, which nobody reads, left out)
	dropin = None
	if not args.no_extras:
		n = max(3, min(args.steps, 5))
		engine.Net.skipInputGrad = True
		step()
		t_skip = timeSteps(step, n, lib, grid) / n
		engine.Net.skipInputGrad = False
		lazy.enabled = False
		step()
		t_literal = timeSteps(step, n, lib, grid) / n
		lazy.enabled = True
		step()
		# the sixteen 3x3 layers on the implicit GEMM in all three passes (what `auto` gives them is Winograd F(4x4) / F(2x2),
		# held to |err| <= 6e-5 max|ref| instead of the per-element bound the implicit GEMM meets: `tolerance`)
		igemm = (surf.Dnn.ConvFwdAlgo.implicitGemm, surf.Dnn.ConvBwdDataAlgo.implicitGemm, surf.Dnn.ConvBwdFilterAlgo.implicitGemm)
		threes = [(layer, layer.cfg["algos"]) for layer in convLayers(net) if tuple(layer.params["W"].data.shape[2:]) == (3, 3)]
		for layer, _ in threes:
			layer.cfg["algos"] = igemm
		step(); step()
		t_igemm = timeSteps(step, n, lib, grid) / n
		for layer, algos in threes:
			layer.cfg["algos"] = algos
		step()
		dropin = {
			"caller": "reference-literal call sequence through wrappers with the reference's signatures only "
					  "(tests/golden/trace_resnet50_b8.json, recorded from the reference's own Python on this backend)",
			"images_per_sec": world * args.batch * args.steps / elapsed, "ms_per_step": elapsed / args.steps * 1e3,
			"note": "this IS `value`: there is no patched caller any more; fusion is decided inside the backend",
			"lazy_fusion_off": {"images_per_sec": world * args.batch / t_literal, "ms_per_step": t_literal * 1e3,
								"what": "PUZZLE_MI355_LAZY=0: one kernel (or two) per reference call, nothing deferred"},
			"winograd_off": {"images_per_sec": world * args.batch / t_igemm, "ms_per_step": t_igemm * 1e3, "layers": len(threes),
							 "what": "the 3x3 stride-1 layers pinned to the implicit GEMM (forward, backward-data, backward-filter): every "
									 "convolution of the step then meets |err| <= 1e-5*s + 1e-4*|ref| per element (tolerance.implicit_gemm_and_stem)"},
			"harness_skips_conv1_input_grad": {"images_per_sec": world * args.batch / t_skip, "ms_per_step": t_skip * 1e3,
											   "what": "engine.Net.skipInputGrad=True: updGrad=False honoured (the reference "
													   "means to, Containers/Sequential.py:215-218 is unreachable)"},
		}

	# the same steps with the MFMA kernels in a split math mode (backend.DnnContext.MATH; off by default: `value` is
	# the fp32-MFMA path unless PUZZLE_MI355_MATH says otherwise)
	mathModes = None
	if not args.no_extras:
		current = bnd.dnn.convMath
		mathModes = {"value_measured_with": current}
		for mode in ("f32", "split6", "split9"):
			if mode == current:
				continue
			bnd.dnn.setConvMath(mode)
			step(); step()
			t_mode = timeSteps(step, n, lib, grid) / n
			mathModes[mode] = {"images_per_sec": world * args.batch / t_mode, "ms_per_step": t_mode * 1e3}
		bnd.dnn.setConvMath(current)
		step()
		mathModes["what"] = (
			"f32 = v_mfma_f32_32x32x2_f32. split6 / split9 = every fp32 operand of the 1x1 convolutions (forward, backward-data, "
			"backward-filter of >= 128-channel layers) split exactly into three bf16 terms, 6 / 9 partial products on "
			"v_mfma_f32_32x32x16_bf16, fp32 accumulation; inputs, outputs and every other kernel unchanged; error against "
			"float64 equal to the fp32 MFMA's (tools/probes/split_probe.hip, tools/split_wgrad_check.py, DESIGN.md 3.1e). "
			"Split modes run on one stream (no filter-gradient overlap)."
		)

	extras = None
	if rank == 0 and world == 1 and not args.no_extras and args.batch == BATCH:
		net.reset()
		extras = sideConfigs(gpuarray, lib, optim, nets, bnd)

	if nodeinfo is not None:                    # leave together: no rank tears its communicator down under a peer's collective
		lib.pz_device_sync()
		grid.barrier()
		nodeinfo.close()
		grid.barrier()

	if rank != 0:
		return

	images_per_sec = world * args.batch * args.steps / elapsed

	fams = []
	for i in range(NFAM):
		if launches[i] > 0:
			fams.append({
				"kernel": FAMILY[i], "launches": int(launches[i]), "avg_launch_ms": ms[i] / launches[i],
				"total_ms_per_step": ms[i] / roof_steps, "achieved_tflops": flops[i] / (ms[i] * 1e-3) / 1e12
			})
			# fraction of the fp32-MFMA peak by FLOP the matrix pipe executes: all of them for the implicit GEMM families; for the
			# Winograd family 1/4 of the forward + backward-data share (two of its three passes) and 1/2.25 of the backward-filter share
			share = (2.0 / 3.0 / 4.0 + 1.0 / 3.0 / 2.25) if i == 3 else 1.0
			fams[-1]["frac"] = fams[-1]["achieved_tflops"] * share / PEAK_F32_MFMA_TFLOPS
			fams[-1]["frac_is"] = "executed FLOP / time / %.1f TFLOP/s" % PEAK_F32_MFMA_TFLOPS + (
				" (direct-equivalent rate x %.4f)" % share if i == 3 else "")
			if concurrent and timed[i][2] > 0:          # the same family inside the timed region, launches sharing the device
				fams[-1]["timed_region_concurrent"] = {
					"launches": int(timed[i][2]), "avg_launch_ms": timed[i][0] / timed[i][2],
					"apparent_tflops": timed[i][1] / (timed[i][0] * 1e-3) / 1e12
				}
			if i == 3:         # direct-convolution FLOP / time; the matrix pipe executes 1/4 (F(4x4)) or 1/2.25 (F(2x2)) of them
				fams[-1]["note"] = ("algorithmic (direct-convolution) TFLOP/s; the matrix pipe executes 1/4 of the forward / "
									"backward-data share (F(4x4,3x3)) and 1/2.25 of the backward-filter share (F(2x2,3x3))")
	dom = max(range(NFAM), key=lambda i: ms[i])
	achieved = flops[dom] / (ms[dom] * 1e-3) / 1e12 if ms[dom] > 0 else 0.0

	# HBM bytes per launch of the dominant kernel family: memory-side L2 counters of the same command, collected with
	# rocprofv3 --pmc in separate passes and corrected as MI355X_MICROARCH.md prescribes (tools/pmc_bench.sh ->
	# profiles/*_hbm_traffic.json). PMC collection cannot run inside the timed process; the summary is only used when it was
	# taken from a library built from the very sources loaded now (build id recorded in the summary), else null.
	traffic, traffic_note = None, "no PMC summary for build %s under profiles/" % lib.buildId()
	try:
		import glob
		for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")), reverse=True):
			prof = json.load(open(path))
			if prof.get("build_id") != lib.buildId() or args.batch != BATCH:
				continue
			key = ["igemm_conv_kernel<128, 128", "igemm_conv_kernel<64, 256", "wgrad_conv_kernel<", "wino_"][dom]
			rows = [v for k, v in prof["kernels"].items() if key in k]
			n = sum(v["dispatches"] for v in rows)
			if n > 0:
				traffic = sum(v["hbm_bytes_per_launch"] * v["dispatches"] for v in rows) / n
				traffic_note = "bytes per launch, mean over %d profiled launches (%s); %s" % (
					n, os.path.basename(path), prof["calibration"]["note"]
				)
				break
	except (OSError, KeyError, ValueError):
		pass

	per_gpu = images_per_sec / world
	wino_executed = WINO_FLOP_PER_IMAGE * (2.0 / 3.0 / 4.0 + 1.0 / 3.0 / 2.25)      # fwd + bwd-data on F(4x4), bwd-filter on F(2x2)
	executed = (FLOP_PER_IMAGE - WINO_FLOP_PER_IMAGE + wino_executed + FLOP_CONV1_DGRAD) * per_gpu / 1e12
	result = {
		"metric": "images/sec fwd+bwd+Adam ResNet-50 224x224 fp32 b256 per GPU", "value": images_per_sec,
		"unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
		"ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
		"dtype": "f32", "data": "synthetic", "math": bnd.dnn.convMath,
		"config": {
			"workload": "ResNet-50 (PuzzleLib variant, 55x55 stage 2) synthetic ImageNet 224x224 fp32, batch %d per GPU, "
						"fwd+CE+zeroGrad+bwd+Adam, random-init (he) weights, loadResNet(actInplace=True), reference-literal "
						"call sequence (conv1 input gradient computed as the reference does)" % args.batch,
			"global_batch": world * args.batch, "parallelism": "dp%d" % world,
			"grad_allreduce": "none" if nodeinfo is None else (
				"single-rank rehearsal (PUZZLE_MI355_FORCE_COMM=1): " if world == 1 else "") + (
				"RCCL (ncclCommCount = %d ranks) sum + 1/N, 25 MB buckets in reverse execution order, overlapped with backward"
				% nodeinfo.commRanks if nodeinfo.transport == "rccl" else
				"FALLBACK: host-staged all-reduce over TCP (RCCL communicator could not be created)"
			),
			"grad_transport": "none" if nodeinfo is None else nodeinfo.transport,
			"rccl_nranks": 0 if nodeinfo is None else nodeinfo.commRanks,
			"comm": comm,
		},
		"tolerance": {
			"what": "per-element parity bounds the GPU suite enforces on the kernels this number was measured with "
					"(tests/test_gpu_6_fulltensor.py: whole y / dx / dw of every ResNet-50 convolution at batch 256 against the fp64 "
					"oracle; tests/test_gpu_0_ops.py at oracle sizes)",
			"implicit_gemm_and_stem": "|err| <= 1e-5*s + 1e-4*|ref|, s = max(1, rms(ref)) for y and dx, sqrt(N*P*Q) for dw",
			"winograd_f4x4_fwd_bwd_data": "|err| <= 6e-5 * max|ref|",
			"winograd_f2x2_bwd_filter": "|err| <= 2e-5 * max|ref| + 1e-5*sqrt(N*P*Q)",
			"batchnorm_pool_eltwise_softmax": "atol 1e-5..1e-4, rtol 1e-4 against the oracle (tests/test_gpu_0_ops.py)",
			"full_depth_training_step": "every parameter gradient of ResNet-50 (b16, train mode, all fusions): relative L2 error "
										"<= max(5e-5, 2x the fp32-summing oracle's own distance from the fp64-summing oracle); median <= 5e-5",
		},
		"build": {"library_build_id": lib.buildId(), "source_id": lib.sourceId(),
				  "matches_sources": lib.sourceId() in (None, lib.buildId()),
				  "flags": lib.buildFlags(), "shipped_flags": lib.buildFlags() == lib.defaultFlags()},
		"model_tflops_note": "direct-convolution FLOP of the network (conv1 dgrad not counted) / step time",
		"model_tflops_per_gpu": per_gpu * FLOP_PER_IMAGE / 1e12,
		"pct_of_f32_mfma_peak": per_gpu * FLOP_PER_IMAGE / 1e12 / PEAK_F32_MFMA_TFLOPS * 100.0,
		"pct_of_f32_mfma_peak_executed": executed / PEAK_F32_MFMA_TFLOPS * 100.0,
		"pct_executed_note": "FLOP the matrix pipe actually executes per step: the 3x3 layers' Winograd kernels do 1/4 (forward, "
							 "backward-data: F(4x4,3x3)) and 1/2.25 (backward-filter: F(2x2,3x3)) of their direct-convolution share, "
							 "tile padding not counted; conv1's input gradient is executed and counted here",
		"final_loss": loss,
		"pool_out_of_memory_events": oomEvents(lib),
		"timed_region_host": {
			"issue_ms_per_step": issued / args.steps * 1e3,
			"driver_allocations": allocs1[0] - allocs0[0], "driver_allocation_ms": (allocs1[1] - allocs0[1]) * 1e3,
			"side_stream_launches_per_step": side_launches / args.steps,
			"note": "host time to issue the timed steps (no device wait inside), and pool misses that went to hipMalloc during "
					"them: an outlier step time with normal kernel times shows up here",
		},
		"backend_fusion_counts_total": fusion_counts,
		"roofline": {
			"kernel": FAMILY[dom], "bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
			"frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_note": traffic_note,
			"avg_launch_ms": ms[dom] / max(launches[dom], 1), "launches_measured": int(launches[dom]),
			"measured_over": (
				"%d extra steps of the timed loop, run right after it with the filter-gradient side stream off: in the timed "
				"region backward-data and backward-filter launches run concurrently and share the CUs, so their event-to-event "
				"times are not the kernels' own; see conv_kernel_families[].timed_region_concurrent" % ROOF_STEPS
			) if concurrent else (
				"%d steps of the timed loop run again right behind the timed region with HIP events around every convolution launch "
				"(%.2f ms per step with the events, %.2f ms without): every launch runs alone, on the stream the events are recorded on"
				% (prof_steps, profiled_elapsed / prof_steps * 1e3, elapsed / args.steps * 1e3)
			)
		},
		"conv_kernel_families": fams,
	}
	if single is not None:
		result["single_gpu_same_run"] = single
	if dropin is not None:
		result["dropin"] = dropin
	if mathModes is not None:
		result["math_modes"] = mathModes
	if extras is not None:
		result["configs"] = extras

	if world == 1 and not args.no_cpu_baseline:
		result["cpu_baseline"] = cpu_baseline()

	if world > 1 and nodeinfo.transport != "rccl" and os.environ.get("PUZZLE_MI355_ALLOW_FALLBACK", "0") != "1":
		# a multi-GPU number measured over the TCP fallback is not the design's number: no JSON line on stdout, rc != 0
		# (PUZZLE_MI355_ALLOW_FALLBACK=1 keeps the line — rehearsals with several ranks on one device)
		print(json.dumps(result), file=sys.stderr)
		print("bench.py: %d ranks ended on the host-staged fallback transport (RCCL communicator could not be created) — "
			  "refusing to report it as the data-parallel result" % world, file=sys.stderr, flush=True)
		sys.exit(3)
	print(json.dumps(result))


if __name__ == "__main__":
	main()
