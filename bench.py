"""
bench.py — ResNet-50 (the reference's variant, Models/Nets/ResNet.py:69-121) training step on synthetic ImageNet-shaped
fp32 data, batch 256 per GPU: forward + cross-entropy + zero-grad + backward + Adam (Handlers/Trainer.py:28-35), data
already resident in HBM. One process per GPU; for N > 1 launch with
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
(weak scaling: 256 images per GPU, gradients mean-all-reduced over RCCL, overlapped with backward).

Prints ONE JSON line on rank 0 with the driver's contract plus
  roofline      dominant kernel family (fp32 MFMA implicit-GEMM convolution): algorithmic FLOP / measured launch time
                (HIP events around every launch of the timed region, recorded on the launch stream) vs 157.3 TFLOP/s
  cpu_baseline  the numpy oracle (a restatement of the reference's CPU algorithm, extended with backward) timed on this
                host's cores on a bounded sample (rank 0, N == 1 only)
"""
import argparse, json, os, sys, time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 256
FLOP_PER_IMAGE = 22.770e9        # fwd + dgrad + wgrad, conv1 dgrad excluded; reported, not used for `value`
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
FAMILY = ["igemm_conv_kernel<128,128> (fwd + bwd-data)", "igemm_conv_kernel<64,256> (fwd + bwd-data)",
		  "wgrad_conv_kernel (bwd-filter)", "wino_conv_kernel / wino_wgrad_kernel F(2x2,3x3) (all three passes of the 3x3 layers)"]
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

	threads = os.cpu_count()
	try:                                               # the threads numpy's BLAS actually runs its GEMMs on
		from threadpoolctl import threadpool_info
		blas = [p["num_threads"] for p in threadpool_info() if p.get("user_api") == "blas"]
		threads = max(blas) if blas else threads
	except Exception:
		pass

	return {
		"value": sample_batch / dt, "unit": "images/sec", "cores": threads, "kind": "port",
		"sample": "1 training step (fwd+CE+bwd+Adam) of the same ResNet-50 at batch %d, %.1f s wall, numpy %s, BLAS threads %d "
				  "of %d host CPUs (im2col / element-wise parts are single-threaded numpy)" % (
			sample_batch, dt, np.__version__, threads, os.cpu_count()
		)
	}


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--gpus", type=int, default=1)
	ap.add_argument("--steps", type=int, default=10)
	ap.add_argument("--warmup", type=int, default=3)
	ap.add_argument("--batch", type=int, default=BATCH, help=argparse.SUPPRESS)
	ap.add_argument("--no-cpu-baseline", action="store_true")
	args = ap.parse_args()

	world = int(os.environ.get("WORLD_SIZE", "1"))
	rank = int(os.environ.get("RANK", "0"))
	local = int(os.environ.get("PUZZLE_MI355_DEVICE", os.environ.get("LOCAL_RANK", "0")))
	assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world)

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

	from puzzlelib_amd import nets, train, lib, backend
	from puzzlelib_amd.surface import bound
	import ctypes

	gpuarray = bound().gpuarray

	np.random.seed(1234)                        # identical seeds -> identical initial parameters on every rank
	# actInplace=True is the reference's own flag (Models/Nets/ResNet.py:63): ReLUs overwrite their input, which lets
	# this backend fold them into the neighbouring BatchNorm / Add / Replicate kernels (bit-identical results)
	net = nets.loadResNet(None, "50", actInplace=True, initscheme="he")

	rng = np.random.RandomState(1234 + rank)    # each rank trains on its own shard of the global mini-batch
	data = gpuarray.to_gpu(rng.randn(args.batch, 3, 224, 224).astype(np.float32))
	labels = gpuarray.to_gpu(rng.randint(0, 1000, size=(args.batch, )).astype(np.int32))

	optimizer = train.Adam(alpha=1e-3, nodeinfo=nodeinfo)
	optimizer.setupOn(net, useGlobalState=True)
	if nodeinfo is not None:
		grid.enableOverlap(optimizer, nodeinfo)

	cost = train.CrossEntropy()
	trainer = train.Trainer(net, cost, optimizer, batchsize=args.batch)
	net.trainMode()

	def step():
		trainer.handleBatch([data, labels], 0, None)
		net.reset()

	for _ in range(args.warmup):
		step()

	lib.pz_device_sync()
	grid.barrier()
	lib.pz_conv_profile_enable(1)

	t0 = time.perf_counter()
	for _ in range(args.steps):
		step()
	lib.pz_device_sync()
	grid.barrier()
	elapsed = time.perf_counter() - t0

	lib.pz_conv_profile_enable(0)
	ms = (ctypes.c_double * NFAM)()
	flops = (ctypes.c_double * NFAM)()
	launches = (ctypes.c_longlong * NFAM)()
	lib.pz_conv_profile_collect(ms, flops, launches)
	timed = [(ms[i], flops[i], launches[i]) for i in range(NFAM)]

	elapsed = grid.maxOverRanks(elapsed)
	loss = float(cost.getMeanError())

	# Kernel roofline. In the timed region the filter-gradient launches of every layer run on a second stream next to the
	# backward-data chain (DnnContext.overlapFilterGrad): the step gets shorter, but two kernels then share the CUs and a
	# launch's event-to-event time contains its neighbour's work. A kernel's own rate is therefore taken from ROOF_STEPS
	# more steps of the same loop with that overlap switched off (same kernels, same launches, one at a time); the
	# timed-region figures are reported next to it.
	concurrent = backend.DnnContext.overlapFilterGrad
	if concurrent:
		backend.DnnContext.overlapFilterGrad = False
		lib.pz_conv_profile_enable(1)
		for _ in range(ROOF_STEPS):
			step()
		lib.pz_device_sync()
		lib.pz_conv_profile_enable(0)
		lib.pz_conv_profile_collect(ms, flops, launches)
		backend.DnnContext.overlapFilterGrad = True
	roof_steps = ROOF_STEPS if concurrent else args.steps

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
			if concurrent and timed[i][2] > 0:          # the same family inside the timed region, launches sharing the device
				fams[-1]["timed_region_concurrent"] = {
					"launches": int(timed[i][2]), "avg_launch_ms": timed[i][0] / timed[i][2],
					"apparent_tflops": timed[i][1] / (timed[i][0] * 1e-3) / 1e12
				}
			if i == 3:         # direct-convolution FLOP / time; the matrix pipe executes 1/2.25 of them
				fams[-1]["note"] = "algorithmic (direct-convolution) TFLOP/s; MFMA-executed = achieved / 2.25"
	dom = max(range(NFAM), key=lambda i: ms[i])
	achieved = flops[dom] / (ms[dom] * 1e-3) / 1e12 if ms[dom] > 0 else 0.0

	# HBM bytes per launch of the dominant kernel family: memory-side L2 counters of the same command, collected with
	# rocprofv3 --pmc in separate passes and corrected as MI355X_MICROARCH.md prescribes (tools/pmc_bench.sh ->
	# profiles/r01_hbm_traffic.json). PMC collection cannot run inside the timed process, hence the committed summary.
	traffic, traffic_note = None, None
	try:
		prof = json.load(open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")))
		key = ["igemm_conv_kernel<128, 128", "igemm_conv_kernel<64, 256", "wgrad_conv_kernel<", "wino_"][dom]
		rows = [v for k, v in prof["kernels"].items() if key in k]
		n = sum(v["dispatches"] for v in rows)
		if n > 0 and args.batch == BATCH:
			traffic = sum(v["hbm_bytes_per_launch"] * v["dispatches"] for v in rows) / n
			traffic_note = "bytes per launch, mean over %d profiled launches; %s" % (n, prof["calibration"]["note"])
	except (OSError, KeyError, ValueError):
		pass

	result = {
		"metric": "images/sec fwd+bwd+Adam ResNet-50 224x224 fp32 b256 per GPU", "value": images_per_sec,
		"unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
		"ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
		"dtype": "f32", "data": "synthetic",
		"config": {
			"workload": "ResNet-50 (PuzzleLib variant, 55x55 stage 2) synthetic ImageNet 224x224 fp32, batch %d per GPU, "
						"fwd+CE+zeroGrad+bwd+Adam, random-init (he) weights, loadResNet(actInplace=True)" % args.batch,
			"global_batch": world * args.batch, "parallelism": "dp%d" % world,
			"grad_allreduce": "none" if world == 1 else (
				"RCCL sum + 1/N, 25 MB buckets overlapped with backward" if nodeinfo.transport == "rccl" else
				"FALLBACK: host-staged gloo all-reduce (RCCL communicator could not be created)"
			)
		},
		"model_tflops_note": "direct-convolution FLOP of the network / step time; the 3x3 layers' Winograd kernels execute "
							 "1/2.25 of their share on the matrix pipe",
		"model_tflops_per_gpu": images_per_sec / world * FLOP_PER_IMAGE / 1e12,
		"pct_of_f32_mfma_peak": images_per_sec / world * FLOP_PER_IMAGE / 1e12 / PEAK_F32_MFMA_TFLOPS * 100.0,
		"final_loss": loss,
		"roofline": {
			"kernel": FAMILY[dom], "bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
			"frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_note": traffic_note,
			"avg_launch_ms": ms[dom] / max(launches[dom], 1), "launches_measured": int(launches[dom]),
			"measured_over": (
				"%d extra steps of the timed loop, run right after it with the filter-gradient side stream off "
				"(PUZZLE_MI355_OVERLAP_WGRAD=0 semantics): in the timed region backward-data and backward-filter launches "
				"run concurrently and share the CUs, so their event-to-event times are not the kernels' own; see "
				"conv_kernel_families[].timed_region_concurrent" % ROOF_STEPS
			) if concurrent else "the timed region (every launch runs alone)"
		},
		"conv_kernel_families": fams,
	}

	if world == 1 and not args.no_cpu_baseline:
		result["cpu_baseline"] = cpu_baseline()

	print(json.dumps(result))


if __name__ == "__main__":
	main()
