"""A/B of the persistent implicit GEMM (csrc/conv.hip igemm_persist_kernel) against one workgroup per tile on the census layers
with short reductions: per layer and pass the sha1 of the result (must be equal: same products, same order) and the
steady-state time (--reps launches behind a warm-up). Spawns one process per setting of PUZZLE_MI355_IG_PERSIST.
    python tools/dev/persist_ab.py [--settings 0 6144,16 3000,16 ...] [--layers 3 4 7 9]"""
import argparse, hashlib, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def worker(layers, reps, batch):
	import numpy as np
	from conv_census import CENSUS
	from puzzlelib_amd import backend, lib, lazy
	lazy.disabled.add("sidestream")
	lazy.disabled.add("up2")
	bnd = backend.getBackend(0, initmode=2)
	G = bnd.GPUArray
	rng = np.random.RandomState(0)
	out = {}

	def timed(fn):
		for _ in range(max(3, reps // 5)):
			fn()
		lib.pz_device_sync()
		start, end = bnd.Driver.Event(), bnd.Driver.Event()
		start.record()
		for _ in range(reps):
			fn()
		end.record()
		end.synchronize()
		return start.timeTill(end) / reps

	for idx in layers:
		(c, h, w), (k, size, stride, pad), count = CENSUS[idx]
		x = G.toGpu(rng.randn(batch, c, h, w).astype(np.float32))
		W = G.toGpu((rng.randn(k, c, size, size) / np.sqrt(c * size * size)).astype(np.float32))
		y = bnd.dnn.convNd(x, W, None, stride, pad, allocator=bnd.memoryPool)
		dy = G.toGpu(rng.randn(*y.shape).astype(np.float32))
		dx = bnd.dnn.convNdBackwardData(dy, W, None, x, stride, pad, allocator=bnd.memoryPool)
		gflop = 2.0 * batch * k * y.shape[2] * y.shape[3] * c * size * size / 1e9
		row = {"name": "(%d,%d,%d)->(%d,%dx%d,s%d)" % (c, h, w, k, size, size, stride), "gflop": gflop,
			   "y": hashlib.sha1(y.get().tobytes()).hexdigest()[:12], "dx": hashlib.sha1(dx.get().tobytes()).hexdigest()[:12]}
		del dx, y
		row["fwd_ms"] = timed(lambda: bnd.dnn.convNd(x, W, None, stride, pad, allocator=bnd.memoryPool))
		row["dgrad_ms"] = timed(lambda: bnd.dnn.convNdBackwardData(dy, W, None, x, stride, pad, allocator=bnd.memoryPool))
		out[idx] = row
		del x, W, dy
	print("RESULT " + json.dumps(out))


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--settings", nargs="*", default=["0", "6144,16"])
	ap.add_argument("--env", default="PUZZLE_MI355_IG_PERSIST", help="the environment variable the settings are values of")
	ap.add_argument("--layers", nargs="*", type=int, default=[3, 4, 7, 9])
	ap.add_argument("--reps", type=int, default=300)
	ap.add_argument("--batch", type=int, default=256)
	ap.add_argument("--worker", action="store_true")
	args = ap.parse_args()
	if args.worker:
		worker(args.layers, args.reps, args.batch)
		return

	results = {}
	for setting in args.settings:
		env = dict(os.environ)
		env[args.env] = setting
		cmd = [sys.executable, os.path.abspath(__file__), "--worker", "--reps", str(args.reps), "--batch", str(args.batch), "--layers"] + [str(l) for l in args.layers]
		res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
		line = [l for l in res.stdout.splitlines() if l.startswith("RESULT ")]
		if not line:
			print("setting %s failed:\n%s\n%s" % (setting, res.stdout[-2000:], res.stderr[-3000:]))
			continue
		results[setting] = json.loads(line[0][7:])

	base = results.get(args.settings[0])
	print("%-30s %-10s | %9s %6s %s | %9s %6s %s" % ("layer", args.env[-10:], "fwd ms", "TF", "y==", "dgrad ms", "TF", "dx=="))
	for idx in args.layers:
		for setting in args.settings:
			r = results.get(setting, {}).get(str(idx))
			if r is None:
				continue
			b = base[str(idx)] if base else r
			print("%-30s %-10s | %9.4f %6.1f %s | %9.4f %6.1f %s" % (r["name"], setting, r["fwd_ms"], r["gflop"] / r["fwd_ms"], "ok " if r["y"] == b["y"] else "DIFF",
																r["dgrad_ms"], r["gflop"] / r["dgrad_ms"], "ok " if r["dx"] == b["dx"] else "DIFF"))


if __name__ == "__main__":
	main()
