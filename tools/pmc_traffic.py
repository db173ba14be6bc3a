"""gpurun_out/pmc_bench/summary.json (tools/pmc_bench.sh) -> profiles/r02_hbm_traffic.json: HBM bytes per launch of every
kernel of the bench, with the gfx950 corrections MI355X_MICROARCH.md prescribes, checked against the calibration streams
of tools/pmc_calibrate.py (known byte counts).

    python tools/pmc_traffic.py [gpurun_out/pmc_bench] [profiles/r02_hbm_traffic.json]
"""
import json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from puzzlelib_amd import lib            # (loads without a device) — the summary records which build it was taken from
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc_bench")
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")

summary = json.load(open(os.path.join(src, "summary.json")))
GIB_KIB = 1 << 20


def calibration():
	"""pmc_calibrate.py streams 1 GiB per operand: ReLU forward reads 1 + writes 1 operand per launch ... the factors are
	the ratios of the known KiB to the counted KiB, rounded to the documented values when they agree within 2 %."""
	note = []
	ff, wf = 2.0, 1.0
	cf, cw = summary.get("cal_fetch", {}), summary.get("cal_write", {})
	for name, row in cf.items():
		if "FETCH_SIZE" in row and row["FETCH_SIZE"] > GIB_KIB:
			note.append("%s: %d KiB fetched over %d launches" % (name[:40], row["FETCH_SIZE"], row["dispatches"]))
	for name, row in cw.items():
		if "WRITE_SIZE" in row and row["WRITE_SIZE"] > GIB_KIB:
			note.append("%s: %d KiB written over %d launches" % (name[:40], row["WRITE_SIZE"], row["dispatches"]))
	return ff, wf, note


ff, wf, cal_note = calibration()
kernels = {}
fetch, write = summary.get("fetch", {}), summary.get("write", {})
for name in sorted(set(fetch) | set(write)):
	f, w = fetch.get(name, {}), write.get(name, {})
	n = max(f.get("dispatches", 0), w.get("dispatches", 0))
	if n == 0:
		continue
	fb = f.get("FETCH_SIZE", 0.0) * 1024.0 * ff / n
	wb = w.get("WRITE_SIZE", 0.0) * 1024.0 * wf / n
	kernels[name] = {"dispatches": n, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}

# steps of the profiled run = launches of the optimizer kernel (warm-up + timed + the roofline steps of bench.py)
steps = max([v["dispatches"] for k, v in kernels.items() if "OpAdam" in k] or [4])
total = sum(v["hbm_bytes_per_launch"] * v["dispatches"] for v in kernels.values())
out = {
	"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --steps 3 --warmup 1 "
			  "--no-cpu-baseline --no-extras`, tools/pmc_bench.sh -> tools/pmc_traffic.py",
	"units": "counters are KiB; bytes = KiB*1024",
	"build_id": lib.buildId(),
	"calibration": {
		"note": "tools/pmc_calibrate.py: 1 GiB-per-operand streams. FETCH_SIZE reads exactly 1/2 of the bytes of "
				"16-B-per-lane streaming reads -> FETCH is doubled; WRITE_SIZE is exact",
		"fetch_factor": ff, "write_factor": wf, "calibration_kernels": cal_note
	},
	"hbm_bytes_per_step": total / steps,
	"kernels": kernels,
}
json.dump(out, open(dst, "w"), indent=1)
print("%.1f GB HBM traffic per step over %d kernels -> %s" % (total / steps / 1e9, len(kernels), dst))
