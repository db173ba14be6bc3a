#!/bin/bash
# One NiN (config 3) training step as the device sees it: kernel sequence with start offsets and durations of the LAST step
# of a short run (rocprofv3 --kernel-trace), main and filter-gradient stream side by side.
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=/tmp/nin_trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $ROOT/tools/nin_step.py 30 > $OUT/log.txt 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step starts at the last but one OpClassicMomSGD / optimizer kernel
opt = [i for i, r in enumerate(rows) if "MomSGD" in r["Kernel_Name"]]
lo, hi = opt[-2] + 1, opt[-1] + 1
t0 = int(rows[lo]["Start_Timestamp"])
tot = 0
for r in rows[lo:hi]:
	s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
	tot += e - s
	name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
	name = re.sub(r"\(.*", "", name)[:70]
	print("%9.1f us  +%7.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), name))
print("step: %d kernels, %.1f us of kernel time, %.1f us wall" % (hi - lo, tot / 1e3, (int(rows[hi - 1]["End_Timestamp"]) - t0) / 1e3))
PY
