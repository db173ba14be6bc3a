#!/bin/bash
# Package power and shader clock (rocm-smi, ~5 samples per second) while (a) one pointwise layer's forward launch repeats back to back
# and (b) the training step runs: does the matrix kernel alone clock higher than inside the step?   -> gpurun_out/power_clock.txt
cd "$(dirname "$0")/.."
sample() {       # sample <tag> <pid>: until the process ends
	while kill -0 $2 2>/dev/null; do
		rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | sed -E 's/.*sclk clock level: [^(]*\(([0-9]+)Mhz\).*/sclk \1/; s/.*Power \(W\): ([0-9.]+).*/power \1/' | tr '\n' ' '
		echo " $1"
		sleep 0.15
	done
}
{
for spec in "12 fwd" "3 fwd" "6 fwd"; do
	set -- $spec
	python tools/conv_census.py --reps 40000 --passes $2 --only $1 > /tmp/pc_census.txt 2>&1 &
	pid=$!
	sample "layer$1_$2" $pid
	wait $pid; grep -E "1x1|3x3" /tmp/pc_census.txt | cut -c1-70
done
python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-extras > /tmp/pc_bench.txt 2>&1 &
pid=$!
sample "step" $pid
wait $pid
} > gpurun_out/power_clock.txt 2>&1
python - <<'PY'
import re, collections
acc = collections.defaultdict(lambda: [[], []])
for l in open("gpurun_out/power_clock.txt"):
	m = re.match(r"sclk (\d+) power ([\d.]+)\s+(\S+)", l.strip())
	if m and float(m.group(2)) > 900:          # (samples taken while the process imports / allocates are idle ones)
		acc[m.group(3)][0].append(int(m.group(1))); acc[m.group(3)][1].append(float(m.group(2)))
for k, (c, p) in acc.items():
	c, p = sorted(c), sorted(p)
	print("%-14s %3d samples  sclk median %4d MHz (min %4d max %4d)   power median %6.0f W (max %6.0f)" % (k, len(c), c[len(c) // 2], c[0], c[-1], p[len(p) // 2], p[-1]))
PY
