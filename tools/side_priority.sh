#!/bin/bash
# Does a LOW-PRIORITY second queue for the filter gradients fill the tails of the main stream's launches without competing
# with them? Same bench step (no side measurements) under: one stream (default policy), two streams of equal priority,
# two streams with the filter-gradient stream at the device's lowest / highest queue priority.
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms/step  %.0f img/s  side launches/step %.0f' % (d['ms_per_step'], d['value'], d['timed_region_host']['side_stream_launches_per_step']))"; }
for rep in 1 2; do
run PUZZLE_MI355_SIDE_MAX_GFLOP=15
run PUZZLE_MI355_SIDE_MAX_GFLOP=inf
run PUZZLE_MI355_SIDE_MAX_GFLOP=inf PUZZLE_MI355_SIDE_PRIORITY=low
run PUZZLE_MI355_SIDE_MAX_GFLOP=inf PUZZLE_MI355_SIDE_PRIORITY=high
done
