#!/bin/bash
# repeated bench processes on one box: is a later process slowed by earlier ones? (VRAM scrubbing after exit, clocks)
cd "$(dirname "$0")/.."
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$1', round(d['value']), round(d['ms_per_step'],2))"; }
run a; run b; run c; echo "sleep 30"; sleep 30; run d; rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -2
