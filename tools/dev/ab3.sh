#!/bin/bash
# development aid: order-balanced A/B of the ResNet-50 step on one box: the library in the tree ("new") against
# tools/dev/abl/lib_old.so ("old", built from another commit's csrc): new old old new, twice (boxes drift by 0.1 ms per run)
cd "$(dirname "$0")/../.."
run() { env $2 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: %.2f ms/step  frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"; }
OLD="PUZZLE_MI355_LIB=$PWD/tools/dev/abl/lib_old.so"
run warm X=1 > /dev/null
for rep in 1 2; do
run new X=1; run old $OLD; run old $OLD; run new X=1
done
