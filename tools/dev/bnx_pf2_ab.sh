for i in 1 2; do
for v in 0 32 16; do
PUZZLE_MI355_IG_BNX_PF2=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bnx_pf2 $v', round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"
done
done
