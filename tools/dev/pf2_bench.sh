cd /root/repo
for rep in 1 2; do for v in 0 16 32; do
  PUZZLE_MI355_IG_PF2=$v python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PF2 min ktiles $v: %.2f ms/step  frac %.3f  dominant %.3f ms/launch' % (d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms']))"
done; done
