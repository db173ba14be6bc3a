cd /root/repo
for rep in 1 2; do for v in base tg8 tg64 tgoff; do
  if [ $v = base ]; then L=""; else L="PUZZLE_MI355_LIB=$PWD/tools/dev/abl/lib_$v.so"; fi
  env $L python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v: %.2f ms/step  frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"
done; done
