cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q -k "fusion or stats or bn or nets" 2>&1 | tail -3
for rep in 1 2 3; do for v in old new; do
  if [ $v = new ]; then L=""; else L="PUZZLE_MI355_LIB=$PWD/tools/dev/abl/lib_$v.so"; fi
  env $L python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v: %.2f ms/step  frac %.3f  dominant %.3f ms/launch' % (d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms']))"
done; done
