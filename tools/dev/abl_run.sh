cd /root/repo
for v in new prio2 prio3 new prio2 prio3; do
  echo "== $v"
  if [ $v = new ]; then L=""; else L="PUZZLE_MI355_LIB=$PWD/tools/dev/abl/lib_$v.so"; fi
  env $L timeout 300 python tools/conv_census.py --reps 10 --passes fwd,dgrad 2>&1 | grep "1x1" | cut -c1-80
done
