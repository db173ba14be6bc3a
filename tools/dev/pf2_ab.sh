cd /root/repo
for rep in 1 2; do
for v in 0 1; do
  echo "== PF2=$v"
  PUZZLE_MI355_IG_PF2=$v timeout 300 python tools/conv_census.py --reps 10 --passes fwd,dgrad 2>&1 | grep "1x1" | cut -c1-80
done
done
