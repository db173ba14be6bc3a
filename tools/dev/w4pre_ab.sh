cd /root/repo
for mb in 0 200 500; do
  echo "== PUZZLE_MI355_WINO_PRE_MB=$mb"
  PUZZLE_MI355_WINO_PRE_MB=$mb python tools/wino_check.py --reps 20 2>&1 | grep -v "^\[Puzzle" | tail -6
done
