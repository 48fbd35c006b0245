cd $GRAFT_REPO_ROOT
for CFG in "3 8" "4 5" "4 8" "5 4" "6 8" "3 12" "2 10" "4 6" "5 8" "3 7"; do
  set -- $CFG
  for S in 20 200; do
    V=$(python bench.py --gpus 1 --steps $S --warmup 5 --no-extras --no-cpu-baseline --no-quatro --in-flight $1 --lanes $2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['config']['value_repeats']['median'])")
    echo "in_flight $1 lanes $2 steps $S: value/median-of-repeats $V"
  done
done
