# step time of two checkouts of the repository in one GPU session (bench.py --quick, alternating):
#   bash tools/ab_trees.sh _old . [rounds]
set -e
cd $GRAFT_REPO_ROOT
A=$1; B=$2; R=${3:-3}
for i in $(seq 1 $R); do
  for T in $A $B; do
    (cd $T && timeout -k 10 200 python3 bench.py --no-cpu-baseline --quick --steps 60 --warmup 15 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$T:', d['ms_per_step'], 'ms', d['value'], d['unit'])")
  done
done
