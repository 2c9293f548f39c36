# step time (bench.py --quick) for several values of one environment variable in one GPU session:
#   bash tools/ab_envval.sh FCD_BPRE_MINM 0 600 2100 8200
set -e
cd $GRAFT_REPO_ROOT
V=$1; shift
for rnd in 1 2; do
for val in "$@"; do
  env $V=$val timeout -k 10 200 python3 bench.py --no-cpu-baseline --quick --steps 60 --warmup 15 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=$val:', d['ms_per_step'], 'ms')"
done
done
