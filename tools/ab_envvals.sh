# bench.py --quick alternating over several values of one environment variable in one GPU session.
# usage: bash tools/ab_envvals.sh VAR rounds v1 v2 ...
set -e
cd $GRAFT_REPO_ROOT
V=$1; R=$2; shift 2
for i in $(seq 1 $R); do
  for x in "$@"; do
    env $V=$x timeout -k 10 200 python3 bench.py --no-cpu-baseline --quick --steps 60 --warmup 15 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=$x :', d['ms_per_step'], 'ms', d['value'], d['unit'])"
  done
done
