# A/B of one environment switch inside one GPU session: bench.py --quick alternating with and without it.
# usage: bash tools/ab_env.sh FCD_NO_HALO7X [rounds]
set -e
cd $GRAFT_REPO_ROOT
V=$1; R=${2:-2}
for i in $(seq 1 $R); do
  env $V=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --quick --steps 60 --warmup 15 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('with   $V=1:', d['ms_per_step'], 'ms', d['value'], d['unit'])"
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --quick --steps 60 --warmup 15 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('without $V  :', d['ms_per_step'], 'ms', d['value'], d['unit'])"
done
