# A/B of one environment switch whose OFF value is 0 (default on): bench.py --quick alternating VAR=0 / unset.
# usage: bash tools/ab_env0.sh FCD_MERGE_WGRAD [rounds]
set -e
cd $GRAFT_REPO_ROOT
V=$1; R=${2:-2}
for i in $(seq 1 $R); do
  env $V=0 timeout -k 10 200 python3 bench.py --no-cpu-baseline --quick --steps 60 --warmup 15 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=0  :', d['ms_per_step'], 'ms', d['value'], d['unit'])"
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --quick --steps 60 --warmup 15 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default:', d['ms_per_step'], 'ms', d['value'], d['unit'])"
done
