# A/B of two builds of the library inside one GPU session: bench.py --quick alternating between fcd_amd/<lib A> and
# fcd_amd/<lib B> (FCD_HIP_LIB).   usage: bash tools/ab_lib_step.sh libfcd_hip_prev.so libfcd_hip.so [rounds]
set -e
cd $GRAFT_REPO_ROOT
A=$1; B=$2; R=${3:-2}
for i in $(seq 1 $R); do
  for L in $A $B; do
    FCD_HIP_LIB=$GRAFT_REPO_ROOT/fcd_amd/$L timeout -k 10 200 python3 bench.py --no-cpu-baseline --quick --steps 60 --warmup 15 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', d['ms_per_step'], 'ms', d['value'], d['unit'])"
  done
done
