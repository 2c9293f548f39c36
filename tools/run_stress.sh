set -e
cd $GRAFT_REPO_ROOT
P=${1:-fp32x2}
timeout -k 10 300 python3 tools/stress_probe.py $P 32 nopair > gpurun_out/stress_a.log 2>&1 && tail -1 gpurun_out/stress_a.log
timeout -k 10 300 python3 tools/stress_probe.py $P 32 pair > gpurun_out/stress_b.log 2>&1 && tail -1 gpurun_out/stress_b.log
timeout -k 10 500 python3 -u -m pytest tests/test_stress_gpu.py -x -v --timeout 400 --timeout-method thread > gpurun_out/stress_t.log 2>&1; tail -15 gpurun_out/stress_t.log
