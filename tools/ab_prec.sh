#!/bin/bash
# bench.py --quick in both split precisions, then a 40-step kernel trace of FCD_PROF_PREC (default fp32x2).
set -e
mkdir -p gpurun_out
for p in fp32x3 fp32x2 fp32x3 fp32x2; do
  timeout -k 10 200 python3 bench.py --quick --precision $p > gpurun_out/b_$p.log 2>&1
  python3 - "$p" <<'PY'
import json, sys
p = sys.argv[1]
l = [x for x in open("gpurun_out/b_%s.log" % p) if x.startswith("{")][-1]
j = json.loads(l)
print(p, "images/s", j["value"], "ms", j["ms_per_step"])
PY
done
FCD_PROF_PREC=${FCD_PROF_PREC:-fp32x2} timeout -k 10 300 bash tools/prof_steps.sh > gpurun_out/ps.log 2>&1
python3 tools/trace_shapes.py gpurun_out/prof_s/s_kernel_trace.csv 40 14
