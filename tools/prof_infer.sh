# kernel trace of the inference leg: bash tools/prof_infer.sh [precision]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf $O/prof_i
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_i -o i -- python3 $R/tools/infer_probe.py ${1:-fp32x2} 10 > $O/prof_i.log 2>&1
tail -1 $O/prof_i.log
python3 $R/tools/trace_shapes.py $O/prof_i/i_kernel_trace.csv 13 40
