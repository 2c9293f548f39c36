set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf $O/prof_s
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s -o s -- python3 $R/tools/busy_probe.py ${FCD_PROF_PREC:-fp32x3} 40 > $O/prof_s.log 2>&1
echo done
