# kernel trace + stats of the default bench command (no counter passes): bash tools/prof_bench.sh
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf $O/prof_r
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r -o r -- python3 $R/bench.py --no-cpu-baseline > $O/prof_r.log 2>&1
tail -1 $O/prof_r.log | cut -c1-200
echo done
