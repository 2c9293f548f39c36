set -e
P=${1:-fp32x2}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf $O/prof_stress
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stress -o st -- python3 $R/tools/stress_probe.py $P 32 pair > $O/prof_stress.log 2>&1
python3 $R/tools/profile_summary.py $O/prof_stress/st_kernel_stats.csv "stress config (image 256, batch 32, conv_dim 128), 6 $P steps" > $O/prof_stress_summary.md
echo done
