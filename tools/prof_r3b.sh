# Final-tree record of round 3's second session: full GPU test suite, the driver's bench command, 40-step traces
# (one stream / overlapped), bench trace, inference trace.   bash tools/prof_r3b.sh   (inside gpurun)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/r3b_gputests.log 2>&1
tail -2 $O/r3b_gputests.log
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r3b_bench.json 2> $O/r3b_bench.err
export FCD_PROF_PREC=fp32x2
FCD_SIDE_STREAM=0 bash tools/prof_steps.sh
python3 tools/trace_shapes.py $O/prof_s/s_kernel_trace.csv 40 100 > $O/r3b_fp32x2_40steps_onestream_by_launch_shape.txt
cp $O/prof_s/s_kernel_stats.csv $O/r3b_fp32x2_40steps_onestream_kernel_stats.csv
cp $O/prof_s.log $O/r3b_onestream.log
bash tools/prof_steps.sh
python3 tools/trace_shapes.py $O/prof_s/s_kernel_trace.csv 40 100 > $O/r3b_fp32x2_40steps_by_launch_shape.txt
cp $O/prof_s/s_kernel_stats.csv $O/r3b_fp32x2_40steps_kernel_stats.csv
cp $O/prof_s.log $O/r3b_overlap.log
bash tools/prof_bench.sh
cp $O/prof_r/r_kernel_stats.csv $O/r3b_fp32x2_bench_kernel_stats.csv
bash tools/prof_infer.sh fp32x2 > $O/r3b_inference_fp32x2_by_launch_shape.txt 2>&1
echo all done
