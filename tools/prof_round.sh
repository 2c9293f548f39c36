set -e
P=${1:-3}      # precision code of the dominant convolution's counter passes: 2 = BF16X3, 3 = F16X2 (bench.py runs its default)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf $O/prof_r $O/pmc_r1 $O/pmc_r2 $O/pmc_r3
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r -o r -- python3 $R/bench.py --no-cpu-baseline > $O/prof_r.log 2>&1
timeout -s KILL 90 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/pmc_r1 -o r -- python3 $R/tools/conv_once.py $P 0 32 > $O/pmc_r1.log 2>&1
timeout -s KILL 90 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/pmc_r2 -o r -- python3 $R/tools/conv_once.py $P 0 32 > $O/pmc_r2.log 2>&1
timeout -s KILL 90 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/pmc_r3 -o r -- python3 $R/tools/conv_once.py $P 0 32 > $O/pmc_r3.log 2>&1
echo done
