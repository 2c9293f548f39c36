# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, one rocprofv3 --pmc pass each) of the kernels bench.py's roofline
# objects describe: the dominant convolution, the InstanceNorm apply, the weight gradient.
set -e
P=${1:-3}      # precision code of the convolution and the weight gradient: 2 = BF16X3, 3 = F16X2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
for w in conv in wgrad; do rm -rf $O/pmc_${w}_f $O/pmc_${w}_w; done
timeout -s KILL 90 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/pmc_conv_f -o r -- python3 $R/tools/conv_once.py $P 0 32 > $O/pmc_conv_f.log 2>&1
timeout -s KILL 90 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/pmc_conv_w -o r -- python3 $R/tools/conv_once.py $P 0 32 > $O/pmc_conv_w.log 2>&1
timeout -s KILL 90 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/pmc_in_f -o r -- python3 $R/tools/in_once.py > $O/pmc_in_f.log 2>&1
timeout -s KILL 90 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/pmc_in_w -o r -- python3 $R/tools/in_once.py > $O/pmc_in_w.log 2>&1
timeout -s KILL 90 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/pmc_wgrad_f -o r -- python3 $R/tools/wgrad_once.py $P 32 > $O/pmc_wgrad_f.log 2>&1
timeout -s KILL 90 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/pmc_wgrad_w -o r -- python3 $R/tools/wgrad_once.py $P 32 > $O/pmc_wgrad_w.log 2>&1
echo done
# afterwards, in the repository (reads gpurun_out/, writes profiles/r2_traffic.json and the trimmed CSVs):
#   [TRAFFIC_JSON=profiles/r2_traffic.json] python3 tools/pmc_traffic.py r2 resblock_conv_x3_b32=gpurun_out/pmc_conv_f,gpurun_out/pmc_conv_w \
#       in_apply=gpurun_out/pmc_in_f,gpurun_out/pmc_in_w resblock_wgrad_x3_b32=gpurun_out/pmc_wgrad_f,gpurun_out/pmc_wgrad_w
