set -e
P=${1:-2}      # precision code: 2 = BF16X3, 3 = F16X2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf $O/pmc_a1 $O/pmc_a2
timeout -s KILL 90 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/pmc_a1 -o r -- python3 $R/tools/conv_once.py $P 0 32 > $O/pmc_a1.log 2>&1
timeout -s KILL 90 rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT -d $O/pmc_a2 -o r -- python3 $R/tools/conv_once.py $P 0 32 > $O/pmc_a2.log 2>&1
cd $R && timeout -k 10 120 python3 tools/clock_probe.py $P 2>&1 | grep -v "^$" | grep "sclk\|us per" | cut -c1-120
