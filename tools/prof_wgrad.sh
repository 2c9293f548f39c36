set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf $O/pmc_w1 $O/pmc_w2 $O/pmc_w3 $O/pmc_w4
timeout -s KILL 90 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/pmc_w1 -o w -- python3 $R/tools/wgrad_once.py ${1:-3} 32 > $O/pmc_w1.log 2>&1
timeout -s KILL 90 rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT -d $O/pmc_w2 -o w -- python3 $R/tools/wgrad_once.py ${1:-3} 32 > $O/pmc_w2.log 2>&1
timeout -s KILL 90 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/pmc_w3 -o w -- python3 $R/tools/wgrad_once.py ${1:-3} 32 > $O/pmc_w3.log 2>&1
timeout -s KILL 90 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/pmc_w4 -o w -- python3 $R/tools/wgrad_once.py ${1:-3} 32 > $O/pmc_w4.log 2>&1
echo done
