#!/bin/bash
# kernel trace + SQ counters of tools/bench_dbias.py (the in-kernel bias gradient at config 4's size)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pd
SQ1="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD"
SQ2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA"
SQ3="SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE"
SQ4="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum"
rocprofv3 --kernel-trace --stats -d /tmp/pd/kt -o kt -- python $R/tools/bench_dbias.py > /tmp/pd/kt.log 2>&1
rocprofv3 --pmc $SQ1 -d /tmp/pd/p1 -o pmc -- python $R/tools/bench_dbias.py > /dev/null 2>&1
rocprofv3 --pmc $SQ2 -d /tmp/pd/p2 -o pmc -- python $R/tools/bench_dbias.py > /dev/null 2>&1
rocprofv3 --pmc $SQ3 -d /tmp/pd/p3 -o pmc -- python $R/tools/bench_dbias.py > /dev/null 2>&1
rocprofv3 --pmc $SQ4 -d /tmp/pd/p4 -o pmc -- python $R/tools/bench_dbias.py > /dev/null 2>&1
python3 $R/tools/pmc_summary.py /tmp/pd dbias | sed 's/.*\] //' | cut -c1-150
