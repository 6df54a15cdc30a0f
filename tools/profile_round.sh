#!/bin/bash
# GPU box: the round's evidence in one call -- kernel traces (--kernel-trace --stats) of the headline workload, of the
# long-read shape at FULL chrY size and of the site pileup, then the PMC groups of the headline workload, each in its own
# pass.  Output under gpurun_out/prof_*; tools/summarize_profiles.py <tag> condenses the headline passes.
mkdir -p gpurun_out
tools/rocprof_pass.sh trace --kernel-trace --stats > /dev/null && echo "trace ok"
KB_LONG_LEN=${KB_LONG_LEN:-57227415} tools/profile_secondary.sh 2>&1 | tail -24
tools/rocprof_pass.sh fetch --pmc FETCH_SIZE > /dev/null && echo "fetch ok"
tools/rocprof_pass.sh write --pmc WRITE_SIZE > /dev/null && echo "write ok"
tools/rocprof_pass.sh sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS > /dev/null && echo "sq1 ok"
tools/rocprof_pass.sh sq2 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE > /dev/null && echo "sq2 ok"
tools/rocprof_pass.sh sq3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA > /dev/null && echo "sq3 ok"
