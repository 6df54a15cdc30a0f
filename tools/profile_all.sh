#!/bin/bash
# GPU box: the un-profiled bench line, then one rocprofv3 pass per counter group (never combined).
mkdir -p gpurun_out
timeout -k 10 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -c 1500 gpurun_out/bench_final.json
tools/rocprof_pass.sh trace --kernel-trace --stats > /dev/null && echo trace ok
tools/rocprof_pass.sh fetch --pmc FETCH_SIZE > /dev/null && echo fetch ok
tools/rocprof_pass.sh write --pmc WRITE_SIZE > /dev/null && echo write ok
tools/rocprof_pass.sh sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS > /dev/null && echo sq1 ok
tools/rocprof_pass.sh sq2 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE > /dev/null && echo sq2 ok
