#!/bin/bash
# end-to-end pass of the chr21 contig for a few host thread settings (tooling; run on the GPU box)
python3 -c "import os; print('cpus', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))"
cat /sys/fs/cgroup/cpu.max 2>/dev/null
for S in 0 1; do
for T in 12 16; do
  for C in 4 8; do
    echo "== DUT_PIN_SPIN=$S DUT_THREADS=$T DUT_COPY_THREADS=$C"
    DUT_PIN_SPIN=$S DUT_THREADS=$T DUT_COPY_THREADS=$C python3 bench.py --no-secondary --no-traffic --cpu-sample 0 --min-time 0 --max-blocks 1 2>&1 >/dev/null | grep "end-to-end"
  done
done
done
