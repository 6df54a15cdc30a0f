mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r4z_tests.txt 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/r4z_tests.txt
timeout -k 10 600 python tools/first_pass.py > gpurun_out/r4ac_first_pass.txt 2>&1
grep "====.*ms" gpurun_out/r4ac_first_pass.txt
awk '/==== engine 1 pass 0$/,/==== engine 1 pass 0:/' gpurun_out/r4ac_first_pass.txt | grep -v "collect\|finish: run\|order\|patched\|ring thread"
