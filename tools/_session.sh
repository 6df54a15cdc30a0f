mkdir -p gpurun_out
L=decodingustools_amd/lib
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "kats or adversarial or shapes or counter_planes or short_reads_2mb or long_reads_indel or deep or rows or deeper" > gpurun_out/r4z_tests.txt 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/r4z_tests.txt
timeout -k 10 800 python tools/ab_kernel.py --rounds 4 --steps 100 $L/libcallable_hip_base.so $L/libcallable_hip.so > gpurun_out/r4aa_ab.txt 2>&1
grep "^==" gpurun_out/r4aa_ab.txt
grep DIFFERS gpurun_out/r4aa_ab.txt | head -3
