mkdir -p gpurun_out
for hp in 1 0 1 0; do
  echo "== DUT_HUGEPAGES=$hp"
  DUT_HUGEPAGES=$hp timeout -k 10 300 python tools/first_pass.py --engines 2 --passes 2 2>&1 | grep "====.*ms"
  DUT_HUGEPAGES=$hp E2E_THREADS=16 E2E_REPS=3 E2E_QUIET=1 E2E_TEARDOWN=0 timeout -k 10 300 python tools/e2e_bench.py 2>&1 | grep -- "--- DUT\|since the last decode\|BAM read"
done
