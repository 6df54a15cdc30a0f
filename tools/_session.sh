mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "kats or adversarial or heads or shapes or packed or counter_planes or short_reads_2mb or long_reads_indel or deep" > gpurun_out/r4z_tests.txt 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/r4z_tests.txt
timeout -k 10 500 python bench.py > gpurun_out/r4z_bench.json 2> gpurun_out/r4z_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4z_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", d["value"], "ms", d["ms_per_step"], "frac", r["frac"], "layout_frac", r.get("layout_frac"), "traffic", r.get("traffic"), "layout", r.get("layout_bytes_per_launch"))
print({k: v for k, v in r.items() if k in ("achieved", "kernel", "kernel_ms", "kernels")})
for k in ("end_to_end_first_pass_s", "end_to_end_next_pass_s"):
    print(k, d.get(k) or d.get("config", {}).get(k))
PY
