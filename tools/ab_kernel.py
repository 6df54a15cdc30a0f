#!/usr/bin/env python3
"""Same-box A/B of several builds of the library on one workload, the inputs generated ONCE (tooling, not the product):

    python tools/ab_kernel.py [--rounds 3] [--steps 40] [--workload chr21|long] [--length N] libA.so libB.so[:VAR=value,...] ...

(a library may carry environment settings for its children: lib.so:DUT_QUAL_FORM=bytes)

The parent generates the synthetic contig, caches it as .npy files and then starts one child per (round, library) with
DUT_CALLABLE_LIB set; a child pushes the contig through the module API once, checks the BED text against the first
library's (md5), runs `steps` resident steps and prints k_pileup's HIP-event time.  Boxes and processes differ by a
few percent, so variants are compared within one call, interleaved, over several rounds."""
import argparse, hashlib, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(d, steps, workload):
    import bench
    from decodingustools_amd import CallableOptions, CallableProfiler, ContigProfiler, Engine, process_single_contig
    rec, ref, _ = bench.load_records(d, workload)
    opt = CallableOptions()
    bed = os.path.join(tempfile.mkdtemp(), "x.bed")
    with Engine(opt, 0) as eng:
        counter = CallableProfiler(bed)
        st = ContigProfiler("c", ref.shape[0])
        t1 = time.perf_counter()
        process_single_contig(eng, counter, st, opt, 20, rec, ref)
        first_pass = time.perf_counter() - t1
        counter.close()
        lay = eng.contig_layout(); nbytes = eng.contig_bytes()
        md5 = hashlib.md5(open(bed, "rb").read()).hexdigest()
        eng.set_profiling(True)
        for _ in range(5):
            eng.contig_run()
        eng.sync(); eng.reset_kernel_ms()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.contig_run()
        eng.sync()
        dt = (time.perf_counter() - t0) / steps * 1e3
        ms, n = eng.kernel_ms()
        print(json.dumps(dict(ms_step=round(dt, 4), first_pass_s=round(first_pass, 4), form=lay["form"], row_groups=lay["row_groups"], max_groups=lay["max_groups"],
                              device_mb=round(lay["device_bytes"] / 1e6, 1), input_mb=round(nbytes[0] / 1e6, 1), bed_md5=md5, sums=[st.n_covered_bases, st.summed_coverage, st.summed_baseq, st.summed_mapq, st.quality_bases],
                              **{k: round(v / n, 4) for k, v in ms.items() if v})), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--workload", default="chr21")
    ap.add_argument("--length", type=int, default=0)
    ap.add_argument("--child", default=None)
    ap.add_argument("libs", nargs="*")
    a = ap.parse_args()
    if a.child:
        return child(a.child, a.steps, a.workload)
    import bench
    from decodingustools_amd import synth
    d = tempfile.mkdtemp(prefix="ab_inputs_", dir="/tmp")
    if a.workload == "chr21":
        L = a.length or 46_709_983
        seed = synth.seed_for(2, 20)
        rec = synth.short_read_contig(L, 30.0, seed)
    else:
        L = a.length or 57_227_415
        seed = synth.seed_for(3, 23)
        rec = synth.long_read_contig(L, 50.0, seed)
    ref = synth.make_reference(L, seed)
    assert bench.save_records(d, a.workload, rec, ref)
    del rec, ref
    first = None
    res = {l: [] for l in a.libs}
    for r in range(a.rounds):
        for l in a.libs:
            lib, _, sets = l.partition(":")
            env = dict(os.environ, DUT_CALLABLE_LIB=os.path.abspath(lib))
            for kv in filter(None, sets.split(",")):
                k, _, v = kv.partition("=")
                env[k] = v
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", d, "--steps", str(a.steps), "--workload", a.workload],
                               env=env, capture_output=True, text=True, timeout=600)
            if p.returncode != 0:
                print(l, "FAILED", p.stderr[-400:], flush=True)
                continue
            o = json.loads(p.stdout.strip().splitlines()[-1])
            if first is None:
                first = (o["bed_md5"], o["sums"])
            same = (o["bed_md5"], o["sums"]) == first
            res[l].append(o.get("pileup", 0.0))
            print(os.path.basename(l), "pileup_ms", o.get("pileup"), "step_ms", o["ms_step"], "first_pass_s", o.get("first_pass_s"), "form", o.get("form"),
                  "groups", o.get("row_groups"), "max", o.get("max_groups"), "device_MB", o.get("device_mb"), "input_MB", o.get("input_mb"),
                  "same_as_first" if same else "DIFFERS", flush=True)
            if p.stderr.strip() and os.environ.get("DUT_TIMING") == "1":
                print(p.stderr[-6000:], flush=True)
    for l, v in res.items():
        if v:
            print("== %-40s min %.4f  mean %.4f  (%d runs)" % (os.path.basename(l), min(v), sum(v) / len(v), len(v)))
    import shutil
    shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
