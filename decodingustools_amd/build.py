"""Builds libcallable_hip.so (gfx950 device code + host mirror) in-tree with hipcc.

The shared library is the product's only native artefact; it lands in
decodingustools_amd/lib/ so that it travels with the source tree (it is git-ignored).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcallable_hip.so")
SOURCES = [os.path.join(CSRC, "callable_loci.hip"), os.path.join(CSRC, "qual_pack.cpp"), os.path.join(CSRC, "host_coverage.cpp"),
           os.path.join(CSRC, "bam_io.cpp"), os.path.join(CSRC, "report.cpp"),
           os.path.join(CSRC, "haplogroup.cpp")]
CLI = os.path.join(LIBDIR, "dut-coverage")
CLI_SRC = os.path.join(CSRC, "coverage_main.cpp")
HEADERS = [os.path.join(CSRC, "kernels.hip.h"), os.path.join(CSRC, "host_parallel.h"),
           os.path.join(CSRC, "qual_pack.h"), os.path.join(CSRC, "pass_rows.h"),
           os.path.join(HERE, "..", "include", "callable_loci.h"),
           os.path.join(HERE, "..", "include", "dut_coverage.h"),
           os.path.join(HERE, "..", "include", "dut_bam.h"),
           os.path.join(HERE, "..", "include", "dut_report.h"),
           os.path.join(HERE, "..", "include", "dut_haplogroup.h"), CLI_SRC]


def _stale():
    if not os.path.exists(LIB) or not os.path.exists(CLI):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS)


TUNING_LIB = os.path.join(LIBDIR, "libcallable_hip_tuning.so")


def build_tuning(verbose=False):
    """The same library with -DCL_TUNING: the timing-experiment hooks of tools/ (CL_ABLATE skips kernel phases,
    CL_FORCE_LONG picks the form of k_pileup by hand).  Never loaded by the product: tools select it through
    DUT_CALLABLE_LIB."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(LIBDIR, exist_ok=True)
    tmp = f".tmp{os.getpid()}"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DCL_TUNING",
           "-Wall", "-Wno-unused-function"] + SOURCES + ["-lz", "-ldl", "-o", TUNING_LIB + tmp]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    os.replace(TUNING_LIB + tmp, TUNING_LIB)
    return TUNING_LIB


def build(force=False, verbose=False):
    """Compile for gfx950 (cross-compiles without a GPU). Returns the library path."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libcallable_hip.so")
    os.makedirs(LIBDIR, exist_ok=True)
    tmp = f".tmp{os.getpid()}"                    # several ranks may find the library stale at once
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function"] + SOURCES + ["-lz", "-ldl", "-o", LIB + tmp]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    os.replace(LIB + tmp, LIB)
    # the `coverage` command line tool, linked against the library beside it
    cmd = [hipcc, "-O2", "-std=c++17", CLI_SRC, "-L" + LIBDIR, "-lcallable_hip", "-Wl,-rpath,$ORIGIN", "-o", CLI + tmp]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building dut-coverage failed:\n" + r.stdout + r.stderr)
    os.replace(CLI + tmp, CLI)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--tuning" in sys.argv:
        print(build_tuning(verbose=True))
