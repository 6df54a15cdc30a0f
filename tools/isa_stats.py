#!/usr/bin/env python3
"""Per-kernel figures of the gfx950 code object inside a built library: VGPRs, SGPRs, LDS, scratch, spills and the
count of a few instruction classes (v_bitop3, v_lerp, ds_add, global_load, mfma ...).  CPU only (llvm-objdump /
llvm-readelf on the extracted code object).

    python tools/isa_stats.py [lib.so] [kernel-name-substring ...]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def extract(lib, out):
    data = open(lib, "rb").read()
    # the clang offload bundle lives in .hip_fatbin; the gfx950 ELF inside starts with \x7fELF after the host ELF header
    idx = [m.start() for m in re.finditer(b"\x7fELF", data)]
    for i in idx[1:]:
        # e_machine EM_AMDGPU = 224
        if int.from_bytes(data[i + 18:i + 20], "little") == 224:
            # size: section header offset + shnum * shentsize
            shoff = int.from_bytes(data[i + 40:i + 48], "little")
            shentsize = int.from_bytes(data[i + 58:i + 60], "little")
            shnum = int.from_bytes(data[i + 60:i + 62], "little")
            open(out, "wb").write(data[i:i + shoff + shnum * shentsize])
            return True
    return False


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(
        os.path.dirname(os.path.abspath(__file__)), "..", "decodingustools_amd", "lib", "libcallable_hip.so")
    pats = [a for a in sys.argv[1:] if not a.endswith(".so")]
    with tempfile.TemporaryDirectory() as d:
        co = os.path.join(d, "k.co")
        if not extract(lib, co):
            sys.exit("no gfx950 code object found in " + lib)
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        dis = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
    # metadata: one YAML list entry per kernel under amdhsa.kernels (keys in alphabetical order, .agpr_count first;
    # the entries of .args are indented deeper)
    kern = {}
    block = None
    indent = None
    for line in notes.splitlines():
        m = re.match(r"^(\s*)- \.agpr_count:\s+(\S+)", line)
        if m and (indent is None or len(m.group(1)) == indent):
            indent = len(m.group(1))
            block = {".agpr_count": m.group(2)}
            continue
        if block is None:
            continue
        mm = re.match(r"^\s+(\.[a-z_]+):\s+(\S+)", line)
        if not mm or len(line) - len(line.lstrip()) != indent + 2:
            continue
        block[mm.group(1)] = mm.group(2)
        if mm.group(1) == ".vgpr_spill_count":
            kern[block.get(".name")] = block
            block = None
    # instruction counts per function
    counts = {}
    fn = None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            fn = m.group(1)
            counts[fn] = {}
            continue
        if fn is None:
            continue
        t = line.split()
        if not t:
            continue
        op = t[0]
        c = counts[fn]
        c["total"] = c.get("total", 0) + 1
        for cls in ("v_bitop3", "v_lerp", "v_dot4", "ds_add", "ds_read", "ds_write", "ds_load", "ds_store", "global_load", "global_store",
                    "global_atomic", "buffer_load", "s_waitcnt", "s_barrier", "v_mfma", "scratch_", "v_bcnt", "_dpp"):
            if cls in op or (cls == "_dpp" and "dpp" in line):
                c[cls] = c.get(cls, 0) + 1
    for name in sorted(counts):
        if pats and not any(p in name for p in pats):
            continue
        if name not in kern:
            continue
        k = kern.get(name, {})
        c = counts[name]
        print(f"{name}\n    vgpr {k.get('.vgpr_count')} agpr {k.get('.agpr_count')} sgpr {k.get('.sgpr_count')} lds {k.get('.group_segment_fixed_size')} "
              f"scratch {k.get('.private_segment_fixed_size')} vgpr_spill {k.get('.vgpr_spill_count')} sgpr_spill {k.get('.sgpr_spill_count')}")
        print("    " + " ".join(f"{a}={b}" for a, b in sorted(c.items())))


if __name__ == "__main__":
    main()
