#!/usr/bin/env python3
"""Compile pygsp_amd/csrc/gspx.hip with -save-temps and print per-kernel register / occupancy
figures from the gfx950 assembly; optionally dump one kernel's ISA.  CPU-only (cross compile).

usage: tools/kernel_report.py [substring-of-kernel-name-to-dump]
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "pygsp_amd", "csrc", "gspx.hip")


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else None
    tmp = tempfile.mkdtemp(prefix="gspx_rep_")
    subprocess.check_call(
        ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         SRC, "-o", os.path.join(tmp, "t.so"), "-save-temps"], cwd=tmp)
    asm = open(os.path.join(tmp, "gspx-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    # metadata blocks
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?.*\.sgpr_count:\s+(\d+)(?:.*\n)*?.*\.vgpr_count:\s+(\d+)", asm):
        pass
    kern = re.findall(r"^(_Z\S+):\s*; @", asm, flags=re.M)
    for k in kern:
        body = asm[asm.index("\n" + k + ":"):]
        end = body.index("s_endpgm")
        tail = body[end:end + 6000]
        def grab(key):
            mm = re.search(r"; %s: (\d+)" % key, tail)
            return int(mm.group(1)) if mm else -1
        demangled = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        short = re.sub(r"gspx::StepArgs<\w+>", "Args", demangled)[:90]
        print(f"{short:90s} vgpr={grab('NumVgprs'):3d} sgpr={grab('NumSgprs'):3d} "
              f"occ={grab('Occupancy'):2d} scratch={grab('ScratchSize'):3d} lds={grab('LDSByteSize')}")
        if want and want in demangled:
            isa = body[:end + 8]
            lines = [l for l in isa.split("\n") if not l.strip().startswith(";")]
            open(os.path.join(tmp, "dump.s"), "w").write("\n".join(lines))
            print("   -> ISA dumped to", os.path.join(tmp, "dump.s"))


if __name__ == "__main__":
    main()
