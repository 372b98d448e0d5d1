#!/usr/bin/env python3
"""VGPRs, spilled dwords and scratch bytes of every k_eval_forest instantiation (compiles rdf_hip.hip to ISA text).
usage: python3 tools/kernel_regs.py [substring of the demangled template arguments]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "3d-beats_amd", "csrc", "rdf_hip.hip")


def main(filt=""):
    d = tempfile.mkdtemp(prefix="rdf_isa_")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-fast-math", "-ffp-contract=off",
                           "--save-temps", "-c", "-o", os.path.join(d, "rdf.o"), SRC], cwd=d, stderr=subprocess.DEVNULL)
    s = open(os.path.join(d, "rdf_hip-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    md = s[s.find("amdhsa.kernels"):]
    for b in md.split("  - .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", b).group(1)
        if "k_eval_forest" not in name:
            continue
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        short = dem[dem.find("k_eval_forest"):dem.find(">(") + 1]
        if filt not in short:
            continue
        i = s.find("\n" + name + ":")
        body = s[i:s.find(".end_amdhsa_kernel", i)]
        def field(key):
            return re.search(key + r":\s+(\d+)", b).group(1)
        print(f"{short:62s} vgpr {field('.vgpr_count'):>3s} spill {field('.vgpr_spill_count'):>3s} "
              f"scratch {field('.private_segment_fixed_size'):>4s} B  lines {body.count(chr(10)):5d}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "")
