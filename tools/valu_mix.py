#!/usr/bin/env python3
"""What the level loop of the forest kernel issues per level, by issue class.

tools/ubench_valu.hip measured two VALU issue rates on gfx950 with several waves per SIMD: ~2.35 cycles per wave64
instruction for v_fma_f32 / v_add_f32 / v_mul_f32 / v_add_u32 / v_mov_b32, ~4.2 cycles for everything else the kernel
uses (conversions, v_pk_*_f32, compares, v_cndmask, shifts, v_mad_u32_u24, v_perm_b32, ...).  This script compiles
rdf_hip.hip to ISA text, takes the innermost loop of one instantiation of k_eval_forest (the level loop), leaves out
the blocks of the IEEE-divide path (they run only for nodes flagged kFlagExact) and counts the instructions of each
class.  Its `half_rate_share` is tools/roofline.py's VALU_HALF_RATE_SHARE.

    python3 tools/valu_mix.py [mangled-name substring, default: the packed 512-thread 4-tree kernel of the bench batch]
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "3d-beats_amd", "csrc", "rdf_hip.hip")
FULL_RATE = {"v_fma_f32", "v_fmac_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_add_u32", "v_sub_u32",
             "v_subrev_u32", "v_mov_b32"}
# blocks of the IEEE-divide path: the divide itself, the fetch of the exact record (64-bit addressed), the mode switches
EXACT_PATH_MARKS = ("v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32", "v_rcp_f32", "v_lshl_add_u64", "s_setreg_imm32_b32")


def isa_text():
    out = os.path.join(tempfile.gettempdir(), "rdf_hip_gfx950.s")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(SRC):
        subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-fast-math", "-ffp-contract=off",
                               "-S", "--cuda-device-only", "-o", out, SRC], stderr=subprocess.DEVNULL)
    return open(out).read().splitlines()


def main(sub):
    lines = isa_text()
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and sub in l and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end]
    depth = max(int(m.group(1)) for l in body for m in [re.search(r"Depth=(\d+)", l)] if m)
    # split into basic blocks; a block's annotation says which loop it belongs to
    blocks, cur = [], None
    for l in body:
        if re.match(r"^(\.LBB\S+:|; %bb\.\d+:)", l):
            cur = {"head": l, "ins": []}
            blocks.append(cur)
        elif cur is not None and l.startswith("\t") and not l.strip().startswith((";", ".")):
            cur["ins"].append(l.split()[0])
    inner = [b for b in blocks if f"Depth={depth}" in b["head"]]
    kept = [b for b in inner if not any(i in EXACT_PATH_MARKS for i in b["ins"])]
    count = collections.Counter(i.replace("_e32", "").replace("_e64", "").replace("_sdwa", "") for b in kept for i in b["ins"])
    valu = {k: v for k, v in count.items() if k.startswith("v_")}
    full = sum(v for k, v in valu.items() if k in FULL_RATE)
    half = sum(valu.values()) - full
    out = {"kernel": lines[start].split(":")[0], "loop_depth": depth, "blocks_in_level_loop": len(inner),
           "blocks_counted": len(kept), "valu_full_rate": full, "valu_half_rate": half,
           "half_rate_share": round(half / max(1, full + half), 3),
           "salu": sum(v for k, v in count.items() if k.startswith("s_") and not k.startswith(("s_waitcnt", "s_nop"))),
           "vmem": sum(v for k, v in count.items() if k.startswith(("global_", "buffer_", "flat_"))),
           "lds": sum(v for k, v in count.items() if k.startswith("ds_")),
           "valu_by_opcode": dict(sorted(valu.items(), key=lambda kv: -kv[1]))}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "k_eval_forestILi512ELb1ELi4ELb0ELi4ELb0ELi1ELb0E")
